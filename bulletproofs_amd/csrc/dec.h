// 256-bit integer -> decimal digits on the device.
//
// The reference's Fiat-Shamir transcript hashes the decimal `show` text of points and scalars
// (`pointToBS = show`, /root/reference/Bulletproofs/Utils.hs:32-33,145-174).  The hashing stays on the
// host (transcript.h); what this file moves is the FORMATTING of numbers that already live on the GPU:
// the kernels hand the host the digits of a coordinate next to its 32 bytes, so that the host's part
// of a challenge is concatenation + SHA-256.  Measured on the batch verifier, the 256-bit -> decimal
// conversions were two thirds of the host's time per round.
//
// Record format (80 bytes per number, DEC_REC): byte 0 = n, the number of digits (1..78, no leading
// zeros; zero is "0"); the digits are ASCII, right-aligned: bytes [80 - n, 80).  Bytes in between are
// '0' padding (byte 1 is 0).  The host appends  rec + 80 - n, n bytes.
#pragma once
#include "fe.h"

namespace bp {

constexpr int DEC_REC = 80;        // bytes per record
constexpr int DEC_REC_WORDS = 20;  // dwords per record

// floor(t / 10^9) and t mod 10^9 for t < 2^62, by multiplication with M = floor(2^93 / 10^9) + 1:
// M * 10^9 - 2^93 < 10^9 < 2^(93 - 62), so the quotient is exact on the whole range.
BP_HD uint32_t dec_divmod_1e9(uint64_t t, uint32_t& rem) {
  const uint64_t M = 9903520314283042200ULL;
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t hi = __umul64hi(t, M);
#else
  uint64_t hi = (uint64_t)(((unsigned __int128)t * M) >> 64);
#endif
  uint32_t q = (uint32_t)(hi >> 29);
  rem = (uint32_t)(t - (uint64_t)q * 1000000000u);
  return q;
}

// number of decimal digits of c, 1 <= c < 10^9
BP_HD uint32_t dec_ndigits(uint32_t c) {
  return 1u + (c >= 10u) + (c >= 100u) + (c >= 1000u) + (c >= 10000u) + (c >= 100000u) + (c >= 1000000u) +
         (c >= 10000000u) + (c >= 100000000u);
}

// v -> record words w[0..20) (little-endian dwords of the 80-byte record)
BP_HD void u256_to_dec_record(const u256& v, uint32_t (&w)[DEC_REC_WORDS]) {
  // base-10^9 chunks, little-endian, by Horner over the limbs from the top: after k limbs the value is
  // below 2^(32k), which needs at most k + 1 chunks
  uint32_t c[9];
#pragma unroll
  for (int j = 0; j < 9; j++) c[j] = 0;
#pragma unroll
  for (int k = 1; k <= 8; k++) {
    uint32_t carry = v.v[8 - k];
#pragma unroll
    for (int j = 0; j < 9; j++) {
      if (j <= k) {
        uint64_t t = ((uint64_t)c[j] << 32) | carry;  // c[j] < 10^9: t < 2^62
        carry = dec_divmod_1e9(t, c[j]);
      }
    }
  }
  // 81 digit positions, most significant first: chunk j covers positions 9 (8 - j) .. 9 (8 - j) + 8; the value is
  // below 10^78, so positions 0..2 are zero and position P goes to byte P - 1 (bytes 2..79 hold 78 digits)
#pragma unroll
  for (int i = 0; i < DEC_REC_WORDS; i++) w[i] = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    uint32_t x = c[j];
#pragma unroll
    for (int d = 8; d >= 0; d--) {
      uint32_t q = (uint32_t)(((uint64_t)x * 0xCCCCCCCDull) >> 35);  // x / 10
      uint32_t digit = x - q * 10u;
      x = q;
      const int pos = 9 * (8 - j) + d;
      if (pos >= 3) {
        const int b = pos - 1;
        w[b >> 2] |= (digit + 48u) << (8 * (b & 3));
      }
    }
  }
  // digit count: position of the top non-zero chunk, digits inside it
  uint32_t top = c[0], topj = 0;
#pragma unroll
  for (int j = 1; j < 9; j++) {
    bool nz = c[j] != 0;
    top = nz ? c[j] : top;
    topj = nz ? (uint32_t)j : topj;
  }
  uint32_t n = top == 0 ? 1u : 9u * topj + dec_ndigits(top);
  w[0] = (w[0] & 0xFFFF0000u) | n;  // byte 0 = n, byte 1 = 0
}

}  // namespace bp
