// Fixed-base window tables for the generator set of the reference
// (`gs`, `hs`, `h`, `gen`; /root/reference/Bulletproofs/Utils.hs:14-25).
//
// For every base B and every 8-bit window w the table holds d * 2^(8w) * B for
// d = 1..128 as affine points, so k*B costs one mixed add per window of the
// signed-digit form of k (digits in [-127, 128], 33 windows, least significant
// first; the 33rd window only ever holds the final carry).
#pragma once
#include "ec.h"

namespace bp {

constexpr int FB_WBITS = 8;
constexpr int FB_WINDOWS = 33;
constexpr int FB_ENTRIES = 128;     // magnitudes 1..128
constexpr int FB_DWORDS = 9;        // digit bytes packed 4 per u32: ceil(33/4)

// base numbering inside the table for a context with `ng` generator pairs
BP_HD uint32_t fb_base_g(uint32_t i) { return i; }
BP_HD uint32_t fb_base_h(uint32_t ng, uint32_t i) { return ng + i; }
BP_HD uint32_t fb_base_G(uint32_t ng) { return 2 * ng; }      // curve generator `gen`
BP_HD uint32_t fb_base_H(uint32_t ng) { return 2 * ng + 1; }  // `h`
// GS_k = sum_{i < 2^k} gs_i  (RangeProof/Internal.hs:200), k = 0..log2(ng)
BP_HD uint32_t fb_base_gsum(uint32_t ng, uint32_t k) { return 2 * ng + 2 + k; }

// Signed 8-bit digits of a canonical scalar, packed one byte per window:
// byte = digit & 0xFF, where 0x80 means +128 (-128 is never produced).
// out[FB_DWORDS] u32, window w in byte (w & 3) of word (w >> 2).
BP_HD void fb_recode(const u256& k, uint32_t out[FB_DWORDS]) {
  uint32_t carry = 0;
#pragma unroll
  for (int g = 0; g < FB_DWORDS; g++) {
    uint32_t word = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      int w = 4 * g + b;
      uint32_t v = carry;
      if (w < 32) v += (k.v[w >> 2] >> (8 * (w & 3))) & 0xFFu;
      if (v > 128) {
        carry = 1;
        v = (v - 256) & 0xFFu;  // negative digit, two's complement byte (v = 256 -> 0)
      } else {
        carry = 0;
      }
      word |= v << (8 * b);
    }
    out[g] = word;
  }
}

}  // namespace bp
