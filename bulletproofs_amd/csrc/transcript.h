// Host-side Fiat-Shamir transcript of the reference, kept on the CPU as
// BASELINE.json's north_star asks:
//   oracle = fromInteger . integerDigest . sha256   (/root/reference/Bulletproofs/Utils.hs:29-30)
//   pointToBS = show                                 (Utils.hs:32-33)
//   shamirY / shamirZ / shamirX / shamirX' / shamirU (Utils.hs:145-174)
// The hashed bytes are the Haskell `show` text of points and scalars (decimal),
// so this file is: 256-bit -> decimal, the `show` formats, SHA-256 (SHA-NI when
// the CPU has it), and the reduction of the digest into Fr / Fq.
//
// PARITY NOTE: the `show` format is defined by the un-vendored
// packages elliptic-curve-0.3.0 / galois-field-1.0.0.  Both candidates are
// implemented; SHOW_DERIVED ("A (P x) (P y)", "P s") is the default.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#endif

namespace bp {
namespace host {

enum ShowStyle { SHOW_DERIVED = 0, SHOW_PLAIN = 1 };

// ---- SHA-256 ---------------------------------------------------------------------------
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
    0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
    0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
    0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static inline void sha256_blocks_portable(uint32_t st[8], const uint8_t* p, size_t nblk) {
  for (; nblk; nblk--, p += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
      w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
      uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = h + S1 + ch + SHA_K[i] + w[i];
      uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
  }
}

#if defined(__x86_64__)
#define BP_SHA_QROUND(M, KOFF)                                               \
  msg = _mm_add_epi32(M, _mm_loadu_si128((const __m128i*)&SHA_K[KOFF]));     \
  s1 = _mm_sha256rnds2_epu32(s1, s0, msg);                                   \
  msg = _mm_shuffle_epi32(msg, 0x0E);                                        \
  s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
#define BP_SHA_SCHED(W0, W1, W2, W3) \
  W0 = _mm_sha256msg2_epu32(_mm_add_epi32(_mm_sha256msg1_epu32(W0, W1), _mm_alignr_epi8(W3, W2, 4)), W3);

__attribute__((target("sha,sse4.1,ssse3"))) static inline void sha256_blocks_shani(uint32_t st[8], const uint8_t* p,
                                                                                    size_t nblk) {
  const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
  __m128i tmp = _mm_loadu_si128((const __m128i*)&st[0]);     // DCBA
  __m128i s1 = _mm_loadu_si128((const __m128i*)&st[4]);      // HGFE
  tmp = _mm_shuffle_epi32(tmp, 0xB1);                        // CDAB
  s1 = _mm_shuffle_epi32(s1, 0x1B);                          // EFGH
  __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                  // ABEF
  s1 = _mm_blend_epi16(s1, tmp, 0xF0);                       // CDGH
  for (; nblk; nblk--, p += 64) {
    __m128i abef = s0, cdgh = s1, msg;
    __m128i m0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 0)), MASK);
    __m128i m1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16)), MASK);
    __m128i m2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 32)), MASK);
    __m128i m3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 48)), MASK);
    BP_SHA_QROUND(m0, 0)  BP_SHA_SCHED(m0, m1, m2, m3)
    BP_SHA_QROUND(m1, 4)  BP_SHA_SCHED(m1, m2, m3, m0)
    BP_SHA_QROUND(m2, 8)  BP_SHA_SCHED(m2, m3, m0, m1)
    BP_SHA_QROUND(m3, 12) BP_SHA_SCHED(m3, m0, m1, m2)
    BP_SHA_QROUND(m0, 16) BP_SHA_SCHED(m0, m1, m2, m3)
    BP_SHA_QROUND(m1, 20) BP_SHA_SCHED(m1, m2, m3, m0)
    BP_SHA_QROUND(m2, 24) BP_SHA_SCHED(m2, m3, m0, m1)
    BP_SHA_QROUND(m3, 28) BP_SHA_SCHED(m3, m0, m1, m2)
    BP_SHA_QROUND(m0, 32) BP_SHA_SCHED(m0, m1, m2, m3)
    BP_SHA_QROUND(m1, 36) BP_SHA_SCHED(m1, m2, m3, m0)
    BP_SHA_QROUND(m2, 40) BP_SHA_SCHED(m2, m3, m0, m1)
    BP_SHA_QROUND(m3, 44) BP_SHA_SCHED(m3, m0, m1, m2)
    BP_SHA_QROUND(m0, 48)
    BP_SHA_QROUND(m1, 52)
    BP_SHA_QROUND(m2, 56)
    BP_SHA_QROUND(m3, 60)
    s0 = _mm_add_epi32(s0, abef);
    s1 = _mm_add_epi32(s1, cdgh);
  }
  tmp = _mm_shuffle_epi32(s0, 0x1B);                         // FEBA
  s1 = _mm_shuffle_epi32(s1, 0xB1);                          // DCHG
  s0 = _mm_blend_epi16(tmp, s1, 0xF0);                       // DCBA
  s1 = _mm_alignr_epi8(s1, tmp, 8);                          // HGFE
  _mm_storeu_si128((__m128i*)&st[0], s0);
  _mm_storeu_si128((__m128i*)&st[4], s1);
}
// Two independent messages in lock-step: sha256rnds2 has a multi-cycle latency and a message is one
// dependent chain of 32 of them per block, so a second chain interleaved with it nearly doubles the
// throughput of a core.
#define BP_SHA_LOAD_STATE(ST, S0, S1)                               \
  {                                                                 \
    __m128i t_ = _mm_loadu_si128((const __m128i*)&(ST)[0]);         \
    S1 = _mm_loadu_si128((const __m128i*)&(ST)[4]);                 \
    t_ = _mm_shuffle_epi32(t_, 0xB1);                               \
    S1 = _mm_shuffle_epi32(S1, 0x1B);                               \
    S0 = _mm_alignr_epi8(t_, S1, 8);                                \
    S1 = _mm_blend_epi16(S1, t_, 0xF0);                             \
  }
#define BP_SHA_STORE_STATE(ST, S0, S1)                              \
  {                                                                 \
    __m128i t_ = _mm_shuffle_epi32(S0, 0x1B);                       \
    __m128i u_ = _mm_shuffle_epi32(S1, 0xB1);                       \
    _mm_storeu_si128((__m128i*)&(ST)[0], _mm_blend_epi16(t_, u_, 0xF0)); \
    _mm_storeu_si128((__m128i*)&(ST)[4], _mm_alignr_epi8(u_, t_, 8));    \
  }
#define BP_SHA_QROUND2(MA, MB, KOFF)                                  \
  {                                                                   \
    __m128i k_ = _mm_loadu_si128((const __m128i*)&SHA_K[KOFF]);       \
    __m128i ga = _mm_add_epi32(MA, k_), gb = _mm_add_epi32(MB, k_);   \
    a1 = _mm_sha256rnds2_epu32(a1, a0, ga);                           \
    b1 = _mm_sha256rnds2_epu32(b1, b0, gb);                           \
    ga = _mm_shuffle_epi32(ga, 0x0E);                                 \
    gb = _mm_shuffle_epi32(gb, 0x0E);                                 \
    a0 = _mm_sha256rnds2_epu32(a0, a1, ga);                           \
    b0 = _mm_sha256rnds2_epu32(b0, b1, gb);                           \
  }
__attribute__((target("sha,sse4.1,ssse3"))) static inline void sha256_blocks_shani_x2(uint32_t stA[8], const uint8_t* pa,
                                                                                       uint32_t stB[8], const uint8_t* pb,
                                                                                       size_t nblk) {
  const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
  __m128i a0, a1, b0, b1;
  BP_SHA_LOAD_STATE(stA, a0, a1)
  BP_SHA_LOAD_STATE(stB, b0, b1)
  for (; nblk; nblk--, pa += 64, pb += 64) {
    __m128i sa0 = a0, sa1 = a1, sb0 = b0, sb1 = b1;
    __m128i x0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 0)), MASK);
    __m128i x1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 16)), MASK);
    __m128i x2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 32)), MASK);
    __m128i x3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pa + 48)), MASK);
    __m128i y0 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 0)), MASK);
    __m128i y1 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 16)), MASK);
    __m128i y2 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 32)), MASK);
    __m128i y3 = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(pb + 48)), MASK);
#define BP_SHA_STEP2(W0, W1, W2, W3, V0, V1, V2, V3, KOFF) \
  BP_SHA_QROUND2(W0, V0, KOFF) BP_SHA_SCHED(W0, W1, W2, W3) BP_SHA_SCHED(V0, V1, V2, V3)
    BP_SHA_STEP2(x0, x1, x2, x3, y0, y1, y2, y3, 0)
    BP_SHA_STEP2(x1, x2, x3, x0, y1, y2, y3, y0, 4)
    BP_SHA_STEP2(x2, x3, x0, x1, y2, y3, y0, y1, 8)
    BP_SHA_STEP2(x3, x0, x1, x2, y3, y0, y1, y2, 12)
    BP_SHA_STEP2(x0, x1, x2, x3, y0, y1, y2, y3, 16)
    BP_SHA_STEP2(x1, x2, x3, x0, y1, y2, y3, y0, 20)
    BP_SHA_STEP2(x2, x3, x0, x1, y2, y3, y0, y1, 24)
    BP_SHA_STEP2(x3, x0, x1, x2, y3, y0, y1, y2, 28)
    BP_SHA_STEP2(x0, x1, x2, x3, y0, y1, y2, y3, 32)
    BP_SHA_STEP2(x1, x2, x3, x0, y1, y2, y3, y0, 36)
    BP_SHA_STEP2(x2, x3, x0, x1, y2, y3, y0, y1, 40)
    BP_SHA_STEP2(x3, x0, x1, x2, y3, y0, y1, y2, 44)
#undef BP_SHA_STEP2
    BP_SHA_QROUND2(x0, y0, 48)
    BP_SHA_QROUND2(x1, y1, 52)
    BP_SHA_QROUND2(x2, y2, 56)
    BP_SHA_QROUND2(x3, y3, 60)
    a0 = _mm_add_epi32(a0, sa0);
    a1 = _mm_add_epi32(a1, sa1);
    b0 = _mm_add_epi32(b0, sb0);
    b1 = _mm_add_epi32(b1, sb1);
  }
  BP_SHA_STORE_STATE(stA, a0, a1)
  BP_SHA_STORE_STATE(stB, b0, b1)
}
#undef BP_SHA_QROUND2
#undef BP_SHA_LOAD_STATE
#undef BP_SHA_STORE_STATE
#undef BP_SHA_QROUND
#undef BP_SHA_SCHED
static inline bool cpu_has_shani() {
  static int cached = -1;
  if (cached < 0) {
    unsigned a, b, c, d;
    cached = (__get_cpuid_count(7, 0, &a, &b, &c, &d) && (b & (1u << 29))) ? 1 : 0;
    if (getenv("BP_NO_SHANI")) cached = 0;  // test hook: force the portable path
  }
  return cached == 1;
}
#endif

static inline void sha256_blocks(uint32_t st[8], const uint8_t* p, size_t nblk) {
#if defined(__x86_64__)
  if (cpu_has_shani()) {
    sha256_blocks_shani(st, p, nblk);
    return;
  }
#endif
  sha256_blocks_portable(st, p, nblk);
}

// one-shot digest of a message with >= 72 bytes of writable slack after it (padding is
// written in place; callers build messages in scratch buffers)
static inline void sha256_inplace(uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t total = len;
  msg[total++] = 0x80;
  while ((total & 63) != 56) msg[total++] = 0;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 7; i >= 0; i--) msg[total++] = (uint8_t)(bits >> (8 * i));
  sha256_blocks(st, msg, total / 64);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(st[i] >> 24);
    out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8);
    out[4 * i + 3] = (uint8_t)st[i];
  }
}

// two messages with the same number of blocks
static inline void sha256_blocks_x2(uint32_t stA[8], const uint8_t* pa, uint32_t stB[8], const uint8_t* pb, size_t nblk) {
#if defined(__x86_64__)
  if (cpu_has_shani()) {
    sha256_blocks_shani_x2(stA, pa, stB, pb, nblk);
    return;
  }
#endif
  sha256_blocks_portable(stA, pa, nblk);
  sha256_blocks_portable(stB, pb, nblk);
}

// SHA-256 state after the first `done` bytes (a multiple of 64) of a message: transcripts that share a
// prefix (every message starts with `show _r`; shamirY/Z/X share `show _r <> A <> S`) hash it once
struct ShaMid {
  uint32_t st[8];
  size_t done;
};
static inline ShaMid sha_mid_init() {
  return ShaMid{{0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}, 0};
}
// absorb the full blocks of msg[0, upto) that `mid` has not seen yet
static inline void sha_mid_advance(ShaMid& mid, const uint8_t* msg, size_t upto) {
  size_t full = upto & ~(size_t)63;
  if (full > mid.done) {
    sha256_blocks(mid.st, msg + mid.done, (full - mid.done) / 64);
    mid.done = full;
  }
}
// digest of msg[0, len) given the state after its first mid.done bytes; same slack rule as sha256_inplace
static inline void sha256_resume_inplace(const ShaMid& mid, uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t st[8];
  memcpy(st, mid.st, sizeof st);
  size_t total = len;
  msg[total++] = 0x80;
  while ((total & 63) != 56) msg[total++] = 0;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 7; i >= 0; i--) msg[total++] = (uint8_t)(bits >> (8 * i));
  sha256_blocks(st, msg + mid.done, (total - mid.done) / 64);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(st[i] >> 24);
    out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8);
    out[4 * i + 3] = (uint8_t)st[i];
  }
}

static inline size_t sha256_pad_inplace(uint8_t* msg, size_t len) {
  size_t total = len;
  msg[total++] = 0x80;
  while ((total & 63) != 56) msg[total++] = 0;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 7; i >= 0; i--) msg[total++] = (uint8_t)(bits >> (8 * i));
  return total;
}
static inline void sha256_digest_out(const uint32_t st[8], uint8_t out[32]) {
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(st[i] >> 24);
    out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8);
    out[4 * i + 3] = (uint8_t)st[i];
  }
}
// sha256_resume_inplace for two messages at once (their common blocks run interleaved)
static inline void sha256_resume_inplace_x2(const ShaMid& midA, uint8_t* msgA, size_t lenA, uint8_t outA[32],
                                            const ShaMid& midB, uint8_t* msgB, size_t lenB, uint8_t outB[32]) {
  uint32_t sa[8], sb[8];
  memcpy(sa, midA.st, sizeof sa);
  memcpy(sb, midB.st, sizeof sb);
  size_t na = (sha256_pad_inplace(msgA, lenA) - midA.done) / 64, nb = (sha256_pad_inplace(msgB, lenB) - midB.done) / 64;
  size_t common = na < nb ? na : nb;
  sha256_blocks_x2(sa, msgA + midA.done, sb, msgB + midB.done, common);
  if (na > common) sha256_blocks(sa, msgA + midA.done + 64 * common, na - common);
  if (nb > common) sha256_blocks(sb, msgB + midB.done + 64 * common, nb - common);
  sha256_digest_out(sa, outA);
  sha256_digest_out(sb, outB);
}

// ---- 256-bit little-endian integer -> decimal text ------------------------------------------
// (u1:u0) / 10^19 for u1 < 10^19 by multiplication with a precomputed reciprocal (Moeller-Granlund
// 2-by-1 division; 10^19 has its top bit set, so it is its own normalised divisor): ~10 cycles
// against a 128/64-bit `div`.
static inline uint64_t div_1e19(uint64_t u1, uint64_t u0, uint64_t* rem) {
  const uint64_t D = 10000000000000000000ULL;
  const uint64_t V = 0xd83c94fb6d2ac34aULL;  // floor((2^128 - 1) / D) - 2^64
  unsigned __int128 q = (unsigned __int128)V * u1 + (((unsigned __int128)u1 << 64) | u0);
  uint64_t q1 = (uint64_t)(q >> 64) + 1, q0 = (uint64_t)q;
  uint64_t r = u0 - q1 * D;
  if (r > q0) {
    q1--;
    r += D;
  }
  if (r >= D) {
    q1++;
    r -= D;
  }
  *rem = r;
  return q1;
}

static const char DEC_PAIRS[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960"
    "616263646566676869707172737475767778798081828384858687888990919293949596979899";

// exactly 8 digits of x < 10^8: four independent two-digit lookups
static inline void put8(char* o, uint32_t x) {
  uint32_t hi = x / 10000u, lo = x - hi * 10000u;
  uint32_t a = hi / 100u, b = hi - a * 100u, c = lo / 100u, d = lo - c * 100u;
  memcpy(o, DEC_PAIRS + 2 * a, 2);
  memcpy(o + 2, DEC_PAIRS + 2 * b, 2);
  memcpy(o + 4, DEC_PAIRS + 2 * c, 2);
  memcpy(o + 6, DEC_PAIRS + 2 * d, 2);
}
// exactly 19 digits of c < 10^19: 3 + 8 + 8
static inline void put19(char* o, uint64_t c) {
  uint64_t q = c / 100000000ULL;
  uint32_t e = (uint32_t)(c - q * 100000000ULL);
  uint32_t a = (uint32_t)(q / 100000000ULL);      // < 1000
  uint32_t b = (uint32_t)(q - (uint64_t)a * 100000000ULL);
  uint32_t a1 = a / 100u, a0 = a - a1 * 100u;
  o[0] = (char)('0' + a1);
  memcpy(o + 1, DEC_PAIRS + 2 * a0, 2);
  put8(o + 3, b);
  put8(o + 11, e);
}

// writes the decimal digits of the 32-byte LE integer, returns the length (no terminator)
static inline size_t u256_to_dec(const uint8_t le[32], char* out) {
  uint64_t v[4];
  memcpy(v, le, 32);
  int top = 3;
  while (top >= 0 && v[top] == 0) top--;
  if (top < 0) {
    out[0] = '0';
    return 1;
  }
  uint64_t chunk[5];
  int nchunk = 0;
  while (top >= 0) {
    uint64_t rem = 0;
    for (int i = top; i >= 0; i--) v[i] = div_1e19(rem, v[i], &rem);
    chunk[nchunk++] = rem;
    while (top >= 0 && v[top] == 0) top--;
  }
  char* o = out;
  // most significant chunk without leading zeros
  {
    char tmp[20];
    int n = 0;
    uint64_t c = chunk[nchunk - 1];
    do {
      tmp[n++] = (char)('0' + c % 10);
      c /= 10;
    } while (c);
    while (n) *o++ = tmp[--n];
  }
  // every further chunk is exactly 19 digits
  for (int k = nchunk - 2; k >= 0; k--) {
    put19(o, chunk[k]);
    o += 19;
  }
  return (size_t)(o - out);
}

// `show (s :: Fr)`  (Utils.hs:153,165,174)
static inline size_t show_fr(ShowStyle st, const uint8_t s[32], char* out) {
  char* o = out;
  if (st == SHOW_DERIVED) {
    *o++ = 'P';
    *o++ = ' ';
  }
  o += u256_to_dec(s, o);
  return (size_t)(o - out);
}

// `pointToBS = show`  (Utils.hs:32-33); 64 zero bytes encode `O`
static inline size_t show_point(ShowStyle st, const uint8_t pt[64], char* out) {
  uint64_t w[8];
  memcpy(w, pt, 64);
  bool inf = (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0;
  char* o = out;
  if (inf) {
    *o++ = 'O';
    return 1;
  }
  if (st == SHOW_DERIVED) {
    memcpy(o, "A (P ", 5);
    o += 5;
    o += u256_to_dec(pt, o);
    memcpy(o, ") (P ", 5);
    o += 5;
    o += u256_to_dec(pt + 32, o);
    *o++ = ')';
  } else {
    *o++ = 'A';
    *o++ = ' ';
    o += u256_to_dec(pt, o);
    *o++ = ' ';
    o += u256_to_dec(pt + 32, o);
  }
  return (size_t)(o - out);
}

// The same texts from numbers the GPU has already formatted (dec.h: 80-byte records, byte 0 = digit count n,
// digits right-aligned in [80 - n, 80)); the host only concatenates.
static inline size_t put_dec_record(const uint8_t* rec, char* out) {
  size_t n = rec[0];
  memcpy(out, rec + 80 - n, n);
  return n;
}
static inline bool dec_record_is_zero(const uint8_t* rec) { return rec[0] == 1 && rec[79] == '0'; }
static inline size_t show_fr_rec(ShowStyle st, const uint8_t* rec, char* out) {
  char* o = out;
  if (st == SHOW_DERIVED) {
    *o++ = 'P';
    *o++ = ' ';
  }
  o += put_dec_record(rec, o);
  return (size_t)(o - out);
}
static inline size_t show_point_rec(ShowStyle st, const uint8_t* recx, const uint8_t* recy, char* out) {
  char* o = out;
  if (dec_record_is_zero(recx) && dec_record_is_zero(recy)) {  // (0, 0) encodes `O`
    *o++ = 'O';
    return 1;
  }
  if (st == SHOW_DERIVED) {
    memcpy(o, "A (P ", 5);
    o += 5;
    o += put_dec_record(recx, o);
    memcpy(o, ") (P ", 5);
    o += 5;
    o += put_dec_record(recy, o);
    *o++ = ')';
  } else {
    *o++ = 'A';
    *o++ = ' ';
    o += put_dec_record(recx, o);
    *o++ = ' ';
    o += put_dec_record(recy, o);
  }
  return (size_t)(o - out);
}

static const char SHOW_R[] = "115792089237316195423570985008687907852837564279074904382605163141518161494337";
static const size_t SHOW_R_LEN = sizeof(SHOW_R) - 1;
// little-endian 64-bit limbs
static const uint64_t MOD_R[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL,
                                  0xFFFFFFFFFFFFFFFFULL};
static const uint64_t MOD_P[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                                  0xFFFFFFFFFFFFFFFFULL};

// digest (big-endian integer) mod m -> 32 bytes little-endian; 2^256 < 2m for both moduli
static inline void digest_mod(const uint8_t digest[32], const uint64_t m[4], uint8_t out_le[32]) {
  uint64_t v[4];
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    for (int j = 0; j < 8; j++) w = (w << 8) | digest[8 * (3 - i) + j];
    v[i] = w;
  }
  uint64_t d[4];
  unsigned __int128 bw = 0;
  for (int i = 0; i < 4; i++) {
    unsigned __int128 t = (unsigned __int128)v[i] - m[i] - (uint64_t)bw;
    d[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
  memcpy(out_le, bw ? v : d, 32);
}

// `oracle` (Utils.hs:29-30); msg needs 72 bytes of slack (sha256_inplace)
static inline void oracle(uint8_t* msg, size_t len, const uint64_t m[4], uint8_t out_le[32]) {
  uint8_t dg[32];
  sha256_inplace(msg, len, dg);
  digest_mod(dg, m, out_le);
}

// state after the first block of `show _r` (78 characters: the first 64 are one SHA block)
static inline const ShaMid& sha_mid_show_r() {
  static const ShaMid mid = [] {
    ShaMid m = sha_mid_init();
    sha_mid_advance(m, (const uint8_t*)SHOW_R, SHOW_R_LEN);
    return m;
  }();
  return mid;
}

// Incremental message builder over a caller-provided scratch buffer
struct Msg {
  uint8_t* buf;
  size_t len;
  ShowStyle st;
  bool opens_with_r = false;
  Msg(uint8_t* b, ShowStyle s) : buf(b), len(0), st(s) {}
  void r() {
    if (len == 0) opens_with_r = true;
    memcpy(buf + len, SHOW_R, SHOW_R_LEN);
    len += SHOW_R_LEN;
  }
  void point(const uint8_t* p) { len += show_point(st, p, (char*)buf + len); }
  void fr(const uint8_t* s) { len += show_fr(st, s, (char*)buf + len); }
  // the same from GPU-formatted records (a point is two consecutive records: x, y)
  void point_rec(const uint8_t* rec2) { len += show_point_rec(st, rec2, rec2 + 80, (char*)buf + len); }
  void fr_rec(const uint8_t* rec) { len += show_fr_rec(st, rec, (char*)buf + len); }
  void raw(const void* p, size_t n) {
    memcpy(buf + len, p, n);
    len += n;
  }
  void finish_fr(uint8_t out_le[32]) {
    if (opens_with_r)
      finish_fr_from(sha_mid_show_r(), out_le);
    else
      oracle(buf, len, MOD_R, out_le);
  }
  // as finish_fr, for a message whose first mid.done bytes are already absorbed in `mid`
  void finish_fr_from(const ShaMid& mid, uint8_t out_le[32]) {
    uint8_t dg[32];
    sha256_resume_inplace(mid, buf, len, dg);
    digest_mod(dg, MOD_R, out_le);
  }
};
// finish two messages together (interleaved SHA-256)
static inline void finish_fr_pair(Msg& a, const ShaMid& ma, uint8_t outa_le[32], Msg& b, const ShaMid& mb, uint8_t outb_le[32]) {
  uint8_t da[32], db[32];
  sha256_resume_inplace_x2(ma, a.buf, a.len, da, mb, b.buf, b.len, db);
  digest_mod(da, MOD_R, outa_le);
  digest_mod(db, MOD_R, outb_le);
}
constexpr size_t MSG_SCRATCH = 1280;  // longest transcript: shamirX, 8 numbers + show_r + slack

}  // namespace host
}  // namespace bp
