// Host-side scalar field Fr of secp256k1 (4 x 64-bit limbs, Montgomery form) for the
// protocol control code that stays on the CPU: the O(nm) vector work of the prover
// (/root/reference/Bulletproofs/MultiRangeProof/Prover.hs:133-169, Utils.hs:50-69) and the
// benchmark's proof generator.  Replaces `Fr` of galois-field as used on the host.
#pragma once
#include <stdint.h>
#include <string.h>

namespace bp {
namespace host {

struct Fr4 {
  uint64_t v[4];
};

static const uint64_t FR_M[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL,
                                 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t FR_R1[4] = {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1ULL, 0ULL};
static const uint64_t FR_R2[4] = {0x896CF21467D7D140ULL, 0x741496C20E7CF878ULL, 0xE697F5E45BCD07C6ULL,
                                  0x9D671CD581C69BC5ULL};
static const uint64_t FR_N0INV = 0x4B0DFF665588B13FULL;
typedef unsigned __int128 u128;

static inline bool fr_geq_m(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > FR_M[i]) return true;
    if (a[i] < FR_M[i]) return false;
  }
  return true;
}
static inline void fr_sub_m(uint64_t a[4]) {
  u128 bw = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a[i] - FR_M[i] - (uint64_t)bw;
    a[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
}
static inline Fr4 fr_zero() { return Fr4{{0, 0, 0, 0}}; }
static inline Fr4 fr_one() { return Fr4{{FR_R1[0], FR_R1[1], FR_R1[2], FR_R1[3]}}; }
static inline bool fr_is_zero(const Fr4& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static inline bool fr_eq(const Fr4& a, const Fr4& b) { return memcmp(a.v, b.v, 32) == 0; }

static inline Fr4 fr_add(const Fr4& a, const Fr4& b) {
  Fr4 r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.v[i] + b.v[i];
    r.v[i] = (uint64_t)c;
    c >>= 64;
  }
  if (c || fr_geq_m(r.v)) fr_sub_m(r.v);
  return r;
}
static inline Fr4 fr_sub(const Fr4& a, const Fr4& b) {
  Fr4 r;
  u128 bw = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a.v[i] - b.v[i] - (uint64_t)bw;
    r.v[i] = (uint64_t)t;
    bw = (t >> 64) & 1;
  }
  if (bw) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)r.v[i] + FR_M[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
static inline Fr4 fr_neg(const Fr4& a) { return fr_sub(fr_zero(), a); }

// Montgomery product (CIOS)
static inline Fr4 fr_mul(const Fr4& a, const Fr4& b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a.v[i] * b.v[j] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t q = t[0] * FR_N0INV;
    c = (u128)q * FR_M[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)q * FR_M[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  Fr4 r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || fr_geq_m(r.v)) fr_sub_m(r.v);
  return r;
}
static inline Fr4 fr_sqr(const Fr4& a) { return fr_mul(a, a); }

// canonical little-endian bytes <-> Montgomery form
static inline Fr4 fr_from_bytes(const uint8_t le[32]) {
  Fr4 a;
  memcpy(a.v, le, 32);
  if (fr_geq_m(a.v)) fr_sub_m(a.v);
  Fr4 r2 = {{FR_R2[0], FR_R2[1], FR_R2[2], FR_R2[3]}};
  return fr_mul(a, r2);
}
static inline void fr_to_bytes(const Fr4& a, uint8_t le[32]) {
  Fr4 one = {{1, 0, 0, 0}};
  Fr4 c = fr_mul(a, one);
  memcpy(le, c.v, 32);
}
static inline Fr4 fr_from_u64(uint64_t x) {
  uint8_t b[32] = {0};
  memcpy(b, &x, 8);
  return fr_from_bytes(b);
}
// ---- modular inversion: binary extended Euclid on four 64-bit limbs (a few microseconds) ----------------------
typedef unsigned __int128 inv_u128;

static inline bool inv_geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > b[i]) return true;
    if (a[i] < b[i]) return false;
  }
  return true;
}
static inline uint64_t inv_add(uint64_t a[4], const uint64_t b[4]) {  // a += b, returns the carry
  inv_u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (inv_u128)a[i] + b[i];
    a[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
static inline void inv_sub(uint64_t a[4], const uint64_t b[4]) {  // a -= b (mod 2^256)
  uint64_t bw = 0;
  for (int i = 0; i < 4; i++) {
    inv_u128 t = (inv_u128)a[i] - b[i] - bw;
    a[i] = (uint64_t)t;
    bw = (uint64_t)(t >> 64) & 1;
  }
}
static inline void inv_shr1(uint64_t a[4], uint64_t top) {  // a = (top:a) >> 1
  a[0] = (a[0] >> 1) | (a[1] << 63);
  a[1] = (a[1] >> 1) | (a[2] << 63);
  a[2] = (a[2] >> 1) | (a[3] << 63);
  a[3] = (a[3] >> 1) | (top << 63);
}
// x <- x / 2 mod m (m odd)
static inline void inv_half(uint64_t x[4], const uint64_t m[4]) {
  uint64_t top = 0;
  if (x[0] & 1) top = inv_add(x, m);
  inv_shr1(x, top);
}
static inline bool inv_is_one(const uint64_t a[4]) { return a[0] == 1 && (a[1] | a[2] | a[3]) == 0; }

// out = a^-1 mod m for an odd prime m and a < m; 0 for a = 0.  Binary extended Euclid (HAC 14.61 for odd moduli):
// invariants x1 * a = u, x2 * a = v (mod m).
static inline void modinv256(const uint64_t a[4], const uint64_t m[4], uint64_t out[4]) {
  uint64_t u[4] = {a[0], a[1], a[2], a[3]}, v[4] = {m[0], m[1], m[2], m[3]};
  uint64_t x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
  if ((u[0] | u[1] | u[2] | u[3]) == 0) {
    memset(out, 0, 32);
    return;
  }
  while (!inv_is_one(u) && !inv_is_one(v)) {
    while (!(u[0] & 1)) {
      inv_shr1(u, 0);
      inv_half(x1, m);
    }
    while (!(v[0] & 1)) {
      inv_shr1(v, 0);
      inv_half(x2, m);
    }
    if (inv_geq(u, v)) {
      inv_sub(u, v);
      if (!inv_geq(x1, x2)) inv_add(x1, m);  // the sum wraps 2^256 only together with the subtraction below
      inv_sub(x1, x2);
    } else {
      inv_sub(v, u);
      if (!inv_geq(x2, x1)) inv_add(x2, m);
      inv_sub(x2, x1);
    }
  }
  memcpy(out, inv_is_one(u) ? x1 : x2, 32);
}

// a^-1 (Montgomery form in and out); inverse of 0 is 0
static inline Fr4 fr_inv(const Fr4& a) {
  // the plain inverse of a R is a^-1 R^-1; two Montgomery products with R^2 bring it to a^-1 R
  Fr4 x, r2 = {{FR_R2[0], FR_R2[1], FR_R2[2], FR_R2[3]}};
  modinv256(a.v, FR_M, x.v);
  return fr_mul(fr_mul(x, r2), r2);
}

}  // namespace host
}  // namespace bp
