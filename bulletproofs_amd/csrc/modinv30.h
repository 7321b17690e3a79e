// Modular inversion by Bernstein-Yang division steps ("safegcd", https://gcd.cr.yp.to/papers.html#safegcd) on nine
// signed 30-bit limbs, for both moduli of secp256k1.
//
// Why: the rounds of the batch verifier invert one scalar (x_r^-1 mod r) and one base-field element (1/ZZZ mod p)
// per proof and round, lane by lane.  Kaliski's almost-inverse (fe.h) costs ~66 k VALU instructions, the Fermat chain
// on 29-bit limbs (fq29.h) ~37 k; 600 division steps in batches of 30 on 32-bit words plus twenty 2x2-matrix updates
// cost ~21 k, with a fixed instruction sequence for every input (no divergence between lanes).
//
// The algorithm keeps f, g (initially m, x) and d, e (initially 0, 1) with  d x = f,  e x = g  (mod m).  A batch of 30
// division steps only looks at the low 30 bits of f and g and yields a transition matrix t = (u v; q r) with
// (f, g) <- t (f, g) / 2^30; d and e follow by the same matrix, with a multiple of m added so that the division by
// 2^30 is exact.  After 20 batches (600 >= 590 steps, the bound for 256-bit inputs with the "half-delta" start
// zeta = -1) g is 0 and f is +-1, so +-d is the inverse.  0 maps to 0.
#pragma once
#include "fe.h"

namespace bp {

struct S30 {
  int32_t v[9];
};
struct ModInv30 {
  int32_t m[9];      // the modulus, limbs in [0, 2^30)
  uint32_t m_inv30;  // m^-1 mod 2^30
};
constexpr int32_t MI_M30 = (int32_t)0x3FFFFFFF;

BP_HD ModInv30 modinv30_r() {
  return ModInv30{{0x10364141, 0x3f497a33, 0x348a03bb, 0x2bb739ab, 0x3ffffeba, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff},
                  0x2a774ec1u};
}
BP_HD ModInv30 modinv30_p() {
  return ModInv30{{0x3ffffc2f, 0x3ffffffb, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff},
                  0x2ddacacfu};
}

BP_HD S30 s30_from_u256(const u256& a) {
  S30 r;
  r.v[0] = (int32_t)(a.v[0] & 0x3FFFFFFFu);
  r.v[1] = (int32_t)(((a.v[0] >> 30) | (a.v[1] << 2)) & 0x3FFFFFFFu);
  r.v[2] = (int32_t)(((a.v[1] >> 28) | (a.v[2] << 4)) & 0x3FFFFFFFu);
  r.v[3] = (int32_t)(((a.v[2] >> 26) | (a.v[3] << 6)) & 0x3FFFFFFFu);
  r.v[4] = (int32_t)(((a.v[3] >> 24) | (a.v[4] << 8)) & 0x3FFFFFFFu);
  r.v[5] = (int32_t)(((a.v[4] >> 22) | (a.v[5] << 10)) & 0x3FFFFFFFu);
  r.v[6] = (int32_t)(((a.v[5] >> 20) | (a.v[6] << 12)) & 0x3FFFFFFFu);
  r.v[7] = (int32_t)(((a.v[6] >> 18) | (a.v[7] << 14)) & 0x3FFFFFFFu);
  r.v[8] = (int32_t)(a.v[7] >> 16);
  return r;
}
// limbs in [0, 2^30), value below 2^256
BP_HD u256 s30_to_u256(const S30& a) {
  const uint32_t* l = (const uint32_t*)a.v;
  u256 r;
  r.v[0] = l[0] | (l[1] << 30);
  r.v[1] = (l[1] >> 2) | (l[2] << 28);
  r.v[2] = (l[2] >> 4) | (l[3] << 26);
  r.v[3] = (l[3] >> 6) | (l[4] << 24);
  r.v[4] = (l[4] >> 8) | (l[5] << 22);
  r.v[5] = (l[5] >> 10) | (l[6] << 20);
  r.v[6] = (l[6] >> 12) | (l[7] << 18);
  r.v[7] = (l[7] >> 14) | (l[8] << 16);
  return r;
}

// 30 division steps on the low words: (zeta, f, g) -> zeta', and the matrix t = (u v; q r), entries in [-2^30, 2^30]
BP_HD int32_t modinv30_divsteps(int32_t zeta, uint32_t f0, uint32_t g0, int32_t (&t)[4]) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 30; i++) {
    uint32_t c1 = (uint32_t)(zeta >> 31);  // all ones when zeta < 0
    const uint32_t c2 = 0u - (g & 1u);     // all ones when g is odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // +-f, +-u, +-v
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;  // swap as well
    zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t[0] = (int32_t)u;
  t[1] = (int32_t)v;
  t[2] = (int32_t)q;
  t[3] = (int32_t)r;
  return zeta;
}

// (d, e) <- t (d, e) / 2^30 mod m, both kept in (-2 m, m)
BP_HD void modinv30_update_de(S30& d, S30& e, const int32_t (&t)[4], const ModInv30& mi) {
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;  // sign masks
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int32_t di = d.v[0], ei = e.v[0];
  int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
  // the multiple of m that makes the low 30 bits vanish
  md -= (int32_t)((mi.m_inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)MI_M30);
  me -= (int32_t)((mi.m_inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)MI_M30);
  cd += (int64_t)mi.m[0] * md;
  ce += (int64_t)mi.m[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    di = d.v[i];
    ei = e.v[i];
    cd += (int64_t)u * di + (int64_t)v * ei;
    ce += (int64_t)q * di + (int64_t)r * ei;
    cd += (int64_t)mi.m[i] * md;
    ce += (int64_t)mi.m[i] * me;
    d.v[i - 1] = (int32_t)cd & MI_M30;
    cd >>= 30;
    e.v[i - 1] = (int32_t)ce & MI_M30;
    ce >>= 30;
  }
  d.v[8] = (int32_t)cd;
  e.v[8] = (int32_t)ce;
}

// (f, g) <- t (f, g) / 2^30 (exact)
BP_HD void modinv30_update_fg(S30& f, S30& g, const int32_t (&t)[4]) {
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  int32_t fi = f.v[0], gi = g.v[0];
  int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; i++) {
    fi = f.v[i];
    gi = g.v[i];
    cf += (int64_t)u * fi + (int64_t)v * gi;
    cg += (int64_t)q * fi + (int64_t)r * gi;
    f.v[i - 1] = (int32_t)cf & MI_M30;
    cf >>= 30;
    g.v[i - 1] = (int32_t)cg & MI_M30;
    cg >>= 30;
  }
  f.v[8] = (int32_t)cf;
  g.v[8] = (int32_t)cg;
}

// r in (-2 m, m), negated when sign < 0, brought to [0, m) with limbs in [0, 2^30)
BP_HD void modinv30_normalize(S30& r, int32_t sign, const ModInv30& mi) {
  int32_t c[9];
#pragma unroll
  for (int i = 0; i < 9; i++) c[i] = r.v[i];
  int32_t add = c[8] >> 31;  // negative: add m
  const int32_t neg = sign >> 31;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c[i] += mi.m[i] & add;
    c[i] = (c[i] ^ neg) - neg;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c[i + 1] += c[i] >> 30;
    c[i] &= MI_M30;
  }
  add = c[8] >> 31;  // still negative after the negation: add m once more
#pragma unroll
  for (int i = 0; i < 9; i++) c[i] += mi.m[i] & add;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c[i + 1] += c[i] >> 30;
    c[i] &= MI_M30;
  }
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c[i];
}

// x^-1 mod m for canonical x in [0, m); 0 -> 0
BP_HD u256 modinv30(const u256& x, const ModInv30& mi) {
  S30 d, e, f, g = s30_from_u256(x);
#pragma unroll
  for (int i = 0; i < 9; i++) {
    d.v[i] = 0;
    e.v[i] = 0;
    f.v[i] = mi.m[i];
  }
  e.v[0] = 1;
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; it++) {
    int32_t t[4];
    zeta = modinv30_divsteps(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    modinv30_update_de(d, e, t, mi);
    modinv30_update_fg(f, g, t);
  }
  // g = 0, f = +-1 (or +-m for x = 0, where d = 0)
  modinv30_normalize(d, f.v[8], mi);
  return s30_to_u256(d);
}

// 1/a mod p for a weakly reduced base-field element (fe.h form: below 2^256); 0 -> 0
BP_HD u256 fq_inv_gcd(const u256& a) { return modinv30(fq_canon(a), modinv30_p()); }

// Montgomery inverse in the scalar field: A R -> A^-1 R; 0 -> 0.  The plain inverse of A R is A^-1 R^-1, and one
// Montgomery product with R^3 (= 2^768 mod r) turns it into A^-1 R.
BP_HD u256 fr_inv_gcd(const u256& a_mont) {
  const uint32_t R3[8] = {0xE9FF41EDu, 0x7BC0CFE0u, 0x44D4322Cu, 0x00176484u, 0xF1D0B2DAu, 0xB1B31347u, 0x18EF116Du, 0x555D800Cu};
  u256 r3;
#pragma unroll
  for (int i = 0; i < 8; i++) r3.v[i] = R3[i];
  return Fr::mul(modinv30(a_mont, modinv30_r()), r3);
}

}  // namespace bp
