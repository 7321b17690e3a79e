// Variable-base scalar multiplication for the verifier's non-generator points
// (L_k, R_k, S, V_j, T1, T2; /root/reference/Bulletproofs/InnerProductProof/Verifier.hs:49,
// MultiRangeProof/Verifier.hs:59-65, RangeProof/Internal.hs:185), replacing the reference's
// per-term double-and-add `mul`.
//
// secp256k1 has the endomorphism phi(x, y) = (beta*x, y) = lambda*(x, y).  A scalar k is split
// as k = k1 + k2*lambda (mod r) with |k1|, |k2| < 2^128, so k*P = k1*P + k2*phi(P) needs 128
// doublings instead of 256.  Up to four half-length terms share one doubling chain through a
// joint table of all their subset sums (Straus): one table addition per bit.
// The group element computed is exactly k*P (+ ...); only the evaluation order differs.
#pragma once
#include "ec.h"
#include "fq29.h"

namespace bp {

struct GlvHalf {
  uint32_t m[5];  // magnitude, < 2^129 (proven < 2^128; one spare bit is carried)
  bool neg;
};

BP_HD u256 glv_const(const uint32_t (&t)[8]) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return r;
}

// round(k * g / 2^384) for 256-bit k, g
BP_HD u256 glv_mul_shift_384(const u256& k, const u256& g) {
  uint32_t t[16];
  mul_wide(t, k, g);
  u256 r = u256_zero();
  uint64_t c = (t[11] >> 31) & 1u;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += t[12 + i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  r.v[4] = (uint32_t)c;
  return r;
}

// k (canonical, < r) -> (k1, k2), k = k1 + k2*lambda (mod r)
BP_HD void glv_split(const u256& k, GlvHalf& h1, GlvHalf& h2) {
  const uint32_t G1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
  const uint32_t G2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};
  // (-b1), (-b2), lambda premultiplied by 2^256 mod r so that one Montgomery product with a plain operand gives x*c mod r
  const uint32_t MB1_M[8] = {0x0AD9263Cu, 0xC50468D0u, 0xFAA6ED42u, 0x1B1C8205u, 0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu};
  const uint32_t MB2_M[8] = {0x6A144696u, 0x0CAC5E50u, 0xF3BA5939u, 0x1E8A8DC5u, 0xBA244FCEu, 0x176CDF65u, 0x8E173580u, 0xC25575EBu};
  const uint32_t LAM_M[8] = {0xC9926C9Eu, 0xF07DEB3Du, 0x83C6944Cu, 0x2C93E7ADu, 0x52697D91u, 0x73A96606u, 0x8558D639u, 0x53284017u};
  const uint32_t HALF_N[8] = {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
  u256 c1 = glv_mul_shift_384(k, glv_const(G1));
  u256 c2 = glv_mul_shift_384(k, glv_const(G2));
  u256 r2 = Fr::add(Fr::mul(c1, glv_const(MB1_M)), Fr::mul(c2, glv_const(MB2_M)));
  u256 r1 = Fr::sub(k, Fr::mul(r2, glv_const(LAM_M)));
  u256 half = glv_const(HALF_N);
  u256 t;
  h1.neg = u256_sub(t, half, r1) != 0;  // r1 > (r-1)/2
  if (h1.neg) r1 = Fr::neg(r1);
  h2.neg = u256_sub(t, half, r2) != 0;
  if (h2.neg) r2 = Fr::neg(r2);
#pragma unroll
  for (int i = 0; i < 5; i++) {
    h1.m[i] = r1.v[i];
    h2.m[i] = r2.v[i];
  }
}

// k (canonical, < r) -> non-negative halves m1, m2 < 2^129 with k = m1 + m2*lambda (mod r).
// glv_split's signed halves satisfy |k1| < 0.64 * 2^128, |k2| < 0.55 * 2^128 (rounding to the lattice
// basis (a1, b1), (a2, b2)); adding the lattice vector (a2 - a1, b2 - b1) = (|b1|, a2), which is = 0 under
// (x, y) -> x + y*lambda, moves both into (0.25, 1.63) * 2^128.  With no signs left, a joint table
// over P, phi(P), ... depends on the points only and can be built before the scalars are known.
struct GlvPos {
  uint32_t m1[5], m2[5];
};
BP_HD GlvPos glv_split_pos(const u256& k) {
  const uint32_t G1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
  const uint32_t G2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};
  const uint32_t MB1_M[8] = {0x0AD9263Cu, 0xC50468D0u, 0xFAA6ED42u, 0x1B1C8205u, 0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu};
  const uint32_t MB2_M[8] = {0x6A144696u, 0x0CAC5E50u, 0xF3BA5939u, 0x1E8A8DC5u, 0xBA244FCEu, 0x176CDF65u, 0x8E173580u, 0xC25575EBu};
  const uint32_t LAM_M[8] = {0xC9926C9Eu, 0xF07DEB3Du, 0x83C6944Cu, 0x2C93E7ADu, 0x52697D91u, 0x73A96606u, 0x8558D639u, 0x53284017u};
  const uint32_t OFF1[8] = {0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0u, 0u, 0u, 0u};           // |b1| = a2 - a1
  const uint32_t OFF2[8] = {0x9D44CFD8u, 0x57C1108Du, 0xA8E2F3F6u, 0x14CA50F7u, 0x00000001u, 0u, 0u, 0u};  // a2 = b2 - b1
  u256 c1 = glv_mul_shift_384(k, glv_const(G1));
  u256 c2 = glv_mul_shift_384(k, glv_const(G2));
  u256 r2 = Fr::add(Fr::mul(c1, glv_const(MB1_M)), Fr::mul(c2, glv_const(MB2_M)));
  u256 r1 = Fr::sub(k, Fr::mul(r2, glv_const(LAM_M)));
  // r1, r2 are the signed halves mod r; the offsets bring both into [0, 2^129)
  r1 = Fr::add(r1, glv_const(OFF1));
  r2 = Fr::add(r2, glv_const(OFF2));
  GlvPos g;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    g.m1[i] = r1.v[i];
    g.m2[i] = r2.v[i];
  }
  return g;
}

// phi(P) = (beta * x, y)
BP_HD affine glv_phi(const affine& p) {
  const uint32_t BETA[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};
  affine r;
  r.x = fq_mul(p.x, glv_const(BETA));  // (0,0) stays the identity encoding
  r.y = p.y;
  return r;
}

// P + phi(P) = -phi^2(P) = (beta^2 x, -y): the three images of P under the endomorphism add up to O (1 + beta + beta^2 = 0
// puts them on one line y = const), so this entry of a joint table costs one product
BP_HD affine glv_p_plus_phi(const affine& p) {
  const uint32_t BETA2[8] = {0x8E6AFA40u, 0x3EC693D6u, 0xED0A766Au, 0x630FB68Au, 0x53CBCB16u, 0x919BB861u, 0x9A83F8EFu, 0x851695D4u};
  if (affine_is_inf(p)) return affine_inf();
  affine r;
  r.x = fq_mul(p.x, glv_const(BETA2));
  r.y = fq_neg(p.y);
  return r;
}

// 129-bit magnitudes as a left-aligned 160-bit shift register: top bit first
struct BitReg {
  uint32_t w[5];
};
BP_HD BitReg bitreg_init(const GlvHalf& h) {
  BitReg b;
  b.w[4] = (h.m[4] << 31) | (h.m[3] >> 1);
  b.w[3] = (h.m[3] << 31) | (h.m[2] >> 1);
  b.w[2] = (h.m[2] << 31) | (h.m[1] >> 1);
  b.w[1] = (h.m[1] << 31) | (h.m[0] >> 1);
  b.w[0] = h.m[0] << 31;
  return b;
}
BP_HD uint32_t bitreg_next(BitReg& b) {
  uint32_t top = b.w[4] >> 31;
  b.w[4] = (b.w[4] << 1) | (b.w[3] >> 31);
  b.w[3] = (b.w[3] << 1) | (b.w[2] >> 31);
  b.w[2] = (b.w[2] << 1) | (b.w[1] >> 31);
  b.w[1] = (b.w[1] << 1) | (b.w[0] >> 31);
  b.w[0] <<= 1;
  return top;
}
constexpr int GLV_BITS = 129;
BP_HD BitReg bitreg_init5(const uint32_t (&m)[5]) {
  GlvHalf h;
#pragma unroll
  for (int i = 0; i < 5; i++) h.m[i] = m[i];
  h.neg = false;
  return bitreg_init(h);
}

// Joint table over T signed half-terms q[0..T): entry e in [1, 2^T) = sum_{j in bits(e)} q[j], XYZZ.
// The caller stores the single-term entries (e = 2^j) with xyzz_from_affine(q[j]); this routine then
// forms every composite entry from two STORED entries (the entry without its top term, and the top
// term read back as an affine point), so the loop has one addition site and nothing is indexed by
// a runtime value in registers.
template <int T, class Store, class Load>
BP_HD void joint_table_composites(Store store, Load load) {
#pragma unroll 1
  for (int e = 3; e < (1 << T); e++) {
    int top = 31 - __builtin_clz((unsigned)e);
    int rest = e & ~(1 << top);
    if (rest == 0) continue;  // single-term entry, already stored
    xyzz single = load(1 << top);  // ZZ = ZZZ = 1, or the identity
    affine qt;
    qt.x = single.X;
    qt.y = single.Y;
    if (xyzz_is_inf(single)) qt = affine_inf();
    store(e, xyzz_madd(load(rest), qt));
  }
}

// Split k, store +-P and +-phi(P) as the single-term entries ea, eb and return their bit streams.
struct HalfPair {
  BitReg a, b;
};
template <class Store>
BP_HD HalfPair glv_prep_point(const u256& k, const affine& P, int ea, int eb, Store store) {
  GlvHalf h1, h2;
  glv_split(k, h1, h2);
  affine p2 = glv_phi(P);
  store(ea, xyzz_from_affine(h1.neg ? affine_neg(P) : P));
  store(eb, xyzz_from_affine(h2.neg ? affine_neg(p2) : p2));
  HalfPair hp;
  hp.a = bitreg_init(h1);
  hp.b = bitreg_init(h2);
  return hp;
}

// k0*P0 + k1*P1 through the sign-free joint table of the pair: load(e), e in [1, 16), returns the AFFINE sum
// over the bits of e of {P0, phi(P0), P1, phi(P1)} ((0,0) for the identity).  One doubling and one mixed
// addition per bit; the entry of the next bit is requested before the current step's arithmetic.
template <class LoadAff>
BP_HD xyzz glv_pair_multiexp_tab(const u256& k0, const u256& k1, LoadAff load) {
  GlvPos s0 = glv_split_pos(k0), s1 = glv_split_pos(k1);
  BitReg b0 = bitreg_init5(s0.m1), b1 = bitreg_init5(s0.m2), b2 = bitreg_init5(s1.m1), b3 = bitreg_init5(s1.m2);
  auto next_idx = [&]() -> uint32_t {
    return bitreg_next(b0) | (bitreg_next(b1) << 1) | (bitreg_next(b2) << 2) | (bitreg_next(b3) << 3);
  };
  xyzz acc = xyzz_inf();
  uint32_t idx = next_idx();
  affine q = affine_inf();
  if (idx) q = load((int)idx);
#pragma unroll 1
  for (int i = 0; i < GLV_BITS; i++) {
    affine cur = q;
    q = affine_inf();
    if (i + 1 < GLV_BITS) {
      idx = next_idx();
      if (idx) q = load((int)idx);
    }
    acc = xyzz_dbl(acc);
    acc = xyzz_madd(acc, cur);
  }
  return acc;
}

// bit i of a 129-bit magnitude
BP_HD uint32_t mag_bit(const uint32_t (&m)[5], int i) {
  const int w = i >> 5;
  uint32_t v = m[0];
  v = w == 1 ? m[1] : v;
  v = w == 2 ? m[2] : v;
  v = w == 3 ? m[3] : v;
  v = w >= 4 ? m[4] : v;
  return (v >> (i & 31)) & 1u;
}
BP_HD xyzz xyzz_phi(const xyzz& p) {  // (beta X, Y, ZZ, ZZZ); the identity stays the identity
  const uint32_t BETA[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};
  xyzz r = p;
  r.X = fq_mul(p.X, glv_const(BETA));
  return r;
}
// The same chain with the accumulator on nine 29-bit limbs (fq29.h): lazy doubling and mixed addition, one addition
// site; the table load is unconditional (index 0 re-reads entry 1 and drops it) so that nothing waits for it before the
// step's arithmetic has been issued.
template <class LoadAff>
BP_HD xyzz glv_pair_multiexp_tab29(const u256& k0, const u256& k1, LoadAff load) {
  GlvPos s0 = glv_split_pos(k0), s1 = glv_split_pos(k1);
  BitReg b0 = bitreg_init5(s0.m1), b1 = bitreg_init5(s0.m2), b2 = bitreg_init5(s1.m1), b3 = bitreg_init5(s1.m2);
  auto next_idx = [&]() -> uint32_t {
    return bitreg_next(b0) | (bitreg_next(b1) << 1) | (bitreg_next(b2) << 2) | (bitreg_next(b3) << 3);
  };
  xyzz29 acc = xyzz29_inf();
  uint32_t idx = next_idx();
  affine q = load((int)(idx ? idx : 1u));
#pragma unroll 1
  for (int i = 0; i < GLV_BITS; i++) {
    const affine cur = q;
    const uint32_t icur = idx;
    idx = i + 1 < GLV_BITS ? next_idx() : 0u;
    q = load((int)(idx ? idx : 1u));
    acc = xyzz29_dbl(acc);
    if (icur && !affine_is_inf(cur)) acc = xyzz29_madd(acc, cur, false);
  }
  return xyzz29_to(acc);
}

// NT sign-free joint tables on ONE doubling chain: sum over t of k[2t] P_t0 + k[2t+1] P_t1, load(t, e) the AFFINE
// entry e in [1, 16) of table t as above.  A table of a single point (present[2t+1] false) has entries 1..3 only; its
// absent half-scalars stay zero, so nothing above 3 is ever asked for.  One doubling and up to NT mixed additions
// per bit through ONE addition site; the entries of the next bit are requested before the current step's arithmetic.
template <int NT, class LoadAff>
BP_HD xyzz glv_tables_multiexp29(const u256 (&k)[2 * NT], const bool (&present)[2 * NT], LoadAff load) {
  static_assert(NT == 1 || NT == 2, "one or two tables per chain");
  BitReg b[NT][4];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      GlvPos s;
#pragma unroll
      for (int i = 0; i < 5; i++) s.m1[i] = s.m2[i] = 0;
      if (present[2 * t + j]) s = glv_split_pos(k[2 * t + j]);
      b[t][2 * j] = bitreg_init5(s.m1);
      b[t][2 * j + 1] = bitreg_init5(s.m2);
    }
  uint32_t idx[NT];
  affine q[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    idx[t] = bitreg_next(b[t][0]) | (bitreg_next(b[t][1]) << 1) | (bitreg_next(b[t][2]) << 2) | (bitreg_next(b[t][3]) << 3);
    q[t] = load(t, (int)(idx[t] ? idx[t] : 1u));
  }
  xyzz29 acc = xyzz29_inf();
#pragma unroll 1
  for (int i = 0; i < GLV_BITS; i++) {
    affine cur[NT];
    uint32_t icur[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      cur[t] = q[t];
      icur[t] = idx[t];
      idx[t] = 0u;
      if (i + 1 < GLV_BITS)
        idx[t] = bitreg_next(b[t][0]) | (bitreg_next(b[t][1]) << 1) | (bitreg_next(b[t][2]) << 2) | (bitreg_next(b[t][3]) << 3);
      q[t] = load(t, (int)(idx[t] ? idx[t] : 1u));
    }
    acc = xyzz29_dbl(acc);
    if constexpr (NT == 1) {
      if (icur[0] && !affine_is_inf(cur[0])) acc = xyzz29_madd(acc, cur[0], false);
    } else {
#pragma unroll 1
      for (int t = 0; t < NT; t++) {
        const affine c = t ? cur[NT - 1] : cur[0];
        const uint32_t ic = t ? icur[NT - 1] : icur[0];
        if (ic && !affine_is_inf(c)) acc = xyzz29_madd(acc, c, false);
      }
    }
  }
  return xyzz29_to(acc);
}

// ---- k*P with one lane per bit -------------------------------------------------------------------------------
// Lane j < GLV_LANE_BITS holds Tj = 2^j P (XYZZ) and contributes [bit j of |k1|] (+-Tj) + [bit j of |k2|] (+-phi(Tj)),
// k = k1 + k2*lambda the signed split: no doubling and at most one addition per lane, and the lanes' terms add up
// to k*P.  |k1|, |k2| < 2^128 (glv_split); the spare bit 128 the halves carry is served by the last lane.
constexpr int GLV_LANE_BITS = 128;
BP_HD xyzz glv_bit_lane(const GlvHalf& h1, const GlvHalf& h2, const xyzz& Tj, int j) {
  xyzz acc = xyzz_inf();
  if (mag_bit(h1.m, j)) acc = h1.neg ? xyzz_neg(Tj) : Tj;
  if (mag_bit(h2.m, j)) {
    xyzz q = xyzz_phi(Tj);
    acc = xyzz_add(acc, h2.neg ? xyzz_neg(q) : q);
  }
  if (j == GLV_LANE_BITS - 1 && ((h1.m[4] | h2.m[4]) & 1u)) {
    xyzz t2 = xyzz_dbl(Tj);
    if (h1.m[4] & 1u) acc = xyzz_add(acc, h1.neg ? xyzz_neg(t2) : t2);
    if (h2.m[4] & 1u) {
      xyzz q = xyzz_phi(t2);
      acc = xyzz_add(acc, h2.neg ? xyzz_neg(q) : q);
    }
  }
  return acc;
}

// Host-testable construction of that table (the device builds it with one shared inversion, verify.hip):
// tab[e], e in [1, 16); tab[0] unused.
BP_HD void glv_pair_table_local(const affine& P0, const affine& P1, affine* tab) {
  xyzz t[16];
  t[1] = xyzz_from_affine(P0);
  t[2] = xyzz_from_affine(glv_phi(P0));
  t[4] = xyzz_from_affine(P1);
  t[8] = xyzz_from_affine(glv_phi(P1));
  auto store = [&](int e, const xyzz& v) { t[e] = v; };
  auto load = [&](int e) { return t[e]; };
  joint_table_composites<4>(store, load);
  for (int e = 1; e < 16; e++) tab[e] = xyzz_to_affine(t[e]);
}

// Host-testable reference of the whole construction for one or two points:
//   k[0]*P[0] (+ k[1]*P[1]) via GLV + one joint table in `tab` (size >= 16).
template <int NP>
BP_HD xyzz glv_multiexp_local(const u256 (&k)[NP], const affine (&P)[NP], xyzz* tab) {
  auto store = [&](int e, const xyzz& v) { tab[e] = v; };
  auto load = [&](int e) { return tab[e]; };
  HalfPair h0 = glv_prep_point(k[0], P[0], 1, 2, store);
  HalfPair h1 = h0;
  if (NP == 2) h1 = glv_prep_point(k[NP - 1], P[NP - 1], 4, 8, store);
  joint_table_composites<2 * NP>(store, load);
  xyzz acc = xyzz_inf();
  for (int i = 0; i < GLV_BITS; i++) {
    acc = xyzz_dbl(acc);
    uint32_t idx = bitreg_next(h0.a) | (bitreg_next(h0.b) << 1);
    if (NP == 2) idx |= (bitreg_next(h1.a) << 2) | (bitreg_next(h1.b) << 3);
    if (idx) acc = xyzz_add(acc, tab[idx]);
  }
  return acc;
}

}  // namespace bp
