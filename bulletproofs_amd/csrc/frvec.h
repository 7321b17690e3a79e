// Scalar-field (Fr) vector work of the provers and of IPP.verifyProof, per element.
//
// The bodies below are what one lane of the kernels in frvec.hip computes; they are BP_HD so that the CPU tests
// (tests/hostsim) run exactly this source against the oracle.  Reference:
//   powerVector, hadamard, dot           /root/reference/Bulletproofs/Utils.hs:50-69
//   computeLRPolys, computeTPoly         MultiRangeProof/Prover.hs:133-169
//   ls', rs', cL, cR of an IPP round     InnerProductProof/Prover.hs:60-61,84-85
//   mkOtherExponents                     InnerProductProof/Verifier.hs:55-73
//   computePolynomials, evaluatePolynomial, vectorMatrixProduct
//                                        ArithmeticCircuit/Prover.hs:104-117, Internal.hs:123-152
// Values are canonical residues in Montgomery form (Fr of fe.h) unless a name says `plain`: the Montgomery product
// of a Montgomery-form and a plain operand is the plain product, which is how scalars leave these kernels canonical
// without a conversion pass.
#pragma once
#include "fe.h"

namespace bp {

// a^e for a small exponent, a in Montgomery form
BP_HD u256 fr_pow_u32(const u256& a, uint32_t e) {
  u256 r = Fr::one(), b = a;
  while (e) {
    if (e & 1u) r = Fr::mul(r, b);
    e >>= 1;
    if (e) b = Fr::sqr(b);
  }
  return r;
}

// element i of `powerVector a n` (Utils.hs:50-52): a^i, except that powerVector 0 n is all zeros (0^0 = 0 there)
BP_HD u256 fr_power_elem(const u256& a, uint32_t i) {
  if (u256_is_zero(a)) return u256_zero();
  return fr_pow_u32(a, i);
}

BP_HD u256 fr_from_plain(const u256& plain) { return Fr::to_mont(Fr::reduce_once(plain)); }

// 2^b as a Montgomery-form scalar, b < 256
BP_HD u256 fr_two_pow(uint32_t b) {
  u256 p = u256_zero();
  p.v[b >> 5] = 1u << (b & 31);
  return fr_from_plain(p);
}

// computeLRPolys (MultiRangeProof/Prover.hs:133-154), element i = j * n + b of the n*m-vectors:
//   l0 = aL - z, l1 = sL, r0 = y^i (aR + z) + z^(j+2) 2^b, r1 = y^i sR       (aR = aL - 1; j from 0 here)
struct RpPolyElem {
  u256 l0, r0, r1;
};
BP_HD RpPolyElem rp_poly_elem(uint32_t i, uint32_t n_bits, uint32_t a_l_bit, const u256& s_r, const u256& y,
                              const u256& z) {
  const uint32_t j = i / n_bits, b = i % n_bits;
  const u256 one = Fr::one();
  const u256 a_l = a_l_bit ? one : u256_zero();
  const u256 a_r = Fr::sub(a_l, one);
  const u256 yi = fr_power_elem(y, i);
  RpPolyElem e;
  e.l0 = Fr::sub(a_l, z);
  e.r0 = Fr::add(Fr::mul(yi, Fr::add(a_r, z)), Fr::mul(fr_pow_u32(z, j + 2), fr_two_pow(b)));
  e.r1 = Fr::mul(yi, s_r);
  return e;
}

// One IPP round on the witness (InnerProductProof/Prover.hs:84-85), index j < h of vectors of length 2h:
//   ls'_j = x ls_j + xInv ls_{h+j},  rs'_j = xInv rs_j + x rs_{h+j}
BP_HD void ipp_fold_elem(u256& l_lo, const u256& l_hi, u256& r_lo, const u256& r_hi, const u256& x, const u256& xi) {
  l_lo = Fr::add(Fr::mul(x, l_lo), Fr::mul(xi, l_hi));
  r_lo = Fr::add(Fr::mul(xi, r_lo), Fr::mul(x, r_hi));
}

// The fold of the generator vectors (Prover.hs:81-82) as coefficients of the ORIGINAL generators (prover.h,
// ipp_prove_fixed): generator i sits in the low half of its block of nk_prev when (i mod nk_prev) < nk_prev / 2;
// gs'' = xInv gs_lo + x gs_hi, hs'' = x hs_lo + xInv hs_hi.  cg, ch are plain-form coefficients.
BP_HD void ipp_coeff_elem(uint32_t i, uint32_t nk_prev, u256& cg_plain, u256& ch_plain, const u256& x, const u256& xi) {
  const bool lo = (i & (nk_prev - 1)) < nk_prev / 2;
  cg_plain = Fr::mul(cg_plain, lo ? xi : x);
  ch_plain = Fr::mul(ch_plain, lo ? x : xi);
}

// The six canonical scalars generator pair i contributes to the three sums of a round over the original generators:
//   C = commitment of (ls, rs) under the folded base, L = lCommit, R = rCommit (Prover.hs:63-73); nk = current length.
struct IppScalars {
  u256 c_g, c_h, l_g, l_h, r_g, r_h;  // canonical (plain)
};
template <class LoadL, class LoadR>
BP_HD IppScalars ipp_scalar_elem(uint32_t i, uint32_t nk, const u256& cg_plain, const u256& ch_plain, LoadL ls,
                                 LoadR rs) {
  const uint32_t j = i & (nk - 1), h = nk / 2;
  IppScalars s;
  s.c_g = Fr::mul(ls(j), cg_plain);
  s.c_h = Fr::mul(rs(j), ch_plain);
  s.l_g = s.l_h = s.r_g = s.r_h = u256_zero();
  if (nk >= 2) {
    if (j >= h) {
      s.l_g = Fr::mul(ls(j - h), cg_plain);  // lCommit = <lsLeft, gsRight> + <rsRight, hsLeft> + cL bH   (:63-67)
      s.r_h = Fr::mul(rs(j - h), ch_plain);  // rCommit = <lsRight, gsLeft> + <rsLeft, hsRight> + cR bH   (:69-73)
    } else {
      s.l_h = Fr::mul(rs(j + h), ch_plain);
      s.r_g = Fr::mul(ls(j + h), cg_plain);
    }
  }
  return s;
}

// mkOtherExponents (InnerProductProof/Verifier.hs:55-73) for a power-of-two n = 2^lg: the Map's first insert of
// index i1 comes from i = i1 - 2^j with j the top bit of i1, so s_i = s_0 * prod over the set bits j of i of c_j^2,
// where c_j = challenges !! j and `challenges` is the REVERSED round order (Verifier.hs:50).
// c2_rev[j] = c_j^2 in Montgomery form.
BP_HD u256 ipp_other_exponent(uint32_t i, uint32_t lg, const u256& s0, const u256* c2_rev) {
  u256 s = s0;
  for (uint32_t j = 0; j < lg; j++)
    if ((i >> j) & 1u) s = Fr::mul(s, c2_rev[j]);
  return s;
}

// computePolynomials (ArithmeticCircuit/Prover.hs:104-117), element i:
//   l1 = aL + y^-i zwR, l2 = aO, l3 = sL;  r0 = zwO - y^i, r1 = y^i aR + zwL, r3 = y^i sR   (l0 = r2 = 0)
struct AcPolyElem {
  u256 l1, r0, r1, r3;
};
BP_HD AcPolyElem ac_poly_elem(const u256& a_l, const u256& a_r, const u256& s_r, const u256& yi, const u256& yinv_i,
                              const u256& zw_l, const u256& zw_r, const u256& zw_o) {
  AcPolyElem e;
  e.l1 = Fr::add(a_l, Fr::mul(yinv_i, zw_r));
  e.r0 = Fr::sub(zw_o, yi);
  e.r1 = Fr::add(Fr::mul(yi, a_r), zw_l);
  e.r3 = Fr::mul(yi, s_r);
  return e;
}

}  // namespace bp
