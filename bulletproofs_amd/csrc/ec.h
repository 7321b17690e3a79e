// secp256k1 group law in XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2).
//
// Replaces `<>`/`add`, `inv`, `mul`, `mempty` of elliptic-curve-0.3.0 as used by
// the reference (/root/reference/Bulletproofs/Utils.hs:71-85,
// RangeProof/Internal.hs:183-205, InnerProductProof/Prover.hs:63-92).  The
// reference's group law is total, so every routine here handles the identity,
// P + P and P + (-P) exactly.
//
// Affine points: (x, y) canonical or weakly reduced; the identity `O` is
// encoded as (0, 0), which is not on y^2 = x^3 + 7.
// XYZZ points: identity <=> ZZ == 0 (mod p).
#pragma once
#include "fe.h"
#include "modinv30.h"

namespace bp {

struct affine {
  u256 x, y;
};
struct xyzz {
  u256 X, Y, ZZ, ZZZ;
};

BP_HD bool affine_is_inf(const affine& a) { return u256_is_zero(a.x) && u256_is_zero(a.y); }
BP_HD affine affine_inf() {
  affine a;
  a.x = u256_zero();
  a.y = u256_zero();
  return a;
}
BP_HD affine affine_neg(const affine& a) {
  affine r;
  r.x = a.x;
  r.y = affine_is_inf(a) ? a.y : fq_neg(a.y);
  return r;
}
BP_HD affine affine_gen() {
  affine g;
  const uint32_t gx[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu,
                          0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
  const uint32_t gy[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u,
                          0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    g.x.v[i] = gx[i];
    g.y.v[i] = gy[i];
  }
  return g;
}
// y^2 == x^3 + 7, or the identity encoding
BP_HD bool affine_on_curve(const affine& a) {
  if (affine_is_inf(a)) return true;
  u256 seven = u256_zero();
  seven.v[0] = 7;
  u256 rhs = fq_add(fq_mul(fq_sqr(a.x), a.x), seven);
  return fq_eq(fq_sqr(a.y), rhs);
}

BP_HD xyzz xyzz_inf() {
  xyzz r;
  r.X = u256_zero();
  r.Y = u256_zero();
  r.ZZ = u256_zero();
  r.ZZZ = u256_zero();
  return r;
}
BP_HD bool xyzz_is_inf(const xyzz& p) { return fq_is_zero(p.ZZ); }
BP_HD xyzz xyzz_from_affine(const affine& a) {
  xyzz r;
  if (affine_is_inf(a)) return xyzz_inf();
  r.X = a.x;
  r.Y = a.y;
  r.ZZ = u256_one();
  r.ZZZ = u256_one();
  return r;
}
BP_HD xyzz xyzz_neg(const xyzz& p) {
  xyzz r = p;
  r.Y = fq_neg(p.Y);
  return r;
}

// 2*(x, y) for an affine, non-identity point (mdbl-2008-s-1, a = 0). y != 0 on
// secp256k1 (no 2-torsion), so the result is never the identity.
BP_HD xyzz xyzz_dbl_affine(const affine& a) {
  xyzz r;
  u256 U = fq_dbl(a.y);
  u256 V = fq_sqr(U);
  u256 W = fq_mul(U, V);
  u256 S = fq_mul(a.x, V);
  u256 X2 = fq_sqr(a.x);
  u256 M = fq_add(fq_dbl(X2), X2);
  r.X = fq_sub(fq_sqr(M), fq_dbl(S));
  r.Y = fq_sub(fq_mul(M, fq_sub(S, r.X)), fq_mul(W, a.y));
  r.ZZ = V;
  r.ZZZ = W;
  return r;
}

// 2*P (dbl-2008-s-1, a = 0): 6M + 3S
BP_HD xyzz xyzz_dbl(const xyzz& p) {
  if (xyzz_is_inf(p)) return xyzz_inf();
  xyzz r;
  u256 U = fq_dbl(p.Y);
  u256 V = fq_sqr(U);
  u256 W = fq_mul(U, V);
  u256 S = fq_mul(p.X, V);
  u256 X2 = fq_sqr(p.X);
  u256 M = fq_add(fq_dbl(X2), X2);
  r.X = fq_sub(fq_sqr(M), fq_dbl(S));
  r.Y = fq_sub(fq_mul(M, fq_sub(S, r.X)), fq_mul(W, p.Y));
  r.ZZ = fq_mul(V, p.ZZ);
  r.ZZZ = fq_mul(W, p.ZZZ);
  return r;
}

// P + Q, Q affine (madd-2008-s): 8M + 2S on the generic path
BP_HD xyzz xyzz_madd(const xyzz& p, const affine& q) {
  if (affine_is_inf(q)) return p;
  if (xyzz_is_inf(p)) return xyzz_from_affine(q);
  u256 U2 = fq_mul(q.x, p.ZZ);
  u256 S2 = fq_mul(q.y, p.ZZZ);
  u256 Pd = fq_sub(U2, p.X);
  u256 Rd = fq_sub(S2, p.Y);
  if (fq_is_zero(Pd)) {
    if (fq_is_zero(Rd)) return xyzz_dbl_affine(q);
    return xyzz_inf();
  }
  xyzz r;
  u256 PP = fq_sqr(Pd);
  u256 PPP = fq_mul(Pd, PP);
  u256 Q = fq_mul(p.X, PP);
  r.X = fq_sub(fq_sub(fq_sqr(Rd), PPP), fq_dbl(Q));
  r.Y = fq_sub(fq_mul(Rd, fq_sub(Q, r.X)), fq_mul(p.Y, PPP));
  r.ZZ = fq_mul(p.ZZ, PP);
  r.ZZZ = fq_mul(p.ZZZ, PPP);
  return r;
}

// P + Q (add-2008-s): 12M + 2S on the generic path
BP_HD xyzz xyzz_add(const xyzz& p, const xyzz& q) {
  if (xyzz_is_inf(q)) return p;
  if (xyzz_is_inf(p)) return q;
  u256 U1 = fq_mul(p.X, q.ZZ);
  u256 U2 = fq_mul(q.X, p.ZZ);
  u256 S1 = fq_mul(p.Y, q.ZZZ);
  u256 S2 = fq_mul(q.Y, p.ZZZ);
  u256 Pd = fq_sub(U2, U1);
  u256 Rd = fq_sub(S2, S1);
  if (fq_is_zero(Pd)) {
    if (fq_is_zero(Rd)) return xyzz_dbl(p);
    return xyzz_inf();
  }
  xyzz r;
  u256 PP = fq_sqr(Pd);
  u256 PPP = fq_mul(Pd, PP);
  u256 Q = fq_mul(U1, PP);
  r.X = fq_sub(fq_sub(fq_sqr(Rd), PPP), fq_dbl(Q));
  r.Y = fq_sub(fq_mul(Rd, fq_sub(Q, r.X)), fq_mul(S1, PPP));
  r.ZZ = fq_mul(fq_mul(p.ZZ, q.ZZ), PP);
  r.ZZZ = fq_mul(fq_mul(p.ZZZ, q.ZZZ), PPP);
  return r;
}

// Normalise with a known inverse of ZZZ*ZZ... see xyzz_to_affine for the
// single-point form.  Given zi = 1/ZZZ:  1/ZZ = (ZZ * zi)^2 since ZZ^3 = ZZZ^2.
BP_HD affine xyzz_to_affine_with_inv(const xyzz& p, const u256& zzz_inv) {
  if (xyzz_is_inf(p)) return affine_inf();
  affine a;
  u256 t = fq_mul(p.ZZ, zzz_inv);  // = 1/Z for Z with ZZ = Z^2, ZZZ = Z^3
  u256 zz_inv = fq_sqr(t);
  a.x = fq_canon(fq_mul(p.X, zz_inv));
  a.y = fq_canon(fq_mul(p.Y, zzz_inv));
  return a;
}
BP_HD affine xyzz_to_affine(const xyzz& p) {
  if (xyzz_is_inf(p)) return affine_inf();
  return xyzz_to_affine_with_inv(p, fq_inv_gcd(p.ZZZ));  // division steps (modinv30.h): a third of Kaliski's instructions
}

// Same group element?  (cross-multiplied comparison, no inversion)
BP_HD bool xyzz_eq(const xyzz& p, const xyzz& q) {
  bool pi = xyzz_is_inf(p), qi = xyzz_is_inf(q);
  if (pi || qi) return pi && qi;
  return fq_eq(fq_mul(p.X, q.ZZ), fq_mul(q.X, p.ZZ)) && fq_eq(fq_mul(p.Y, q.ZZZ), fq_mul(q.Y, p.ZZZ));
}

// k*P by left-to-right double-and-add over the bits of a canonical scalar.
// (The shape of the reference's `mul`; used for one-off products and tests.)
BP_HD xyzz xyzz_mul_affine(const affine& p, const u256& k) {
  xyzz acc = xyzz_inf();
  u256 kk = k;
  for (int i = 255; i >= 0; i--) {
    acc = xyzz_dbl(acc);
    if (u256_pop_msb(kk)) acc = xyzz_madd(acc, p);
  }
  return acc;
}

}  // namespace bp
