// One point operation spread over neighbouring lanes (device only).
//
// A chain of DEPENDENT point operations in one lane (the Horner chain of the MSM's windows, the 129 double-and-add steps
// of an IPP round) is pure latency: one wave issues an instruction every ~5 cycles and nothing else can use the result.
// The products inside one operation are partly independent, so two or four neighbouring lanes take one product each
// per round and pass operands with quad-permute DPP moves (`__builtin_amdgcn_update_dpp`: register to register inside a
// quad, no LDS).  All lanes run ONE instruction stream; a lane without a useful product in some round multiplies whatever
// it holds.  Arithmetic, limb classes and the exceptional-case filter are those of fq29.h (xyzz29_dbl, xyzz29_madd).
//
//   quad  (4 lanes): doubling in 3 rounds instead of 9 products (2.7 us against 4.2 us in one lane), full addition in
//         4 rounds instead of 14 (3.1 us against 5.7 us; benchmarks/microbench --lanes, one wave per SIMD).  msm.hip k_final / k_window_sum_quad, ipp.hip k_bit_tables, and
//         block_sum_quad: the workgroup sums of the one-lane-per-bit products and of the provers' fixed-base terms.
//   pair  (2 lanes): doubling + mixed addition in 5 + 5 rounds instead of 9 + 10, (5.2 us against 6.8 us) for only ~1/3 more instructions in
//         total (the quad form would be 3 + 4 rounds for more than twice the instructions).  verify.hip k_round_pair,
//         k_stage1_tail_pair.
// Every caller keeps whole quads / pairs active together: the permutes only read lanes of the same group.
#pragma once
#include "devio.h"
#include "fq29.h"

namespace bp {

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ fe29 quad_bcast(const fe29& v) {  // every lane reads lane SRC of its quad
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = dpp_quad<SRC * 0x55>(v.v[i]);
  return r;
}
__device__ __forceinline__ fe29 pair_xchg(const fe29& v) {  // the neighbour's value: quad_perm [1, 0, 3, 2]
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = dpp_quad<0xB1>(v.v[i]);
  return r;
}
__device__ __forceinline__ fe29 fe29_sel(bool c, const fe29& a, const fe29& b) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
__device__ __forceinline__ fe29 fe29_shl(const fe29& a, uint32_t sh) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << sh;
  return r;
}
__device__ __forceinline__ fe29 fe29_times3_norm(const fe29& a) {  // 3a < 2^30.6 -> N
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] * 3u;
  return fe29_norm(r);
}

// ---- quad: lane 0 holds Y, lane 1 X, lane 2 ZZZ, lane 3 ZZ ----------------------------------------------------------
//   round A     V = (2Y)^2           XX = X^2           -                   -
//   round B     W = 2Y V             S = X V            MM = M M, M = 3 XX  ZZ3 = ZZ V
//   round C     W (64 p - Y)         M (S - X3)         ZZZ3 = W ZZZ        -            X3 = MM - 2 S
//   new state   Y3 = sum of lanes 0, 1                  X3                  ZZZ3         ZZ3
// 2Y < 2^30.001 meets only class N operands; the differences are normalised before they are multiplied; the identity
// (all limbs zero) stays zero.
__device__ __forceinline__ fe29 quad_spread(const xyzz29& p, uint32_t lane) {
  return fe29_sel(lane == 0, p.Y, fe29_sel(lane == 1, p.X, fe29_sel(lane == 2, p.ZZZ, p.ZZ)));
}
__device__ __forceinline__ xyzz29 quad_gather(const fe29& P) {
  xyzz29 r;
  r.Y = quad_bcast<0>(P);
  r.X = quad_bcast<1>(P);
  r.ZZZ = quad_bcast<2>(P);
  r.ZZ = quad_bcast<3>(P);
  return r;
}
__device__ __forceinline__ fe29 quad_dbl(const fe29& P, uint32_t lane) {
  const fe29 A = fe29_shl(P, lane == 0 ? 1u : 0u);  // lane 0: U = 2Y; the others: their own coordinate
  const fe29 Q1 = fe29_sqr(A);                      // V | XX | - | -
  const fe29 V = quad_bcast<0>(Q1);
  const fe29 M = fe29_times3_norm(quad_bcast<1>(Q1));
  const bool l1 = lane == 1, l2 = lane == 2;
  const fe29 Q2 = fe29_mul(fe29_sel(l2, M, A), fe29_sel(l2, M, V));  // W | S | MM | ZZ3
  const fe29 W = quad_bcast<0>(Q2), S = quad_bcast<1>(Q2), MM = quad_bcast<2>(Q2);
  const fe29 X3 = fe29_norm(fe29_add(MM, fe29_neg2(fe29_shl(S, 1))));  // M^2 + (128 p - 2S)
  const fe29 T = fe29_add(S, fe29_neg(X3));                           // S - X3, lazy
  const fe29 Q3 = fe29_mul(fe29_sel(l1, M, W), fe29_sel(l1, T, fe29_sel(lane == 0, fe29_neg(P), P)));  // -WY | MT | ZZZ3 | -
  const fe29 Y3 = fe29_norm(fe29_add(Q3, quad_bcast<1>(Q3)));  // lane 0: M (S - X3) - W Y
  return fe29_sel(lane == 0, Y3, fe29_sel(l1, X3, fe29_sel(l2, Q3, Q2)));
}

// p + q for two XYZZ points, both spread over the quad (add-2008-s; 12 products and 2 squares in four rounds):
//   lane:       0 (Y)                1 (X)              2 (ZZZ)             3 (ZZ)
//   round A     S1 = Y1 ZZZ2         U1 = X1 ZZ2        S2 = Y2 ZZZ1        U2 = X2 ZZ1    R = S2 - S1, P = U2 - U1
//   round B     RR = R^2             PP = P^2           ZZZ1 ZZZ2           ZZ1 ZZ2
//   round C     PPP = P PP           Q = U1 PP          PPP                 ZZ3 = ZZ1 ZZ2 PP
//   round D     PPP (64 p - S1)      R (Q - X3)         ZZZ3 = ZZZ1 ZZZ2 PPP  -            X3 = RR - PPP - 2Q
//   new state   Y3 = sum of lanes 0, 1                  X3                  ZZZ3           ZZ3
// ZZ3 = 0 (mod p) - equal or opposite points, an identity on either side - goes to the exact routine of ec.h.
__device__ __forceinline__ fe29 quad_add(const fe29& pc, const fe29& qc, uint32_t lane) {
  const bool lo = lane < 2, even = (lane & 1u) == 0, l0 = lane == 0, l1 = lane == 1;
  fe29 qx, Q1x;
#pragma unroll
  for (int i = 0; i < 9; i++) qx.v[i] = dpp_quad<0x4E>(qc.v[i]);  // quad_perm [2, 3, 0, 1]: ZZZ2 | ZZ2 | Y2 | X2
  const fe29 Q1 = fe29_mul(pc, qx);                               // S1 | U1 | S2 | U2
#pragma unroll
  for (int i = 0; i < 9; i++) Q1x.v[i] = dpp_quad<0x4E>(Q1.v[i]);  // S2 | U2 | S1 | U1
  const fe29 D = fe29_norm(fe29_add(fe29_sel(lo, Q1x, Q1), fe29_neg(fe29_sel(lo, Q1, Q1x))));  // R | P | R | P
  const fe29 Q2 = fe29_mul(fe29_sel(lo, D, pc), fe29_sel(lo, D, qc));                           // RR | PP | ZZZ12 | ZZ12
  const fe29 PP = quad_bcast<1>(Q2), Pd = quad_bcast<1>(D);
  const fe29 Q3 = fe29_mul(fe29_sel(even, Pd, fe29_sel(l1, Q1, Q2)), PP);                       // PPP | Q | PPP | ZZ3
  const uint32_t l6 = Q3.v[6], l7 = Q3.v[7];
  const uint32_t mine = lane == 3 && l6 == l7 && (l6 == 0u || l6 == M29) ? 1u : 0u;
  if (dpp_quad<3 * 0x55>(mine) != 0u)
    return quad_spread(xyzz29_add_exact(quad_gather(pc), quad_gather(qc)), lane);
  const fe29 RR = quad_bcast<0>(Q2), PPP = quad_bcast<0>(Q3), Q = quad_bcast<1>(Q3);
  fe29 W;  // PPP + 2Q < 2^30.6
#pragma unroll
  for (int i = 0; i < 9; i++) W.v[i] = PPP.v[i] + (Q.v[i] << 1);
  const fe29 X3 = fe29_norm(fe29_add(RR, fe29_neg2(W)));
  const fe29 T = fe29_add(Q, fe29_neg(X3));  // lazy
  const fe29 Rb = quad_bcast<0>(D);
  const fe29 Q4 = fe29_mul(fe29_sel(l0, fe29_neg(Q1), fe29_sel(l1, Rb, Q2)), fe29_sel(l1, T, Q3));  // -S1 PPP | R T | ZZZ3 | -
  const fe29 Y3 = fe29_norm(fe29_add(Q4, quad_bcast<1>(Q4)));
  return fe29_sel(l0, Y3, fe29_sel(l1, X3, fe29_sel(lane == 2, Q4, Q3)));
}
// The same for sums in which identities are COMMON (the terms of a one-lane-per-bit product: half of them are O): an
// identity operand is recognised by its all-zero ZZ and answered by selects, so that only P + P and P - P take the
// exact routine.  (An identity written as a non-zero multiple of p would still reach it through the filter.)
__device__ __forceinline__ uint32_t quad_is_inf(const fe29& c) {  // lane 3 holds ZZ
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) nz |= c.v[i];
  return dpp_quad<3 * 0x55>(nz == 0u ? 1u : 0u);
}
__device__ __forceinline__ fe29 quad_add_total(const fe29& pc, const fe29& qc, uint32_t lane) {
  const uint32_t pinf = quad_is_inf(pc), qinf = quad_is_inf(qc);
  if ((pinf | qinf) != 0u) return fe29_sel(pinf != 0u, qc, pc);  // whole quads
  return quad_add(pc, qc, lane);
}
// Sum of one XYZZ term per thread over a workgroup of 64, 128 or 256 threads (thread 0 returns it), by quads: the terms
// cross LDS coordinate by coordinate, quad q adds the terms q, q + n/4, q + n/2, q + 3n/4, then a tree over the n/4 quads.
// 3 + log2(n/4) additions of 3-4 us deep, against log2(n) levels of ~12 us for shuffled single-lane additions.
// lds: n * 36 words.
__device__ __forceinline__ xyzz block_sum_quad(const xyzz& term, uint32_t* lds) {
  const uint32_t t = threadIdx.x, n = blockDim.x, nq = n >> 2, q = t >> 2, lane = t & 3u;
  {
    const xyzz29 v = xyzz29_from(term);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      lds[t * 36 + i] = v.Y.v[i];
      lds[t * 36 + 9 + i] = v.X.v[i];
      lds[t * 36 + 18 + i] = v.ZZZ.v[i];
      lds[t * 36 + 27 + i] = v.ZZ.v[i];
    }
  }
  __syncthreads();
  fe29 acc;
#pragma unroll
  for (int i = 0; i < 9; i++) acc.v[i] = lds[q * 36 + lane * 9 + i];
#pragma unroll 1
  for (uint32_t k = 1; k < 4; k++) {
    fe29 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.v[i] = lds[(q + k * nq) * 36 + lane * 9 + i];
    acc = quad_add_total(acc, o, lane);
  }
  __syncthreads();  // every term has been read: the area now carries the tree
#pragma unroll 1
  for (uint32_t m = nq >> 1; m >= 1; m >>= 1) {
    if (q >= m && q < 2 * m) {
#pragma unroll
      for (int i = 0; i < 9; i++) lds[q * 36 + lane * 9 + i] = acc.v[i];
    }
    __syncthreads();
    if (q < m) {
      fe29 o;
#pragma unroll
      for (int i = 0; i < 9; i++) o.v[i] = lds[(q + m) * 36 + lane * 9 + i];
      acc = quad_add_total(acc, o, lane);
    }
    __syncthreads();
  }
  return xyzz29_to(quad_gather(acc));  // meaningful in quad 0
}

// the lane's coordinate of an XYZZ point in memory (8 x 32 limbs, weakly reduced): Y | X | ZZZ | ZZ
__device__ __forceinline__ uint32_t quad_coord_offset(uint32_t lane) { return lane == 0 ? 32u : lane == 1 ? 0u : lane == 2 ? 96u : 64u; }
__device__ __forceinline__ fe29 quad_load(const xyzz* p, uint32_t lane) {
  return fe29_from_u256(ld_u256((const char*)p + quad_coord_offset(lane)));
}
__device__ __forceinline__ void quad_store(xyzz* p, const fe29& v, uint32_t lane) {
  st_u256((char*)p + quad_coord_offset(lane), fe29_to_u256(v));
}

// ---- pair: the even lane holds (Y, ZZZ), the odd lane (X, ZZ) --------------------------------------------------------
struct pair29 {
  fe29 c0, c1;
};
__device__ __forceinline__ pair29 pair_spread(const xyzz29& p, bool odd) {
  pair29 s;
  s.c0 = fe29_sel(odd, p.X, p.Y);
  s.c1 = fe29_sel(odd, p.ZZ, p.ZZZ);
  return s;
}
__device__ __forceinline__ xyzz29 pair_gather(const pair29& s, bool odd) {
  const fe29 o0 = pair_xchg(s.c0), o1 = pair_xchg(s.c1);
  xyzz29 r;
  r.X = fe29_sel(odd, s.c0, o0);
  r.Y = fe29_sel(odd, o0, s.c0);
  r.ZZ = fe29_sel(odd, s.c1, o1);
  r.ZZZ = fe29_sel(odd, o1, s.c1);
  return r;
}
// doubling:      even lane            odd lane
//   round A      V = (2Y)^2           XX = X^2
//   round B      W = 2Y V             S = X V                  (V crosses)
//   round C      ZZZ3 = W ZZZ         MM = M M, M = 3 XX       X3 = MM - 2S, T = S - X3 in the odd lane
//   round D      W (64 p - Y)         ZZ3 = V ZZ
//   round E      -                    M T                      (M T crosses: Y3 = M T - W Y)
__device__ __forceinline__ pair29 pair_dbl(const pair29& s, bool odd) {
  const fe29 A = fe29_shl(s.c0, odd ? 0u : 1u);       // 2Y | X
  const fe29 Q1 = fe29_sqr(A);                         // V | XX
  const fe29 V = fe29_sel(odd, pair_xchg(Q1), Q1);     // V in both lanes
  const fe29 Q2 = fe29_mul(A, V);                      // W | S
  const fe29 M = fe29_times3_norm(Q1);                 // (unused) | M
  const fe29 Q3 = fe29_mul(fe29_sel(odd, M, Q2), fe29_sel(odd, M, s.c1));  // ZZZ3 | MM
  const fe29 X3 = fe29_norm(fe29_add(Q3, fe29_neg2(fe29_shl(Q2, 1))));      // (unused) | MM - 2S
  const fe29 T = fe29_add(Q2, fe29_neg(X3));                                // (unused) | S - X3, lazy
  const fe29 Q4 = fe29_mul(fe29_sel(odd, V, Q2), fe29_sel(odd, s.c1, fe29_neg(s.c0)));  // W (64 p - Y) | ZZ3
  const fe29 Q5 = fe29_mul(M, T);                      // (unused) | M T
  const fe29 Y3 = fe29_norm(fe29_add(Q4, pair_xchg(Q5)));  // M T - W Y | (unused)
  pair29 r;
  r.c0 = fe29_sel(odd, X3, Y3);
  r.c1 = fe29_sel(odd, Q4, Q3);
  return r;
}
// mixed addition of the affine point at `entry` (x at +0, y at +32, both < 2^256; (0, 0) is the identity):
//                even lane            odd lane
//   round A      S2 = ZZZ y           U2 = ZZ x                R = S2 - Y | P = U2 - X, normalised
//   round B      RR = R^2             PP = P^2                 (P, R, PP, RR cross)
//   round C      PPP = P PP           Q = X PP                 (PPP, Q cross)  X3 = RR - PPP - 2Q, T = Q - X3 in both
//   round D      PPP (64 p - Y)       ZZ3 = ZZ PP
//   round E      ZZZ3 = ZZZ PPP       R T                      (R T crosses: Y3 = R T - Y PPP)
// ZZ3 = 0 (mod p) - P + P, P - P, accumulator at O - goes to the exact routine of ec.h, as in xyzz29_madd.
__device__ __forceinline__ pair29 pair_madd(const pair29& s, const uint8_t* entry, bool odd) {
  const u256 own = ld_u256(entry + (odd ? 0 : 32));  // the odd lane needs x, the even lane y
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) nz |= own.v[i];
  if (((nz | dpp_quad<0xB1>(nz)) == 0u)) return s;   // the identity; both lanes of the pair agree
  const fe29 D = fe29_mul_add(s.c1, fe29_from_u256(own), fe29_neg(s.c0));  // R = S2 - Y | P = U2 - X, out of one reduction
  const fe29 Q2 = fe29_sqr(D);                                    // RR | PP
  const fe29 Dx = pair_xchg(D), Q2x = pair_xchg(Q2);              // P | R,  PP | RR
  const fe29 PP = fe29_sel(odd, Q2, Q2x), RR = fe29_sel(odd, Q2x, Q2);
  const fe29 Q3 = fe29_mul(fe29_sel(odd, s.c0, Dx), PP);          // PPP = P PP | Q = X PP
  const fe29 Q3x = pair_xchg(Q3);
  const fe29 PPP = fe29_sel(odd, Q3x, Q3), Q = fe29_sel(odd, Q3, Q3x);
  fe29 W;                                                         // PPP + 2Q < 2^30.6
#pragma unroll
  for (int i = 0; i < 9; i++) W.v[i] = PPP.v[i] + (Q.v[i] << 1);
  const fe29 X3 = fe29_norm(fe29_add(RR, fe29_neg2(W)));          // both lanes
  const fe29 T = fe29_add(Q, fe29_neg(X3));                       // lazy
  const fe29 Q4 = fe29_mul(fe29_sel(odd, PP, PPP), fe29_sel(odd, s.c1, fe29_neg(s.c0)));  // PPP (64 p - Y) | ZZ3
  const fe29 Q5 = fe29_mul(fe29_sel(odd, Dx, PPP), fe29_sel(odd, T, s.c1));                // ZZZ3 | R T
  const uint32_t l6 = Q4.v[6], l7 = Q4.v[7];
  const uint32_t mine = odd && l6 == l7 && (l6 == 0u || l6 == M29) ? 1u : 0u;
  if ((mine | dpp_quad<0xB1>(mine)) != 0u)
    return pair_spread(xyzz29_madd_exact(pair_gather(s, odd), ld_affine(entry), false), odd);
  const fe29 Y3 = fe29_norm(fe29_add(Q4, pair_xchg(Q5)));         // R T - Y PPP | (unused)
  pair29 r;
  r.c0 = fe29_sel(odd, X3, Y3);
  r.c1 = fe29_sel(odd, Q4, Q5);
  return r;
}

}  // namespace bp
