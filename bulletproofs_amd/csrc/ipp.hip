// Inner-product-argument prover kernels on gfx950.
//
// One recursion step of generateProof' (/root/reference/Bulletproofs/InnerProductProof/Prover.hs:52-92):
//   lCommit = <l_lo, G_hi> + <r_hi, H_lo> + cL*u        (:63-67)   -> one MSM over gathered points
//   rCommit = <l_hi, G_lo> + <r_lo, H_hi> + cR*u        (:69-73)   -> one MSM
//   gs''_i  = xInv*G_lo_i + x*G_hi_i                     (:81)      -> k_fold_points
//   hs''_i  = x*H_lo_i + xInv*H_hi_i                     (:82)      -> k_fold_points
//   P'      = x^2*L + xInv^2*R + P                       (:87-92)   -> k_lincomb2_add
// and the generator rescale hs'_i = hs_i * y^-i (MultiRangeProof/Prover.hs:107) -> k_scale_points.
// The scalar folds ls', rs' (:84-85) and the dot products cL, cR (:60-61) are O(n) Fr work and
// stay with the host control code (prover_host.hip).
#include <string.h>
#include "lanes29.h"
#include "engine.h"
#include "devio.h"
#include "glv.h"

namespace bp {

// out[i] = s_lo * lo[i] + s_hi * hi[i]  (two scalars shared by all i): GLV + joint table, one lane per i.
// tab: VT entries x n xyzz scratch, tab[(e-1)*n + i].
__global__ void __launch_bounds__(256) k_fold_points(const uint8_t* __restrict__ lo, const uint8_t* __restrict__ hi,
                                                    uint32_t n, u256 s_lo, u256 s_hi, xyzz* __restrict__ tab,
                                                    uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto store = [&](int e, const xyzz& v) { st_xyzz(&tab[(size_t)(e - 1) * n + i], v); };
  auto load = [&](int e) { return ld_xyzz(&tab[(size_t)(e - 1) * n + i]); };
  HalfPair h0 = glv_prep_point(s_lo, ld_affine(lo + (size_t)i * 64), 1, 2, store);
  HalfPair h1 = glv_prep_point(s_hi, ld_affine(hi + (size_t)i * 64), 4, 8, store);
  joint_table_composites<4>(store, load);
  xyzz acc = xyzz_inf();
#pragma unroll 1
  for (int b = 0; b < GLV_BITS; b++) {
    uint32_t idx = bitreg_next(h0.a) | (bitreg_next(h0.b) << 1) | (bitreg_next(h1.a) << 2) | (bitreg_next(h1.b) << 3);
    acc = xyzz_dbl(acc);
    if (idx) acc = xyzz_add(acc, load((int)idx));
  }
  st_affine(out + (size_t)i * 64, xyzz_to_affine(acc));
}

// out = k0*p0 + k1*p1 + p2 (single group element; one lane)
__global__ void __launch_bounds__(64) k_lincomb2_add(const uint8_t* __restrict__ p0, const uint8_t* __restrict__ p1,
                                                    const uint8_t* __restrict__ p2, u256 k0, u256 k1,
                                                    xyzz* __restrict__ tab, uint8_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  auto store = [&](int e, const xyzz& v) { st_xyzz(&tab[e - 1], v); };
  auto load = [&](int e) { return ld_xyzz(&tab[e - 1]); };
  HalfPair h0 = glv_prep_point(k0, ld_affine(p0), 1, 2, store);
  HalfPair h1 = glv_prep_point(k1, ld_affine(p1), 4, 8, store);
  joint_table_composites<4>(store, load);
  xyzz acc = xyzz_inf();
#pragma unroll 1
  for (int b = 0; b < GLV_BITS; b++) {
    uint32_t idx = bitreg_next(h0.a) | (bitreg_next(h0.b) << 1) | (bitreg_next(h1.a) << 2) | (bitreg_next(h1.b) << 3);
    acc = xyzz_dbl(acc);
    if (idx) acc = xyzz_add(acc, load((int)idx));
  }
  acc = xyzz_madd(acc, ld_affine(p2));
  st_affine(out, xyzz_to_affine(acc));
}

// ---- short multi-exponentiations on the latency path: one lane per scalar bit ---------------------------------
// A single proof's verifier walks one k0*p0 + k1*p1 + acc per round (mkChallenges,
// InnerProductProof/Verifier.hs:47-50) and a few short sums over the proof's own points, 129-step double-and-add
// chains that one lane runs in ~1.5 ms while the rest of the GPU idles.  The points are known before any
// challenge: k_bit_tables writes 2^j * p, j < 128, once per point (one chain of doublings for all of them side by
// side), and a product k * p is then one lane per bit (glv.h glv_bit_lane: at most one addition) and a
// workgroup sum.  Exact: the lanes' terms add up to the same group element.
// The chain of 127 doublings is what a single proof's verifier waits for first (x * S needs S's table), so a point is
// spread over a quad of lanes (lanes29.h: a doubling in three rounds of products, 2.7 us against 4.2 us); every lane
// stores its own coordinate of each multiple.
__global__ void __launch_bounds__(64) k_bit_tables(const uint8_t* __restrict__ pts, uint32_t n, xyzz* __restrict__ sh /*[n][128]*/) {
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, lane = threadIdx.x & 3u;
  if (i >= n) return;  // whole quads
  fe29 P = quad_spread(xyzz29_from(xyzz_from_affine(ld_affine(pts + (size_t)i * 64))), lane);
  quad_store(&sh[(size_t)i * GLV_LANE_BITS], P, lane);
#pragma unroll 1
  for (int j = 1; j < GLV_LANE_BITS; j++) {
    P = quad_dbl(P, lane);
    quad_store(&sh[(size_t)i * GLV_LANE_BITS + j], P, lane);
  }
}

// acc <- k0*p0 + k1*p1 + acc, p0 / p1 by their bit tables; 256 lanes, the result stays XYZZ (the host normalises)
__global__ void __launch_bounds__(256) k_bits_lincomb2(const xyzz* __restrict__ sh0, const xyzz* __restrict__ sh1, u256 k0,
                                                      u256 k1, xyzz* __restrict__ acc) {
  __shared__ uint32_t red[256 * 36];
  const int t = (int)threadIdx.x, side = t >> 7, j = t & (GLV_LANE_BITS - 1);
  xyzz term = bit_lane_term(side ? k1 : k0, side ? sh1 : sh0, j);
  xyzz sum = block_sum_quad(term, red);
  if (t == 0) st_xyzz(acc, xyzz_add(sum, ld_xyzz(acc)));
}

// out[b] = sum_t scalars[first_b + t] * point(first_b + t), t < count_b, for up to four sums b = blockIdx.x over
// consecutive tables; two terms per pass over the bit positions, one workgroup sum at the end
struct BitSums {
  uint32_t first[4], count[4];
};
__global__ void __launch_bounds__(256) k_bits_multiexp(const xyzz* __restrict__ sh, const uint8_t* __restrict__ scalars,
                                                      BitSums bs, xyzz* __restrict__ out) {
  __shared__ uint32_t red[256 * 36];
  const int t = (int)threadIdx.x, side = t >> 7, j = t & (GLV_LANE_BITS - 1);
  const uint32_t first = bs.first[blockIdx.x], count = bs.count[blockIdx.x];
  xyzz acc = xyzz_inf();
#pragma unroll 1
  for (uint32_t term = (uint32_t)side; term < count; term += 2) {
    u256 k = Fr::reduce_once(ld_u256(scalars + (size_t)(first + term) * 32));
    acc = xyzz_add(acc, bit_lane_term(k, sh + (size_t)(first + term) * GLV_LANE_BITS, j));
  }
  xyzz sum = block_sum_quad(acc, red);
  if (t == 0) st_xyzz(&out[blockIdx.x], sum);
}

// out[i] = scalars[i] * pts[i], independent scalars (canonical, 32 B each): GLV, one lane per i
__global__ void __launch_bounds__(256) k_scale_points(const uint8_t* __restrict__ pts, const uint8_t* __restrict__ scalars,
                                                     uint32_t n, xyzz* __restrict__ tab, uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto store = [&](int e, const xyzz& v) { st_xyzz(&tab[(size_t)(e - 1) * n + i], v); };
  auto load = [&](int e) { return ld_xyzz(&tab[(size_t)(e - 1) * n + i]); };
  u256 k = Fr::reduce_once(ld_u256(scalars + (size_t)i * 32));
  HalfPair h0 = glv_prep_point(k, ld_affine(pts + (size_t)i * 64), 1, 2, store);
  joint_table_composites<2>(store, load);
  xyzz acc = xyzz_inf();
#pragma unroll 1
  for (int b = 0; b < GLV_BITS; b++) {
    uint32_t idx = bitreg_next(h0.a) | (bitreg_next(h0.b) << 1);
    acc = xyzz_dbl(acc);
    if (idx) acc = xyzz_add(acc, load((int)idx));
  }
  st_affine(out + (size_t)i * 64, xyzz_to_affine(acc));
}

__device__ __forceinline__ void lincomb_lane(const LinJob& j, uint32_t i, xyzz* __restrict__ tab, size_t T, size_t t) {
  auto store = [&](int e, const xyzz& v) { st_xyzz(&tab[(size_t)(e - 1) * T + t], v); };
  auto load = [&](int e) { return ld_xyzz(&tab[(size_t)(e - 1) * T + t]); };
  u256 ka = j.sa ? Fr::reduce_once(ld_u256(j.sa + (size_t)i * 32)) : j.ka;
  u256 kb = j.sb ? Fr::reduce_once(ld_u256(j.sb + (size_t)i * 32)) : j.kb;
  HalfPair h0 = glv_prep_point(ka, ld_affine(j.a + (size_t)i * j.stride_a), 1, 2, store);
  HalfPair h1 = glv_prep_point(kb, ld_affine(j.b + (size_t)i * j.stride_b), 4, 8, store);
  joint_table_composites<4>(store, load);
  xyzz acc = xyzz_inf();
#pragma unroll 1
  for (int b = 0; b < GLV_BITS; b++) {
    uint32_t idx = bitreg_next(h0.a) | (bitreg_next(h0.b) << 1) | (bitreg_next(h1.a) << 2) | (bitreg_next(h1.b) << 3);
    acc = xyzz_dbl(acc);
    if (idx) acc = xyzz_add(acc, load((int)idx));
  }
  if (j.c) acc = xyzz_madd(acc, ld_affine(j.c + (size_t)i * j.stride_c));
  st_affine(j.out + (size_t)i * 64, xyzz_to_affine(acc));
}

__global__ void __launch_bounds__(256) k_lincomb_jobs(LinJob j0, LinJob j1, LinJob j2, uint32_t total,
                                                     xyzz* __restrict__ tab) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  // pick the lane's job by data, not by branch: lanes of different jobs share a wave in the late rounds
  const bool in0 = t < j0.n, in1 = t < j0.n + j1.n;
  LinJob j = in0 ? j0 : (in1 ? j1 : j2);
  uint32_t i = in0 ? t : (in1 ? t - j0.n : t - j0.n - j1.n);
  lincomb_lane(j, i, tab, total, t);
}

int ipp_lincomb_jobs(bp_ctx* ctx, const LinJob* jobs, int njobs, xyzz* d_tab) {
  LinJob j[3];
  memset((void*)j, 0, sizeof(j));
  uint32_t total = 0;
  for (int i = 0; i < njobs && i < 3; i++) {
    j[i] = jobs[i];
    total += jobs[i].n;
  }
  if (total == 0) return BP_OK;
  hipLaunchKernelGGL(k_lincomb_jobs, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, j[0], j[1], j[2], total,
                     d_tab);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int ipp_fold_points(bp_ctx* ctx, const uint8_t* d_lo, const uint8_t* d_hi, uint32_t n, const uint8_t s_lo[32],
                    const uint8_t s_hi[32], xyzz* d_tab, uint8_t* d_out) {
  u256 a, b;
  memcpy(a.v, s_lo, 32);
  memcpy(b.v, s_hi, 32);
  hipLaunchKernelGGL(k_fold_points, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_lo, d_hi, n, a, b, d_tab, d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int ipp_lincomb2_add(bp_ctx* ctx, const uint8_t* d_p0, const uint8_t* d_p1, const uint8_t* d_p2, const uint8_t k0[32],
                     const uint8_t k1[32], xyzz* d_tab, uint8_t* d_out) {
  u256 a, b;
  memcpy(a.v, k0, 32);
  memcpy(b.v, k1, 32);
  hipLaunchKernelGGL(k_lincomb2_add, dim3(1), dim3(64), 0, ctx->stream, d_p0, d_p1, d_p2, a, b, d_tab, d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int ipp_bit_tables(bp_ctx* ctx, const uint8_t* d_pts, uint32_t n, xyzz* d_sh, hipStream_t st) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(k_bit_tables, dim3((n + 15) / 16), dim3(64), 0, st, d_pts, n, d_sh);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int ipp_bits_lincomb2(bp_ctx* ctx, const xyzz* d_sh0, const xyzz* d_sh1, const uint8_t k0[32], const uint8_t k1[32],
                      xyzz* d_acc) {
  u256 a, b;
  memcpy(a.v, k0, 32);
  memcpy(b.v, k1, 32);
  hipLaunchKernelGGL(k_bits_lincomb2, dim3(1), dim3(256), 0, ctx->stream, d_sh0, d_sh1, a, b, d_acc);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int ipp_bits_multiexp(bp_ctx* ctx, const xyzz* d_sh, const uint8_t* d_scalars, const uint32_t* first, const uint32_t* count,
                      int nsums, xyzz* d_out) {
  if (nsums < 1 || nsums > 4) {
    ctx->err = "ipp_bits_multiexp: one to four sums";
    return BP_ERR_ARG;
  }
  BitSums bs;
  for (int i = 0; i < 4; i++) {
    bs.first[i] = i < nsums ? first[i] : 0;
    bs.count[i] = i < nsums ? count[i] : 0;
  }
  hipLaunchKernelGGL(k_bits_multiexp, dim3(nsums), dim3(256), 0, ctx->stream, d_sh, d_scalars, bs, d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int ipp_scale_points(bp_ctx* ctx, const uint8_t* d_pts, const uint8_t* d_scalars, uint32_t n, xyzz* d_tab,
                     uint8_t* d_out) {
  hipLaunchKernelGGL(k_scale_points, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_pts, d_scalars, n, d_tab,
                     d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

}  // namespace bp
