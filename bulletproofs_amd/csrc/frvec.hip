// Scalar-field vector kernels of the provers and of IPP.verifyProof on gfx950.
//
// The witness vectors of a proof stay resident in HBM from the first commitment to the last inner-product round;
// per round only the challenge (two kernel arguments) goes down and three points come back.  The per-element
// arithmetic is in frvec.h (shared with the CPU tests); this file holds the kernels, the workgroup reductions
// (wave shuffles + LDS) and the launch wrappers.  Reference:
//   powerVector, hadamard, dot                /root/reference/Bulletproofs/Utils.hs:50-69
//   computeLRPolys, computeTPoly              MultiRangeProof/Prover.hs:133-169
//   cL, cR, ls', rs' of an IPP round          InnerProductProof/Prover.hs:60-61,84-85
//   mkOtherExponents                          InnerProductProof/Verifier.hs:55-73
//   computePolynomials, evaluatePolynomial, multiplyPoly, vectorMatrixProduct
//                                             ArithmeticCircuit/Prover.hs:104-117, Internal.hs:123-152
// These are HBM-trivial, launch-latency-bound kernels (n <= 8,192 elements of 32 B): what matters is that there are
// few of them per round and that nothing crosses PCIe but the challenge.
#include "prover.h"
#include "devio.h"
#include "frvec.h"

namespace bp {

// ---- reductions ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u256 fr_wave_sum(u256 v) {
#pragma unroll 1
  for (int d = 32; d > 0; d >>= 1) v = Fr::add(v, shfl_down_u256(v, d));
  return v;
}
// sum over a workgroup of up to 1024 threads, valid in thread 0; lds: blockDim.x / 64 entries
__device__ __forceinline__ u256 fr_block_sum(u256 v, u256* lds) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = fr_wave_sum(v);
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  u256 r = u256_zero();
  if (wid == 0) {
    if (lane < nw) r = lds[lane];
    r = fr_wave_sum(r);
  }
  __syncthreads();
  return r;
}

// ---- generic vector kernels ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fr_from_bytes(const uint8_t* __restrict__ src, uint32_t n, u256* __restrict__ dst) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_u256(&dst[i], fr_from_plain(ld_u256(src + (size_t)i * 32)));
}
__global__ void __launch_bounds__(256) k_fr_to_bytes(const u256* __restrict__ src, uint32_t n, uint8_t* __restrict__ dst) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_u256(dst + (size_t)i * 32, Fr::from_mont(ld_u256(&src[i])));
}
__global__ void __launch_bounds__(256) k_fr_power_vector(u256 a, uint32_t n, u256* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_u256(&out[i], fr_power_elem(a, i));
}
__global__ void __launch_bounds__(256) k_fr_hadamard(const u256* __restrict__ a, const u256* __restrict__ b, uint32_t n,
                                                    u256* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_u256(&out[i], Fr::mul(ld_u256(&a[i]), ld_u256(&b[i])));
}
// out[0] = <a, b> (Montgomery form); one workgroup
__global__ void __launch_bounds__(1024) k_fr_dot(const u256* __restrict__ a, const u256* __restrict__ b, uint32_t n,
                                                u256* __restrict__ out) {
  __shared__ u256 lds[16];
  u256 acc = u256_zero();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc = Fr::add(acc, Fr::mul(ld_u256(&a[i]), ld_u256(&b[i])));
  acc = fr_block_sum(acc, lds);
  if (threadIdx.x == 0) st_u256(&out[0], acc);
}

// ---- range proof ---------------------------------------------------------------------------------------------
// The scalars of commitBitVectors (RangeProof/Internal.hs:126-149) for the fixed-base sums A and S:
//   sc[0] = [aL (nm), aR (nm), aBlinding], sc[1] = [sL (nm), sR (nm), sBlinding], canonical
__global__ void __launch_bounds__(256) k_rp_commit_scalars(uint32_t nm, const uint8_t* __restrict__ bits,
                                                          const uint8_t* __restrict__ s_l, const uint8_t* __restrict__ s_r,
                                                          u256 alpha, u256 rho, uint8_t* __restrict__ sc) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nt = (size_t)2 * nm + 1;
  if (i < nm) {
    u256 one = u256_one(), zero = u256_zero();
    u256 minus1 = Fr::modulus();
    minus1.v[0] -= 1;
    const bool bit = bits[i] != 0;
    st_u256(sc + (size_t)i * 32, bit ? one : zero);
    st_u256(sc + ((size_t)nm + i) * 32, bit ? zero : minus1);  // aR = aL - 1
    st_u256(sc + (nt + i) * 32, Fr::reduce_once(ld_u256(s_l + (size_t)i * 32)));
    st_u256(sc + (nt + nm + i) * 32, Fr::reduce_once(ld_u256(s_r + (size_t)i * 32)));
  }
  if (i == 0) {
    st_u256(sc + (size_t)2 * nm * 32, Fr::reduce_once(alpha));
    st_u256(sc + (nt + (size_t)2 * nm) * 32, Fr::reduce_once(rho));
  }
}

__global__ void __launch_bounds__(256) k_rp_polys(uint32_t n_bits, uint32_t nm, const uint8_t* __restrict__ bits,
                                                 const u256* __restrict__ s_r, u256 y, u256 z, u256* __restrict__ l0,
                                                 u256* __restrict__ r0, u256* __restrict__ r1) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nm) return;
  RpPolyElem e = rp_poly_elem(i, n_bits, bits[i] != 0, ld_u256(&s_r[i]), y, z);
  st_u256(&l0[i], e.l0);
  st_u256(&r0[i], e.r0);
  st_u256(&r1[i], e.r1);
}

// computeTPoly (MultiRangeProof/Prover.hs:160-169): t0 = <l0, r0>, t2 = <l1, r1>, t1 = <l0 + l1, r0 + r1> - t0 - t2
// = <l1, r0> + <l0, r1>.  out[0..3): canonical bytes.  One workgroup.
__global__ void __launch_bounds__(1024) k_rp_tpoly(uint32_t nm, const u256* __restrict__ l0, const u256* __restrict__ l1,
                                                  const u256* __restrict__ r0, const u256* __restrict__ r1,
                                                  u256* __restrict__ out) {
  __shared__ u256 lds[16];
  u256 t0 = u256_zero(), t1 = u256_zero(), t2 = u256_zero();
  for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) {
    u256 a0 = ld_u256(&l0[i]), a1 = ld_u256(&l1[i]), b0 = ld_u256(&r0[i]), b1 = ld_u256(&r1[i]);
    t0 = Fr::add(t0, Fr::mul(a0, b0));
    t2 = Fr::add(t2, Fr::mul(a1, b1));
    t1 = Fr::add(t1, Fr::add(Fr::mul(a1, b0), Fr::mul(a0, b1)));
  }
  t0 = fr_block_sum(t0, lds);
  t1 = fr_block_sum(t1, lds);
  t2 = fr_block_sum(t2, lds);
  if (threadIdx.x == 0) {
    st_u256(&out[0], Fr::from_mont(t0));
    st_u256(&out[1], Fr::from_mont(t1));
    st_u256(&out[2], Fr::from_mont(t2));
  }
}

// ls = l0 + x l1, rs = r0 + x r1   (MultiRangeProof/Prover.hs:90-91)
__global__ void __launch_bounds__(256) k_rp_eval(uint32_t nm, const u256* __restrict__ l0, const u256* __restrict__ l1,
                                                const u256* __restrict__ r0, const u256* __restrict__ r1, u256 x,
                                                u256* __restrict__ ls, u256* __restrict__ rs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nm) return;
  st_u256(&ls[i], Fr::add(ld_u256(&l0[i]), Fr::mul(x, ld_u256(&l1[i]))));
  st_u256(&rs[i], Fr::add(ld_u256(&r0[i]), Fr::mul(x, ld_u256(&r1[i]))));
}

// ---- inner-product argument, prover ------------------------------------------------------------------------------
// cG_i = 1, cH_i = hscale_i as plain-form coefficients of the original generators
__global__ void __launch_bounds__(256) k_ipp_init(uint32_t n, const u256* __restrict__ hscale, u256* __restrict__ cg,
                                                 u256* __restrict__ ch) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st_u256(&cg[i], u256_one());
  st_u256(&ch[i], Fr::from_mont(ld_u256(&hscale[i])));
}

// One workgroup: [fold the witness of length nk_prev to nk = nk_prev / 2 in place,] then over the current vectors
//   dot = <ls, rs>, cL = <lsLeft, rsRight>, cR = <lsRight, rsLeft>   (InnerProductProof/Prover.hs:60-61)
// and the three scalars on bH = ub * gen: sc[s][2n] = ub * {dot, cL, cR}, canonical.  sm[0] = dot (canonical),
// sm[1], sm[2] = ls[0], rs[0] (canonical: the proof's l, r once nk = 1).
__global__ void __launch_bounds__(1024) k_ipp_witness(uint32_t n, uint32_t nk_prev, uint32_t fold, u256 x, u256 xi, u256 ub,
                                                     u256* __restrict__ ls, u256* __restrict__ rs, uint8_t* __restrict__ sc,
                                                     u256* __restrict__ sm) {
  __shared__ u256 lds[16];
  uint32_t nk = nk_prev;
  if (fold) {
    nk = nk_prev / 2;
    for (uint32_t j = threadIdx.x; j < nk; j += blockDim.x) {
      u256 a = ld_u256(&ls[j]), b = ld_u256(&rs[j]);
      ipp_fold_elem(a, ld_u256(&ls[nk + j]), b, ld_u256(&rs[nk + j]), x, xi);
      st_u256(&ls[j], a);
      st_u256(&rs[j], b);
    }
    __syncthreads();
  }
  const uint32_t h = nk / 2;
  u256 dot = u256_zero(), cl = u256_zero(), cr = u256_zero();
  for (uint32_t j = threadIdx.x; j < nk; j += blockDim.x) {
    u256 a = ld_u256(&ls[j]), b = ld_u256(&rs[j]);
    dot = Fr::add(dot, Fr::mul(a, b));
    if (j < h) {
      cl = Fr::add(cl, Fr::mul(a, ld_u256(&rs[h + j])));
      cr = Fr::add(cr, Fr::mul(ld_u256(&ls[h + j]), b));
    }
  }
  dot = fr_block_sum(dot, lds);
  cl = fr_block_sum(cl, lds);
  cr = fr_block_sum(cr, lds);
  if (threadIdx.x == 0) {
    const size_t nt = (size_t)2 * n + 1;
    st_u256(sc + ((size_t)2 * n) * 32, Fr::from_mont(Fr::mul(ub, dot)));
    st_u256(sc + (nt + (size_t)2 * n) * 32, Fr::from_mont(Fr::mul(ub, cl)));
    st_u256(sc + (2 * nt + (size_t)2 * n) * 32, Fr::from_mont(Fr::mul(ub, cr)));
    st_u256(&sm[0], Fr::from_mont(dot));
    st_u256(&sm[1], Fr::from_mont(ld_u256(&ls[0])));
    st_u256(&sm[2], Fr::from_mont(ld_u256(&rs[0])));
  }
}

// [coefficients of the previous fold,] then the scalars of the round's three sums over the 2n original generators:
//   sc[s][i] (gs_i) and sc[s][n + i] (hs_i) for s = 0 (commitment), 1 (lCommit), 2 (rCommit); nk = current length.
__global__ void __launch_bounds__(256) k_ipp_scalars(uint32_t n, uint32_t nk_prev, uint32_t fold, u256 x, u256 xi,
                                                    const u256* __restrict__ ls, const u256* __restrict__ rs,
                                                    u256* __restrict__ cg, u256* __restrict__ ch, uint8_t* __restrict__ sc) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 g = ld_u256(&cg[i]), hh = ld_u256(&ch[i]);
  uint32_t nk = nk_prev;
  if (fold) {
    ipp_coeff_elem(i, nk_prev, g, hh, x, xi);
    st_u256(&cg[i], g);
    st_u256(&ch[i], hh);
    nk = nk_prev / 2;
  }
  IppScalars s = ipp_scalar_elem(i, nk, g, hh, [&](uint32_t j) { return ld_u256(&ls[j]); },
                                 [&](uint32_t j) { return ld_u256(&rs[j]); });
  const size_t nt = (size_t)2 * n + 1;
  st_u256(sc + (size_t)i * 32, s.c_g);
  st_u256(sc + ((size_t)n + i) * 32, s.c_h);
  st_u256(sc + (nt + i) * 32, s.l_g);
  st_u256(sc + (nt + n + i) * 32, s.l_h);
  st_u256(sc + (2 * nt + i) * 32, s.r_g);
  st_u256(sc + (2 * nt + n + i) * 32, s.r_h);
}

// ---- inner-product argument, verifier ------------------------------------------------------------------------------
// Scalars of cProof = l <s, Gs> + r <reverse s, Hs> + (l r) bH   (InnerProductProof/Verifier.hs:27-38), n = 2^lg:
//   sc[i] = l s_i, sc[n + (n - 1 - i)] = r s_i, sc[2n] = l r, canonical.  l_plain, r_plain: the proof's l, r reduced.
__global__ void __launch_bounds__(256) k_ipp_vscalars(uint32_t n, uint32_t lg, IppVArgs a, uint8_t* __restrict__ sc) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 s = ipp_other_exponent(i, lg, a.s0, a.c2_rev);
  st_u256(sc + (size_t)i * 32, Fr::mul(a.l_plain, s));
  st_u256(sc + ((size_t)n + (n - 1 - i)) * 32, Fr::mul(a.r_plain, s));
  if (i == 0) st_u256(sc + (size_t)2 * n * 32, Fr::mul(a.l_plain, Fr::to_mont(a.r_plain)));
}

// ---- arithmetic circuits -------------------------------------------------------------------------------------------
// out[c] = sum_r zs[r] * W[r][c]  (`vectorMatrixProduct`, ArithmeticCircuit/Internal.hs:146-152) for a dense q x n
// matrix of canonical scalars (bytes); columns n..n2 are zero (padCircuit, Internal.hs:76-89).  One lane per column.
__global__ void __launch_bounds__(256) k_ac_vec_mat(const u256* __restrict__ zs, const uint8_t* __restrict__ W, uint32_t q,
                                                   uint32_t n, uint32_t n2, u256* __restrict__ out) {
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n2) return;
  u256 acc = u256_zero();
  if (c < n)
    for (uint32_t r = 0; r < q; r++) {
      u256 e = ld_u256(W + ((size_t)r * n + c) * 32);
      if (!u256_is_zero(e)) acc = Fr::add(acc, Fr::mul(ld_u256(&zs[r]), fr_from_plain(e)));
    }
  st_u256(&out[c], acc);
}

__global__ void __launch_bounds__(256) k_ac_polys(uint32_t n2, const u256* __restrict__ a_l, const u256* __restrict__ a_r,
                                                 const u256* __restrict__ s_r, u256 y, u256 yinv,
                                                 const u256* __restrict__ zw_l, const u256* __restrict__ zw_r,
                                                 const u256* __restrict__ zw_o, u256* __restrict__ l1, u256* __restrict__ r0,
                                                 u256* __restrict__ r1, u256* __restrict__ r3) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  AcPolyElem e = ac_poly_elem(ld_u256(&a_l[i]), ld_u256(&a_r[i]), ld_u256(&s_r[i]), fr_power_elem(y, i),
                              fr_power_elem(yinv, i), ld_u256(&zw_l[i]), ld_u256(&zw_r[i]), ld_u256(&zw_o[i]));
  st_u256(&l1[i], e.l1);
  st_u256(&r0[i], e.r0);
  st_u256(&r1[i], e.r1);
  st_u256(&r3[i], e.r3);
}

// tPoly = multiplyPoly lPoly rPoly (ArithmeticCircuit/Internal.hs:130-139) with l = (0, l1, aO, sL), r = (r0, r1, 0, r3):
// coefficients 1..6, canonical, into out[1..7); out[0] = 0.  One workgroup.
__global__ void __launch_bounds__(1024) k_ac_tpoly(uint32_t n2, const u256* __restrict__ l1, const u256* __restrict__ a_o,
                                                  const u256* __restrict__ s_l, const u256* __restrict__ r0,
                                                  const u256* __restrict__ r1, const u256* __restrict__ r3,
                                                  u256* __restrict__ out) {
  __shared__ u256 lds[16];
  u256 t[6];
#pragma unroll
  for (int k = 0; k < 6; k++) t[k] = u256_zero();
  for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) {
    u256 L1 = ld_u256(&l1[i]), L2 = ld_u256(&a_o[i]), L3 = ld_u256(&s_l[i]);
    u256 R0 = ld_u256(&r0[i]), R1 = ld_u256(&r1[i]), R3 = ld_u256(&r3[i]);
    t[0] = Fr::add(t[0], Fr::mul(L1, R0));
    t[1] = Fr::add(t[1], Fr::add(Fr::mul(L1, R1), Fr::mul(L2, R0)));
    t[2] = Fr::add(t[2], Fr::add(Fr::mul(L2, R1), Fr::mul(L3, R0)));
    t[3] = Fr::add(t[3], Fr::add(Fr::mul(L1, R3), Fr::mul(L3, R1)));
    t[4] = Fr::add(t[4], Fr::mul(L2, R3));
    t[5] = Fr::add(t[5], Fr::mul(L3, R3));
  }
#pragma unroll 1
  for (int k = 0; k < 6; k++) {
    u256 s = fr_block_sum(t[k], lds);
    if (threadIdx.x == 0) st_u256(&out[1 + k], Fr::from_mont(s));
  }
  if (threadIdx.x == 0) st_u256(&out[0], u256_zero());
}

// ls = evaluatePolynomial lPoly x = x l1 + x^2 aO + x^3 sL, rs = r0 + x r1 + x^3 r3   (ArithmeticCircuit/Prover.hs:63-64)
__global__ void __launch_bounds__(256) k_ac_eval(uint32_t n2, const u256* __restrict__ l1, const u256* __restrict__ a_o,
                                                const u256* __restrict__ s_l, const u256* __restrict__ r0,
                                                const u256* __restrict__ r1, const u256* __restrict__ r3, u256 x, u256 x2,
                                                u256 x3, u256* __restrict__ ls, u256* __restrict__ rs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  st_u256(&ls[i], Fr::add(Fr::add(Fr::mul(x, ld_u256(&l1[i])), Fr::mul(x2, ld_u256(&a_o[i]))), Fr::mul(x3, ld_u256(&s_l[i]))));
  st_u256(&rs[i], Fr::add(Fr::add(ld_u256(&r0[i]), Fr::mul(x, ld_u256(&r1[i]))), Fr::mul(x3, ld_u256(&r3[i]))));
}

// The scalars of the circuit's three commitBitVector sums (ArithmeticCircuit/Internal.hs:102-107, Prover.hs:29-33):
//   sc[0] = [aL, aR, aiBlinding], sc[1] = [aO, 0.., aoBlinding], sc[2] = [sL, sR, sBlinding], canonical
__global__ void __launch_bounds__(256) k_ac_commit_scalars(uint32_t n2, const u256* __restrict__ a_l, const u256* __restrict__ a_r,
                                                          const u256* __restrict__ a_o, const u256* __restrict__ s_l,
                                                          const u256* __restrict__ s_r, u256 b0, u256 b1, u256 b2,
                                                          uint8_t* __restrict__ sc) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nt = (size_t)2 * n2 + 1;
  if (i < n2) {
    st_u256(sc + (size_t)i * 32, Fr::from_mont(ld_u256(&a_l[i])));
    st_u256(sc + ((size_t)n2 + i) * 32, Fr::from_mont(ld_u256(&a_r[i])));
    st_u256(sc + (nt + i) * 32, Fr::from_mont(ld_u256(&a_o[i])));
    st_u256(sc + (nt + n2 + i) * 32, u256_zero());
    st_u256(sc + (2 * nt + i) * 32, Fr::from_mont(ld_u256(&s_l[i])));
    st_u256(sc + (2 * nt + n2 + i) * 32, Fr::from_mont(ld_u256(&s_r[i])));
  }
  if (i == 0) {
    st_u256(sc + (size_t)2 * n2 * 32, Fr::reduce_once(b0));
    st_u256(sc + (nt + (size_t)2 * n2) * 32, Fr::reduce_once(b1));
    st_u256(sc + (2 * nt + (size_t)2 * n2) * 32, Fr::reduce_once(b2));
  }
}

// ---- launch wrappers (asynchronous on ctx->stream) ---------------------------------------------------------------
#define FV_GRID(n) dim3((unsigned)((((size_t)(n)) + 255) / 256)), dim3(256), 0, ctx->stream
#define FV_ONE dim3(1), dim3(1024), 0, ctx->stream

int frv_from_bytes(bp_ctx* ctx, const uint8_t* d_bytes, uint32_t n, u256* d_out) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(k_fr_from_bytes, FV_GRID(n), d_bytes, n, d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_to_bytes(bp_ctx* ctx, const u256* d_in, uint32_t n, uint8_t* d_bytes) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(k_fr_to_bytes, FV_GRID(n), d_in, n, d_bytes);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_power_vector(bp_ctx* ctx, const u256& a, uint32_t n, u256* d_out) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(k_fr_power_vector, FV_GRID(n), a, n, d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_hadamard(bp_ctx* ctx, const u256* a, const u256* b, uint32_t n, u256* out) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(k_fr_hadamard, FV_GRID(n), a, b, n, out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_dot(bp_ctx* ctx, const u256* a, const u256* b, uint32_t n, u256* out) {
  hipLaunchKernelGGL(k_fr_dot, FV_ONE, a, b, n, out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_rp_commit_scalars(bp_ctx* ctx, uint32_t nm, const uint8_t* bits, const uint8_t* s_l, const uint8_t* s_r,
                          const u256& alpha, const u256& rho, uint8_t* sc) {
  hipLaunchKernelGGL(k_rp_commit_scalars, FV_GRID(nm), nm, bits, s_l, s_r, alpha, rho, sc);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_rp_polys(bp_ctx* ctx, uint32_t n_bits, uint32_t nm, const uint8_t* bits, const u256* s_r, const u256& y,
                 const u256& z, u256* l0, u256* r0, u256* r1) {
  hipLaunchKernelGGL(k_rp_polys, FV_GRID(nm), n_bits, nm, bits, s_r, y, z, l0, r0, r1);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_rp_tpoly(bp_ctx* ctx, uint32_t nm, const u256* l0, const u256* l1, const u256* r0, const u256* r1, u256* out3) {
  hipLaunchKernelGGL(k_rp_tpoly, FV_ONE, nm, l0, l1, r0, r1, out3);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_rp_eval(bp_ctx* ctx, uint32_t nm, const u256* l0, const u256* l1, const u256* r0, const u256* r1, const u256& x,
                u256* ls, u256* rs) {
  hipLaunchKernelGGL(k_rp_eval, FV_GRID(nm), nm, l0, l1, r0, r1, x, ls, rs);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ipp_init(bp_ctx* ctx, uint32_t n, const u256* hscale, u256* cg, u256* ch) {
  hipLaunchKernelGGL(k_ipp_init, FV_GRID(n), n, hscale, cg, ch);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ipp_round(bp_ctx* ctx, uint32_t n, uint32_t nk_prev, bool fold, const u256& x, const u256& xi, const u256& ub,
                  u256* ls, u256* rs, u256* cg, u256* ch, uint8_t* sc, u256* sm, bool want_scalars) {
  hipLaunchKernelGGL(k_ipp_witness, FV_ONE, n, nk_prev, fold ? 1u : 0u, x, xi, ub, ls, rs, sc, sm);
  if (want_scalars)
    hipLaunchKernelGGL(k_ipp_scalars, FV_GRID(n), n, nk_prev, fold ? 1u : 0u, x, xi, ls, rs, cg, ch, sc);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ipp_vscalars(bp_ctx* ctx, uint32_t n, uint32_t lg, const IppVArgs& a, uint8_t* sc) {
  hipLaunchKernelGGL(k_ipp_vscalars, FV_GRID(n), n, lg, a, sc);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ac_vec_mat(bp_ctx* ctx, const u256* zs, const uint8_t* W, uint32_t q, uint32_t n, uint32_t n2, u256* out) {
  if (n2 == 0) return BP_OK;
  hipLaunchKernelGGL(k_ac_vec_mat, FV_GRID(n2), zs, W, q, n, n2, out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ac_polys(bp_ctx* ctx, uint32_t n2, const u256* a_l, const u256* a_r, const u256* s_r, const u256& y,
                 const u256& yinv, const u256* zw_l, const u256* zw_r, const u256* zw_o, u256* l1, u256* r0, u256* r1,
                 u256* r3) {
  if (n2 == 0) return BP_OK;
  hipLaunchKernelGGL(k_ac_polys, FV_GRID(n2), n2, a_l, a_r, s_r, y, yinv, zw_l, zw_r, zw_o, l1, r0, r1, r3);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ac_tpoly(bp_ctx* ctx, uint32_t n2, const u256* l1, const u256* a_o, const u256* s_l, const u256* r0,
                 const u256* r1, const u256* r3, u256* out7) {
  hipLaunchKernelGGL(k_ac_tpoly, FV_ONE, n2, l1, a_o, s_l, r0, r1, r3, out7);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ac_eval(bp_ctx* ctx, uint32_t n2, const u256* l1, const u256* a_o, const u256* s_l, const u256* r0,
                const u256* r1, const u256* r3, const u256& x, const u256& x2, const u256& x3, u256* ls, u256* rs) {
  if (n2 == 0) return BP_OK;
  hipLaunchKernelGGL(k_ac_eval, FV_GRID(n2), n2, l1, a_o, s_l, r0, r1, r3, x, x2, x3, ls, rs);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
int frv_ac_commit_scalars(bp_ctx* ctx, uint32_t n2, const u256* a_l, const u256* a_r, const u256* a_o, const u256* s_l,
                          const u256* s_r, const u256& b0, const u256& b1, const u256& b2, uint8_t* sc) {
  hipLaunchKernelGGL(k_ac_commit_scalars, FV_GRID(n2 ? n2 : 1), n2, a_l, a_r, a_o, s_l, s_r, b0, b1, b2, sc);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

}  // namespace bp
