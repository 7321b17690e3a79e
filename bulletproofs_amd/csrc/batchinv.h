// Simultaneous inversion across a workgroup (Montgomery's trick over lanes).
//
// The verifier inverts one scalar-field and one base-field element per proof and round (x_k^-1 for
// mkChallenges, /root/reference/Bulletproofs/InnerProductProof/Verifier.hs:47-50, and the affine form of
// the running commitment that the next challenge hashes); the reference calls `recip` / the affine group
// law once per element.  With one proof per lane every lane of a wave would run the same inversion (~21 k
// instructions by division steps, modinv30.h; ~66 k by Kaliski's almost-inverse before that).  Here the workgroup multiplies its elements together (prefix and suffix products by wave
// shuffles, wave totals through LDS), ONE wave inverts the product, and every lane gets its own inverse as
//   1/x_i = (1 / prod) * (prod of the elements before i) * (prod of the elements after i):
// about 20 multiplications per lane; the other waves wait at a barrier and leave their issue slots to
// whatever else is resident on the SIMD.  The values are exact: (1/x_i) mod m is unique.
#pragma once
#include <hip/hip_runtime.h>
#include "fe.h"
#include "fq29.h"
#include "modinv30.h"

namespace bp {

struct FqInvOps {
  static __device__ __forceinline__ u256 mul(const u256& a, const u256& b) { return fq_mul(a, b); }
  static __device__ __forceinline__ u256 inv(const u256& a) { return fq_inv_gcd(a); }  // division steps (modinv30.h)
  static __device__ __forceinline__ u256 one() { return u256_one(); }
  static __device__ __forceinline__ bool is_zero(const u256& a) { return fq_is_zero(a); }
};
struct FrInvOps {  // Montgomery form in, Montgomery form out
  static __device__ __forceinline__ u256 mul(const u256& a, const u256& b) { return Fr::mul(a, b); }
  static __device__ __forceinline__ u256 inv(const u256& a) { return fr_inv_gcd(a); }
  static __device__ __forceinline__ u256 one() { return Fr::one(); }
  static __device__ __forceinline__ bool is_zero(const u256& a) { return u256_is_zero(a); }
};

__device__ __forceinline__ u256 shfl_up_u256(const u256& a, int d) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_up(a.v[i], d, 64);
  return r;
}
__device__ __forceinline__ u256 shfl_dn_u256(const u256& a, int d) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_down(a.v[i], d, 64);
  return r;
}

constexpr int BATCHINV_MAX_WAVES = 16;
constexpr int BATCHINV_LDS = BATCHINV_MAX_WAVES + 1;  // u256 slots the caller provides

// Every thread of the workgroup must call this (threads without an element pass one()); blockDim.x is a
// multiple of 64, at most 1024.  Returns 1/x, and 0 for x = 0 (as fq_inv / Fr::inv do).
template <class F>
__device__ __forceinline__ u256 block_batch_inverse(const u256& x, u256* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool zero = F::is_zero(x);
  const u256 v = zero ? F::one() : x;
  // inclusive prefix and suffix products inside the wave
  u256 pre = v, suf = v;
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    u256 a = shfl_up_u256(pre, d);
    u256 b = shfl_dn_u256(suf, d);
    u256 pa = F::mul(pre, a), sb = F::mul(suf, b);
    if (lane >= d) pre = pa;
    if (lane + d < 64) suf = sb;
  }
  u256 pex = shfl_up_u256(pre, 1), sex = shfl_dn_u256(suf, 1);  // products of the lanes before / after this one
  if (lane == 0) pex = F::one();
  if (lane == 63) {
    sex = F::one();
    lds[wave] = pre;  // the wave's total
  }
  __syncthreads();
  if (wave == 0) {
    u256 g = lds[0];
    for (int w = 1; w < nw; w++) g = F::mul(g, lds[w]);
    u256 gi = F::inv(g);  // all lanes of this wave hold the same value; the other waves wait below
    if (lane == 0) lds[BATCHINV_MAX_WAVES] = gi;
  }
  __syncthreads();
  u256 r = lds[BATCHINV_MAX_WAVES];
  for (int w = 0; w < nw; w++)
    if (w != wave) r = F::mul(r, lds[w]);  // wave-uniform
  r = F::mul(F::mul(r, pex), sex);
  __syncthreads();  // lds may be reused by the next call
  if (zero) r = u256_zero();
  return r;
}

}  // namespace bp
