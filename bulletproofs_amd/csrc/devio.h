// Device-side load/store and cross-lane helpers for 256-bit values and points.
#pragma once
#include <hip/hip_runtime.h>
#include "ec.h"
#include "glv.h"

namespace bp {

__device__ __forceinline__ u256 ld_u256(const void* p) {
  const uint4* q = (const uint4*)p;
  uint4 a = q[0], b = q[1];
  u256 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void st_u256(void* p, const u256& r) {
  uint4* q = (uint4*)p;
  q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
  q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ affine ld_affine(const void* p) {
  affine a;
  a.x = ld_u256(p);
  a.y = ld_u256((const char*)p + 32);
  return a;
}
__device__ __forceinline__ void st_affine(void* p, const affine& a) {
  st_u256(p, a.x);
  st_u256((char*)p + 32, a.y);
}
__device__ __forceinline__ xyzz ld_xyzz(const xyzz* p) {
  xyzz r;
  r.X = ld_u256(&p->X);
  r.Y = ld_u256(&p->Y);
  r.ZZ = ld_u256(&p->ZZ);
  r.ZZZ = ld_u256(&p->ZZZ);
  return r;
}
__device__ __forceinline__ void st_xyzz(xyzz* p, const xyzz& r) {
  st_u256(&p->X, r.X);
  st_u256(&p->Y, r.Y);
  st_u256(&p->ZZ, r.ZZ);
  st_u256(&p->ZZZ, r.ZZZ);
}

__device__ __forceinline__ u256 shfl_down_u256(const u256& a, int d) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_down(a.v[i], d, 64);
  return r;
}
__device__ __forceinline__ xyzz shfl_down_xyzz(const xyzz& p, int d) {
  xyzz r;
  r.X = shfl_down_u256(p.X, d);
  r.Y = shfl_down_u256(p.Y, d);
  r.ZZ = shfl_down_u256(p.ZZ, d);
  r.ZZZ = shfl_down_u256(p.ZZZ, d);
  return r;
}
// Sum over the 64 lanes of a wave; lane 0 holds the result.
__device__ __forceinline__ xyzz wave_sum(xyzz v) {
#pragma unroll 1
  for (int d = 32; d > 0; d >>= 1) {
    xyzz o = shfl_down_xyzz(v, d);
    v = xyzz_add(v, o);
  }
  return v;
}
// Sum over the first `width` lanes of a wave (width a power of two <= 64); lane 0 holds the result.  Each step is a
// full XYZZ addition (~3,500 instructions) on the critical path, so sums over few values take few steps.
__device__ __forceinline__ xyzz wave_sum_width(xyzz v, int width) {
#pragma unroll 1
  for (int d = width >> 1; d > 0; d >>= 1) {
    xyzz o = shfl_down_xyzz(v, d);
    v = xyzz_add(v, o);
  }
  return v;
}
// Inclusive suffix scan over a workgroup of up to 256 threads: thread t gets the sum over threads t' >= t.
// Shuffle steps inside a wave (6 additions deep), then each wave adds the totals of the waves behind it.
// `lds` must hold blockDim.x/64 xyzz values.
__device__ __forceinline__ xyzz block_suffix_scan(xyzz v, xyzz* lds) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    xyzz o = shfl_down_xyzz(v, d);
    if (lane + d < 64) v = xyzz_add(v, o);
  }
  if (nw == 1) return v;
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  xyzz later = xyzz_inf();
  for (int w2 = nw - 1; w2 > wid; w2--) later = xyzz_add(later, lds[w2]);
  __syncthreads();
  return xyzz_add(v, later);
}

// This lane's term of k * point (k canonical, < r) from the point's bit table tab[j] = 2^j * point (glv.h glv_bit_lane)
__device__ __forceinline__ xyzz bit_lane_term(const u256& k, const xyzz* __restrict__ tab, int j) {
  GlvHalf h1, h2;
  glv_split(k, h1, h2);
  return glv_bit_lane(h1, h2, ld_xyzz(&tab[j]), j);
}

// Sum over a workgroup of up to 1024 threads; thread 0 holds the result.
// `lds` must hold blockDim.x/64 xyzz values.
__device__ __forceinline__ xyzz block_sum(xyzz v, xyzz* lds) {
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  if (nw == 1) return v;
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  xyzz r = xyzz_inf();
  if (wid == 0) {
    if (lane < nw) r = lds[lane];
    int width = 1;
    while (width < nw) width <<= 1;
    r = wave_sum_width(r, width);  // log2(waves) steps, not six
  }
  __syncthreads();
  return r;
}

}  // namespace bp
