// gfx950 integer-VALU microbenchmarks: measures the roofline denominator
// (v_mad_u64_u32 issue rate) and the field / point primitive throughput.
// Build: hipcc -O3 --offload-arch=gfx950 -o microbench microbench.hip
#include <string.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../bulletproofs_amd/csrc/lanes29.h"
#include "../bulletproofs_amd/csrc/ec.h"
#include "../bulletproofs_amd/csrc/fq29.h"
using namespace bp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- raw instruction streams: NCH independent chains per lane --------------
template <int NCH>
__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x + i;
  uint32_t x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
__global__ void k_mullo(uint32_t* out, uint32_t a, int iters) {
  uint32_t acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
__global__ void k_mulhi(uint32_t* out, uint32_t a, int iters) {
  uint32_t acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
__global__ void k_mad24(uint32_t* out, uint32_t a, int iters) {
  uint32_t acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(a));
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
__global__ void k_addc(uint32_t* out, uint32_t a, int iters) {
  uint32_t acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : "vcc");
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
__global__ void k_fma64(double* out, double a, int iters) {
  double acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(a));
    }
  }
  double s = 0;
  for (int i = 0; i < NCH; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- field / point primitives ------------------------------------------------
__device__ __forceinline__ u256 seed_fe(uint32_t s) {
  u256 r;
  for (int i = 0; i < 8; i++) r.v[i] = s * 2654435761u + i * 0x9E3779B9u;
  return r;
}
__device__ __forceinline__ u256 fq_mul_rows(const u256& a, const u256& b) {
  uint32_t t[16];
  mul_wide_rows(t, a, b);
  return fq_reduce_wide(t);
}
// op 0: fq_mul  1: fq_sqr  2: FqMont::mul  3: Fr::mul  4: fq_add  5: fq_sub  6: fq_mul (row-wise arm)
template <int OP>
__global__ void __launch_bounds__(256) k_field(uint32_t* out, int iters) {
  u256 a = seed_fe(threadIdx.x + blockIdx.x * blockDim.x), b = seed_fe(threadIdx.x * 7 + 1);
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { a = fq_mul(a, b); b = fq_mul(b, a); }
    if (OP == 1) { a = fq_sqr(a); b = fq_sqr(b); }
    if (OP == 2) { a = FqMont::mul(a, b); b = FqMont::mul(b, a); }
    if (OP == 3) { a = Fr::mul(a, b); b = Fr::mul(b, a); }
    if (OP == 4) { a = fq_add(a, b); b = fq_add(b, a); }
    if (OP == 5) { a = fq_sub(a, b); b = fq_sub(b, a); }
    if (OP == 6) { a = fq_mul_rows(a, b); b = fq_mul_rows(b, a); }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a.v[i] ^ b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// op 0: madd chain  1: dbl chain  2: full add chain
template <int OP>
__global__ void __launch_bounds__(256) k_point(uint32_t* out, int iters) {
  affine g = affine_gen();
  u256 k = seed_fe(threadIdx.x + blockIdx.x * blockDim.x + 1);
  k.v[7] = 0; k.v[6] = 0; k.v[5] = 0; k.v[4] = 0; k.v[3] = 0; k.v[2] = 0; k.v[1] = 0; k.v[0] = (k.v[0] & 0xffff) + 2;
  xyzz acc = xyzz_mul_affine(g, k);  // distinct per lane
  xyzz q = xyzz_dbl(xyzz_dbl(acc));
  for (int it = 0; it < iters; it++) {
    if (OP == 0) acc = xyzz_madd(acc, g);
    if (OP == 1) acc = xyzz_dbl(acc);
    if (OP == 2) acc = xyzz_add(acc, q);
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= acc.X.v[i] ^ acc.Y.v[i] ^ acc.ZZ.v[i] ^ acc.ZZZ.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same chains on nine 29-bit limbs (fq29.h).  op 0: fe29_mul  1: fe29_sqr
template <int OP>
__global__ void __launch_bounds__(256) k_field29(uint32_t* out, int iters) {
  fe29 a = fe29_from_u256(seed_fe(threadIdx.x + blockIdx.x * blockDim.x)), b = fe29_from_u256(seed_fe(threadIdx.x * 7 + 1));
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { a = fe29_mul(a, b); b = fe29_mul(b, a); }
    if (OP == 1) { a = fe29_sqr(a); b = fe29_sqr(b); }
  }
  uint32_t s = 0;
  for (int i = 0; i < 9; i++) s ^= a.v[i] ^ b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_point29(uint32_t* out, int iters) {
  affine g = affine_gen();
  u256 k = seed_fe(threadIdx.x + blockIdx.x * blockDim.x + 1);
  k.v[7] = 0; k.v[6] = 0; k.v[5] = 0; k.v[4] = 0; k.v[3] = 0; k.v[2] = 0; k.v[1] = 0; k.v[0] = (k.v[0] & 0xffff) + 2;
  xyzz29 acc = xyzz29_from(xyzz_mul_affine(g, k));  // distinct per lane
  for (int it = 0; it < iters; it++) acc = xyzz29_madd(acc, g, (it & 1) != 0);
  xyzz r = xyzz29_to(acc);
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= r.X.v[i] ^ r.Y.v[i] ^ r.ZZ.v[i] ^ r.ZZZ.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// reference for k_point29's result: the same alternating chain through ec.h
__global__ void __launch_bounds__(256) k_point_alt(uint32_t* out, int iters) {
  affine g = affine_gen();
  u256 k = seed_fe(threadIdx.x + blockIdx.x * blockDim.x + 1);
  k.v[7] = 0; k.v[6] = 0; k.v[5] = 0; k.v[4] = 0; k.v[3] = 0; k.v[2] = 0; k.v[1] = 0; k.v[0] = (k.v[0] & 0xffff) + 2;
  xyzz acc = xyzz_mul_affine(g, k);
  for (int it = 0; it < iters; it++) acc = xyzz_madd(acc, (it & 1) ? affine_neg(g) : g);
  affine a = xyzz_to_affine(acc);
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a.x.v[i] ^ a.y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_point29_aff(uint32_t* out, int iters) {
  affine g = affine_gen();
  u256 k = seed_fe(threadIdx.x + blockIdx.x * blockDim.x + 1);
  k.v[7] = 0; k.v[6] = 0; k.v[5] = 0; k.v[4] = 0; k.v[3] = 0; k.v[2] = 0; k.v[1] = 0; k.v[0] = (k.v[0] & 0xffff) + 2;
  xyzz29 acc = xyzz29_from(xyzz_mul_affine(g, k));
  for (int it = 0; it < iters; it++) acc = xyzz29_madd(acc, g, (it & 1) != 0);
  affine a = xyzz_to_affine(xyzz29_to(acc));
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a.x.v[i] ^ a.y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- one point operation over a quad / a pair of lanes (lanes29.h), and the one-lane 29-bit forms beside them ----
// op 0: xyzz29_dbl  1: xyzz29_add  2: quad_dbl  3: quad_add  4: pair_dbl  5: pair_dbl + pair_madd  6: xyzz29_dbl + xyzz29_madd
// Every group of lanes (1, 4 or 2) works on its own point; the result is the affine x, y of lane group's point.
template <int OP>
__global__ void __launch_bounds__(256) k_lanes(uint32_t* out, uint8_t* scratch, int iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int G = (OP == 2 || OP == 3) ? 4 : ((OP == 4 || OP == 5) ? 2 : 1);
  const uint32_t grp = t / G, lane = t % G;
  affine g = affine_gen();
  u256 k = seed_fe(grp + 1);
  k.v[7] = 0; k.v[6] = 0; k.v[5] = 0; k.v[4] = 0; k.v[3] = 0; k.v[2] = 0; k.v[1] = 0; k.v[0] = (k.v[0] & 0xffff) + 2;
  const xyzz start = xyzz_mul_affine(g, k);
  const xyzz29 q29 = xyzz29_from(xyzz_dbl(xyzz_dbl(start)));
  st_affine(scratch + (size_t)t * 64, g);  // the mixed additions read their point from memory, like the kernels
  xyzz29 acc = xyzz29_from(start);
  if (OP == 0) for (int it = 0; it < iters; it++) acc = xyzz29_dbl(acc);
  if (OP == 1) for (int it = 0; it < iters; it++) acc = xyzz29_add(acc, q29);
  if (OP == 6) for (int it = 0; it < iters; it++) acc = xyzz29_madd(xyzz29_dbl(acc), ld_affine(scratch + (size_t)t * 64), false);
  if (OP == 2 || OP == 3) {
    fe29 P = quad_spread(acc, lane);
    const fe29 Q = quad_spread(q29, lane);
    for (int it = 0; it < iters; it++) P = OP == 2 ? quad_dbl(P, lane) : quad_add(P, Q, lane);
    acc = quad_gather(P);
  }
  if (OP == 4 || OP == 5) {
    const bool odd = lane != 0;
    pair29 P = pair_spread(acc, odd);
    for (int it = 0; it < iters; it++) {
      P = pair_dbl(P, odd);
      if (OP == 5) P = pair_madd(P, scratch + (size_t)t * 64, odd);
    }
    acc = pair_gather(P, odd);
  }
  affine a = xyzz_to_affine(xyzz29_to(acc));
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a.x.v[i] ^ (a.y.v[i] * 3u);
  out[t] = s;
}

template <class F>
static double time_ms(F launch, int reps = 3) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();  // warm
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, cus, prop.clockRate / 1000);
  void* buf; CK(hipMalloc(&buf, 64u << 20));
  const int threads = 256;
  if (argc > 1 && !strcmp(argv[1], "--lanes")) {
    // latency of ONE dependent point operation at one wave per SIMD (and two), one lane against a quad / a pair
    uint8_t* scr; CK(hipMalloc(&scr, (size_t)cus * 8 * 256 * 64));
    for (int wps = 1; wps <= 2; wps *= 2) {
      int blocks = cus * wps;
      const int iters = 300;
#define LN(name, OP) { double ms = time_ms([&] { hipLaunchKernelGGL((k_lanes<OP>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, scr, iters); }); \
      printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"us_per_op\": %.2f}\n", name, wps, ms * 1e3 / iters); }
      LN("xyzz29_dbl (one lane)", 0) LN("quad_dbl", 2) LN("pair_dbl", 4) LN("xyzz29_add (one lane)", 1) LN("quad_add", 3)
      LN("xyzz29_dbl + xyzz29_madd (one lane)", 6) LN("pair_dbl + pair_madd", 5)
#undef LN
    }
    // parity: the same chains end at the same affine points (compared group by group)
    const int blocks = 4, iters = 23;
    uint32_t* o[7];
    std::vector<uint32_t> h[7];
    for (int i = 0; i < 7; i++) { CK(hipMalloc(&o[i], blocks * 256 * 4)); h[i].resize(blocks * 256); }
    hipLaunchKernelGGL((k_lanes<0>), dim3(blocks), dim3(256), 0, 0, o[0], scr, iters);
    hipLaunchKernelGGL((k_lanes<1>), dim3(blocks), dim3(256), 0, 0, o[1], scr, iters);
    hipLaunchKernelGGL((k_lanes<2>), dim3(blocks), dim3(256), 0, 0, o[2], scr, iters);
    hipLaunchKernelGGL((k_lanes<3>), dim3(blocks), dim3(256), 0, 0, o[3], scr, iters);
    hipLaunchKernelGGL((k_lanes<4>), dim3(blocks), dim3(256), 0, 0, o[4], scr, iters);
    hipLaunchKernelGGL((k_lanes<5>), dim3(blocks), dim3(256), 0, 0, o[5], scr, iters);
    hipLaunchKernelGGL((k_lanes<6>), dim3(blocks), dim3(256), 0, 0, o[6], scr, iters);
    for (int i = 0; i < 7; i++) CK(hipMemcpy(h[i].data(), o[i], h[i].size() * 4, hipMemcpyDeviceToHost));
    auto cmp = [&](const char* name, int one, int many, int G) {
      size_t bad = 0, n = h[many].size() / G;
      for (size_t g2 = 0; g2 < n; g2++)
        for (int l = 0; l < G; l++) bad += h[many][g2 * G + l] != h[one][g2];
      printf("{\"check\": \"%s\", \"groups\": %zu, \"mismatches\": %zu}\n", name, n, bad);
    };
    cmp("quad_dbl chain == xyzz29_dbl chain", 0, 2, 4);
    cmp("quad_add chain == xyzz29_add chain", 1, 3, 4);
    cmp("pair_dbl chain == xyzz29_dbl chain", 0, 4, 2);
    cmp("pair_dbl + pair_madd chain == xyzz29_dbl + xyzz29_madd chain", 6, 5, 2);
    return 0;
  }
  for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
    int blocks = cus * wps;                // 256 threads = 4 waves = 1 wave/SIMD per block
    const int iters = 4000;
    double ops = (double)blocks * threads * iters * 8 * 4;
#define RAW(name, kern, T) { double ms = time_ms([&] { hipLaunchKernelGGL((kern<4>), dim3(blocks), dim3(threads), 0, 0, (T*)buf, 12345, 777, iters); }); \
      printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"Gops\": %.1f, \"cycles_per_wave_instr\": %.2f}\n", name, wps, ops / ms / 1e6, (double)wps * ms * 1e-3 * prop.clockRate * 1e3 / ((double)iters * 8 * 4 * wps)); }
    RAW("v_mad_u64_u32", k_mad64, uint64_t)
#undef RAW
#define RAW1(name, kern, T, arg) { double ms = time_ms([&] { hipLaunchKernelGGL((kern<4>), dim3(blocks), dim3(threads), 0, 0, (T*)buf, arg, iters); }); \
      printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"Gops\": %.1f, \"cycles_per_wave_instr\": %.2f}\n", name, wps, ops / ms / 1e6, ms * 1e-3 * prop.clockRate * 1e3 / ((double)iters * 8 * 4)); }
    RAW1("v_mul_lo_u32", k_mullo, uint32_t, 12345u)
    RAW1("v_mul_hi_u32", k_mulhi, uint32_t, 12345u)
    RAW1("v_mad_u32_u24", k_mad24, uint32_t, 12345u)
    RAW1("v_addc_co_u32", k_addc, uint32_t, 12345u)
    RAW1("v_fma_f64", k_fma64, double, 1.000001)
#undef RAW1
  }
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = cus * wps;
    const int iters = 2000;
#define FLD(name, OP) { double ms = time_ms([&] { hipLaunchKernelGGL((k_field<OP>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, iters); }); \
      double n = (double)blocks * threads * iters * 2; \
      printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"Gop_per_s\": %.2f, \"ns_per_wave_op\": %.1f}\n", name, wps, n / ms / 1e6, ms * 1e6 / (iters * 2.0)); }
    FLD("fq_mul", 0) FLD("fq_sqr", 1) FLD("fqmont_mul", 2) FLD("fr_mul", 3) FLD("fq_add", 4) FLD("fq_sub", 5) FLD("fq_mul_rows", 6)
#undef FLD
  }
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = cus * wps;
    const int iters = 500;
#define PT(name, OP) { double ms = time_ms([&] { hipLaunchKernelGGL((k_point<OP>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, iters); }); \
      double n = (double)blocks * threads * iters; \
      printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"Gop_per_s\": %.3f, \"us_per_wave_op\": %.2f}\n", name, wps, n / ms / 1e6, ms * 1e3 / iters); }
    PT("xyzz_madd", 0) PT("xyzz_dbl", 1) PT("xyzz_add", 2)
#undef PT
    { double ms = time_ms([&] { hipLaunchKernelGGL(k_point29, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, iters); });
      double n = (double)blocks * threads * iters;
      printf("{\"bench\": \"xyzz29_madd\", \"waves_per_simd\": %d, \"Gop_per_s\": %.3f, \"us_per_wave_op\": %.2f}\n", wps, n / ms / 1e6, ms * 1e3 / iters); }
  }
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = cus * wps;
    const int iters = 2000;
#define FLD29(name, OP) { double ms = time_ms([&] { hipLaunchKernelGGL((k_field29<OP>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, iters); }); \
      double n = (double)blocks * threads * iters * 2; \
      printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"Gop_per_s\": %.2f, \"ns_per_wave_op\": %.1f}\n", name, wps, n / ms / 1e6, ms * 1e6 / (iters * 2.0)); }
    FLD29("fe29_mul", 0) FLD29("fe29_sqr", 1)
#undef FLD29
  }
  {  // parity of the lazy chain against ec.h: same alternating +-G chain, affine results compared lane by lane
    const int blocks = 8, iters = 37;
    uint32_t *a, *b;
    CK(hipMalloc(&a, blocks * 256 * 4)); CK(hipMalloc(&b, blocks * 256 * 4));
    hipLaunchKernelGGL(k_point_alt, dim3(blocks), dim3(256), 0, 0, a, iters);
    hipLaunchKernelGGL(k_point29_aff, dim3(blocks), dim3(256), 0, 0, b, iters);
    std::vector<uint32_t> ha(blocks * 256), hb(blocks * 256);
    CK(hipMemcpy(ha.data(), a, ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, hb.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ha.size(); i++) bad += ha[i] != hb[i];
    printf("{\"check\": \"xyzz29_madd chain == xyzz_madd chain\", \"lanes\": %zu, \"mismatches\": %zu}\n", ha.size(), bad);
  }
  return 0;
}
