// Random 64-byte gathers from a large table: what the memory system sustains for the access shape of the fixed-base
// multi-exponentiation (one table entry per mixed addition).  ./gather_probe [table GiB] 
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// every lane: `iters` steps of DEPTH independent 64 B loads (4 x uint4) at hashed entry indices, `alu` dummy multiplies per step
template <int DEPTH>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ tab, uint64_t entries, int iters, int alu, uint32_t* out) {
  uint64_t h = mix(blockIdx.x * 256u + threadIdx.x + 1);
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint4 v[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      h = mix(h + d);
      const uint4* e = tab + (h % entries) * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) v[d][j] = e[j];
    }
    uint32_t x = (uint32_t)h | 1u;
    for (int a = 0; a < alu; a++) x = x * 2654435761u + 12345u;  // dependent ALU work between the loads and their use
    acc ^= x;
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc ^= v[d][j].x ^ v[d][j].y ^ v[d][j].z ^ v[d][j].w;
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  size_t gib = argc > 1 ? atol(argv[1]) : 60;
  size_t bytes = gib << 30;
  uint4* tab; uint32_t* out;
  CK(hipMalloc(&tab, bytes));
  CK(hipMemset(tab, 1, bytes));
  CK(hipMalloc(&out, 4u << 20));
  uint64_t entries = bytes / 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int lanes_k : {128, 256, 512}) {
    for (int alu : {0, 400, 1600}) {
      int blocks = lanes_k * 1024 / 256, iters = 200;
#define RUN(D) { hipLaunchKernelGGL(k_gather<D>, dim3(blocks), dim3(256), 0, 0, tab, entries, 10, alu, out); CK(hipDeviceSynchronize()); \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_gather<D>, dim3(blocks), dim3(256), 0, 0, tab, entries, iters, alu, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double n = (double)blocks * 256 * iters * D; \
        printf("{\"table_gib\": %zu, \"lanes\": %d, \"depth\": %d, \"alu\": %d, \"Ggather_per_s\": %.2f, \"GB_per_s\": %.0f, \"us_per_step\": %.2f}\n", gib, lanes_k * 1024, D, alu, n / ms / 1e6, n * 64 / ms / 1e6, ms * 1e3 / iters); }
      RUN(1) RUN(2)
    }
  }
  return 0;
}
