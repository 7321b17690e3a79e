// Which kernels overlap across streams on this GPU?  (diagnostic, see DESIGN.md "streams")
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include "../bulletproofs_amd/csrc/ec.h"
using namespace bp;
__global__ void __launch_bounds__(256) k_busy(uint32_t* out, int iters) {
  affine g = affine_gen();
  xyzz acc = xyzz_from_affine(g);
  acc = xyzz_dbl(acc);
  for (int i = 0; i < iters; i++) acc = xyzz_madd(xyzz_dbl(acc), g);
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= acc.X.v[i] ^ acc.ZZ.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_spin(uint32_t* out, int iters) {
  uint32_t x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
  if (x == 0x12345u) out[0] = x;
}
__global__ void __launch_bounds__(256) k_fq(uint32_t* out, int iters) {  // field muls only, few registers
  u256 a = u256_one(), b = u256_one();
  a.v[0] = threadIdx.x + 3; b.v[1] = blockIdx.x + 5;
  for (int i = 0; i < iters; i++) { a = fq_mul(a, b); b = fq_mul(b, a); }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a.v[i] ^ b.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  const int NS = 8;
  hipStream_t st[NS];
  uint32_t* d[NS];
  for (int i = 0; i < NS; i++) { (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); (void)hipMalloc(&d[i], 1 << 22); }
  auto timeit = [&](int kind, int blocks, int a, int b) {
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int which = 0; which < 2; which++) {
        int s = which ? b : a;
        if (s < 0) continue;
        if (kind == 0) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[s], d[s], 100000);
        if (kind == 1) hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, st[s], d[s], 150);
        if (kind == 2) hipLaunchKernelGGL(k_fq, dim3(blocks), dim3(256), 0, st[s], d[s], 1500);
      }
      (void)hipStreamSynchronize(st[a]);
      if (b >= 0) (void)hipStreamSynchronize(st[b]);
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms < best) best = ms;
    }
    return best;
  };
  const char* names[3] = {"spin", "xyzz", "fqmul"};
  for (int kind = 0; kind < 3; kind++)
    for (int blocks : {16, 32, 64, 128, 256, 512}) {
      double t1 = timeit(kind, blocks, 0, -1);
      printf("%-6s blocks=%3d alone %.3f | (0,1) %.3f  (0,2) %.3f  (0,4) %.3f  (1,5) %.3f\n", names[kind], blocks, t1,
             timeit(kind, blocks, 0, 1), timeit(kind, blocks, 0, 2), timeit(kind, blocks, 0, 4), timeit(kind, blocks, 1, 5));
    }
  return 0;
}
