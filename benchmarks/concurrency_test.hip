// Do small long-running kernels on different streams overlap on this GPU?  (diagnostic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#include "../bulletproofs_amd/csrc/ec.h"
using namespace bp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_busy(uint32_t* out, int iters) {
  affine g = affine_gen();
  xyzz acc = xyzz_from_affine(g);
  acc = xyzz_dbl(acc);
  for (int i = 0; i < iters; i++) acc = xyzz_madd(xyzz_dbl(acc), g);
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= acc.X.v[i] ^ acc.ZZ.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  const int NS = 8;
  hipStream_t st[NS];
  uint32_t* d[NS];
  uint8_t* h[NS];
  for (int i = 0; i < NS; i++) {
    CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    CK(hipMalloc(&d[i], 1 << 22));
    CK(hipHostMalloc(&h[i], 1 << 22, hipHostMallocDefault));
  }
  auto run = [&](int ns, int blocks, bool copies, const char* label) {
    for (int rep = 0; rep < 2; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < ns; i++) {
        if (copies) (void)hipMemcpyAsync(d[i], h[i], 1 << 20, hipMemcpyHostToDevice, st[i]);
        hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, st[i], d[i], 300);
        if (copies) (void)hipMemcpyAsync(h[i], d[i], 1 << 20, hipMemcpyDeviceToHost, st[i]);
      }
      for (int i = 0; i < ns; i++) (void)hipStreamSynchronize(st[i]);
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("%-44s streams=%d blocks/stream=%d copies=%d : %.3f ms\n", label, ns, blocks, (int)copies, ms);
    }
  };
  // which pairs of streams overlap?  (streams map onto a few hardware queues / pipes)
  for (int a = 0; a < NS; a++)
    for (int b = a + 1; b < NS; b++) {
      double best = 1e9;
      for (int rep = 0; rep < 2; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_busy, dim3(128), dim3(256), 0, st[a], d[a], 300);
        hipLaunchKernelGGL(k_busy, dim3(128), dim3(256), 0, st[b], d[b], 300);
        (void)hipStreamSynchronize(st[a]);
        (void)hipStreamSynchronize(st[b]);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best) best = ms;
      }
      printf("pair (%d,%d): %.3f ms\n", a, b, best);
    }
  run(1, 64, false, "one small kernel");
  run(4, 64, false, "four small kernels on four streams");
  run(1, 256, false, "one kernel with 4x the blocks");
  run(1, 64, true, "one small kernel + H2D/D2H");
  run(4, 64, true, "four small kernels + H2D/D2H");
  run(2, 128, true, "two kernels + copies");
  run(2, 128, false, "two kernels");
  return 0;
}
