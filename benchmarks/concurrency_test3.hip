// Staggered launches with and without async copies on same-pipe streams (diagnostic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
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
int main() {
  const int NS = 16;
  hipStream_t st[NS];
  uint32_t* d[NS];
  uint8_t* h[NS];
  for (int i = 0; i < NS; i++) {
    (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    (void)hipMalloc(&d[i], 1 << 22);
    (void)hipHostMalloc(&h[i], 1 << 22, hipHostMallocDefault);
  }
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (int copies = 0; copies < 4; copies++)
    for (int stride : {1, 4}) {
      for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        double done[4];
        for (int c = 0; c < 4; c++) {
          int s = c * stride;
          if (copies >= 1) (void)hipMemcpyAsync(d[s], h[s], 1 << 19, hipMemcpyHostToDevice, st[s]);
          hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, st[s], copies == 3 ? (uint32_t*)h[s] : d[s] + (1 << 18), 220);
          if (copies == 2) (void)hipMemcpyAsync(h[s] + (1 << 20), d[s] + (1 << 18), 1 << 20, hipMemcpyDeviceToHost, st[s]);
          std::this_thread::sleep_for(std::chrono::microseconds(500));
        }
        for (int c = 0; c < 4; c++) {
          (void)hipStreamSynchronize(st[c * stride]);
          done[c] = ms(t0, now());
        }
        if (rep) printf("copies=%d stride=%d: chunk completion times %.2f %.2f %.2f %.2f ms (kernel alone ~2 ms, launches 0.5 ms apart)\n",
                        copies, stride, done[0], done[1], done[2], done[3]);
      }
    }
  return 0;
}
