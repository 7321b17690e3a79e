// Element-wise group / field operations behind the C ABI: batches of independent
// `mul`, `<>`, `inv`, `addTwoMulP`, `def` and field-law hooks.
// Reference call sites: /root/reference/Bulletproofs/Utils.hs:72-73,84-85,
// RangeProof/Internal.hs:183-205, InnerProductProof/Prover.hs:81-92, test/Test/Field.hs.
#include "engine.h"
#include "verify.h"
#include "devio.h"

namespace bp {

__global__ void __launch_bounds__(256) k_point_mul(const uint8_t* __restrict__ scalars,
                                                  const uint8_t* __restrict__ points, uint32_t n,
                                                  uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 k = Fr::reduce_once(ld_u256(scalars + (size_t)i * 32));
  affine p = ld_affine(points + (size_t)i * 64);
  st_affine(out + (size_t)i * 64, xyzz_to_affine(xyzz_mul_affine(p, k)));
}

__global__ void __launch_bounds__(256) k_point_add(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                  uint32_t n, uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine p = ld_affine(a + (size_t)i * 64), q = ld_affine(b + (size_t)i * 64);
  st_affine(out + (size_t)i * 64, xyzz_to_affine(xyzz_madd(xyzz_from_affine(p), q)));
}

__global__ void __launch_bounds__(256) k_point_neg(const uint8_t* __restrict__ a, uint32_t n,
                                                  uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine p = affine_neg(ld_affine(a + (size_t)i * 64));
  p.x = fq_canon(p.x);
  p.y = fq_canon(p.y);
  st_affine(out + (size_t)i * 64, p);
}

__global__ void __launch_bounds__(256) k_add_two_mul(const uint8_t* __restrict__ e0, const uint8_t* __restrict__ p0,
                                                    const uint8_t* __restrict__ e1, const uint8_t* __restrict__ p1,
                                                    uint32_t n, uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 k0 = Fr::reduce_once(ld_u256(e0 + (size_t)i * 32)), k1 = Fr::reduce_once(ld_u256(e1 + (size_t)i * 32));
  affine a = ld_affine(p0 + (size_t)i * 64), b = ld_affine(p1 + (size_t)i * 64);
  // interleaved double-and-add (shared doublings)
  xyzz acc = xyzz_inf();
  for (int j = 255; j >= 0; j--) {
    acc = xyzz_dbl(acc);
    if (u256_pop_msb(k0)) acc = xyzz_madd(acc, a);
    if (u256_pop_msb(k1)) acc = xyzz_madd(acc, b);
  }
  st_affine(out + (size_t)i * 64, xyzz_to_affine(acc));
}

// out = sum of n affine points; one workgroup (the per-GPU partial sums of a sharded MSM: n is the number of GPUs)
__global__ void __launch_bounds__(256) k_point_sum(const uint8_t* __restrict__ pts, uint32_t n, uint8_t* __restrict__ out) {
  __shared__ xyzz lds[4];
  xyzz acc = xyzz_inf();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc = xyzz_madd(acc, ld_affine(pts + (size_t)i * 64));
  xyzz r = block_sum(acc, lds);
  if (threadIdx.x == 0) st_affine(out, xyzz_to_affine(r));
}

__global__ void __launch_bounds__(256) k_on_curve(const uint8_t* __restrict__ points, uint32_t n,
                                                 uint8_t* __restrict__ ok) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // O, or canonical coordinates (x, y < p: an `Fq` is always reduced) that satisfy the curve equation
  const affine a = ld_affine(points + (size_t)i * 64);
  const u256 q = fq_p();
  ok[i] = (affine_is_inf(a) || (!u256_geq(a.x, q) && !u256_geq(a.y, q) && affine_on_curve(a))) ? 1 : 0;
}

// *bad |= 1 when some point is neither O nor a canonically encoded point of the curve (the caller clears *bad)
__global__ void __launch_bounds__(256) k_points_invalid(const uint8_t* __restrict__ points, uint32_t n,
                                                       uint32_t* __restrict__ bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const affine a = ld_affine(points + (size_t)i * 64);
  const u256 q = fq_p();
  if (!(affine_is_inf(a) || (!u256_geq(a.x, q) && !u256_geq(a.y, q) && affine_on_curve(a)))) atomicOr(bad, 1u);
}

template <class F>
__device__ __forceinline__ u256 mont_op(int op, const u256& a, const u256& b) {
  u256 x = F::to_mont(a), y = F::to_mont(b), r;
  switch (op) {
    case 0: r = F::add(x, y); break;
    case 1: r = F::sub(x, y); break;
    case 2: r = F::mul(x, y); break;
    case 3: r = F::sqr(x); break;
    case 4: r = F::inv(x); break;
    default: r = F::neg(x); break;
  }
  return F::from_mont(r);
}

__global__ void __launch_bounds__(256) k_field_op(int field, int op, const uint8_t* __restrict__ a,
                                                 const uint8_t* __restrict__ b, uint32_t n,
                                                 uint8_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 x = ld_u256(a + (size_t)i * 32), y = ld_u256(b + (size_t)i * 32), r;
  if (field == 0) {
    switch (op) {
      case 0: r = fq_add(x, y); break;
      case 1: r = fq_sub(x, y); break;
      case 2: r = fq_mul(x, y); break;
      case 3: r = fq_sqr(x); break;
      case 4: r = fq_inv(x); break;
      case 6: r = fq_sqrt_candidate(x); break;
      default: r = fq_neg(x); break;
    }
    r = fq_canon(r);
  } else if (field == 1) {
    r = mont_op<Fr>(op, x, y);
  } else {
    r = mont_op<FqMont>(op, x, y);
  }
  st_u256(out + (size_t)i * 32, r);
}

}  // namespace bp

using namespace bp;

// Upload `nin` host arrays, run fn(device pointers, device out), download out.
template <class Fn>
static int with_staging(bp_ctx* ctx, std::initializer_list<std::pair<const void*, size_t>> ins, void* out,
                        size_t out_bytes, Fn fn) {
  size_t total = 0;
  std::vector<size_t> offs;
  for (auto& in : ins) {
    offs.push_back(total);
    total += (in.second + 255) & ~(size_t)255;
  }
  size_t out_off = total;
  total += (out_bytes + 255) & ~(size_t)255;
  if (ctx->io.reserve(total) != 0) {
    ctx->err = "staging allocation failed";
    return BP_ERR_NOMEM;
  }
  char* base = (char*)ctx->io.p;
  std::vector<const uint8_t*> dptr;
  size_t k = 0;
  for (auto& in : ins) {
    if (in.second) BP_TRY(hipMemcpyAsync(base + offs[k], in.first, in.second, hipMemcpyHostToDevice, ctx->stream));
    dptr.push_back((const uint8_t*)(base + offs[k]));
    k++;
  }
  int rc = fn(dptr, (uint8_t*)(base + out_off));
  if (rc != BP_OK) return rc;
  if (out_bytes) BP_TRY(hipMemcpyAsync(out, base + out_off, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  BP_TRY(hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

#define BP_CHECK_N(n) BP_REQUIRE((n) < (1ull << 31), "n must be below 2^31")
#define GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, ctx->stream

namespace bp {
// device pointers, asynchronous on ctx->stream: *d_bad (one u32) = 1 iff some of the n points has an invalid encoding
int points_invalid_dev(bp_ctx* ctx, const uint8_t* d_points, size_t n, uint32_t* d_bad) {
  BP_TRY(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), ctx->stream));
  if (n == 0) return BP_OK;
  BP_CHECK_N(n);
  hipLaunchKernelGGL(k_points_invalid, GRID(n), d_points, (uint32_t)n, d_bad);
  BP_TRY(hipGetLastError());
  return BP_OK;
}
}  // namespace bp

extern "C" {

int bp_point_mul(bp_ctx* ctx, const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (scalars && points && out), "null pointer");
  BP_CHECK_N(n);
  if (n == 0) return BP_OK;
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{scalars, n * 32}, {points, n * 64}}, out, n * 64,
                      [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
                        hipLaunchKernelGGL(k_point_mul, GRID(n), d[0], d[1], (uint32_t)n, dout);
                        BP_TRY(hipGetLastError());
                        return BP_OK;
                      });
}

int bp_point_add(bp_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (a && b && out), "null pointer");
  BP_CHECK_N(n);
  if (n == 0) return BP_OK;
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{a, n * 64}, {b, n * 64}}, out, n * 64,
                      [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
                        hipLaunchKernelGGL(k_point_add, GRID(n), d[0], d[1], (uint32_t)n, dout);
                        BP_TRY(hipGetLastError());
                        return BP_OK;
                      });
}

int bp_point_neg(bp_ctx* ctx, const uint8_t* a, size_t n, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (a && out), "null pointer");
  BP_CHECK_N(n);
  if (n == 0) return BP_OK;
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{a, n * 64}}, out, n * 64, [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
    hipLaunchKernelGGL(k_point_neg, GRID(n), d[0], (uint32_t)n, dout);
    BP_TRY(hipGetLastError());
    return BP_OK;
  });
}

int bp_add_two_mul(bp_ctx* ctx, const uint8_t* e0, const uint8_t* p0, const uint8_t* e1, const uint8_t* p1,
                   size_t n, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (e0 && p0 && e1 && p1 && out), "null pointer");
  BP_CHECK_N(n);
  if (n == 0) return BP_OK;
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{e0, n * 32}, {p0, n * 64}, {e1, n * 32}, {p1, n * 64}}, out, n * 64,
                      [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
                        hipLaunchKernelGGL(k_add_two_mul, GRID(n), d[0], d[1], d[2], d[3], (uint32_t)n, dout);
                        BP_TRY(hipGetLastError());
                        return BP_OK;
                      });
}

int bp_point_on_curve(bp_ctx* ctx, const uint8_t* points, size_t n, uint8_t* ok) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (points && ok), "null pointer");
  BP_CHECK_N(n);
  if (n == 0) return BP_OK;
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{points, n * 64}}, ok, n, [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
    hipLaunchKernelGGL(k_on_curve, GRID(n), d[0], (uint32_t)n, dout);
    BP_TRY(hipGetLastError());
    return BP_OK;
  });
}

int bp_field_op(bp_ctx* ctx, int field, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (a && b && out), "null pointer");
  BP_REQUIRE(field >= 0 && field <= 2 && op >= 0 && op <= 6 && (op != 6 || field == 0), "bad field/op selector");
  BP_CHECK_N(n);
  if (n == 0) return BP_OK;
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{a, n * 32}, {b, n * 32}}, out, n * 32,
                      [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
                        hipLaunchKernelGGL(k_field_op, GRID(n), field, op, d[0], d[1], (uint32_t)n, dout);
                        BP_TRY(hipGetLastError());
                        return BP_OK;
                      });
}

int bp_show_numbers(bp_ctx* ctx, const uint8_t* numbers, size_t n, uint8_t* records) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (numbers && records), "null pointer");
  BP_CHECK_N(n);
  if (n == 0) return BP_OK;
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{numbers, n * 32}}, records, n * 80, [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
    hipLaunchKernelGGL(k_dec_numbers, GRID(n), d[0], (uint32_t)n, dout);
    BP_TRY(hipGetLastError());
    return BP_OK;
  });
}

int bp_msm_dev(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits, void* d_out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(d_out && (n == 0 || (d_scalars && d_points)), "null pointer");
  BP_REQUIRE(window_bits >= 0 && window_bits <= 16, "window_bits must be 0 (auto) or 1..16");
  BP_TRY(hipSetDevice(ctx->device));
  return msm_run(ctx, d_scalars, d_points, n, window_bits, d_out, nullptr);
}

int bp_msm_dev_hs(bp_ctx* ctx, const uint8_t* h_scalars, const void* d_points, size_t n, int window_bits, void* d_out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(d_out && (n == 0 || (h_scalars && d_points)), "null pointer");
  BP_REQUIRE(window_bits >= 0 && window_bits <= 16, "window_bits must be 0 (auto) or 1..16");
  BP_TRY(hipSetDevice(ctx->device));
  return msm_run_host_scalars(ctx, h_scalars, d_points, n, window_bits, d_out);
}

int bp_point_sum_dev(bp_ctx* ctx, const void* d_points, size_t n, void* d_out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(d_out && (n == 0 || d_points), "null pointer");
  BP_CHECK_N(n);
  BP_TRY(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_point_sum, dim3(1), dim3(n <= 64 ? 64 : 256), 0, ctx->stream, (const uint8_t*)d_points, (uint32_t)n,
                     (uint8_t*)d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int bp_msm_dev_static(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits,
                      void* d_out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(d_out && (n == 0 || (d_scalars && d_points)), "null pointer");
  BP_REQUIRE(window_bits >= 0 && window_bits <= 16, "window_bits must be 0 (auto) or 1..16");
  BP_REQUIRE(n <= BP_MSM_STATIC_MAX_N, "bp_msm_dev_static: n exceeds BP_MSM_STATIC_MAX_N (use bp_msm_dev)");
  BP_TRY(hipSetDevice(ctx->device));
  return msm_run_static(ctx, d_scalars, d_points, n, window_bits, d_out);
}

int bp_msm_points_create(bp_ctx* ctx, const uint8_t* points, size_t n, bp_msm_points** out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(points && out, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  return msm_points_create(ctx, points, nullptr, n, (MsmPoints**)out);
}

int bp_msm_points_create_dev(bp_ctx* ctx, const void* d_points, size_t n, bp_msm_points** out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(d_points && out, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  return msm_points_create(ctx, nullptr, d_points, n, (MsmPoints**)out);
}

void bp_msm_points_destroy(bp_msm_points* mp) { msm_points_destroy((MsmPoints*)mp); }

int bp_msm_points_run(bp_ctx* ctx, bp_msm_points* mp, const uint8_t* scalars, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(mp && scalars && out, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  return msm_points_run((MsmPoints*)mp, nullptr, scalars, nullptr, out);
}

int bp_msm_points_run_dev(bp_ctx* ctx, bp_msm_points* mp, const void* d_scalars, void* d_out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(mp && d_scalars && d_out, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  return msm_points_run((MsmPoints*)mp, d_scalars, nullptr, d_out, nullptr);
}

int bp_msm(bp_ctx* ctx, const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(out && (n == 0 || (scalars && points)), "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  return with_staging(ctx, {{scalars, n * 32}, {points, n * 64}}, out, 64,
                      [&](std::vector<const uint8_t*>& d, uint8_t* dout) {
                        return msm_run(ctx, d[0], d[1], n, 0, dout, nullptr);
                      });
}

}  // extern "C"
