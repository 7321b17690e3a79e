// Generator state of a context: the device-resident mirror of the reference's
// lazy CAF lists `gs`, `hs` and the points `h`, `gen`
// (/root/reference/Bulletproofs/Utils.hs:14-25), plus the fixed-base window
// tables (fixedbase.h) used by the batch verifier and by bp_commit.
#include "engine.h"
#include "devio.h"
#include "fixedbase.h"

namespace bp {

// bases[b] (affine) for b in [0, nbases): gs, hs, G, H, then the prefix sums GS_k.
__global__ void __launch_bounds__(256) k_gsum(const uint8_t* __restrict__ gs, uint32_t ng, uint32_t nk,
                                             uint8_t* __restrict__ out) {
  // one thread per k: GS_k = sum_{i < 2^k} gs_i  (tiny, one-time)
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nk) return;
  uint32_t cnt = 1u << k;
  xyzz acc = xyzz_inf();
  for (uint32_t i = 0; i < cnt && i < ng; i++) acc = xyzz_madd(acc, ld_affine(gs + (size_t)i * 64));
  st_affine(out + (size_t)k * 64, xyzz_to_affine(acc));
}

// start[b][w] = 2^(8w) * bases[b] in XYZZ; one thread per base
__global__ void __launch_bounds__(256) k_fb_starts(const uint8_t* __restrict__ bases, uint32_t nbases,
                                                  xyzz* __restrict__ starts) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbases) return;
  xyzz cur = xyzz_from_affine(ld_affine(bases + (size_t)b * 64));
  for (int w = 0; w < FB_WINDOWS; w++) {
    st_xyzz(&starts[(size_t)b * FB_WINDOWS + w], cur);
    if (w + 1 < FB_WINDOWS)
      for (int i = 0; i < FB_WBITS; i++) cur = xyzz_dbl(cur);
  }
}

// table[(b*33 + w)*128 + d-1] = d * start[b][w], affine.  One thread per (b, w):
// 127 XYZZ additions, then one shared inversion for the 128 entries
// (Montgomery's trick, with the prefix products parked in the output rows).
__global__ void __launch_bounds__(256) k_fb_fill(const xyzz* __restrict__ starts, uint32_t nrows,
                                                xyzz* __restrict__ scratch, uint8_t* __restrict__ table) {
  uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  xyzz s = ld_xyzz(&starts[row]);
  xyzz* tmp = scratch + (size_t)row * FB_ENTRIES;
  uint8_t* out = table + (size_t)row * FB_ENTRIES * 64;
  bool inf = xyzz_is_inf(s);
  // entries in XYZZ
  xyzz cur = s;
  for (int d = 0; d < FB_ENTRIES; d++) {
    st_xyzz(&tmp[d], cur);
    cur = xyzz_add(cur, s);
  }
  if (inf) {
    for (int d = 0; d < FB_ENTRIES; d++) st_affine(out + (size_t)d * 64, affine_inf());
    return;
  }
  // prefix products of ZZZ (d*S is never the identity for d <= 128 < r)
  u256 run = u256_one();
  for (int d = 0; d < FB_ENTRIES; d++) {
    st_u256(out + (size_t)d * 64, run);  // product of ZZZ_0..ZZZ_{d-1}
    run = fq_mul(run, ld_u256(&tmp[d].ZZZ));
  }
  u256 inv = fq_inv(run);
  for (int d = FB_ENTRIES - 1; d >= 0; d--) {
    xyzz e = ld_xyzz(&tmp[d]);
    u256 pre = ld_u256(out + (size_t)d * 64);
    u256 zi = fq_mul(inv, pre);  // 1 / ZZZ_d
    inv = fq_mul(inv, e.ZZZ);
    st_affine(out + (size_t)d * 64, xyzz_to_affine_with_inv(e, zi));
  }
}

}  // namespace bp

using namespace bp;

extern "C" {

int bp_ctx_set_generators(bp_ctx* ctx, const uint8_t* gs, const uint8_t* hs, const uint8_t* h, uint32_t n_gens) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(gs && hs && h, "null pointer");
  BP_REQUIRE(n_gens >= 1 && n_gens <= 8192 && (n_gens & (n_gens - 1)) == 0,
             "n_gens must be a power of two in 1..8192");
  BP_TRY(hipSetDevice(ctx->device));
  uint32_t nk = 0;
  while ((1u << nk) < n_gens) nk++;
  nk += 1;  // GS_k for k = 0..log2(n_gens)
  uint32_t nbases = 2 * n_gens + 2 + nk;
  size_t nrows = (size_t)nbases * FB_WINDOWS;
  size_t table_bytes = nrows * FB_ENTRIES * 64;
  Carver cv(nullptr);
  auto carve = [&](void* base) {
    Carver c(base);
    ctx->d_bases = c.take<uint8_t>((size_t)nbases * 64);
    ctx->d_table = c.take<uint8_t>(table_bytes);
    return c.off;
  };
  size_t need = carve(nullptr);
  if (ctx->gen.reserve(need) != 0) {
    ctx->err = "generator table allocation failed";
    return BP_ERR_NOMEM;
  }
  carve(ctx->gen.p);
  hipStream_t st = ctx->stream;
  affine G = affine_gen();
  BP_TRY(hipMemcpyAsync(ctx->d_bases, gs, (size_t)n_gens * 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(ctx->d_bases + (size_t)n_gens * 64, hs, (size_t)n_gens * 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(ctx->d_bases + (size_t)2 * n_gens * 64, &G, 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(ctx->d_bases + (size_t)(2 * n_gens + 1) * 64, h, 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipStreamSynchronize(st));  // G lives on the host stack
  hipLaunchKernelGGL(k_gsum, dim3(1), dim3(256), 0, st, ctx->d_bases, n_gens, nk,
                     ctx->d_bases + (size_t)(2 * n_gens + 2) * 64);
  // temporary buffers for the build
  DevBuf tmp;
  size_t tmp_bytes = nrows * sizeof(xyzz) + 256 + nrows * FB_ENTRIES * sizeof(xyzz);
  if (tmp.reserve(tmp_bytes) != 0) {
    ctx->err = "generator table scratch allocation failed";
    return BP_ERR_NOMEM;
  }
  xyzz* starts = (xyzz*)tmp.p;
  xyzz* scratch = (xyzz*)((char*)tmp.p + ((nrows * sizeof(xyzz) + 255) & ~(size_t)255));
  hipLaunchKernelGGL(k_fb_starts, dim3((nbases + 255) / 256), dim3(256), 0, st, ctx->d_bases, nbases, starts);
  hipLaunchKernelGGL(k_fb_fill, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, starts, (uint32_t)nrows,
                     scratch, ctx->d_table);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  tmp.release();
  if (e != hipSuccess) {
    ctx->err = std::string("generator table build: ") + hipGetErrorString(e);
    return BP_ERR_DEVICE;
  }
  ctx->n_gens = n_gens;
  ctx->n_bases = nbases;
  return BP_OK;
}

}  // extern "C"
