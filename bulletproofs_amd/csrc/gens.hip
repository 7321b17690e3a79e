// Generator state of a context: the device-resident mirror of the reference's
// lazy CAF lists `gs`, `hs` and the points `h`, `gen`
// (/root/reference/Bulletproofs/Utils.hs:14-25), plus the fixed-base window
// tables (fixedbase.h) used by the batch verifier and by bp_commit.
#include <stdlib.h>
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

// start[b][w] = 2^(wb*w) * bases[b] in XYZZ; one thread per base
__global__ void __launch_bounds__(256) k_fb_starts(const uint8_t* __restrict__ bases, uint32_t nbases, int wb,
                                                  xyzz* __restrict__ starts) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbases) return;
  const int W = fb_windows(wb);
  xyzz cur = xyzz_from_affine(ld_affine(bases + (size_t)b * 64));
  for (int w = 0; w < W; w++) {
    st_xyzz(&starts[(size_t)b * W + w], cur);
    if (w + 1 < W)
      for (int i = 0; i < wb; i++) cur = xyzz_dbl(cur);
  }
}

// table[(row * E + d - 1)] = d * start[row], affine, for rows [row0, row0 + nrows), E = 2^(wb-1) entries each.
// One thread per (row, block of `blk` consecutive d): a short double-and-add to the block's first
// multiple, blk - 1 XYZZ additions, then one shared inversion for the block (Montgomery's trick, with
// the prefix products parked in the output slots).  scratch[(j * nthreads + t)] keeps the block's
// XYZZ values, strided so that lanes coalesce.
__global__ void __launch_bounds__(256) k_fb_fill(const xyzz* __restrict__ starts, uint32_t row0, uint32_t nrows,
                                                uint32_t entries, uint32_t blk, xyzz* __restrict__ scratch,
                                                uint8_t* __restrict__ table) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t per_row = entries / blk;
  uint32_t nthreads = nrows * per_row;
  if (t >= nthreads) return;
  uint32_t row = row0 + t / per_row, part = t % per_row;
  xyzz s = ld_xyzz(&starts[row]);
  uint8_t* out = table + ((size_t)row * entries + (size_t)part * blk) * 64;
  if (xyzz_is_inf(s)) {
    for (uint32_t d = 0; d < blk; d++) st_affine(out + (size_t)d * 64, affine_inf());
    return;
  }
  // first = (part * blk + 1) * s
  xyzz cur = xyzz_inf();
  uint32_t first = part * blk + 1;
  for (int bit = 19; bit >= 0; bit--) {
    cur = xyzz_dbl(cur);
    if ((first >> bit) & 1u) cur = xyzz_add(cur, s);
  }
  for (uint32_t d = 0; d < blk; d++) {
    st_xyzz(&scratch[(size_t)d * nthreads + t], cur);
    cur = xyzz_add(cur, s);
  }
  // prefix products of ZZZ (d*S is never the identity for d <= 2^19 < r)
  u256 run = u256_one();
  for (uint32_t d = 0; d < blk; d++) {
    st_u256(out + (size_t)d * 64, run);  // product of ZZZ_0..ZZZ_{d-1}
    run = fq_mul(run, ld_u256(&scratch[(size_t)d * nthreads + t].ZZZ));
  }
  u256 inv = fq_inv_gcd(run);
  for (int d = (int)blk - 1; d >= 0; d--) {
    xyzz e = ld_xyzz(&scratch[(size_t)d * nthreads + t]);
    u256 pre = ld_u256(out + (size_t)d * 64);
    u256 zi = fq_mul(inv, pre);  // 1 / ZZZ_d
    inv = fq_mul(inv, e.ZZZ);
    st_affine(out + (size_t)d * 64, xyzz_to_affine_with_inv(e, zi));
  }
}

// bad[0] += 1 for every entry that is not a finite point of the curve with canonical coordinates (< p)
__global__ void __launch_bounds__(256) k_gen_check(const uint8_t* __restrict__ pts, uint32_t n, uint32_t* __restrict__ bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine a = ld_affine(pts + (size_t)i * 64);
  const u256 p = fq_p();
  bool ok = !affine_is_inf(a) && !u256_geq(a.x, p) && !u256_geq(a.y, p) && affine_on_curve(a);
  if (!ok) atomicAdd(bad, 1u);
}

// window width for a generator set: the widest whose table fits the budget (env BP_FB_WBITS forces one)
static int choose_fb_wbits(uint32_t nbases) {
  if (const char* e = getenv("BP_FB_WBITS")) {
    int v = atoi(e);
    if (v == 8 || v == 12 || v == 16 || v == 20) return v;
  }
  size_t budget = (size_t)64 << 30;  // of 288 GB HBM3E
  if (const char* e = getenv("BP_FB_TABLE_GIB")) {
    long v = atol(e);
    if (v >= 1 && v <= 200) budget = (size_t)v << 30;
  }
  const int cand[4] = {20, 16, 12, 8};
  for (int wb : cand) {
    size_t bytes = (size_t)nbases * fb_windows(wb) * fb_entries(wb) * 64;
    if (bytes <= budget) return wb;
  }
  return 8;
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
  // Whatever happens below, the previous generator set is gone (the table buffer is re-used or freed): the context
  // holds NO generators until the new tables are complete, and batch objects created against the old set are
  // refused from now on (they compare this epoch).
  ctx->n_gens = 0;
  ctx->n_bases = 0;
  ctx->d_bases = nullptr;
  ctx->d_table = nullptr;
  ctx->gen_epoch++;
  int wb = choose_fb_wbits(nbases);
  const bool forced = getenv("BP_FB_WBITS") != nullptr;
  uint32_t W = 0, E = 0, blk = 0;
  size_t nrows = 0, table_bytes = 0, row_scratch = 0, slab_rows = 0, starts_bytes = 0;
  uint8_t *d_bases = nullptr, *d_table = nullptr;
  DevBuf tmp;
  struct TmpGuard {
    DevBuf& b;
    ~TmpGuard() { b.release(); }
  } tmp_guard{tmp};
  for (;;) {  // on an allocation failure fall back to the next narrower window
    W = fb_windows(wb);
    E = fb_entries(wb);
    blk = E < (uint32_t)FB_BUILD_BLOCK ? E : (uint32_t)FB_BUILD_BLOCK;
    nrows = (size_t)nbases * W;
    table_bytes = nrows * E * 64;
    auto carve = [&](void* base) {
      Carver c(base);
      d_bases = c.take<uint8_t>((size_t)nbases * 64);
      d_table = c.take<uint8_t>(table_bytes);
      return c.off;
    };
    size_t need = carve(nullptr);
    // temporary buffers for the build: window starts + XYZZ scratch for one slab of rows (<= 2 GiB)
    row_scratch = (size_t)E * sizeof(xyzz);
    slab_rows = ((size_t)2 << 30) / row_scratch;
    if (slab_rows < 1) slab_rows = 1;
    if (slab_rows > nrows) slab_rows = nrows;
    starts_bytes = (nrows * sizeof(xyzz) + 255) & ~(size_t)255;
    if (ctx->gen.reserve(need) == 0 && tmp.reserve(starts_bytes + slab_rows * row_scratch) == 0) {
      carve(ctx->gen.p);
      break;
    }
    tmp.release();
    (void)hipGetLastError();  // clear the allocation error
    if (forced || wb == 8) {
      ctx->gen.release();
      ctx->err = "generator table allocation failed";
      return BP_ERR_NOMEM;
    }
    wb = wb == 20 ? 16 : (wb == 16 ? 12 : 8);
  }
  hipStream_t st = ctx->stream;
  affine G = affine_gen();
  BP_TRY(hipMemcpyAsync(d_bases, gs, (size_t)n_gens * 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(d_bases + (size_t)n_gens * 64, hs, (size_t)n_gens * 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(d_bases + (size_t)2 * n_gens * 64, &G, 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(d_bases + (size_t)(2 * n_gens + 1) * 64, h, 64, hipMemcpyHostToDevice, st));
  {
    hipError_t e0 = hipStreamSynchronize(st);  // G lives on the host stack
    if (e0 != hipSuccess) {
      ctx->err = std::string("generator upload: ") + hipGetErrorString(e0);
      return BP_ERR_DEVICE;
    }
  }
  {
    // gs, hs, gen, h replace group elements the reference computes itself: every entry must be a finite curve point
    // in canonical encoding (the kernels' "d * B is never the identity" and the transcript text rely on it)
    uint32_t* d_bad = (uint32_t*)tmp.p;
    uint32_t bad = 0;
    hipError_t e1 = hipMemsetAsync(d_bad, 0, 4, st);
    hipLaunchKernelGGL(k_gen_check, dim3((2 * n_gens + 2 + 255) / 256), dim3(256), 0, st, d_bases, 2 * n_gens + 2, d_bad);
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);
    if (e1 != hipSuccess) {
      ctx->err = std::string("generator check: ") + hipGetErrorString(e1);
      return BP_ERR_DEVICE;
    }
    if (bad) {
      ctx->err = "bp_ctx_set_generators: a generator is not a finite, canonically encoded point of the curve";
      return BP_ERR_ARG;
    }
  }
  hipLaunchKernelGGL(k_gsum, dim3(1), dim3(256), 0, st, d_bases, n_gens, nk,
                     d_bases + (size_t)(2 * n_gens + 2) * 64);
  xyzz* starts = (xyzz*)tmp.p;
  xyzz* scratch = (xyzz*)((char*)tmp.p + starts_bytes);
  hipLaunchKernelGGL(k_fb_starts, dim3((nbases + 255) / 256), dim3(256), 0, st, d_bases, nbases, wb, starts);
  for (size_t r0 = 0; r0 < nrows; r0 += slab_rows) {
    uint32_t nr = (uint32_t)(nrows - r0 < slab_rows ? nrows - r0 : slab_rows);
    size_t nthreads = (size_t)nr * (E / blk);
    hipLaunchKernelGGL(k_fb_fill, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, starts, (uint32_t)r0, nr,
                       E, blk, scratch, d_table);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    ctx->err = std::string("generator table build: ") + hipGetErrorString(e);
    return BP_ERR_DEVICE;
  }
  // commit: only a complete table becomes visible
  ctx->d_bases = d_bases;
  ctx->d_table = d_table;
  ctx->fb_wbits = wb;
  ctx->n_gens = n_gens;
  ctx->n_bases = nbases;
  return BP_OK;
}

}  // extern "C"
