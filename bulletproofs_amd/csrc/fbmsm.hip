// A few large multi-exponentiations over the context's OWN generators (gs_i, hs_i, gen) through the
// fixed-base window tables (fixedbase.h): `sumExps` (/root/reference/Bulletproofs/Utils.hs:76-80) when every
// point is one of the reference's generator CAFs.  The inner-product prover uses it to evaluate
// lCommit, rCommit and commitmentLR' of a round (InnerProductProof/Prover.hs:63-92) over the ORIGINAL
// generators with expanded scalars instead of folding the generator vectors (prover_host.hip).
//
// One lane per term walks all windows of its scalar (one gathered table entry and one mixed addition per
// non-zero digit, no doublings), workgroups sum their lanes, a second launch sums the workgroups and adds an
// optional offset point to the first result.  The sums leave as XYZZ: the host, which waits for them anyway,
// normalises them with one shared inversion in microseconds (host_inv.h), against ~0.14 ms for a GPU lane.
#include <stdlib.h>
#include "engine.h"
#include "devio.h"
#include "fixedbase.h"
#include "fq29.h"
#include "lanes29.h"
#include "msm_core.h"
#include "prover.h"

namespace bp {

// lpt = 2^lpt_log lanes share a term: lane `sub` of a term walks windows sub, sub + lpt, ...  A lane's chain is the
// critical path of every prover round (the host waits for the three sums), so short chains beat few lanes here.
template <int WB>
__global__ void __launch_bounds__(256) k_fb_msm_terms(uint32_t nterms, uint32_t lpt_log, const uint8_t* __restrict__ scalars /*[nmsm][nterms][32]*/,
                                                     const uint32_t* __restrict__ term_base, const uint8_t* __restrict__ table,
                                                     xyzz* __restrict__ partial /*[nmsm][gridDim.x]*/) {
  constexpr int W = fb_windows(WB), E = fb_entries(WB), DW = fb_dwords(WB), SLOT = fb_slot_bits(WB), PER = 32 / SLOT;
  __shared__ uint32_t red[256 * 36];
  __shared__ uint32_t sdig[DW * 256];  // the lane's digit words, read back by window index
  const uint32_t msm = blockIdx.y, gt = blockIdx.x * 256u + threadIdx.x;
  const uint32_t t = gt >> lpt_log, sub = gt & ((1u << lpt_log) - 1u), lpt = 1u << lpt_log;
  xyzz29 acc = xyzz29_inf();
  if (t < nterms) {
    u256 k = ld_u256(scalars + ((size_t)msm * nterms + t) * 32);
    if (!u256_is_zero(k)) {
      uint32_t d[DW];
      fb_recode<WB>(k, d);
#pragma unroll
      for (int g = 0; g < DW; g++) sdig[g * 256 + threadIdx.x] = d[g];
      const size_t row0 = (size_t)term_base[t] * W * E;
#pragma unroll 1
      for (int w = (int)sub; w < W; w += (int)lpt) {
        uint32_t word = sdig[(w / PER) * 256 + threadIdx.x];
        uint32_t code = SLOT == 32 ? word : ((word >> ((SLOT * (w % PER)) & 31)) & (uint32_t)((1ull << SLOT) - 1ull));
        if (code == 0) continue;
        bool neg;
        uint32_t mag = fb_decode<WB>(code, neg);
        affine q = ld_affine(table + (row0 + (size_t)w * E + (mag - 1)) * 64);
        acc = xyzz29_madd(acc, q, neg);
      }
    }
  }
  xyzz s = block_sum_quad(xyzz29_to(acc), red);  // quads of lanes (lanes29.h): the sums are what a prover's round waits for
  if (threadIdx.x == 0) st_xyzz(&partial[(size_t)msm * gridDim.x + blockIdx.x], s);
}

// out[msm] = sum of the msm's partials (+ offset for msm 0), XYZZ; one wave when there are at most 64 partials
__global__ void __launch_bounds__(256) k_fb_msm_finish(uint32_t nblk, const xyzz* __restrict__ partial,
                                                      const uint8_t* __restrict__ offset /*64 B affine, or null*/,
                                                      xyzz* __restrict__ out /*[nmsm]*/) {
  __shared__ uint32_t red[256 * 36];
  const uint32_t msm = blockIdx.x;
  xyzz acc = xyzz_inf(), s;
  for (uint32_t i = threadIdx.x; i < nblk; i += blockDim.x) acc = xyzz_add(acc, ld_xyzz(&partial[(size_t)msm * nblk + i]));
  s = block_sum_quad(acc, red);  // 64 or 256 threads
  if (threadIdx.x == 0) {
    if (msm == 0 && offset) s = xyzz_madd(s, ld_affine(offset));
    st_xyzz(&out[msm], s);
  }
}

// lanes per term: as many as keep the launch within about one wave per SIMD (65,536 lanes) and leave every lane at
// least two windows
static uint32_t fb_lpt_log(uint32_t nmsm, uint32_t nterms, int wbits) {
  static const int forced = [] {
    const char* e = getenv("BP_FB_LPT_LOG");
    return e ? atoi(e) : -1;
  }();
  if (forced >= 0 && forced <= 3) return (uint32_t)forced;
  const uint32_t W = (uint32_t)((256 + wbits) / wbits);
  uint32_t l = 0;
  while (l < 3 && ((uint64_t)nmsm * nterms << (l + 1)) <= 65536 && (2u << (l + 1)) <= W) l++;
  return l;
}

// partial sums fb_msm_launch needs at d_partial for these dimensions (XYZZ entries)
size_t fb_msm_partials(bp_ctx* ctx, uint32_t nmsm, uint32_t nterms) {
  const uint32_t l = fb_lpt_log(nmsm, nterms, ctx->fb_wbits);
  return (size_t)nmsm * ((((size_t)nterms << l) + 255) / 256);
}

// nmsm sums over the same nterms bases; d_scalars [nmsm][nterms][32] canonical, d_term_base [nterms] table base
// indices (fixedbase.h), d_partial: fb_msm_partials() XYZZ of scratch, d_out: nmsm sums, XYZZ (not normalised).
int fb_msm_launch(bp_ctx* ctx, uint32_t nmsm, uint32_t nterms, const uint8_t* d_scalars, const uint32_t* d_term_base,
                  xyzz* d_partial, const uint8_t* d_offset, xyzz* d_out) {
  if (!ctx->d_table || ctx->n_gens == 0) {
    ctx->err = "generators not set (bp_ctx_set_generators)";
    return BP_ERR_STATE;
  }
  const uint32_t l = fb_lpt_log(nmsm, nterms, ctx->fb_wbits);
  const uint32_t nblk = (uint32_t)((((size_t)nterms << l) + 255u) / 256u);
  dim3 grid(nblk, nmsm), block(256);
  hipStream_t st = ctx->stream;
  switch (ctx->fb_wbits) {
    case 20: hipLaunchKernelGGL(k_fb_msm_terms<20>, grid, block, 0, st, nterms, l, d_scalars, d_term_base, ctx->d_table, d_partial); break;
    case 16: hipLaunchKernelGGL(k_fb_msm_terms<16>, grid, block, 0, st, nterms, l, d_scalars, d_term_base, ctx->d_table, d_partial); break;
    case 12: hipLaunchKernelGGL(k_fb_msm_terms<12>, grid, block, 0, st, nterms, l, d_scalars, d_term_base, ctx->d_table, d_partial); break;
    default: hipLaunchKernelGGL(k_fb_msm_terms<8>, grid, block, 0, st, nterms, l, d_scalars, d_term_base, ctx->d_table, d_partial); break;
  }
  hipLaunchKernelGGL(k_fb_msm_finish, dim3(nmsm), dim3(nblk <= 64 ? 64 : 256), 0, st, nblk, d_partial, d_offset, d_out);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

}  // namespace bp
