// Host control of the prover path, in C++ above the engine (the reference is compiled Haskell):
//   InnerProductProof.generateProof            (/root/reference/Bulletproofs/InnerProductProof/Prover.hs:17-92)
//   MultiRangeProof.generateProof(Unsafe)      (MultiRangeProof/Prover.hs:19-123), computeLRPolys / computeTPoly (:133-169)
//   RangeProof.generateProof                   (RangeProof/Prover.hs:15-32)
//   commitBitVectors                           (RangeProof/Internal.hs:126-149)
// Every `sumExps` is one bucket MSM on the GPU over device-resident generator arrays, the
// generator folds and hs' rescale are GLV kernels (ipp.hip); the O(nm) scalar-field vector
// algebra, bit decomposition, validation and the Fiat-Shamir transcript stay on the host.
// Randomness: the reference draws sL, sR, aBlinding, sBlinding, t1Blinding,
// t2Blinding from MonadRandom in that order; here they are explicit arguments.
#include "prover.h"
#include "fixedbase.h"
#include "glv.h"
#include "host_inv.h"
#include "validate.h"

using namespace bp;
using namespace bp::host;

namespace bp {

int carve_ws(bp_ctx* ctx, uint32_t n, ProverWs& w, uint32_t extra) {
  auto carve = [&](void* base) {
    Carver c(base);
    w.G = c.take<uint8_t>((size_t)n * 64);
    w.H = c.take<uint8_t>((size_t)n * 64);
    w.G2 = c.take<uint8_t>((size_t)n * 64);
    w.H2 = c.take<uint8_t>((size_t)n * 64);
    w.seg_cap = (((size_t)2 * n + 16 + extra) + 255) / 256 * 256;
    w.pts = c.take<uint8_t>(4 * w.seg_cap * 64);
    w.sc = c.take<uint8_t>(4 * w.seg_cap * 32);
    w.small = c.take<uint8_t>(16 * 64);
    w.tab = c.take<xyzz>((size_t)15 * (n + 2));
    w.bit_cap = 64 + extra;
    w.bit_pts = c.take<uint8_t>((size_t)w.bit_cap * 64);
    w.bit_sh = c.take<xyzz>((size_t)w.bit_cap * GLV_LANE_BITS);
    w.bit_out = c.take<xyzz>(8);
    w.xout = c.take<xyzz>(16);
    w.fr.cap = n;
    w.fr.ls = c.take<u256>(n);
    w.fr.rs = c.take<u256>(n);
    w.fr.cg = c.take<u256>(n);
    w.fr.ch = c.take<u256>(n);
    for (int i = 0; i < 12; i++) w.fr.v[i] = c.take<u256>(n);
    w.fr.sm = c.take<u256>(64);
    w.fr.bits = c.take<uint8_t>(n);
    w.fr.raw = c.take<uint8_t>((size_t)6 * n * 32);
    return c.off;
  };
  size_t need = carve(nullptr);
  if (ctx->pws.reserve(need) != 0) {
    ctx->err = "prover workspace allocation failed";
    return BP_ERR_NOMEM;
  }
  carve(ctx->pws.p);
  return BP_OK;
}

int msm_gather(bp_ctx* ctx, ProverWs& w, const MsmSeg* segs, int n_segs, const std::vector<Fr4>& scalars,
               uint8_t out[64]) {
  hipStream_t st = ctx->stream;
  size_t total = 0;
  for (int i = 0; i < n_segs; i++) {
    if (segs[i].count)
      BP_TRY(hipMemcpyAsync(w.pts + total * 64, segs[i].pts, segs[i].count * 64,
                            segs[i].on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    total += segs[i].count;
  }
  BP_REQUIRE(total == scalars.size(), "msm_gather: scalar count does not match the segments");
  uint8_t* d_out = w.small + 15 * 64;
  if (total == 0) {
    memset(out, 0, 64);
    return BP_OK;
  }
  std::vector<uint8_t> sc(total * 32);
  for (size_t i = 0; i < total; i++) fr_to_bytes(scalars[i], &sc[i * 32]);
  BP_TRY(hipMemcpyAsync(w.sc, sc.data(), sc.size(), hipMemcpyHostToDevice, st));
  int rc = msm_run(ctx, w.sc, w.pts, total, 0, d_out, nullptr);
  if (rc != BP_OK) return rc;
  BP_TRY(hipMemcpyAsync(out, d_out, 64, hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));  // `sc` and host segments are temporaries
  return BP_OK;
}

int msm_multi(bp_ctx* ctx, ProverWs& w, const MsmJob* jobs, int njobs) {
  BP_REQUIRE(njobs >= 1 && njobs <= 4, "msm_multi: 1..4 jobs");
  size_t longest = 0;
  for (int j = 0; j < njobs; j++) {
    size_t total = 0;
    for (int s2 = 0; s2 < jobs[j].nsegs; s2++) total += jobs[j].segs[s2].count;
    BP_REQUIRE(total == jobs[j].scalars->size(), "msm_multi: scalar count does not match the segments");
    BP_REQUIRE(total <= w.seg_cap, "msm_multi: job exceeds the workspace");
    if (total > longest) longest = total;
  }
  if (longest > msm_direct_limit() || longest == 0) {
    for (int j = 0; j < njobs; j++) {
      int rc = msm_gather(ctx, w, jobs[j].segs, jobs[j].nsegs, *jobs[j].scalars, jobs[j].out);
      if (rc != BP_OK) return rc;
    }
    return BP_OK;
  }
  hipStream_t st = ctx->stream;
  const uint32_t stride = (uint32_t)((longest + 255) / 256 * 256);
  uint32_t n[4] = {0, 0, 0, 0};
  std::vector<uint8_t> sc((size_t)njobs * stride * 32);
  for (int j = 0; j < njobs; j++) {
    size_t off = (size_t)j * stride, total = 0;
    for (int s2 = 0; s2 < jobs[j].nsegs; s2++) {
      const MsmSeg& sg = jobs[j].segs[s2];
      if (sg.count)
        BP_TRY(hipMemcpyAsync(w.pts + (off + total) * 64, sg.pts, sg.count * 64,
                              sg.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
      total += sg.count;
    }
    for (size_t i = 0; i < total; i++) fr_to_bytes((*jobs[j].scalars)[i], &sc[(off + i) * 32]);
    n[j] = (uint32_t)total;
  }
  BP_TRY(hipMemcpyAsync(w.sc, sc.data(), sc.size(), hipMemcpyHostToDevice, st));
  uint8_t* d_out = w.small + 12 * 64;
  int rc = msm_direct_multi(ctx, w.sc, w.pts, stride, n, njobs, d_out);
  if (rc != BP_OK) return rc;
  uint8_t res[4 * 64];
  BP_TRY(hipMemcpyAsync(res, d_out, (size_t)njobs * 64, hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));
  for (int j = 0; j < njobs; j++) memcpy(jobs[j].out, res + 64 * j, 64);
  return BP_OK;
}

// MSM over [ptsA (na), ptsB (nb), single] with scalars [sa, sb, s1] -> affine result to host
int msm3(bp_ctx* ctx, ProverWs& w, const uint8_t* dA, const std::vector<Fr4>& sa, size_t a0, const uint8_t* dB,
         const std::vector<Fr4>& sb, size_t b0, size_t cnt, const uint8_t* d_single, const Fr4& s1, uint8_t out[64]) {
  hipStream_t st = ctx->stream;
  std::vector<uint8_t> sc((2 * cnt + 1) * 32);
  for (size_t i = 0; i < cnt; i++) {
    fr_to_bytes(sa[a0 + i], &sc[i * 32]);
    fr_to_bytes(sb[b0 + i], &sc[(cnt + i) * 32]);
  }
  fr_to_bytes(s1, &sc[2 * cnt * 32]);
  if (cnt) {
    BP_TRY(hipMemcpyAsync(w.pts, dA, cnt * 64, hipMemcpyDeviceToDevice, st));
    BP_TRY(hipMemcpyAsync(w.pts + cnt * 64, dB, cnt * 64, hipMemcpyDeviceToDevice, st));
  }
  BP_TRY(hipMemcpyAsync(w.pts + 2 * cnt * 64, d_single, 64, hipMemcpyDeviceToDevice, st));
  BP_TRY(hipMemcpyAsync(w.sc, sc.data(), sc.size(), hipMemcpyHostToDevice, st));
  uint8_t* d_out = w.small + 15 * 64;
  int rc = msm_run(ctx, w.sc, w.pts, 2 * cnt + 1, 0, d_out, nullptr);
  if (rc != BP_OK) return rc;
  BP_TRY(hipMemcpyAsync(out, d_out, 64, hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));  // `sc` is a host temporary
  return BP_OK;
}

// The recursion of generateProof' on device-resident G, H (w.G, w.H hold n points each).
// ls, rs in Montgomery form; u, P affine (host bytes).  Appends L_k, R_k (64 B each) and writes l, r.
int ipp_prove_core(bp_ctx* ctx, ProverWs& w, uint32_t n, const uint8_t u[64], const uint8_t P0[64],
                   std::vector<Fr4>& ls, std::vector<Fr4>& rs, ShowStyle style, std::vector<uint8_t>& Ls,
                   std::vector<uint8_t>& Rs, uint8_t l_out[32], uint8_t r_out[32]) {
  hipStream_t st = ctx->stream;
  uint8_t* d_u = w.small;
  uint8_t* d_P = w.small + 64;
  uint8_t* d_L = w.small + 128;
  uint8_t* d_R = w.small + 192;
  uint8_t* d_Pn = w.small + 256;
  uint8_t P[64];
  memcpy(P, P0, 64);
  BP_TRY(hipMemcpyAsync(d_u, u, 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(d_P, P, 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipStreamSynchronize(st));
  if (n == 0) {  // ([], []) -> InnerProductProof [] [] 0 0   (Prover.hs:42)
    memset(l_out, 0, 32);
    memset(r_out, 0, 32);
    return BP_OK;
  }
  int rc;
  while (n > 1) {
    uint32_t h = n / 2;
    Fr4 cL = fr_zero(), cR = fr_zero();
    for (uint32_t i = 0; i < h; i++) {
      cL = fr_add(cL, fr_mul(ls[i], rs[h + i]));  // dot lsLeft rsRight  (:60)
      cR = fr_add(cR, fr_mul(ls[h + i], rs[i]));  // dot lsRight rsLeft  (:61)
    }
    uint8_t L[64], R[64];
    {
      // lCommit = sumExps lsLeft gsRight <> sumExps rsRight hsLeft <> (bH `mul` cL)   (:63-67)
      // rCommit = sumExps lsRight gsLeft <> sumExps rsLeft hsRight <> (bH `mul` cR)   (:69-73)
      // both in one launch: two segments of 2h + 1 terms
      MsmSeg segL[3] = {{w.G + (size_t)h * 64, h, true}, {w.H, h, true}, {d_u, 1, true}};
      MsmSeg segR[3] = {{w.G, h, true}, {w.H + (size_t)h * 64, h, true}, {d_u, 1, true}};
      std::vector<Fr4> scL(2 * (size_t)h + 1), scR(2 * (size_t)h + 1);
      for (uint32_t i = 0; i < h; i++) {
        scL[i] = ls[i];
        scL[h + i] = rs[h + i];
        scR[i] = ls[h + i];
        scR[h + i] = rs[i];
      }
      scL[2 * (size_t)h] = cL;
      scR[2 * (size_t)h] = cR;
      MsmJob jobs[2] = {{segL, 3, &scL, L}, {segR, 3, &scR, R}};
      if ((rc = msm_multi(ctx, w, jobs, 2)) != BP_OK) return rc;
    }
    Ls.insert(Ls.end(), L, L + 64);
    Rs.insert(Rs.end(), R, R + 64);
    // x = shamirX' commitmentLR lCommit rCommit   (:75)
    uint8_t buf[MSG_SCRATCH], xb[32], xib[32], x2b[32], xi2b[32];
    Msg mm(buf, style);
    mm.r();
    mm.point(L);
    mm.point(R);
    mm.point(P);
    mm.finish_fr(xb);
    Fr4 x = fr_from_bytes(xb), xi = fr_inv(x);
    fr_to_bytes(xi, xib);
    fr_to_bytes(fr_sqr(x), x2b);
    fr_to_bytes(fr_sqr(xi), xi2b);
    // gs'' = xInv*gsLeft + x*gsRight ; hs'' = x*hsLeft + xInv*hsRight   (:81-82) and
    // commitmentLR' = x^2 L + xInv^2 R + commitmentLR   (:87-92): 2h + 1 lanes of one launch
    BP_TRY(hipMemcpyAsync(d_L, L, 64, hipMemcpyHostToDevice, st));
    BP_TRY(hipMemcpyAsync(d_R, R, 64, hipMemcpyHostToDevice, st));
    {
      LinJob jb[3];
      memset((void*)jb, 0, sizeof(jb));
      jb[0].a = w.G;
      jb[0].b = w.G + (size_t)h * 64;
      jb[0].out = w.G2;
      memcpy(jb[0].ka.v, xib, 32);
      memcpy(jb[0].kb.v, xb, 32);
      jb[0].n = h;
      jb[0].stride_a = jb[0].stride_b = 64;
      jb[1].a = w.H;
      jb[1].b = w.H + (size_t)h * 64;
      jb[1].out = w.H2;
      memcpy(jb[1].ka.v, xb, 32);
      memcpy(jb[1].kb.v, xib, 32);
      jb[1].n = h;
      jb[1].stride_a = jb[1].stride_b = 64;
      jb[2].a = d_L;
      jb[2].b = d_R;
      jb[2].c = d_P;
      jb[2].out = d_Pn;
      memcpy(jb[2].ka.v, x2b, 32);
      memcpy(jb[2].kb.v, xi2b, 32);
      jb[2].n = 1;
      if ((rc = ipp_lincomb_jobs(ctx, jb, 3, w.tab)) != BP_OK) return rc;
    }
    std::swap(w.G, w.G2);
    std::swap(w.H, w.H2);
    BP_TRY(hipMemcpyAsync(P, d_Pn, 64, hipMemcpyDeviceToHost, st));
    BP_TRY(hipMemcpyAsync(d_P, d_Pn, 64, hipMemcpyDeviceToDevice, st));
    // ls' = x*lsLeft + xInv*lsRight ; rs' = xInv*rsLeft + x*rsRight   (:84-85) while the GPU folds
    for (uint32_t i = 0; i < h; i++) {
      ls[i] = fr_add(fr_mul(x, ls[i]), fr_mul(xi, ls[h + i]));
      rs[i] = fr_add(fr_mul(xi, rs[i]), fr_mul(x, rs[h + i]));
    }
    ls.resize(h);
    rs.resize(h);
    BP_TRY(hipStreamSynchronize(st));
    n = h;
  }
  fr_to_bytes(ls[0], l_out);
  fr_to_bytes(rs[0], r_out);
  return BP_OK;
}

int fb_msm_host(bp_ctx* ctx, ProverWs& w, const std::vector<uint32_t>& term_base, uint32_t nmsm,
                const std::vector<Fr4>* const* scalars, uint8_t* const* outs) {
  hipStream_t st = ctx->stream;
  const uint32_t nterms = (uint32_t)term_base.size();
  BP_REQUIRE(nmsm >= 1 && nmsm <= 15 && nterms >= 1, "fb_msm_host: 1..15 sums over at least one term");
  const size_t tb_bytes = ((size_t)nterms * 4 + 255) & ~(size_t)255;
  if ((size_t)nmsm * nterms * 32 > 4 * w.seg_cap * 32 || tb_bytes + fb_msm_partials(ctx, nmsm, nterms) * sizeof(xyzz) > 4 * w.seg_cap * 64) {
    ctx->err = "fb_msm_host: prover workspace too small";
    return BP_ERR_STATE;
  }
  std::vector<uint8_t> hsc((size_t)nmsm * nterms * 32);
  for (uint32_t j = 0; j < nmsm; j++) {
    BP_REQUIRE(scalars[j]->size() == nterms, "fb_msm_host: scalar count does not match the terms");
    for (uint32_t t = 0; t < nterms; t++) fr_to_bytes((*scalars[j])[t], &hsc[((size_t)j * nterms + t) * 32]);
  }
  uint32_t* d_tb = (uint32_t*)w.pts;
  xyzz* d_part = (xyzz*)(w.pts + tb_bytes);
  BP_TRY(hipMemcpyAsync(d_tb, term_base.data(), (size_t)nterms * 4, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(w.sc, hsc.data(), hsc.size(), hipMemcpyHostToDevice, st));
  int rc = fb_msm_launch(ctx, nmsm, nterms, w.sc, d_tb, d_part, nullptr, w.xout);
  if (rc != BP_OK) return rc;
  uint8_t out[15 * 64];
  if ((rc = fb_msm_fetch(ctx, w, nmsm, out)) != BP_OK) return rc;
  for (uint32_t j = 0; j < nmsm; j++) memcpy(outs[j], out + (size_t)j * 64, 64);
  return BP_OK;
}

int fb_msm_fetch(bp_ctx* ctx, ProverWs& w, uint32_t nmsm, uint8_t* out, void* tail_host, size_t tail_bytes) {
  BP_REQUIRE(nmsm >= 1 && (size_t)nmsm * sizeof(xyzz) + tail_bytes <= 16 * sizeof(xyzz), "fb_msm_fetch: too many results");
  alignas(16) uint8_t stage[16 * sizeof(xyzz)];
  BP_TRY(hipMemcpyAsync(stage, w.xout, (size_t)nmsm * sizeof(xyzz) + tail_bytes, hipMemcpyDeviceToHost, ctx->stream));
  BP_TRY(hipStreamSynchronize(ctx->stream));
  affine a[16];
  xyzz px[16];
  memcpy((void*)px, stage, (size_t)nmsm * sizeof(xyzz));
  xyzz_batch_to_affine_host(px, (int)nmsm, a);
  memcpy(out, (const void*)a, (size_t)nmsm * 64);
  if (tail_bytes) memcpy(tail_host, stage + (size_t)nmsm * sizeof(xyzz), tail_bytes);
  return BP_OK;
}

bool prover_uses_fixed_base() {
  static const bool fold = getenv("BP_PROVER_FOLD") && atoi(getenv("BP_PROVER_FOLD")) != 0;
  return !fold;
}

int ipp_prove_fixed(bp_ctx* ctx, ProverWs& w, uint32_t n, const std::vector<Fr4>& hscale, const uint8_t ub[32],
                    const uint8_t D[64], std::vector<Fr4>& ls, std::vector<Fr4>& rs, ShowStyle style,
                    std::vector<uint8_t>& Ls, std::vector<uint8_t>& Rs, uint8_t l_out[32], uint8_t r_out[32]) {
  hipStream_t st = ctx->stream;
  if (n == 0) {  // ([], []) -> InnerProductProof [] [] 0 0   (Prover.hs:42)
    memset(l_out, 0, 32);
    memset(r_out, 0, 32);
    return BP_OK;
  }
  const uint32_t ng = ctx->n_gens;
  if (n > ng || (n & (n - 1)) != 0) {
    ctx->err = "ipp_prove_fixed: n must be a power of two within the context's generators";
    return BP_ERR_ARG;
  }
  if (n == 1) {  // ([l], [r]): no round, commitmentLR is never hashed (Prover.hs:43)
    fr_to_bytes(ls[0], l_out);
    fr_to_bytes(rs[0], r_out);
    return BP_OK;
  }
  const uint32_t nterms = 2 * n + 1;
  // device scratch out of the prover workspace: scalars in w.sc, base indices + partial sums in w.pts
  if ((size_t)3 * nterms * 32 > 4 * w.seg_cap * 32 ||
      (((size_t)nterms * 4 + 255) & ~(size_t)255) + fb_msm_partials(ctx, 3, nterms) * sizeof(xyzz) > 4 * w.seg_cap * 64) {
    ctx->err = "ipp_prove_fixed: prover workspace too small";
    return BP_ERR_STATE;
  }
  uint8_t* d_sc = w.sc;
  uint32_t* d_tb = (uint32_t*)w.pts;
  xyzz* d_part = (xyzz*)(w.pts + (((size_t)nterms * 4 + 255) & ~(size_t)255));
  uint8_t* d_off = w.small;
  {
    std::vector<uint32_t> tb(nterms);
    for (uint32_t i = 0; i < n; i++) {
      tb[i] = fb_base_g(i);
      tb[n + i] = fb_base_h(ng, i);
    }
    tb[2 * n] = fb_base_G(ng);
    BP_TRY(hipMemcpyAsync(d_tb, tb.data(), (size_t)nterms * 4, hipMemcpyHostToDevice, st));
    BP_TRY(hipMemcpyAsync(d_off, D, 64, hipMemcpyHostToDevice, st));
    BP_TRY(hipStreamSynchronize(st));  // tb lives on this frame
  }
  // fold coefficients of the original generators, in PLAIN form: a Montgomery product of a Montgomery-form
  // and a plain operand is the plain product, so every scalar below comes out canonical with one multiplication
  auto plain = [](const Fr4& a) {
    Fr4 p;
    fr_to_bytes(a, (uint8_t*)p.v);
    return p;
  };
  std::vector<Fr4> cG(n), cH(n);
  for (uint32_t i = 0; i < n; i++) {
    cG[i] = Fr4{{1, 0, 0, 0}};
    cH[i] = plain(hscale[i]);
  }
  const Fr4 ubm = fr_from_bytes(ub);
  std::vector<uint8_t> hsc((size_t)3 * nterms * 32);
  uint8_t out[3][64];
  // commitment of (ls, rs) under the folded base, and the next lCommit / rCommit, as three sums over 2n + 1 generators
  auto launch = [&](uint32_t nk) -> int {
    const uint32_t h = nk / 2;
    memset(hsc.data(), 0, hsc.size());
    Fr4 dot = fr_zero(), cL = fr_zero(), cR = fr_zero();
    for (uint32_t j = 0; j < nk; j++) dot = fr_add(dot, fr_mul(ls[j], rs[j]));
    for (uint32_t j = 0; j < h; j++) {
      cL = fr_add(cL, fr_mul(ls[j], rs[h + j]));  // dot lsLeft rsRight  (Prover.hs:60)
      cR = fr_add(cR, fr_mul(ls[h + j], rs[j]));  // dot lsRight rsLeft  (:61)
    }
    uint8_t* C = hsc.data();
    uint8_t* L = C + (size_t)nterms * 32;
    uint8_t* R = L + (size_t)nterms * 32;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t j = i & (nk - 1);
      Fr4 v = fr_mul(ls[j], cG[i]);
      memcpy(C + (size_t)i * 32, v.v, 32);
      v = fr_mul(rs[j], cH[i]);
      memcpy(C + (size_t)(n + i) * 32, v.v, 32);
      if (j >= h) {
        // lCommit = sumExps lsLeft gsRight <> sumExps rsRight hsLeft <> bH * cL   (:63-67)
        v = fr_mul(ls[j - h], cG[i]);
        memcpy(L + (size_t)i * 32, v.v, 32);
        // rCommit = sumExps lsRight gsLeft <> sumExps rsLeft hsRight <> bH * cR   (:69-73)
        v = fr_mul(rs[j - h], cH[i]);
        memcpy(R + (size_t)(n + i) * 32, v.v, 32);
      } else {
        v = fr_mul(rs[j + h], cH[i]);
        memcpy(L + (size_t)(n + i) * 32, v.v, 32);
        v = fr_mul(ls[j + h], cG[i]);
        memcpy(R + (size_t)i * 32, v.v, 32);
      }
    }
    fr_to_bytes(fr_mul(ubm, dot), C + (size_t)2 * n * 32);
    fr_to_bytes(fr_mul(ubm, cL), L + (size_t)2 * n * 32);
    fr_to_bytes(fr_mul(ubm, cR), R + (size_t)2 * n * 32);
    BP_TRY(hipMemcpyAsync(d_sc, hsc.data(), hsc.size(), hipMemcpyHostToDevice, st));
    int rc = fb_msm_launch(ctx, 3, nterms, d_sc, d_tb, d_part, d_off, w.xout);
    if (rc != BP_OK) return rc;
    return fb_msm_fetch(ctx, w, 3, (uint8_t*)out);
  };
  int rc;
  uint32_t nk = n;
  if ((rc = launch(nk)) != BP_OK) return rc;
  while (nk > 1) {
    const uint32_t h = nk / 2;
    const uint8_t *P = out[0], *L = out[1], *R = out[2];
    Ls.insert(Ls.end(), L, L + 64);
    Rs.insert(Rs.end(), R, R + 64);
    // x = shamirX' commitmentLR lCommit rCommit   (:75)
    uint8_t buf[MSG_SCRATCH], xb[32];
    Msg mm(buf, style);
    mm.r();
    mm.point(L);
    mm.point(R);
    mm.point(P);
    mm.finish_fr(xb);
    const Fr4 x = fr_from_bytes(xb), xi = fr_inv(x);
    // gs'' = xInv*gsLeft + x*gsRight ; hs'' = x*hsLeft + xInv*hsRight   (:81-82), as coefficients
    for (uint32_t i = 0; i < n; i++) {
      const bool lo = (i & (nk - 1)) < h;
      cG[i] = fr_mul(cG[i], lo ? xi : x);
      cH[i] = fr_mul(cH[i], lo ? x : xi);
    }
    // ls' = x*lsLeft + xInv*lsRight ; rs' = xInv*rsLeft + x*rsRight   (:84-85)
    for (uint32_t j = 0; j < h; j++) {
      ls[j] = fr_add(fr_mul(x, ls[j]), fr_mul(xi, ls[h + j]));
      rs[j] = fr_add(fr_mul(xi, rs[j]), fr_mul(x, rs[h + j]));
    }
    ls.resize(h);
    rs.resize(h);
    nk = h;
    if (nk > 1 && (rc = launch(nk)) != BP_OK) return rc;  // commitmentLR' (:87-92) and the next lCommit, rCommit
  }
  fr_to_bytes(ls[0], l_out);
  fr_to_bytes(rs[0], r_out);
  return BP_OK;
}

int mk_challenges_dev(bp_ctx* ctx, ProverWs& w, ShowStyle style, uint8_t acc[64], const uint8_t* Lp, const uint8_t* Rp,
                      uint32_t k, std::vector<Fr4>& chal, const xyzz* d_sh_lr) {
  hipStream_t st = ctx->stream;
  int rc;
  chal.assign(k, fr_zero());
  if (k == 0) return BP_OK;
  // The proof's L_j, R_j are known up front: with their bit tables a round is one short launch, the running
  // commitment stays XYZZ on the device, and the host normalises its copy between the hashes.
  const bool bits = d_sh_lr || (prover_uses_fixed_base() && 2 * k <= w.bit_cap);
  if (bits) {
    if (!d_sh_lr) {
      BP_TRY(hipMemcpyAsync(w.bit_pts, Lp, (size_t)k * 64, hipMemcpyHostToDevice, st));
      BP_TRY(hipMemcpyAsync(w.bit_pts + (size_t)k * 64, Rp, (size_t)k * 64, hipMemcpyHostToDevice, st));
      if ((rc = ipp_bit_tables(ctx, w.bit_pts, 2 * k, w.bit_sh, st)) != BP_OK) return rc;
      d_sh_lr = w.bit_sh;
    }
    affine a0;
    memcpy(&a0, acc, 64);
    xyzz ax = xyzz_from_affine(a0);
    BP_TRY(hipMemcpyAsync(w.bit_out, &ax, sizeof(xyzz), hipMemcpyHostToDevice, st));
    for (uint32_t j = 0; j < k; j++) {
      uint8_t buf[MSG_SCRATCH], xjb[32], x2b[32], xi2b[32];
      Msg mm(buf, style);
      mm.r();
      mm.point(Lp + (size_t)j * 64);
      mm.point(Rp + (size_t)j * 64);
      mm.point(acc);
      mm.finish_fr(xjb);
      chal[j] = fr_from_bytes(xjb);
      fr_to_bytes(fr_sqr(chal[j]), x2b);
      fr_to_bytes(fr_sqr(fr_inv(chal[j])), xi2b);
      if ((rc = ipp_bits_lincomb2(ctx, d_sh_lr + (size_t)j * GLV_LANE_BITS, d_sh_lr + (size_t)(k + j) * GLV_LANE_BITS, x2b,
                                  xi2b, w.bit_out)) != BP_OK)
        return rc;
      BP_TRY(hipMemcpyAsync(&ax, w.bit_out, sizeof(xyzz), hipMemcpyDeviceToHost, st));
      BP_TRY(hipStreamSynchronize(st));
      affine an = host::xyzz_to_affine_host(ax);
      memcpy(acc, &an, 64);
    }
    return BP_OK;
  }
  uint8_t* d_acc = w.small + 64;
  uint8_t* d_L = w.small + 128;
  uint8_t* d_R = w.small + 192;
  uint8_t* d_new = w.small + 256;
  BP_TRY(hipMemcpyAsync(d_acc, acc, 64, hipMemcpyHostToDevice, st));
  for (uint32_t j = 0; j < k; j++) {
    const uint8_t* Lj = Lp + (size_t)j * 64;
    const uint8_t* Rj = Rp + (size_t)j * 64;
    uint8_t buf[MSG_SCRATCH], xjb[32], x2b[32], xi2b[32];
    Msg mm(buf, style);
    mm.r();
    mm.point(Lj);
    mm.point(Rj);
    mm.point(acc);
    mm.finish_fr(xjb);
    chal[j] = fr_from_bytes(xjb);
    fr_to_bytes(fr_sqr(chal[j]), x2b);
    fr_to_bytes(fr_sqr(fr_inv(chal[j])), xi2b);
    BP_TRY(hipMemcpyAsync(d_L, Lj, 64, hipMemcpyHostToDevice, st));
    BP_TRY(hipMemcpyAsync(d_R, Rj, 64, hipMemcpyHostToDevice, st));
    if ((rc = ipp_lincomb2_add(ctx, d_L, d_R, d_acc, x2b, xi2b, w.tab, d_new)) != BP_OK) return rc;
    BP_TRY(hipMemcpyAsync(acc, d_new, 64, hipMemcpyDeviceToHost, st));
    BP_TRY(hipMemcpyAsync(d_acc, d_new, 64, hipMemcpyDeviceToDevice, st));
    BP_TRY(hipStreamSynchronize(st));
  }
  return BP_OK;
}

bool prover_vectors_on_device() {
  static const bool host = getenv("BP_PROVER_HOSTVEC") && atoi(getenv("BP_PROVER_HOSTVEC")) != 0;
  return !host;
}

int ipp_prove_fixed_dev(bp_ctx* ctx, ProverWs& w, uint32_t n, const u256* d_hscale, const uint8_t ub[32],
                        const uint8_t D[64], ShowStyle style, std::vector<uint8_t>& Ls, std::vector<uint8_t>& Rs,
                        uint8_t l_out[32], uint8_t r_out[32], uint8_t* dot0_out) {
  hipStream_t st = ctx->stream;
  if (n == 0) {  // ([], []) -> InnerProductProof [] [] 0 0   (Prover.hs:42)
    memset(l_out, 0, 32);
    memset(r_out, 0, 32);
    if (dot0_out) memset(dot0_out, 0, 32);
    return BP_OK;
  }
  const uint32_t ng = ctx->n_gens;
  if (n > ng || (n & (n - 1)) != 0 || n > w.fr.cap) {
    ctx->err = "ipp_prove_fixed_dev: n must be a power of two within the context's generators and the workspace";
    return BP_ERR_ARG;
  }
  const uint32_t nterms = 2 * n + 1;
  if ((size_t)3 * nterms * 32 > 4 * w.seg_cap * 32 ||
      (((size_t)nterms * 4 + 255) & ~(size_t)255) + fb_msm_partials(ctx, 3, nterms) * sizeof(xyzz) > 4 * w.seg_cap * 64) {
    ctx->err = "ipp_prove_fixed_dev: prover workspace too small";
    return BP_ERR_STATE;
  }
  FrDev& f = w.fr;
  uint8_t* d_sc = w.sc;
  uint32_t* d_tb = (uint32_t*)w.pts;
  xyzz* d_part = (xyzz*)(w.pts + (((size_t)nterms * 4 + 255) & ~(size_t)255));
  uint8_t* d_off = w.small;
  int rc;
  const u256 ubm = fr4_as_u256(fr_from_bytes(ub));
  const u256 zero = u256_zero();
  struct {
    uint8_t pts[3][64];
    uint8_t sm[3][32];
  } res;
  u256* d_sm = (u256*)(w.xout + 3);  // dot, l, r right behind the three sums: one copy per round
  auto fetch = [&](bool points) -> int {
    if (points) return fb_msm_fetch(ctx, w, 3, &res.pts[0][0], res.sm, sizeof(res.sm));
    BP_TRY(hipMemcpyAsync(res.sm, d_sm, sizeof(res.sm), hipMemcpyDeviceToHost, st));
    BP_TRY(hipStreamSynchronize(st));
    return BP_OK;
  };
  if (n == 1) {  // ([l], [r]): no round, commitmentLR is never hashed (Prover.hs:43)
    if ((rc = frv_ipp_round(ctx, n, 1, false, zero, zero, ubm, f.ls, f.rs, f.cg, f.ch, d_sc, d_sm, false)) != BP_OK) return rc;
    if ((rc = fetch(false)) != BP_OK) return rc;
    if (dot0_out) memcpy(dot0_out, res.sm[0], 32);
    memcpy(l_out, res.sm[1], 32);
    memcpy(r_out, res.sm[2], 32);
    return BP_OK;
  }
  {
    std::vector<uint32_t> tb(nterms);
    for (uint32_t i = 0; i < n; i++) {
      tb[i] = fb_base_g(i);
      tb[n + i] = fb_base_h(ng, i);
    }
    tb[2 * n] = fb_base_G(ng);
    BP_TRY(hipMemcpyAsync(d_tb, tb.data(), (size_t)nterms * 4, hipMemcpyHostToDevice, st));
    BP_TRY(hipMemcpyAsync(d_off, D, 64, hipMemcpyHostToDevice, st));
    if ((rc = frv_ipp_init(ctx, n, d_hscale, f.cg, f.ch)) != BP_OK) return rc;
    BP_TRY(hipStreamSynchronize(st));  // tb lives on this frame
  }
  // commitment of (ls, rs) under the folded base and the next lCommit / rCommit: three sums over the 2n + 1 generators
  auto launch = [&](uint32_t nk_prev, bool fold, const u256& x, const u256& xi) -> int {
    int r2 = frv_ipp_round(ctx, n, nk_prev, fold, x, xi, ubm, f.ls, f.rs, f.cg, f.ch, d_sc, d_sm, true);
    if (r2 != BP_OK) return r2;
    if ((r2 = fb_msm_launch(ctx, 3, nterms, d_sc, d_tb, d_part, d_off, w.xout)) != BP_OK) return r2;
    return fetch(true);
  };
  uint32_t nk = n;
  if ((rc = launch(nk, false, zero, zero)) != BP_OK) return rc;
  if (dot0_out) memcpy(dot0_out, res.sm[0], 32);
  while (nk > 1) {
    const uint8_t *P = res.pts[0], *L = res.pts[1], *R = res.pts[2];
    Ls.insert(Ls.end(), L, L + 64);
    Rs.insert(Rs.end(), R, R + 64);
    // x = shamirX' commitmentLR lCommit rCommit   (:75)
    uint8_t buf[MSG_SCRATCH], xb[32];
    Msg mm(buf, style);
    mm.r();
    mm.point(L);
    mm.point(R);
    mm.point(P);
    mm.finish_fr(xb);
    const Fr4 x = fr_from_bytes(xb), xi = fr_inv(x);
    // ls', rs' (:84-85), the fold coefficients (:81-82) and the next round's scalars: on the device
    if (nk > 2) {
      if ((rc = launch(nk, true, fr4_as_u256(x), fr4_as_u256(xi))) != BP_OK) return rc;
    } else {  // last fold: only l, r are left
      if ((rc = frv_ipp_round(ctx, n, nk, true, fr4_as_u256(x), fr4_as_u256(xi), ubm, f.ls, f.rs, f.cg, f.ch, d_sc, d_sm,
                              false)) != BP_OK)
        return rc;
      if ((rc = fetch(false)) != BP_OK) return rc;
    }
    nk /= 2;
  }
  memcpy(l_out, res.sm[1], 32);
  memcpy(r_out, res.sm[2], 32);
  return BP_OK;
}


// generateProofUnsafe (MultiRangeProof/Prover.hs:47-123) with every O(nm) scalar-field vector on the device:
// commitBitVectors' scalars, computeLRPolys, computeTPoly, ls / rs, powerVector (recip y) and the whole inner-product
// recursion (frvec.hip).  The host keeps the transcript, the scalars tBlinding / mu / t and the control flow.
// bits: aL, one byte per entry (nm of them: every value in range).
static int rangeproof_prove_dev(bp_ctx* ctx, ProverWs& w, uint32_t n, uint32_t m, uint32_t m2, uint32_t nm, uint32_t k,
                                const std::vector<uint8_t>& bits, const uint8_t* blindings, const uint8_t* sL,
                                const uint8_t* sR, const uint8_t* alpha, const uint8_t* rho, const uint8_t* tau1,
                                const uint8_t* tau2, uint8_t* proof_scalars, uint8_t* proof_points) {
  const ShowStyle style = (ShowStyle)ctx->show_style;
  hipStream_t st = ctx->stream;
  const uint32_t ng = ctx->n_gens;
  FrDev& f = w.fr;
  int rc;
  uint8_t* A = proof_points;
  uint8_t* S = proof_points + 64;
  uint8_t* T1 = proof_points + 128;
  uint8_t* T2 = proof_points + 192;
  u256 *l0 = f.v[0], *r0 = f.v[1], *r1 = f.v[2], *l1 = f.v[3], *sRm = f.v[4], *hscale = f.v[5];
  uint8_t* raw_sL = f.raw;
  uint8_t* raw_sR = f.raw + (size_t)nm * 32;
  const uint32_t nterms = 2 * nm + 1;
  const size_t tb_bytes = ((size_t)nterms * 4 + 255) & ~(size_t)255;
  if ((size_t)3 * nterms * 32 > 4 * w.seg_cap * 32 || tb_bytes + fb_msm_partials(ctx, 3, nterms) * sizeof(xyzz) > 4 * w.seg_cap * 64) {
    ctx->err = "rangeproof_prove_dev: prover workspace too small";
    return BP_ERR_STATE;
  }
  uint32_t* d_tb = (uint32_t*)w.pts;
  xyzz* d_part = (xyzz*)(w.pts + tb_bytes);
  // commitBitVectors: A = aBlinding*h + <aL, gs> + <aR, hs>, S likewise   (RangeProof/Internal.hs:136-147)
  std::vector<uint32_t> tb(nterms);
  for (uint32_t i = 0; i < nm; i++) {
    tb[i] = fb_base_g(i);
    tb[nm + i] = fb_base_h(ng, i);
  }
  tb[2 * (size_t)nm] = fb_base_H(ng);
  BP_TRY(hipMemcpyAsync(d_tb, tb.data(), (size_t)nterms * 4, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(f.bits, bits.data(), nm, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(raw_sL, sL, (size_t)nm * 32, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(raw_sR, sR, (size_t)nm * 32, hipMemcpyHostToDevice, st));
  if ((rc = frv_rp_commit_scalars(ctx, nm, f.bits, raw_sL, raw_sR, bytes_as_u256(alpha), bytes_as_u256(rho), w.sc)) != BP_OK)
    return rc;
  if ((rc = fb_msm_launch(ctx, 2, nterms, w.sc, d_tb, d_part, nullptr, w.xout)) != BP_OK) return rc;
  if ((rc = fb_msm_fetch(ctx, w, 2, A)) != BP_OK) return rc;  // A, S are in; tb, bits live on this frame
  // queued now, they run while the host hashes y and z
  if ((rc = frv_from_bytes(ctx, raw_sL, nm, l1)) != BP_OK) return rc;  // l1 = sL (Prover.hs:143)
  if ((rc = frv_from_bytes(ctx, raw_sR, nm, sRm)) != BP_OK) return rc;
  // y, z   (Prover.hs:75-76)
  uint8_t buf[MSG_SCRATCH], keep[MSG_SCRATCH], yb[32], zb[32], xb[32];
  Msg my(buf, style);
  my.r();
  my.point(A);
  my.point(S);
  size_t pre = my.len;
  memcpy(keep, buf, pre);
  my.finish_fr(yb);
  Msg mz(buf, style);
  mz.raw(keep, pre);
  mz.fr(yb);
  mz.finish_fr(zb);
  const Fr4 y = fr_from_bytes(yb), z = fr_from_bytes(zb);
  // computeLRPolys / computeTPoly   (Prover.hs:133-169)
  if ((rc = frv_rp_polys(ctx, n, nm, f.bits, sRm, fr4_as_u256(y), fr4_as_u256(z), l0, r0, r1)) != BP_OK) return rc;
  if ((rc = frv_rp_tpoly(ctx, nm, l0, l1, r0, r1, f.sm + 8)) != BP_OK) return rc;
  uint8_t tpoly[3][32];
  BP_TRY(hipMemcpyAsync(tpoly, f.sm + 8, sizeof(tpoly), hipMemcpyDeviceToHost, st));
  const Fr4 yinv = fr_inv(y);  // while the kernels run
  BP_TRY(hipStreamSynchronize(st));
  // hs' = hs_i * recip(y^i) (Prover.hs:107) enters the inner-product argument as scalar factors; runs behind the
  // host's work on T1, T2
  if ((rc = frv_power_vector(ctx, fr4_as_u256(yinv), nm, hscale)) != BP_OK) return rc;
  const Fr4 t0 = fr_from_bytes(tpoly[0]), t1 = fr_from_bytes(tpoly[1]), t2 = fr_from_bytes(tpoly[2]);
  // t1Commit = commit t1 t1Blinding, t2Commit likewise   (Prover.hs:84-85; Utils.hs:84-85)
  {
    std::vector<uint32_t> tbc{fb_base_G(ng), fb_base_H(ng)};
    std::vector<Fr4> s1{t1, fr_from_bytes(tau1)}, s2{t2, fr_from_bytes(tau2)};
    const std::vector<Fr4>* scs[2] = {&s1, &s2};
    uint8_t* outs[2] = {T1, T2};
    if ((rc = fb_msm_host(ctx, w, tbc, 2, scs, outs)) != BP_OK) return rc;
  }
  // x   (Prover.hs:88)
  Msg mx(buf, style);
  mx.raw(keep, pre);
  mx.point(T1);
  mx.point(T2);
  mx.fr(yb);
  mx.fr(zb);
  mx.finish_fr(xb);
  const Fr4 x = fr_from_bytes(xb), x2 = fr_sqr(x);
  // ls, rs on the device; t, tBlinding, mu   (Prover.hs:90-103)
  if ((rc = frv_rp_eval(ctx, nm, l0, l1, r0, r1, fr4_as_u256(x), f.ls, f.rs)) != BP_OK) return rc;
  const Fr4 t = fr_add(fr_add(t0, fr_mul(t1, x)), fr_mul(t2, x2));
  Fr4 zp = fr_mul(z, z), tbl = fr_add(fr_mul(fr_from_bytes(tau2), x2), fr_mul(fr_from_bytes(tau1), x));
  for (uint32_t j = 0; j < m; j++) {
    tbl = fr_add(tbl, fr_mul(zp, fr_from_bytes(blindings + (size_t)j * 32)));
    zp = fr_mul(zp, z);
  }
  (void)m2;
  const Fr4 mu = fr_add(fr_from_bytes(alpha), fr_mul(fr_from_bytes(rho), x));
  fr_to_bytes(tbl, proof_scalars);
  fr_to_bytes(mu, proof_scalars + 32);
  fr_to_bytes(t, proof_scalars + 64);
  // u = gen * shamirU tBlinding mu t   (Prover.hs:105-106): a scalar factor on gen
  uint8_t ub[32];
  Msg mu_(buf, style);
  mu_.r();
  mu_.fr(proof_scalars);
  mu_.fr(proof_scalars + 32);
  mu_.fr(proof_scalars + 64);
  mu_.finish_fr(ub);
  // productProof = IPP.generateProof (gs, hs', u) commitmentLR (ls, rs)   (Prover.hs:108-112); commitmentLR is the
  // commitment of the prover's own ls, rs, t (offset O)
  uint8_t D[64], dot0[32];
  memset(D, 0, 64);
  std::vector<uint8_t> Lv, Rv;
  rc = ipp_prove_fixed_dev(ctx, w, nm, hscale, ub, D, style, Lv, Rv, proof_scalars + 96, proof_scalars + 128, dot0);
  if (rc != BP_OK) return rc;
  if (memcmp(dot0, proof_scalars + 64, 32) != 0) {  // the reference panics here (Prover.hs:94-95)
    ctx->err = "Error on: t = dot l r";
    return BP_ERR_STATE;
  }
  if (k) {
    memcpy(proof_points + 256, Lv.data(), (size_t)k * 64);
    memcpy(proof_points + 256 + (size_t)k * 64, Rv.data(), (size_t)k * 64);
  }
  return BP_OK;
}

}  // namespace bp

extern "C" {

// IPP.generateProof on an arbitrary base (InnerProductProof/Prover.hs:17-26).  n = 0 or a power of two.
// Gs, Hs: n points; u (bH), P (commitmentLR): one point; ls, rs: n scalars.
// Outputs: L, R: log2(n) points each (round order), l, r: 32 bytes each.
int bp_ipp_prove(bp_ctx* ctx, uint32_t n, const uint8_t* Gs, const uint8_t* Hs, const uint8_t* u, const uint8_t* P,
                 const uint8_t* ls, const uint8_t* rs, uint8_t* L_out, uint8_t* R_out, uint8_t* l_out,
                 uint8_t* r_out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n == 0 || (n & (n - 1)) == 0, "n must be 0 or a power of two");
  BP_REQUIRE(n <= (1u << 20), "n too large");
  BP_REQUIRE(u && P && l_out && r_out && (n == 0 || (Gs && Hs && ls && rs)), "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  ProverWs w;
  int rc = carve_ws(ctx, n ? n : 1, w);
  if (rc != BP_OK) return rc;
  if (n) {
    BP_TRY(hipMemcpyAsync(w.G, Gs, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream));
    BP_TRY(hipMemcpyAsync(w.H, Hs, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream));
  }
  std::vector<Fr4> l(n), r(n);
  for (uint32_t i = 0; i < n; i++) {
    l[i] = fr_from_bytes(ls + (size_t)i * 32);
    r[i] = fr_from_bytes(rs + (size_t)i * 32);
  }
  std::vector<uint8_t> Lv, Rv;
  rc = ipp_prove_core(ctx, w, n, u, P, l, r, (ShowStyle)ctx->show_style, Lv, Rv, l_out, r_out);
  if (rc != BP_OK) return rc;
  if (!Lv.empty()) {
    BP_REQUIRE(L_out && R_out, "null pointer");
    memcpy(L_out, Lv.data(), Lv.size());
    memcpy(R_out, Rv.data(), Rv.size());
  }
  return BP_OK;
}

// IPP.verifyProof n base commitmentLR proof (InnerProductProof/Verifier.hs:18-38) on a caller-supplied base.
//   n       the reference's `n` (an Integer there: the vector length the proof is for)
//   Gs, Hs  bGs (n_gs points), bHs (n_hs points); u = bH; P = commitmentLR
//   L, R    k points each = zip lCommits rCommits (:53); l, r: the proof's scalars
// mkChallenges walks the k rounds (hash on the host, x^2 L + x^-2 R + acc on the GPU); mkOtherExponents and the
// scalars of cProof = l <s, Gs> + r <reverse s, Hs> + (l r) bH are a kernel when n is a power of two that the bases
// cover (frvec.hip), else the reference's Map walk on the host (first insert wins; `sumExps` truncates to the shorter
// list, Utils.hs:79); one variable-base MSM of 2n + 1 terms; verdict = (acc == cProof) on affine bytes.
// Where the reference would fail with an index error (fewer rounds than logBase2 n), this returns
// BP_ERR_REFERENCE_FAILS.  Encodings the reference's types cannot hold (a point that is neither O nor on the curve, a
// coordinate >= p, l or r >= r) do not verify (validate.h), as in the batch verifier.
int bp_ipp_verify(bp_ctx* ctx, uint64_t n64, const uint8_t* Gs, uint32_t n_gs, const uint8_t* Hs, uint32_t n_hs,
                  const uint8_t* u, const uint8_t* P, const uint8_t* L, const uint8_t* R, uint32_t k, const uint8_t* l,
                  const uint8_t* r, uint8_t* verdict) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(u && P && l && r && verdict, "null pointer");
  BP_REQUIRE(n64 <= (1u << 20) && n_gs <= (1u << 20) && n_hs <= (1u << 20), "n too large (limit 2^20)");
  BP_REQUIRE(k <= 64 && (k == 0 || (L && R)), "bad round count or null pointer");
  BP_REQUIRE((n_gs == 0 || Gs) && (n_hs == 0 || Hs), "null pointer");
  const uint32_t n = (uint32_t)n64;
  // logBase2 n = floor (logBase 2.0 n) (Utils.hs:93-94); for n = 0 the reference's range [0 .. logBase2 n - 1] is empty
  uint32_t lg = 0;
  while (n >= 2 && (2u << lg) <= n) lg++;
  if (!point_encoding_valid(u) || !point_encoding_valid(P) || !points_encoding_valid(L, k) ||
      !points_encoding_valid(R, k) || !scalar_encoding_valid(l) || !scalar_encoding_valid(r)) {
    *verdict = 0;
    return BP_OK;
  }
  if (n >= 2 && lg > k) {
    ctx->err = "bp_ipp_verify: fewer L/R rounds than logBase2 n (the reference fails with an index error here)";
    return BP_ERR_REFERENCE_FAILS;
  }
  BP_TRY(hipSetDevice(ctx->device));
  const ShowStyle style = (ShowStyle)ctx->show_style;
  hipStream_t st = ctx->stream;
  const uint32_t cap = n > n_gs ? (n > n_hs ? n : n_hs) : (n_gs > n_hs ? n_gs : n_hs);
  ProverWs w;
  int rc = carve_ws(ctx, cap ? cap : 1, w);
  if (rc != BP_OK) return rc;
  // mkChallenges (:40-53)
  uint8_t acc[64];
  memcpy(acc, P, 64);
  std::vector<Fr4> chal;
  if ((rc = mk_challenges_dev(ctx, w, style, acc, L, R, k, chal)) != BP_OK) return rc;
  Fr4 prod = fr_one();
  for (uint32_t j = 0; j < k; j++) prod = fr_mul(prod, chal[j]);
  const Fr4 s0 = fr_inv(prod);  // recip (product challenges); recip 0 = 0 cannot arise from a hash in [0, r) but stays total
  const Fr4 lm = fr_from_bytes(l), rm = fr_from_bytes(r);
  uint8_t lc[32], rcn[32];
  fr_to_bytes(lm, lc);
  fr_to_bytes(rm, rcn);
  size_t a = 0, b = 0;  // terms taken from Gs, Hs
  const bool pow2 = n >= 1 && (n & (n - 1)) == 0;
  if (pow2 && n_gs >= n && n_hs >= n && lg <= 24) {
    a = b = n;
    IppVArgs va;
    memset((void*)&va, 0, sizeof(va));
    for (uint32_t j = 0; j < lg; j++) va.c2_rev[j] = fr4_as_u256(fr_sqr(chal[k - 1 - j]));  // challenges !! j
    va.s0 = fr4_as_u256(s0);
    va.l_plain = bytes_as_u256(lc);
    va.r_plain = bytes_as_u256(rcn);
    if ((rc = frv_ipp_vscalars(ctx, n, lg, va, w.sc)) != BP_OK) return rc;
  } else {
    // mkOtherExponents as written (:55-73): a Map from index to exponent, first insert wins
    std::vector<Fr4> accm(n ? n : 1, fr_zero());
    std::vector<uint8_t> present(n ? n : 1, 0);
    accm[0] = s0;
    present[0] = 1;
    for (uint32_t i = 0; i < n / 2; i++)
      for (uint32_t j = 0; j < lg; j++) {
        const uint64_t i1 = ((uint64_t)1 << j) + i;
        if (i1 >= n || present[i1]) continue;
        if (!present[i]) {
          ctx->err = "bp_ipp_verify: mkOtherExponents reads a missing key (the reference fails here)";
          return BP_ERR_REFERENCE_FAILS;
        }
        accm[i1] = fr_mul(accm[i], fr_sqr(chal[k - 1 - j]));
        present[i1] = 1;
      }
    std::vector<Fr4> other;  // Map.elems: ascending keys
    for (uint32_t i = 0; i < (n ? n : 1); i++)
      if (present[i]) other.push_back(accm[i]);
    const size_t len = other.size();
    a = len < n_gs ? len : n_gs;  // sumExps otherExponents bGs truncates to the shorter list (Utils.hs:79)
    b = len < n_hs ? len : n_hs;  // sumExps (reverse otherExponents) bHs
    std::vector<uint8_t> sc((a + b + 1) * 32);
    for (size_t t = 0; t < a; t++) fr_to_bytes(fr_mul(lm, other[t]), &sc[t * 32]);
    for (size_t t = 0; t < b; t++) fr_to_bytes(fr_mul(rm, other[len - 1 - t]), &sc[(a + t) * 32]);
    fr_to_bytes(fr_mul(lm, rm), &sc[(a + b) * 32]);
    BP_TRY(hipMemcpyAsync(w.sc, sc.data(), sc.size(), hipMemcpyHostToDevice, st));
    BP_TRY(hipStreamSynchronize(st));  // `sc` is a host temporary
  }
  if (a) BP_TRY(hipMemcpyAsync(w.pts, Gs, a * 64, hipMemcpyHostToDevice, st));
  if (b) BP_TRY(hipMemcpyAsync(w.pts + a * 64, Hs, b * 64, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(w.pts + (a + b) * 64, u, 64, hipMemcpyHostToDevice, st));
  uint32_t* d_bad = (uint32_t*)(w.small + 11 * 64);  // the bases are untrusted too: checked where they landed
  if ((rc = points_invalid_dev(ctx, w.pts, a + b, d_bad)) != BP_OK) return rc;
  uint8_t* d_out = w.small + 15 * 64;
  if ((rc = msm_run(ctx, w.sc, w.pts, a + b + 1, 0, d_out, nullptr)) != BP_OK) return rc;
  uint8_t cproof[64];
  uint32_t bad = 0;
  BP_TRY(hipMemcpyAsync(cproof, d_out, 64, hipMemcpyDeviceToHost, st));
  BP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));
  *verdict = (!bad && memcmp(cproof, acc, 64) == 0) ? 1 : 0;  // c == cProof   (:27)
  return BP_OK;
}

// `generateProof` validation (MultiRangeProof/Prover.hs:25-38): upper_bound is a little-endian integer.
// Returns BP_OK and n = log2(upper_bound), or BP_ERR_BOUND_TOO_LARGE / BP_ERR_N_NOT_POW2.
int bp_upper_bound_bits(const uint8_t* upper_bound, size_t len, uint32_t* n_bits) {
  if (!upper_bound || !n_bits) return BP_ERR_ARG;
  // upperBound < _r ?
  uint8_t ub[40] = {0};
  size_t top = len;
  while (top > 0 && upper_bound[top - 1] == 0) top--;
  if (top > 32) return BP_ERR_BOUND_TOO_LARGE;
  memcpy(ub, upper_bound, top);
  uint64_t v[4];
  memcpy(v, ub, 32);
  if (fr_geq_m(v)) return BP_ERR_BOUND_TOO_LARGE;
  // isLogBase2 upperBound, then isLogBase2 n   (Utils.hs:87-100)
  int bits = 0, ones = 0;
  for (int i = 0; i < 256; i++)
    if ((v[i >> 6] >> (i & 63)) & 1) {
      ones++;
      bits = i;
    }
  if (ones != 1) return BP_ERR_N_NOT_POW2;
  uint32_t n = (uint32_t)bits;
  if (n == 0 || (n & (n - 1)) != 0) return BP_ERR_N_NOT_POW2;
  *n_bits = n;
  return BP_OK;
}

// MultiRangeProof.generateProof (checked = 1) / generateProofUnsafe (checked = 0) for m values of n_bits bits.
// values, blindings: m x 32 B.  sL, sR: n_bits * m2 scalars each (m2 = m padded to a power of two with (0, 0),
// Prover.hs:41-43); alpha = aBlinding, rho = sBlinding, tau1, tau2 = t1Blinding, t2Blinding.
// proof_scalars: 5 x 32 B (tBlinding, mu, t, l, r); proof_points: (4 + 2k) x 64 B (A, S, T1, T2, L.., R..),
// k = log2(n_bits * m2).  Returns BP_ERR_NOT_IN_RANGE when checked and some value >= 2^n_bits.
int bp_rangeproof_prove(bp_ctx* ctx, uint32_t n_bits, uint32_t m, const uint8_t* values, const uint8_t* blindings,
                        const uint8_t* sL, const uint8_t* sR, const uint8_t* alpha, const uint8_t* rho,
                        const uint8_t* tau1, const uint8_t* tau2, int checked, uint8_t* proof_scalars,
                        uint8_t* proof_points) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(values && blindings && sL && sR && alpha && rho && tau1 && tau2 && proof_scalars && proof_points,
             "null pointer");
  BP_REQUIRE(n_bits >= 1 && n_bits <= 128 && (n_bits & (n_bits - 1)) == 0, "n_bits must be a power of two <= 128");
  BP_REQUIRE(m >= 1, "m must be >= 1");
  uint32_t m2 = 1;
  while (m2 < m) m2 <<= 1;
  const uint32_t n = n_bits, nm = n * m2;
  if (ctx->n_gens == 0) {
    ctx->err = "generators not set (bp_generators_derive / bp_ctx_set_generators)";
    return BP_ERR_STATE;
  }
  BP_REQUIRE(nm <= ctx->n_gens, "n_bits * m exceeds the generator count of the context");
  uint32_t k = 0;
  while ((1u << k) < nm) k++;
  if (checked) {  // checkRanges (RangeProof/Internal.hs:164-165)
    for (uint32_t j = 0; j < m; j++) {
      const uint8_t* v = values + (size_t)j * 32;
      for (uint32_t bit = n; bit < 256; bit++)
        if ((v[bit >> 3] >> (bit & 7)) & 1) {
          ctx->err = "ValuesNotInRange";
          return BP_ERR_NOT_IN_RANGE;
        }
    }
  }
  BP_TRY(hipSetDevice(ctx->device));
  const ShowStyle style = (ShowStyle)ctx->show_style;
  hipStream_t st = ctx->stream;
  const uint32_t ng = ctx->n_gens;
  // aL = reversedEncodeBitMulti n vs, aR = aL - 1   (RangeProof/Internal.hs:73-88): the binary digits of each
  // value, least significant first, zero-filled to n.  fillWithZeros never truncates, so an out-of-range value
  // (possible only when unchecked) contributes all its digits and shifts the ones after it; the vector algebra
  // below then zips down to nm entries while A still commits to every digit (sumExps against the infinite gs/hs).
  const Fr4 one = fr_one(), zero = fr_zero(), minus1 = fr_neg(one);
  std::vector<Fr4> aL, aR;
  aL.reserve(nm);
  for (uint32_t j = 0; j < m2; j++) {
    uint8_t v[32] = {0};
    if (j < m) fr_to_bytes(fr_from_bytes(values + (size_t)j * 32), v);  // Fr: canonical residue
    uint32_t len = n;
    for (uint32_t bit = 255; bit >= n; bit--)
      if ((v[bit >> 3] >> (bit & 7)) & 1) {
        len = bit + 1;
        break;
      }
    for (uint32_t b = 0; b < len; b++) aL.push_back(((v[b >> 3] >> (b & 7)) & 1) ? one : zero);
  }
  const uint32_t n_digits = (uint32_t)aL.size();  // == nm unless some value is out of range
  BP_REQUIRE(n_digits <= ng, "out-of-range values need more generators than the context holds");
  aR.resize(n_digits);
  for (uint32_t i = 0; i < n_digits; i++) aR[i] = fr_is_zero(aL[i]) ? minus1 : zero;
  ProverWs w;
  int rc = carve_ws(ctx, n_digits, w);
  if (rc != BP_OK) return rc;
  if (prover_uses_fixed_base() && prover_vectors_on_device() && n_digits == nm) {
    std::vector<uint8_t> bits(nm);
    for (uint32_t i = 0; i < nm; i++) bits[i] = fr_is_zero(aL[i]) ? 0 : 1;
    return rangeproof_prove_dev(ctx, w, n, m, m2, nm, k, bits, blindings, sL, sR, alpha, rho, tau1, tau2, proof_scalars,
                                proof_points);
  }
  const uint8_t* d_gs = ctx->d_bases;
  const uint8_t* d_hs = ctx->d_bases + (size_t)ng * 64;
  const uint8_t* d_gen = ctx->d_bases + (size_t)2 * ng * 64;
  const uint8_t* d_h = ctx->d_bases + (size_t)(2 * ng + 1) * 64;
  uint8_t gen_h[128];
  BP_TRY(hipMemcpyAsync(gen_h, d_gen, 128, hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));
  std::vector<Fr4> sLv(nm), sRv(nm);
  for (uint32_t i = 0; i < nm; i++) {
    sLv[i] = fr_from_bytes(sL + (size_t)i * 32);
    sRv[i] = fr_from_bytes(sR + (size_t)i * 32);
  }
  uint8_t* A = proof_points;
  uint8_t* S = proof_points + 64;
  uint8_t* T1 = proof_points + 128;
  uint8_t* T2 = proof_points + 192;
  // commitBitVectors: A = aBlinding*h + <aL, gs> + <aR, hs>, S likewise   (RangeProof/Internal.hs:136-147)
  if (prover_uses_fixed_base() && n_digits == nm) {  // both commitments through the fixed-base tables, one launch
    std::vector<uint32_t> tb(2 * (size_t)nm + 1);
    for (uint32_t i = 0; i < nm; i++) {
      tb[i] = fb_base_g(i);
      tb[nm + i] = fb_base_h(ng, i);
    }
    tb[2 * (size_t)nm] = fb_base_H(ng);
    std::vector<Fr4> scA(aL), scS(sLv);
    scA.insert(scA.end(), aR.begin(), aR.end());
    scA.push_back(fr_from_bytes(alpha));
    scS.insert(scS.end(), sRv.begin(), sRv.end());
    scS.push_back(fr_from_bytes(rho));
    const std::vector<Fr4>* scs[2] = {&scA, &scS};
    uint8_t* outs[2] = {A, S};
    if ((rc = fb_msm_host(ctx, w, tb, 2, scs, outs)) != BP_OK) return rc;
  } else {  // both commitments in one launch
    MsmSeg segA[3] = {{d_gs, n_digits, true}, {d_hs, n_digits, true}, {d_h, 1, true}};
    MsmSeg segS[3] = {{d_gs, nm, true}, {d_hs, nm, true}, {d_h, 1, true}};
    std::vector<Fr4> scA(aL), scS(sLv);
    scA.insert(scA.end(), aR.begin(), aR.end());
    scA.push_back(fr_from_bytes(alpha));
    scS.insert(scS.end(), sRv.begin(), sRv.end());
    scS.push_back(fr_from_bytes(rho));
    MsmJob jobs[2] = {{segA, 3, &scA, A}, {segS, 3, &scS, S}};
    if ((rc = msm_multi(ctx, w, jobs, 2)) != BP_OK) return rc;
  }
  // y, z   (Prover.hs:75-76)
  uint8_t buf[MSG_SCRATCH], keep[MSG_SCRATCH], yb[32], zb[32], xb[32];
  Msg my(buf, style);
  my.r();
  my.point(A);
  my.point(S);
  size_t pre = my.len;
  memcpy(keep, buf, pre);
  my.finish_fr(yb);
  Msg mz(buf, style);
  mz.raw(keep, pre);
  mz.fr(yb);
  mz.finish_fr(zb);
  const Fr4 y = fr_from_bytes(yb), z = fr_from_bytes(zb);
  // computeLRPolys / computeTPoly   (Prover.hs:133-169)
  std::vector<Fr4> l0(nm), r0(nm), r1(nm), ypow(nm), zpow(m2 + 3);
  zpow[0] = one;
  for (uint32_t e = 1; e < m2 + 3; e++) zpow[e] = fr_mul(zpow[e - 1], z);
  {
    Fr4 yi = one, two = fr_add(one, one);
    for (uint32_t j = 0; j < m2; j++) {
      Fr4 two_k = one;
      for (uint32_t b = 0; b < n; b++) {
        uint32_t i = j * n + b;
        ypow[i] = yi;  // powerVector y nm (y = 0 has probability 2^-256; its 0^0 = 0 quirk is not reproduced)
        l0[i] = fr_sub(aL[i], z);
        r0[i] = fr_add(fr_mul(yi, fr_add(aR[i], z)), fr_mul(zpow[j + 2], two_k));
        r1[i] = fr_mul(yi, sRv[i]);
        yi = fr_mul(yi, y);
        two_k = fr_mul(two_k, two);
      }
    }
  }
  Fr4 t0 = zero, t1 = zero, t2 = zero;
  for (uint32_t i = 0; i < nm; i++) {
    t0 = fr_add(t0, fr_mul(l0[i], r0[i]));
    t2 = fr_add(t2, fr_mul(sLv[i], r1[i]));
    t1 = fr_add(t1, fr_add(fr_mul(sLv[i], r0[i]), fr_mul(l0[i], r1[i])));
  }
  // t1Commit = commit t1 t1Blinding, t2Commit likewise   (Prover.hs:84-85; Utils.hs:84-85)
  if (prover_uses_fixed_base()) {
    std::vector<uint32_t> tb{fb_base_G(ng), fb_base_H(ng)};
    std::vector<Fr4> s1{t1, fr_from_bytes(tau1)}, s2{t2, fr_from_bytes(tau2)};
    const std::vector<Fr4>* scs[2] = {&s1, &s2};
    uint8_t* outs[2] = {T1, T2};
    if ((rc = fb_msm_host(ctx, w, tb, 2, scs, outs)) != BP_OK) return rc;
  } else {
    MsmSeg seg[1] = {{gen_h, 2, false}};
    std::vector<Fr4> s1{t1, fr_from_bytes(tau1)}, s2{t2, fr_from_bytes(tau2)};
    MsmJob jobs[2] = {{seg, 1, &s1, T1}, {seg, 1, &s2, T2}};
    if ((rc = msm_multi(ctx, w, jobs, 2)) != BP_OK) return rc;
  }
  // x   (Prover.hs:88)
  Msg mx(buf, style);
  mx.raw(keep, pre);
  mx.point(T1);
  mx.point(T2);
  mx.fr(yb);
  mx.fr(zb);
  mx.finish_fr(xb);
  const Fr4 x = fr_from_bytes(xb), x2 = fr_sqr(x);
  // ls, rs, t, tBlinding, mu   (Prover.hs:90-103)
  std::vector<Fr4> ls(nm), rs(nm);
  Fr4 t = fr_add(fr_add(t0, fr_mul(t1, x)), fr_mul(t2, x2));
  Fr4 dotlr = zero;
  for (uint32_t i = 0; i < nm; i++) {
    ls[i] = fr_add(l0[i], fr_mul(x, sLv[i]));
    rs[i] = fr_add(r0[i], fr_mul(x, r1[i]));
    dotlr = fr_add(dotlr, fr_mul(ls[i], rs[i]));
  }
  if (!fr_eq(t, dotlr)) {  // the reference panics here (Prover.hs:94-95)
    ctx->err = "Error on: t = dot l r";
    return BP_ERR_STATE;
  }
  Fr4 tb = fr_add(fr_mul(fr_from_bytes(tau2), x2), fr_mul(fr_from_bytes(tau1), x));
  for (uint32_t j = 0; j < m; j++) tb = fr_add(tb, fr_mul(zpow[j + 2], fr_from_bytes(blindings + (size_t)j * 32)));
  Fr4 mu = fr_add(fr_from_bytes(alpha), fr_mul(fr_from_bytes(rho), x));
  fr_to_bytes(tb, proof_scalars);
  fr_to_bytes(mu, proof_scalars + 32);
  fr_to_bytes(t, proof_scalars + 64);
  // u = gen * shamirU tBlinding mu t   (Prover.hs:105-106)
  uint8_t ub[32], u[64];
  Msg mu_(buf, style);
  mu_.r();
  mu_.fr(proof_scalars);
  mu_.fr(proof_scalars + 32);
  mu_.fr(proof_scalars + 64);
  mu_.finish_fr(ub);
  std::vector<uint8_t> Lv, Rv;
  if (prover_uses_fixed_base()) {
    // hs' = hs_i * recip(y^i) (Prover.hs:107) enters as scalar factors; commitmentLR (Prover.hs:108) is, for the
    // prover's own ls, rs, t, the commitment <ls, gs> + <rs, hs'> + t*u (expand A, S, mu in
    // RangeProof/Internal.hs:183-198) plus the digits of out-of-range values that only aCommit carries
    std::vector<Fr4> hscale(nm);
    Fr4 yinv = fr_inv(y), cur = one;
    for (uint32_t i = 0; i < nm; i++) {
      hscale[i] = cur;
      cur = fr_mul(cur, yinv);
    }
    uint8_t D[64];
    memset(D, 0, 64);
    if (n_digits > nm) {
      uint8_t* d_u = w.small;
      BP_TRY(hipMemcpyAsync(d_u, gen_h, 64, hipMemcpyHostToDevice, st));  // any point: its scalar is zero
      if ((rc = msm3(ctx, w, d_gs + (size_t)nm * 64, aL, nm, d_hs + (size_t)nm * 64, aR, nm, n_digits - nm, d_u, zero,
                     D)) != BP_OK)
        return rc;
    }
    // productProof = IPP.generateProof (gs, hs', u) commitmentLR (ls, rs)   (Prover.hs:109-112)
    rc = ipp_prove_fixed(ctx, w, nm, hscale, ub, D, ls, rs, style, Lv, Rv, proof_scalars + 96, proof_scalars + 128);
    if (rc != BP_OK) return rc;
  } else {
    {
      MsmSeg seg[1] = {{gen_h, 1, false}};
      std::vector<Fr4> s1{fr_from_bytes(ub)};
      MsmJob jobs[1] = {{seg, 1, &s1, u}};
      if ((rc = msm_multi(ctx, w, jobs, 1)) != BP_OK) return rc;
    }
    // hs' = hs_i * recip(y^i)   (Prover.hs:107): one batched inversion on the host, one GLV kernel
    {
      std::vector<uint8_t> sc((size_t)nm * 32);
      Fr4 yinv = fr_inv(y), cur = one;
      for (uint32_t i = 0; i < nm; i++) {
        fr_to_bytes(cur, &sc[(size_t)i * 32]);
        cur = fr_mul(cur, yinv);
      }
      BP_TRY(hipMemcpyAsync(w.sc, sc.data(), sc.size(), hipMemcpyHostToDevice, st));
      if ((rc = ipp_scale_points(ctx, d_hs, w.sc, nm, w.tab, w.H)) != BP_OK) return rc;
      BP_TRY(hipMemcpyAsync(w.G, d_gs, (size_t)nm * 64, hipMemcpyDeviceToDevice, st));
      BP_TRY(hipStreamSynchronize(st));
    }
    // commitmentLR = computeLRCommitment ...   (Prover.hs:108).  For the prover's own ls, rs, t this is the
    // same group element as <ls, gs> + <rs, hs'> + t*u (expand A, S, mu in RangeProof/Internal.hs:183-198),
    // which is one MSM.
    uint8_t P[64];
    {
      uint8_t* d_u = w.small;
      BP_TRY(hipMemcpyAsync(d_u, u, 64, hipMemcpyHostToDevice, st));
      if ((rc = msm3(ctx, w, w.G, ls, 0, w.H, rs, 0, nm, d_u, t, P)) != BP_OK) return rc;
      if (n_digits > nm) {  // digits of out-of-range values that only aCommit carries
        uint8_t extra[64], sum[64];
        if ((rc = msm3(ctx, w, d_gs + (size_t)nm * 64, aL, nm, d_hs + (size_t)nm * 64, aR, nm, n_digits - nm, d_u, zero,
                       extra)) != BP_OK)
          return rc;
        if ((rc = bp_point_add(ctx, P, extra, 1, sum)) != BP_OK) return rc;
        memcpy(P, sum, 64);
      }
    }
    // productProof = IPP.generateProof (gs, hs', u) commitmentLR (ls, rs)   (Prover.hs:109-112)
    rc = ipp_prove_core(ctx, w, nm, u, P, ls, rs, style, Lv, Rv, proof_scalars + 96, proof_scalars + 128);
    if (rc != BP_OK) return rc;
  }
  if (k) {
    memcpy(proof_points + 256, Lv.data(), (size_t)k * 64);
    memcpy(proof_points + 256 + (size_t)k * 64, Rv.data(), (size_t)k * 64);
  }
  return BP_OK;
}

// ---- self-test hooks of the scalar-field vector kernels (frvec.hip), used by the parity tests ----------------------
// op 0: out[0..n) = powerVector a n (Utils.hs:50-52; a: one scalar)   1: out[0..n) = hadamard a b (n each, :55-58)
// op 2: out[0] = dot a b (:60-61).  Canonical 32-byte scalars in host memory.
int bp_fr_vector_op(bp_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint32_t n, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(op >= 0 && op <= 2 && a && out && (op == 0 || b), "bad selector or null pointer");
  BP_REQUIRE(n <= (1u << 20), "n too large");
  BP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  ProverWs w;
  int rc = carve_ws(ctx, n ? n : 1, w);
  if (rc != BP_OK) return rc;
  FrDev& f = w.fr;
  const size_t vec = (size_t)n * 32;
  uint32_t n_out = n;
  if (op == 0) {
    if ((rc = frv_power_vector(ctx, fr4_as_u256(fr_from_bytes(a)), n, f.v[2])) != BP_OK) return rc;
  } else {
    if (n) {
      BP_TRY(hipMemcpyAsync(f.raw, a, vec, hipMemcpyHostToDevice, st));
      BP_TRY(hipMemcpyAsync(f.raw + vec, b, vec, hipMemcpyHostToDevice, st));
    }
    if ((rc = frv_from_bytes(ctx, f.raw, n, f.v[0])) != BP_OK) return rc;
    if ((rc = frv_from_bytes(ctx, f.raw + vec, n, f.v[1])) != BP_OK) return rc;
    if (op == 1) {
      if ((rc = frv_hadamard(ctx, f.v[0], f.v[1], n, f.v[2])) != BP_OK) return rc;
    } else {
      if ((rc = frv_dot(ctx, f.v[0], f.v[1], n, f.v[2])) != BP_OK) return rc;
      n_out = 1;
    }
  }
  if ((rc = frv_to_bytes(ctx, f.v[2], n_out, f.raw + 2 * vec)) != BP_OK) return rc;
  if (n_out) BP_TRY(hipMemcpyAsync(out, f.raw + 2 * vec, (size_t)n_out * 32, hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));
  return BP_OK;
}

// computeLRPolys / computeTPoly / ls, rs (MultiRangeProof/Prover.hs:90-91,133-169) as the range prover's kernels
// evaluate them.  bits: aL, nm = n_bits * m bytes (0/1); s_l, s_r: nm scalars; y, z, x: one scalar each.
// out: (5 nm + 3) scalars: l0 | r0 | r1 | ls | rs | t0 t1 t2.
int bp_rp_polys_selftest(bp_ctx* ctx, uint32_t n_bits, uint32_t m, const uint8_t* bits, const uint8_t* s_l,
                         const uint8_t* s_r, const uint8_t* y, const uint8_t* z, const uint8_t* x, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(bits && s_l && s_r && y && z && x && out, "null pointer");
  BP_REQUIRE(n_bits >= 1 && m >= 1 && (uint64_t)n_bits * m <= 8192, "n_bits * m out of range");
  const uint32_t nm = n_bits * m;
  BP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  ProverWs w;
  int rc = carve_ws(ctx, nm, w);
  if (rc != BP_OK) return rc;
  FrDev& f = w.fr;
  const size_t vec = (size_t)nm * 32;
  BP_TRY(hipMemcpyAsync(f.bits, bits, nm, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(f.raw, s_l, vec, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(f.raw + vec, s_r, vec, hipMemcpyHostToDevice, st));
  u256 *l0 = f.v[0], *r0 = f.v[1], *r1 = f.v[2], *l1 = f.v[3], *sRm = f.v[4];
  if ((rc = frv_from_bytes(ctx, f.raw, nm, l1)) != BP_OK) return rc;
  if ((rc = frv_from_bytes(ctx, f.raw + vec, nm, sRm)) != BP_OK) return rc;
  if ((rc = frv_rp_polys(ctx, n_bits, nm, f.bits, sRm, fr4_as_u256(fr_from_bytes(y)), fr4_as_u256(fr_from_bytes(z)), l0, r0,
                         r1)) != BP_OK)
    return rc;
  if ((rc = frv_rp_tpoly(ctx, nm, l0, l1, r0, r1, f.sm + 8)) != BP_OK) return rc;
  if ((rc = frv_rp_eval(ctx, nm, l0, l1, r0, r1, fr4_as_u256(fr_from_bytes(x)), f.ls, f.rs)) != BP_OK) return rc;
  const u256* vs[5] = {l0, r0, r1, f.ls, f.rs};
  for (int i = 0; i < 5; i++) {
    if ((rc = frv_to_bytes(ctx, vs[i], nm, f.raw + 2 * vec)) != BP_OK) return rc;
    BP_TRY(hipMemcpyAsync(out + (size_t)i * vec, f.raw + 2 * vec, vec, hipMemcpyDeviceToHost, st));
  }
  BP_TRY(hipMemcpyAsync(out + 5 * vec, f.sm + 8, 96, hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));
  return BP_OK;
}

// `rounds` recursion steps of generateProof' (InnerProductProof/Prover.hs:52-92) on the witness alone, as the
// provers' kernels run them: ls, rs (n scalars, n a power of two), hscale (n: the factors y^-i on hs), ub (bH = ub * gen),
// xs: the challenges of the rounds.  out: nk = n >> rounds; ls' (nk) | rs' (nk) | dot, cL, cR of the folded
// witness (3) | the scalars of the three sums over the 2n + 1 original generators (3 (2n + 1)).
int bp_ipp_fold_selftest(bp_ctx* ctx, uint32_t n, const uint8_t* ls, const uint8_t* rs, const uint8_t* hscale,
                         const uint8_t* ub, uint32_t rounds, const uint8_t* xs, uint8_t* out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(ls && rs && hscale && ub && out && (rounds == 0 || xs), "null pointer");
  BP_REQUIRE(n >= 1 && n <= 8192 && (n & (n - 1)) == 0 && (n >> rounds) >= 1, "n must be a power of two >= 2^rounds");
  BP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  ProverWs w;
  int rc = carve_ws(ctx, n, w);
  if (rc != BP_OK) return rc;
  FrDev& f = w.fr;
  const size_t vec = (size_t)n * 32, nt = (size_t)2 * n + 1;
  BP_TRY(hipMemcpyAsync(f.raw, ls, vec, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(f.raw + vec, rs, vec, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(f.raw + 2 * vec, hscale, vec, hipMemcpyHostToDevice, st));
  if ((rc = frv_from_bytes(ctx, f.raw, n, f.ls)) != BP_OK) return rc;
  if ((rc = frv_from_bytes(ctx, f.raw + vec, n, f.rs)) != BP_OK) return rc;
  if ((rc = frv_from_bytes(ctx, f.raw + 2 * vec, n, f.v[0])) != BP_OK) return rc;
  if ((rc = frv_ipp_init(ctx, n, f.v[0], f.cg, f.ch)) != BP_OK) return rc;
  const u256 ubm = fr4_as_u256(fr_from_bytes(ub)), zero = u256_zero();
  uint32_t nk = n;
  if ((rc = frv_ipp_round(ctx, n, nk, false, zero, zero, ubm, f.ls, f.rs, f.cg, f.ch, w.sc, f.sm, true)) != BP_OK) return rc;
  for (uint32_t r = 0; r < rounds; r++) {
    const Fr4 x = fr_from_bytes(xs + (size_t)r * 32), xi = fr_inv(x);
    if ((rc = frv_ipp_round(ctx, n, nk, true, fr4_as_u256(x), fr4_as_u256(xi), ubm, f.ls, f.rs, f.cg, f.ch, w.sc, f.sm,
                            true)) != BP_OK)
      return rc;
    nk /= 2;
  }
  const size_t vk = (size_t)nk * 32;
  if ((rc = frv_to_bytes(ctx, f.ls, nk, f.raw)) != BP_OK) return rc;
  if ((rc = frv_to_bytes(ctx, f.rs, nk, f.raw + vk)) != BP_OK) return rc;
  BP_TRY(hipMemcpyAsync(out, f.raw, 2 * vk, hipMemcpyDeviceToHost, st));
  // dot, cL, cR: recover them from the bH scalars would need ub^-1; sm[0] holds dot, the sums' bH entries hold ub * {dot, cL, cR}
  BP_TRY(hipMemcpyAsync(out + 2 * vk, f.sm, 32, hipMemcpyDeviceToHost, st));
  BP_TRY(hipMemcpyAsync(out + 2 * vk + 32, w.sc + (nt + 2 * (size_t)n) * 32, 32, hipMemcpyDeviceToHost, st));
  BP_TRY(hipMemcpyAsync(out + 2 * vk + 64, w.sc + (2 * nt + 2 * (size_t)n) * 32, 32, hipMemcpyDeviceToHost, st));
  BP_TRY(hipMemcpyAsync(out + 2 * vk + 96, w.sc, 3 * nt * 32, hipMemcpyDeviceToHost, st));
  BP_TRY(hipStreamSynchronize(st));
  return BP_OK;
}

}  // extern "C"
