// Host control of the arithmetic-circuit protocol, in C++ above the engine:
//   ArithmeticCircuit.generateProof   (/root/reference/Bulletproofs/ArithmeticCircuit/Prover.hs:16-117)
//   ArithmeticCircuit.verifyProof     (ArithmeticCircuit/Verifier.hs:16-71)
//   padCircuit / padAssignment / delta / commitBitVector / shamirGxGxG / shamirGs / shamirZ
//                                     (ArithmeticCircuit/Internal.hs:76-117)
// Every sumExps / commitBitVector / commitmentLR is one bucket MSM on the GPU over the context's
// device-resident gs / hs; hs' and the IPP folds are the GLV kernels of ipp.hip; the accumulator
// walk of mkChallenges runs k_lincomb2_add once per round.  The O(Q n) weight algebra, the
// polynomial coefficients and the Fiat-Shamir transcript stay on the host.
// Randomness: aiBlinding, aoBlinding, sBlinding, sL, sR and the five
// tBlindings are explicit arguments in the reference's draw order.
#include "prover.h"
#include "fixedbase.h"
#include "glv.h"
#include "host_inv.h"
#include "validate.h"

using namespace bp;
using namespace bp::host;

namespace {

constexpr size_t AC_MSG = 2048;  // shamirGs over 7 points is the longest transcript here

// `powerVector a n` (Utils.hs:50-52), including 0^0 = 0
std::vector<Fr4> power_vector(const Fr4& a, uint32_t n) {
  std::vector<Fr4> v(n);
  if (fr_is_zero(a)) {
    for (auto& e : v) e = fr_zero();
    return v;
  }
  Fr4 cur = fr_one();
  for (uint32_t i = 0; i < n; i++) {
    v[i] = cur;
    cur = fr_mul(cur, a);
  }
  return v;
}

uint32_t pad_pow2(uint32_t n) {  // padToNearestPowerOfTwo (Utils.hs:107-121): [] stays []
  if (n == 0) return 0;
  uint32_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

// Everything both sides derive from (circuit, y, z): ArithmeticCircuit/Prover.hs:37-42 == Verifier.hs:32-36
struct Derived {
  std::vector<Fr4> zs, ys, yinv, zwL, zwR, zwO;
};

// zs `vectorMatrixProduct` W for a dense q x n matrix of canonical scalars, zero-padded to n2 columns
std::vector<Fr4> zs_times(const std::vector<Fr4>& zs, const uint8_t* W, uint32_t q, uint32_t n, uint32_t n2) {
  std::vector<Fr4> out(n2, fr_zero());
  for (uint32_t r = 0; r < q; r++)
    for (uint32_t c = 0; c < n; c++) {
      const uint8_t* e = W + ((size_t)r * n + c) * 32;
      bool zero = true;
      for (int b = 0; b < 32; b++)
        if (e[b]) {
          zero = false;
          break;
        }
      if (!zero) out[c] = fr_add(out[c], fr_mul(zs[r], fr_from_bytes(e)));
    }
  return out;
}

Derived derive(uint32_t n, uint32_t n2, uint32_t q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
               const Fr4& y, const Fr4& z) {
  Derived d;
  std::vector<Fr4> zp = power_vector(z, q + 1);
  d.zs.assign(zp.begin() + 1, zp.end());  // drop 1 (powerVector z (qLen + 1))
  d.ys = power_vector(y, n2);
  d.yinv = power_vector(fr_inv(y), n2);
  d.zwL = zs_times(d.zs, wL, q, n, n2);
  d.zwR = zs_times(d.zs, wR, q, n, n2);
  d.zwO = zs_times(d.zs, wO, q, n, n2);
  return d;
}

void challenge_y_z(ShowStyle style, const uint8_t* AI, const uint8_t* AO, const uint8_t* S, uint8_t yb[32],
                   uint8_t zb[32]) {
  uint8_t buf[AC_MSG];
  Msg my(buf, style);  // shamirGxGxG aiCommit aoCommit sCommit
  my.r();
  my.point(AI);
  my.point(AO);
  my.point(S);
  my.finish_fr(yb);
  Msg mz(buf, style);  // shamirZ y = oracle (show _r <> show y)
  mz.r();
  mz.fr(yb);
  mz.finish_fr(zb);
}

void challenge_x(ShowStyle style, const uint8_t* T7, uint8_t xb[32]) {
  uint8_t buf[AC_MSG];
  Msg mx(buf, style);  // shamirGs tCommits
  mx.r();
  for (int i = 0; i < 7; i++) mx.point(T7 + 64 * i);
  mx.finish_fr(xb);
}

void challenge_u(ShowStyle style, const uint8_t* three_scalars, uint8_t ub[32]) {
  uint8_t buf[AC_MSG];
  Msg m(buf, style);
  m.r();
  m.fr(three_scalars);
  m.fr(three_scalars + 32);
  m.fr(three_scalars + 64);
  m.finish_fr(ub);
}

struct Gens {
  const uint8_t *d_gs, *d_hs, *d_gen, *d_h;
  uint8_t gen_h[128];  // host copies of gen, h
};

int get_gens(bp_ctx* ctx, Gens& g) {
  const uint32_t ng = ctx->n_gens;
  g.d_gs = ctx->d_bases;
  g.d_hs = ctx->d_bases + (size_t)ng * 64;
  g.d_gen = ctx->d_bases + (size_t)2 * ng * 64;
  g.d_h = ctx->d_bases + (size_t)(2 * ng + 1) * 64;
  BP_TRY(hipMemcpyAsync(g.gen_h, g.d_gen, 128, hipMemcpyDeviceToHost, ctx->stream));
  BP_TRY(hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

// hs' = zipWith mul hs (powerVector (recip y) n) into w.H; gs into w.G   (Prover.hs:76, Verifier.hs:38)
int load_ipp_base(bp_ctx* ctx, ProverWs& w, const Gens& g, const std::vector<Fr4>& yinv, uint32_t n2) {
  if (n2 == 0) return BP_OK;
  hipStream_t st = ctx->stream;
  std::vector<uint8_t> sc((size_t)n2 * 32);
  for (uint32_t i = 0; i < n2; i++) fr_to_bytes(yinv[i], &sc[(size_t)i * 32]);
  BP_TRY(hipMemcpyAsync(w.sc, sc.data(), sc.size(), hipMemcpyHostToDevice, st));
  int rc = ipp_scale_points(ctx, g.d_hs, w.sc, n2, w.tab, w.H);
  if (rc != BP_OK) return rc;
  BP_TRY(hipMemcpyAsync(w.G, g.d_gs, (size_t)n2 * 64, hipMemcpyDeviceToDevice, st));
  BP_TRY(hipStreamSynchronize(st));
  return BP_OK;
}

// u = gen * k as a one-term direct MSM (GLV, 129 steps instead of the 256-step ladder of bp_point_mul)
int point_mul_gen(bp_ctx* ctx, ProverWs& w, const Gens& g, const uint8_t k[32], uint8_t out[64]) {
  MsmSeg seg[1] = {{g.d_gen, 1, true}};
  std::vector<Fr4> s1{fr_from_bytes(k)};
  MsmJob jobs[1] = {{seg, 1, &s1, out}};
  return msm_multi(ctx, w, jobs, 1);
}

int check_dims(bp_ctx* ctx, uint32_t n, uint32_t q, uint32_t m, uint32_t& n2, uint32_t& k) {
  BP_REQUIRE(n <= 8192 && q <= (1u << 20) && m <= (1u << 20), "circuit dimensions out of range");
  if (ctx->n_gens == 0) {
    ctx->err = "generators not set (bp_generators_derive / bp_ctx_set_generators)";
    return BP_ERR_STATE;
  }
  n2 = pad_pow2(n);
  BP_REQUIRE(n2 <= ctx->n_gens, "gate count exceeds the generator count of the context");
  k = 0;
  while ((2u << k) <= n2) k++;
  return BP_OK;
}

// generateProof (ArithmeticCircuit/Prover.hs:21-101) with the O(n) and O(Q n) scalar-field vector work on the device
// (frvec.hip): the assignment and blinding vectors are uploaded once and stay in HBM through commitBitVector's
// scalars, zs . W (`vectorMatrixProduct`), computePolynomials, multiplyPoly, evaluatePolynomial and the inner-product
// recursion.  The host keeps the transcript, zs, the O(Q m) commitment-weight sums and the proof's three scalars.
int ac_prove_dev(bp_ctx* ctx, ProverWs& w, uint32_t n, uint32_t n2, uint32_t k, uint32_t q, uint32_t m, const uint8_t* wL,
                 const uint8_t* wR, const uint8_t* wO, const uint8_t* wV, const uint8_t* a_l, const uint8_t* a_r,
                 const uint8_t* a_o, const uint8_t* commit_blinders, const uint8_t* blind3, const uint8_t* s_l,
                 const uint8_t* s_r, const uint8_t* tau5, uint8_t* proof_scalars, uint8_t* proof_points) {
  const ShowStyle style = (ShowStyle)ctx->show_style;
  hipStream_t st = ctx->stream;
  const uint32_t ng = ctx->n_gens;
  FrDev& f = w.fr;
  int rc;
  const Fr4 zero = fr_zero();
  uint8_t* AI = proof_points;
  uint8_t* AO = proof_points + 64;
  uint8_t* S = proof_points + 128;
  uint8_t* T = proof_points + 192;  // 7 points
  u256 *aL = f.v[0], *aR = f.v[1], *aO = f.v[2], *sL = f.v[3], *sR = f.v[4], *zwL = f.v[5], *zwR = f.v[6], *zwO = f.v[7];
  u256 *l1 = f.v[8], *r0 = f.v[9], *r1 = f.v[10], *r3 = f.v[11];
  u256* hscale = f.v[5];  // after computePolynomials the zs . W vectors are dead
  const uint32_t nterms = 2 * n2 + 1;
  const size_t tb_bytes = ((size_t)nterms * 4 + 255) & ~(size_t)255;
  if ((size_t)3 * nterms * 32 > 4 * w.seg_cap * 32 || tb_bytes + fb_msm_partials(ctx, 3, nterms) * sizeof(xyzz) > 4 * w.seg_cap * 64) {
    ctx->err = "ac_prove_dev: prover workspace too small";
    return BP_ERR_STATE;
  }
  // padAssignment (Internal.hs:91-97): zero-filled to n2; sL, sR already have n2 entries (chooseBlindingVectors n, :32)
  const size_t vec = (size_t)n2 * 32;
  BP_TRY(hipMemsetAsync(f.raw, 0, 3 * vec, st));
  if (n) {
    BP_TRY(hipMemcpyAsync(f.raw, a_l, (size_t)n * 32, hipMemcpyHostToDevice, st));
    BP_TRY(hipMemcpyAsync(f.raw + vec, a_r, (size_t)n * 32, hipMemcpyHostToDevice, st));
    BP_TRY(hipMemcpyAsync(f.raw + 2 * vec, a_o, (size_t)n * 32, hipMemcpyHostToDevice, st));
  }
  BP_TRY(hipMemcpyAsync(f.raw + 3 * vec, s_l, vec, hipMemcpyHostToDevice, st));
  BP_TRY(hipMemcpyAsync(f.raw + 4 * vec, s_r, vec, hipMemcpyHostToDevice, st));
  u256* vin[5] = {aL, aR, aO, sL, sR};
  for (int i = 0; i < 5; i++)
    if ((rc = frv_from_bytes(ctx, f.raw + (size_t)i * vec, n2, vin[i])) != BP_OK) return rc;
  // commitBitVector (Internal.hs:102-107): aiCommit, aoCommit (vR = []), sCommit   (Prover.hs:29-33)
  uint32_t* d_tb = (uint32_t*)w.pts;
  xyzz* d_part = (xyzz*)(w.pts + tb_bytes);
  std::vector<uint32_t> tb(nterms);
  for (uint32_t i = 0; i < n2; i++) {
    tb[i] = fb_base_g(i);
    tb[n2 + i] = fb_base_h(ng, i);
  }
  tb[2 * (size_t)n2] = fb_base_H(ng);
  BP_TRY(hipMemcpyAsync(d_tb, tb.data(), (size_t)nterms * 4, hipMemcpyHostToDevice, st));
  if ((rc = frv_ac_commit_scalars(ctx, n2, aL, aR, aO, sL, sR, bytes_as_u256(blind3), bytes_as_u256(blind3 + 32),
                                  bytes_as_u256(blind3 + 64), w.sc)) != BP_OK)
    return rc;
  if ((rc = fb_msm_launch(ctx, 3, nterms, w.sc, d_tb, d_part, nullptr, w.xout)) != BP_OK) return rc;
  if ((rc = fb_msm_fetch(ctx, w, 3, AI)) != BP_OK) return rc;
  const Fr4 aiB = fr_from_bytes(blind3), aoB = fr_from_bytes(blind3 + 32), sB = fr_from_bytes(blind3 + 64);

  uint8_t yb[32], zb[32], xb[32];
  challenge_y_z(style, AI, AO, S, yb, zb);
  const Fr4 y = fr_from_bytes(yb), z = fr_from_bytes(zb), yinv = fr_inv(y);
  // zs = tail (powerVector z (q + 1)); zwL, zwR, zwO = zs `vectorMatrixProduct` W   (Prover.hs:37-42)
  std::vector<Fr4> zp = power_vector(z, q + 1);
  std::vector<Fr4> zs(zp.begin() + 1, zp.end());
  {
    const size_t zs_bytes = (((size_t)q * 32) + 255) & ~(size_t)255, w_bytes = (size_t)q * n * 32;
    if (ctx->io.reserve(zs_bytes + w_bytes + 256) != 0) {
      ctx->err = "staging allocation failed";
      return BP_ERR_NOMEM;
    }
    u256* d_zs = (u256*)ctx->io.p;
    uint8_t* d_W = (uint8_t*)ctx->io.p + zs_bytes;
    if (q) BP_TRY(hipMemcpyAsync(d_zs, zs.data(), (size_t)q * 32, hipMemcpyHostToDevice, st));  // Fr4 == u256 layout
    const uint8_t* Ws[3] = {wL, wR, wO};
    u256* outs[3] = {zwL, zwR, zwO};
    for (int i = 0; i < 3; i++) {
      if (w_bytes) BP_TRY(hipMemcpyAsync(d_W, Ws[i], w_bytes, hipMemcpyHostToDevice, st));
      if ((rc = frv_ac_vec_mat(ctx, d_zs, d_W, q, n, n2, outs[i])) != BP_OK) return rc;
    }
  }
  // computePolynomials (Prover.hs:104-117) and tPoly = multiplyPoly (Internal.hs:130-139)
  if ((rc = frv_ac_polys(ctx, n2, aL, aR, sR, fr4_as_u256(y), fr4_as_u256(yinv), zwL, zwR, zwO, l1, r0, r1, r3)) != BP_OK)
    return rc;
  if ((rc = frv_ac_tpoly(ctx, n2, l1, aO, sL, r0, r1, r3, f.sm + 8)) != BP_OK) return rc;
  uint8_t tpoly[7][32];
  BP_TRY(hipMemcpyAsync(tpoly, f.sm + 8, sizeof(tpoly), hipMemcpyDeviceToHost, st));
  // hs' = hs_i * y^-i (:76) enters the inner-product argument as scalar factors
  if ((rc = frv_power_vector(ctx, fr4_as_u256(yinv), n2, hscale)) != BP_OK) return rc;
  BP_TRY(hipStreamSynchronize(st));  // zs, tb and the weight matrices are read: host temporaries may go
  Fr4 tc[7];
  for (int i = 0; i < 7; i++) tc[i] = fr_from_bytes(tpoly[i]);
  // tBlindings = insertAt 2 0 (0 : five randoms); tCommits = zipWith commit tPoly tBlindings   (Prover.hs:57-58)
  Fr4 tau[7] = {zero, fr_from_bytes(tau5), zero, fr_from_bytes(tau5 + 32), fr_from_bytes(tau5 + 64),
                fr_from_bytes(tau5 + 96), fr_from_bytes(tau5 + 128)};
  {
    std::vector<uint32_t> tbc{fb_base_G(ng), fb_base_H(ng)};
    std::vector<Fr4> sc7[7];
    const std::vector<Fr4>* scs[7];
    uint8_t* outs[7];
    for (int i = 0; i < 7; i++) {
      sc7[i] = {tc[i], tau[i]};
      scs[i] = &sc7[i];
      outs[i] = T + 64 * i;
    }
    if ((rc = fb_msm_host(ctx, w, tbc, 7, scs, outs)) != BP_OK) return rc;
  }
  challenge_x(style, T, xb);
  const Fr4 x = fr_from_bytes(xb), x2 = fr_sqr(x), x3 = fr_mul(x2, x);
  // ls = evaluatePolynomial n lPoly x, rs likewise (Prover.hs:63-64), on the device; t = ls . rs = tPoly(x) (:65)
  if ((rc = frv_ac_eval(ctx, n2, l1, aO, sL, r0, r1, r3, fr4_as_u256(x), fr4_as_u256(x2), fr4_as_u256(x3), f.ls, f.rs)) != BP_OK)
    return rc;
  Fr4 t = zero, xp = fr_one();
  for (int i = 0; i < 7; i++) {
    t = fr_add(t, fr_mul(tc[i], xp));
    xp = fr_mul(xp, x);
  }
  // tBlinding = sum tau_i x^i + x^2 * (zs . (commitBlinders `vectorMatrixProductT` commitmentWeights))   (:67-70)
  Fr4 zGamma = zero;
  for (uint32_t r = 0; r < q; r++) {
    Fr4 row = zero;
    for (uint32_t j = 0; j < m; j++)
      row = fr_add(row, fr_mul(fr_from_bytes(commit_blinders + (size_t)j * 32),
                               fr_from_bytes(wV + ((size_t)r * m + j) * 32)));
    zGamma = fr_add(zGamma, fr_mul(zs[r], row));
  }
  Fr4 tB = fr_mul(x2, zGamma);
  xp = fr_one();
  for (int i = 0; i < 7; i++) {
    tB = fr_add(tB, fr_mul(tau[i], xp));
    xp = fr_mul(xp, x);
  }
  const Fr4 mu = fr_add(fr_add(fr_mul(aiB, x), fr_mul(aoB, x2)), fr_mul(sB, x3));  // (:72)
  fr_to_bytes(tB, proof_scalars);
  fr_to_bytes(mu, proof_scalars + 32);
  fr_to_bytes(t, proof_scalars + 64);
  // u = gen * shamirU tBlinding mu t   (:74-75)
  uint8_t ub[32], D[64], dot0[32];
  challenge_u(style, proof_scalars, ub);
  memset(D, 0, 64);
  std::vector<uint8_t> Lv, Rv;
  rc = ipp_prove_fixed_dev(ctx, w, n2, hscale, ub, D, style, Lv, Rv, proof_scalars + 96, proof_scalars + 128, dot0);
  if (rc != BP_OK) return rc;
  if (memcmp(dot0, proof_scalars + 64, 32) != 0) {  // t = ls . rs must equal tPoly(x)
    ctx->err = "arithmetic-circuit prover: dot ls rs differs from tPoly(x)";
    return BP_ERR_STATE;
  }
  if (k) {
    memcpy(proof_points + 640, Lv.data(), (size_t)k * 64);
    memcpy(proof_points + 640 + (size_t)k * 64, Rv.data(), (size_t)k * 64);
  }
  return BP_OK;
}

}  // namespace

extern "C" {

int bp_ac_proof_points(uint32_t n_gates) {
  uint32_t n2 = pad_pow2(n_gates), k = 0;
  while ((2u << k) <= n2) k++;
  return (int)(10 + 2 * k);
}

int bp_ac_prove(bp_ctx* ctx, uint32_t n, uint32_t q, uint32_t m, const uint8_t* wL, const uint8_t* wR,
                const uint8_t* wO, const uint8_t* wV, const uint8_t* a_l, const uint8_t* a_r, const uint8_t* a_o,
                const uint8_t* commit_blinders, const uint8_t* blind3, const uint8_t* s_l, const uint8_t* s_r,
                const uint8_t* tau5, uint8_t* proof_scalars, uint8_t* proof_points) {
  if (!ctx) return BP_ERR_ARG;
  uint32_t n2, k;
  int rc = check_dims(ctx, n, q, m, n2, k);
  if (rc != BP_OK) return rc;
  BP_REQUIRE(blind3 && tau5 && proof_scalars && proof_points, "null pointer");
  BP_REQUIRE(n == 0 || (a_l && a_r && a_o && s_l && s_r), "null pointer");
  BP_REQUIRE(q == 0 || n == 0 || (wL && wR && wO), "null pointer");
  BP_REQUIRE(q == 0 || m == 0 || (wV && commit_blinders), "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  const ShowStyle style = (ShowStyle)ctx->show_style;
  ProverWs w;
  if ((rc = carve_ws(ctx, n2 ? n2 : 1, w)) != BP_OK) return rc;
  Gens g;
  if ((rc = get_gens(ctx, g)) != BP_OK) return rc;
  const Fr4 zero = fr_zero();
  if (prover_uses_fixed_base() && prover_vectors_on_device() && n2 > 0 && (size_t)q * n * 32 <= ((size_t)1 << 30))
    return ac_prove_dev(ctx, w, n, n2, k, q, m, wL, wR, wO, wV, a_l, a_r, a_o, commit_blinders, blind3, s_l, s_r, tau5,
                        proof_scalars, proof_points);

  // padAssignment (Internal.hs:91-97); sL, sR already have the padded length (chooseBlindingVectors n, :32)
  std::vector<Fr4> aL(n2, zero), aR(n2, zero), aO(n2, zero), sL(n2), sR(n2);
  for (uint32_t i = 0; i < n; i++) {
    aL[i] = fr_from_bytes(a_l + (size_t)i * 32);
    aR[i] = fr_from_bytes(a_r + (size_t)i * 32);
    aO[i] = fr_from_bytes(a_o + (size_t)i * 32);
  }
  for (uint32_t i = 0; i < n2; i++) {
    sL[i] = fr_from_bytes(s_l + (size_t)i * 32);
    sR[i] = fr_from_bytes(s_r + (size_t)i * 32);
  }
  const Fr4 aiB = fr_from_bytes(blind3), aoB = fr_from_bytes(blind3 + 32), sB = fr_from_bytes(blind3 + 64);
  uint8_t* AI = proof_points;
  uint8_t* AO = proof_points + 64;
  uint8_t* S = proof_points + 128;
  uint8_t* T = proof_points + 192;  // 7 points
  // commitBitVector (Internal.hs:102-107): aiCommit, aoCommit (vR = []), sCommit   (Prover.hs:29-33)
  if (prover_uses_fixed_base() && n2 > 0) {  // the three commitments through the fixed-base tables, one launch
    const uint32_t ng = ctx->n_gens;
    std::vector<uint32_t> tb(2 * (size_t)n2 + 1);
    for (uint32_t i = 0; i < n2; i++) {
      tb[i] = fb_base_g(i);
      tb[n2 + i] = fb_base_h(ng, i);
    }
    tb[2 * (size_t)n2] = fb_base_H(ng);
    std::vector<Fr4> scI(aL), scO(aO), scS(sL);
    scI.insert(scI.end(), aR.begin(), aR.end());
    scI.push_back(aiB);
    scO.resize(2 * (size_t)n2, zero);  // vR = []
    scO.push_back(aoB);
    scS.insert(scS.end(), sR.begin(), sR.end());
    scS.push_back(sB);
    const std::vector<Fr4>* scs[3] = {&scI, &scO, &scS};
    uint8_t* outs[3] = {AI, AO, S};
    if ((rc = fb_msm_host(ctx, w, tb, 3, scs, outs)) != BP_OK) return rc;
  } else {  // the three commitments in one launch
    MsmSeg segI[3] = {{g.d_gs, n2, true}, {g.d_hs, n2, true}, {g.d_h, 1, true}};
    MsmSeg segO[2] = {{g.d_gs, n2, true}, {g.d_h, 1, true}};
    std::vector<Fr4> scI(aL), scO(aO), scS(sL);
    scI.insert(scI.end(), aR.begin(), aR.end());
    scI.push_back(aiB);
    scO.push_back(aoB);
    scS.insert(scS.end(), sR.begin(), sR.end());
    scS.push_back(sB);
    MsmJob jobs[3] = {{segI, 3, &scI, AI}, {segO, 2, &scO, AO}, {segI, 3, &scS, S}};
    if ((rc = msm_multi(ctx, w, jobs, 3)) != BP_OK) return rc;
  }

  uint8_t yb[32], zb[32], xb[32];
  challenge_y_z(style, AI, AO, S, yb, zb);
  const Fr4 y = fr_from_bytes(yb), z = fr_from_bytes(zb);
  Derived d = derive(n, n2, q, wL, wR, wO, y, z);

  // computePolynomials (Prover.hs:104-117): l0 = 0, l1 = aL + y^-n o zwR, l2 = aO, l3 = sL;
  // r0 = zwO - y^n, r1 = y^n o aR + zwL, r2 = 0, r3 = y^n o sR.  tPoly = multiplyPoly (Internal.hs:130-139).
  std::vector<Fr4> l1(n2), r0(n2), r1(n2), r3(n2);
  for (uint32_t i = 0; i < n2; i++) {
    l1[i] = fr_add(aL[i], fr_mul(d.yinv[i], d.zwR[i]));
    r0[i] = fr_sub(d.zwO[i], d.ys[i]);
    r1[i] = fr_add(fr_mul(d.ys[i], aR[i]), d.zwL[i]);
    r3[i] = fr_mul(d.ys[i], sR[i]);
  }
  Fr4 tc[7] = {zero, zero, zero, zero, zero, zero, zero};
  for (uint32_t i = 0; i < n2; i++) {
    tc[1] = fr_add(tc[1], fr_mul(l1[i], r0[i]));
    tc[2] = fr_add(tc[2], fr_add(fr_mul(l1[i], r1[i]), fr_mul(aO[i], r0[i])));
    tc[3] = fr_add(tc[3], fr_add(fr_mul(aO[i], r1[i]), fr_mul(sL[i], r0[i])));
    tc[4] = fr_add(tc[4], fr_add(fr_mul(l1[i], r3[i]), fr_mul(sL[i], r1[i])));
    tc[5] = fr_add(tc[5], fr_mul(aO[i], r3[i]));
    tc[6] = fr_add(tc[6], fr_mul(sL[i], r3[i]));
  }
  // tBlindings = insertAt 2 0 (0 : five randoms); tCommits = zipWith commit tPoly tBlindings   (Prover.hs:57-58)
  Fr4 tau[7] = {zero, fr_from_bytes(tau5), zero, fr_from_bytes(tau5 + 32), fr_from_bytes(tau5 + 64),
                fr_from_bytes(tau5 + 96), fr_from_bytes(tau5 + 128)};
  if (prover_uses_fixed_base()) {  // seven two-term sums over gen, h through the fixed-base tables
    const uint32_t ng = ctx->n_gens;
    std::vector<uint32_t> tb{fb_base_G(ng), fb_base_H(ng)};
    std::vector<Fr4> sc7[7];
    const std::vector<Fr4>* scs[7];
    uint8_t* outs[7];
    for (int i = 0; i < 7; i++) {
      sc7[i] = {tc[i], tau[i]};
      scs[i] = &sc7[i];
      outs[i] = T + 64 * i;
    }
    if ((rc = fb_msm_host(ctx, w, tb, 7, scs, outs)) != BP_OK) return rc;
  } else {  // seven lanes of one GLV kernel: T_i = t_i * gen + tau_i * h
    uint8_t e[14 * 32];
    for (int i = 0; i < 7; i++) {
      fr_to_bytes(tc[i], e + 32 * i);
      fr_to_bytes(tau[i], e + 32 * (7 + i));
    }
    hipStream_t st = ctx->stream;
    BP_TRY(hipMemcpyAsync(w.sc, e, sizeof(e), hipMemcpyHostToDevice, st));
    LinJob jb;
    memset((void*)&jb, 0, sizeof(jb));
    jb.a = g.d_gen;
    jb.b = g.d_h;
    jb.sa = w.sc;
    jb.sb = w.sc + 7 * 32;
    jb.out = w.pts;
    jb.n = 7;
    if ((rc = ipp_lincomb_jobs(ctx, &jb, 1, w.tab)) != BP_OK) return rc;
    BP_TRY(hipMemcpyAsync(T, w.pts, 7 * 64, hipMemcpyDeviceToHost, st));
    BP_TRY(hipStreamSynchronize(st));
  }
  challenge_x(style, T, xb);
  const Fr4 x = fr_from_bytes(xb), x2 = fr_sqr(x), x3 = fr_mul(x2, x);
  // ls = evaluatePolynomial n lPoly x, rs likewise, t = ls . rs   (Prover.hs:63-65)
  std::vector<Fr4> ls(n2), rs(n2);
  Fr4 t = zero;
  for (uint32_t i = 0; i < n2; i++) {
    ls[i] = fr_add(fr_add(fr_mul(x, l1[i]), fr_mul(x2, aO[i])), fr_mul(x3, sL[i]));
    rs[i] = fr_add(fr_add(r0[i], fr_mul(x, r1[i])), fr_mul(x3, r3[i]));
    t = fr_add(t, fr_mul(ls[i], rs[i]));
  }
  // tBlinding = sum tau_i x^i + x^2 * (zs . (commitBlinders `vectorMatrixProductT` commitmentWeights))   (:67-70)
  Fr4 zGamma = zero;
  for (uint32_t r = 0; r < q; r++) {
    Fr4 row = zero;
    for (uint32_t j = 0; j < m; j++)
      row = fr_add(row, fr_mul(fr_from_bytes(commit_blinders + (size_t)j * 32),
                               fr_from_bytes(wV + ((size_t)r * m + j) * 32)));
    zGamma = fr_add(zGamma, fr_mul(d.zs[r], row));
  }
  Fr4 tB = fr_mul(x2, zGamma), xp = fr_one();
  for (int i = 0; i < 7; i++) {
    tB = fr_add(tB, fr_mul(tau[i], xp));
    xp = fr_mul(xp, x);
  }
  Fr4 mu = fr_add(fr_add(fr_mul(aiB, x), fr_mul(aoB, x2)), fr_mul(sB, x3));  // (:72)
  fr_to_bytes(tB, proof_scalars);
  fr_to_bytes(mu, proof_scalars + 32);
  fr_to_bytes(t, proof_scalars + 64);
  // u = gen * shamirU tBlinding mu t   (:74-75)
  uint8_t ub[32], u[64];
  challenge_u(style, proof_scalars, ub);
  std::vector<uint8_t> Lv, Rv;
  if (prover_uses_fixed_base()) {
    // commitmentLR (:77-85) is, for the prover's own ls, rs, the commitment <ls, gs> + <rs, hs'> + t*u (expand
    // aiCommit, aoCommit, sCommit and mu: an identity in the exponents, no circuit constraint needed), t = ls . rs:
    // offset O; hs' = hs_i * y^-i (:76) enters as scalar factors
    uint8_t D[64];
    memset(D, 0, 64);
    rc = ipp_prove_fixed(ctx, w, n2, d.yinv, ub, D, ls, rs, style, Lv, Rv, proof_scalars + 96, proof_scalars + 128);
    if (rc != BP_OK) return rc;
  } else {
    if ((rc = point_mul_gen(ctx, w, g, ub, u)) != BP_OK) return rc;
    if ((rc = load_ipp_base(ctx, w, g, d.yinv, n2)) != BP_OK) return rc;
    // the same group element as one MSM over the materialised IPP base
    uint8_t P[64];
    {
      uint8_t* d_u = w.small;
      BP_TRY(hipMemcpyAsync(d_u, u, 64, hipMemcpyHostToDevice, ctx->stream));
      if ((rc = msm3(ctx, w, w.G, ls, 0, w.H, rs, 0, n2, d_u, t, P)) != BP_OK) return rc;
    }
    rc = ipp_prove_core(ctx, w, n2, u, P, ls, rs, style, Lv, Rv, proof_scalars + 96, proof_scalars + 128);
    if (rc != BP_OK) return rc;
  }
  if (k) {
    memcpy(proof_points + 640, Lv.data(), (size_t)k * 64);
    memcpy(proof_points + 640 + (size_t)k * 64, Rv.data(), (size_t)k * 64);
  }
  return BP_OK;
}

int bp_ac_verify(bp_ctx* ctx, uint32_t n, uint32_t q, uint32_t m, const uint8_t* wL, const uint8_t* wR,
                 const uint8_t* wO, const uint8_t* wV, const uint8_t* cs, const uint8_t* v_commits,
                 const uint8_t* proof_scalars, const uint8_t* proof_points, uint8_t* verdict) {
  if (!ctx) return BP_ERR_ARG;
  uint32_t n2, k;
  int rc = check_dims(ctx, n, q, m, n2, k);
  if (rc != BP_OK) return rc;
  BP_REQUIRE(proof_scalars && proof_points && verdict, "null pointer");
  BP_REQUIRE(q == 0 || ((n == 0 || (wL && wR && wO)) && cs), "null pointer");
  BP_REQUIRE(m == 0 || (v_commits && (q == 0 || wV)), "null pointer");
  // untrusted encodings the reference's types cannot hold do not verify (validate.h; as in the batch range verifier)
  {
    bool ok = points_encoding_valid(proof_points, 10 + (size_t)2 * k) && points_encoding_valid(v_commits, m);
    for (int i = 0; i < 5; i++) ok = ok && scalar_encoding_valid(proof_scalars + 32 * i);
    if (!ok) {
      *verdict = 0;
      return BP_OK;
    }
  }
  BP_TRY(hipSetDevice(ctx->device));
  const ShowStyle style = (ShowStyle)ctx->show_style;
  ProverWs w;
  if ((rc = carve_ws(ctx, n2 ? n2 : 1, w, m)) != BP_OK) return rc;
  Gens g;
  if ((rc = get_gens(ctx, g)) != BP_OK) return rc;
  const Fr4 zero = fr_zero();
  const uint8_t* AI = proof_points;
  const uint8_t* AO = proof_points + 64;
  const uint8_t* S = proof_points + 128;
  const uint8_t* T = proof_points + 192;
  const uint8_t* Lp = proof_points + 640;
  const uint8_t* Rp = Lp + (size_t)k * 64;
  const Fr4 tB = fr_from_bytes(proof_scalars), mu = fr_from_bytes(proof_scalars + 32),
            t = fr_from_bytes(proof_scalars + 64), l = fr_from_bytes(proof_scalars + 96),
            r = fr_from_bytes(proof_scalars + 128);

  // With the fixed-base tables every remaining variable-base term is one of the proof's own points (or gen, h, the
  // commitments): their bit tables (ipp.hip) are built by one launch now, beside the host's vector algebra, and the
  // short sums and the rounds of mkChallenges then run one lane per scalar bit.
  // table order: AI, AO, S | T_1..T_7 | gen, h | V_0..V_{m-1} | L_0.. | R_0..
  const bool fixed = prover_uses_fixed_base() && n2 > 0;
  const uint32_t bt_T = 3, bt_G = 10, bt_V = 12, bt_L = 12 + m;
  const bool bits = fixed && bt_L + 2 * k <= w.bit_cap;
  if (bits) {
    hipStream_t st = ctx->stream;
    BP_TRY(hipMemcpyAsync(w.bit_pts, proof_points, (size_t)10 * 64, hipMemcpyHostToDevice, st));
    BP_TRY(hipMemcpyAsync(w.bit_pts + (size_t)bt_G * 64, g.gen_h, (size_t)2 * 64, hipMemcpyHostToDevice, st));
    if (m) BP_TRY(hipMemcpyAsync(w.bit_pts + (size_t)bt_V * 64, v_commits, (size_t)m * 64, hipMemcpyHostToDevice, st));
    if (k) BP_TRY(hipMemcpyAsync(w.bit_pts + (size_t)bt_L * 64, Lp, (size_t)2 * k * 64, hipMemcpyHostToDevice, st));
    if ((rc = ipp_bit_tables(ctx, w.bit_pts, bt_L + 2 * k, w.bit_sh, st)) != BP_OK) return rc;
  }

  // x, y, z   (Verifier.hs:28-30)
  uint8_t yb[32], zb[32], xb[32];
  challenge_x(style, T, xb);
  challenge_y_z(style, AI, AO, S, yb, zb);
  const Fr4 x = fr_from_bytes(xb), y = fr_from_bytes(yb), z = fr_from_bytes(zb);
  const Fr4 x2 = fr_sqr(x), x3 = fr_mul(x2, x);
  Derived d = derive(n, n2, q, wL, wR, wO, y, z);
  // u = gen * shamirU tBlinding mu t   (:40-41); the proof's scalars are hashed as canonical residues
  uint8_t canon3[96], ub[32], u[64];
  fr_to_bytes(tB, canon3);
  fr_to_bytes(mu, canon3 + 32);
  fr_to_bytes(t, canon3 + 64);
  challenge_u(style, canon3, ub);
  // with the fixed-base tables u = gen * ub and hs' = hs_i * y^-i never exist as points: both enter as scalar factors
  const Fr4 ubm = fr_from_bytes(ub);
  std::vector<uint32_t> tb_gh;  // gs_0.., hs_0.., h, gen
  if (fixed) {
    const uint32_t ng = ctx->n_gens;
    tb_gh.resize(2 * (size_t)n2 + 2);
    for (uint32_t i = 0; i < n2; i++) {
      tb_gh[i] = fb_base_g(i);
      tb_gh[n2 + i] = fb_base_h(ng, i);
    }
    tb_gh[2 * (size_t)n2] = fb_base_H(ng);
    tb_gh[2 * (size_t)n2 + 1] = fb_base_G(ng);
  } else if ((rc = point_mul_gen(ctx, w, g, ub, u)) != BP_OK) {
    return rc;
  }

  // verifyTPoly (:43-54): commit t tBlinding == gen*gExp + sumExps xs tCommits + sumExps vExp vCommits,
  // evaluated as one multiexp of the difference, which is O exactly when both sides are equal.
  bool t_ok;
  xyzz bit_res[2];  // latency path: x*AI + x^2*AO + x^3*S, and the difference of verifyTPoly's two sides
  {
    Fr4 kdelta = zero, cQ = zero;  // delta n y zwL zwR (Internal.hs:99-100); cQ = zs . cs
    for (uint32_t i = 0; i < n2; i++) kdelta = fr_add(kdelta, fr_mul(fr_mul(d.yinv[i], d.zwR[i]), d.zwL[i]));
    for (uint32_t rr = 0; rr < q; rr++) cQ = fr_add(cQ, fr_mul(d.zs[rr], fr_from_bytes(cs + (size_t)rr * 32)));
    Fr4 gExp = fr_mul(x2, fr_add(kdelta, cQ));
    std::vector<Fr4> sc;
    sc.push_back(fr_sub(t, gExp));
    sc.push_back(tB);
    Fr4 xp = x3;  // xs = 0 : x : 0 : x^3 .. x^6
    Fr4 xs[7] = {zero, x, zero, zero, zero, zero, zero};
    for (int i = 3; i < 7; i++) {
      xs[i] = xp;
      xp = fr_mul(xp, x);
    }
    for (int i = 0; i < 7; i++) sc.push_back(fr_neg(xs[i]));
    for (uint32_t j = 0; j < m; j++) {  // vExp = x^2 * (zs `vectorMatrixProduct` commitmentWeights)
      Fr4 col = zero;
      for (uint32_t rr = 0; rr < q; rr++)
        col = fr_add(col, fr_mul(d.zs[rr], fr_from_bytes(wV + ((size_t)rr * m + j) * 32)));
      sc.push_back(fr_neg(fr_mul(x2, col)));
    }
    if (bits) {
      // scalars in table order; the first three are those of x*AI + x^2*AO + x^3*S (verifyLRCommitment below): both
      // sums leave in one launch
      std::vector<uint8_t> sb((size_t)bt_L * 32);
      fr_to_bytes(x, &sb[0]);
      fr_to_bytes(x2, &sb[32]);
      fr_to_bytes(x3, &sb[64]);
      for (int i = 0; i < 7; i++) fr_to_bytes(sc[2 + i], &sb[(size_t)(bt_T + i) * 32]);
      fr_to_bytes(sc[0], &sb[(size_t)bt_G * 32]);
      fr_to_bytes(sc[1], &sb[(size_t)(bt_G + 1) * 32]);
      for (uint32_t j = 0; j < m; j++) fr_to_bytes(sc[9 + j], &sb[(size_t)(bt_V + j) * 32]);
      BP_TRY(hipMemcpyAsync(w.sc, sb.data(), sb.size(), hipMemcpyHostToDevice, ctx->stream));
      const uint32_t first[2] = {0, bt_T}, count[2] = {3, 9 + m};
      if ((rc = ipp_bits_multiexp(ctx, w.bit_sh, w.sc, first, count, 2, w.bit_out + 1)) != BP_OK) return rc;
      BP_TRY(hipMemcpyAsync(bit_res, w.bit_out + 1, 2 * sizeof(xyzz), hipMemcpyDeviceToHost, ctx->stream));
      BP_TRY(hipStreamSynchronize(ctx->stream));  // also releases w.sc for the sums below
      t_ok = xyzz_is_inf(bit_res[1]);
    } else {
      MsmSeg segs[3] = {{g.gen_h, 2, false}, {T, 7, false}, {v_commits, m, false}};
      uint8_t diff[64];
      if ((rc = msm_gather(ctx, w, segs, 3, sc, diff)) != BP_OK) return rc;
      t_ok = true;
      for (int i = 0; i < 64; i++) t_ok &= diff[i] == 0;
    }
  }

  // verifyLRCommitment (:56-71)
  uint8_t acc[64];
  if (fixed) {
    // commitmentLR = x*AI + x^2*AO + x^3*S + sumExps gExp gs + sumExps hExp hs' - mu*h + t*u: the generator part
    // through the tables, the three proof points as one short multiexp, one addition
    std::vector<Fr4> sc(2 * (size_t)n2 + 2);
    for (uint32_t i = 0; i < n2; i++) {
      sc[i] = fr_mul(x, fr_mul(d.yinv[i], d.zwR[i]));
      sc[n2 + i] = fr_mul(fr_sub(fr_add(fr_mul(x, d.zwL[i]), d.zwO[i]), d.ys[i]), d.yinv[i]);
    }
    sc[2 * (size_t)n2] = fr_neg(mu);
    sc[2 * (size_t)n2 + 1] = fr_mul(t, ubm);
    uint8_t fpart[64], vpart[64];
    const std::vector<Fr4>* scs[1] = {&sc};
    uint8_t* outs[1] = {fpart};
    if ((rc = fb_msm_host(ctx, w, tb_gh, 1, scs, outs)) != BP_OK) return rc;
    if (bits) {
      affine fa;
      memcpy(&fa, fpart, 64);
      affine sum = host::xyzz_to_affine_host(xyzz_madd(bit_res[0], fa));  // on the host: microseconds
      memcpy(acc, &sum, 64);
    } else {
      uint8_t three[3 * 64];
      memcpy(three, AI, 64);
      memcpy(three + 64, AO, 64);
      memcpy(three + 128, S, 64);
      std::vector<Fr4> sv{x, x2, x3};
      MsmSeg segs[1] = {{three, 3, false}};
      MsmJob vjobs[1] = {{segs, 1, &sv, vpart}};
      if ((rc = msm_multi(ctx, w, vjobs, 1)) != BP_OK) return rc;
      if ((rc = bp_point_add(ctx, fpart, vpart, 1, acc)) != BP_OK) return rc;
    }
  } else {
    if ((rc = load_ipp_base(ctx, w, g, d.yinv, n2)) != BP_OK) return rc;
    // commitmentLR = x*AI + x^2*AO + x^3*S + sumExps gExp gs + sumExps hExp hs' - mu*h + t*u
    std::vector<Fr4> sc(2 * (size_t)n2);
    for (uint32_t i = 0; i < n2; i++) {
      sc[i] = fr_mul(x, fr_mul(d.yinv[i], d.zwR[i]));
      sc[n2 + i] = fr_sub(fr_add(fr_mul(x, d.zwL[i]), d.zwO[i]), d.ys[i]);
    }
    sc.push_back(x);
    sc.push_back(x2);
    sc.push_back(x3);
    sc.push_back(fr_neg(mu));
    sc.push_back(t);
    uint8_t five[5 * 64];
    memcpy(five, AI, 64);
    memcpy(five + 64, AO, 64);
    memcpy(five + 128, S, 64);
    memcpy(five + 192, g.gen_h + 64, 64);
    memcpy(five + 256, u, 64);
    MsmSeg segs[3] = {{w.G, n2, true}, {w.H, n2, true}, {five, 5, false}};
    if ((rc = msm_gather(ctx, w, segs, 3, sc, acc)) != BP_OK) return rc;
  }
  // IPP.verifyProof n (gs, hs', u) commitmentLR productProof   (InnerProductProof/Verifier.hs:18-38)
  // mkChallenges (:40-53): x_j = shamirX' acc L_j R_j; acc <- x_j^2 L_j + x_j^-2 R_j + acc
  std::vector<Fr4> chal;  // round order; the reference's list is this reversed
  if ((rc = mk_challenges_dev(ctx, w, style, acc, Lp, Rp, k, chal,
                              bits && k ? w.bit_sh + (size_t)bt_L * GLV_LANE_BITS : nullptr)) != BP_OK)
    return rc;
  // mkOtherExponents (:55-73): s_0 = (prod x)^-1, s_{i + 2^j} = s_i * c_j^2 with c_0 the LAST round's challenge
  bool lr_ok;
  {
    std::vector<Fr4> s(n2 ? n2 : 1);
    Fr4 prod = fr_one();
    for (uint32_t j = 0; j < k; j++) prod = fr_mul(prod, chal[j]);
    s[0] = fr_inv(prod);
    for (uint32_t i1 = 1; i1 < n2; i1++) {
      uint32_t j = 31 - __builtin_clz(i1);
      s[i1] = fr_mul(s[i1 - (1u << j)], fr_sqr(chal[k - 1 - j]));
    }
    // cProof = l*<s, gs> + r*<reverse s, hs'> + (l*r)*u   (:27-36)
    uint8_t cproof[64];
    if (fixed) {
      std::vector<Fr4> sc(2 * (size_t)n2 + 2);
      for (uint32_t i = 0; i < n2; i++) {
        sc[i] = fr_mul(l, s[i]);
        sc[n2 + i] = fr_mul(fr_mul(r, s[n2 - 1 - i]), d.yinv[i]);
      }
      sc[2 * (size_t)n2] = zero;  // h
      sc[2 * (size_t)n2 + 1] = fr_mul(fr_mul(l, r), ubm);
      const std::vector<Fr4>* scs[1] = {&sc};
      uint8_t* outs[1] = {cproof};
      if ((rc = fb_msm_host(ctx, w, tb_gh, 1, scs, outs)) != BP_OK) return rc;
    } else {
      std::vector<Fr4> sc(2 * (size_t)n2 + 1);
      for (uint32_t i = 0; i < n2; i++) {
        sc[i] = fr_mul(l, s[i]);
        sc[n2 + i] = fr_mul(r, s[n2 - 1 - i]);
      }
      sc[2 * (size_t)n2] = fr_mul(l, r);
      MsmSeg segs[3] = {{w.G, n2, true}, {w.H, n2, true}, {u, 1, false}};
      if ((rc = msm_gather(ctx, w, segs, 3, sc, cproof)) != BP_OK) return rc;
    }
    lr_ok = memcmp(cproof, acc, 64) == 0;
  }
  *verdict = (lr_ok && t_ok) ? 1 : 0;
  return BP_OK;
}

}  // extern "C"
