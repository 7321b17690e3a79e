// bp_rangeproof_verify: MultiRangeProof.verifyProof for ONE proof, total over its arguments as the reference's
// function is (/root/reference/Bulletproofs/MultiRangeProof/Verifier.hs:19-88).  The reference checks nothing about
// `upperBound` or about the number of L/R pairs in the proof on the verifier side:
//   n = logBase2 upperBound (Double arithmetic, Utils.hs:93-94) may be any integer, not only a power of two <= 128;
//   mkChallenges folds over however many (L, R) pairs the proof holds (InnerProductProof/Verifier.hs:44-53);
//   mkOtherExponents walks a Map for any n (:55-73) and `sumExps` truncates to the shorter list (Utils.hs:79);
//   `challenges !! j` fails when the proof has fewer pairs than logBase2 (n m) (:72).
// The batch engine (verify.hip) covers the shape every honest proof has - n a power of two <= 128 and exactly
// log2(n m) pairs - and this entry point routes such proofs to it.  Every other input takes the general path below,
// which follows the reference's definitions literally (hs' materialised as points, one sumExps per definition) on the
// same GPU primitives: bp_msm (sumExps), bp_point_mul (mul), bp_ipp_verify (IPP.verifyProof for any n and k).  This is
// the path of the reference's negative test prop_differentUpperBound (test/Test/Protocol/RangeProof.hs:173-181).
#include <stdlib.h>
#include <string>
#include <vector>

#include "engine.h"
#include "host_fr.h"
#include "transcript.h"
#include "validate.h"

using namespace bp;
using namespace bp::host;

namespace {

// `powerVector a n` (Utils.hs:50-52), including the 0^0 = 0 special case (every entry of powerVector 0 n is 0)
std::vector<Fr4> power_vector(const Fr4& a, uint64_t n) {
  std::vector<Fr4> v((size_t)n);
  if (fr_is_zero(a)) {
    for (auto& e : v) e = fr_zero();
    return v;
  }
  Fr4 cur = fr_one();
  for (uint64_t i = 0; i < n; i++) {
    v[(size_t)i] = cur;
    cur = fr_mul(cur, a);
  }
  return v;
}

// a ^ e for e >= 1 ((^) of Prelude: 0 ^ e = 0)
Fr4 fr_pow(const Fr4& a, uint64_t e) {
  Fr4 r = fr_one(), b = a;
  while (e) {
    if (e & 1) r = fr_mul(r, b);
    b = fr_sqr(b);
    e >>= 1;
  }
  return r;
}

Fr4 fr_sum(const std::vector<Fr4>& v) {
  Fr4 s = fr_zero();
  for (auto& e : v) s = fr_add(s, e);
  return s;
}

// log2Ceil (Utils.hs:124-130) for x >= 1
uint32_t log2_ceil(uint32_t x) {
  const uint32_t floor_log = 31 - (uint32_t)__builtin_clz(x);
  return floor_log + ((uint32_t)__builtin_ctz(x) < floor_log ? 1 : 0);
}

}  // namespace

extern "C" {

int bp_rangeproof_verify(bp_ctx* ctx, const uint8_t* upper_bound, size_t ub_len, const uint8_t* v_commits, uint32_t m,
                         const uint8_t* proof_scalars, const uint8_t* proof_points, uint32_t k, uint8_t* verdict) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(upper_bound && proof_scalars && proof_points && verdict && (m == 0 || v_commits), "null pointer");
  BP_REQUIRE(k <= 64, "more than 64 L/R pairs");
  BP_REQUIRE(ub_len <= 4096, "upper bound longer than 4096 bytes");
  if (m == 0) {
    // log2Ceil 0 = -1 and `2 ^ (-1)` raises "Negative exponent" as soon as vCommitsExp2 is measured (Verifier.hs:37-39)
    ctx->err = "bp_rangeproof_verify: no commitments (the reference fails with `Negative exponent` here)";
    return BP_ERR_REFERENCE_FAILS;
  }
  BP_REQUIRE(m <= 8192, "more than 8192 commitments");
  // n = logBase2 upperBound.  For upperBound = 0 GHC's floor (-Infinity) is -2^1024: every list the verifier builds from
  // a negative length is empty and every `take` / `drop` count wraps to 0, exactly as for n = 0.
  const int lb = bp_log_base2(upper_bound, ub_len);
  const uint64_t n = lb < 0 ? 0 : (uint64_t)lb;
  const uint32_t m2 = 1u << log2_ceil(m);  // vCommits ++ replicate (2 ^ log2Ceil m - m) mempty   (Verifier.hs:37-38)
  const uint64_t nm = n * m2;
  if (ctx->n_gens == 0) {
    ctx->err = "generators not set (bp_generators_derive / bp_ctx_set_generators)";
    return BP_ERR_STATE;
  }
  // `gs` and `hs` are infinite lists in the reference; the context holds the first n_gens of each.  mkOtherExponents
  // always holds key 0, so even n m = 0 reads gs !! 0.
  const uint64_t need = nm ? nm : 1;
  if (need > ctx->n_gens) {
    ctx->err = "bp_rangeproof_verify: logBase2(upperBound) * m exceeds the context's generators (derive " +
               std::to_string((unsigned long long)need) + " or more)";
    return BP_ERR_STATE;
  }
  const uint8_t *A = proof_points, *S = proof_points + 64, *T1 = proof_points + 128, *T2 = proof_points + 192;
  const uint8_t* Lp = proof_points + 256;
  const uint8_t* Rp = Lp + (size_t)k * 64;

  // ---- the shape the batch engine covers: one call of it, batch = 1 ------------------------------------------------
  {
    uint32_t lg = 0;
    while (((uint64_t)1 << lg) < nm) lg++;
    const bool pow2 = n >= 1 && n <= 128 && (n & (n - 1)) == 0;
    // BP_VERIFY_GENERAL=1 sends every proof down the general path (the tests compare the two paths with it)
    const char* force = getenv("BP_VERIFY_GENERAL");
    if (pow2 && k == lg && !(force && atoi(force) != 0)) {
      std::vector<uint8_t> pts((size_t)(4 + 2 * k + m2) * 64, 0);
      memcpy(pts.data(), proof_points, (size_t)(4 + 2 * k) * 64);
      memcpy(pts.data() + (size_t)(4 + 2 * k) * 64, v_commits, (size_t)m * 64);  // the residue stays O
      return bp_rangeproof_verify_batch(ctx, (uint32_t)n, m2, 1, proof_scalars, pts.data(), verdict);
    }
  }

  // ---- the general path --------------------------------------------------------------------------------------------
  // Encodings the reference's types cannot hold do not verify (validate.h), as in the batch engine (k_vcheck).
  {
    bool ok = points_encoding_valid(proof_points, 4 + (size_t)2 * k) && points_encoding_valid(v_commits, m);
    for (int i = 0; i < 5; i++) ok = ok && scalar_encoding_valid(proof_scalars + 32 * i);
    if (!ok) {
      *verdict = 0;
      return BP_OK;
    }
  }
  BP_TRY(hipSetDevice(ctx->device));
  const ShowStyle style = (ShowStyle)ctx->show_style;
  int rc;
  // y, z, x   (Verifier.hs:31-33; Utils.hs:145-165) and uChallenge = shamirU tBlinding mu t   (Verifier.hs:86)
  uint8_t yb[32], zb[32], xb[32], ub[32];
  {
    uint8_t buf[MSG_SCRATCH];
    Msg my(buf, style);
    my.r();
    my.point(A);
    my.point(S);
    my.finish_fr(yb);
    Msg mz(buf, style);
    mz.r();
    mz.point(A);
    mz.point(S);
    mz.fr(yb);
    mz.finish_fr(zb);
    Msg mx(buf, style);
    mx.r();
    mx.point(A);
    mx.point(S);
    mx.point(T1);
    mx.point(T2);
    mx.fr(yb);
    mx.fr(zb);
    mx.finish_fr(xb);
    Msg mu(buf, style);
    mu.r();
    mu.fr(proof_scalars);
    mu.fr(proof_scalars + 32);
    mu.fr(proof_scalars + 64);
    mu.finish_fr(ub);
  }
  const Fr4 y = fr_from_bytes(yb), z = fr_from_bytes(zb), x = fr_from_bytes(xb);
  const Fr4 tB = fr_from_bytes(proof_scalars), mu = fr_from_bytes(proof_scalars + 32),
            t = fr_from_bytes(proof_scalars + 64);
  const Fr4 z2 = fr_sqr(z), x2 = fr_sqr(x);
  // generators: gen, h and the needed prefixes of gs, hs from the context's device-resident lists
  const uint32_t ng = ctx->n_gens;
  std::vector<uint8_t> gsb((size_t)need * 64), hsb((size_t)(nm ? nm : 1) * 64);
  uint8_t gen_h[128];
  BP_TRY(hipMemcpyAsync(gsb.data(), ctx->d_bases, (size_t)need * 64, hipMemcpyDeviceToHost, ctx->stream));
  if (nm)
    BP_TRY(hipMemcpyAsync(hsb.data(), ctx->d_bases + (size_t)ng * 64, (size_t)nm * 64, hipMemcpyDeviceToHost, ctx->stream));
  BP_TRY(hipMemcpyAsync(gen_h, ctx->d_bases + (size_t)2 * ng * 64, 128, hipMemcpyDeviceToHost, ctx->stream));
  BP_TRY(hipStreamSynchronize(ctx->stream));

  // 2^n mod r and dot (powerVector 1 n) (powerVector 2 n) = 2^n - 1
  const Fr4 two = fr_from_u64(2);
  const std::vector<Fr4> pow2n = power_vector(two, n);
  const Fr4 sum2n = fr_sum(pow2n);
  const std::vector<Fr4> ynm = power_vector(y, nm);

  // ---- verifyTPoly (Verifier.hs:44-65): commit t tBlinding == sumExps (z^2 * powerVector z m) vCommits <> gen * delta
  //      <> t1Commit * x <> t2Commit * x^2, as ONE sumExps of the difference, which is O exactly when the sides agree
  {
    // delta n m y z (RangeProof/Internal.hs:152-157)
    Fr4 delta = fr_mul(fr_sub(z, z2), fr_sum(ynm));
    for (uint32_t j = 1; j <= m2; j++) delta = fr_sub(delta, fr_mul(fr_pow(z, (uint64_t)j + 2), sum2n));
    const std::vector<Fr4> zm = power_vector(z, m2);
    const size_t nt = (size_t)m2 + 4;
    std::vector<uint8_t> sc(nt * 32), pts(nt * 64, 0);
    memcpy(&pts[0], gen_h, 128);  // gen, h
    fr_to_bytes(fr_sub(t, delta), &sc[0]);
    fr_to_bytes(tB, &sc[32]);
    for (uint32_t j = 0; j < m2; j++) {
      if (j < m) memcpy(&pts[(size_t)(2 + j) * 64], v_commits + (size_t)j * 64, 64);
      fr_to_bytes(fr_neg(fr_mul(z2, zm[j])), &sc[(size_t)(2 + j) * 32]);
    }
    memcpy(&pts[(size_t)(2 + m2) * 64], T1, 64);
    memcpy(&pts[(size_t)(3 + m2) * 64], T2, 64);
    fr_to_bytes(fr_neg(x), &sc[(size_t)(2 + m2) * 32]);
    fr_to_bytes(fr_neg(x2), &sc[(size_t)(3 + m2) * 32]);
    uint8_t diff[64];
    if ((rc = bp_msm(ctx, sc.data(), pts.data(), nt, diff)) != BP_OK) return rc;
    bool t_ok = true;
    for (int i = 0; i < 64; i++) t_ok = t_ok && diff[i] == 0;
    if (!t_ok) {  // `and` stops at the first False: verifyLRCommitment is never evaluated (Verifier.hs:26-29)
      *verdict = 0;
      return BP_OK;
    }
  }

  // ---- verifyLRCommitment (Verifier.hs:68-88) ----------------------------------------------------------------------
  // hs' = zipWith (\yi hi -> hi `mul` recip yi) (powerVector y nm) hs   (:85)
  std::vector<uint8_t> hsp((size_t)(nm ? nm : 1) * 64);
  if (nm) {
    if (fr_is_zero(y)) {  // powerVector 0 nm is all zeros and `recip 0` divides by zero
      ctx->err = "bp_rangeproof_verify: y = 0 (the reference fails with a division by zero here)";
      return BP_ERR_REFERENCE_FAILS;
    }
    const std::vector<Fr4> yinv = power_vector(fr_inv(y), nm);
    std::vector<uint8_t> sc((size_t)nm * 32);
    for (uint64_t i = 0; i < nm; i++) fr_to_bytes(yinv[(size_t)i], &sc[(size_t)i * 32]);
    if ((rc = bp_point_mul(ctx, sc.data(), hsb.data(), (size_t)nm, hsp.data())) != BP_OK) return rc;
  }
  // u = gen `mul` uChallenge   (:86-87)
  uint8_t u[64];
  if ((rc = bp_point_mul(ctx, ub, gen_h, 1, u)) != BP_OK) return rc;
  // computeLRCommitment (RangeProof/Internal.hs:169-206):
  //   aCommit <> sCommit*x <> inv (gsSum*z) <> sumExps hExp hs' <> foldl' (sumExps (hExp' j) (slice n j hs')) [1..m]
  //   <> inv (h*mu) <> u*t
  uint8_t P[64];
  {
    const size_t nt = (size_t)2 * nm + 4;
    std::vector<uint8_t> sc(nt * 32), pts(nt * 64);
    memcpy(&pts[0], A, 64);
    memcpy(&pts[64], S, 64);
    fr_to_bytes(fr_one(), &sc[0]);
    fr_to_bytes(x, &sc[32]);
    const Fr4 negz = fr_neg(z);
    if (nm) {
      memcpy(&pts[128], gsb.data(), (size_t)nm * 64);
      memcpy(&pts[128 + (size_t)nm * 64], hsp.data(), (size_t)nm * 64);
    }
    Fr4 zj1 = z;  // z ^ (j + 1), j from 1
    for (uint32_t j = 1; j <= m2; j++) {
      zj1 = fr_mul(zj1, z);
      for (uint64_t i = 0; i < n; i++) {
        const uint64_t g = (uint64_t)(j - 1) * n + i;
        fr_to_bytes(negz, &sc[(size_t)(2 + g) * 32]);
        // hExp !! g = z * y^g, hExp' j !! i = z^(j+1) * 2^i on the same hs' !! g
        fr_to_bytes(fr_add(fr_mul(z, ynm[(size_t)g]), fr_mul(zj1, pow2n[(size_t)i])), &sc[(size_t)(2 + nm + g) * 32]);
      }
    }
    memcpy(&pts[(size_t)(2 + 2 * nm) * 64], gen_h + 64, 64);  // h
    memcpy(&pts[(size_t)(3 + 2 * nm) * 64], u, 64);
    fr_to_bytes(fr_neg(mu), &sc[(size_t)(2 + 2 * nm) * 32]);
    fr_to_bytes(t, &sc[(size_t)(3 + 2 * nm) * 32]);
    if ((rc = bp_msm(ctx, sc.data(), pts.data(), nt, P)) != BP_OK) return rc;
  }
  // IPP.verifyProof nm (InnerProductBase gs hs' u) commitmentLR productProof   (:78-82)
  return bp_ipp_verify(ctx, nm, gsb.data(), (uint32_t)need, hsp.data(), (uint32_t)nm, u, P, Lp, Rp, k, proof_scalars + 96,
                       proof_scalars + 128, verdict);
}

}  // extern "C"
