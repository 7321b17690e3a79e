// Benchmark / test input generator: honest 64-bit-style range proofs (m = 1) for the batch
// verifier, produced at a rate the bench can afford.
//
// The proofs are exactly those Bulletproofs.RangeProof.generateProofUnsafe would emit for the
// same (v, gamma) and blinders (/root/reference/Bulletproofs/MultiRangeProof/Prover.hs:47-123):
// the reference's generators are gs_i = gen * g_i, hs_i = gen * h_i with PUBLIC g_i, h_i
// (Utils.hs:19-25), so every proof point is a*gen + b*h for scalars the prover can compute in
// Fr.  This tool does that scalar bookkeeping on the host (multi-threaded) and asks the engine
// for one batched double-multiplication per protocol step; the group elements, and therefore
// the proofs, are identical (checked against the oracle's literal prover in tests/).
// It is test/bench tooling, not part of the product path.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <thread>
#include <vector>

#include "../bulletproofs_amd/csrc/ec.h"
#include "../bulletproofs_amd/csrc/host_fr.h"
#include "../bulletproofs_amd/csrc/transcript.h"
#include "../include/bpgpu.h"

using namespace bp::host;

namespace {

struct SplitMix {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
};

template <class F>
void parallel_for(size_t n, int threads, F f) {
  if (threads < 1) threads = 1;
  std::vector<std::thread> th;
  size_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    size_t a = t * per, b = a + per < n ? a + per : n;
    if (a >= b) break;
    th.emplace_back([=] { f(a, b); });
  }
  for (auto& x : th) x.join();
}

struct ProofState {
  std::vector<Fr4> ls, rs, gv, hv;  // witness and generator discrete logs (folded in place)
  Fr4 aP;                           // discrete log of the running commitment (b-part is 0)
  Fr4 u;                            // uChallenge
};

}  // namespace

extern "C" {

// Inputs (all [batch][32] little-endian canonical unless noted): v, gamma, alpha (aBlinding), rho (sBlinding),
// tau1, tau2 (t1Blinding, t2Blinding); sL, sR: [batch][n_bits][32].  h: the reference's `h` point.
// Outputs in the bp_vbatch_load layout over the whole batch (m = 1): scalars 5 rows, points 4 + 2k + 1 rows.
int bpbench_make_proofs(bp_ctx* ctx, int show_style, uint32_t n_bits, uint32_t batch, const uint8_t* h_point,
                        const uint8_t* v, const uint8_t* gamma, const uint8_t* sL, const uint8_t* sR,
                        const uint8_t* alpha, const uint8_t* rho, const uint8_t* tau1, const uint8_t* tau2,
                        int threads, uint8_t* scalars_out, uint8_t* points_out) {
  const uint32_t n = n_bits;
  uint32_t k = 0;
  while ((1u << k) < n) k++;
  if ((1u << k) != n || n > 128) return BP_ERR_ARG;
  const size_t B = batch;
  const ShowStyle st = (ShowStyle)show_style;
  bp::affine Gaff = bp::affine_gen();
  uint8_t gen[64];
  memcpy(gen, &Gaff, 64);
  // public discrete logs of the generators (Utils.hs:20,25)
  std::vector<Fr4> gdl(n), hdl(n);
  for (uint32_t i = 0; i < n; i++) {
    uint8_t buf[MSG_SCRATCH], out[32];
    char num[16];
    int nl = snprintf(num, sizeof num, "%u", i + 1);
    Msg mg(buf, st);
    mg.raw(num, nl);
    mg.point(gen);
    mg.finish_fr(out);
    gdl[i] = fr_from_bytes(out);
    Msg mh(buf, st);
    mh.raw(num, nl);
    mh.point(h_point);
    mh.finish_fr(out);
    hdl[i] = fr_from_bytes(out);
  }
  auto row_s = [&](uint32_t r, size_t p) { return scalars_out + ((size_t)r * B + p) * 32; };
  auto row_p = [&](uint32_t r, size_t p) { return points_out + ((size_t)r * B + p) * 64; };
  const uint32_t ROW_A = 0, ROW_S = 1, ROW_T1 = 2, ROW_T2 = 3, ROW_L = 4, ROW_R = 4 + k, ROW_V = 4 + 2 * k;

  std::vector<ProofState> stt(B);
  std::vector<uint8_t> e0(3 * B * 32), e1(3 * B * 32), p0(3 * B * 64), p1(3 * B * 64), outp(3 * B * 64);
  for (size_t i = 0; i < 3 * B; i++) {
    memcpy(&p0[i * 64], gen, 64);
    memcpy(&p1[i * 64], h_point, 64);
  }
  std::vector<Fr4> yv(B), zv(B), xv(B), t1v(B), t2v(B);
  int rc;

  // ---- A, S, V  (RangeProof/Internal.hs:126-149, Utils.hs:84-85) ----
  parallel_for(B, threads, [&](size_t a, size_t b) {
    for (size_t p = a; p < b; p++) {
      uint64_t vv[4];
      memcpy(vv, v + p * 32, 32);
      Fr4 aA = fr_zero(), aS = fr_zero();
      for (uint32_t i = 0; i < n; i++) {
        bool bit = (vv[i >> 6] >> (i & 63)) & 1;
        // aL_i = bit, aR_i = bit - 1
        if (bit) aA = fr_add(aA, gdl[i]); else aA = fr_sub(aA, hdl[i]);
        Fr4 sl = fr_from_bytes(sL + (p * n + i) * 32), sr = fr_from_bytes(sR + (p * n + i) * 32);
        aS = fr_add(aS, fr_add(fr_mul(sl, gdl[i]), fr_mul(sr, hdl[i])));
      }
      fr_to_bytes(aA, &e0[(0 * B + p) * 32]);
      memcpy(&e1[(0 * B + p) * 32], alpha + p * 32, 32);
      fr_to_bytes(aS, &e0[(1 * B + p) * 32]);
      memcpy(&e1[(1 * B + p) * 32], rho + p * 32, 32);
      memcpy(&e0[(2 * B + p) * 32], v + p * 32, 32);
      memcpy(&e1[(2 * B + p) * 32], gamma + p * 32, 32);
    }
  });
  if ((rc = bp_add_two_mul(ctx, e0.data(), p0.data(), e1.data(), p1.data(), 3 * B, outp.data())) != BP_OK) return rc;
  for (size_t p = 0; p < B; p++) {
    memcpy(row_p(ROW_A, p), &outp[(0 * B + p) * 64], 64);
    memcpy(row_p(ROW_S, p), &outp[(1 * B + p) * 64], 64);
    memcpy(row_p(ROW_V, p), &outp[(2 * B + p) * 64], 64);
  }

  // ---- y, z, polynomials, T1, T2  (MultiRangeProof/Prover.hs:74-85,133-169) ----
  parallel_for(B, threads, [&](size_t a, size_t b) {
    for (size_t p = a; p < b; p++) {
      uint8_t buf[MSG_SCRATCH], yb[32], zb[32];
      Msg my(buf, st);
      my.r(); my.point(row_p(ROW_A, p)); my.point(row_p(ROW_S, p));
      size_t pre = my.len;
      uint8_t keep[MSG_SCRATCH];
      memcpy(keep, buf, pre);
      my.finish_fr(yb);
      Msg mz(buf, st);
      mz.raw(keep, pre); mz.fr(yb);
      mz.finish_fr(zb);
      Fr4 y = fr_from_bytes(yb), z = fr_from_bytes(zb), z2 = fr_sqr(z);
      yv[p] = y; zv[p] = z;
      uint64_t vv[4];
      memcpy(vv, v + p * 32, 32);
      ProofState& s = stt[p];
      s.ls.resize(n); s.rs.resize(n); s.gv.resize(n); s.hv.resize(n);
      // l0 = aL - z, l1 = sL, r0 = y^i (aR + z) + z^2 2^i, r1 = y^i sR; stash l0,r0 in ls,rs and l1,r1 in gv,hv
      Fr4 yi = fr_one(), two_i = fr_one(), one = fr_one(), two = fr_add(one, one);
      Fr4 t0 = fr_zero(), t1 = fr_zero(), t2 = fr_zero();
      for (uint32_t i = 0; i < n; i++) {
        bool bit = (vv[i >> 6] >> (i & 63)) & 1;
        Fr4 aL = bit ? one : fr_zero(), aR = bit ? fr_zero() : fr_neg(one);
        Fr4 l0 = fr_sub(aL, z);
        Fr4 r0 = fr_add(fr_mul(yi, fr_add(aR, z)), fr_mul(z2, two_i));
        Fr4 l1 = fr_from_bytes(sL + (p * n + i) * 32);
        Fr4 r1 = fr_mul(yi, fr_from_bytes(sR + (p * n + i) * 32));
        t0 = fr_add(t0, fr_mul(l0, r0));
        t2 = fr_add(t2, fr_mul(l1, r1));
        t1 = fr_add(t1, fr_add(fr_mul(l1, r0), fr_mul(l0, r1)));
        s.ls[i] = l0; s.rs[i] = r0; s.gv[i] = l1; s.hv[i] = r1;
        yi = fr_mul(yi, y);
        two_i = fr_mul(two_i, two);
      }
      t1v[p] = t1; t2v[p] = t2;
      s.aP = t0;  // parked until x is known
      fr_to_bytes(t1, &e0[(0 * B + p) * 32]);
      memcpy(&e1[(0 * B + p) * 32], tau1 + p * 32, 32);
      fr_to_bytes(t2, &e0[(1 * B + p) * 32]);
      memcpy(&e1[(1 * B + p) * 32], tau2 + p * 32, 32);
    }
  });
  if ((rc = bp_add_two_mul(ctx, e0.data(), p0.data(), e1.data(), p1.data(), 2 * B, outp.data())) != BP_OK) return rc;
  for (size_t p = 0; p < B; p++) {
    memcpy(row_p(ROW_T1, p), &outp[(0 * B + p) * 64], 64);
    memcpy(row_p(ROW_T2, p), &outp[(1 * B + p) * 64], 64);
  }

  // ---- x, ls, rs, t, tBlinding, mu, u, hs', P  (Prover.hs:87-112) ----
  parallel_for(B, threads, [&](size_t a, size_t b) {
    for (size_t p = a; p < b; p++) {
      uint8_t buf[MSG_SCRATCH], yb[32], zb[32], xb[32], ub[32];
      fr_to_bytes(yv[p], yb);
      fr_to_bytes(zv[p], zb);
      Msg mx(buf, st);
      mx.r(); mx.point(row_p(ROW_A, p)); mx.point(row_p(ROW_S, p)); mx.point(row_p(ROW_T1, p));
      mx.point(row_p(ROW_T2, p)); mx.fr(yb); mx.fr(zb);
      mx.finish_fr(xb);
      Fr4 x = fr_from_bytes(xb), x2 = fr_sqr(x), z = zv[p], z2 = fr_sqr(z), y = yv[p];
      xv[p] = x;
      ProofState& s = stt[p];
      Fr4 t = fr_add(fr_add(s.aP, fr_mul(t1v[p], x)), fr_mul(t2v[p], x2));
      Fr4 tb = fr_add(fr_add(fr_mul(z2, fr_from_bytes(gamma + p * 32)), fr_mul(fr_from_bytes(tau2 + p * 32), x2)),
                      fr_mul(fr_from_bytes(tau1 + p * 32), x));
      Fr4 mu = fr_add(fr_from_bytes(alpha + p * 32), fr_mul(fr_from_bytes(rho + p * 32), x));
      fr_to_bytes(tb, row_s(0, p));
      fr_to_bytes(mu, row_s(1, p));
      fr_to_bytes(t, row_s(2, p));
      Msg mu_(buf, st);
      mu_.r(); mu_.fr(row_s(0, p)); mu_.fr(row_s(1, p)); mu_.fr(row_s(2, p));
      mu_.finish_fr(ub);
      s.u = fr_from_bytes(ub);
      Fr4 yinv = fr_inv(y), yi = fr_one();
      Fr4 aP = fr_mul(t, s.u);
      for (uint32_t i = 0; i < n; i++) {
        Fr4 l = fr_add(s.ls[i], fr_mul(x, s.gv[i]));
        Fr4 r = fr_add(s.rs[i], fr_mul(x, s.hv[i]));
        s.ls[i] = l; s.rs[i] = r;
        s.gv[i] = gdl[i];
        s.hv[i] = fr_mul(hdl[i], yi);  // hs'_i = hs_i * y^-i
        yi = fr_mul(yi, yinv);
        aP = fr_add(aP, fr_add(fr_mul(l, s.gv[i]), fr_mul(r, s.hv[i])));
      }
      s.aP = aP;  // P = <ls,G> + <rs,hs'> + t*u  (= computeLRCommitment for an honest prover)
    }
  });

  // ---- inner-product argument  (InnerProductProof/Prover.hs:28-92) ----
  uint32_t len = n;
  for (uint32_t r = 0; r < k; r++) {
    uint32_t half = len / 2;
    parallel_for(B, threads, [&](size_t a, size_t b) {
      for (size_t p = a; p < b; p++) {
        ProofState& s = stt[p];
        Fr4 cL = fr_zero(), cR = fr_zero(), aL = fr_zero(), aR = fr_zero();
        for (uint32_t i = 0; i < half; i++) {
          cL = fr_add(cL, fr_mul(s.ls[i], s.rs[half + i]));
          cR = fr_add(cR, fr_mul(s.ls[half + i], s.rs[i]));
          aL = fr_add(aL, fr_add(fr_mul(s.ls[i], s.gv[half + i]), fr_mul(s.rs[half + i], s.hv[i])));
          aR = fr_add(aR, fr_add(fr_mul(s.ls[half + i], s.gv[i]), fr_mul(s.rs[i], s.hv[half + i])));
        }
        aL = fr_add(aL, fr_mul(cL, s.u));
        aR = fr_add(aR, fr_mul(cR, s.u));
        fr_to_bytes(aL, &e0[(0 * B + p) * 32]);
        fr_to_bytes(aR, &e0[(1 * B + p) * 32]);
        fr_to_bytes(s.aP, &e0[(2 * B + p) * 32]);
        // stash for the fold
        s.ls.push_back(aL);
        s.ls.push_back(aR);
      }
    });
    if ((rc = bp_point_mul(ctx, e0.data(), p0.data(), 3 * B, outp.data())) != BP_OK) return rc;
    parallel_for(B, threads, [&](size_t a, size_t b) {
      for (size_t p = a; p < b; p++) {
        ProofState& s = stt[p];
        memcpy(row_p(ROW_L + r, p), &outp[(0 * B + p) * 64], 64);
        memcpy(row_p(ROW_R + r, p), &outp[(1 * B + p) * 64], 64);
        uint8_t buf[MSG_SCRATCH], xb[32];
        Msg mm(buf, st);
        mm.r(); mm.point(&outp[(0 * B + p) * 64]); mm.point(&outp[(1 * B + p) * 64]); mm.point(&outp[(2 * B + p) * 64]);
        mm.finish_fr(xb);
        Fr4 x = fr_from_bytes(xb), xi = fr_inv(x);
        Fr4 aR = s.ls.back(); s.ls.pop_back();
        Fr4 aL = s.ls.back(); s.ls.pop_back();
        for (uint32_t i = 0; i < half; i++) {
          Fr4 g2 = fr_add(fr_mul(xi, s.gv[i]), fr_mul(x, s.gv[half + i]));
          Fr4 h2 = fr_add(fr_mul(x, s.hv[i]), fr_mul(xi, s.hv[half + i]));
          Fr4 l2 = fr_add(fr_mul(x, s.ls[i]), fr_mul(xi, s.ls[half + i]));
          Fr4 r2 = fr_add(fr_mul(xi, s.rs[i]), fr_mul(x, s.rs[half + i]));
          s.gv[i] = g2; s.hv[i] = h2; s.ls[i] = l2; s.rs[i] = r2;
        }
        s.ls.resize(half); s.rs.resize(half); s.gv.resize(half); s.hv.resize(half);
        s.aP = fr_add(s.aP, fr_add(fr_mul(fr_sqr(x), aL), fr_mul(fr_sqr(xi), aR)));
      }
    });
    len = half;
  }
  for (size_t p = 0; p < B; p++) {
    fr_to_bytes(stt[p].ls[0], row_s(3, p));
    fr_to_bytes(stt[p].rs[0], row_s(4, p));
  }
  return BP_OK;
}

// Deterministic inputs for bpbench_make_proofs from a seed (a documented stand-in for the
// reference's MonadRandom draws): v uniform in [0, 2^n), blinders in [1, r),
// sL/sR in [1, 2^min(n,64)).  Buffers as in bpbench_make_proofs.
void bpbench_random_inputs(uint64_t seed, uint32_t n_bits, uint32_t batch, uint8_t* v, uint8_t* gamma, uint8_t* sL,
                           uint8_t* sR, uint8_t* alpha, uint8_t* rho, uint8_t* tau1, uint8_t* tau2) {
  SplitMix rng{seed};
  auto scalar = [&](uint8_t* out) {
    uint64_t w[4];
    do {
      for (int i = 0; i < 4; i++) w[i] = rng.next();
      w[3] &= 0x7FFFFFFFFFFFFFFFULL;  // < 2^255 < r
    } while ((w[0] | w[1] | w[2] | w[3]) == 0);
    memcpy(out, w, 32);
  };
  auto small = [&](uint8_t* out, uint32_t bits, bool nonzero) {
    uint64_t w[4] = {0, 0, 0, 0};
    do {
      for (uint32_t i = 0; i < (bits + 63) / 64 && i < 4; i++) w[i] = rng.next();
      if (bits < 256 && (bits & 63)) w[bits >> 6] &= (1ULL << (bits & 63)) - 1;
    } while (nonzero && (w[0] | w[1] | w[2] | w[3]) == 0);
    memcpy(out, w, 32);
  };
  for (size_t p = 0; p < batch; p++) {
    small(v + p * 32, n_bits, false);
    scalar(gamma + p * 32);
    for (uint32_t i = 0; i < n_bits; i++) small(sL + (p * n_bits + i) * 32, n_bits < 64 ? n_bits : 64, true);
    for (uint32_t i = 0; i < n_bits; i++) small(sR + (p * n_bits + i) * 32, n_bits < 64 ? n_bits : 64, true);
    scalar(alpha + p * 32);
    scalar(rho + p * 32);
    scalar(tau1 + p * 32);
    scalar(tau2 + p * 32);
  }
}

}  // extern "C"
