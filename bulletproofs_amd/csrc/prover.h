// Shared pieces of the host-side prover/verifier control code (prover_host.hip, circuit_host.hip):
// the device workspace, gathered MSMs over generator slices, and the IPP recursion.
#pragma once
#include <vector>

#include "engine.h"
#include "host_fr.h"
#include "transcript.h"

namespace bp {

// Device-resident scalar-field vectors of a proof (frvec.hip): n entries of 32 B each, Montgomery form unless noted.
struct FrDev {
  u256 *ls, *rs;     // the inner-product witness, folded in place round by round
  u256 *cg, *ch;     // fold coefficients of the original generators (plain form)
  u256* v[12];       // polynomial coefficient vectors of the provers
  u256* sm;          // 64 small results (dots, polynomial coefficients, the proof's l and r)
  uint8_t* bits;     // aL of the range prover, one byte per entry
  uint8_t* raw;      // staging for byte vectors uploaded from the caller: 6 vectors of n x 32 B
  uint32_t cap;      // n
};

struct ProverWs {
  uint8_t *G, *H;    // current generator vectors (affine), n points each
  uint8_t *G2, *H2;  // fold outputs (ping-pong)
  uint8_t* pts;      // gathered MSM points: 4 segments of `seg_cap` entries
  size_t seg_cap;    // entries per segment (multiple of 256, >= 2n + 16 + extra)
  uint8_t* sc;       // MSM scalars, same layout
  uint8_t* small;    // 16 single points: u, P, L, R, ...; slots 12..15 = MSM outputs
  xyzz* tab;         // 15 * (n + 2) joint-table scratch
  // the latency path's short sums (ipp.hip, ipp_bit_tables): up to bit_cap points of a proof, affine, their bit
  // tables (16 KB a point) and a few XYZZ results
  uint32_t bit_cap;
  uint8_t* bit_pts;
  xyzz* bit_sh;
  xyzz* bit_out;  // 8 entries; entry 0 is the running commitment of mk_challenges_dev
  xyzz* xout;     // 16 entries: the sums of fb_msm_launch, XYZZ (fb_msm_fetch normalises them on the host)
  FrDev fr;            // scalar-field vectors (frvec.hip)
};

// carve the workspace for vectors of n points out of ctx->pws (grown on demand); `extra` more gathered terms
int carve_ws(bp_ctx* ctx, uint32_t n, ProverWs& w, uint32_t extra = 0);

// One slice of a gathered MSM: `count` points at `pts` (device memory if on_device, else host memory).
struct MsmSeg {
  const uint8_t* pts;
  size_t count;
  bool on_device;
};
// sum_i scalars[i] * P_i over the concatenated segments -> affine result in host memory (synchronous).
int msm_gather(bp_ctx* ctx, ProverWs& w, const MsmSeg* segs, int n_segs, const std::vector<host::Fr4>& scalars,
               uint8_t out[64]);

// Up to four gathered MSMs in one launch pair (direct path); falls back to one bucket MSM each when a job
// has more terms than msm_direct_limit().  Results land in jobs[j].out (host memory); synchronous.
struct MsmJob {
  const MsmSeg* segs;
  int nsegs;
  const std::vector<host::Fr4>* scalars;
  uint8_t* out;
};
int msm_multi(bp_ctx* ctx, ProverWs& w, const MsmJob* jobs, int njobs);

// MSM over [dA (cnt), dB (cnt), single] with scalars [sa[a0..], sb[b0..], s1]
int msm3(bp_ctx* ctx, ProverWs& w, const uint8_t* dA, const std::vector<host::Fr4>& sa, size_t a0, const uint8_t* dB,
         const std::vector<host::Fr4>& sb, size_t b0, size_t cnt, const uint8_t* d_single, const host::Fr4& s1,
         uint8_t out[64]);

// The recursion of generateProof' (InnerProductProof/Prover.hs:28-92) on device-resident w.G, w.H.
int ipp_prove_core(bp_ctx* ctx, ProverWs& w, uint32_t n, const uint8_t u[64], const uint8_t P0[64],
                   std::vector<host::Fr4>& ls, std::vector<host::Fr4>& rs, host::ShowStyle style,
                   std::vector<uint8_t>& Ls, std::vector<uint8_t>& Rs, uint8_t l_out[32], uint8_t r_out[32]);

// The same recursion when the base is the context's own generators: G = gs[0..n), H_i = hscale[i] * hs_i (the callers'
// hs'), bH = gen * ub.  Nothing is folded on the device: with c_i the product of the challenges that the folds would
// have put on generator i, every lCommit / rCommit / commitmentLR' is a multi-exponentiation over the ORIGINAL
// generators with scalars l_j c_i (fbmsm.hip, fixed-base tables, no doublings); commitmentLR' = D + <l', G'> + <r', H'>
// + <l', r'> bH with D = commitmentLR - (that sum for the initial l, r), which the folds leave unchanged
// (commitmentLR' - commitment' = commitmentLR - commitment is the identity behind Prover.hs:87-92).
// D: the caller's offset (O when commitmentLR is the commitment of ls, rs).  Same bytes as ipp_prove_core.
int ipp_prove_fixed(bp_ctx* ctx, ProverWs& w, uint32_t n, const std::vector<host::Fr4>& hscale, const uint8_t ub[32],
                    const uint8_t D[64], std::vector<host::Fr4>& ls, std::vector<host::Fr4>& rs, host::ShowStyle style,
                    std::vector<uint8_t>& Ls, std::vector<uint8_t>& Rs, uint8_t l_out[32], uint8_t r_out[32]);
// Up to 15 sums over the context's generators in one launch pair through the fixed-base tables:
// outs[j] = sum_t (*scalars[j])[t] * base(term_base[t]) (fixedbase.h numbering; scalars in Montgomery form).  Synchronous.
int fb_msm_host(bp_ctx* ctx, ProverWs& w, const std::vector<uint32_t>& term_base, uint32_t nmsm,
                const std::vector<host::Fr4>* const* scalars, uint8_t* const* outs);
// The same with the witness RESIDENT ON THE DEVICE (w.fr.ls, w.fr.rs; d_hscale: n Montgomery-form factors on the device):
// the dots cL, cR, the scalar folds ls', rs' and the expanded scalars of every round are computed by the kernels of
// frvec.hip; per round only the challenge goes down (as kernel arguments) and three points come back.  dot0_out
// (optional) receives <ls, rs> of the initial witness, canonical (the range prover's `t = dot l r` check).
int ipp_prove_fixed_dev(bp_ctx* ctx, ProverWs& w, uint32_t n, const u256* d_hscale, const uint8_t ub[32],
                        const uint8_t D[64], host::ShowStyle style, std::vector<uint8_t>& Ls, std::vector<uint8_t>& Rs,
                        uint8_t l_out[32], uint8_t r_out[32], uint8_t* dot0_out);
// mkChallenges (InnerProductProof/Verifier.hs:40-53): x_j = shamirX' acc L_j R_j; acc <- x_j^2 L_j + x_j^-2 R_j + acc for
// j < k.  acc: commitmentLR in, the final accumulator out (host bytes); Lp, Rp: k points each (host); chal: the
// challenges in ROUND order (the reference's list is this reversed).  Hashing on the host, the group work on the GPU
// (one lane per scalar bit on the bit tables of L_j, R_j, ipp.hip; the host normalises the running commitment).
// d_sh_lr: bit tables of L_0..L_{k-1}, R_0..R_{k-1} the caller has built already (ipp_bit_tables), or null.
int mk_challenges_dev(bp_ctx* ctx, ProverWs& w, host::ShowStyle style, uint8_t acc[64], const uint8_t* Lp,
                      const uint8_t* Rp, uint32_t k, std::vector<host::Fr4>& chal, const xyzz* d_sh_lr = nullptr);
// BP_PROVER_HOSTVEC=1 keeps the provers' scalar-field vector work on the host (the round-1 path; A/B timing)
bool prover_vectors_on_device();
// BP_PROVER_FOLD=1 keeps the provers on ipp_prove_core (device-side generator folds)
bool prover_uses_fixed_base();
// fbmsm.hip
// XYZZ partial sums fb_msm_launch needs at d_partial for these dimensions
size_t fb_msm_partials(bp_ctx* ctx, uint32_t nmsm, uint32_t nterms);
int fb_msm_launch(bp_ctx* ctx, uint32_t nmsm, uint32_t nterms, const uint8_t* d_scalars, const uint32_t* d_term_base,
                  xyzz* d_partial, const uint8_t* d_offset, xyzz* d_out);
// The sums of the last fb_msm_launch(..., w.xout) as affine points in host memory: one device-to-host copy of the
// nmsm XYZZ results and of `tail_bytes` more device bytes right behind them (into tail_host), a stream
// synchronisation, and the normalisation on the host (one shared inversion, host_inv.h).
int fb_msm_fetch(bp_ctx* ctx, ProverWs& w, uint32_t nmsm, uint8_t* out /*[nmsm][64]*/, void* tail_host = nullptr,
                 size_t tail_bytes = 0);

// ---- frvec.hip: launch wrappers, asynchronous on ctx->stream; all pointers are device memory ----------------------
struct IppVArgs {
  u256 c2_rev[24];  // squares of the challenges in the REVERSED round order of mkChallenges, Montgomery form
  u256 s0;          // recip (product challenges), Montgomery form
  u256 l_plain, r_plain;
};
int frv_from_bytes(bp_ctx* ctx, const uint8_t* d_bytes, uint32_t n, u256* d_out);
int frv_to_bytes(bp_ctx* ctx, const u256* d_in, uint32_t n, uint8_t* d_bytes);
int frv_power_vector(bp_ctx* ctx, const u256& a, uint32_t n, u256* d_out);
int frv_hadamard(bp_ctx* ctx, const u256* a, const u256* b, uint32_t n, u256* out);
int frv_dot(bp_ctx* ctx, const u256* a, const u256* b, uint32_t n, u256* out);
int frv_rp_commit_scalars(bp_ctx* ctx, uint32_t nm, const uint8_t* bits, const uint8_t* s_l, const uint8_t* s_r,
                          const u256& alpha, const u256& rho, uint8_t* sc);
int frv_rp_polys(bp_ctx* ctx, uint32_t n_bits, uint32_t nm, const uint8_t* bits, const u256* s_r, const u256& y,
                 const u256& z, u256* l0, u256* r0, u256* r1);
int frv_rp_tpoly(bp_ctx* ctx, uint32_t nm, const u256* l0, const u256* l1, const u256* r0, const u256* r1, u256* out3);
int frv_rp_eval(bp_ctx* ctx, uint32_t nm, const u256* l0, const u256* l1, const u256* r0, const u256* r1, const u256& x,
                u256* ls, u256* rs);
int frv_ipp_init(bp_ctx* ctx, uint32_t n, const u256* hscale, u256* cg, u256* ch);
int frv_ipp_round(bp_ctx* ctx, uint32_t n, uint32_t nk_prev, bool fold, const u256& x, const u256& xi, const u256& ub,
                  u256* ls, u256* rs, u256* cg, u256* ch, uint8_t* sc, u256* sm, bool want_scalars);
int frv_ipp_vscalars(bp_ctx* ctx, uint32_t n, uint32_t lg, const IppVArgs& a, uint8_t* sc);
int frv_ac_vec_mat(bp_ctx* ctx, const u256* zs, const uint8_t* W, uint32_t q, uint32_t n, uint32_t n2, u256* out);
int frv_ac_polys(bp_ctx* ctx, uint32_t n2, const u256* a_l, const u256* a_r, const u256* s_r, const u256& y,
                 const u256& yinv, const u256* zw_l, const u256* zw_r, const u256* zw_o, u256* l1, u256* r0, u256* r1,
                 u256* r3);
int frv_ac_tpoly(bp_ctx* ctx, uint32_t n2, const u256* l1, const u256* a_o, const u256* s_l, const u256* r0,
                 const u256* r1, const u256* r3, u256* out7);
int frv_ac_eval(bp_ctx* ctx, uint32_t n2, const u256* l1, const u256* a_o, const u256* s_l, const u256* r0,
                const u256* r1, const u256* r3, const u256& x, const u256& x2, const u256& x3, u256* ls, u256* rs);
int frv_ac_commit_scalars(bp_ctx* ctx, uint32_t n2, const u256* a_l, const u256* a_r, const u256* a_o, const u256* s_l,
                          const u256* s_r, const u256& b0, const u256& b1, const u256& b2, uint8_t* sc);

// host Fr4 (4 x u64, Montgomery form with R = 2^256) and the device's u256 (8 x u32) share their bit layout
static inline u256 fr4_as_u256(const host::Fr4& a) {
  u256 r;
  memcpy(r.v, a.v, 32);
  return r;
}
static inline u256 bytes_as_u256(const uint8_t b[32]) {
  u256 r;
  memcpy(r.v, b, 32);
  return r;
}

}  // namespace bp
