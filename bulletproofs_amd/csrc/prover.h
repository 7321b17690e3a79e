// Shared pieces of the host-side prover/verifier control code (prover_host.hip, circuit_host.hip):
// the device workspace, gathered MSMs over generator slices, and the IPP recursion.
#pragma once
#include <vector>

#include "engine.h"
#include "host_fr.h"
#include "transcript.h"

namespace bp {

struct ProverWs {
  uint8_t *G, *H;    // current generator vectors (affine), n points each
  uint8_t *G2, *H2;  // fold outputs (ping-pong)
  uint8_t* pts;      // gathered MSM points: 4 segments of `seg_cap` entries
  size_t seg_cap;    // entries per segment (multiple of 256, >= 2n + 16 + extra)
  uint8_t* sc;       // MSM scalars, same layout
  uint8_t* small;    // 16 single points: u, P, L, R, ...; slots 12..15 = MSM outputs
  xyzz* tab;         // 15 * (n + 2) joint-table scratch
  xyzz* split_tab;   // 15 x 64 joint tables of the lane-split round kernel (ipp.hip)
  xyzz* split_sh;    // shifted forms of up to 32 points (L_j, R_j of a proof), 8 each
  uint8_t* split_pts;  // those points, affine
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
// BP_PROVER_FOLD=1 keeps the provers on ipp_prove_core (device-side generator folds)
bool prover_uses_fixed_base();
// fbmsm.hip
int fb_msm_launch(bp_ctx* ctx, uint32_t nmsm, uint32_t nterms, const uint8_t* d_scalars, const uint32_t* d_term_base,
                  xyzz* d_partial, const uint8_t* d_offset, uint8_t* d_out);

}  // namespace bp
