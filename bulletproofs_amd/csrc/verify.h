// Shared definitions of the batch verifier (verify.hip, host/rangeproof_host.cpp).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <functional>
#include "engine.h"
#include "fixedbase.h"

namespace bp {

struct VDims {
  uint32_t B;   // proofs in the batch
  uint32_t n;   // bit width (upper bound 2^n)
  uint32_t m;   // aggregated values (power of two; commitments padded with O)
  uint32_t nm;  // vector length
  uint32_t k;   // IPP rounds = log2(nm)
  uint32_t wb;  // window width of the context's fixed-base table (fixedbase.h)
};

// rows of the per-batch point array pts[row][proof][64]
enum { VP_A = 0, VP_S = 1, VP_T1 = 2, VP_T2 = 3, VP_L0 = 4 };
// then L_0..L_{k-1}, R_0..R_{k-1}, V_0..V_{m-1}
// rows of the per-batch scalar array sc[row][proof][32]: tBlinding, mu, t, l, r
enum { VS_TAU = 0, VS_MU = 1, VS_T = 2, VS_L = 3, VS_R = 4 };

// lanes cooperating on one proof in the fixed-base kernel: enough threads to fill the chip
inline uint32_t vbatch_lpp(uint32_t B) {
  static const uint64_t target = [] {
    const char* e = getenv("BP_FIXED_LANES");
    uint64_t v = e ? strtoull(e, nullptr, 10) : 0;
    return v ? v : (uint64_t)131072;
  }();
  uint32_t lpp = 1;
  while (lpp < 64 && (uint64_t)B * lpp < target) lpp <<= 1;
  return lpp;
}

}  // namespace bp

// One batch of proofs of a single shape, in lock-step (include/bpgpu.h bp_vbatch_*).
// Owns its stream and pinned staging so that two batches can ping-pong: the host
// hashes one while the GPU works on the other.
struct bp_vbatch {
  // host loops over the proofs of the latency path (inversions between the hashes): the job's hashing threads when the
  // batch belongs to a bp_vjob, the calling thread otherwise
  std::function<void(size_t, const std::function<void(size_t, size_t)>&)> pfor;
  bp_ctx* ctx;
  bp::VDims d;
  bp::DevBuf buf;
  hipStream_t stream;
  hipStream_t aux;        // side stream of the t-polynomial check (verifyTPoly), which needs y, z, x only
  bool t_pending;         // that check is in flight on `aux`
  bool tab_pending;       // the rounds' joint tables (k_round_tables) are in flight on `aux`
  bool begun_complete;    // the begin stage has been waited for (its constants are complete)
  // device
  uint8_t *sc, *pts, *chal, *xk, *acc_aff, *vs, *verdict;
  uint8_t* bad;           // per proof: the loaded encoding was invalid (k_vcheck); such a proof does not verify
  uint64_t epoch;         // ctx->gen_epoch at creation: the generator set the term tables refer to
  bp::u256 *cst, *ypowinv, *zpow, *xk_m /* x_k^2 */, *xkinv_m, *stab /* half-tables of the s-vector */;
  uint32_t *dig1, *dig3a, *dig3b, *tb1, *tb3a, *tb3b;
  bp::xyzz *acc1, *acc2, *vtab;  // vtab: S and phi(S) per proof (k_stage1_tail_pair)
  uint32_t *fsmsg, *fsmsg2, *fspre;  // device transcript (fsdev.h): message buffers, and the rounds' hashed openings
  // joint tables of the rounds, of S and of the t-check's terms (k_round_tables): affine entries, and the XYZZ
  // composites / running products they are normalised from
  uint8_t* rtab;
  uint32_t* rtx;   // k_round_tables' composite entries on 29-bit limbs, [11 T][36][B]
  uint32_t* rpre;  // ... and the running products of their ZZZ, [11 T][9][B]
  // small batches (B <= VSPLIT_MAX_B): latency path.  Every variable-base product runs one lane per scalar bit on the
  // bit tables sh[point][128] = 2^j * point of the proof's points (rows VP_S..), the running commitment stays in
  // XYZZ on the device (acc_x) and is normalised by the host from the pinned copy hx, and the host supplies the
  // field inversions (hinv: y^-1, (y-1)^-1, x_r^-1 per proof, pinned) it can do in microseconds.
  bool split;
  bp::xyzz *sh, *acc_x, *fpart;  // fpart: workgroup sums of the flat fixed-base kernel
  bp::xyzz* hx;                  // pinned [B]
  uint8_t* hinv;                 // pinned [3][B][32], Montgomery form
  bool collect_pending;          // hx holds a commitment the caller has not been given yet (vbatch_collect)
  uint32_t nt1, nt3a, nt3b, npts;
  int stage;  // 0 created, 1 loaded, 2 begun, 2+r after r rounds
  // buffers of the *_async calls are pinned host memory: kernels read challenges from and store results to them directly
  bool direct_host;
  // profiling: event pairs around the three k_fixed_multiexp launches
  hipEvent_t ev[3][2];
  hipEvent_t fs_ev[3];  // device-transcript pass: hand-offs between the batch's stream and its side stream
  hipEvent_t done;  // recorded behind the last command of each *_async call (vbatch_ready)
  bool ev_pending[3];
  double ev_madds[3];
};

namespace bp {
// record i (dec.h, 80 bytes) = decimal text of the 32-byte little-endian integer i of src; dst may be pinned host memory
__global__ void k_dec_numbers(const uint8_t* __restrict__ src, uint32_t count, uint8_t* __restrict__ dst);
// Asynchronous forms (enqueue on vb->stream, no synchronisation); the host buffers
// must stay valid until the stream has been synchronised (use pinned memory for overlap).
int vbatch_load_async(bp_vbatch* vb, const uint8_t* scalars, const uint8_t* points);
// out_text (pinned, [B][2][80], or null): the decimal text of the commitment's coordinates as dec.h records, for
// the host's transcript; lr_text (pinned, [2k][B][2][80], or null): the same for the proof's L_r, R_r rows, complete
// once round 0 has been enqueued (ignored when vbatch_prepare_async has run for this pass).  Both need direct_host.
int vbatch_begin_async(bp_vbatch* vb, const uint8_t* challenges, uint8_t* out_commitment, uint8_t* out_text,
                       uint8_t* lr_text);
int vbatch_round_async(bp_vbatch* vb, uint32_t round, const uint8_t* x, uint8_t* out_commitment, uint8_t* out_text);
// the pass's challenge-independent side work (L_r, R_r text into lr_text, the rounds' joint tables), ahead of begin
int vbatch_prepare_async(bp_vbatch* vb, uint8_t* lr_text);
// decimal records of A, S, T1, T2 ([4][B][2]) and tBlinding, mu, t ([3][B]) into pinned init_text, on the batch's stream
int vbatch_init_text_async(bp_vbatch* vb, uint8_t* init_text);
int vbatch_finish_async(bp_vbatch* vb, uint8_t* verdicts);
// the whole pass with the Fiat-Shamir transcript on the device (fsdev.h): no host step between load and verdicts
int vbatch_run_fsdev_async(bp_vbatch* vb, uint8_t* verdicts);
int vbatch_wait(bp_vbatch* vb);
// bp_vbatch_create with the caller's promise to set bp_vbatch::pfor (its threads run the latency path's host loops)
int vbatch_create_ex(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, bool pooled_host, bp_vbatch** out);
// After vbatch_wait of a begin or round stage: the commitment as affine bytes in out_commitment ([B][64]).  Batches
// on the latency path (vb->split) normalise on the host here; for the others the kernels have written it already.
int vbatch_collect(bp_vbatch* vb, uint8_t* out_commitment);
// joins the side stream of vbatch_prepare_async (L_r, R_r text, round tables); call before hashing round 0
int vbatch_join_side(bp_vbatch* vb);
// has everything enqueued by the last *_async call completed?  (non-blocking)
bool vbatch_ready(bp_vbatch* vb);
}  // namespace bp
