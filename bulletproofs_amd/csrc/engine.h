// Internal engine state behind the C ABI of include/bpgpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/bpgpu.h"
#include "ec.h"
#include "msm_core.h"

namespace bp {

// A grow-only device buffer owned by the context.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    size_t want = need + need / 4;
    if (hipMalloc(&p, want) != hipSuccess) return -1;
    bytes = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

// Carves aligned sub-buffers out of one allocation.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <class T>
  T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* r = (T*)(base ? base + off : nullptr);
    off += count * sizeof(T);
    return r;
  }
};

}  // namespace bp

struct bp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  bp::DevBuf msm_ws;    // MSM workspace
  bp::DevBuf io;        // staging for host-pointer entry points
  bp::DevBuf gen;       // generator tables (verifier / prover)
  bp::DevBuf vws;       // batch-verifier workspace
  bp::DevBuf pws;       // prover workspace (IPP state, gathered MSM inputs)
  // bp_msm_dev_hs: staging for scalars uploaded from host memory, its copy stream and per-piece events (lazy)
  bp::DevBuf msm_stage;
  hipStream_t msm_copy_stream = nullptr;
  hipEvent_t msm_copy_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t msm_stage_free = nullptr;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // generator state (filled by bp_ctx_set_generators); pointers into `gen`
  uint32_t n_gens = 0;       // generator pairs (gs_i, hs_i); 0 = no generators installed
  uint64_t gen_epoch = 0;    // bumped by every bp_ctx_set_generators: batch objects of an older set are refused
  uint32_t n_bases = 0;      // rows of d_bases: gs, hs, G, H, GS_k
  uint8_t* d_bases = nullptr;  // affine, 64 B each
  uint8_t* d_table = nullptr;  // fixed-base window table, see fixedbase.h
  // streams for the batch verifier's chunks: a set that the hardware runs concurrently (ctx.hip)
  std::vector<hipStream_t> spool;
  std::vector<int> sgroup;  // indices into spool that overlap with each other
  bool sgroup_probed = false;  // the probe found a concurrent set (else sgroup = every stream)
  size_t spool_next = 0;
  size_t aux_next = 0;
  int fb_wbits = 8;            // window width of d_table (8, 12 or 16), chosen in bp_ctx_set_generators
  // transcript conventions the reference inherits from un-vendored packages
  int show_style = 0;    // 0: derived Show "A (P x) (P y)" / "P s"; 1: plain "A x y" / "s"
  int sqrt_branch = 0;   // 0: sr returns a^((p+1)/4); 1: the other root
  // the batch job bp_rangeproof_verify_batch keeps between calls of one shape (rangeproof_host.hip)
  void* vjob_cache = nullptr;
  void (*vjob_cache_free)(void*) = nullptr;
  // optional kernel timing with HIP events on the launching stream (bp_ctx_profile)
  int profile = 0;
  double prof_ms[2] = {0, 0};        // [0] k_fixed_multiexp, [1] k_accumulate
  uint64_t prof_calls[2] = {0, 0};
  double prof_madds[2] = {0, 0};     // mixed additions those launches performed (nominal)
  hipEvent_t msm_ev[2] = {nullptr, nullptr};
  bool prof_pending_light = false;
};

#define BP_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
      return BP_ERR_DEVICE;                                                                   \
    }                                                                                         \
  } while (0)

#define BP_REQUIRE(cond, msg)          \
  do {                                 \
    if (!(cond)) {                     \
      if (ctx) ctx->err = (msg);       \
      return BP_ERR_ARG;               \
    }                                  \
  } while (0)

namespace bp {
// ops.hip: *d_bad (one u32, device) = 1 iff some of the n device-resident points is neither O nor a canonically
// encoded point of the curve; asynchronous on ctx->stream
int points_invalid_dev(bp_ctx* ctx, const uint8_t* d_points, size_t n, uint32_t* d_bad);
// msm.hip
// A stream for one chunk of a batch job, from the context's concurrent set (round-robin).  Not owned by the caller.
hipStream_t ctx_chunk_stream(bp_ctx* ctx);
// a pool stream outside the concurrent chunk set (side work that nothing on a chunk stream waits for)
hipStream_t ctx_aux_stream(bp_ctx* ctx);
int msm_run(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int c_override, void* d_out_affine,
            void* d_out_xyzz);
// the same with the scalars still in (pinned) host memory: uploaded in pieces behind which the bucket work runs
int msm_run_host_scalars(bp_ctx* ctx, const uint8_t* h_scalars, const void* d_points, size_t n, int c_override,
                         void* d_out_affine);
void msm_host_scalars_release(bp_ctx* ctx);
// The same over the file-scope scratch array of msm.hip instead of ctx->msm_ws (bp_msm_dev_static): calls no
// hipMalloc / hipFree.  msm_static_init(device) resolves that device's copy of the array once (bp_ctx_create calls it
// with the device current); msm_static_bytes() is its size.
int msm_static_init(int device);
size_t msm_static_bytes();
int msm_run_static(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int c_override, void* d_out);
// A fixed point set prepared once for many MSMs (bp_msm_points_*): shifted copies 2^(c w) P_i, own workspace.
struct MsmPoints;
int msm_points_create(bp_ctx* ctx, const uint8_t* h_points, const void* d_points, size_t n, MsmPoints** out);
void msm_points_destroy(MsmPoints* mp);
int msm_points_run(MsmPoints* mp, const void* d_scalars, const uint8_t* h_scalars, void* d_out, uint8_t* h_out);
// ipp.hip (device pointers, asynchronous on ctx->stream)
// One batch of two-scalar linear combinations: out[i] = ka*a[i] + kb*b[i] (+ c[i] when c != null), i < n.
// sa / sb, when non-null, give per-lane canonical scalars (32 B each) instead of the shared ka / kb.
// a / b / c advance by `stride_*` bytes per lane (0 = the same point for every lane).
struct LinJob {
  const uint8_t *a, *b, *c;
  const uint8_t *sa, *sb;
  uint8_t* out;
  u256 ka, kb;
  uint32_t n, stride_a, stride_b, stride_c;
};
// up to three jobs in ONE launch (GLV + 15-entry joint table per lane; d_tab: 15 * total lanes XYZZ)
int ipp_lincomb_jobs(bp_ctx* ctx, const LinJob* jobs, int njobs, xyzz* d_tab);
// up to four independent direct MSMs in ONE launch: segment s holds n[s] terms at offset s * stride
// (stride a multiple of 256) of d_scalars / d_points; d_out: nseg affine points.  n[s] <= msm_direct_limit().
int msm_direct_multi(bp_ctx* ctx, const uint8_t* d_scalars, const uint8_t* d_points, uint32_t stride, const uint32_t* n,
                     int nseg, uint8_t* d_out);
size_t msm_direct_limit();
int ipp_fold_points(bp_ctx* ctx, const uint8_t* d_lo, const uint8_t* d_hi, uint32_t n, const uint8_t s_lo[32],
                    const uint8_t s_hi[32], xyzz* d_tab, uint8_t* d_out);
// The latency path's short sums (ipp.hip): bit tables d_sh[point][128] = 2^j * point (16 KB a point), then one lane
// per scalar bit.  ipp_bits_lincomb2: *d_acc <- k0*p0 + k1*p1 + *d_acc (XYZZ, stays on the device);
// ipp_bits_multiexp: d_out[b] = sum over the tables first[b] .. first[b] + count[b] with the scalars (canonical,
// 32 B, device) of the same indices, up to four sums in one launch.
int ipp_bit_tables(bp_ctx* ctx, const uint8_t* d_pts, uint32_t n, xyzz* d_sh, hipStream_t st);
int ipp_bits_lincomb2(bp_ctx* ctx, const xyzz* d_sh0, const xyzz* d_sh1, const uint8_t k0[32], const uint8_t k1[32],
                      xyzz* d_acc);
int ipp_bits_multiexp(bp_ctx* ctx, const xyzz* d_sh, const uint8_t* d_scalars, const uint32_t* first, const uint32_t* count,
                      int nsums, xyzz* d_out);
int ipp_lincomb2_add(bp_ctx* ctx, const uint8_t* d_p0, const uint8_t* d_p1, const uint8_t* d_p2, const uint8_t k0[32],
                     const uint8_t k1[32], xyzz* d_tab, uint8_t* d_out);
int ipp_scale_points(bp_ctx* ctx, const uint8_t* d_pts, const uint8_t* d_scalars, uint32_t n, xyzz* d_tab,
                     uint8_t* d_out);
}  // namespace bp
