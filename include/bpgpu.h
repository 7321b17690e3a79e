/* bpgpu.h - C ABI of the MI355X engine for the elliptic-curve hot path of
 * sdiehl/bulletproofs (secp256k1).
 *
 * This is the boundary a `foreign import ccall` in the reference's Haskell
 * modules binds.  The reference has no FFI today: each
 * entry point below names the pure Haskell function (file:line under
 * /root/reference) whose evaluation it replaces.
 *
 * Encodings (all little-endian, caller-owned buffers):
 *   scalar  : 32 bytes, integer in [0, 2^256); values >= r are accepted and act
 *             modulo the group order r (as `mul` does for an `Fr`).
 *   point   : 64 bytes, affine x || y, each 32 bytes, canonical (< p).  The
 *             identity `O` (`mempty`) is 64 zero bytes ((0,0) is not on the curve).
 *   Outputs are always canonical; points are affine-normalised.
 *
 * Every function returns BP_OK (0) or a negative error; bp_last_error(ctx)
 * returns a message for the most recent failure on that context.  Nothing
 * throws across this boundary.  A context is bound to one GPU and one HIP
 * stream and is not thread-safe: use one context per host thread (the
 * reference is single-threaded, so there are no concurrency semantics to keep).
 *
 * Entry points come in two flavours:
 *   bp_*      host pointers in, host pointers out (what the Haskell shim calls);
 *   bp_*_dev  device pointers (hipMalloc memory on ctx's device), asynchronous on
 *             the context's stream - for callers that keep data resident in HBM.
 */
#ifndef BPGPU_H
#define BPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BP_OK 0
#define BP_ERR_ARG (-1)     /* bad argument (null pointer, size out of range, ...) */
#define BP_ERR_DEVICE (-2)  /* HIP runtime error */
#define BP_ERR_NOMEM (-3)   /* device allocation failed */
#define BP_ERR_STATE (-4)   /* call out of order (e.g. generators not set) */
#define BP_ERR_NOT_IN_RANGE (-5)    /* RangeProofError ValuesNotInRange  (RangeProof/Internal.hs:40-45) */
#define BP_ERR_BOUND_TOO_LARGE (-6) /* RangeProofError UpperBoundTooLarge */
#define BP_ERR_N_NOT_POW2 (-7)      /* RangeProofError NNotPowerOf2 */
#define BP_ERR_REFERENCE_FAILS (-8) /* the reference's own evaluation raises on these arguments (`challenges !! j` with
                                       too few L/R pairs, `2 ^ (-1)` for an empty commitment list, `recip 0`): the
                                       shim raises the same way; no Bool exists to return */

typedef struct bp_ctx bp_ctx;

/* ABI version of this header (major*100 + minor). */
int bp_version(void);

/* Create / destroy an engine context on HIP device `device`. */
int bp_ctx_create(bp_ctx** out, int device);
void bp_ctx_destroy(bp_ctx* ctx);
const char* bp_last_error(const bp_ctx* ctx);
/* Block until all work queued on the context's stream has finished. */
int bp_ctx_sync(bp_ctx* ctx);
/* The context's hipStream_t, as an opaque pointer. */
void* bp_ctx_stream(bp_ctx* ctx);

/* What the context's stream-pool probe found: the value of GPU_MAX_HW_QUEUES the process runs under (0 = unset,
 * HIP's default of 4 applies), the pool size, and how many of the pool's streams were measured to run kernels
 * side by side (0 = none found: the batch verifier's chunks then serialise and a step is ~15 % slower).  The
 * library never changes the environment; the host sets GPU_MAX_HW_QUEUES=16 before HIP initialises
 * (INTEGRATION.md).  Any pointer may be NULL.  No counterpart in the reference. */
int bp_ctx_queue_info(bp_ctx* ctx, int* hw_queues_env, int* pool_streams, int* concurrent_streams);

/* Kernel timing for the benchmark's roofline line: when enabled, the two dominant kernels are
 * bracketed with HIP events on the stream they are launched on.  kernel 0 = k_fixed_multiexp
 * (batch verifier), 1 = k_accumulate (MSM bucket accumulation).  Returns the accumulated device
 * time, launch count and the number of mixed point additions those launches performed.
 * bp_ctx_profile(ctx, x) also resets the counters.  No counterpart in the reference. */
int bp_ctx_profile(bp_ctx* ctx, int enable);
int bp_ctx_profile_read(bp_ctx* ctx, int kernel, double* total_ms, uint64_t* launches, double* mixed_adds);

/* ---- element-wise group operations (batches of n independent operations) ----
 *
 * bp_point_mul      out[i] = scalars[i] * points[i]
 *                   replaces `mul` (Data.Curve.Weierstrass, imported at
 *                   Bulletproofs/Utils.hs:9; call sites Utils.hs:73,
 *                   MultiRangeProof/Prover.hs:107, RangeProof/Internal.hs:185-205).
 * bp_point_add      out[i] = a[i] + b[i]
 *                   replaces `<>` / `add` (Utils.hs:73,78; test/Test/Field.hs:23).
 * bp_point_neg      out[i] = -a[i]
 *                   replaces `inv` (RangeProof/Internal.hs:187,196).
 * bp_add_two_mul    out[i] = e0[i]*p0[i] + e1[i]*p1[i]
 *                   replaces `addTwoMulP` (Utils.hs:72-73), incl. the generator
 *                   folds of InnerProductProof/Prover.hs:81-82.
 * bp_point_on_curve ok[i] = 1 if points[i] is O or has canonical coordinates (< p) that satisfy y^2 = x^3 + 7
 *                   replaces `def` (Utils.hs:42).
 */
int bp_point_mul(bp_ctx* ctx, const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out);
int bp_point_add(bp_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);
int bp_point_neg(bp_ctx* ctx, const uint8_t* a, size_t n, uint8_t* out);
int bp_add_two_mul(bp_ctx* ctx, const uint8_t* e0, const uint8_t* p0, const uint8_t* e1, const uint8_t* p1,
                   size_t n, uint8_t* out);
int bp_point_on_curve(bp_ctx* ctx, const uint8_t* points, size_t n, uint8_t* ok);

/* ---- multi-scalar multiplication ----
 *
 * bp_msm            out = sum_i scalars[i] * points[i]
 *                   replaces `sumExps` (Bulletproofs/Utils.hs:76-80).  The
 *                   caller truncates to the shorter list as the reference does
 *                   (Utils.hs:79); n = 0 gives O (Utils.hs:80).
 * bp_msm_dev        same with device-resident inputs/outputs; `d_out` receives
 *                   one 64-byte affine point.  window_bits = 0 picks the window
 *                   size from n; 1..16 forces it (benchmark / test knob).
 */
int bp_msm(bp_ctx* ctx, const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out);
int bp_msm_dev(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits, void* d_out);
/* bp_msm_dev_hs     the same `sumExps` with the POINTS resident in HBM and the SCALARS still in host memory (pinned
 *                   memory for the overlap below; pageable works, synchronously): a Pedersen vector commitment over
 *                   resident generators whose exponents arrive per call (BASELINE config 3: "points uploaded once").
 *                   One call instead of a copy, a synchronisation and bp_msm_dev.  BP_MSM_HS_PARTS=2..8 cuts the
 *                   point range into pieces whose uploads are queued at once, piece i's bucket work running while
 *                   piece i + 1 crosses PCIe (each piece fills its own bucket array, one reduction follows); on MI355X
 *                   that overlap costs more than it returns, so the default is one piece.  Same result
 *                   bytes as bp_msm_dev either way.
 *                   Asynchronous on the context's stream; h_scalars must stay valid until that stream has passed the
 *                   call (bp_ctx_sync). */
int bp_msm_dev_hs(bp_ctx* ctx, const uint8_t* h_scalars, const void* d_points, size_t n, int window_bits, void* d_out);

/* ---- multi-scalar multiplication over a FIXED point list ----
 * `sumExps es ps` where `ps` is the same list call after call - a Pedersen vector commitment over fixed generators,
 * the shape of commitBitVectors (RangeProof/Internal.hs:136-147) and of BASELINE config 3 ("points uploaded once").
 * bp_msm_points_create(_dev) uploads the n points once (host / device pointer) and precomputes, per point, the
 * copies 2^(c w) P for the bucket method's windows (W x n x 64 B of HBM), so that every window's buckets carry
 * weight 1: no window doublings and one bucket reduction instead of W.  Results are the same group elements as
 * bp_msm's, byte for byte.
 *   bp_msm_points_run      scalars in HOST memory (n x 32 B; pinned memory lets the upload overlap the recoding:
 *                          it is cut into chunks, each recoded as soon as it has landed), out: 64 B host; synchronous
 *   bp_msm_points_run_dev  scalars and the 64-byte result in device memory, asynchronous on the context's stream
 * A handle belongs to the context it was created on; calls on one handle must not overlap.
 */
typedef struct bp_msm_points bp_msm_points;
int bp_msm_points_create(bp_ctx* ctx, const uint8_t* points, size_t n, bp_msm_points** out);
int bp_msm_points_create_dev(bp_ctx* ctx, const void* d_points, size_t n, bp_msm_points** out);
void bp_msm_points_destroy(bp_msm_points* mp);
int bp_msm_points_run(bp_ctx* ctx, bp_msm_points* mp, const uint8_t* scalars, uint8_t* out);
int bp_msm_points_run_dev(bp_ctx* ctx, bp_msm_points* mp, const void* d_scalars, void* d_out);

/* bp_point_sum_dev   d_out = sum of n affine points, device pointers, asynchronous on the context's stream: the fold
 *                   of `<>` over a list (the right-nested sum of Utils.hs:78-80 without the scalars).  Combines the
 *                   per-GPU partial sums of a sumExps whose point range is sharded over GPUs. */
int bp_point_sum_dev(bp_ctx* ctx, const void* d_points, size_t n, void* d_out);

/* bp_msm_dev_static  bp_msm_dev for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 *                   harnesses that forbid hipMalloc.  It has NO counterpart in the reference.  Same arguments,
 *                   same result bytes, asynchronous on the context's stream; it calls no hipMalloc, hipFree or
 *                   other allocating routine.  Its workspace (digit codes, bucket lists, histograms, bucket and
 *                   partial sums) is a 32 MiB static device array that every GPU the library runs on holds
 *                   until the process ends.  Limits: n <= BP_MSM_STATIC_MAX_N, and a forced window_bits whose
 *                   workspace exceeds the array is refused (BP_ERR_ARG) - nothing is launched then.  All calls
 *                   on one GPU share that GPU's array: calls that overlap in time on one GPU (two contexts, two
 *                   host threads) race; calls on one context's stream are ordered and safe.  Kernel timing
 *                   (bp_ctx_profile) does not cover this entry point.
 */
#define BP_MSM_STATIC_MAX_N 65536u
int bp_msm_dev_static(bp_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits,
                      void* d_out);

/* ---- generator set ----
 *
 * bp_ctx_set_generators  uploads the reference's generator lists and builds the
 *                   device-resident fixed-base tables.  gs, hs: n_gens affine
 *                   points each (`take n gs`, `take n hs`; Bulletproofs/Utils.hs:19-25),
 *                   h: one point (Utils.hs:14-15); `gen` is built in.  n_gens is a
 *                   power of two in 1..8192 and bounds n*m of every later proof.
 *                   The lists are CAFs in the reference (evaluated once per
 *                   process); here they are evaluated once per context.
 */
int bp_ctx_set_generators(bp_ctx* ctx, const uint8_t* gs, const uint8_t* hs, const uint8_t* h, uint32_t n_gens);

/* ---- batched range-proof verification, lock-step state machine ----
 *
 * Verifies `batch` independent proofs of one shape (n_bits, m) with the
 * Fiat-Shamir transcript kept on the host: the caller hashes, this engine does
 * the scalar-field and curve work of
 *   MultiRangeProof.verifyProof   (Bulletproofs/MultiRangeProof/Verifier.hs:19-88),
 *   computeLRCommitment           (Bulletproofs/RangeProof/Internal.hs:169-206),
 *   InnerProductProof.verifyProof (Bulletproofs/InnerProductProof/Verifier.hs:18-73).
 * All arrays are structure-of-arrays: row r of an array holds that field for
 * proofs 0..batch-1 consecutively.
 *
 *   bp_vbatch_create   n_bits = log2(upperBound) (power of two <= 128), m = number of
 *                      commitments after padding to a power of two (Verifier.hs:37-38)
 *   bp_vbatch_load     scalars: 5 rows x 32 B: tBlinding, mu, t, l, r
 *                      points : (4 + 2k + m) rows x 64 B: aCommit, sCommit, t1Commit,
 *                      t2Commit, lCommits[0..k), rCommits[0..k), vCommits[0..m), k = log2(n_bits*m)
 *   bp_vbatch_begin    challenges: 4 rows x 32 B: y, z, x (shamirY/Z/X, Utils.hs:145-165) and
 *                      uChallenge (shamirU, Utils.hs:172-174); returns the commitment
 *                      P = computeLRCommitment(...) per proof (64 B affine each)
 *   bp_vbatch_round    round r = 0..k-1 of mkChallenges (Verifier.hs:44-53): x = shamirX'(acc, L_r, R_r)
 *                      per proof; returns acc' = x^2 L_r + x^-2 R_r + acc per proof
 *   bp_vbatch_finish   verdicts[i] = 1 iff verifyTPoly && verifyLRCommitment hold for proof i
 */
typedef struct bp_vbatch bp_vbatch;
int bp_vbatch_create(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, bp_vbatch** out);
void bp_vbatch_destroy(bp_vbatch* vb);
uint32_t bp_vbatch_rounds(const bp_vbatch* vb);
uint32_t bp_vbatch_point_rows(const bp_vbatch* vb);
int bp_vbatch_load(bp_vbatch* vb, const uint8_t* scalars, const uint8_t* points);
int bp_vbatch_begin(bp_vbatch* vb, const uint8_t* challenges, uint8_t* out_commitment);
int bp_vbatch_round(bp_vbatch* vb, uint32_t round, const uint8_t* x, uint8_t* out_commitment);
int bp_vbatch_finish(bp_vbatch* vb, uint8_t* verdicts);

/* ---- host-side protocol layer (transcript on the CPU, curve work on the GPU) ----
 *
 * These mirror the reference's own functions one level above the engine: they
 * contain the Fiat-Shamir hashing (`oracle`, `shamir*`, Utils.hs:29-33,145-174)
 * and the control flow, multi-threaded over independent proofs.
 *
 * bp_ctx_set_show_style   text format `show` gives points/scalars in transcripts
 *                   (0 = derived Show "A (P x) (P y)" / "P s", the default; 1 = "A x y" / "s").
 *                   The format is defined by packages outside the reference tree.
 * bp_ctx_set_sqrt_branch  which root `sr` returns in generateH (0 = a^((p+1)/4), 1 = the other).
 * bp_generators_derive    evaluates `h`, `take n gs`, `take n hs` (Utils.hs:14-25,38-47), returns
 *                   them (outputs may be NULL) and installs them with bp_ctx_set_generators.
 * bp_log_base2      `logBase2` (Utils.hs:93-94) on a little-endian integer, in Double arithmetic
 *                   as the reference does; -1 for zero.
 * bp_transcript     the reference's hash-to-scalar functions, for parity tests:
 *                   which = 0 shamirY(A,S)  1 shamirZ(A,S,y)  2 shamirX(A,S,T1,T2,y,z)
 *                   3 shamirX'(acc,L,R) [pts = acc,L,R]  4 shamirU(tBlinding,mu,t)
 *                   5 oracle(raw) into Fr  6 oracle(raw) into Fq
 *                   7 shamirGxGxG(p1,p2,p3)  8 the circuit protocol's shamirZ(z) [scs = z]
 *                   9 shamirGs(ps) over raw_len points (raw unused)
 *                   (ArithmeticCircuit/Internal.hs:109-117).  out: 32 bytes.
 *
 * bp_rangeproof_verify_batch   verdicts[i] = MultiRangeProof.verifyProof 2^n_bits vCommits_i proof_i
 *                   (MultiRangeProof/Verifier.hs:19-39; m = 1 is RangeProof.verifyProof,
 *                   RangeProof/Verifier.hs:17-24) for `batch` independent proofs = `map verifyProof`.
 *                   scalars / points use the bp_vbatch_load layout over the whole batch;
 *                   vCommits are already padded to m = 2^ceil(log2 m) with O.
 * bp_vjob_*         the same split into create (workspaces, pinned staging, worker threads),
 *                   load (proofs -> pinned host + HBM) and run (the transcript + GPU pipeline);
 *                   n_chunks = 0 and host_threads = 0 pick defaults (env BP_HOST_THREADS).
 */
int bp_ctx_set_show_style(bp_ctx* ctx, int style);
int bp_ctx_set_sqrt_branch(bp_ctx* ctx, int branch);
int bp_generators_derive(bp_ctx* ctx, uint32_t n_gens, uint8_t* gs_out, uint8_t* hs_out, uint8_t* h_out);
/* The evaluated CAFs `h`, `take n gs`, `take n hs` (Utils.hs:14-25) as a file: save writes the generators the
 * context holds (with the show style / sqrt branch they were derived under and a SHA-256 checksum); load checks
 * the file (magic, size, checksum, every entry a finite curve point), sets the context's style / branch to the
 * recorded ones and installs the generators like bp_ctx_set_generators.  What a long-lived Haskell process gets from
 * the memoised top-level lists, a fresh process gets from the file (benchmarks/gen_tables.py writes one). */
int bp_generators_save(bp_ctx* ctx, const char* path);
int bp_generators_load(bp_ctx* ctx, const char* path);
int bp_log_base2(const uint8_t* upper_bound, size_t len);
int bp_transcript(int style, int which, const uint8_t* pts, const uint8_t* scs, const uint8_t* raw, size_t raw_len,
                  uint8_t* out);
int bp_rangeproof_verify_batch(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, const uint8_t* scalars,
                               const uint8_t* points, uint8_t* verdicts);
/* bp_rangeproof_verify   *verdict = MultiRangeProof.verifyProof upperBound vCommits proof for ONE proof
 *                   (MultiRangeProof/Verifier.hs:19-88), TOTAL over its arguments like the reference's function, which
 *                   checks neither the bound nor the proof's shape: n = logBase2 upperBound in Double arithmetic
 *                   (Utils.hs:93-94; any integer, e.g. upperBound = 12 gives n = 3, upperBound = 0 acts as n = 0),
 *                   m = length vCommits (unpadded; the padding with O to 2^log2Ceil m happens inside, Verifier.hs:37-38),
 *                   k = the number of (L, R) pairs `zip lCommits rCommits` yields - whatever the proof holds.
 *                   upper_bound: little-endian integer of ub_len bytes.  proof_scalars: 5 x 32 B (tBlinding, mu, t,
 *                   l, r); proof_points: (4 + 2k) x 64 B (aCommit, sCommit, t1Commit, t2Commit, L_0.., R_0..).
 *                   A proof of the honest shape (n a power of two <= 128, k = log2(n m)) runs on the batch engine;
 *                   every other input follows the reference's definitions literally on the same GPU primitives
 *                   (computeLRCommitment over materialised hs', RangeProof/Internal.hs:169-206; mkChallenges over k
 *                   pairs and mkOtherExponents for any n, InnerProductProof/Verifier.hs:40-73).  This is the domain
 *                   of the reference's negative test prop_differentUpperBound (test/Test/Protocol/RangeProof.hs:173-181).
 *                   Returns BP_ERR_REFERENCE_FAILS where the reference raises instead of returning a Bool: m = 0, or
 *                   verifyTPoly holds and the proof has fewer pairs than logBase2 (n m) (`and` is lazy: a failing
 *                   verifyTPoly gives False first).  BP_ERR_STATE when n * 2^log2Ceil m exceeds the context's
 *                   generators (`gs`, `hs` are infinite lists in the reference; derive more and call again). */
int bp_rangeproof_verify(bp_ctx* ctx, const uint8_t* upper_bound, size_t ub_len, const uint8_t* v_commits, uint32_t m,
                         const uint8_t* proof_scalars, const uint8_t* proof_points, uint32_t k, uint8_t* verdict);
/* bp_vjob_*         the batch job behind bp_rangeproof_verify_batch (map (verifyProof upperBound), MultiRangeProof/
 *                   Verifier.hs:19-39): create (workspaces, pinned staging, hashing threads) / load (untimed upload) /
 *                   run (the timed pass) / submit + wait (the pass on a driver thread of the job).  n_chunks = 0 and
 *                   host_threads = 0 pick defaults.  Where the Fiat-Shamir transcript is formed is fixed when the job
 *                   is created: on the host's threads (default; the reference's `oracle` / `shamir*`, Utils.hs:29-33,
 *                   145-174, in C++), or - BP_FS_DEVICE=1 in the environment - by kernels on the GPU from the same
 *                   bytes (no host step inside a pass, one chunk by default, no hashing threads used).  Verdicts are
 *                   identical; batches on the latency path (<= 768 proofs) always use the host transcript. */
typedef struct bp_vjob bp_vjob;
int bp_vjob_create(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, uint32_t n_chunks, int host_threads,
                   bp_vjob** out);
void bp_vjob_destroy(bp_vjob* job);
int bp_vjob_load(bp_vjob* job, const uint8_t* scalars, const uint8_t* points);
int bp_vjob_run(bp_vjob* job, uint8_t* verdicts);
/* bp_vjob_submit / bp_vjob_wait   bp_vjob_run in two halves, for a serving loop that keeps TWO jobs in flight on one
 *                   context (`map verifyProof` over a stream of batches): submit starts the pass on a driver thread
 *                   owned by the job and returns at once, wait blocks until it is done and returns its verdicts
 *                   (the same bytes bp_vjob_run gives).  The beginning and the end of a pass leave the GPU partly
 *                   idle (first hashes, last rounds, final multi-exponentiation of the last chunk: ~16 % of a
 *                   65,536-proof pass); with a second job submitted meanwhile those ends overlap.  One pass per job at
 *                   a time (BP_ERR_STATE otherwise: load / run / submit while in flight, wait without submit); the
 *                   jobs must belong to the same context only in the sense that they share its generators - each has
 *                   its own streams, buffers and hashing threads.  bp_ctx_profile must be off.  No counterpart in the
 *                   reference (a pure function has no notion of overlap); the verdicts are its Bools. */
int bp_vjob_submit(bp_vjob* job);
int bp_vjob_wait(bp_vjob* job, uint8_t* verdicts);

/* ---- prover path ----
 *
 * bp_ipp_prove      InnerProductProof.generateProof on an arbitrary base
 *                   (Bulletproofs/InnerProductProof/Prover.hs:17-92): Gs, Hs n points, u = bH, P = commitmentLR,
 *                   ls, rs n scalars; n = 0 or a power of two.  L_out, R_out: log2(n) points each in round
 *                   order; l_out, r_out: 32 bytes.  Per round the L/R commitments are one bucket MSM each, the
 *                   generator folds (Prover.hs:81-82) one GLV kernel each; x = shamirX' is hashed on the host.
 * bp_upper_bound_bits   the validation of generateProof (MultiRangeProof/Prover.hs:25-38): n = log2(upperBound),
 *                   BP_ERR_BOUND_TOO_LARGE if upperBound >= r, BP_ERR_N_NOT_POW2 unless upperBound = 2^n with
 *                   n a power of two.  upper_bound: little-endian integer of `len` bytes.
 * bp_rangeproof_prove   MultiRangeProof.generateProof (checked != 0, with checkRanges -> BP_ERR_NOT_IN_RANGE)
 *                   or generateProofUnsafe (checked = 0) (MultiRangeProof/Prover.hs:19-123); m = 1 is
 *                   RangeProof.generateProof (RangeProof/Prover.hs:15-32).  The MonadRandom draws become explicit
 *                   inputs in draw order: sL, sR (n_bits*m2 scalars each, m2 = m rounded up to a power of two,
 *                   Utils.hs:135-139), alpha = aBlinding, rho = sBlinding (Prover.hs:69-70), tau1, tau2 =
 *                   t1Blinding, t2Blinding (:81-82).  Outputs: proof_scalars 5 x 32 B (tBlinding, mu, t, l, r),
 *                   proof_points (4 + 2k) x 64 B (aCommit, sCommit, t1Commit, t2Commit, lCommits, rCommits).
 */
int bp_ipp_prove(bp_ctx* ctx, uint32_t n, const uint8_t* Gs, const uint8_t* Hs, const uint8_t* u, const uint8_t* P,
                 const uint8_t* ls, const uint8_t* rs, uint8_t* L_out, uint8_t* R_out, uint8_t* l_out,
                 uint8_t* r_out);
/* bp_ipp_verify     InnerProductProof.verifyProof n base commitmentLR proof on an arbitrary base
 *                   (Bulletproofs/InnerProductProof/Verifier.hs:18-73, exported at InnerProductProof.hs:1-7).
 *                   n: the reference's Integer argument (<= 2^20); Gs = bGs (n_gs points), Hs = bHs (n_hs points),
 *                   u = bH, P = commitmentLR; L, R: k points each (zip lCommits rCommits); l, r: 32 bytes each.
 *                   *verdict = 1/0.  Lists shorter than n truncate the sums exactly as `sumExps` does
 *                   (Utils.hs:79); fewer rounds than logBase2 n, where the reference's `!!` fails, is
 *                   BP_ERR_REFERENCE_FAILS.  Points that are neither O nor canonical curve points (proof or bases) and
 *                   l, r >= r do not verify: the reference's types cannot hold them. */
int bp_ipp_verify(bp_ctx* ctx, uint64_t n, const uint8_t* Gs, uint32_t n_gs, const uint8_t* Hs, uint32_t n_hs,
                  const uint8_t* u, const uint8_t* P, const uint8_t* L, const uint8_t* R, uint32_t k, const uint8_t* l,
                  const uint8_t* r, uint8_t* verdict);
int bp_upper_bound_bits(const uint8_t* upper_bound, size_t len, uint32_t* n_bits);
int bp_rangeproof_prove(bp_ctx* ctx, uint32_t n_bits, uint32_t m, const uint8_t* values, const uint8_t* blindings,
                        const uint8_t* sL, const uint8_t* sR, const uint8_t* alpha, const uint8_t* rho,
                        const uint8_t* tau1, const uint8_t* tau2, int checked, uint8_t* proof_scalars,
                        uint8_t* proof_points);

/* ---- arithmetic circuits (host control in C++ over the same engine) ----
 * A circuit (ArithCircuit, ArithmeticCircuit/Internal.hs:42-58) is passed as dense row-major matrices of
 * canonical scalars: wL, wR, wO are q x n (n = multiplication gates, unpadded), wV = commitmentWeights is
 * q x m, cs has q entries.  padCircuit / padAssignment (Internal.hs:76-97) happen inside: n2 = n rounded up
 * to a power of two (0 stays 0), k = log2(n2).
 *
 * bp_ac_proof_points   number of points in a proof for n gates: 10 + 2k.
 * bp_ac_prove       ArithmeticCircuit.generateProof (ArithmeticCircuit/Prover.hs:16-117).  a_l, a_r, a_o: n
 *                   scalars (Assignment); commit_blinders: m scalars (ArithWitness.commitBlinders; the witness's
 *                   `commitments` are not read by the prover).  The MonadRandom draws are explicit inputs in
 *                   draw order: blind3 = aiBlinding, aoBlinding, sBlinding (:25-27); s_l, s_r: n2 scalars each
 *                   (chooseBlindingVectors n, :32); tau5: the five tBlindings of :57 (they land at degrees
 *                   1, 3, 4, 5, 6).  Outputs: proof_scalars 5 x 32 B (tBlinding, mu, t, l, r); proof_points
 *                   (10 + 2k) x 64 B (aiCommit, aoCommit, sCommit, tCommits[0..6], lCommits, rCommits).
 * bp_ac_verify      ArithmeticCircuit.verifyProof (ArithmeticCircuit/Verifier.hs:16-71): *verdict = 1/0.
 *                   v_commits: m points.  Needs generators for n2 (bp_generators_derive).
 */
int bp_ac_proof_points(uint32_t n_gates);
int bp_ac_prove(bp_ctx* ctx, uint32_t n, uint32_t q, uint32_t m, const uint8_t* wL, const uint8_t* wR,
                const uint8_t* wO, const uint8_t* wV, const uint8_t* a_l, const uint8_t* a_r, const uint8_t* a_o,
                const uint8_t* commit_blinders, const uint8_t* blind3, const uint8_t* s_l, const uint8_t* s_r,
                const uint8_t* tau5, uint8_t* proof_scalars, uint8_t* proof_points);
int bp_ac_verify(bp_ctx* ctx, uint32_t n, uint32_t q, uint32_t m, const uint8_t* wL, const uint8_t* wR,
                 const uint8_t* wO, const uint8_t* wV, const uint8_t* cs, const uint8_t* v_commits,
                 const uint8_t* proof_scalars, const uint8_t* proof_points, uint8_t* verdict);

/* ---- canonical proof bytes and `show` text (host only) ----
 * The reference defines no serialisation (its records derive Show/Eq/Generic/NFData only,
 * RangeProof/Internal.hs:14-38, InnerProductProof/Internal.hs:10-24); this layout is this library's:
 *   tBlinding | mu | t | aCommit | sCommit | t1Commit | t2Commit | k | L_1..L_k | R_1..R_k | l | r
 * scalars 32 B big-endian canonical; points 0x04|x|y (big-endian, canonical, on the curve) or 0x00 for O;
 * k one byte.  In memory: scalars 5 x 32 B little-endian (tBlinding, mu, t, l, r), points (4 + 2k) x 64 B as
 * bp_rangeproof_prove returns them.
 * bp_rangeproof_wire_size   upper bound of the encoded size for k rounds.
 * bp_rangeproof_serialize / _deserialize   strict: non-canonical values, off-curve points, unknown tags,
 *                   truncation and trailing bytes are BP_ERR_ARG.
 * bp_rangeproof_show   the text GHC's derived Show gives for `RangeProof Fr PA` (record syntax), with the
 *                   element format of bp_ctx_set_show_style: for eyeball diffs against a GHC run.
 */
size_t bp_rangeproof_wire_size(uint32_t k);
int bp_rangeproof_serialize(const uint8_t* scalars, const uint8_t* points, uint32_t k, uint8_t* out, size_t cap,
                            size_t* len);
int bp_rangeproof_deserialize(const uint8_t* in, size_t len, uint32_t* k, uint8_t* scalars, uint8_t* points,
                              size_t points_cap);
int bp_rangeproof_show(const uint8_t* scalars, const uint8_t* points, uint32_t k, int style, char* out, size_t cap,
                       size_t* len);

/* ---- field arithmetic self-test hooks (used by the parity tests) ----
 * op: 0 add, 1 sub, 2 mul, 3 square(a), 4 inverse(a), 5 negate(a),
 *     6 a^((p+1)/4), the square-root candidate `sr` uses (field 0 only; Utils.hs:47)
 * field: 0 = Fq (base field, pseudo-Mersenne path), 1 = Fr (Montgomery path),
 *        2 = Fq through the generic Montgomery path
 * a, b, out: n canonical 32-byte elements.  Replaces `+ - * recip negate` on
 * `Fr`/`Fq` (galois-field, imported at Utils.hs:8; laws in test/Test/Field.hs:39-64).
 */
int bp_field_op(bp_ctx* ctx, int field, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);

/* ---- scalar-field vector kernels: self-test hooks (used by the parity tests) ----
 * The provers keep their witness vectors in HBM and run the reference's vector algebra as kernels; these hooks run the
 * same kernels on caller data (canonical 32-byte scalars, host pointers).
 * bp_fr_vector_op      op 0 powerVector a n (Utils.hs:50-52, incl. powerVector 0 n = zeros; a: one scalar, b unused)
 *                      op 1 hadamard a b (Utils.hs:55-58)   op 2 dot a b (Utils.hs:60-61; one scalar out)
 * bp_rp_polys_selftest computeLRPolys, computeTPoly and ls = l0 + x l1, rs = r0 + x r1
 *                      (MultiRangeProof/Prover.hs:90-91,133-169).  bits: aL as n_bits * m bytes of 0/1.
 *                      out: 5 n m + 3 scalars: l0 | r0 | r1 | ls | rs | t0 t1 t2.
 * bp_ipp_fold_selftest `rounds` steps of generateProof' on the witness (InnerProductProof/Prover.hs:60-61,84-85) with
 *                      challenges xs, and the scalars that express lCommit, rCommit and the folded commitment over
 *                      the ORIGINAL generators (hs' = hscale_i * hs_i, bH = ub * gen).  out: nk = n >> rounds;
 *                      ls' (nk) | rs' (nk) | dot, ub*cL, ub*cR | 3 x (2n + 1) scalars (gs part, hs part, gen).
 */
int bp_fr_vector_op(bp_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint32_t n, uint8_t* out);
int bp_rp_polys_selftest(bp_ctx* ctx, uint32_t n_bits, uint32_t m, const uint8_t* bits, const uint8_t* s_l,
                         const uint8_t* s_r, const uint8_t* y, const uint8_t* z, const uint8_t* x, uint8_t* out);
int bp_ipp_fold_selftest(bp_ctx* ctx, uint32_t n, const uint8_t* ls, const uint8_t* rs, const uint8_t* hscale,
                         const uint8_t* ub, uint32_t rounds, const uint8_t* xs, uint8_t* out);

/* ---- decimal `show` digits on the device (self-test hook of the verifier's text path) ----
 * The transcript hashes `show` text of points and scalars (`pointToBS = show`, Utils.hs:32-33,145-174).  The
 * batch verifier's kernels format the numbers they produce (and the proof's L_r, R_r) themselves and hand
 * the host 80-byte records: byte 0 = digit count n (1..78), ASCII digits right-aligned in [80 - n, 80).
 * This entry point runs the same kernel on n 32-byte little-endian integers (host pointers) and returns
 * their records, so the tests can compare them with the oracle's decimal text.  Hashing stays on the host.
 */
int bp_show_numbers(bp_ctx* ctx, const uint8_t* numbers, size_t n, uint8_t* records);

#ifdef __cplusplus
}
#endif
#endif /* BPGPU_H */
