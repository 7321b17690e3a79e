// Host side of the drop-in: the control flow and Fiat-Shamir transcript of
//   Bulletproofs.MultiRangeProof.verifyProof  (/root/reference/Bulletproofs/MultiRangeProof/Verifier.hs:19-39)
//   Bulletproofs.RangeProof.verifyProof       (RangeProof/Verifier.hs:17-24)
//   mkChallenges                              (InnerProductProof/Verifier.hs:40-53)
//   gs / hs / h / generateH                   (Utils.hs:14-25,38-47)
// written in C++ above the engine ABI (the reference is compiled Haskell and no
// GHC exists in this image).  Hashing is multi-threaded over independent proofs
// and ping-pongs with the GPU: the batch is cut into chunks, and while the GPU
// runs one chunk's round the host hashes the other chunk's transcript.
#include <sched.h>
#include <stdio.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"
#include "transcript.h"
#include "verify.h"

using namespace bp;
using namespace bp::host;

namespace {

// Minimal persistent worker pool: run(f, n) calls f(i) for i in [0, n) in blocks.
class Pool {
 public:
  explicit Pool(int nthreads) : n_(nthreads < 1 ? 1 : nthreads) {
    for (int t = 1; t < n_; t++) workers_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  int size() const { return n_; }
  void run(const std::function<void(size_t, size_t)>& f, size_t n, size_t grain = 64) {
    if (n == 0) return;
    if (n <= grain || n_ == 1) {  // one block: waking the workers (~30 us) would cost more than the work
      f(0, n);
      return;
    }
    fn_ = &f;
    total_ = n;
    grain_ = grain;
    next_.store(0);
    {
      std::lock_guard<std::mutex> lk(mu_);
      pending_ = n_ - 1;
      gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
  }

 private:
  void work() {
    for (;;) {
      size_t a = next_.fetch_add(grain_);
      if (a >= total_) break;
      size_t b = a + grain_ < total_ ? a + grain_ : total_;
      (*fn_)(a, b);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  uint64_t gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
  const std::function<void(size_t, size_t)>* fn_ = nullptr;
  size_t total_ = 0, grain_ = 64;
  std::atomic<size_t> next_{0};
};

// Hashing threads of a job when the caller names none: BP_HOST_THREADS, else the cores this process may run on (its
// affinity mask: the box's CPU share, not the machine's core count) divided by the ranks that share the host
// (BP_LOCAL_RANKS, else torchrun's LOCAL_WORLD_SIZE), at most 16 (a 65,536-proof step fits 18 ms + 114 ms / threads).
// Eight ranks on one node therefore do not start 8 x 16 threads.
int default_threads() {
  const char* e = getenv("BP_HOST_THREADS");
  if (e && atoi(e) > 0) return atoi(e);
  int cores = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
  if (cores <= 0) cores = (int)std::thread::hardware_concurrency();
  if (cores <= 0) cores = 1;
  int ranks = 1;
  const char* lr = getenv("BP_LOCAL_RANKS");
  if (!lr) lr = getenv("LOCAL_WORLD_SIZE");
  if (lr && atoi(lr) > 0) ranks = atoi(lr);
  int t = cores / ranks;
  if (t < 1) t = 1;
  return t > 16 ? 16 : t;
}

// Where the Fiat-Shamir transcript is formed.  The default keeps it on the host's threads (north_star: "transcript
// hashing and protocol control stay on the host"; SURVEY.md 8f-1 allows a device-side SHA-256 + decimal kernel only as an
// explicit choice).  BP_FS_DEVICE=1 is that choice: text, SHA-256 and challenges on the GPU (fsdev.h), a pass without a
// host step.  Read when a job is created (the job keeps its form), so one process can hold jobs of both forms.
bool fs_device_enabled() {
  const char* e = getenv("BP_FS_DEVICE");
  return e && atoi(e) != 0;
}

struct Chunk {
  bp_vbatch* vb = nullptr;
  uint32_t B = 0;       // proofs in this chunk
  size_t first = 0;     // index of its first proof in the job
  // pinned host memory
  uint8_t *sc = nullptr, *pts = nullptr, *chal = nullptr, *xk = nullptr, *acc = nullptr, *verdict = nullptr;
  // decimal text written by the kernels (dec.h records): the running commitment, and the rows L_r, R_r
  uint8_t *acc_txt = nullptr, *lr_txt = nullptr;
  uint8_t* init_txt = nullptr;  // A, S, T1, T2 and tBlinding, mu, t (vbatch_init_text_async)
};

}  // namespace

struct bp_vjob {
  bp_ctx* ctx;
  uint32_t n_bits, m, batch, k, npts;
  ShowStyle style;
  std::vector<Chunk> chunks;
  Pool* pool;
  void* pinned = nullptr;
  bool loaded = false;
  bool fsdev = false;  // the transcript on the device (vbatch_run_fsdev_async): set at creation, off for the latency path
  // bp_vjob_submit / bp_vjob_wait: the run driven by a thread of its own, so that a second job's pass can fill the
  // GPU while this one's drains
  std::thread driver;
  bool in_flight = false;
  int driver_rc = BP_OK;
  std::string driver_err;
  std::vector<uint8_t> driver_verdicts;
};

static int vjob_run_impl(bp_vjob* job, uint8_t* verdicts, std::string& err);

extern "C" {

int bp_ctx_set_show_style(bp_ctx* ctx, int style) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(style == 0 || style == 1, "style must be 0 (derived Show) or 1 (plain)");
  ctx->show_style = style;
  return BP_OK;
}

int bp_ctx_set_sqrt_branch(bp_ctx* ctx, int branch) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(branch == 0 || branch == 1, "branch must be 0 or 1");
  ctx->sqrt_branch = branch;
  return BP_OK;
}

void bp_vjob_destroy(bp_vjob* job) {
  if (!job) return;
  if (job->driver.joinable()) job->driver.join();  // a submitted pass finishes before its buffers go
  (void)hipSetDevice(job->ctx->device);
  for (auto& c : job->chunks)
    if (c.vb) bp_vbatch_destroy(c.vb);
  if (job->pinned) (void)hipHostFree(job->pinned);
  delete job->pool;
  delete job;
}

int bp_vjob_create(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, uint32_t n_chunks, int host_threads,
                   bp_vjob** out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(out, "null pointer");
  *out = nullptr;
  BP_REQUIRE(batch >= 1, "batch must be >= 1");
  // Host transcript (the default): chunks ping-pong between host hashing and the GPU; four streams of the context's
  // concurrent set overlap.  Device transcript (BP_FS_DEVICE=1): no host step inside a pass, so ONE chunk - its chain
  // kernels then have a wave on every SIMD and its multi-exponentiations several.
  if (n_chunks == 0) n_chunks = fs_device_enabled() ? 1 : (batch >= 16384 ? 4 : (batch >= 4096 ? 2 : 1));
  BP_REQUIRE(n_chunks <= 64 && n_chunks <= batch, "n_chunks out of range");
  BP_TRY(hipSetDevice(ctx->device));
  bp_vjob* job = new bp_vjob();
  job->ctx = ctx;
  job->n_bits = n_bits;
  job->m = m;
  job->batch = batch;
  job->style = (ShowStyle)ctx->show_style;
  job->fsdev = fs_device_enabled();
  job->pool = new Pool(host_threads > 0 ? host_threads : default_threads());
  job->chunks.resize(n_chunks);
  size_t per = (batch + n_chunks - 1) / n_chunks;
  // diagnostic: BP_CHUNK_WEIGHTS="w0,w1,..." (one weight per chunk) sizes the chunks in those proportions, in multiples
  // of 256 proofs, instead of equally
  std::vector<size_t> sizes(n_chunks, 0);
  {
    std::vector<double> wts;
    if (const char* e = getenv("BP_CHUNK_WEIGHTS")) {
      for (const char* q = e; *q;) {
        char* end = nullptr;
        double v = strtod(q, &end);
        if (end == q) break;
        wts.push_back(v > 0 ? v : 0);
        q = *end == ',' ? end + 1 : end;
      }
    }
    double sum = 0;
    for (double v : wts) sum += v;
    size_t left = batch;
    for (uint32_t c = 0; c < n_chunks; c++) {
      size_t want = per;
      if (wts.size() == n_chunks && sum > 0) want = ((size_t)((double)batch * wts[c] / sum) + 255) / 256 * 256;
      if (want < 1) want = 1;
      const size_t others = n_chunks - 1 - c;  // every later chunk keeps at least one proof
      if (want + others > left) want = left - others;
      if (c + 1 == n_chunks) want = left;
      sizes[c] = want;
      left -= want;
    }
  }
  size_t first = 0, pinned_bytes = 0;
  for (uint32_t c = 0; c < n_chunks; c++) {
    Chunk& ch = job->chunks[c];
    ch.first = first;
    ch.B = (uint32_t)sizes[c];
    first += ch.B;
    int rc = vbatch_create_ex(ctx, n_bits, m, ch.B, true, &ch.vb);
    if (rc != BP_OK) {
      bp_vjob_destroy(job);
      return rc;
    }
    job->k = bp_vbatch_rounds(ch.vb);
    job->npts = bp_vbatch_point_rows(ch.vb);
    pinned_bytes += (size_t)ch.B * (5 * 32 + (size_t)job->npts * 64 + 4 * 32 + 32 + 64 + 1 + 160 +
                                    (size_t)2 * job->k * 160 + 11 * 80) + 9 * 256;
  }
  if (hipHostMalloc(&job->pinned, pinned_bytes, hipHostMallocDefault) != hipSuccess) {
    job->pinned = nullptr;
    bp_vjob_destroy(job);
    ctx->err = "pinned host allocation failed";
    return BP_ERR_NOMEM;
  }
  Carver cv(job->pinned);
  for (auto& ch : job->chunks) {
    size_t B = ch.B;
    ch.sc = cv.take<uint8_t>(5 * B * 32);
    ch.pts = cv.take<uint8_t>((size_t)job->npts * B * 64);
    ch.chal = cv.take<uint8_t>(4 * B * 32);
    ch.xk = cv.take<uint8_t>(B * 32);
    ch.acc = cv.take<uint8_t>(B * 64);
    ch.verdict = cv.take<uint8_t>(B);
    ch.acc_txt = cv.take<uint8_t>(B * 160);
    ch.lr_txt = cv.take<uint8_t>((size_t)2 * job->k * B * 160 + 16);
    ch.init_txt = cv.take<uint8_t>(B * 11 * 80);
    ch.vb->direct_host = true;  // acc / verdict above are pinned: the kernels store to them directly
    Pool* pool = job->pool;
    ch.vb->pfor = [pool](size_t n, const std::function<void(size_t, size_t)>& f) { pool->run(f, n, 4); };
    job->fsdev = job->fsdev && !ch.vb->split;
  }
  *out = job;
  return BP_OK;
}

// scalars: 5 rows x batch x 32 B, points: rows x batch x 64 B (layout of bp_vbatch_load over the WHOLE batch).
// Copies the proofs into pinned chunk-local arrays and uploads them to HBM.
int bp_vjob_load(bp_vjob* job, const uint8_t* scalars, const uint8_t* points) {
  if (!job) return BP_ERR_ARG;
  bp_ctx* ctx = job->ctx;
  BP_REQUIRE(scalars && points, "null pointer");
  if (job->in_flight) {
    ctx->err = "bp_vjob_load while a submitted pass is in flight (bp_vjob_wait first)";
    return BP_ERR_STATE;
  }
  BP_TRY(hipSetDevice(ctx->device));
  size_t Bt = job->batch;
  for (auto& ch : job->chunks) {
    size_t B = ch.B;
    for (uint32_t r = 0; r < 5; r++) memcpy(ch.sc + r * B * 32, scalars + (r * Bt + ch.first) * 32, B * 32);
    for (uint32_t r = 0; r < job->npts; r++) memcpy(ch.pts + r * B * 64, points + (r * Bt + ch.first) * 64, B * 64);
    int rc = vbatch_load_async(ch.vb, ch.sc, ch.pts);
    if (rc != BP_OK) return rc;
  }
  for (auto& ch : job->chunks) {
    int rc = vbatch_wait(ch.vb);
    if (rc != BP_OK) return rc;
  }
  job->loaded = true;
  return BP_OK;
}

// The timed path: transcript hashing on the host, everything else on the GPU.
int bp_vjob_run(bp_vjob* job, uint8_t* verdicts) {
  if (!job) return BP_ERR_ARG;
  bp_ctx* ctx = job->ctx;
  BP_REQUIRE(verdicts, "null pointer");
  if (!job->loaded) {
    ctx->err = "bp_vjob_run before bp_vjob_load";
    return BP_ERR_STATE;
  }
  if (job->in_flight) {
    ctx->err = "bp_vjob_run while a submitted pass is in flight (bp_vjob_wait first)";
    return BP_ERR_STATE;
  }
  std::string err;
  int rc = vjob_run_impl(job, verdicts, err);
  if (rc != BP_OK && !err.empty()) ctx->err = err;
  return rc;
}

// Two jobs in flight: bp_vjob_submit starts the pass on a driver thread owned by the job and returns; bp_vjob_wait
// joins it and hands out the verdicts.  While job A walks its last rounds and final multi-exponentiations (the GPU is
// no longer full: the ends of a pass are ~16 % of it) job B's first stages already run.  Each job keeps its own
// chunks, streams, pinned buffers and hashing pool; the two drivers only share the context's device and its
// immutable generator tables.  Kernel timing (bp_ctx_profile) must be off while passes overlap.
int bp_vjob_submit(bp_vjob* job) {
  if (!job) return BP_ERR_ARG;
  bp_ctx* ctx = job->ctx;
  if (!job->loaded || job->in_flight) {
    ctx->err = job->in_flight ? "bp_vjob_submit: the previous pass of this job has not been waited for"
                              : "bp_vjob_submit before bp_vjob_load";
    return BP_ERR_STATE;
  }
  if (ctx->profile) {
    ctx->err = "bp_vjob_submit: kernel timing (bp_ctx_profile) covers synchronous passes only";
    return BP_ERR_STATE;
  }
  if (job->driver.joinable()) job->driver.join();
  job->driver_verdicts.assign(job->batch, 0);
  job->driver_err.clear();
  job->in_flight = true;
  job->driver = std::thread([job] { job->driver_rc = vjob_run_impl(job, job->driver_verdicts.data(), job->driver_err); });
  return BP_OK;
}

int bp_vjob_wait(bp_vjob* job, uint8_t* verdicts) {
  if (!job) return BP_ERR_ARG;
  bp_ctx* ctx = job->ctx;
  BP_REQUIRE(verdicts, "null pointer");
  if (!job->in_flight) {
    ctx->err = "bp_vjob_wait without bp_vjob_submit";
    return BP_ERR_STATE;
  }
  job->driver.join();
  job->in_flight = false;
  if (job->driver_rc != BP_OK) {
    if (!job->driver_err.empty()) ctx->err = job->driver_err;
    return job->driver_rc;
  }
  memcpy(verdicts, job->driver_verdicts.data(), job->batch);
  return BP_OK;
}

}  // extern "C"

// error reporting of the pass: into `err` (a driver thread must not write the context's message under another
// call's feet); hipSetDevice is per thread
#define VJ_TRY(expr)                                                   \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) {                                            \
      err = std::string(#expr) + ": " + hipGetErrorString(e_);        \
      return BP_ERR_DEVICE;                                            \
    }                                                                  \
  } while (0)

static int vjob_run_impl(bp_vjob* job, uint8_t* verdicts, std::string& err) {
  bp_ctx* ctx = job->ctx;
  VJ_TRY(hipSetDevice(ctx->device));
  const ShowStyle st = job->style;
  const uint32_t k = job->k;
  int rc;
  // BP_TRACE=1: print a host-side timeline (ms since entry) of hashing / waiting per chunk
  static const bool trace = getenv("BP_TRACE") != nullptr;
  auto t_entry = std::chrono::steady_clock::now();
  auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(); };
#define TRACE(what, c, r, t0) do { if (trace) fprintf(stderr, "[vjob] %-8s chunk %d round %d  %8.3f -> %8.3f ms\n", what, (int)(c), (int)(r), t0, now_ms()); } while (0)
  // The transcript on the device (BP_FS_DEVICE=1 when the job was created): every chunk's pass is one in-order
  // sequence of kernels, challenges included, and the host only collects verdicts.  The latency path of small batches
  // keeps the host transcript (its chains run one lane per scalar bit and hand XYZZ sums to the host anyway).
  if (job->fsdev) {
    for (auto& ch : job->chunks)
      if ((rc = vbatch_run_fsdev_async(ch.vb, ch.verdict)) != BP_OK) {
        err = ctx->err;
        return rc;
      }
    for (auto& ch : job->chunks) {
      if ((rc = vbatch_wait(ch.vb)) != BP_OK) {
        err = ctx->err;
        return rc;
      }
      memcpy(verdicts + ch.first, ch.verdict, ch.B);
    }
    return BP_OK;
  }
  const ShaMid r_mid = sha_mid_show_r();  // every transcript opens with the first block of `show _r`
  // y, z, x (MultiRangeProof/Verifier.hs:31-33) and uChallenge (Verifier.hs:86)
  // Two proofs are hashed side by side (finish_fr_pair: interleaved SHA-256 chains); a lone last proof pairs
  // with itself in spare buffers.
  // BP_INIT_TEXT=1: the GPU formats A, S, T1, T2, tBlinding, mu, t as well (vbatch_init_text_async) and the host only
  // concatenates and hashes (0.95 instead of 1.55 ms per 16,384-proof chunk).  Off by default: the pass gets SLOWER,
  // 24.5-24.7 against 23.1-23.3 ms (profiles/r06_host_transcript_stagger_sweep.txt).  With the side work of all chunks
  // queued at the start of the pass the first chunk's text waits 3.3 ms behind 126 MB of L_r, R_r text on the host
  // link; with the side work started per chunk (what this switch does) the GPU idles while the host hashes the first
  // challenges; spacing the chunks by hand (BP_STAGGER_US) changes neither.
  static const bool init_text_env = getenv("BP_INIT_TEXT") && atoi(getenv("BP_INIT_TEXT")) != 0;
  auto hash0 = [&](Chunk& ch) {
    const size_t B = ch.B;
    const bool recs = init_text_env && !ch.vb->split;  // the kernels formatted the proof's own numbers (dec.h records)
    const uint8_t* ptxt = ch.init_txt;
    const uint8_t* stxt = ch.init_txt + 8 * B * 80;
    auto put_point = [&](Msg& m, uint32_t row, size_t p) {
      if (recs) m.point_rec(ptxt + ((size_t)row * B + p) * 160);
      else m.point(ch.pts + ((size_t)row * B + p) * 64);
    };
    auto put_scalar = [&](Msg& m, uint32_t row, size_t p) {
      if (recs) m.fr_rec(stxt + ((size_t)row * B + p) * 80);
      else m.fr(ch.sc + ((size_t)row * B + p) * 32);
    };
    job->pool->run(
        [&](size_t a, size_t b) {
          uint8_t buf0[MSG_SCRATCH], buf1[MSG_SCRATCH];
          uint8_t spare[4][32];
          for (size_t p = a; p < b; p += 2) {
            const bool two = p + 1 < b;
            const size_t q = two ? p + 1 : p;
            uint8_t* bufs[2] = {buf0, buf1};
            const size_t pr[2] = {p, q};
            uint8_t* y[2] = {ch.chal + (0 * B + p) * 32, two ? ch.chal + (0 * B + q) * 32 : spare[0]};
            uint8_t* z[2] = {ch.chal + (1 * B + p) * 32, two ? ch.chal + (1 * B + q) * 32 : spare[1]};
            uint8_t* x[2] = {ch.chal + (2 * B + p) * 32, two ? ch.chal + (2 * B + q) * 32 : spare[2]};
            uint8_t* u[2] = {ch.chal + (3 * B + p) * 32, two ? ch.chal + (3 * B + q) * 32 : spare[3]};
            // shamirY: show _r <> A <> S   (Utils.hs:145-148).  The text of show _r <> A <> S opens all three
            // messages: its full SHA blocks are absorbed once, and it stays in place in the buffer (padding
            // is only ever written behind it)
            Msg my[2] = {Msg(bufs[0], st), Msg(bufs[1], st)};
            ShaMid mid[2] = {r_mid, r_mid};
            size_t as_len[2];
            for (int i = 0; i < 2; i++) {
              my[i].r();
              put_point(my[i], VP_A, pr[i]);
              put_point(my[i], VP_S, pr[i]);
              as_len[i] = my[i].len;
              sha_mid_advance(mid[i], bufs[i], as_len[i]);
            }
            finish_fr_pair(my[0], mid[0], y[0], my[1], mid[1], y[1]);
            // shamirZ: ... <> show y   (Utils.hs:150-153)
            Msg mz[2] = {Msg(bufs[0], st), Msg(bufs[1], st)};
            for (int i = 0; i < 2; i++) {
              mz[i].len = as_len[i];
              mz[i].fr(y[i]);
            }
            finish_fr_pair(mz[0], mid[0], z[0], mz[1], mid[1], z[1]);
            // shamirX: show _r <> A <> S <> T1 <> T2 <> show y <> show z   (Utils.hs:155-165)
            Msg mx[2] = {Msg(bufs[0], st), Msg(bufs[1], st)};
            for (int i = 0; i < 2; i++) {
              mx[i].len = as_len[i];
              put_point(mx[i], VP_T1, pr[i]);
              put_point(mx[i], VP_T2, pr[i]);
              mx[i].fr(y[i]);
              mx[i].fr(z[i]);
            }
            finish_fr_pair(mx[0], mid[0], x[0], mx[1], mid[1], x[1]);
            // shamirU: show _r <> show tBlinding <> show mu <> show t   (Utils.hs:172-174)
            Msg mu[2] = {Msg(bufs[0], st), Msg(bufs[1], st)};
            for (int i = 0; i < 2; i++) {
              mu[i].r();
              put_scalar(mu[i], VS_TAU, pr[i]);
              put_scalar(mu[i], VS_MU, pr[i]);
              put_scalar(mu[i], VS_T, pr[i]);
            }
            finish_fr_pair(mu[0], r_mid, u[0], mu[1], r_mid, u[1]);
          }
        },
        B);
  };
  // mkChallenges (InnerProductProof/Verifier.hs:44-53): shamirX' acc L R = oracle (show _r <> L <> R <> acc)
  auto hashR = [&](Chunk& ch, uint32_t r) {
    const size_t B = ch.B;
    job->pool->run(
        [&](size_t a, size_t b) {
          uint8_t buf0[MSG_SCRATCH], buf1[MSG_SCRATCH];
          uint8_t spare[32];
          for (size_t p = a; p < b; p += 2) {
            const bool two = p + 1 < b;
            const size_t pr[2] = {p, two ? p + 1 : p};
            Msg mm[2] = {Msg(buf0, st), Msg(buf1, st)};
            for (int i = 0; i < 2; i++) {
              mm[i].r();
              if (ch.vb->split) {
                // latency path: the host formats the three points itself (the commitment was normalised by
                // vbatch_collect a moment ago)
                mm[i].point(ch.pts + ((size_t)(VP_L0 + r) * B + pr[i]) * 64);
                mm[i].point(ch.pts + ((size_t)(VP_L0 + k + r) * B + pr[i]) * 64);
                mm[i].point(ch.acc + pr[i] * 64);
              } else {
                // every number of this message was formatted on the GPU (dec.h records): L_r, R_r beside
                // stage 1, the running commitment by the kernel that produced it
                mm[i].point_rec(ch.lr_txt + ((size_t)r * B + pr[i]) * 160);
                mm[i].point_rec(ch.lr_txt + ((size_t)(k + r) * B + pr[i]) * 160);
                mm[i].point_rec(ch.acc_txt + pr[i] * 160);
              }
            }
            finish_fr_pair(mm[0], r_mid, ch.xk + p * 32, mm[1], r_mid, two ? ch.xk + (p + 1) * 32 : spare);
          }
        },
        B);
  };
  // Every chunk walks the stages 0 (hash y,z,x,u -> begin), 1..k (hash x_r -> round r-1), k+1 (-> finish),
  // k+2 (verdicts in).  The host always serves the READY chunk that is furthest along, so the chunks
  // spread out in time: while the host hashes later chunks the GPU already runs the final multiexp of
  // earlier ones, and nothing waits for a whole round to finish across chunks.
  const size_t nc = job->chunks.size();
  // challenge-independent GPU work of every chunk first (the rounds' joint tables, the text of L_r, R_r): it
  // fills the GPU while the host hashes the first chunk's y, z, x, u.  (Formatting A, S, T1, T2, tBlinding, mu, t
  // on the GPU as well was measured slower: the first chunk then waits for its text, 31.9 against 31.0 ms.)
  // ... and in front of it, on every chunk's own stream, the text of the numbers y, z, x, u hash from the proof itself
  if (init_text_env)
    for (auto& ch : job->chunks)
      if (!ch.vb->split && (rc = vbatch_init_text_async(ch.vb, ch.init_txt)) != BP_OK) return rc;
  // (with the init text on, a chunk's side work starts with its stage 1 instead - vbatch_begin_async queues it: the
  // text of L_r, R_r is 31 MB per 16,384-proof chunk over the host link, and the init text must not wait behind it)
  if (!init_text_env)
    for (auto& ch : job->chunks)
      if ((rc = vbatch_prepare_async(ch.vb, ch.lr_txt)) != BP_OK) return rc;
  static const int sched_bias = [] {
    const char* e = getenv("BP_SCHED_BIAS");
    return e ? atoi(e) : -1;
  }();
  // BP_STAGGER_US: chunk c does not begin before c x that many microseconds into the pass (diagnostic: how far apart
  // the chunks' phases should be; by default the time y, z, x, u take to hash spaces them)
  static const long stagger_us = getenv("BP_STAGGER_US") ? atol(getenv("BP_STAGGER_US")) : 0;
  std::vector<uint32_t> stage(nc, 0);
  size_t finished = 0;
  while (finished < nc) {
    int pick = -1;
    long pick_key = 0;
    for (size_t c = 0; c < nc; c++) {
      if (stage[c] > k + 1) continue;  // verdicts pending or done
      if (stage[c] > 0 && !vbatch_ready(job->chunks[c].vb)) continue;
      if (stage[c] == 0 && stagger_us > 0 && now_ms() * 1000.0 < (double)stagger_us * (double)c) continue;  // not yet
      // sched_bias < 0: the ready chunk that is furthest along; otherwise the least advanced one, with chunk c
      // handicapped by sched_bias * c stages (0: all chunks advance together; large: lowest index first)
      long key = sched_bias < 0 ? -(long)stage[c] : (long)stage[c] + (long)sched_bias * (long)c;
      if (pick < 0 || key < pick_key) {
        pick = (int)c;
        pick_key = key;
      }
    }
    if (pick < 0) {
      // nothing to hash right now: collect a chunk whose verdicts are in, else poll until some chunk's
      // GPU work completes (blocking on one particular stream could sit out a 6 ms final phase while a
      // 2 ms round of another chunk finishes)
      bool collected = false;
      for (size_t c = 0; c < nc; c++)
        if (stage[c] == k + 2 && vbatch_ready(job->chunks[c].vb)) {
          Chunk& ch = job->chunks[c];
          if ((rc = vbatch_wait(ch.vb)) != BP_OK) return rc;
          memcpy(verdicts + ch.first, ch.verdict, ch.B);
          stage[c] = k + 3;
          finished++;
          collected = true;
        }
      if (!collected) std::this_thread::yield();
      continue;
    }
    Chunk& ch = job->chunks[pick];
    double t_h = now_ms();
    if (stage[pick] == 0) {
      if (init_text_env && !ch.vb->split && (rc = vbatch_wait(ch.vb)) != BP_OK) return rc;  // the init text is in
      TRACE("inittxt", pick, -1, t_h);
      hash0(ch);
      TRACE("hash0", pick, -1, t_h);
      if ((rc = vbatch_begin_async(ch.vb, ch.chal, ch.acc, ch.acc_txt, ch.lr_txt)) != BP_OK) return rc;
    } else if (stage[pick] <= k) {
      if ((rc = vbatch_wait(ch.vb)) != BP_OK) return rc;  // completed already: folds the profiling events in
      TRACE("wait", pick, stage[pick] - 1, t_h);
      if ((rc = vbatch_collect(ch.vb, ch.acc)) != BP_OK) return rc;
      TRACE("collect", pick, stage[pick] - 1, t_h);
      uint32_t r = stage[pick] - 1;
      // round 0 hashes the text of L_0, R_0 that k_dec_numbers wrote on the SIDE stream: join it first
      if (r == 0 && (rc = vbatch_join_side(ch.vb)) != BP_OK) return rc;
      hashR(ch, r);
      TRACE("hashR", pick, r, t_h);
      if ((rc = vbatch_round_async(ch.vb, r, ch.xk, ch.acc, ch.acc_txt)) != BP_OK) return rc;
      TRACE("launch", pick, r, t_h);
    } else {
      if ((rc = vbatch_wait(ch.vb)) != BP_OK) return rc;
      if ((rc = vbatch_finish_async(ch.vb, ch.verdict)) != BP_OK) return rc;
      TRACE("finish", pick, -1, t_h);
    }
    stage[pick]++;
  }
#undef TRACE
  return BP_OK;
}

extern "C" {

// One job (workspaces in HBM, pinned staging, the hashing threads) stays cached on the context, so that repeated calls of
// one shape - the shim's single-proof verifyProof in a loop, or `map verifyProof` batch after batch - do not
// allocate and free pinned memory, device workspaces and a thread pool every time.  A call of another shape, new
// generators or another `show` style replaces it; bp_ctx_destroy frees it.  BP_NO_VJOB_CACHE=1 turns the cache off.
static void vjob_cache_free(void* j) { bp_vjob_destroy((bp_vjob*)j); }

int bp_rangeproof_verify_batch(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, const uint8_t* scalars,
                               const uint8_t* points, uint8_t* verdicts) {
  if (!ctx) return BP_ERR_ARG;
  static const bool no_cache = getenv("BP_NO_VJOB_CACHE") && atoi(getenv("BP_NO_VJOB_CACHE")) != 0;
  bp_vjob* job = (bp_vjob*)ctx->vjob_cache;
  if (job && (no_cache || job->n_bits != n_bits || job->m != m || job->batch != batch ||
              job->style != (ShowStyle)ctx->show_style || job->chunks[0].vb->epoch != ctx->gen_epoch)) {
    bp_vjob_destroy(job);
    job = nullptr;
    ctx->vjob_cache = nullptr;
  }
  int rc;
  if (!job) {
    rc = bp_vjob_create(ctx, n_bits, m, batch, 0, 0, &job);
    if (rc != BP_OK) return rc;
    if (!no_cache) {
      ctx->vjob_cache = job;
      ctx->vjob_cache_free = vjob_cache_free;
    }
  }
  rc = bp_vjob_load(job, scalars, points);
  if (rc == BP_OK) rc = bp_vjob_run(job, verdicts);
  if (no_cache) bp_vjob_destroy(job);
  return rc;
}

// n = logBase2 upperBound = floor (logBase 2.0 (fromIntegral upperBound))   (Utils.hs:93-94,
// MultiRangeProof/Verifier.hs:34): Double arithmetic, as in the reference.  upper_bound is a
// little-endian integer of `len` bytes.
int bp_log_base2(const uint8_t* upper_bound, size_t len) {
  double v = 0.0;
  for (size_t i = len; i-- > 0;) v = v * 256.0 + (double)upper_bound[i];  // exact prefix, then rounds like fromIntegral
  if (!(v > 0.0)) return -1;
  return (int)__builtin_floor(__builtin_log(v) / __builtin_log(2.0));
}

// Transcript self-test hook: which = 0 shamirY(A,S) 1 shamirZ(A,S,y) 2 shamirX(A,S,T1,T2,y,z)
// 3 shamirX'(acc,L,R) 4 shamirU(tau,mu,t) 5 oracle over raw bytes into Fr 6 oracle into Fq.
// pts: up to 4 points (64 B each), scs: up to 3 scalars (32 B each), raw/raw_len for 5, 6.
int bp_transcript(int style, int which, const uint8_t* pts, const uint8_t* scs, const uint8_t* raw, size_t raw_len,
                  uint8_t* out) {
  if (!out || style < 0 || style > 1) return BP_ERR_ARG;
  ShowStyle st = (ShowStyle)style;
  if (which == 9 && raw_len > 4096) return BP_ERR_ARG;
  std::vector<uint8_t> big((which == 9 ? raw_len * 170 : raw_len) + MSG_SCRATCH);
  Msg m(big.data(), st);
  switch (which) {
    case 0: m.r(); m.point(pts); m.point(pts + 64); break;
    case 1: m.r(); m.point(pts); m.point(pts + 64); m.fr(scs); break;
    case 2: m.r(); for (int i = 0; i < 4; i++) m.point(pts + 64 * i); m.fr(scs); m.fr(scs + 32); break;
    case 3: m.r(); m.point(pts + 64); m.point(pts + 128); m.point(pts); break;
    case 4: m.r(); m.fr(scs); m.fr(scs + 32); m.fr(scs + 64); break;
    case 5: m.raw(raw, raw_len); break;
    case 6: m.raw(raw, raw_len); oracle(m.buf, m.len, MOD_P, out); return BP_OK;
    case 7: m.r(); for (int i = 0; i < 3; i++) m.point(pts + 64 * i); break;
    case 8: m.r(); m.fr(scs); break;
    case 9: m.r(); for (size_t i = 0; i < raw_len; i++) m.point(pts + 64 * i); break;
    default: return BP_ERR_ARG;
  }
  m.finish_fr(out);
  return BP_OK;
}

// The reference's generator CAFs (Utils.hs:14-25,38-47): h = generateH "", gs_i / hs_i =
// gen * oracle(show i <> show gen / show h), i from 1.  Writes n_gens points each to gs_out,
// hs_out and one to h_out (any may be null), and installs them with bp_ctx_set_generators.
int bp_generators_derive(bp_ctx* ctx, uint32_t n_gens, uint8_t* gs_out, uint8_t* hs_out, uint8_t* h_out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(n_gens >= 1 && n_gens <= 8192 && (n_gens & (n_gens - 1)) == 0,
             "n_gens must be a power of two in 1..8192");
  ShowStyle st = (ShowStyle)ctx->show_style;
  affine G = affine_gen();
  uint8_t gen[64];
  memcpy(gen, &G, 64);
  // generateH (Utils.hs:38-47): x = oracle(show gen <> extra) in Fq, y = sr(x^3 + 7), retry with '1':extra
  uint8_t hpt[64];
  std::string extra;
  int rc;
  for (int attempt = 0;; attempt++) {
    if (attempt > 256) {
      ctx->err = "generateH did not terminate";
      return BP_ERR_STATE;
    }
    uint8_t buf[MSG_SCRATCH + 300];
    Msg m(buf, st);
    m.point(gen);
    m.raw(extra.data(), extra.size());
    uint8_t x[32], y[32], x2[32], x3[32], rhs[32], seven[32] = {7}, chk[32];
    oracle(m.buf, m.len, MOD_P, x);
    if ((rc = bp_field_op(ctx, 0, 3, x, x, 1, x2)) != BP_OK) return rc;
    if ((rc = bp_field_op(ctx, 0, 2, x2, x, 1, x3)) != BP_OK) return rc;
    if ((rc = bp_field_op(ctx, 0, 0, x3, seven, 1, rhs)) != BP_OK) return rc;
    if ((rc = bp_field_op(ctx, 0, 6, rhs, rhs, 1, y)) != BP_OK) return rc;  // sr: (p+1)/4 power
    if ((rc = bp_field_op(ctx, 0, 3, y, y, 1, chk)) != BP_OK) return rc;
    if (memcmp(chk, rhs, 32) == 0) {
      if (ctx->sqrt_branch == 1) {  // U2 alternative: the other root
        uint8_t ny[32];
        if ((rc = bp_field_op(ctx, 0, 5, y, y, 1, ny)) != BP_OK) return rc;
        memcpy(y, ny, 32);
      }
      memcpy(hpt, x, 32);
      memcpy(hpt + 32, y, 32);
      break;
    }
    extra = "1" + extra;
  }
  // gs, hs
  std::vector<uint8_t> sc((size_t)2 * n_gens * 32), base((size_t)2 * n_gens * 64), out((size_t)2 * n_gens * 64);
  for (uint32_t i = 0; i < n_gens; i++) {
    uint8_t buf[MSG_SCRATCH];
    char num[16];
    int nl = snprintf(num, sizeof num, "%u", i + 1);
    Msg mg(buf, st);
    mg.raw(num, (size_t)nl);
    mg.point(gen);
    mg.finish_fr(&sc[(size_t)i * 32]);
    Msg mh(buf, st);
    mh.raw(num, (size_t)nl);
    mh.point(hpt);
    mh.finish_fr(&sc[(size_t)(n_gens + i) * 32]);
    memcpy(&base[(size_t)i * 64], gen, 64);
    memcpy(&base[(size_t)(n_gens + i) * 64], gen, 64);
  }
  if ((rc = bp_point_mul(ctx, sc.data(), base.data(), (size_t)2 * n_gens, out.data())) != BP_OK) return rc;
  if (gs_out) memcpy(gs_out, out.data(), (size_t)n_gens * 64);
  if (hs_out) memcpy(hs_out, out.data() + (size_t)n_gens * 64, (size_t)n_gens * 64);
  if (h_out) memcpy(h_out, hpt, 64);
  return bp_ctx_set_generators(ctx, out.data(), out.data() + (size_t)n_gens * 64, hpt, n_gens);
}

// The generator CAFs as a file: `h`, `take n gs`, `take n hs` written once and installed by later
// processes without the hashing and the 2n scalar multiplications of bp_generators_derive.
// Layout: "BPGENS01", u32 n_gens, u32 show style, u32 sqrt branch, u32 0, gs (n x 64 B), hs (n x 64 B), h (64 B),
// SHA-256 of everything before it.  The style / branch the table was derived under are recorded because h, and with it
// every hs_i, depends on them (Utils.hs:38-47); loading sets the context to the recorded values.
int bp_generators_save(bp_ctx* ctx, const char* path) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(path, "null pointer");
  if (ctx->n_gens == 0) {
    ctx->err = "generators not set (bp_generators_derive)";
    return BP_ERR_STATE;
  }
  BP_TRY(hipSetDevice(ctx->device));
  const size_t n = ctx->n_gens, body = 24 + (2 * n + 1) * 64;
  std::vector<uint8_t> buf(body + 32 + 72);
  memcpy(buf.data(), "BPGENS01", 8);
  uint32_t hdr[4] = {(uint32_t)n, (uint32_t)ctx->show_style, (uint32_t)ctx->sqrt_branch, 0u};
  memcpy(buf.data() + 8, hdr, 16);
  // device layout of the bases: gs, hs, G, h (gens.hip)
  BP_TRY(hipMemcpy(buf.data() + 24, ctx->d_bases, 2 * n * 64, hipMemcpyDeviceToHost));
  BP_TRY(hipMemcpy(buf.data() + 24 + 2 * n * 64, ctx->d_bases + (2 * n + 1) * 64, 64, hipMemcpyDeviceToHost));
  uint8_t dg[32];
  std::vector<uint8_t> tmp(buf.begin(), buf.begin() + body);
  tmp.resize(body + 72);
  sha256_inplace(tmp.data(), body, dg);
  memcpy(buf.data() + body, dg, 32);
  FILE* f = fopen(path, "wb");
  if (!f) {
    ctx->err = std::string("cannot open ") + path + " for writing";
    return BP_ERR_ARG;
  }
  bool ok = fwrite(buf.data(), 1, body + 32, f) == body + 32;
  ok = (fclose(f) == 0) && ok;
  if (!ok) {
    ctx->err = std::string("short write to ") + path;
    return BP_ERR_STATE;
  }
  return BP_OK;
}

int bp_generators_load(bp_ctx* ctx, const char* path) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(path, "null pointer");
  FILE* f = fopen(path, "rb");
  if (!f) {
    ctx->err = std::string("cannot open ") + path;
    return BP_ERR_ARG;
  }
  uint8_t head[24];
  uint32_t hdr[4] = {0, 0, 0, 0};
  bool ok = fread(head, 1, 24, f) == 24 && memcmp(head, "BPGENS01", 8) == 0;
  if (ok) memcpy(hdr, head + 8, 16);
  const size_t n = hdr[0];
  ok = ok && n >= 1 && n <= 8192 && (n & (n - 1)) == 0 && hdr[1] <= 1 && hdr[2] <= 1;
  std::vector<uint8_t> buf;
  const size_t body = 24 + (2 * n + 1) * 64;
  if (ok) {
    buf.resize(body + 72);
    memcpy(buf.data(), head, 24);
    uint8_t want[32], got[32], extra;
    ok = fread(buf.data() + 24, 1, body - 24, f) == body - 24 && fread(want, 1, 32, f) == 32 && fread(&extra, 1, 1, f) == 0;
    if (ok) {
      std::vector<uint8_t> tmp(buf);
      sha256_inplace(tmp.data(), body, got);
      ok = memcmp(want, got, 32) == 0;
    }
  }
  fclose(f);
  if (!ok) {
    ctx->err = std::string(path) + " is not a generator table (bad magic, size or checksum)";
    return BP_ERR_ARG;
  }
  // every entry must be a finite curve point: the table replaces group elements the reference computes itself
  std::vector<uint8_t> oc(2 * n + 1);
  int rc = bp_point_on_curve(ctx, buf.data() + 24, 2 * n + 1, oc.data());
  if (rc != BP_OK) return rc;
  for (size_t i = 0; i < 2 * n + 1; i++) {
    bool inf = true;
    for (int b = 0; b < 64; b++) inf = inf && buf[24 + i * 64 + b] == 0;
    if (!oc[i] || inf) {
      ctx->err = std::string(path) + ": entry is not a finite point of the curve";
      return BP_ERR_ARG;
    }
  }
  ctx->show_style = (int)hdr[1];
  ctx->sqrt_branch = (int)hdr[2];
  return bp_ctx_set_generators(ctx, buf.data() + 24, buf.data() + 24 + n * 64, buf.data() + 24 + 2 * n * 64, (uint32_t)n);
}

}  // extern "C"
