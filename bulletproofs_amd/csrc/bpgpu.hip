// C ABI of the MI355X backend (include/bpgpu.h).  Host orchestration of the kernels in
// kernels.cuh plus the short serial tails (Horner combine over window sums,
// normalisation, joint-sparse-form recoding) in host_field.h.
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <memory>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <thread>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bpgpu.h"
#include "host_field.h"
#include "kernels.cuh"

using namespace bpgpu;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(BPGPU_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define TRY(expr)          \
  do {                     \
    int r_ = (expr);       \
    if (r_ != 0) return r_; \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) HIP_TRY(hipHostFree(p));
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// N persistent worker threads; run(f) executes f(w) on every worker w and returns when all are done.
// (The staging copies of a large upload and the per-GPU legs of a sharded MSM: a fresh std::thread per
// call costs tens of microseconds each and a new HIP thread context.)
class Pool {
 public:
  ~Pool() { stop(); }
  // A pool may be started again after stop() (bpgpu_shutdown followed by bpgpu_init): the generation counter
  // survives, so every new worker starts from the CURRENT generation -- it must never run the job of an
  // earlier run, whose captures point into a frame that is gone -- and the job is dropped once its run ends.
  void start(int n) {
    if (!th_.empty()) return;
    uint64_t gen0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = false;
      n_ = n;
      pending_ = 0;
      job_ = nullptr;
      gen0 = gen_;
    }
    for (int w = 0; w < n; w++)
      th_.emplace_back([this, w, gen0] {
        uint64_t seen = gen0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
          cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
          if (quit_) return;
          seen = gen_;
          const std::function<void(int)>* job = &job_;  // stays valid until pending_ reaches 0
          lk.unlock();
          (*job)(w);
          lk.lock();
          if (--pending_ == 0) {
            job_ = nullptr;
            done_.notify_all();
          }
        }
      });
  }
  void stop() {
    if (th_.empty()) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
    th_.clear();
    std::lock_guard<std::mutex> lk(mu_);
    pending_ = 0;
    job_ = nullptr;
  }
  int size() const { return n_; }
  // one run at a time (callers hold the owning context's mutex, or the registry's for the shard pool).
  // begin() starts f(w) on every worker and returns; wait() blocks until all are done; run() = both.
  void begin(const std::function<void(int)>& f) {
    std::lock_guard<std::mutex> lk(mu_);
    job_ = f;
    pending_ = n_;
    gen_++;
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return pending_ == 0; });
  }
  void run(const std::function<void(int)>& f) {
    begin(f);
    wait();
  }

 private:
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::function<void(int)> job_;
  uint64_t gen_ = 0;
  int pending_ = 0, n_ = 0;
  bool quit_ = false;
};

// Host threads a context may keep beside its caller (Horner workers, staging threads): three where the host has
// the cores, fewer when many contexts (bpgpu_sharded_init) or many processes (one rank per GPU: LOCAL_WORLD_SIZE,
// set by torch.distributed.run) share it -- max(1, min(3, cores / (2 P))), P = contexts on this host.  The core
// count is BPGPU_HOST_CORES when set (a container's CPU share is not what the OS reports), else what the OS reports.
std::atomic<int> g_ctx_count{1};  // contexts of this process (primary + shards), for the budget
int helper_threads() {
  long cores = (long)std::thread::hardware_concurrency();
  if (const char* e = getenv("BPGPU_HOST_CORES")) {
    const long v = atol(e);
    if (v > 0) cores = v;
  }
  if (cores < 1) cores = 1;
  long procs = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) {
    const long v = atol(e);
    if (v > 0) procs = v;
  }
  long P = (long)g_ctx_count.load() * procs;
  if (P < 1) P = 1;
  long t = cores / (2 * P);
  if (t > 3) t = 3;
  if (t < 1) t = 1;
  return (int)t;
}

// Large host-to-device copies.  hipMemcpyAsync from pageable memory moves 29 GB/s on the MI355X box, from
// pinned memory 57 GB/s (tools/h2d_probe.py), and one host core copies 32 GB/s: three persistent worker
// threads stage 4 MiB chunks into pinned buffers, each chunk's DMA following on the worker's own stream.
struct Uploader {
  static constexpr int T = 8, SLOTS = 2;  // T: the most staging threads; nt of them run (staging_threads())
  static constexpr size_t CH = 4u << 20, MIN_BYTES = 8u << 20;
  bool ready = false;
  int nt = T;
  void* buf[T][SLOTS] = {};
  hipStream_t ws[T] = {};
  hipStream_t cs[T] = {};  // point uploads: the chunks' conversion kernels (ws carries the DMAs only, see pt_upload_begin)
  hipEvent_t ev[T][SLOTS] = {};
  hipEvent_t fence = nullptr, done[T] = {};
  hipEvent_t pdone[2][T] = {};  // point uploads (PtUpload): worker w has enqueued every chunk of phase p
  Pool pool;
  hipError_t init() {
    if (ready) return hipSuccess;
    nt = helper_threads();
    if (const char* ev_nt = getenv("BPGPU_STAGING_THREADS")) {  // tuning / experiments
      const int v = atoi(ev_nt);
      if (v >= 1 && v <= T) nt = v;
    }
    hipError_t e = hipEventCreateWithFlags(&fence, hipEventDisableTiming);
    for (int w = 0; w < nt && e == hipSuccess; w++) {
      e = hipStreamCreateWithFlags(&ws[w], hipStreamNonBlocking);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs[w], hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&done[w], hipEventDisableTiming);
      for (int ph = 0; ph < 2 && e == hipSuccess; ph++) e = hipEventCreateWithFlags(&pdone[ph][w], hipEventDisableTiming);
      for (int k = 0; k < SLOTS && e == hipSuccess; k++) {
        e = hipHostMalloc(&buf[w][k], CH, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[w][k], hipEventDisableTiming);
      }
    }
    if (e == hipSuccess) pool.start(nt);
    ready = e == hipSuccess;
    if (!ready) release();  // a half-built uploader keeps nothing
    return e;
  }
  void release() {
    pool.stop();
    for (int w = 0; w < T; w++) {
      if (ws[w]) (void)hipStreamSynchronize(ws[w]);
      if (cs[w]) (void)hipStreamSynchronize(cs[w]);
      for (int k = 0; k < SLOTS; k++) {
        if (buf[w][k]) (void)hipHostFree(buf[w][k]);
        if (ev[w][k]) (void)hipEventDestroy(ev[w][k]);
        buf[w][k] = nullptr;
        ev[w][k] = nullptr;
      }
      if (done[w]) (void)hipEventDestroy(done[w]);
      for (int ph = 0; ph < 2; ph++) {
        if (pdone[ph][w]) (void)hipEventDestroy(pdone[ph][w]);
        pdone[ph][w] = nullptr;
      }
      if (ws[w]) (void)hipStreamDestroy(ws[w]);
      if (cs[w]) (void)hipStreamDestroy(cs[w]);
      done[w] = nullptr;
      ws[w] = nullptr;
      cs[w] = nullptr;
    }
    if (fence) (void)hipEventDestroy(fence);
    fence = nullptr;
    ready = false;
  }
};

// One context = one HIP device, one stream, one grow-only workspace, one mutex.  The single-device entry
// points use the calling thread's current context: the primary one (bpgpu_init) unless the thread is a
// shard worker or is inside a call on an IPA session, which carries its own context.  Several contexts may
// sit on one physical GPU (bpgpu_sharded_init with repeated device numbers: rehearsal of the multi-GPU path).
struct Ctx {
  std::mutex mu;
  bool ready = false;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};
  hipEvent_t op_ev[2] = {};  // around the last ip / fold / vector-utility operation (bpgpu_last_op_ms)
  int op_depth = 0;          // OpSpan nesting: only the outermost span records the events
  int num_cus = 256;
  float msm_ms[8] = {};
  bool msm_timed = false;  // ev[0..5] bracket the phases of a completed MSM
  uint64_t msm_entries = 0;  // non-zero digits of the last MSM batch = bucket-accumulation additions
  uint64_t fold_ops[4] = {};  // last generator-fold launch: outputs, doublings, mixed additions (all lanes), kernel kind
  int open_sessions = 0;     // IPA sessions that still point at this context
  uint32_t* noalloc_scratch = nullptr;  // this device's copy of the file-scope scratch array (bpgpu_ip_device_noalloc)
  // MSM workspace (grow-only)
  DevBuf counts, offs, order, cursor, canon, ent_idx, ent_fine, sorted, bsum, partial, winsum, heavy, bcount, redtmp;
  // staging for host-buffer entry points
  DevBuf st_pts_in, st_pts, st_sc, st_out, st_ip;
  DevBuf aux[6];  // sparse products (row pointers, indices, permutation, values, vector, result), point sampling
  Pool hpool;  // up to three host threads (helper_threads()) that share the Horner tail with the caller's thread
  PinnedBuf h_winsum;  // window sums land here (pinned: the copy is a plain DMA)
  Uploader up;
  // chunked point upload in flight (pt_upload_begin .. pt_upload_end)
  struct PtChunk {
    const uint64_t* src;
    uint32_t count;
    int fmt;
    uint32_t* d_dst;
    uint8_t* d_stage;
    int phase;
  };
  std::vector<PtChunk> pt_chunks;
  // phases whose conversion is deferred to pt_upload_wait (one kernel per segment on the waiting stream instead of one per
  // chunk behind its DMA): the segments, and the per-chunk formats the staging threads report through pinned memory
  struct PtDeferred {
    const uint32_t* d_stage;
    uint32_t* d_dst;
    uint32_t n;
    size_t first_chunk;
    int phase;
  };
  std::vector<PtDeferred> pt_deferred;
  int pt_defer_mask = 0;
  int pt_curve = 0;
  PinnedBuf pt_fmt;
  std::atomic<int> pt_phase_count[2];
  hipError_t pt_err[Uploader::T] = {};
  bool pt_active = false;
  // the buffers of the last closed IPA session, kept for the next one (a prover makes proof after proof of
  // one size; hipMalloc + hipFree of 460 MB cost about a millisecond per proof)
  DevBuf ipa_pts, ipa_sc;
  DevBuf ipa_table, ipa_table_tmp;  // fixed-base table of the pooled session's frozen generators (and its XYZZ staging)
  DevBuf fold_alpha;                // k_fold_multi: the alpha-multiples of a launch's generators
  bool ipa_pool_busy = false;
};

Ctx g_primary;                              // bpgpu_init
std::vector<std::unique_ptr<Ctx>> g_shards;  // bpgpu_sharded_init, one per (logical) GPU of the sharded MSM
Pool g_shard_pool;                          // one persistent host thread per shard context
// One sharded call at a time is the contract: the legs of a call occupy every shard context and the one shard pool
// (a second call could only queue behind them), so g_shard_run_mu is held for the whole call -- and by
// bpgpu_sharded_init / bpgpu_shutdown, which therefore wait for a running call.  g_reg_mu only guards the vector
// itself and is never held across device work (bpgpu_sharded_count does not wait for an MSM).
std::mutex g_shard_run_mu;
std::mutex g_reg_mu;
thread_local Ctx* tl_ctx = nullptr;
inline Ctx& cx() { return tl_ctx ? *tl_ctx : g_primary; }

// Tuning knobs are process-wide (documented so in bpgpu.h): results never depend on them.
constexpr size_t DEFAULT_FREEZE_LEN = 1u << 14, DEFAULT_IPA_TABLE_MIN = 1u << 12;
std::atomic<int> g_window_bits{0};
std::atomic<size_t> g_freeze_len{DEFAULT_FREEZE_LEN};  // vectors this short defer the generator fold (0 = never)
std::atomic<bool> g_pair_fold{true};         // long rounds fold their generators two rounds at a time (k_fold_joint4)
std::atomic<size_t> g_ipa_table_min{DEFAULT_IPA_TABLE_MIN};  // frozen generators at least this many get a fixed-base table (0 = never)

// Serialises the calls on the current context; with an explicit context (an IPA session's) that context is
// also made current for the scope.
struct CtxLock {
  Ctx* prev;
  std::unique_lock<std::mutex> lk;
  CtxLock() : prev(tl_ctx), lk(cx().mu) {}
  explicit CtxLock(Ctx* c) : prev(tl_ctx) {
    tl_ctx = c;
    lk = std::unique_lock<std::mutex>(c->mu);
  }
  ~CtxLock() {
    lk.unlock();
    tl_ctx = prev;
  }
};

// Brackets the kernels of one operation with the context's op events (bpgpu_last_op_ms).  Spans nest: an operation made
// of several launches (the k-ary generator fold: k - 1 monic folds) is timed as a whole by its outermost span.
struct OpSpan {
  Ctx& C;
  explicit OpSpan(Ctx& c) : C(c) {
    if (C.op_depth++ == 0) (void)hipEventRecord(C.op_ev[0], C.stream);
  }
  ~OpSpan() {
    if (--C.op_depth == 0) (void)hipEventRecord(C.op_ev[1], C.stream);
  }
};

// Keeps the spans inside its scope from recording (the prover's short rounds: four event packets a round for timers
// nobody reads there)
struct OpMute {
  Ctx& C;
  bool on;
  OpMute(Ctx& c, bool mute) : C(c), on(mute) {
    if (on) C.op_depth++;
  }
  ~OpMute() {
    if (on) C.op_depth--;
  }
};

inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

int check_ready() {
  if (!cx().ready) return fail(BPGPU_E_NOINIT, "bpgpu_init has not been called");
  HIP_TRY(hipSetDevice(cx().device));  // the current device is per host thread
  return 0;
}
int check_curve(int curve) {
  if (curve != BPGPU_PALLAS && curve != BPGPU_VESTA) return fail(BPGPU_E_ARG, "unknown curve id %d", curve);
  return 0;
}

// hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) for a pageable src: ordered after the work
// already on st and before the work that follows on it.  Returns once src has been read.
hipError_t upload_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
  Uploader& U = cx().up;
  if (bytes < Uploader::MIN_BYTES) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  hipError_t e = U.init();
  if (e != hipSuccess) return e;
  if ((e = hipEventRecord(U.fence, st)) != hipSuccess) return e;
  const size_t nch = (bytes + Uploader::CH - 1) / Uploader::CH;
  const int device = cx().device;
  const int nt = U.nt;
  hipError_t err[Uploader::T];
  U.pool.run([&](int w) {
    hipError_t r = hipSetDevice(device);
    if (r == hipSuccess) r = hipStreamWaitEvent(U.ws[w], U.fence, 0);
    for (size_t k = (size_t)w; k < nch && r == hipSuccess; k += (size_t)nt) {
      const int slot = (int)((k / (size_t)nt) % Uploader::SLOTS);
      const size_t off = k * Uploader::CH, len = bytes - off < Uploader::CH ? bytes - off : Uploader::CH;
      if ((r = hipEventSynchronize(U.ev[w][slot])) != hipSuccess) break;  // the slot's previous DMA has drained
      memcpy(U.buf[w][slot], (const char*)src + off, len);
      if ((r = hipMemcpyAsync((char*)dst + off, U.buf[w][slot], len, hipMemcpyHostToDevice, U.ws[w])) != hipSuccess) break;
      r = hipEventRecord(U.ev[w][slot], U.ws[w]);
    }
    if (r == hipSuccess) r = hipEventRecord(U.done[w], U.ws[w]);
    err[w] = r;
  });
  for (int w = 0; w < nt; w++) {
    if (err[w] != hipSuccess) return err[w];
    if ((e = hipStreamWaitEvent(st, U.done[w], 0)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---- host tail helpers ------------------------------------------------------
template <int BF>
host::HPoint<BF> hpoint_from_words(const uint32_t* w) {
  host::HPoint<BF> p;
  memcpy(p.x.v, w, 32);
  memcpy(p.y.v, w + 8, 32);
  memcpy(p.zz.v, w + 16, 32);
  memcpy(p.zzz.v, w + 24, 32);
  p.inf = host::HF<BF>::is_zero(p.zz);
  return p;
}

template <int BF>
void write_xy_inf(const host::HPoint<BF>& p, uint64_t out[9]) {
  host::H4 x, y;
  bool finite = host::HEc<BF>::to_affine(p, x, y);
  memcpy(out, x.v, 32);
  memcpy(out + 4, y.v, 32);
  out[8] = finite ? 0 : 1;
}

// The same for the M results of a batch with ONE field inversion (Montgomery's trick): a host inversion is a Fermat ladder
// of ~320 multiplications (8 us), as much as the rest of a short round's tail, and a k-ary round normalises up to 32 points.
template <int BF>
void write_xy_inf_batch(const host::HPoint<BF>* p, int M, uint64_t (*out)[9]) {
  using F = host::HF<BF>;
  host::H4 z[MSM_MAX_BATCH], pre[MSM_MAX_BATCH];
  host::H4 run = F::one();
  for (int m = 0; m < M; m++) {
    const bool finite = !p[m].inf && !F::is_zero(p[m].zz);
    z[m] = finite ? F::mul(p[m].zz, p[m].zzz) : F::one();
    pre[m] = run;
    run = F::mul(run, z[m]);
  }
  host::H4 inv = F::inv(run);
  for (int m = M - 1; m >= 0; m--) {
    const host::H4 zi = F::mul(inv, pre[m]);  // 1 / (zz zzz) of result m
    inv = F::mul(inv, z[m]);
    const bool finite = !p[m].inf && !F::is_zero(p[m].zz);
    host::H4 x = F::zero(), y = F::zero();
    if (finite) {
      x = F::mul(p[m].x, F::mul(zi, p[m].zzz));
      y = F::mul(p[m].y, F::mul(zi, p[m].zz));
    }
    memcpy(out[m], x.v, 32);
    memcpy(out[m] + 4, y.v, 32);
    out[m][8] = finite ? 0 : 1;
  }
}

// Horner combine (msm.rs:177-190), high bit position to low with one doubling per position.
// Window w contributes its component P at position c w and U_i at position c w + s + i
// (kernels.cuh, bucket reduction); comps = 2 + nbits points per window (P comes in two halves), ark-form XYZZ words.
// Only the windows in [w_lo, w_hi) contribute (the doublings run down to position 0 all the same), so that the
// shares of several host threads simply add up.
template <int BF>
host::HPoint<BF> horner(const uint32_t* winsums, int W, int c, int s, int nbits, int w_lo = 0, int w_hi = 1 << 30) {
  using E = host::HEc<BF>;
  const int comps = 2 + nbits;
  host::HPoint<BF> total = E::identity();
  if (w_hi > W) w_hi = W;
  if (w_lo >= w_hi) return total;
  const int top = c * (w_hi - 1) + s + (nbits > 0 ? nbits - 1 : 0);
  for (int pos = top; pos >= 0; pos--) {
    total = E::dbl(total);
    int w = pos / c, off = pos - w * c;
    if (w < w_lo) continue;  // every component of window w sits in [c w, c w + c - 2]: the filter is exact
    if (off == 0) {
      total = E::add(total, hpoint_from_words<BF>(winsums + 32 * (w * comps)));
      total = E::add(total, hpoint_from_words<BF>(winsums + 32 * (w * comps + 1)));
    }
    // U_i of window w sits at off = s + i; with s = 0 and i = 0 it shares position c w with P
    int i = off - s;
    if (i >= 0 && i < nbits) total = E::add(total, hpoint_from_words<BF>(winsums + 32 * (w * comps + 2 + i)));
  }
  return total;
}

// Window width.  Signed c-bit digits give 2^(c-1) buckets per window and W = ceil(256/c) windows.
// Measured on MI355X (tools/wsweep.sh, tools/msm_var_probe.py, tools/ipa_rounds.py; tables in
// profiles/r01_window_sweep.txt):
//   * 2^16 .. 2^24: the bucket accumulation ends in a tail as long as one of its work items (64 buckets,
//     so about one bucket's list): wider windows shorten it, until the bucket reduction (whose additions
//     grow with the bucket count) costs more than the tail saves.  c = 15 and c = 17 also leave the top
//     window empty for scalars below 2^254 (255 = 15 x 17 and bit 254 of a Pasta scalar is clear).
//   * below 2^16 an MSM is a chain of latencies: short lists (about one entry per two buckets) win until
//     the bucket reduction, twice the work in a batch of two, pushes back (single MSM of 2^15: c = 15
//     0.38 ms, c = 13 0.45 ms; the prover's short rounds, two MSMs of 2^15 non-zero scalars: c = 13 0.65 ms,
//     c = 15 0.76 ms per round).
//   * outside the measured range: about 32 points per bucket (c = log2 n - 4), at most 20 bits.
int choose_window(uint32_t n, int M) {
  if (const int forced = g_window_bits.load()) return forced;
  int lg = 0;
  while ((1ull << (lg + 1)) <= n) lg++;
  static const int8_t measured[9] = {15, 15, 16, 16, 17, 17, 17, 19, 20};
  if (lg == 18 && M > 1) return 15;  // a batch of two doubles the buckets to reduce: one bit narrower (1.22 -> 1.16 ms)
  if (lg >= 16 && lg <= 24) return measured[lg - 16];
  if (lg == 14 && M == 1) return 13;  // 14 leaves a three-bit top window: 0.51 ms against 0.44 at 13 (and 0.45 at 15)
  if (lg >= 5 && lg <= 15) return M > 1 ? lg - 2 : lg;
  int c = lg - 4;
  if (c < 2) c = 2;
  if (c > 20) c = 20;
  return c;
}

// Batched MSM over device-resident operands.  d_points: dev affine; d_scalars: ark form.
// out[m] receives xy_inf.
// points_ready (optional): called once the sort has been launched and before the accumulation is; it must put whatever
// makes d_points valid onto the context's stream (the host-buffer entry points upload the points there: the sort needs
// the scalars only, so it runs while the points are still crossing PCIe).
template <int CURVE>
int msm_core(const uint32_t* d_points, const uint32_t* d_scalars, MsmPlan plan, uint64_t (*out)[9],
             const std::function<int()>* points_ready = nullptr) {
  constexpr int BF = CURVE, SF = 1 - CURVE;
  Ctx& C = cx();
  for (int m = 0; m < plan.M; m++) {
    memset(out[m], 0, 72);
    out[m][8] = 1;
  }
  if (plan.n == 0) return 0;
  if (plan.M < 1 || plan.M > MSM_MAX_BATCH) return fail(BPGPU_E_ARG, "a batch holds 1 to %d MSMs", MSM_MAX_BATCH);
  // fixed-base table mode: the width is the one the table was built for, and all windows of an MSM share a bucket set
  const bool table = plan.table_stride != 0;
  const int c = table ? plan.c : choose_window(plan.n_eff ? plan.n_eff : plan.n, plan.M);
  const int W = (256 + c - 1) / c;
  const int Wb = table ? 1 : W;  // bucket sets per MSM
  plan.c = c;
  plan.W = W;
  // the sort keeps entry positions, bin and bucket offsets in 32 bits, and an entry is index | sign << 31
  if ((uint64_t)plan.M * plan.n * (uint64_t)W >= (1ull << 32))
    return fail(BPGPU_E_ARG, "MSM of %d x %u points in %d windows has 2^32 or more (point, window) pairs", plan.M, plan.n, W);
  for (int m = 0; m < plan.M; m++)
    for (int g = 0; g < 3; g++)
      if ((uint64_t)plan.seg[m][g].pt_off + plan.seg[m][g].count + (uint64_t)plan.table_stride * (W - 1) > (1ull << 31))
        return fail(BPGPU_E_ARG, "point index beyond 2^31");
  const uint32_t B = 1u << (c - 1);
  const uint32_t NB = (uint32_t)plan.M * Wb * B;
  const uint64_t slots = (uint64_t)plan.M * plan.n;
  // bucket reduction: runs of S = 2^s buckets, at most 8192 runs per window.  Both stages are bound by the
  // number of additions they issue (a lone wave already fills its SIMD's issue slots), so longer runs pay
  // once the second stage's share is large (profiles/r01_window_sweep.txt, second table)
  int s = c - 1 < 3 ? c - 1 : 3;
  if (NB >= (800u << 10) && c - 1 >= 4) s = 4;
  // two million buckets and more (a batch of two 2^20-slot MSMs): runs of 32 put the first stage on one wave per SIMD
  // again and halve the second stage's lists (reduce 0.73 -> 0.66 ms); no difference at 2^24, slower below
  if (NB >= (1600u << 10) && c - 1 >= 5) s = 5;
  if (c - 1 - s > 13) s = c - 1 - 13;
  if (table && c - 1 >= 2) s = 2;  // few buckets: the reduction is a chain of additions, keep it short
  const uint32_t nruns = NB >> s;
  const uint32_t runs_per_window = B >> s;
  const int nbits = c - 1 - s;  // bits of a window-local run index
  const int comps = 2 + nbits;

  // two-level counting sort: coarse bins of 2^fine_bits buckets
  SortPlan sp;
  sp.fine_bits = c - 1 < 10 ? c - 1 : 10;
  // with few buckets in all keep at least 64 bins, so that the LDS histograms do not pile onto a handful of counters
  while (sp.fine_bits > 0 && (NB >> sp.fine_bits) < 64) sp.fine_bits--;
  sp.nbins = NB >> sp.fine_bits;
  sp.tiles = cdiv(slots, 1024);
  const uint32_t part_grid = sp.tiles < 512 ? sp.tiles : 512;
  const size_t lds_bytes = (size_t)sp.nbins * 4;
  if (lds_bytes > 64 * 1024) return fail(BPGPU_E_ARG, "window width %d needs %zu B of LDS bin counters", c, lds_bytes);

  // bin_count | chunk counter | tickets | bucket counts | bin_cursor: the first four are zeroed by ONE fill per MSM
  TRY(C.counts.ensure((size_t)sp.nbins * BIN_PAD * 4 * 2 + 64 + 4096 + (size_t)NB * 4));
  TRY(C.cursor.ensure((size_t)(sp.nbins + 1) * 4 * 2));      // bin_start | tile_start
  TRY(C.offs.ensure((size_t)(NB + 1) * 4));
  TRY(C.order.ensure((size_t)NB * 4));
  TRY(C.canon.ensure((size_t)slots * W * 4));  // window-major digit words
  TRY(C.ent_idx.ensure((size_t)slots * W * 4 + 4));
  TRY(C.ent_fine.ensure((size_t)slots * W * 2 + 4));
  TRY(C.sorted.ensure((size_t)slots * W * 4 + 4));
  TRY(C.bsum.ensure((size_t)NB * 128));
  TRY(C.partial.ensure((size_t)nruns * 256));
  // The window sums (and the number of sorted entries: instrumentation for the roofline) are written by the last
  // kernels straight into pinned host memory, which the device addresses over PCIe: no copy command between the
  // reduction and the host tail.
  const size_t ws_bytes = (size_t)plan.M * Wb * comps * 128;
  TRY(C.h_winsum.ensure(ws_bytes + 16));
  uint32_t* d_winsum = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&d_winsum, C.h_winsum.p, 0));
  uint32_t* d_entries = d_winsum + ws_bytes / 4;
  uint32_t* bin_count = C.counts.as<uint32_t>();
  uint32_t* chunk_counter = bin_count + (size_t)sp.nbins * BIN_PAD;  // zeroed with bin_count
  uint32_t* red_tickets = chunk_counter + 16;  // k_reduce2: workgroups done per output (1024 words, zeroed with bin_count)
  uint32_t* bcount = red_tickets + 1024;  // per bucket: entries (k_bin_count), then the placing cursor (k_bin_place)
  uint32_t* bin_cursor = bcount + NB;
  // heavy-bucket lists: at most entries/HEAVY_T heavy buckets and entries/HEAVY_PIECE + that many pieces
  const uint64_t entries = slots * W;
  const uint64_t mean_len = entries / NB;
  // A single lane walks a list of at most heavy_t entries; longer lists are cut into wave-cooperative pieces.
  // The floor follows the work there is per lane: in a small MSM (the prover's short rounds) one list of 127
  // entries -- the top window's carry-only bucket -- kept a lane busy for 0.45 ms, longer than the rest of
  // the kernel (tools/msm_var_probe.py: 0.18 vs 0.54 ms depending on which side of 128 that bucket fell).
  // table mode (the prover's short rounds) is a chain of latencies, not of throughput: a wave that has its SIMD to
  // itself issues an instruction every 5 clocks, one that shares it every 8 -- one persistent wave per SIMD there
  // (accumulation 0.116 -> 0.092 ms), two everywhere else
  const int accum_waves = table ? 1 : BP_ACCUM_WAVES;
  const uint64_t lanes = (uint64_t)C.num_cus * 4 * accum_waves * 64;
  uint64_t floor_t = 4 * entries / lanes;
  if (floor_t < 32) floor_t = 32;
  if (floor_t > HEAVY_T_MIN) floor_t = HEAVY_T_MIN;
  // far fewer buckets than lanes and long lists (table mode): 2^glog lanes share a bucket, each about a dozen entries
  int glog = 0;
  while (glog < 6 && (mean_len >> glog) > 12 && ((uint64_t)NB << (glog + 1)) <= lanes) glog++;
  const uint64_t lane_len = mean_len >> glog;  // entries a lane walks
  // (Letting one lane walk lists up to a third of a lane's total work -- the 2^14 top-window buckets of a 2^24-point MSM,
  // 512 entries each against a mean of 32, would then be plain chunks instead of wave pieces -- is slower: 14.5 against
  // 13.7 ms of accumulation; the wave pieces stay.)
  const uint32_t heavy_t = (uint32_t)((4 * lane_len > floor_t ? 4 * lane_len : floor_t) << glog);
  // a wave-piece gives its lanes about as many entries as an ordinary lane walks (a piece of 1024 in an MSM whose lanes
  // walk four entries each was the longest item of the kernel by a factor of three: the sparse top window of any width
  // that does not divide 255 cost 0.15 ms at 2^16 points)
  uint32_t piece_len = (uint32_t)(64 * (mean_len < 4 ? 4 : (mean_len > 16 ? 16 : mean_len)));
  if (piece_len > HEAVY_PIECE_MAX) piece_len = HEAVY_PIECE_MAX;
  const size_t max_heavy = (size_t)(entries / heavy_t) + 1, max_pieces = (size_t)(entries / piece_len) + max_heavy;
  TRY(C.heavy.ensure((3 * max_heavy + 2 * max_pieces + 32 * max_pieces) * 4));
  HeavyLists hv;
  hv.threshold = heavy_t;
  hv.piece = piece_len;
  hv.counters = chunk_counter + 4;
  hv.bucket = C.heavy.as<uint32_t>();
  hv.piece0 = hv.bucket + max_heavy;
  hv.npieces = hv.piece0 + max_heavy;
  hv.piece_lo = hv.npieces + max_heavy;
  hv.piece_hi = hv.piece_lo + max_pieces;
  hv.psum = hv.piece_hi + max_pieces;
  uint32_t* bin_start = C.cursor.as<uint32_t>();
  uint32_t* tile_start = bin_start + sp.nbins + 1;
  const uint32_t max_tiles = (uint32_t)(slots * W / BIN_TILE) + sp.nbins;

  hipStream_t st = C.stream;
  // the phase timers (bpgpu_last_msm_timings) cost five event packets; the prover's short rounds (table mode, 0.25 ms of
  // kernels) go without them: about 0.2 ms per 2^20 proof
  const bool phase_events = !table;  // (on a 2^20-point MSM the five packets are 15-20 us of 1470: kept, the bench reads them)
  if (phase_events) HIP_TRY(hipEventRecord(C.ev[0], st));
  HIP_TRY(hipMemsetAsync(bin_count, 0, (size_t)sp.nbins * BIN_PAD * 4 + 64 + 4096 + (size_t)NB * 4, st));
  k_part_count<SF><<<part_grid, 1024, lds_bytes, st>>>(plan, sp, d_scalars, C.canon.as<uint32_t>(), bin_count);
  if (phase_events) HIP_TRY(hipEventRecord(C.ev[1], st));
  k_bin_scan<<<1, 1024, 0, st>>>(bin_count, bin_start, tile_start, bin_cursor, sp.nbins, d_entries);
  k_part_scatter<<<dim3(cdiv(plan.n, PART_TILE), plan.M * W), 1024, 0, st>>>(
      plan, sp, C.canon.as<uint32_t>(), bin_start, bin_cursor, C.ent_idx.as<uint32_t>(), C.ent_fine.as<uint16_t>());
  if (phase_events) HIP_TRY(hipEventRecord(C.ev[2], st));
  k_bin_count<<<max_tiles, 1024, 0, st>>>(sp, bin_start, tile_start, C.ent_fine.as<uint16_t>(), bcount);
  k_bin_offsets<<<sp.nbins, 1024, 0, st>>>(sp, bin_start, bcount, C.offs.as<uint32_t>(),
                                           C.order.as<uint32_t>(), hv);
  k_bin_place<<<max_tiles, 1024, 0, st>>>(sp, bin_start, tile_start, C.ent_idx.as<uint32_t>(), C.ent_fine.as<uint16_t>(),
                                          C.offs.as<uint32_t>(), bcount, C.sorted.as<uint32_t>());
  if (points_ready) TRY((*points_ready)());
  if (phase_events) HIP_TRY(hipEventRecord(C.ev[3], st));
  k_accum<BF><<<C.num_cus * accum_waves, 256, 0, st>>>(d_points, C.offs.as<uint32_t>(), C.sorted.as<uint32_t>(),
                                             C.order.as<uint32_t>(), C.bsum.as<uint32_t>(), NB, sp.nbins, sp.fine_bits, hv,
                                             chunk_counter, glog);
  k_heavy_merge<BF><<<C.num_cus, 256, 0, st>>>(hv, C.bsum.as<uint32_t>());
  if (!table) HIP_TRY(hipEventRecord(C.ev[4], st));
  // few runs (fewer quads than the chip has lanes at two waves a SIMD): the chain of 2 S additions per run is what the
  // stage takes, so a quad shares each addition (kernels.cuh, quad_add); many runs: one lane each, bound by throughput
  if ((uint64_t)nruns * 4 <= (uint64_t)C.num_cus * 4 * 64 * 2 + 8192)
    k_reduce1_quad<BF><<<cdiv((uint64_t)nruns * 4, 256), 256, 0, st>>>(C.bsum.as<uint32_t>(), C.partial.as<uint32_t>(), nruns, s);
  else
    k_reduce1<BF><<<cdiv(nruns, 256), 256, 0, st>>>(C.bsum.as<uint32_t>(), C.partial.as<uint32_t>(), nruns, s);
  {
    // slices per component: about 1024 waves in all (one per SIMD), at least one value per lane.  Every
    // slice pays a 6-step wave tree on top of its lanes' sums, so more, shorter slices cost more additions
    const uint32_t outputs = (uint32_t)(comps * plan.M * Wb), half = runs_per_window >> 1;
    uint32_t slices = 1;
    // (2048 / 4096 / 8192 waves measured again with the quad-cooperative trees of round 3: slower at every size, 2^14 .. 2^20)
    while (slices < 64 && 2 * slices * outputs <= 1024 && 2 * slices * 64 <= half) slices *= 2;
    const uint32_t groups = cdiv(slices, 4);  // a workgroup adds up its four slices itself
    // More than one workgroup per output (the prover's short rounds): the last one to finish adds up all of them --
    // its addition code is warm, where a third kernel for two additions spent 39 us on a launch and cold fetches.
    // (more than four slices only happen with at most 64 outputs: the loop above keeps slices x outputs <= 1024)
    const bool fused = groups > 1;
    if (fused && (groups > 64 || outputs > 1024)) return fail(BPGPU_E_STATE, "reduction: %u outputs in %u groups", outputs, groups);
    TRY(C.redtmp.ensure((size_t)outputs * groups * 128));
    k_reduce2<BF><<<dim3(comps, plan.M * Wb, groups), 256, 0, st>>>(
        C.partial.as<uint32_t>(), C.redtmp.as<uint32_t>(), d_winsum, runs_per_window, slices, fused ? red_tickets : nullptr);
  }
  if (phase_events) HIP_TRY(hipEventRecord(C.ev[5], st));
  // Horner tail on up to four host threads (the context's pool workers and this one): a batch of two MSMs gives each
  // MSM half of them, a single MSM all.  Share g of an MSM takes the windows [g W / G, (g+1) W / G) and
  // runs its doublings down to bit position 0, so the shares simply add up; the critical path is the 255
  // doublings plus 1 / G of the additions.  The workers are woken once the accumulation has finished and spin
  // on a flag for about as long as the bucket reduction takes (a condition-variable wake-up after the data has
  // arrived costs about as much as the additions it takes over); past that they yield between looks, so that on a
  // host with more contexts than cores a waiting worker never keeps a core from the thread it waits for.
  const uint32_t* hw = (const uint32_t*)C.h_winsum.p;
  const size_t stride = (size_t)Wb * comps * 32;
  const int T = C.hpool.size() + 1;                   // host threads in all
  const int G = T / plan.M > 1 ? T / plan.M : 1;      // shares per MSM (at most 4)
  const int ntasks = plan.M * G;
  host::HPoint<BF> part[MSM_MAX_BATCH];  // ntasks = M G <= max(M, T)
  auto task = [&](int t) {
    for (int k = t; k < ntasks; k += T) {
      const int m = k / G, g = k % G;
      part[k] = horner<BF>(hw + m * stride, Wb, c, s, nbits, g * Wb / G, (g + 1) * Wb / G);
    }
  };
  // Table mode (the prover's short rounds): the tail is a dozen additions per MSM and no doublings -- the calling thread
  // does it alone, without the event, the wake-up and the spinning workers
  const bool solo = table;
  if (!solo) HIP_TRY(hipEventSynchronize(C.ev[4]));
  std::atomic<int> go{0};  // 0: wait, 1: the window sums are in host memory, -1: give up
  if (!solo) C.hpool.begin([&](int w) {
    const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(500);
    uint32_t looks = 0;
    while (go.load(std::memory_order_acquire) == 0) {
      if ((++looks & 63u) == 0 && std::chrono::steady_clock::now() > spin_until)
        std::this_thread::yield();
      else
        __builtin_ia32_pause();
    }
    if (go.load(std::memory_order_acquire) == 1) task(w + 1);
  });
  hipError_t sync_err = hipStreamSynchronize(st);
  if (sync_err == hipSuccess) sync_err = hipGetLastError();
  if (sync_err != hipSuccess) {
    go.store(-1, std::memory_order_release);
    if (!solo) C.hpool.wait();
    return fail(BPGPU_E_HIP, "MSM kernels failed: %s", hipGetErrorString(sync_err));
  }
  {
    uint32_t e32;
    memcpy(&e32, (const char*)C.h_winsum.p + ws_bytes, 4);
    C.msm_entries = e32;
  }
  auto t0 = std::chrono::steady_clock::now();
  go.store(1, std::memory_order_release);
  if (solo) {
    for (int t = 0; t < T; t++) task(t);
  } else {
    task(0);
    C.hpool.wait();
  }
  for (int m = 0; m < plan.M; m++)  // (part[m G] is dead after this: the totals are gathered at the front)
    for (int g = 1; g < G; g++) part[m * G] = host::HEc<BF>::add(part[m * G], part[m * G + g]);
  if (G > 1)
    for (int m = 1; m < plan.M; m++) part[m] = part[m * G];
  write_xy_inf_batch<BF>(part, plan.M, out);
  auto t1 = std::chrono::steady_clock::now();
  C.msm_ms[6] = std::chrono::duration<float, std::milli>(t1 - t0).count();
  C.msm_timed = phase_events;  // the GPU phases are read off the events when bpgpu_last_msm_timings asks
  if (!phase_events)
    for (int i : {0, 1, 2, 3, 5, 7}) C.msm_ms[i] = 0;
  return 0;
}

int msm_dispatch(int curve, const uint32_t* d_points, const uint32_t* d_scalars, const MsmPlan& plan,
                 uint64_t (*out)[9], const std::function<int()>* points_ready = nullptr) {
  return curve == BPGPU_PALLAS ? msm_core<0>(d_points, d_scalars, plan, out, points_ready)
                               : msm_core<1>(d_points, d_scalars, plan, out, points_ready);
}

MsmPlan single_plan(uint32_t n) {
  MsmPlan p;
  memset(&p, 0, sizeof p);
  p.M = 1;
  p.n = n;
  p.seg[0][0] = MsmSeg{0, 0, n};
  return p;
}

int points_in(int curve, const uint32_t* d_in, uint32_t* d_out, size_t n, int fmt) {
  if (n == 0) return 0;
  if (curve == BPGPU_PALLAS)
    k_points_in<0><<<cdiv(n, 64), 64, 0, cx().stream>>>(d_in, d_out, (uint32_t)n, fmt);
  else
    k_points_in<1><<<cdiv(n, 64), 64, 0, cx().stream>>>(d_in, d_out, (uint32_t)n, fmt);
  HIP_TRY(hipGetLastError());
  return 0;
}

int points_out(int curve, const uint32_t* d_in, uint32_t* d_out, size_t n) {
  if (n == 0) return 0;
  if (curve == BPGPU_PALLAS)
    k_points_out<0><<<cdiv(n, 256), 256, 0, cx().stream>>>(d_in, d_out, (uint32_t)n);
  else
    k_points_out<1><<<cdiv(n, 256), 256, 0, cx().stream>>>(d_in, d_out, (uint32_t)n);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- pipelined point upload ------------------------------------------------------------------------------
// Host points (ark Jacobian 96 B or ark affine 64 B) -> dev affine in HBM, in chunks of PT_CHUNK points that the context's
// staging threads push one after the other: copy into a pinned slot, DMA on the thread's own stream, conversion kernel
// (k_points_in) on that same stream -- so conversion of chunk k overlaps the DMA of chunk k+1 and nothing of it sits
// on the context's main stream.  A Jacobian chunk whose Z coordinates are all 1 (every point of both reference recipes:
// G::rand normalises, examples/msm_bench.rs:41, src/relations/ipa.rs:28-36) is compacted to x | y while it is staged: a
// third fewer bytes cross PCIe.  The check reads what the staging copy reads anyway; a chunk with any other Z (or an
// identity, Z = 0) goes up as it is.
// Segments carry a phase (0 or 1): the caller can start device work that needs only the phase-0 points while the
// phase-1 points are still crossing PCIe (the prover's first round: L needs A_hi | B_lo, R the other halves).
//   pt_upload_begin  starts the staging threads and returns; the caller's arrays are read until pt_upload_end returns
//   pt_upload_wait   blocks until every chunk of `phase` has been enqueued, then orders stream `st` after them (and, for a
//                    phase of defer_mask, converts the phase's segments on `st`: one kernel a segment instead of one a chunk)
//   pt_upload_end    joins the staging threads (every chunk enqueued; all phases) and reports their first error
struct PtSeg {
  const uint64_t* src;
  size_t n;
  int fmt;  // 0 jacobian, 1 affine
  uint32_t* d_dst;
  int phase;
};
constexpr uint32_t PT_CHUNK = 32768;  // 3 MiB of Jacobian points: fits a 4 MiB pinned slot

// stages `count` points into buf; returns the bytes to send and the format they are in
template <int BF>
size_t stage_points(const uint64_t* src, uint32_t count, int fmt, void* buf, int* out_fmt) {
  if (fmt == 1) {
    memcpy(buf, src, (size_t)count * 64);
    *out_fmt = 1;
    return (size_t)count * 64;
  }
  const uint64_t o0 = FieldConsts<BF>::H_ONE[0], o1 = FieldConsts<BF>::H_ONE[1], o2 = FieldConsts<BF>::H_ONE[2],
                 o3 = FieldConsts<BF>::H_ONE[3];
  uint64_t* d = (uint64_t*)buf;
  uint64_t bad = 0;
  for (uint32_t i = 0; i < count; i++) {
    const uint64_t* q = src + 12ull * i;
    memcpy(d + 8ull * i, q, 64);
    bad |= (q[8] ^ o0) | (q[9] ^ o1) | (q[10] ^ o2) | (q[11] ^ o3);
  }
  if (bad == 0) {
    *out_fmt = 1;
    return (size_t)count * 64;
  }
  memcpy(buf, src, (size_t)count * 96);
  *out_fmt = 0;
  return (size_t)count * 96;
}

int pt_upload_begin(int curve, const PtSeg* segs, int nseg, hipStream_t st, int defer_mask = 0) {
  Ctx& C = cx();
  Uploader& U = C.up;
  if (C.pt_active) return fail(BPGPU_E_STATE, "a point upload is already in flight on this context");
  HIP_TRY(U.init());
  size_t total = 0;
  for (int g = 0; g < nseg; g++) total += segs[g].n;
  TRY(C.st_pts_in.ensure((total ? total : 1) * 96));
  C.pt_chunks.clear();
  C.pt_deferred.clear();
  C.pt_defer_mask = defer_mask;
  C.pt_curve = curve;
  uint8_t* stage = (uint8_t*)C.st_pts_in.p;
  for (int ph = 0; ph < 2; ph++)
    for (int g = 0; g < nseg; g++) {
      if (segs[g].phase != ph) continue;
      if (((defer_mask >> ph) & 1) && segs[g].n)
        C.pt_deferred.push_back(Ctx::PtDeferred{(const uint32_t*)stage, segs[g].d_dst, (uint32_t)segs[g].n, C.pt_chunks.size(), ph});
      for (size_t o = 0; o < segs[g].n; o += PT_CHUNK) {
        const uint32_t cnt = (uint32_t)(segs[g].n - o < PT_CHUNK ? segs[g].n - o : PT_CHUNK);
        C.pt_chunks.push_back(Ctx::PtChunk{segs[g].src + o * (segs[g].fmt == 0 ? 12 : 8), cnt, segs[g].fmt, segs[g].d_dst + 16ull * o,
                                            stage, ph});
        stage += (size_t)cnt * 96;
      }
    }
  if (defer_mask) TRY(C.pt_fmt.ensure(C.pt_chunks.size() + 1));
  HIP_TRY(hipEventRecord(U.fence, st));  // the staging area and the destination are free once st gets here
  C.pt_phase_count[0].store(0);
  C.pt_phase_count[1].store(0);
  const int device = C.device, nt = U.nt;
  Ctx* ctx = &C;
  C.pt_active = true;
  U.pool.begin([ctx, device, nt, curve](int w) {
    Ctx& C = *ctx;
    Uploader& U = C.up;
    hipError_t r = hipSetDevice(device);
    if (r == hipSuccess) r = hipStreamWaitEvent(U.ws[w], U.fence, 0);
    if (r == hipSuccess) r = hipStreamWaitEvent(U.cs[w], U.fence, 0);
    int cur = 0;
    auto close_phases = [&](int upto) {  // every phase below `upto` is complete for this worker (also after an error)
      while (cur < upto) {
        // (a deferred phase has no conversion kernels of its own: its last operation is a DMA)
        if (r == hipSuccess) r = hipEventRecord(U.pdone[cur][w], ((C.pt_defer_mask >> cur) & 1) ? U.ws[w] : U.cs[w]);
        C.pt_phase_count[cur].fetch_add(1, std::memory_order_release);
        cur++;
      }
    };
    const size_t nch = C.pt_chunks.size();
    size_t mine = 0;
    for (size_t k = (size_t)w; k < nch; k += (size_t)nt, mine++) {
      const Ctx::PtChunk& ch = C.pt_chunks[k];
      close_phases(ch.phase);
      if (r != hipSuccess) continue;
      const int slot = (int)(mine % Uploader::SLOTS);
      if ((r = hipEventSynchronize(U.ev[w][slot])) != hipSuccess) continue;  // the slot's previous DMA has drained
      int cfmt = 0;
      const size_t bytes = curve == BPGPU_PALLAS ? stage_points<0>(ch.src, ch.count, ch.fmt, U.buf[w][slot], &cfmt)
                                                 : stage_points<1>(ch.src, ch.count, ch.fmt, U.buf[w][slot], &cfmt);
      if ((C.pt_defer_mask >> ch.phase) & 1) ((volatile uint8_t*)C.pt_fmt.p)[k] = (uint8_t)cfmt;  // read by the segment's kernel
      if ((r = hipMemcpyAsync(ch.d_stage, U.buf[w][slot], bytes, hipMemcpyHostToDevice, U.ws[w])) != hipSuccess) continue;
      if ((r = hipEventRecord(U.ev[w][slot], U.ws[w])) != hipSuccess) continue;
      if ((C.pt_defer_mask >> ch.phase) & 1) continue;  // converted by pt_upload_wait, a segment at a time
      // The conversion runs on a stream of its own, behind this chunk's DMA: a conversion kernel launched while a bucket
      // accumulation holds the chip does not finish until that kernel ends (kernel trace of a 2^20-element proof: three
      // 8 us conversions took 0.85 ms beside the first round's k_accum -- also as k_points_in_affine, 32 registers and
      // raised priority), and on the DMA's stream it held the worker's later copies back with it.  Worth 0.1 ms of
      // bpgpu_ipa_begin at 2^20 (tools/begin_ab.py, alternating sessions): the transfer itself is bound by the host side.
      if ((r = hipStreamWaitEvent(U.cs[w], U.ev[w][slot], 0)) != hipSuccess) continue;
      hipStream_t ks = U.cs[w];
      if (cfmt == 1) {  // affine rows: the lean kernel that fits beside a running bucket accumulation
        if (curve == BPGPU_PALLAS)
          k_points_in_affine<0><<<cdiv(ch.count, 256), 256, 0, ks>>>((const uint32_t*)ch.d_stage, ch.d_dst, ch.count);
        else
          k_points_in_affine<1><<<cdiv(ch.count, 256), 256, 0, ks>>>((const uint32_t*)ch.d_stage, ch.d_dst, ch.count);
      } else if (curve == BPGPU_PALLAS)
        k_points_in<0><<<cdiv(ch.count, 64), 64, 0, ks>>>((const uint32_t*)ch.d_stage, ch.d_dst, ch.count, cfmt);
      else
        k_points_in<1><<<cdiv(ch.count, 64), 64, 0, ks>>>((const uint32_t*)ch.d_stage, ch.d_dst, ch.count, cfmt);
      r = hipGetLastError();
    }
    close_phases(2);
    C.pt_err[w] = r;
  });
  return 0;
}

int pt_upload_wait(int phase, hipStream_t st) {
  Ctx& C = cx();
  Uploader& U = C.up;
  if (!C.pt_active) return fail(BPGPU_E_STATE, "no point upload in flight");
  uint32_t looks = 0;
  while (C.pt_phase_count[phase].load(std::memory_order_acquire) < U.nt) {
    if ((++looks & 255u) == 0) std::this_thread::yield(); else __builtin_ia32_pause();
  }
  for (int w = 0; w < U.nt; w++) HIP_TRY(hipStreamWaitEvent(st, U.pdone[phase][w], 0));
  if ((C.pt_defer_mask >> phase) & 1) {  // the phase's DMAs are ordered before st now: convert its segments there
    uint8_t* d_fmt = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void**)&d_fmt, C.pt_fmt.p, 0));
    for (const Ctx::PtDeferred& d : C.pt_deferred) {
      if (d.phase != phase) continue;
      if (C.pt_curve == BPGPU_PALLAS)
        k_points_in_chunked<0><<<cdiv(d.n, 256), 256, 0, st>>>(d.d_stage, d.d_dst, d.n, PT_CHUNK, d_fmt + d.first_chunk);
      else
        k_points_in_chunked<1><<<cdiv(d.n, 256), 256, 0, st>>>(d.d_stage, d.d_dst, d.n, PT_CHUNK, d_fmt + d.first_chunk);
      HIP_TRY(hipGetLastError());
    }
  }
  return 0;
}

int pt_upload_end() {
  Ctx& C = cx();
  if (!C.pt_active) return 0;
  C.up.pool.wait();
  C.pt_active = false;
  for (int w = 0; w < C.up.nt; w++)
    if (C.pt_err[w] != hipSuccess) return fail(BPGPU_E_HIP, "point upload failed: %s", hipGetErrorString(C.pt_err[w]));
  return 0;
}

// While `on` is set a chunked upload is in flight: the caller's arrays are read until pt_upload_end returns, on every
// path out of the scope.
struct PipeGuard {
  bool& on;
  ~PipeGuard() {
    if (on) (void)pt_upload_end();
  }
};

// the whole thing for one array on the context's stream: d_out is valid for work enqueued on st after the call
int upload_points(int curve, const uint64_t* points, size_t n, int fmt, uint32_t* d_out, hipStream_t st) {
  if (n == 0) return 0;
  if (n * (fmt == 0 ? 96 : 64) < Uploader::MIN_BYTES) {
    Ctx& C = cx();
    TRY(C.st_pts_in.ensure(n * 96));
    HIP_TRY(hipMemcpyAsync(C.st_pts_in.p, points, n * (fmt == 0 ? 96 : 64), hipMemcpyHostToDevice, st));
    return points_in(curve, C.st_pts_in.as<uint32_t>(), d_out, n, fmt);
  }
  PtSeg seg{points, n, fmt, d_out, 0};
  TRY(pt_upload_begin(curve, &seg, 1, st));
  int rc = pt_upload_wait(0, st);
  int rc2 = pt_upload_end();
  return rc ? rc : rc2;
}

// <a, b> for up to two pairs at once (a's offsets are relative to d_a, b's to d_b); results (ark form) land in
// d_out0 / d_out1.  d_partial: room for 2 * ip_blocks(n) block sums of 32 bytes.
uint32_t ip_blocks(uint32_t n) {
  uint32_t nblk = n == 0 ? 1 : cdiv(n, 256 * 8);
  return nblk > 1024 ? 1024 : nblk;
}
int ip_launch_with(int curve, const uint32_t* d_a, const uint32_t* d_b, IpPair p0, IpPair p1, int npairs, uint32_t n,
                   uint32_t* d_partial, uint32_t* d_out0, uint32_t* d_out1) {
  Ctx& C = cx();
  const uint32_t nblk = ip_blocks(n);
  dim3 grid(nblk, npairs);
  OpSpan span(cx());
  if (curve == BPGPU_PALLAS) {
    k_ip_partial<1><<<grid, 256, 0, C.stream>>>(d_a, d_b, p0, p1, n, d_partial);
    k_ip_final<1><<<npairs, 256, 0, C.stream>>>(d_partial, nblk, d_out0, d_out1);
  } else {
    k_ip_partial<0><<<grid, 256, 0, C.stream>>>(d_a, d_b, p0, p1, n, d_partial);
    k_ip_final<0><<<npairs, 256, 0, C.stream>>>(d_partial, nblk, d_out0, d_out1);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
int ip_launch(int curve, const uint32_t* d_a, const uint32_t* d_b, IpPair p0, IpPair p1, int npairs, uint32_t n,
              uint32_t* d_out0, uint32_t* d_out1) {
  TRY(cx().st_ip.ensure((size_t)2 * ip_blocks(n) * 32));
  return ip_launch_with(curve, d_a, d_b, p0, p1, npairs, n, cx().st_ip.as<uint32_t>(), d_out0, d_out1);
}

FoldScalars make_fold_scalars(const uint64_t lo[4], const uint64_t hi[4]) {
  FoldScalars f;
  memcpy(f.lo, lo, 32);
  memcpy(f.hi, hi, 32);
  return f;
}

int fold_scalars_launch(int curve, uint32_t* v0, uint32_t* o0, const FoldScalars& s0, uint32_t* v1, uint32_t* o1,
                        const FoldScalars& s1, int nvec, uint32_t n) {
  if (n == 0) return 0;
  uint32_t gx = cdiv(n, 256);
  const uint32_t cap = (uint32_t)cx().num_cus * 16;
  if (gx > cap) gx = cap;
  dim3 grid(gx, nvec);
  OpSpan span(cx());
  if (curve == BPGPU_PALLAS)
    k_fold_scalars<1><<<grid, 256, 0, cx().stream>>>(v0, o0, s0, v1, o1, s1, n);
  else
    k_fold_scalars<0><<<grid, 256, 0, cx().stream>>>(v0, o0, s0, v1, o1, s1, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

// non-zero 4-bit codes among the first len positions of a recoding (= mixed additions a lane performs)
inline uint32_t count_codes(const uint32_t* codes, int len) {
  uint32_t c = 0;
  for (int pos = 0; pos < len; pos++) c += ((codes[pos >> 3] >> ((pos & 7) * 4)) & 7u) != 0;
  return c;
}

// ---- recoding of the two fold multipliers (shared by every lane of k_fold_points) -------------
void shr1(uint64_t v[5]) {
  for (int i = 0; i < 4; i++) v[i] = (v[i] >> 1) | (v[i + 1] << 63);
  v[4] >>= 1;
}
bool nonzero(const uint64_t v[5]) { return (v[0] | v[1] | v[2] | v[3] | v[4]) != 0; }

// Joint sparse form (Solinas) of two non-negative 256-bit integers as 4-bit codes; returns the length.
int jsf_codes(const uint64_t a[4], const uint64_t b[4], uint32_t* codes, int max_positions) {
  uint64_t k0[5] = {a[0], a[1], a[2], a[3], 0};
  uint64_t k1[5] = {b[0], b[1], b[2], b[3], 0};
  int d0 = 0, d1 = 0, pos = 0;
  while ((nonzero(k0) || nonzero(k1) || d0 || d1) && pos < max_positions) {
    int l0 = (int)((k0[0] & 7) + d0) & 7, l1 = (int)((k1[0] & 7) + d1) & 7;
    int u0 = 0, u1 = 0;
    if (l0 & 1) {
      u0 = 2 - (l0 & 3);
      if (((l0 & 7) == 3 || (l0 & 7) == 5) && (l1 & 3) == 2) u0 = -u0;
    }
    if (l1 & 1) {
      u1 = 2 - (l1 & 3);
      if (((l1 & 7) == 3 || (l1 & 7) == 5) && (l0 & 3) == 2) u1 = -u1;
    }
    if (2 * d0 == 1 + u0) d0 = 1 - d0;
    if (2 * d1 == 1 + u1) d1 = 1 - d1;
    shr1(k0);
    shr1(k1);
    uint32_t code = 0;
    if (u0 != 0 && u1 == 0) code = 1 | (u0 < 0 ? 8 : 0);
    else if (u0 == 0 && u1 != 0) code = 2 | (u1 < 0 ? 8 : 0);
    else if (u0 != 0 && u0 == u1) code = 3 | (u0 < 0 ? 8 : 0);
    else if (u0 != 0) code = 4 | (u0 < 0 ? 8 : 0);  // u1 = -u0:  u0 (first - second)
    codes[pos >> 3] |= code << ((pos & 7) * 4);
    pos++;
  }
  return pos;
}

// (the GLV split k = k1 + k2 lambda and its 256-bit helpers live in kernels.cuh: host and device share them)
template <int CURVE>
int make_fold_plan(const uint64_t s_lo_ark[4], const uint64_t s_hi_ark[4], FoldPlan& f) {
  using HFs = host::HF<1 - CURVE>;
  host::H4 a, b;
  memcpy(a.v, s_lo_ark, 32);
  memcpy(b.v, s_hi_ark, 32);
  host::H4 ca = HFs::to_canon(a), cb = HFs::to_canon(b);
  memset(&f, 0, sizeof f);
  memcpy(f.s_lo, ca.v, 32);
  memcpy(f.s_hi, cb.v, 32);
  uint64_t k1[4], k2[4], l1[4], l2[4];
  bool n1, n2, m1, m2;
  glv_split<CURVE>(ca.v, k1, n1, k2, n2);
  glv_split<CURVE>(cb.v, l1, m1, l2, m2);
  // the halves must be short (they are < 2^129 by construction); anything else is a bug
  if (k1[3] | k2[3] | l1[3] | l2[3] | (k1[2] >> 4) | (k2[2] >> 4) | (l1[2] >> 4) | (l2[2] >> 4))
    return fail(BPGPU_E_STATE, "GLV split produced an over-long half");
  f.neg = (n1 ? 1u : 0u) | (n2 ? 2u : 0u) | (m1 ? 4u : 0u) | (m2 ? 8u : 0u);
  int lp = jsf_codes(k1, k2, f.codes_p, 136), lq = jsf_codes(l1, l2, f.codes_q, 136);
  f.len = lp > lq ? lp : lq;
  return 0;
}

int fold_multi(int curve, const uint32_t* p, const uint32_t* q, const uint64_t* s, int m, uint32_t n, uint32_t* o);

int fold_points_launch(int curve, const uint32_t* g0, uint32_t* o0, const uint64_t lo0[4], const uint64_t hi0[4],
                       const uint32_t* g1, uint32_t* o1, const uint64_t lo1[4], const uint64_t hi1[4], int nvec,
                       uint32_t n) {
  if (n == 0) return 0;
  Ctx& C = cx();
  // Both multipliers on one doubling chain with Eisenstein width-2 digits (k_fold_multi: 128 doublings + ~110 additions
  // an output, 2 or 4 lanes an output for short vectors) -- 7.5 -> 6.5 ms at 2^19 outputs, 1.04 -> 0.70 ms at 2^14 against
  // k_fold_points below (GLV halves in joint sparse form: ~130 additions), which stays as the path for a schedule that does
  // not fit that kernel's operation list.  In place (o == g) is fine there: a lane reads its own column only, before the
  // output's one store.
  {
    OpSpan span(C);
    C.fold_ops[0] = C.fold_ops[1] = C.fold_ops[2] = 0;
    uint64_t s2[8];
    memcpy(s2, lo0, 32);
    memcpy(s2 + 4, hi0, 32);
    int rc = fold_multi(curve, nullptr, g0, s2, 2, n, o0);
    if (rc == 0 && nvec > 1) {
      memcpy(s2, lo1, 32);
      memcpy(s2 + 4, hi1, 32);
      rc = fold_multi(curve, nullptr, g1, s2, 2, n, o1);
    }
    if (rc <= 0) return rc;
  }
  Plans2<FoldPlan> plans;  // kernel arguments
  if (curve == BPGPU_PALLAS) {
    TRY(make_fold_plan<0>(lo0, hi0, plans.p[0]));
    TRY(make_fold_plan<0>(nvec > 1 ? lo1 : lo0, nvec > 1 ? hi1 : hi0, plans.p[1]));
  } else {
    TRY(make_fold_plan<1>(lo0, hi0, plans.p[0]));
    TRY(make_fold_plan<1>(nvec > 1 ? lo1 : lo0, nvec > 1 ? hi1 : hi0, plans.p[1]));
  }
  C.fold_ops[0] = C.fold_ops[1] = C.fold_ops[2] = 0;
  C.fold_ops[3] = 3;
  for (int v = 0; v < nvec; v++) {
    const FoldPlan& f = plans.p[v];
    C.fold_ops[0] += n;
    C.fold_ops[1] += (uint64_t)n * f.len;
    C.fold_ops[2] += (uint64_t)n * (count_codes(f.codes_p, f.len) + count_codes(f.codes_q, f.len));
  }
  dim3 grid(cdiv(n, 64), nvec);
  OpSpan span(cx());
  if (curve == BPGPU_PALLAS)
    k_fold_points<0><<<grid, 64, 0, C.stream>>>(g0, o0, g1, o1, plans, n);
  else
    k_fold_points<1><<<grid, 64, 0, C.stream>>>(g0, o0, g1, o1, plans, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- monic fold: Eisenstein-unit expansion of s = k1 + k2 lambda ------------------------------
// 192-bit two's complement, enough for |k| < 2^130
struct S192 {
  uint64_t w[3];
};
S192 s192_from(const uint64_t mag[4], bool neg) {
  S192 r{{mag[0], mag[1], mag[2]}};
  if (neg) {
    unsigned __int128 c = 1;
    for (int i = 0; i < 3; i++) {
      c += (uint64_t)~r.w[i];
      r.w[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
bool s192_zero(const S192& a) { return (a.w[0] | a.w[1] | a.w[2]) == 0; }
void s192_sub_small(S192& a, int d) {  // a -= d, |d| <= 2
  if (d == 0) return;
  if (d > 0) {
    const uint64_t old = a.w[0];
    a.w[0] -= (uint64_t)d;
    if (old < (uint64_t)d)
      for (int i = 1; i < 3; i++)
        if (a.w[i]-- != 0) break;
  } else {
    const uint64_t add = (uint64_t)(-d);
    a.w[0] += add;
    if (a.w[0] < add)
      for (int i = 1; i < 3; i++)
        if (++a.w[i] != 0) break;
  }
}
void s192_sar1(S192& a) {
  a.w[0] = (a.w[0] >> 1) | (a.w[1] << 63);
  a.w[1] = (a.w[1] >> 1) | (a.w[2] << 63);
  a.w[2] = (uint64_t)((int64_t)a.w[2] >> 1);
}
// Width-2 expansion of the Eisenstein integer z = a + b w in base 2.  2 is prime in Z[w], and the twelve odd
// residues modulo 4 are the six units u in {+-1, +-w, +-(1 + w)} and their products with alpha = 1 - w (the six
// elements of norm 3: +-(1 - w), +-(1 + 2w), +-(2 + w)).  An odd z takes the digit d congruent to it modulo 4;
// z - d is then divisible by 4, so the next digit is 0: one position in three is non-zero on average (the
// unit-only expansion of round 1 had 56 %).  The points of the digits cost no arithmetic given Q and
// Q' = alpha Q = Q - phi(Q) (one affine addition per generator, set up by the kernel): u Q and u Q' are
// (zeta^k x, +-y).
// 4-bit codes, position 0 = least significant:  code & 3: 0 none, 1: x, 2: zeta x, 3: zeta^2 x;
// code & 4: the point is Q' instead of Q;  code & 8: use -y.
struct EisDigit {
  int da, db;     // the digit as a + b w
  uint32_t code;
};
const EisDigit* eis_digit_mod4(uint32_t a4, uint32_t b4) {
  // unit digits: (1,0) -> x, +y;  (0,1) -> zeta x, +y;  (1,1) = -w^2 -> zeta^2 x, -y;  negatives flip y
  static const EisDigit D[12] = {
      {1, 0, 1},  {-1, 0, 1 | 8},  {0, 1, 2},  {0, -1, 2 | 8},  {1, 1, 3 | 8},  {-1, -1, 3},
      // alpha = 1 - w;  w alpha = 1 + 2w;  (1 + w) alpha = 2 + w: the same three maps applied to Q'
      {1, -1, 4 | 1}, {-1, 1, 4 | 1 | 8}, {1, 2, 4 | 2}, {-1, -2, 4 | 2 | 8}, {2, 1, 4 | 3 | 8}, {-2, -1, 4 | 3}};
  for (const EisDigit& d : D)
    if (((uint32_t)d.da & 3u) == a4 && ((uint32_t)d.db & 3u) == b4) return &d;
  return nullptr;
}
int eisenstein_codes(S192 a, S192 b, uint32_t* codes, int max_positions) {
  int pos = 0;
  while (!(s192_zero(a) && s192_zero(b))) {
    if (pos >= max_positions) return -1;
    const uint32_t a4 = (uint32_t)a.w[0] & 3u, b4 = (uint32_t)b.w[0] & 3u;
    if ((a4 | b4) & 1u) {
      const EisDigit* d = eis_digit_mod4(a4, b4);
      if (!d) return -1;
      codes[pos >> 3] |= d->code << ((pos & 7) * 4);
      s192_sub_small(a, d->da);
      s192_sub_small(b, d->db);
    }
    s192_sar1(a);
    s192_sar1(b);
    pos++;
  }
  return pos;
}

template <int CURVE>
int make_monic_plan(const uint64_t s_ark[4], MonicPlan& f) {
  using HFs = host::HF<1 - CURVE>;
  host::H4 a;
  memcpy(a.v, s_ark, 32);
  host::H4 ca = HFs::to_canon(a);
  memset(&f, 0, sizeof f);
  uint64_t k1[4], k2[4];
  bool n1, n2;
  glv_split<CURVE>(ca.v, k1, n1, k2, n2);
  if (k1[3] | k2[3] | (k1[2] >> 4) | (k2[2] >> 4)) return fail(BPGPU_E_STATE, "GLV split produced an over-long half");
  f.len = eisenstein_codes(s192_from(k1, n1), s192_from(k2, n2), f.codes, 136);
  if (f.len < 0) return fail(BPGPU_E_STATE, "unit expansion too long");
  return 0;
}

// o0[i] = p0[i] + s0 q0[i] (has_p) or s0 q0[i] (!has_p); same for vector 1 when nvec == 2
int fold_monic_pq(int curve, const uint32_t* p0, const uint32_t* q0, uint32_t* o0, const uint64_t s0[4], const uint32_t* p1,
                  const uint32_t* q1, uint32_t* o1, const uint64_t s1[4], int nvec, uint32_t n, int has_p);
// out0[i] = g0[i] + s0 g0[i+n] (has_p) or s0 g0[i] (!has_p); same for vector 1 when nvec == 2
int fold_monic_launch(int curve, const uint32_t* g0, uint32_t* o0, const uint64_t s0[4], const uint32_t* g1, uint32_t* o1,
                      const uint64_t s1[4], int nvec, uint32_t n, int has_p) {
  const size_t off = has_p ? 16ull * n : 0;
  return fold_monic_pq(curve, g0, g0 + off, o0, s0, g1, g1 + off, o1, s1, nvec, n, has_p);
}
int fold_monic_pq(int curve, const uint32_t* p0, const uint32_t* q0, uint32_t* o0, const uint64_t s0[4], const uint32_t* p1,
                  const uint32_t* q1, uint32_t* o1, const uint64_t s1[4], int nvec, uint32_t n, int has_p) {
  if (n == 0) return 0;
  Ctx& C = cx();
  Plans2<MonicPlan> plans;  // kernel arguments
  if (curve == BPGPU_PALLAS) {
    TRY(make_monic_plan<0>(s0, plans.p[0]));
    TRY(make_monic_plan<0>(nvec > 1 ? s1 : s0, plans.p[1]));
  } else {
    TRY(make_monic_plan<1>(s0, plans.p[0]));
    TRY(make_monic_plan<1>(nvec > 1 ? s1 : s0, plans.p[1]));
  }
  C.fold_ops[0] = C.fold_ops[1] = C.fold_ops[2] = 0;
  C.fold_ops[3] = 1;
  for (int v = 0; v < nvec; v++) {
    C.fold_ops[0] += n;
    C.fold_ops[1] += (uint64_t)n * plans.p[v].len;
    C.fold_ops[2] += (uint64_t)n * (count_codes(plans.p[v].codes, plans.p[v].len) + (has_p ? 1u : 0u));
  }
  dim3 grid(cdiv(n, 256), nvec);
  OpSpan span(cx());
  if (curve == BPGPU_PALLAS)
    k_fold_monic<0><<<grid, 256, 0, C.stream>>>(p0, q0, o0, p1, q1, o1, plans, n, has_p);
  else
    k_fold_monic<1><<<grid, 256, 0, C.stream>>>(p0, q0, o0, p1, q1, o1, plans, n, has_p);
  HIP_TRY(hipGetLastError());
  return 0;
}

// o[i] = [p[i] +] sum_{t < m} s[t] q[t n + i] with all m <= MULTI_MAX_M multipliers on one doubling chain (k_fold_multi).
// Returns 1 (nothing launched) when the schedule does not fit the kernel's operation list; the caller then folds one
// multiplier at a time.
int fold_multi(int curve, const uint32_t* p, const uint32_t* q, const uint64_t* s, int m, uint32_t n, uint32_t* o) {
  if (n == 0 || m < 1 || m > MULTI_MAX_M) return 1;
  Ctx& C = cx();
  MonicPlan plan[MULTI_MAX_M];
  int len = 0;
  for (int t = 0; t < m; t++) {
    if (curve == BPGPU_PALLAS) TRY(make_monic_plan<0>(s + 4 * t, plan[t])); else TRY(make_monic_plan<1>(s + 4 * t, plan[t]));
    if (plan[t].len > len) len = plan[t].len;
  }
  static_assert(sizeof(MultiOps) <= 2048, "kernel arguments");
  MultiOps ops;
  memset(&ops, 0, sizeof ops);
  ops.m = (uint32_t)m;
  // short vectors: 2 or 4 lanes an output while that still leaves about a wave a SIMD (k_fold_multi); each lane gets
  // every doubling and its turn of the additions
  const uint64_t chip_lanes = (uint64_t)C.num_cus * 256;
  const uint32_t lanes = (uint64_t)n * 4 <= chip_lanes ? 4u : ((uint64_t)n * 2 <= chip_lanes ? 2u : 1u);
  ops.lanes = lanes;
  std::vector<uint16_t> list[4];
  uint32_t pending[4] = {0, 0, 0, 0}, adds = 0;
  for (int pos = len - 1; pos >= 0; pos--) {
    for (uint32_t r = 0; r < lanes; r++) pending[r]++;
    for (int t = 0; t < m; t++) {
      if (pos >= plan[t].len) continue;
      const uint32_t code = (plan[t].codes[pos >> 3] >> ((pos & 7) * 4)) & 15u;
      if (!(code & 3u)) continue;
      const uint32_t r = adds % lanes;
      for (; pending[r] > 31; pending[r] -= 31) list[r].push_back((uint16_t)(31u << 7));
      list[r].push_back((uint16_t)(0x1000u | (pending[r] << 7) | ((uint32_t)t << 4) | code));
      pending[r] = 0;
      adds++;
    }
  }
  uint32_t nops = 0;
  for (uint32_t r = 0; r < lanes; r++) {
    for (; pending[r]; pending[r] -= (pending[r] > 31 ? 31 : pending[r])) list[r].push_back((uint16_t)((pending[r] > 31 ? 31u : pending[r]) << 7));
    ops.start[r] = nops;
    if (nops + list[r].size() > (size_t)MULTI_MAX_OPS) return 1;
    memcpy(ops.op + nops, list[r].data(), list[r].size() * sizeof(uint16_t));
    nops += (uint32_t)list[r].size();
  }
  for (uint32_t r = lanes; r <= 4; r++) ops.start[r] = nops;
  TRY(C.fold_alpha.ensure((size_t)m * n * 64));
  C.fold_ops[0] += n;
  C.fold_ops[1] += (uint64_t)n * (uint64_t)len * lanes;                     // every lane of an output doubles through all positions
  C.fold_ops[2] += (uint64_t)n * (adds + (p ? 1u : 0u) + (lanes - 1));    // and lanes - 1 additions join the partial sums
  C.fold_ops[3] = 4;
  OpSpan span(C);
  if (curve == BPGPU_PALLAS)
    k_fold_multi<0><<<cdiv(n, 256 / lanes), 256, 0, C.stream>>>(q, p, C.fold_alpha.as<uint32_t>(), o, ops, n);
  else
    k_fold_multi<1><<<cdiv(n, 256 / lanes), 256, 0, C.stream>>>(q, p, C.fold_alpha.as<uint32_t>(), o, ops, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

size_t pow2_at_least(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace

struct bpgpu_ipa {
  Ctx* ctx;     // the context the session was opened on; every later call runs there
  bool failed;  // a call failed half-way through a round: the state is undefined, every later call is refused
  int curve;
  size_t np;   // padded original length
  size_t len;  // current vector length
  // points: A at [0, np), B at [np, 2np), Q at 2np           (dev affine, 64 B each)
  // scalars: a at [0, np), b at [np, 2np), c_L at 2np, c_R at 2np+1 (ark form, 32 B each)
  //          then (when short rounds defer the generator fold) 4 n0 MSM scalars and 2 n0 weights
  uint32_t* d_pts;
  uint32_t* d_sc;
  bool pooled;  // d_pts / d_sc belong to the context's pool
  bool lr_done;
  // L, R of the first round, computed by bpgpu_ipa_begin while the generators were still crossing PCIe
  bool lr_cached;
  uint64_t l_cache[9], r_cache[9];
  // deferred generator folding (kernels.cuh, k_frozen_*)
  size_t freeze_cap;     // room reserved: n0 <= freeze_cap
  size_t freeze_thresh;  // vectors this short freeze for good
  bool frozen;
  bool freeze_disabled;
  // pair folding: the generators of a long round are frozen for ONE round (lazy) and the two pending
  // challenges are then applied together by k_fold_joint4
  bool lazy;
  bool lazy_off;
  size_t n0;  // length at which the generators stopped being folded
  // fixed-base table of the frozen generators (kernels.cuh, k_table_*): 2^(c w) of A0 | B0 | Q, stride 2 n0 + 1
  uint32_t* d_table;  // null: the short rounds run the general MSM over the frozen generators themselves
  int table_c;
  std::vector<std::array<uint64_t, 8>> pending;  // (x, x^-1) of the folds not applied to the points
  // Common factors of the stored generators: the true vectors are muA * A and muB * B.  A fold
  // x^-1 A_lo + x A_hi is done as A_lo + x^2 A_hi (one scalar multiplication per pair, k_fold_monic)
  // with muA *= x^-1; the factors go into the MSM scalars (round_lr), into the weights when the
  // generators freeze, and into the points themselves only when they are exported.
  uint64_t muA[4], muB[4];  // ark form
  bool mu_one;
  uint32_t* scaled() const { return d_sc + 8ull * sc_scaled; }  // 2 np scratch scalars (mu a | mu b)
  size_t sc_scaled;
  uint32_t* fsc() const { return d_sc + 8ull * (2 * np + 2); }
  uint32_t* wA() const { return fsc() + 8ull * 4 * freeze_cap; }
  uint32_t* wB() const { return wA() + 8ull * freeze_cap; }
};

namespace {
template <int SF>
void mu_mul(uint64_t mu[4], const uint64_t f[4]) {
  host::H4 a, b;
  memcpy(a.v, mu, 32);
  memcpy(b.v, f, 32);
  host::H4 r = host::HF<SF>::mul(a, b);
  memcpy(mu, r.v, 32);
}
template <int SF>
bool are_inverses(const uint64_t x[4], const uint64_t xi[4]) {
  host::H4 a, b;
  memcpy(a.v, x, 32);
  memcpy(b.v, xi, 32);
  return host::HF<SF>::eq(host::HF<SF>::mul(a, b), host::HF<SF>::one());
}
template <int SF>
void sqr_ark(const uint64_t x[4], uint64_t out[4]) {
  host::H4 a;
  memcpy(a.v, x, 32);
  host::H4 r = host::HF<SF>::sqr(a);
  memcpy(out, r.v, 32);
}
void ipa_mu_reset(bpgpu_ipa* s) {
  host::H4 one = s->curve == BPGPU_PALLAS ? host::HF<1>::one() : host::HF<0>::one();
  memcpy(s->muA, one.v, 32);
  memcpy(s->muB, one.v, 32);
  s->mu_one = true;
}
// multiply the common factors into the stored generators (vector length s->len)
int ipa_normalise(bpgpu_ipa* s) {
  if (s->mu_one) return 0;
  uint32_t* dA = s->d_pts;
  uint32_t* dB = s->d_pts + 16ull * s->np;
  TRY(fold_monic_launch(s->curve, dA, dA, s->muA, dB, dB, s->muB, 2, (uint32_t)s->len, 0));
  ipa_mu_reset(s);
  return 0;
}
// A' = x^-1 A_lo + x A_hi,  B' = x B_lo + x^-1 B_hi  on the stored generators, vectors of length 2n
int ipa_fold_generators(bpgpu_ipa* s, const uint64_t x[4], const uint64_t x_inv[4], uint32_t n) {
  uint32_t* dA = s->d_pts;
  uint32_t* dB = s->d_pts + 16ull * s->np;
  const bool pallas = s->curve == BPGPU_PALLAS;
  if (pallas ? are_inverses<1>(x, x_inv) : are_inverses<0>(x, x_inv)) {
    uint64_t x2[4], xi2[4];
    if (pallas) {
      sqr_ark<1>(x, x2);
      sqr_ark<1>(x_inv, xi2);
    } else {
      sqr_ark<0>(x, x2);
      sqr_ark<0>(x_inv, xi2);
    }
    TRY(fold_monic_launch(s->curve, dA, dA, x2, dB, dB, xi2, 2, n, 1));
    if (pallas) {
      mu_mul<1>(s->muA, x_inv);
      mu_mul<1>(s->muB, x);
    } else {
      mu_mul<0>(s->muA, x_inv);
      mu_mul<0>(s->muB, x);
    }
    s->mu_one = false;
    return 0;
  }
  // (x, x_inv) are not inverses of each other: the two-scalar fold, on the true generators
  const size_t len = s->len;
  s->len = 2 * (size_t)n;
  int rc = ipa_normalise(s);
  s->len = len;
  TRY(rc);
  return fold_points_launch(s->curve, dA, dA, x_inv, x, dB, dB, x, x_inv, 2, n);
}
// out[i] = g[i] + s[0] g[i+q] + s[1] g[i+2q] + s[2] g[i+3q] for both generator vectors (multipliers in ark form)
int fold_joint4_launch(int curve, uint32_t* gA, const uint64_t sA[3][4], uint32_t* gB, const uint64_t sB[3][4], uint32_t q) {
  if (q == 0) return 0;
  Ctx& C = cx();
  static_assert(sizeof(Plans2<Joint4Plan>) <= 1024, "kernel arguments");
  Plans2<Joint4Plan> plans;
  for (int v = 0; v < 2; v++) {
    plans.p[v].len = 0;
    for (int t = 0; t < 3; t++) {
      const uint64_t* m = v == 0 ? sA[t] : sB[t];
      if (curve == BPGPU_PALLAS) TRY(make_monic_plan<0>(m, plans.p[v].m[t])); else TRY(make_monic_plan<1>(m, plans.p[v].m[t]));
      if (plans.p[v].m[t].len > plans.p[v].len) plans.p[v].len = plans.p[v].m[t].len;
    }
  }
  C.fold_ops[0] = 2ull * q;
  C.fold_ops[1] = C.fold_ops[2] = 0;
  C.fold_ops[3] = 2;
  for (int v = 0; v < 2; v++) {
    C.fold_ops[1] += (uint64_t)q * plans.p[v].len;
    uint32_t adds = 1;
    for (int t = 0; t < 3; t++) adds += count_codes(plans.p[v].m[t].codes, plans.p[v].m[t].len);
    C.fold_ops[2] += (uint64_t)q * adds;
  }
  // short vectors: two lanes an output (half of the chain's additions each) while that still leaves a wave a SIMD
  const bool split = (uint64_t)q * 2 * 2 <= (uint64_t)C.num_cus * 256;
  dim3 grid(split ? cdiv(q, 128) : cdiv(q, 256), 2);
  OpSpan span(cx());
  if (split) {
    if (curve == BPGPU_PALLAS)
      k_fold_joint4_split<0><<<grid, 256, 0, C.stream>>>(gA, gA, gB, gB, plans, q);
    else
      k_fold_joint4_split<1><<<grid, 256, 0, C.stream>>>(gA, gA, gB, gB, plans, q);
  } else if (curve == BPGPU_PALLAS)
    k_fold_joint4<0><<<grid, 256, 0, C.stream>>>(gA, gA, gB, gB, plans, q);
  else
    k_fold_joint4<1><<<grid, 256, 0, C.stream>>>(gA, gA, gB, gB, plans, q);
  HIP_TRY(hipGetLastError());
  return 0;
}

int ipa_replay_pending(bpgpu_ipa* s);

// Both pending challenges (x0 of the round that froze the generators at length n0, x1 of the next) in one
// pass over the stored generators:  A'' = (x0 x1)^-1 [A_i + x1^2 A_{i+q} + x0^2 A_{i+2q} + (x0 x1)^2 A_{i+3q}],
// B'' the same with inverted challenges; the leading factors join the common factors.
template <int SF>
int ipa_pair_fold_t(bpgpu_ipa* s) {
  using HF = host::HF<SF>;
  const uint64_t* c0 = s->pending[0].data();
  const uint64_t* c1 = s->pending[1].data();
  if (!are_inverses<SF>(c0, c0 + 4) || !are_inverses<SF>(c1, c1 + 4)) return ipa_replay_pending(s);
  host::H4 x0, x0i, x1, x1i;
  memcpy(x0.v, c0, 32);
  memcpy(x0i.v, c0 + 4, 32);
  memcpy(x1.v, c1, 32);
  memcpy(x1i.v, c1 + 4, 32);
  const host::H4 a1 = HF::sqr(x1), a2 = HF::sqr(x0), b1 = HF::sqr(x1i), b2 = HF::sqr(x0i);
  const host::H4 a3 = HF::mul(a1, a2), b3 = HF::mul(b1, b2);
  uint64_t sA[3][4], sB[3][4];
  memcpy(sA[0], a1.v, 32);
  memcpy(sA[1], a2.v, 32);
  memcpy(sA[2], a3.v, 32);
  memcpy(sB[0], b1.v, 32);
  memcpy(sB[1], b2.v, 32);
  memcpy(sB[2], b3.v, 32);
  TRY(fold_joint4_launch(s->curve, s->d_pts, sA, s->d_pts + 16ull * s->np, sB, (uint32_t)(s->n0 / 4)));
  const host::H4 fa = HF::mul(x0i, x1i), fb = HF::mul(x0, x1);
  mu_mul<SF>(s->muA, fa.v);
  mu_mul<SF>(s->muB, fb.v);
  s->mu_one = false;
  s->pending.clear();
  s->frozen = false;
  s->lazy = false;
  return 0;
}
int ipa_pair_fold(bpgpu_ipa* s) { return s->curve == BPGPU_PALLAS ? ipa_pair_fold_t<1>(s) : ipa_pair_fold_t<0>(s); }

// stop folding the generators: they stay as they are now (length s->len); the weights start at the
// common factors
int ipa_freeze(bpgpu_ipa* s) {
  s->n0 = s->len;
  s->frozen = true;
  s->lazy = false;
  s->d_table = nullptr;
  s->pending.clear();
  const uint32_t half = (uint32_t)s->freeze_cap;
  FoldScalars m = make_fold_scalars(s->muA, s->muB);
  if (s->curve == BPGPU_PALLAS)
    k_frozen_init<1><<<cdiv(2 * half, 256), 256, 0, cx().stream>>>(s->wA(), half, m);
  else
    k_frozen_init<0><<<cdiv(2 * half, 256), 256, 0, cx().stream>>>(s->wA(), half, m);
  HIP_TRY(hipGetLastError());
  return 0;
}
// Fixed-base table for the rounds that are left (the generators are frozen for good at length n0 = s->len).
// Worth its doubling chain (0.7 ms at 2^14 generators a side, 1.3 ms at 2^15: 255 dependent doublings per lane) only when many
// rounds follow, and only the pooled session builds one (the table lives in the context's buffers).
constexpr int TABLE_C = 15;             // 17 windows of 15 bits = the 255 bits of a Pasta scalar: no sparse top window
int ipa_build_table(bpgpu_ipa* s) {
  s->d_table = nullptr;
  const size_t min_len = g_ipa_table_min.load();  // default 2^12: twelve rounds left at least
  if (min_len == 0 || !s->pooled || !s->frozen || s->lazy || s->n0 < min_len || s->n0 < 2) return 0;
  Ctx& C = cx();
  // 18 rows: a scalar at or just below 2^254 (r - 1, say) carries out of window 16 into window 17
  const int c = TABLE_C, W = (256 + c - 1) / c;
  const uint32_t n0 = (uint32_t)s->n0, stride = 2 * n0 + 1;
  TRY(C.ipa_table.ensure((size_t)W * stride * 64));
  TRY(C.ipa_table_tmp.ensure((size_t)(W - 1) * stride * 96));
  uint32_t* tab = C.ipa_table.as<uint32_t>();
  uint32_t* tmp = C.ipa_table_tmp.as<uint32_t>();
  const uint32_t count = (uint32_t)(W - 1) * stride;
  if (s->curve == BPGPU_PALLAS) {
    k_table_chain<0><<<cdiv(stride, 256), 256, 0, C.stream>>>(s->d_pts, (uint32_t)s->np, n0, tab, tmp, c, W);
    k_table_norm<0><<<cdiv(cdiv(count, TABLE_NORM_BATCH), 256), 256, 0, C.stream>>>(tmp, tab, stride, count);
  } else {
    k_table_chain<1><<<cdiv(stride, 256), 256, 0, C.stream>>>(s->d_pts, (uint32_t)s->np, n0, tab, tmp, c, W);
    k_table_norm<1><<<cdiv(cdiv(count, TABLE_NORM_BATCH), 256), 256, 0, C.stream>>>(tmp, tab, stride, count);
  }
  HIP_TRY(hipGetLastError());
  s->d_table = tab;
  s->table_c = c;
  return 0;
}
// apply the deferred point folds one round at a time (s->len is the CURRENT vector length; the stored
// generators have length n0 and as many folds pending as halvings separate the two)
int ipa_replay_pending(bpgpu_ipa* s) {
  if (!s->frozen) return 0;
  const size_t final_len = s->len;
  size_t len = s->n0;
  int rc = 0;
  for (const auto& ch : s->pending) {
    s->len = len;  // ipa_fold_generators may need to normalise at the current length
    if ((rc = ipa_fold_generators(s, ch.data(), ch.data() + 4, (uint32_t)(len / 2))) != 0) break;
    len /= 2;
  }
  s->len = final_len;
  TRY(rc);
  s->pending.clear();
  s->frozen = false;
  s->lazy = false;
  return 0;
}
// export needs the folded generators themselves; the session folds every round from here on
int ipa_materialise(bpgpu_ipa* s) {
  if (!s->frozen) return 0;
  TRY(ipa_replay_pending(s));
  s->freeze_disabled = true;
  s->lazy_off = true;
  return 0;
}
}  // namespace

extern "C" {

const char* bpgpu_last_error(void) { return g_err.c_str(); }

// ---- contexts ------------------------------------------------------------------
}  // extern "C"

// Scratch of bpgpu_ip_device_noalloc: block partial sums of k_ip_partial (up to 1024 blocks x 32 B) followed
// by nothing else.  File-scope device memory, one copy per GPU for the life of the process; host code only
// ever uses the per-device address that ctx_init obtains with hipGetSymbolAddress.
constexpr uint32_t NOALLOC_IP_BLOCKS = 1024;
static __device__ __align__(256) uint32_t scratch[NOALLOC_IP_BLOCKS * 8];

namespace {
// caller holds C.mu (or is the only one who can see C)
int ctx_init_steps(Ctx& C, int device) {
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) return fail(BPGPU_E_ARG, "device %d out of range (%d visible)", device, count);
  HIP_TRY(hipSetDevice(device));
  C.device = device;
  HIP_TRY(hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking));
  for (auto& e : C.ev) HIP_TRY(hipEventCreate(&e));
  for (auto& e : C.op_ev) HIP_TRY(hipEventCreate(&e));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  C.num_cus = prop.multiProcessorCount;
  // this device's copy of the allocation-free entry point's scratch array (each GPU has its own address)
  void* sp = nullptr;
  HIP_TRY(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
  C.noalloc_scratch = (uint32_t*)sp;
  return 0;
}
// the stream and the events of a context (also of one whose initialisation stopped half-way)
void ctx_destroy_handles(Ctx& C) {
  for (auto& e : C.ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  for (auto& e : C.op_ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  if (C.stream) (void)hipStreamDestroy(C.stream);
  C.stream = nullptr;
  C.noalloc_scratch = nullptr;
}
int ctx_init(Ctx& C, int device) {
  const int rc = ctx_init_steps(C, device);
  if (rc != 0) {
    const std::string why = g_err;
    ctx_destroy_handles(C);
    g_err = why;
    return rc;
  }
  C.hpool.start(helper_threads());
  C.ready = true;
  return 0;
}

void ctx_release(Ctx& C) {
  if (!C.ready) return;
  (void)hipSetDevice(C.device);
  (void)hipStreamSynchronize(C.stream);
  for (DevBuf* b : {&C.counts, &C.offs, &C.order, &C.cursor, &C.canon, &C.ent_idx, &C.ent_fine, &C.sorted, &C.bsum, &C.partial,
                    &C.heavy, &C.bcount, &C.redtmp, &C.winsum, &C.st_pts_in, &C.st_pts, &C.st_sc, &C.st_out, &C.st_ip})
    b->release();
  for (DevBuf& b : C.aux) b.release();
  C.ipa_pts.release();
  C.ipa_sc.release();
  C.ipa_table.release();
  C.ipa_table_tmp.release();
  C.fold_alpha.release();
  C.pt_fmt.release();
  C.ipa_pool_busy = false;
  C.h_winsum.release();
  C.up.release();
  C.hpool.stop();
  ctx_destroy_handles(C);
  C.msm_timed = false;
  C.ready = false;
}
}  // namespace

extern "C" {

int bpgpu_init(int device) {
  std::lock_guard<std::mutex> lk(g_primary.mu);
  if (g_primary.ready) {
    if (g_primary.device != device) return fail(BPGPU_E_ARG, "already initialised on device %d", g_primary.device);
    return 0;
  }
  return ctx_init(g_primary, device);
}

// Open IPA sessions keep device memory of their context: shutting down under them would leave them with
// dangling pointers, so the call is refused (returns without releasing anything) until they are freed.
void bpgpu_shutdown(void) {
  std::lock_guard<std::mutex> run(g_shard_run_mu);
  std::lock_guard<std::mutex> reg(g_reg_mu);
  {
    std::lock_guard<std::mutex> lk(g_primary.mu);
    if (g_primary.open_sessions > 0) {
      (void)fail(BPGPU_E_STATE, "bpgpu_shutdown: %d IPA session(s) still open; free them first", g_primary.open_sessions);
      return;
    }
  }
  g_shard_pool.stop();
  for (auto& c : g_shards) {
    std::lock_guard<std::mutex> lk(c->mu);
    ctx_release(*c);
  }
  g_shards.clear();
  g_ctx_count.store(1);
  std::lock_guard<std::mutex> lk(g_primary.mu);
  ctx_release(g_primary);
}

int bpgpu_set_window_bits(int c) {
  if (c != 0 && (c < 2 || c > 20)) return fail(BPGPU_E_ARG, "window bits %d outside [2, 20]", c);
  g_window_bits.store(c);
  return 0;
}

int bpgpu_set_freeze_len(size_t len) {
  if (len > (1u << 24)) return fail(BPGPU_E_ARG, "freeze length too large");
  g_freeze_len.store(len);
  return 0;
}

int bpgpu_set_pair_fold(int enable) {
  g_pair_fold.store(enable != 0);
  return 0;
}

int bpgpu_set_ipa_table(size_t min_len) {
  g_ipa_table_min.store(min_len == 1 ? DEFAULT_IPA_TABLE_MIN : min_len);
  return 0;
}

void bpgpu_reset_tuning(void) {
  g_window_bits.store(0);
  g_freeze_len.store(DEFAULT_FREEZE_LEN);
  g_pair_fold.store(true);
  g_ipa_table_min.store(DEFAULT_IPA_TABLE_MIN);
}

int bpgpu_get_tuning(size_t out[4]) {
  if (!out) return fail(BPGPU_E_ARG, "null pointer");
  out[0] = (size_t)g_window_bits.load();
  out[1] = g_freeze_len.load();
  out[2] = g_pair_fold.load() ? 1 : 0;
  out[3] = g_ipa_table_min.load();
  return 0;
}

int bpgpu_dev_alloc(size_t bytes, void** dptr) {
  CtxLock lk;
  TRY(check_ready());
  if (!dptr) return fail(BPGPU_E_ARG, "null output pointer");
  HIP_TRY(hipSetDevice(cx().device));
  HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
  return 0;
}
int bpgpu_dev_free(void* dptr) {
  CtxLock lk;
  TRY(check_ready());
  HIP_TRY(hipStreamSynchronize(cx().stream));
  HIP_TRY(hipFree(dptr));
  return 0;
}
int bpgpu_dev_upload(void* dptr, const void* host, size_t bytes) {
  CtxLock lk;
  TRY(check_ready());
  if (bytes == 0) return 0;
  HIP_TRY(upload_async(dptr, host, bytes, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}
int bpgpu_dev_download(void* host, const void* dptr, size_t bytes) {
  CtxLock lk;
  TRY(check_ready());
  if (bytes == 0) return 0;
  HIP_TRY(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_points_to_device(int curve, const uint64_t* points, size_t n, int fmt, void* d_out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (fmt != 0 && fmt != 1) return fail(BPGPU_E_ARG, "fmt must be 0 (jacobian) or 1 (affine)");
  if (n == 0) return 0;
  if (!points || !d_out) return fail(BPGPU_E_ARG, "null pointer");
  TRY(upload_points(curve, points, n, fmt, (uint32_t*)d_out, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_points_from_device(int curve, const void* d_points, size_t n, uint64_t* out_xyz) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n == 0) return 0;
  if (!d_points || !out_xyz) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_out.ensure(n * 96));
  TRY(points_out(curve, (const uint32_t*)d_points, cx().st_out.as<uint32_t>(), n));
  HIP_TRY(hipMemcpyAsync(out_xyz, cx().st_out.p, n * 96, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_msm_device(int curve, const void* d_points, const void* d_scalars, size_t n, uint64_t out_xy_inf[9]) {
  if (n > BPGPU_MSM_MAX_N) return fail(BPGPU_E_ARG, "n = %zu: one MSM call takes at most 2^28 points", n);
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!out_xy_inf) return fail(BPGPU_E_ARG, "null output");
  if (n && (!d_points || !d_scalars)) return fail(BPGPU_E_ARG, "null pointer");
  return msm_dispatch(curve, (const uint32_t*)d_points, (const uint32_t*)d_scalars, single_plan((uint32_t)n),
                      (uint64_t(*)[9])out_xy_inf);
}

static int msm_host(int curve, const uint64_t* points, int fmt, const uint64_t* scalars, size_t n,
                    uint64_t out_xy_inf[9]) {
  if (n > BPGPU_MSM_MAX_N) return fail(BPGPU_E_ARG, "n = %zu: one MSM call takes at most 2^28 points", n);
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!out_xy_inf) return fail(BPGPU_E_ARG, "null output");
  if (n && (!points || !scalars)) return fail(BPGPU_E_ARG, "null pointer");
  // Scalars first, then the points in chunks on the staging threads' streams (pt_upload_begin): the sort needs the
  // scalars only and is enqueued while the -- two to three times larger -- point array is still crossing PCIe; the
  // accumulation is ordered after the last chunk's conversion (points_ready -> pt_upload_wait).
  const bool piped = n * (size_t)(fmt == 0 ? 96 : 64) >= Uploader::MIN_BYTES;
  if (n) {
    TRY(cx().st_pts.ensure(n * 64));
    TRY(cx().st_sc.ensure(n * 32));
    HIP_TRY(upload_async(cx().st_sc.p, scalars, n * 32, cx().stream));
    if (piped) {
      PtSeg seg{points, n, fmt, cx().st_pts.as<uint32_t>(), 0};
      TRY(pt_upload_begin(curve, &seg, 1, cx().stream));
    }
  }
  const std::function<int()> points_ready = [&]() -> int {
    if (!n) return 0;
    if (piped) return pt_upload_wait(0, cx().stream);
    return upload_points(curve, points, n, fmt, cx().st_pts.as<uint32_t>(), cx().stream);
  };
  const int rc = msm_dispatch(curve, cx().st_pts.as<uint32_t>(), cx().st_sc.as<uint32_t>(), single_plan((uint32_t)n),
                              (uint64_t(*)[9])out_xy_inf, &points_ready);
  const int rc_up = pt_upload_end();  // the caller's arrays are read until here, whatever happened above
  return rc ? rc : rc_up;
}

int bpgpu_msm(int curve, const uint64_t* points_xyz, const uint64_t* scalars, size_t n, uint64_t out_xy_inf[9]) {
  return msm_host(curve, points_xyz, 0, scalars, n, out_xy_inf);
}
int bpgpu_msm_affine(int curve, const uint64_t* points_xy, const uint64_t* scalars, size_t n,
                     uint64_t out_xy_inf[9]) {
  return msm_host(curve, points_xy, 1, scalars, n, out_xy_inf);
}

int bpgpu_ip_device(int curve, const void* d_a, const void* d_b, size_t n, uint64_t out[4]) {
  if (n > 0xffffffffu) return fail(BPGPU_E_ARG, "n too large");
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!out) return fail(BPGPU_E_ARG, "null output");
  if (n && (!d_a || !d_b)) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_out.ensure(64));
  IpPair p0{0u, 0u};
  TRY(ip_launch(curve, (const uint32_t*)d_a, (const uint32_t*)d_b, p0, p0, 1, (uint32_t)n, cx().st_out.as<uint32_t>(),
                cx().st_out.as<uint32_t>() + 8));
  HIP_TRY(hipMemcpyAsync(out, cx().st_out.p, 32, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_ip(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
  {
    CtxLock lk;
    TRY(check_ready());
    TRY(check_curve(curve));
    if (!out) return fail(BPGPU_E_ARG, "null output");
    if (n > 0x7fffffffu) return fail(BPGPU_E_ARG, "n too large");
    if (n && (!a || !b)) return fail(BPGPU_E_ARG, "null pointer");
    TRY(cx().st_sc.ensure(2 * n * 32 + 32));
    TRY(cx().st_out.ensure(64));
    if (n) {
      HIP_TRY(upload_async(cx().st_sc.p, a, n * 32, cx().stream));
      HIP_TRY(upload_async((uint8_t*)cx().st_sc.p + n * 32, b, n * 32, cx().stream));
    }
    IpPair p0{0u, (uint32_t)n};
    TRY(ip_launch(curve, cx().st_sc.as<uint32_t>(), cx().st_sc.as<uint32_t>(), p0, p0, 1, (uint32_t)n,
                  cx().st_out.as<uint32_t>(), cx().st_out.as<uint32_t>() + 8));
    HIP_TRY(hipMemcpyAsync(out, cx().st_out.p, 32, hipMemcpyDeviceToHost, cx().stream));
    HIP_TRY(hipStreamSynchronize(cx().stream));
  }
  return 0;
}

int bpgpu_fold_scalars_device(int curve, const void* d_v, size_t n_out, const uint64_t s_lo[4],
                              const uint64_t s_hi[4], void* d_out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n_out > 0x7fffffffu) return fail(BPGPU_E_ARG, "n too large");
  if (n_out && (!d_v || !d_out || !s_lo || !s_hi)) return fail(BPGPU_E_ARG, "null pointer");
  FoldScalars f = make_fold_scalars(s_lo, s_hi);
  TRY(fold_scalars_launch(curve, (uint32_t*)d_v, (uint32_t*)d_out, f, (uint32_t*)d_v, (uint32_t*)d_out, f, 1,
                          (uint32_t)n_out));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_fold_scalars(int curve, const uint64_t* v, size_t n_out, const uint64_t s_lo[4], const uint64_t s_hi[4],
                       uint64_t* out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n_out > 0x3fffffffu) return fail(BPGPU_E_ARG, "n too large");
  if (n_out == 0) return 0;
  if (!v || !out || !s_lo || !s_hi) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_sc.ensure(2 * n_out * 32));
  HIP_TRY(upload_async(cx().st_sc.p, v, 2 * n_out * 32, cx().stream));
  FoldScalars f = make_fold_scalars(s_lo, s_hi);
  uint32_t* d = cx().st_sc.as<uint32_t>();
  TRY(fold_scalars_launch(curve, d, d, f, d, d, f, 1, (uint32_t)n_out));
  HIP_TRY(hipMemcpyAsync(out, d, n_out * 32, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_fold_points_device(int curve, const void* d_g, size_t n_out, const uint64_t s_lo[4],
                             const uint64_t s_hi[4], void* d_out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n_out > 0x7fffffffu) return fail(BPGPU_E_ARG, "n too large");
  if (n_out && (!d_g || !d_out || !s_lo || !s_hi)) return fail(BPGPU_E_ARG, "null pointer");
  TRY(fold_points_launch(curve, (const uint32_t*)d_g, (uint32_t*)d_out, s_lo, s_hi, (const uint32_t*)d_g,
                         (uint32_t*)d_out, s_lo, s_hi, 1, (uint32_t)n_out));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_fold_points(int curve, const uint64_t* g_xyz, size_t n_out, const uint64_t s_lo[4], const uint64_t s_hi[4],
                      uint64_t* out_xyz) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n_out > 0x3fffffffu) return fail(BPGPU_E_ARG, "n too large");
  if (n_out == 0) return 0;
  if (!g_xyz || !out_xyz || !s_lo || !s_hi) return fail(BPGPU_E_ARG, "null pointer");
  size_t n2 = 2 * n_out;
  TRY(cx().st_pts_in.ensure(n2 * 96));
  TRY(cx().st_pts.ensure(n2 * 64));
  TRY(upload_points(curve, g_xyz, n2, 0, cx().st_pts.as<uint32_t>(), cx().stream));  // chunked, Z = 1 rows compacted
  uint32_t* d = cx().st_pts.as<uint32_t>();
  TRY(fold_points_launch(curve, d, d, s_lo, s_hi, d, d, s_lo, s_hi, 1, (uint32_t)n_out));
  TRY(points_out(curve, d, cx().st_pts_in.as<uint32_t>(), n_out));
  HIP_TRY(hipMemcpyAsync(out_xyz, cx().st_pts_in.p, n_out * 96, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

// ---- verifier MSM of Bp::verify ----------------------------------------------------
extern "C++" {
namespace {
template <int SF>
int verify_tables(const uint64_t* challenges, size_t k, const uint64_t final_a[4], const uint64_t final_b[4],
                  std::vector<uint64_t>& tab, std::vector<uint64_t>& tail) {
  using HFs = host::HF<SF>;
  auto ld = [](const uint64_t* p) {
    host::H4 r;
    memcpy(r.v, p, 32);
    return r;
  };
  host::H4 fa = ld(final_a), fb = ld(final_b), base_a = fa, base_b = fb;
  tab.assign((2 + 2 * k) * 4, 0);
  tail.assign((1 + 2 * k) * 4, 0);
  host::H4 ab = HFs::mul(fa, fb);
  memcpy(&tail[0], ab.v, 32);
  for (size_t j = 0; j < k; j++) {
    host::H4 x = ld(challenges + 4 * j);
    if (HFs::is_zero(x)) return fail(BPGPU_E_ARG, "challenge %zu is zero (no inverse)", j);
    host::H4 xi = HFs::inv(x), x2 = HFs::sqr(x), xi2 = HFs::sqr(xi);
    base_a = HFs::mul(base_a, xi);
    base_b = HFs::mul(base_b, x);
    const size_t t = k - 1 - j;  // bit of the index this challenge acts on
    memcpy(&tab[(2 + t) * 4], x2.v, 32);
    memcpy(&tab[(2 + k + t) * 4], xi2.v, 32);
    host::H4 n2 = HFs::neg(x2), ni2 = HFs::neg(xi2);  // -x^2 on L_j, -x^-2 on R_j (bp_iter.rs:166-173)
    memcpy(&tail[(1 + 2 * j) * 4], n2.v, 32);
    memcpy(&tail[(2 + 2 * j) * 4], ni2.v, 32);
  }
  memcpy(&tab[0], base_a.v, 32);
  memcpy(&tab[4], base_b.v, 32);
  return 0;
}

// points: A | B | Q | L_0 R_0 L_1 R_1 ... ; gens from the host (jacobian) or the device (dev affine)
int verify_msm_common(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz, const uint64_t* ip_gen_xyz,
                      const void* d_a, const void* d_b, const void* d_q, size_t n, const uint64_t* ls_xyz,
                      const uint64_t* rs_xyz, size_t k, const uint64_t* challenges, const uint64_t final_a[4],
                      const uint64_t final_b[4], uint64_t out_xy_inf[9]) {
  Ctx& C = cx();
  const bool from_device = d_q != nullptr;
  bool piped = false;  // a chunked generator upload is in flight
  PipeGuard pipe_guard{piped};
  if (k > 30) return fail(BPGPU_E_ARG, "too many rounds");
  const size_t np = (size_t)1 << k;
  if (pow2_at_least(n < 1 ? 1 : n) != np) return fail(BPGPU_E_ARG, "proof has %zu rounds but the generators need %zu elements", k, n);
  const size_t total = 2 * np + 1 + 2 * k;
  if (total > BPGPU_MSM_MAX_N) return fail(BPGPU_E_ARG, "verifier MSM of %zu points: at most 2^28", total);
  std::vector<uint64_t> tab, tail;
  TRY(curve == BPGPU_PALLAS ? verify_tables<1>(challenges, k, final_a, final_b, tab, tail)
                            : verify_tables<0>(challenges, k, final_a, final_b, tab, tail));
  TRY(C.st_pts.ensure(total * 64));
  TRY(C.st_sc.ensure(total * 32));
  TRY(C.st_ip.ensure(tab.size() * 8));
  hipStream_t st = C.stream;
  uint32_t* dp = C.st_pts.as<uint32_t>();
  uint32_t* ds = C.st_sc.as<uint32_t>();
  HIP_TRY(upload_async(C.st_ip.p, tab.data(), tab.size() * 8, st));
  HIP_TRY(upload_async(ds + 8ull * 2 * np, tail.data(), tail.size() * 8, st));
  // generators, zero-padded to np with identity points (bp_iter.rs:137-138)
  if (np > n) {
    HIP_TRY(hipMemsetAsync(dp + 16ull * n, 0, (np - n) * 64, st));
    HIP_TRY(hipMemsetAsync(dp + 16ull * (np + n), 0, (np - n) * 64, st));
  }
  // staging for the jacobian inputs: the proof's points always, the generators when they come from the host
  const size_t stage_pts = (from_device ? 0 : 2 * n + 1) + 2 * k;
  TRY(C.st_pts_in.ensure((stage_pts ? stage_pts : 1) * 96));
  uint8_t* stg = (uint8_t*)C.st_pts_in.p;
  size_t so = 0;
  if (from_device) {
    if (n) {
      HIP_TRY(hipMemcpyAsync(dp, d_a, n * 64, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpyAsync(dp + 16ull * np, d_b, n * 64, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(dp + 16ull * 2 * np, d_q, 64, hipMemcpyDeviceToDevice, st));
  } else {
    if (n) {
      PtSeg segs[2] = {{a_gens_xyz, n, 0, dp, 0}, {b_gens_xyz, n, 0, dp + 16ull * np, 0}};
      if (n * 96 >= Uploader::MIN_BYTES) {
        // the staging area is st_pts_in: the proof's few points below go behind the generators' 2 n slots
        TRY(C.st_pts_in.ensure((2 * n + 1 + 2 * k) * 96));
        stg = (uint8_t*)C.st_pts_in.p;
        // the chunks cross PCIe on the staging threads' streams while this stream expands the scalars and sorts them:
        // the MSM's accumulation is what waits for the last chunk (points_ready below)
        TRY(pt_upload_begin(curve, segs, 2, st));
        piped = true;
      } else {
        HIP_TRY(upload_async(stg, a_gens_xyz, n * 96, st));
        HIP_TRY(upload_async(stg + n * 96, b_gens_xyz, n * 96, st));
        TRY(points_in(curve, (uint32_t*)stg, dp, n, 0));
        TRY(points_in(curve, (uint32_t*)(stg + n * 96), dp + 16ull * np, n, 0));
      }
    }
    HIP_TRY(upload_async(stg + 2 * n * 96, ip_gen_xyz, 96, st));
    TRY(points_in(curve, (uint32_t*)(stg + 2 * n * 96), dp + 16ull * 2 * np, 1, 0));
    so = (2 * n + 1) * 96;
  }
  if (k) {
    // interleave L_j, R_j on the host side of the staging copy
    std::vector<uint64_t> lr(2 * k * 12);
    for (size_t j = 0; j < k; j++) {
      memcpy(&lr[(2 * j) * 12], ls_xyz + 12 * j, 96);
      memcpy(&lr[(2 * j + 1) * 12], rs_xyz + 12 * j, 96);
    }
    HIP_TRY(upload_async(stg + so, lr.data(), 2 * k * 96, st));
    HIP_TRY(hipStreamSynchronize(st));  // lr, tab and tail live on this frame
    TRY(points_in(curve, (uint32_t*)(stg + so), dp + 16ull * (2 * np + 1), 2 * k, 0));
  } else {
    HIP_TRY(hipStreamSynchronize(st));
  }
  uint32_t gx = cdiv(np, 256);
  if (gx > (uint32_t)C.num_cus * 8) gx = (uint32_t)C.num_cus * 8;
  if (curve == BPGPU_PALLAS)
    k_verify_scalars<1><<<dim3(gx, 2), 256, 0, st>>>(C.st_ip.as<uint32_t>(), ds, (uint32_t)np, (int)k);
  else
    k_verify_scalars<0><<<dim3(gx, 2), 256, 0, st>>>(C.st_ip.as<uint32_t>(), ds, (uint32_t)np, (int)k);
  HIP_TRY(hipGetLastError());
  if (!piped) return msm_dispatch(curve, dp, ds, single_plan((uint32_t)total), (uint64_t(*)[9])out_xy_inf);
  const std::function<int()> ready = [&]() -> int { return pt_upload_wait(0, st); };
  const int rc = msm_dispatch(curve, dp, ds, single_plan((uint32_t)total), (uint64_t(*)[9])out_xy_inf, &ready);
  piped = false;
  const int rc_up = pt_upload_end();
  return rc ? rc : rc_up;
}
}  // namespace
}  // extern "C++"

int bpgpu_ipa_verify_msm(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz, const uint64_t* ip_gen_xyz,
                         size_t n, const uint64_t* ls_xyz, const uint64_t* rs_xyz, size_t k, const uint64_t* challenges,
                         const uint64_t final_a[4], const uint64_t final_b[4], uint64_t out_xy_inf[9]) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!ip_gen_xyz || !final_a || !final_b || !out_xy_inf) return fail(BPGPU_E_ARG, "null pointer");
  if (n && (!a_gens_xyz || !b_gens_xyz)) return fail(BPGPU_E_ARG, "null pointer");
  if (k && (!ls_xyz || !rs_xyz || !challenges)) return fail(BPGPU_E_ARG, "null pointer");
  return verify_msm_common(curve, a_gens_xyz, b_gens_xyz, ip_gen_xyz, nullptr, nullptr, nullptr, n, ls_xyz, rs_xyz, k,
                           challenges, final_a, final_b, out_xy_inf);
}

int bpgpu_ipa_verify_msm_device(int curve, const void* d_a_gens, const void* d_b_gens, const void* d_ip_gen, size_t n,
                                const uint64_t* ls_xyz, const uint64_t* rs_xyz, size_t k, const uint64_t* challenges,
                                const uint64_t final_a[4], const uint64_t final_b[4], uint64_t out_xy_inf[9]) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!d_ip_gen || !final_a || !final_b || !out_xy_inf) return fail(BPGPU_E_ARG, "null pointer");
  if (n && (!d_a_gens || !d_b_gens)) return fail(BPGPU_E_ARG, "null pointer");
  if (k && (!ls_xyz || !rs_xyz || !challenges)) return fail(BPGPU_E_ARG, "null pointer");
  return verify_msm_common(curve, nullptr, nullptr, nullptr, d_a_gens, d_b_gens, d_ip_gen, n, ls_xyz, rs_xyz, k, challenges,
                           final_a, final_b, out_xy_inf);
}

// ---- vector utilities of the k-ary protocols ----------------------------------
extern "C++" {
namespace {
template <int SF>
bool geometric_from_one(const uint64_t* y, size_t k) {
  using HFs = host::HF<SF>;
  host::H4 prev, r, cur;
  memcpy(prev.v, y, 32);
  if (!HFs::eq(prev, HFs::one())) return false;
  memcpy(r.v, y + 4, 32);
  prev = r;
  for (size_t j = 2; j < k; j++) {
    memcpy(cur.v, y + 4 * j, 32);
    prev = HFs::mul(prev, r);
    if (!HFs::eq(prev, cur)) return false;
  }
  return true;
}

int fold_scalars_kary_launch(int curve, const uint32_t* d_v, size_t ck, size_t k, const uint64_t* y, uint32_t* d_out) {
  Ctx& C = cx();
  if (k > 1024) return fail(BPGPU_E_ARG, "k = %zu chunks: at most 1024 are supported", k);
  if (ck == 0) return 0;
  if (k == 0) {  // sum_vecs over no summands: vec![zero; len]
    HIP_TRY(hipMemsetAsync(d_out, 0, ck * 32, C.stream));
    return 0;
  }
  TRY(C.st_ip.ensure(k * 32));
  HIP_TRY(upload_async(C.st_ip.p, y, k * 32, C.stream));
  uint32_t gx = cdiv(ck, 256);
  const uint32_t cap = (uint32_t)C.num_cus * 16;
  if (gx > cap) gx = cap;
  OpSpan span(cx());
  if (curve == BPGPU_PALLAS)
    k_fold_scalars_kary<1><<<gx, 256, k * 36, C.stream>>>(d_v, C.st_ip.as<uint32_t>(), d_out, (uint32_t)ck, (uint32_t)k);
  else
    k_fold_scalars_kary<0><<<gx, 256, k * 36, C.stream>>>(d_v, C.st_ip.as<uint32_t>(), d_out, (uint32_t)ck, (uint32_t)k);
  HIP_TRY(hipGetLastError());
  return 0;
}

// d_out[i] = sum_j y[j] d_g[j ck + i]; d_out (ck points) must not overlap d_g.  Three chunks and more: the multipliers
// ride one doubling chain, seven at a time (k_fold_multi: k = 8 is ONE launch of 128 doublings + ~380 additions per
// output where Horner's rule over k_fold_monic ran seven chains of 128 + 55; 5.2 -> 2.4 ms at ck = 2^16, 56 -> 27.5 ms at
// 2^20).  Two chunks (or a schedule that does not fit the kernel's list): one shared-scalar multiplication per chunk
// (k_fold_monic); when y = (1, r, r^2, ...) -- the only shape the reference uses, powers(x, 0..k) -- Horner's rule
// needs k - 1 of them:  G_0 + r (G_1 + r (G_2 + ...)).
int fold_points_kary_launch(int curve, const uint32_t* d_g, size_t ck, size_t k, const uint64_t* y, uint32_t* d_out) {
  Ctx& C = cx();
  if (ck == 0) return 0;
  if (k == 0) {
    HIP_TRY(hipMemsetAsync(d_out, 0, ck * 64, C.stream));  // identity points
    return 0;
  }
  const uint32_t n = (uint32_t)ck;
  OpSpan span(C);  // the whole fold (however many launches) is one timed operation
  auto chunk = [&](size_t j) { return d_g + 16ull * j * ck; };
  // Two multipliers or more: groups of up to seven ride one doubling chain each (k_fold_multi); chunk 0 is the first
  // group's addend when y[0] = 1 (the reference always passes powers(x, 0..k)), the output so far the later groups'.
  host::H4 y0;
  memcpy(y0.v, y, 32);
  const bool monic = curve == BPGPU_PALLAS ? host::HF<1>::eq(y0, host::HF<1>::one()) : host::HF<0>::eq(y0, host::HF<0>::one());
  // (two chunks with y[0] = 1 are one k_fold_monic launch, operands in registers; with y[0] != 1 they are two of them, or
  // one shared chain here)
  if (k >= 3 || (k == 2 && !monic)) {
    C.fold_ops[0] = C.fold_ops[1] = C.fold_ops[2] = 0;
    size_t j = monic ? 1 : 0;
    const uint32_t* addend = monic ? chunk(0) : nullptr;
    bool ok = true;
    while (j < k && ok) {
      const int m = (int)(k - j < (size_t)MULTI_MAX_M ? k - j : (size_t)MULTI_MAX_M);
      const int rc = fold_multi(curve, addend, chunk(j), y + 4 * j, m, n, d_out);
      if (rc < 0 || rc > 1) return rc;
      if (rc == 1) ok = false;
      addend = d_out;
      j += (size_t)m;
    }
    if (ok) {
      C.fold_ops[0] = n;
      return 0;
    }
    // (a schedule that does not fit the operation list: never with 136-position expansions; fold one at a time instead)
  }
  const bool geo = k >= 2 && (curve == BPGPU_PALLAS ? geometric_from_one<1>(y, k) : geometric_from_one<0>(y, k));
  if (geo) {
    const uint64_t* r = y + 4;
    TRY(fold_monic_pq(curve, chunk(k - 2), chunk(k - 1), d_out, r, nullptr, nullptr, nullptr, r, 1, n, 1));
    for (size_t j = k - 2; j-- > 0;) TRY(fold_monic_pq(curve, chunk(j), d_out, d_out, r, nullptr, nullptr, nullptr, r, 1, n, 1));
    return 0;
  }
  TRY(fold_monic_pq(curve, d_out, chunk(k - 1), d_out, y + 4 * (k - 1), nullptr, nullptr, nullptr, y, 1, n, 0));
  for (size_t j = k - 1; j-- > 0;)
    TRY(fold_monic_pq(curve, d_out, chunk(j), d_out, y + 4 * j, nullptr, nullptr, nullptr, y, 1, n, 1));
  return 0;
}
}  // namespace
}  // extern "C++"

int bpgpu_fold_scalars_kary_device(int curve, const void* d_v, size_t ck, size_t k, const uint64_t* y, void* d_out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (ck > 0x7fffffffu || ck * (k ? k : 1) > 0xffffffffull) return fail(BPGPU_E_ARG, "vector too large");
  if (ck && (!d_out || (k && (!d_v || !y)))) return fail(BPGPU_E_ARG, "null pointer");
  TRY(fold_scalars_kary_launch(curve, (const uint32_t*)d_v, ck, k, y, (uint32_t*)d_out));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_fold_scalars_kary(int curve, const uint64_t* v, size_t ck, size_t k, const uint64_t* y, uint64_t* out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (ck > 0x7fffffffu || ck * (k ? k : 1) > 0x3fffffffull) return fail(BPGPU_E_ARG, "vector too large");
  if (ck == 0) return 0;
  if (!out || (k && (!v || !y))) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_sc.ensure((k ? k : 1) * ck * 32));
  if (k) HIP_TRY(upload_async(cx().st_sc.p, v, k * ck * 32, cx().stream));
  uint32_t* d = cx().st_sc.as<uint32_t>();
  TRY(fold_scalars_kary_launch(curve, d, ck, k, y, d));
  HIP_TRY(hipMemcpyAsync(out, d, ck * 32, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_fold_points_kary_device(int curve, const void* d_g, size_t ck, size_t k, const uint64_t* y, void* d_out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (ck > 0x7fffffffu || ck * (k ? k : 1) > 0x7fffffffull) return fail(BPGPU_E_ARG, "vector too large");
  if (ck && (!d_out || (k && (!d_g || !y)))) return fail(BPGPU_E_ARG, "null pointer");
  if (ck && k) {
    const uint8_t *g0 = (const uint8_t*)d_g, *o0 = (const uint8_t*)d_out;
    if (o0 < g0 + k * ck * 64 && g0 < o0 + ck * 64) return fail(BPGPU_E_ARG, "d_out overlaps d_g");
  }
  TRY(fold_points_kary_launch(curve, (const uint32_t*)d_g, ck, k, y, (uint32_t*)d_out));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_fold_points_kary(int curve, const uint64_t* g_xyz, size_t ck, size_t k, const uint64_t* y, uint64_t* out_xyz) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (ck > 0x7fffffffu || ck * (k ? k : 1) > 0x3fffffffull) return fail(BPGPU_E_ARG, "vector too large");
  if (ck == 0) return 0;
  if (!out_xyz || (k && (!g_xyz || !y))) return fail(BPGPU_E_ARG, "null pointer");
  const size_t tot = k * ck;
  TRY(cx().st_pts_in.ensure((tot > ck ? tot : ck) * 96));
  TRY(cx().st_pts.ensure((tot + ck) * 64));
  uint32_t* d = cx().st_pts.as<uint32_t>();
  uint32_t* d_o = d + 16ull * tot;
  if (tot) {
    TRY(upload_points(curve, g_xyz, tot, 0, d, cx().stream));  // chunked, Z = 1 rows compacted
  }
  TRY(fold_points_kary_launch(curve, d, ck, k, y, d_o));
  TRY(points_out(curve, d_o, cx().st_pts_in.as<uint32_t>(), ck));
  HIP_TRY(hipMemcpyAsync(out_xyz, cx().st_pts_in.p, ck * 96, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

// ---- cross terms of a k-ary round -----------------------------------------------------------------------------
// X[i, j] = msm(a_gen[j] | b_gen[i] | Q,  a[i] | b[j] | <a[i], b[j]>) over chunks of length ck
// (src/reductions/ipa_to_bp_unroll.rs:189-194 == src/proofs/bp_rec_kary.rs:52-60), and
//   pos[t] = sum_{j < k-t-1} X[t+j+1, j],   neg[t] = sum_{j < k-t-1} X[j, t+j+1],   t < k - 1   (:198-215 == :64-76).
// A sum of MSMs is an MSM over the concatenated operands, and here the concatenations are CONTIGUOUS ranges of the padded
// vectors: with off = (t+1) ck and m = (k-t-1) ck,
//   pos[t] = msm(A[..m] | B[off..] | Q,  a[off..] | b[..m] | <a[off..], b[..m]>)
//   neg[t] = msm(A[off..] | B[..m] | Q,  a[..m] | b[off..] | <a[..m], b[off..]>)
// -- the three-segment operand form of the prover's L and R.  So the k (k - 1) MSMs of 2 ck + 1 terms (and as many inner
// products) become 2 (k - 1) MSMs, all of them in ONE batched launch sequence (MsmPlan with M = 2 (k - 1) <= 32; more go in
// several batches), over operands uploaded once.
int bpgpu_kary_cross_terms(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz, const uint64_t* ip_gen_xyz,
                           const uint64_t* a, const uint64_t* b, size_t ck, size_t k, uint64_t* pos_xy_inf,
                           uint64_t* neg_xy_inf) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (k < 2) return 0;  // no cross terms
  if (k > 4096 || ck > ((size_t)1 << 26) || ck * k > ((size_t)1 << 26)) return fail(BPGPU_E_ARG, "k-ary round too large");
  if (!ip_gen_xyz || !pos_xy_inf || !neg_xy_inf) return fail(BPGPU_E_ARG, "null pointer");
  if (ck && (!a_gens_xyz || !b_gens_xyz || !a || !b)) return fail(BPGPU_E_ARG, "null pointer");
  Ctx& C = cx();
  const size_t N = ck * k;
  const uint32_t Nu = (uint32_t)N;
  TRY(C.st_pts.ensure((2 * N + 1) * 64));
  TRY(C.st_sc.ensure((2 * N + 2 * (k - 1)) * 32));
  uint32_t* dp = C.st_pts.as<uint32_t>();
  uint32_t* ds = C.st_sc.as<uint32_t>();
  hipStream_t st = C.stream;
  bool piped = false;  // the generators' chunked upload is in flight
  PipeGuard pipe_guard{piped};
  TRY(upload_points(curve, ip_gen_xyz, 1, 0, dp + 32 * N, st));
  if (N) {
    HIP_TRY(upload_async(ds, a, N * 32, st));
    HIP_TRY(upload_async(ds + 8 * N, b, N * 32, st));
    if (N * 96 >= Uploader::MIN_BYTES) {
      // the generators cross PCIe while the inner products and the first batch's sort run on the scalars: that batch's
      // accumulation is what waits for the last chunk
      PtSeg segs[2] = {{a_gens_xyz, N, 0, dp, 0}, {b_gens_xyz, N, 0, dp + 16 * N, 0}};
      TRY(pt_upload_begin(curve, segs, 2, st));
      piped = true;
    } else {
      TRY(upload_points(curve, a_gens_xyz, N, 0, dp, st));
      TRY(upload_points(curve, b_gens_xyz, N, 0, dp + 16 * N, st));
    }
  }
  // the 2 (k - 1) summed inner products, two per launch
  for (size_t t = 0; t + 1 < k; t++) {
    const uint32_t off = (uint32_t)((t + 1) * ck), m = Nu - off;
    IpPair pp{off, Nu}, pn{0u, Nu + off};
    TRY(ip_launch(curve, ds, ds, pp, pn, 2, m, ds + 8ull * (2 * N + 2 * t), ds + 8ull * (2 * N + 2 * t + 1)));
  }
  static thread_local uint64_t out[MSM_MAX_BATCH][9];
  for (size_t t0 = 0; t0 + 1 < k; t0 += MSM_MAX_BATCH / 2) {
    const size_t cnt = (k - 1 - t0) < (size_t)(MSM_MAX_BATCH / 2) ? (k - 1 - t0) : (size_t)(MSM_MAX_BATCH / 2);
    MsmPlan plan;
    memset(&plan, 0, sizeof plan);
    plan.M = (int)(2 * cnt);
    plan.n = 2 * (Nu - (uint32_t)((t0 + 1) * ck)) + 1;  // the longest of the batch
    for (size_t i = 0; i < cnt; i++) {
      const size_t t = t0 + i;
      const uint32_t off = (uint32_t)((t + 1) * ck), m = Nu - off;
      plan.seg[2 * i][0] = MsmSeg{0u, off, m};
      plan.seg[2 * i][1] = MsmSeg{Nu + off, Nu, m};
      plan.seg[2 * i][2] = MsmSeg{2 * Nu, (uint32_t)(2 * N + 2 * t), 1};
      plan.seg[2 * i + 1][0] = MsmSeg{off, 0u, m};
      plan.seg[2 * i + 1][1] = MsmSeg{Nu, Nu + off, m};
      plan.seg[2 * i + 1][2] = MsmSeg{2 * Nu, (uint32_t)(2 * N + 2 * t + 1), 1};
    }
    if (piped) {
      const std::function<int()> ready = [&]() -> int { return pt_upload_wait(0, st); };
      const int rc = msm_dispatch(curve, dp, ds, plan, out, &ready);
      piped = false;
      const int rc_up = pt_upload_end();
      TRY(rc);
      TRY(rc_up);
    } else {
      TRY(msm_dispatch(curve, dp, ds, plan, out));
    }
    for (size_t i = 0; i < cnt; i++) {
      memcpy(pos_xy_inf + 9 * (t0 + i), out[2 * i], 72);
      memcpy(neg_xy_inf + 9 * (t0 + i), out[2 * i + 1], 72);
    }
  }
  return 0;
}

int bpgpu_hadamard(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n > 0x3fffffffu) return fail(BPGPU_E_ARG, "n too large");
  if (n == 0) return 0;
  if (!a || !b || !out) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_sc.ensure(2 * n * 32));
  uint32_t* d = cx().st_sc.as<uint32_t>();
  HIP_TRY(upload_async(d, a, n * 32, cx().stream));
  HIP_TRY(upload_async(d + 8ull * n, b, n * 32, cx().stream));
  uint32_t gx = cdiv(n, 256);
  const uint32_t cap = (uint32_t)cx().num_cus * 16;
  if (gx > cap) gx = cap;
  if (curve == BPGPU_PALLAS)
    k_hadamard<1><<<gx, 256, 0, cx().stream>>>(d, d + 8ull * n, d, (uint32_t)n);
  else
    k_hadamard<0><<<gx, 256, 0, cx().stream>>>(d, d + 8ull * n, d, (uint32_t)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d, n * 32, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_hadamard_points(int curve, const uint64_t* bases_xyz, const uint64_t* scalars, size_t n, uint64_t* out_xyz) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n > 0x3fffffffu) return fail(BPGPU_E_ARG, "n too large");
  if (n == 0) return 0;
  if (!bases_xyz || !scalars || !out_xyz) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_pts_in.ensure(n * 96));
  TRY(cx().st_pts.ensure(n * 64));
  TRY(cx().st_sc.ensure(n * 32));
  HIP_TRY(upload_async(cx().st_sc.p, scalars, n * 32, cx().stream));
  uint32_t* d = cx().st_pts.as<uint32_t>();
  TRY(upload_points(curve, bases_xyz, n, 0, d, cx().stream));  // chunked, Z = 1 rows compacted
  {
    OpSpan span(cx());
    if (curve == BPGPU_PALLAS)
      k_hadamard_points<0><<<cdiv(n, 256), 256, 0, cx().stream>>>(d, cx().st_sc.as<uint32_t>(), d, (uint32_t)n);
    else
      k_hadamard_points<1><<<cdiv(n, 256), 256, 0, cx().stream>>>(d, cx().st_sc.as<uint32_t>(), d, (uint32_t)n);
  }
  HIP_TRY(hipGetLastError());
  TRY(points_out(curve, d, cx().st_pts_in.as<uint32_t>(), n));
  HIP_TRY(hipMemcpyAsync(out_xyz, cx().st_pts_in.p, n * 96, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_powers(int curve, const uint64_t f[4], uint64_t start, size_t n, uint64_t* out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (n > 0x3fffffffu) return fail(BPGPU_E_ARG, "n too large");
  if (n == 0) return 0;
  if (!f || !out) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_sc.ensure(n * 32));
  PowersArgs args;
  memcpy(args.f, f, 32);
  args.start = start;
  const uint32_t lanes = cdiv(n, 8);
  if (curve == BPGPU_PALLAS)
    k_powers<1><<<cdiv(lanes, 256), 256, 0, cx().stream>>>(args, cx().st_sc.as<uint32_t>(), (uint32_t)n);
  else
    k_powers<0><<<cdiv(lanes, 256), 256, 0, cx().stream>>>(args, cx().st_sc.as<uint32_t>(), (uint32_t)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, cx().st_sc.p, n * 32, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

// ---- sparse matrix - vector products (Com-R1CS compiler) ----------------------------
// ptr/idx: compressed rows of the product to run (already validated, 32-bit); perm: value index per entry or null
static int spmv_run(int curve, const uint32_t* ptr, const uint32_t* idx, const uint32_t* perm, size_t nnz,
                    const uint64_t* vals, size_t n_out, const uint64_t* vec, size_t vec_len, uint64_t* out) {
  DevBuf* B = cx().aux;
  TRY(B[0].ensure((n_out + 1) * 4));
  TRY(B[1].ensure(nnz * 4 + 4));
  if (perm) TRY(B[2].ensure(nnz * 4 + 4));
  TRY(B[3].ensure(nnz * 32 + 32));
  TRY(B[4].ensure(vec_len * 32 + 32));
  TRY(B[5].ensure(n_out * 32));
  hipStream_t st = cx().stream;
  HIP_TRY(upload_async(B[0].p, ptr, (n_out + 1) * 4, st));
  if (nnz) {
    HIP_TRY(upload_async(B[1].p, idx, nnz * 4, st));
    if (perm) HIP_TRY(upload_async(B[2].p, perm, nnz * 4, st));
    HIP_TRY(upload_async(B[3].p, vals, nnz * 32, st));
  }
  if (vec_len) HIP_TRY(upload_async(B[4].p, vec, vec_len * 32, st));
  // lanes per row: two when the rows are short throughout, eight otherwise
  uint32_t max_len = 0;
  for (size_t r = 0; r < n_out; r++) max_len = ptr[r + 1] - ptr[r] > max_len ? ptr[r + 1] - ptr[r] : max_len;
  const bool narrow = max_len <= 64 && nnz <= 6 * n_out;
  const uint32_t group = narrow ? 2u : 8u;
  const uint32_t grid = cdiv((uint64_t)n_out * group, 256);
  const uint32_t* d_perm = perm ? B[2].as<uint32_t>() : nullptr;
  {
    OpSpan span(cx());
#define BP_SPMV(SF, G)                                                                                          \
  k_spmv<SF, G><<<grid, 256, 0, st>>>(B[0].as<uint32_t>(), B[1].as<uint32_t>(), d_perm, B[3].as<uint32_t>(), \
                                      B[4].as<uint32_t>(), B[5].as<uint32_t>(), (uint32_t)n_out)
    if (curve == BPGPU_PALLAS) {
      if (narrow) BP_SPMV(1, 2u); else BP_SPMV(1, 8u);
    } else {
      if (narrow) BP_SPMV(0, 2u); else BP_SPMV(0, 8u);
    }
#undef BP_SPMV
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, B[5].p, n_out * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

// the checks both products share: monotone row pointers starting at 0, sizes inside the kernels' 32-bit indices
static int spmv_check(const uint64_t* row_ptr, const uint64_t* col_idx, const uint64_t* vals, size_t n_rows, size_t* nnz) {
  if (n_rows > 0x0fffffffu) return fail(BPGPU_E_ARG, "too many rows");
  *nnz = 0;
  if (n_rows == 0) return 0;
  if (!row_ptr) return fail(BPGPU_E_ARG, "null pointer");
  if (row_ptr[0] != 0) return fail(BPGPU_E_ARG, "row_ptr[0] must be 0");
  for (size_t r = 0; r < n_rows; r++)
    if (row_ptr[r + 1] < row_ptr[r]) return fail(BPGPU_E_ARG, "row_ptr decreases at row %zu", r);
  *nnz = row_ptr[n_rows];
  if (*nnz > 0x3fffffffu) return fail(BPGPU_E_ARG, "too many entries");
  if (*nnz && (!col_idx || !vals)) return fail(BPGPU_E_ARG, "null pointer");
  return 0;
}

int bpgpu_mat_vec_mult(int curve, const uint64_t* row_ptr, const uint64_t* col_idx, const uint64_t* vals, size_t n_rows,
                       const uint64_t* vec, size_t vec_len, uint64_t* out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  size_t nnz = 0;
  TRY(spmv_check(row_ptr, col_idx, vals, n_rows, &nnz));
  if (n_rows == 0) return 0;
  if (!out || (vec_len && !vec)) return fail(BPGPU_E_ARG, "null pointer");
  if (vec_len > 0x3fffffffu) return fail(BPGPU_E_ARG, "vector too long");
  std::vector<uint32_t> ptr(n_rows + 1), idx(nnz);
  for (size_t r = 0; r <= n_rows; r++) ptr[r] = (uint32_t)row_ptr[r];
  for (size_t e = 0; e < nnz; e++) {
    if (col_idx[e] >= vec_len)  // assert!(c < &vec.len()), com_r1cs.rs:51
      return fail(BPGPU_E_ARG, "column index %llu outside the vector (length %zu)", (unsigned long long)col_idx[e], vec_len);
    idx[e] = (uint32_t)col_idx[e];
  }
  return spmv_run(curve, ptr.data(), idx.data(), nullptr, nnz, vals, n_rows, vec, vec_len, out);
}

int bpgpu_vec_mat_mult(int curve, const uint64_t* vec, size_t vec_len, const uint64_t* row_ptr, const uint64_t* col_idx,
                       const uint64_t* vals, size_t n_rows, size_t output_size, uint64_t* out) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  size_t nnz = 0;
  TRY(spmv_check(row_ptr, col_idx, vals, n_rows, &nnz));
  if (vec_len != n_rows)  // assert_eq!(vec.len(), mat.len()), com_r1cs.rs:74
    return fail(BPGPU_E_ARG, "vector length %zu != matrix rows %zu", vec_len, n_rows);
  if (output_size > 0x0fffffffu) return fail(BPGPU_E_ARG, "output too long");
  for (size_t e = 0; e < nnz; e++)
    if (col_idx[e] >= output_size)  // assert!(mat_cols(mat) <= output_size), com_r1cs.rs:72
      return fail(BPGPU_E_ARG, "matrix has a column %llu >= output size %zu", (unsigned long long)col_idx[e], output_size);
  if (output_size == 0) return 0;
  if (!out || (n_rows && !vec)) return fail(BPGPU_E_ARG, "null pointer");
  // column-compressed form by a counting sort: entry e of row r becomes (row r, value e) in column col_idx[e]
  std::vector<uint32_t> ptr(output_size + 1, 0), rows(nnz), perm(nnz);
  for (size_t e = 0; e < nnz; e++) ptr[col_idx[e] + 1]++;
  for (size_t c = 0; c < output_size; c++) ptr[c + 1] += ptr[c];
  {
    std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
    for (size_t r = 0; r < n_rows; r++)
      for (uint64_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
        const uint32_t at = cur[col_idx[e]]++;
        rows[at] = (uint32_t)r;
        perm[at] = (uint32_t)e;
      }
  }
  return spmv_run(curve, ptr.data(), rows.data(), perm.data(), nnz, vals, output_size, vec, vec_len, out);
}

// ---- seed-faithful point sampling ----------------------------------------------------
extern "C++" {
template <int BF>
static void lift_root(LiftArgs* a) {
  using HF = host::HF<BF>;
  using C = FieldConsts<BF>;
  auto pow = [](host::H4 base, const uint64_t e[4]) {
    host::H4 acc = HF::one();
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        acc = HF::sqr(acc);
        if ((e[i] >> b) & 1) acc = HF::mul(acc, base);
      }
    return acc;
  };
  uint64_t half[4], t[4];  // (p - 1) / 2 and (p - 1) / 2^32
  for (int i = 0; i < 4; i++) half[i] = (C::P64[i] >> 1) | (i < 3 ? C::P64[i + 1] << 63 : 0);
  for (int i = 0; i < 4; i++) t[i] = (C::P64[i] >> 32) | (i < 3 ? C::P64[i + 1] << 32 : 0);
  for (uint64_t g = 2;; g++) {  // the first non-residue: g^((p-1)/2) = -1
    host::H4 gm = HF::from_u64(g);
    if (!HF::eq(pow(gm, half), HF::one())) {
      host::H4 z = pow(gm, t);
      memcpy(a->root, z.v, 32);
      return;
    }
  }
}
}  // extern "C++"

int bpgpu_curve_lift_x(int curve, const uint64_t* xs, const uint8_t* greatest, size_t m, uint64_t* ys, uint8_t* ok) {
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (m > 0x3fffffffu) return fail(BPGPU_E_ARG, "m too large");
  if (m == 0) return 0;
  if (!xs || !greatest || !ys || !ok) return fail(BPGPU_E_ARG, "null pointer");
  static LiftArgs args[2];
  static bool have[2] = {false, false};
  if (!have[curve]) {
    if (curve == BPGPU_PALLAS) lift_root<0>(&args[0]); else lift_root<1>(&args[1]);
    have[curve] = true;
  }
  DevBuf* B = cx().aux;
  TRY(B[0].ensure(m * 32));
  TRY(B[1].ensure(m));
  TRY(B[2].ensure(m * 32));
  TRY(B[3].ensure(m));
  hipStream_t st = cx().stream;
  HIP_TRY(upload_async(B[0].p, xs, m * 32, st));
  HIP_TRY(upload_async(B[1].p, greatest, m, st));
  if (curve == BPGPU_PALLAS)
    k_lift_x<0><<<cdiv(m, 256), 256, 0, st>>>(args[0], B[0].as<uint32_t>(), B[1].as<uint8_t>(), B[2].as<uint32_t>(),
                                              B[3].as<uint8_t>(), (uint32_t)m);
  else
    k_lift_x<1><<<cdiv(m, 256), 256, 0, st>>>(args[1], B[0].as<uint32_t>(), B[1].as<uint8_t>(), B[2].as<uint32_t>(),
                                              B[3].as<uint8_t>(), (uint32_t)m);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ys, B[2].p, m * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ok, B[3].p, m, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

// ---- IPA prover session -----------------------------------------------------
// Shared by the two begin calls: generators come from host memory (ark Jacobian, converted on the
// device) or from device memory (dev affine, as bpgpu_points_to_device leaves them); the witness
// always comes from the host.
static int ipa_begin_common(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz, const uint64_t* ip_gen_xyz,
                            const void* d_a_gens, const void* d_b_gens, const void* d_ip_gen, const uint64_t* a,
                            const uint64_t* b, size_t n, bpgpu_ipa** out) {
  const bool from_device = d_ip_gen != nullptr;
  // zero_pad_to_two_power (util/mod.rs:137-144): next power of two, at least 1
  size_t np = pow2_at_least(n < 1 ? 1 : n);
  bpgpu_ipa* s = new bpgpu_ipa();
  s->ctx = &cx();
  s->failed = false;
  s->curve = curve;
  s->np = np;
  s->len = np;
  s->lr_done = false;
  s->lr_cached = false;
  s->d_pts = nullptr;
  s->d_sc = nullptr;
  const size_t freeze_len = g_freeze_len.load();
  s->freeze_thresh = np < freeze_len ? np : freeze_len;
  s->lazy = false;
  s->lazy_off = !g_pair_fold.load() || np < 4;
  s->freeze_cap = s->lazy_off ? s->freeze_thresh : np;  // pair folding freezes at any length
  s->frozen = false;
  s->d_table = nullptr;
  s->table_c = 0;
  s->freeze_disabled = freeze_len == 0;
  s->n0 = 0;
  s->sc_scaled = 2 * np + 2 + 6 * s->freeze_cap;
  ipa_mu_reset(s);
  const size_t sc_words = (s->sc_scaled + 2 * np) * 8;
  s->pooled = false;
  auto cleanup = [&]() {
    if (s->pooled) {
      cx().ipa_pool_busy = false;
    } else {
      if (s->d_pts) (void)hipFree(s->d_pts);
      if (s->d_sc) (void)hipFree(s->d_sc);
    }
    delete s;
  };
  if (!cx().ipa_pool_busy) {
    int prc = cx().ipa_pts.ensure((2 * np + 1) * 64);
    if (prc == 0) prc = cx().ipa_sc.ensure(sc_words * 4);
    if (prc != 0) {
      delete s;
      return prc;
    }
    s->d_pts = cx().ipa_pts.as<uint32_t>();
    s->d_sc = cx().ipa_sc.as<uint32_t>();
    s->pooled = true;
    cx().ipa_pool_busy = true;
  } else {  // a second session open at the same time owns its buffers
    hipError_t e = hipMalloc((void**)&s->d_pts, (2 * np + 1) * 64);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_sc, sc_words * 4);
    if (e != hipSuccess) {
      cleanup();
      return fail(BPGPU_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
  }
  hipStream_t st = cx().stream;
  int rc = 0;
  do {
    // identity points / zero scalars are all-zero bytes in both device formats
    if (hipMemsetAsync(s->d_pts, 0, (2 * np + 1) * 64, st) != hipSuccess ||
        hipMemsetAsync(s->d_sc, 0, sc_words * 4, st) != hipSuccess) {
      rc = fail(BPGPU_E_HIP, "hipMemsetAsync failed");
      break;
    }
    if (n) {
      if (upload_async(s->d_sc, a, n * 32, st) != hipSuccess ||
          upload_async(s->d_sc + 8 * np, b, n * 32, st) != hipSuccess) {
        rc = fail(BPGPU_E_HIP, "upload failed");
        break;
      }
    }
    if (from_device) {
      if ((n && (hipMemcpyAsync(s->d_pts, d_a_gens, n * 64, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                 hipMemcpyAsync(s->d_pts + 16 * np, d_b_gens, n * 64, hipMemcpyDeviceToDevice, st) != hipSuccess)) ||
          hipMemcpyAsync(s->d_pts + 32 * np, d_ip_gen, 64, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        rc = fail(BPGPU_E_HIP, "device copy failed");
        break;
      }
    } else if (n * 96 >= Uploader::MIN_BYTES && np >= 4 && np / 2 > s->freeze_thresh) {
      // Large vectors from the host: the generators go up in two phases -- what L of the first round reads
      // (A[np/2..], B[..np/2], Q: bp_iter.rs:68-74), then what R reads (:75-81) -- and the first round's two MSMs run
      // here, one after the other, each as soon as its half has arrived: L's sort overlaps the first half's DMA, L's
      // accumulation the second half's.  round_lr then hands the cached pair out.  (n > np / 2 by the choice of np.)
      const size_t h = np / 2;
      uint32_t* dA = s->d_pts;
      uint32_t* dB = s->d_pts + 16 * np;
      PtSeg segs[5] = {{a_gens_xyz + 12 * h, n - h, 0, dA + 16 * h, 0},
                       {b_gens_xyz, h, 0, dB, 0},
                       {ip_gen_xyz, 1, 0, s->d_pts + 32 * np, 0},
                       {a_gens_xyz, h, 0, dA, 1},
                       {b_gens_xyz + 12 * h, n - h, 0, dB + 16 * h, 1}};
      // (phase 1 crosses PCIe while L's k_accum holds the chip: no kernels of its own, or they and the DMAs queued behind
      // them in the shared hardware queues wait for that kernel's end -- pt_upload_begin)
      if ((rc = pt_upload_begin(curve, segs, 5, st, /* defer_mask = */ 2)) != 0) break;
      const uint32_t hn = (uint32_t)h, npu = (uint32_t)np;
      IpPair pl{0u, npu + hn}, pr{hn, npu};
      rc = ip_launch(curve, s->d_sc, s->d_sc, pl, pr, 2, hn, s->d_sc + 8ull * (2 * np), s->d_sc + 8ull * (2 * np + 1));
      for (int side = 0; side < 2 && rc == 0; side++) {
        MsmPlan plan;
        memset(&plan, 0, sizeof plan);
        plan.M = 1;
        plan.n = 2 * hn + 1;
        if (side == 0) {
          plan.seg[0][0] = MsmSeg{hn, 0u, hn};
          plan.seg[0][1] = MsmSeg{npu, npu + hn, hn};
          plan.seg[0][2] = MsmSeg{2 * npu, 2 * npu, 1};
        } else {
          plan.seg[0][0] = MsmSeg{0u, hn, hn};
          plan.seg[0][1] = MsmSeg{npu + hn, npu, hn};
          plan.seg[0][2] = MsmSeg{2 * npu, 2 * npu + 1, 1};
        }
        const std::function<int()> ready = [&]() -> int { return pt_upload_wait(side, st); };
        rc = msm_dispatch(curve, s->d_pts, s->d_sc, plan, (uint64_t(*)[9])(side == 0 ? s->l_cache : s->r_cache), &ready);
      }
      const int rc_up = pt_upload_end();  // the caller's arrays are read until here
      if (rc == 0) rc = rc_up;
      if (rc != 0) break;
      s->lr_cached = true;
    } else {
      if ((rc = upload_points(curve, a_gens_xyz, n, 0, s->d_pts, st)) != 0) break;
      if ((rc = upload_points(curve, b_gens_xyz, n, 0, s->d_pts + 16 * np, st)) != 0) break;
      if ((rc = upload_points(curve, ip_gen_xyz, 1, 0, s->d_pts + 32 * np, st)) != 0) break;
    }
    if (np <= s->freeze_thresh && !s->freeze_disabled && np > 1 && ((rc = ipa_freeze(s)) != 0 || (rc = ipa_build_table(s)) != 0)) break;
    if (hipStreamSynchronize(st) != hipSuccess) rc = fail(BPGPU_E_HIP, "sync failed");
  } while (0);
  if (rc) {
    cleanup();
    return rc;
  }
  cx().open_sessions++;
  *out = s;
  return 0;
}

int bpgpu_ipa_begin(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz, const uint64_t* ip_gen_xyz,
                    const uint64_t* a, const uint64_t* b, size_t n, bpgpu_ipa** out) {
  if (n > BPGPU_IPA_MAX_N) return fail(BPGPU_E_ARG, "n = %zu: an IPA session holds at most 2^26 elements per vector", n);
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!out || !ip_gen_xyz) return fail(BPGPU_E_ARG, "null pointer");
  if (n && (!a_gens_xyz || !b_gens_xyz || !a || !b)) return fail(BPGPU_E_ARG, "null pointer");
  return ipa_begin_common(curve, a_gens_xyz, b_gens_xyz, ip_gen_xyz, nullptr, nullptr, nullptr, a, b, n, out);
}

int bpgpu_ipa_begin_device(int curve, const void* d_a_gens, const void* d_b_gens, const void* d_ip_gen, const uint64_t* a,
                           const uint64_t* b, size_t n, bpgpu_ipa** out) {
  if (n > BPGPU_IPA_MAX_N) return fail(BPGPU_E_ARG, "n = %zu: an IPA session holds at most 2^26 elements per vector", n);
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!out || !d_ip_gen) return fail(BPGPU_E_ARG, "null pointer");
  if (n && (!d_a_gens || !d_b_gens || !a || !b)) return fail(BPGPU_E_ARG, "null pointer");
  return ipa_begin_common(curve, nullptr, nullptr, nullptr, d_a_gens, d_b_gens, d_ip_gen, a, b, n, out);
}

size_t bpgpu_ipa_len(const bpgpu_ipa* s) {
  if (!s) return 0;
  std::lock_guard<std::mutex> lk(s->ctx->mu);
  return s->len;
}

// A failure inside a round leaves a, b, the generators and the weights partly updated: the session is
// marked failed and refuses every later call (a retry would fold the scalars twice).
#define IPA_TRY(expr)       \
  do {                      \
    int r_ = (expr);        \
    if (r_ != 0) {          \
      s->failed = true;     \
      return r_;            \
    }                       \
  } while (0)
#define IPA_HIP_TRY(expr)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      s->failed = true;                                                                            \
      return fail(BPGPU_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                              \
  } while (0)
static int ipa_usable(const bpgpu_ipa* s) {
  if (s->failed) return fail(BPGPU_E_STATE, "the session failed in an earlier call; free it");
  return 0;
}

int bpgpu_ipa_round_lr(bpgpu_ipa* s, uint64_t l_xy_inf[9], uint64_t r_xy_inf[9]) {
  if (!s || !l_xy_inf || !r_xy_inf) return fail(BPGPU_E_ARG, "null pointer");
  CtxLock lk(s->ctx);
  TRY(check_ready());
  TRY(ipa_usable(s));
  if (s->len <= 1) return fail(BPGPU_E_STATE, "the vectors already have length 1");
  if (s->lr_done) return fail(BPGPU_E_STATE, "round_fold must follow round_lr");
  OpMute mute(cx(), s->frozen && !s->lazy);
  const uint32_t n = (uint32_t)(s->len / 2), np = (uint32_t)s->np;
  if (s->lr_cached) {  // the first round of a large session: computed by bpgpu_ipa_begin
    memcpy(l_xy_inf, s->l_cache, 72);
    memcpy(r_xy_inf, s->r_cache, 72);
    s->lr_cached = false;
    s->lr_done = true;
    return 0;
  }
  // c_L = <a[..n], b[n..]>, c_R = <a[n..], b[..n]>   (bp_iter.rs:73,80)
  IpPair pl{0u, np + n}, pr{n, np};
  TRY(ip_launch(s->curve, s->d_sc, s->d_sc, pl, pr, 2, n, s->d_sc + 8ull * (2 * np), s->d_sc + 8ull * (2 * np + 1)));
  MsmPlan plan;
  memset(&plan, 0, sizeof plan);
  plan.M = 2;
  uint64_t out[2][9];
  if (s->frozen) {
    const uint32_t n0 = (uint32_t)s->n0, fs = 2 * np + 2;
    if (s->curve == BPGPU_PALLAS)
      k_frozen_scalars<1><<<cdiv(n0, 256), 256, 0, cx().stream>>>(s->d_sc, s->d_sc + 8ull * np, s->wA(), s->wB(), s->fsc(),
                                                                n0, n);
    else
      k_frozen_scalars<0><<<cdiv(n0, 256), 256, 0, cx().stream>>>(s->d_sc, s->d_sc + 8ull * np, s->wA(), s->wB(), s->fsc(),
                                                                n0, n);
    HIP_TRY(hipGetLastError());
    plan.n = 2 * n0 + 1;
    plan.n_eff = n0 + 1;  // half of the weighted scalars are zero by construction (k_frozen_scalars)
    const bool tab = s->d_table != nullptr && !s->lazy;
    // point indices: the session's A | B | Q array, or the table's compact A0 | B0 | Q rows
    const uint32_t pB = tab ? n0 : np, pQ = tab ? 2 * n0 : 2 * np;
    if (tab) {
      plan.table_stride = 2 * n0 + 1;
      plan.c = s->table_c;
    }
    plan.seg[0][0] = MsmSeg{0u, fs, n0};
    plan.seg[0][1] = MsmSeg{pB, fs + n0, n0};
    plan.seg[0][2] = MsmSeg{pQ, 2 * np, 1};
    plan.seg[1][0] = MsmSeg{0u, fs + 2 * n0, n0};
    plan.seg[1][1] = MsmSeg{pB, fs + 3 * n0, n0};
    plan.seg[1][2] = MsmSeg{pQ, 2 * np + 1, 1};
    TRY(msm_dispatch(s->curve, tab ? s->d_table : s->d_pts, s->d_sc, plan, out));
    memcpy(l_xy_inf, out[0], 72);
    memcpy(r_xy_inf, out[1], 72);
    s->lr_done = true;
    return 0;
  }
  plan.n = 2 * n + 1;
  // the stored generators carry common factors: their MSM scalars are mu_A a and mu_B b
  uint32_t sa = 0u, sb = np;  // scalar indices of a and b
  if (!s->mu_one) {
    sa = (uint32_t)s->sc_scaled;
    sb = sa + np;
    FoldScalars m = make_fold_scalars(s->muA, s->muB);
    const uint32_t blocks = cdiv(2 * n, 256) < 2048 ? cdiv(2 * n, 256) : 2048;
    if (s->curve == BPGPU_PALLAS)
      k_scale_scalars<1><<<dim3(blocks, 2), 256, 0, cx().stream>>>(s->d_sc, s->d_sc + 8ull * sa, s->d_sc + 8ull * np,
                                                                   s->d_sc + 8ull * sb, m, 2 * n);
    else
      k_scale_scalars<0><<<dim3(blocks, 2), 256, 0, cx().stream>>>(s->d_sc, s->d_sc + 8ull * sa, s->d_sc + 8ull * np,
                                                                   s->d_sc + 8ull * sb, m, 2 * n);
    HIP_TRY(hipGetLastError());
  }
  // L = msm(A[n..] | B[..n] | Q,  a[..n] | b[n..] | c_L)   (bp_iter.rs:68-74)
  plan.seg[0][0] = MsmSeg{n, sa, n};
  plan.seg[0][1] = MsmSeg{np, sb + n, n};
  plan.seg[0][2] = MsmSeg{2 * np, 2 * np, 1};
  // R = msm(A[..n] | B[n..] | Q,  a[n..] | b[..n] | c_R)   (bp_iter.rs:75-81)
  plan.seg[1][0] = MsmSeg{0u, sa + n, n};
  plan.seg[1][1] = MsmSeg{np + n, sb, n};
  plan.seg[1][2] = MsmSeg{2 * np, 2 * np + 1, 1};
  TRY(msm_dispatch(s->curve, s->d_pts, s->d_sc, plan, out));
  memcpy(l_xy_inf, out[0], 72);
  memcpy(r_xy_inf, out[1], 72);
  s->lr_done = true;
  return 0;
}

int bpgpu_ipa_round_fold(bpgpu_ipa* s, const uint64_t x[4], const uint64_t x_inv[4]) {
  if (!s || !x || !x_inv) return fail(BPGPU_E_ARG, "null pointer");
  CtxLock lk(s->ctx);
  TRY(check_ready());
  TRY(ipa_usable(s));
  if (!s->lr_done) return fail(BPGPU_E_STATE, "round_lr must precede round_fold");
  OpMute mute(cx(), s->frozen && !s->lazy);
  const uint32_t n = (uint32_t)(s->len / 2), np = (uint32_t)s->np;
  // a' = x a_lo + x^-1 a_hi;  b' = x^-1 b_lo + x b_hi          (bp_iter.rs:88-95)
  FoldScalars fa = make_fold_scalars(x, x_inv), fb = make_fold_scalars(x_inv, x);
  uint32_t* da = s->d_sc;
  uint32_t* db = s->d_sc + 8ull * np;
  IPA_TRY(fold_scalars_launch(s->curve, da, da, fa, db, db, fb, 2, n));
  // A' = x^-1 A_lo + x A_hi;  B' = x B_lo + x^-1 B_hi          (bp_iter.rs:97-104)
  // pair folding: a long round whose successor is long too leaves its generators alone; the next
  // round's L, R run over them with weighted scalars and its fold applies both challenges at once
  if (!s->frozen && !s->lazy_off && s->len >= 4 && s->len / 4 >= (s->freeze_thresh > 1 ? s->freeze_thresh : 1) &&
      (s->curve == BPGPU_PALLAS ? are_inverses<1>(x, x_inv) : are_inverses<0>(x, x_inv))) {
    IPA_TRY(ipa_freeze(s));
    s->lazy = true;
  }
  if (s->frozen) {
    // deferred: the fold only updates the weights of the frozen generators
    const uint32_t n0 = (uint32_t)s->n0;
    if (s->curve == BPGPU_PALLAS)
      k_frozen_update<1><<<cdiv(n0, 256), 256, 0, cx().stream>>>(s->wA(), s->wB(), n0, n, fa);
    else
      k_frozen_update<0><<<cdiv(n0, 256), 256, 0, cx().stream>>>(s->wA(), s->wB(), n0, n, fa);
    IPA_HIP_TRY(hipGetLastError());
    std::array<uint64_t, 8> ch;
    memcpy(ch.data(), x, 32);
    memcpy(ch.data() + 4, x_inv, 32);
    s->pending.push_back(ch);
    if (s->lazy && s->pending.size() == 2) IPA_TRY(ipa_pair_fold(s));
  } else {
    IPA_TRY(ipa_fold_generators(s, x, x_inv, n));
  }
  s->len = n;
  s->lr_done = false;
  if (!s->frozen && !s->freeze_disabled && n > 1 && n <= s->freeze_thresh) {
    IPA_TRY(ipa_freeze(s));
    IPA_TRY(ipa_build_table(s));
  }
  // With frozen generators the round's device work is two small kernels (and once the table build): the call returns
  // without waiting for them -- the next round_lr / finish / export waits on the same stream and reports their errors
  // (a wait here costs a short round 20 us of wake-up latency, and the caller's Fiat-Shamir hop overlaps the table build)
  if (!s->frozen) IPA_HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_ipa_finish(bpgpu_ipa* s, uint64_t final_a[4], uint64_t final_b[4]) {
  if (!s || !final_a || !final_b) return fail(BPGPU_E_ARG, "null pointer");
  CtxLock lk(s->ctx);
  TRY(check_ready());
  TRY(ipa_usable(s));
  if (s->len != 1 || s->lr_done) return fail(BPGPU_E_STATE, "the loop has not finished");
  HIP_TRY(hipMemcpyAsync(final_a, s->d_sc, 32, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipMemcpyAsync(final_b, s->d_sc + 8ull * s->np, 32, hipMemcpyDeviceToHost, cx().stream));
  HIP_TRY(hipStreamSynchronize(cx().stream));
  return 0;
}

int bpgpu_ipa_export(bpgpu_ipa* s, uint64_t* a_gens_xyz, uint64_t* b_gens_xyz, uint64_t* a, uint64_t* b) {
  if (!s) return fail(BPGPU_E_ARG, "null session");
  CtxLock lk(s->ctx);
  TRY(check_ready());
  TRY(ipa_usable(s));
  if (s->lr_done) return fail(BPGPU_E_STATE, "export between round_lr and round_fold");
  if (a_gens_xyz || b_gens_xyz) {
    TRY(ipa_materialise(s));
    TRY(ipa_normalise(s));
  }
  const size_t n = s->len, np = s->np;
  hipStream_t st = cx().stream;
  TRY(cx().st_out.ensure(n * 96));
  if (a_gens_xyz) {
    TRY(points_out(s->curve, s->d_pts, cx().st_out.as<uint32_t>(), n));
    HIP_TRY(hipMemcpyAsync(a_gens_xyz, cx().st_out.p, n * 96, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  if (b_gens_xyz) {
    TRY(points_out(s->curve, s->d_pts + 16ull * np, cx().st_out.as<uint32_t>(), n));
    HIP_TRY(hipMemcpyAsync(b_gens_xyz, cx().st_out.p, n * 96, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  if (a) HIP_TRY(hipMemcpyAsync(a, s->d_sc, n * 32, hipMemcpyDeviceToHost, st));
  if (b) HIP_TRY(hipMemcpyAsync(b, s->d_sc + 8ull * np, n * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

void bpgpu_ipa_free(bpgpu_ipa* s) {
  if (!s) return;
  CtxLock lk(s->ctx);
  if (cx().ready) {
    (void)hipSetDevice(cx().device);
    (void)hipStreamSynchronize(cx().stream);
  }
  cx().open_sessions--;
  if (s->pooled) {
    cx().ipa_pool_busy = false;
  } else {
    if (s->d_pts) (void)hipFree(s->d_pts);
    if (s->d_sc) (void)hipFree(s->d_sc);
  }
  delete s;
}

// ---- allocation-free inner product ---------------------------------------------------------
// bpgpu_ip_device for callers that cannot allocate: the block partial sums go to this device's copy of the
// file-scope `scratch` array (address taken once per context in ctx_init) instead of a grow-only hipMalloc'd
// buffer, the result goes to caller-owned device memory, and nothing here (or below it) calls hipMalloc,
// hipFree or any allocating routine.  Calls on one GPU share that GPU's array: they are serialised per context
// by the context mutex, but two contexts on one GPU (or another process's copy of the library) must not overlap
// such calls on the same device -- documented in bpgpu.h.  Every partial the second kernel reads is written by
// the first in the same call, so nothing needs clearing between calls.
int bpgpu_ip_device_noalloc(int curve, const void* d_a, const void* d_b, size_t n, void* d_out) {
  if (n > BPGPU_IP_NOALLOC_MAX_N) return fail(BPGPU_E_ARG, "n = %zu: bpgpu_ip_device_noalloc takes at most 2^32 - 1 elements", n);
  CtxLock lk;
  TRY(check_ready());
  TRY(check_curve(curve));
  if (!d_out) return fail(BPGPU_E_ARG, "null output");
  if (n && (!d_a || !d_b)) return fail(BPGPU_E_ARG, "null pointer");
  Ctx& C = cx();
  if (!C.noalloc_scratch) return fail(BPGPU_E_STATE, "scratch array address missing");
  static_assert(NOALLOC_IP_BLOCKS >= 1024, "ip_blocks() returns up to 1024 blocks");
  IpPair p0{0u, 0u};
  TRY(ip_launch_with(curve, (const uint32_t*)d_a, (const uint32_t*)d_b, p0, p0, 1, (uint32_t)n, C.noalloc_scratch,
                     (uint32_t*)d_out, (uint32_t*)d_out));
  HIP_TRY(hipStreamSynchronize(C.stream));
  return 0;
}

// ---- sharded (multi-GPU) MSM, one process ------------------------------------------------------
// pippenger_msm (src/util/msm.rs:88-92) is called from ONE process; to use the 8 GPUs of a node from that
// process the library keeps one context per GPU and cuts the pairs into contiguous ranges
// [g n / P, (g+1) n / P).  One persistent host thread per context runs the single-GPU pipeline on
// its range; every context ends with its own Horner tail on the host, so the per-GPU partial (72 bytes) is
// already in host memory when the leg ends and the combine is P - 1 host additions (bpgpu_points_sum's
// loop).  A device-side gather (RCCL) would upload those 72 bytes again to move them between GPUs and bring
// them back: the in-process path does not use it; the one-process-per-GPU path (bench.py, sharded.py) does,
// through torch.distributed, because there the partials live in different address spaces.
extern "C++" {
namespace {
struct ShardJob {
  int rc = 0;
  std::string err;
};
// runs f(g) with shard g's context current and locked on worker g; first failure wins
int run_on_shards(const std::function<int(int)>& f) {
  const int P = (int)g_shards.size();
  std::vector<ShardJob> jobs(P);
  g_shard_pool.run([&](int g) {
    CtxLock lk(g_shards[g].get());
    int rc = check_ready();
    if (rc == 0) rc = f(g);
    jobs[g].rc = rc;
    if (rc) jobs[g].err = g_err;
  });
  for (int g = 0; g < P; g++)
    if (jobs[g].rc) return fail(jobs[g].rc, "shard %d: %s", g, jobs[g].err.c_str());
  return 0;
}
int sum_partials(int curve, const std::vector<std::array<uint64_t, 9>>& parts, uint64_t out[9]) {
  return bpgpu_points_sum(curve, parts.empty() ? nullptr : parts[0].data(), parts.size(), out);
}
}  // namespace
}  // extern "C++"

struct bpgpu_sharded_gens {
  int curve;
  size_t n;
  std::vector<void*> d_pts;  // per shard: its range as dev affine points
  std::vector<size_t> lo, cnt;
};

int bpgpu_sharded_init(int n_ctx, const int* devices) {
  std::lock_guard<std::mutex> run(g_shard_run_mu);
  std::lock_guard<std::mutex> reg(g_reg_mu);
  if (n_ctx < 1 || n_ctx > 64) return fail(BPGPU_E_ARG, "n_ctx = %d outside [1, 64]", n_ctx);
  if (!g_shards.empty()) {
    bool same = (int)g_shards.size() == n_ctx;
    for (int g = 0; same && g < n_ctx; g++) same = g_shards[g]->device == (devices ? devices[g] : g);
    if (same) return 0;
    return fail(BPGPU_E_STATE, "the sharded contexts are already initialised with another device list");
  }
  std::vector<std::unique_ptr<Ctx>> made;
  g_ctx_count.store(n_ctx);  // sizes every context's helper threads (helper_threads())
  for (int g = 0; g < n_ctx; g++) {
    made.emplace_back(new Ctx());
    int rc = ctx_init(*made.back(), devices ? devices[g] : g);
    if (rc) {
      const std::string why = g_err;
      for (auto& c : made) ctx_release(*c);  // the one that failed has already dropped its handles
      g_ctx_count.store(1);
      g_err = why;
      return rc;
    }
  }
  g_shards = std::move(made);
  g_shard_pool.start(n_ctx);
  return 0;
}

int bpgpu_sharded_count(void) {
  std::lock_guard<std::mutex> reg(g_reg_mu);
  return (int)g_shards.size();
}

static int msm_sharded_host(int curve, const uint64_t* points, int fmt, const uint64_t* scalars, size_t n,
                            uint64_t out_xy_inf[9]) {
  std::lock_guard<std::mutex> run(g_shard_run_mu);
  if (g_shards.empty()) return fail(BPGPU_E_NOINIT, "bpgpu_sharded_init has not been called");
  TRY(check_curve(curve));
  if (!out_xy_inf) return fail(BPGPU_E_ARG, "null output");
  if (n && (!points || !scalars)) return fail(BPGPU_E_ARG, "null pointer");
  const size_t P = g_shards.size();
  if ((n + P - 1) / P > BPGPU_MSM_MAX_N) return fail(BPGPU_E_ARG, "n = %zu: more than 2^28 points per shard", n);
  std::vector<std::array<uint64_t, 9>> parts(P);
  const size_t pw = fmt == 0 ? 12 : 8;
  TRY(run_on_shards([&](int g) {
    const size_t lo = (size_t)g * n / P, hi = (size_t)(g + 1) * n / P, cnt = hi - lo;
    Ctx& C = cx();
    if (cnt) {
      TRY(C.st_pts.ensure(cnt * 64));
      TRY(C.st_sc.ensure(cnt * 32));
      HIP_TRY(upload_async(C.st_sc.p, scalars + lo * 4, cnt * 32, C.stream));
      TRY(upload_points(curve, points + lo * pw, cnt, fmt, C.st_pts.as<uint32_t>(), C.stream));
    }
    return msm_dispatch(curve, C.st_pts.as<uint32_t>(), C.st_sc.as<uint32_t>(), single_plan((uint32_t)cnt),
                        (uint64_t(*)[9])parts[g].data());
  }));
  return sum_partials(curve, parts, out_xy_inf);
}

int bpgpu_msm_sharded(int curve, const uint64_t* points_xyz, const uint64_t* scalars, size_t n, uint64_t out_xy_inf[9]) {
  return msm_sharded_host(curve, points_xyz, 0, scalars, n, out_xy_inf);
}
int bpgpu_msm_sharded_affine(int curve, const uint64_t* points_xy, const uint64_t* scalars, size_t n,
                             uint64_t out_xy_inf[9]) {
  return msm_sharded_host(curve, points_xy, 1, scalars, n, out_xy_inf);
}

int bpgpu_sharded_gens_load(int curve, const uint64_t* points, size_t n, int fmt, bpgpu_sharded_gens** out) {
  std::lock_guard<std::mutex> run(g_shard_run_mu);
  if (g_shards.empty()) return fail(BPGPU_E_NOINIT, "bpgpu_sharded_init has not been called");
  TRY(check_curve(curve));
  if (!out || (n && !points)) return fail(BPGPU_E_ARG, "null pointer");
  if (fmt != 0 && fmt != 1) return fail(BPGPU_E_ARG, "fmt must be 0 (jacobian) or 1 (affine)");
  const size_t P = g_shards.size();
  if ((n + P - 1) / P > BPGPU_MSM_MAX_N) return fail(BPGPU_E_ARG, "n = %zu: more than 2^28 points per shard", n);
  bpgpu_sharded_gens* h = new bpgpu_sharded_gens();
  h->curve = curve;
  h->n = n;
  h->d_pts.assign(P, nullptr);
  h->lo.resize(P);
  h->cnt.resize(P);
  const size_t pw = fmt == 0 ? 12 : 8;
  int rc = run_on_shards([&](int g) {
    const size_t lo = (size_t)g * n / P, hi = (size_t)(g + 1) * n / P, cnt = hi - lo;
    h->lo[g] = lo;
    h->cnt[g] = cnt;
    Ctx& C = cx();
    HIP_TRY(hipMalloc(&h->d_pts[g], (cnt ? cnt : 1) * 64));
    if (cnt) {
      TRY(upload_points(curve, points + lo * pw, cnt, fmt, (uint32_t*)h->d_pts[g], C.stream));
      HIP_TRY(hipStreamSynchronize(C.stream));
    }
    return 0;
  });
  if (rc) {
    (void)run_on_shards([&](int g) {
      if (h->d_pts[g]) (void)hipFree(h->d_pts[g]);
      return 0;
    });
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}

int bpgpu_sharded_gens_msm(bpgpu_sharded_gens* h, const uint64_t* scalars, uint64_t out_xy_inf[9]) {
  std::lock_guard<std::mutex> run(g_shard_run_mu);
  if (!h || !out_xy_inf || (h->n && !scalars)) return fail(BPGPU_E_ARG, "null pointer");
  if (g_shards.size() != h->d_pts.size()) return fail(BPGPU_E_STATE, "the sharded contexts changed since the generators were loaded");
  std::vector<std::array<uint64_t, 9>> parts(g_shards.size());
  TRY(run_on_shards([&](int g) {
    Ctx& C = cx();
    const size_t cnt = h->cnt[g];
    if (cnt) {
      TRY(C.st_sc.ensure(cnt * 32));
      HIP_TRY(upload_async(C.st_sc.p, scalars + h->lo[g] * 4, cnt * 32, C.stream));
    }
    return msm_dispatch(h->curve, (const uint32_t*)h->d_pts[g], C.st_sc.as<uint32_t>(), single_plan((uint32_t)cnt),
                        (uint64_t(*)[9])parts[g].data());
  }));
  return sum_partials(h->curve, parts, out_xy_inf);
}

void bpgpu_sharded_gens_free(bpgpu_sharded_gens* h) {
  if (!h) return;
  std::lock_guard<std::mutex> run(g_shard_run_mu);
  if (g_shards.size() == h->d_pts.size())
    (void)run_on_shards([&](int g) {
      (void)hipStreamSynchronize(cx().stream);
      if (h->d_pts[g]) (void)hipFree(h->d_pts[g]);
      return 0;
    });
  delete h;
}

// ---- sharded-MSM helper -----------------------------------------------------
int bpgpu_points_sum(int curve, const uint64_t* xy_inf, size_t count, uint64_t out_xy_inf[9]) {
  TRY(check_curve(curve));
  if (!out_xy_inf || (count && !xy_inf)) return fail(BPGPU_E_ARG, "null pointer");
  auto run = [&](auto tag) {
    constexpr int BF = decltype(tag)::value;
    using E = host::HEc<BF>;
    host::HPoint<BF> acc = E::identity();
    for (size_t i = 0; i < count; i++) {
      const uint64_t* p = xy_inf + 9 * i;
      if (p[8]) continue;
      host::H4 x, y;
      memcpy(x.v, p, 32);
      memcpy(y.v, p + 4, 32);
      acc = E::add(acc, E::from_affine(x, y));
    }
    write_xy_inf<BF>(acc, out_xy_inf);
  };
  if (curve == BPGPU_PALLAS)
    run(std::integral_constant<int, 0>{});
  else
    run(std::integral_constant<int, 1>{});
  return 0;
}

// ---- instrumentation ----------------------------------------------------------
int bpgpu_last_msm_timings(float out_ms[8]) {
  if (!out_ms) return fail(BPGPU_E_ARG, "null pointer");
  CtxLock lk;
  Ctx& C = cx();
  if (C.ready && C.msm_timed) {
    HIP_TRY(hipSetDevice(C.device));
    const int span[6][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 5}, {0, 5}};
    const int slot[6] = {0, 1, 2, 3, 5, 7};
    for (int i = 0; i < 6; i++) HIP_TRY(hipEventElapsedTime(&C.msm_ms[slot[i]], C.ev[span[i][0]], C.ev[span[i][1]]));
    C.msm_ms[4] = 0;
  }
  memcpy(out_ms, C.msm_ms, sizeof C.msm_ms);
  return 0;
}

int bpgpu_last_msm_entries(uint64_t* entries) {
  if (!entries) return fail(BPGPU_E_ARG, "null pointer");
  CtxLock lk;
  *entries = cx().msm_entries;
  return 0;
}

int bpgpu_last_fold_ops(uint64_t out[4]) {
  if (!out) return fail(BPGPU_E_ARG, "null pointer");
  CtxLock lk;
  memcpy(out, cx().fold_ops, sizeof cx().fold_ops);
  return 0;
}

int bpgpu_last_op_ms(float* ms) {
  CtxLock lk;
  TRY(check_ready());
  if (!ms) return fail(BPGPU_E_ARG, "null pointer");
  HIP_TRY(hipEventSynchronize(cx().op_ev[1]));
  HIP_TRY(hipEventElapsedTime(ms, cx().op_ev[0], cx().op_ev[1]));
  return 0;
}

int bpgpu_measure_mad_peak(double* mads_per_sec) {
  CtxLock lk;
  TRY(check_ready());
  if (!mads_per_sec) return fail(BPGPU_E_ARG, "null pointer");
  TRY(cx().st_out.ensure(4096));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, cx().device));
  const int iters = 8192, grid = prop.multiProcessorCount * 8;
  hipStream_t st = cx().stream;
  k_mad_peak<<<grid, 256, 0, st>>>(cx().st_out.as<uint32_t>(), 1u, iters);
  HIP_TRY(hipStreamSynchronize(st));
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    HIP_TRY(hipEventRecord(cx().ev[6], st));
    k_mad_peak<<<grid, 256, 0, st>>>(cx().st_out.as<uint32_t>(), (uint32_t)r + 2u, iters);
    HIP_TRY(hipEventRecord(cx().ev[7], st));
    HIP_TRY(hipEventSynchronize(cx().ev[7]));
    float ms;
    HIP_TRY(hipEventElapsedTime(&ms, cx().ev[6], cx().ev[7]));
    if (ms < best) best = ms;
  }
  *mads_per_sec = (double)grid * 256.0 * iters * 32.0 / (best * 1e-3);
  return 0;
}

}  // extern "C"
