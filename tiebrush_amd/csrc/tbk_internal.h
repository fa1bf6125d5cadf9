// tbk_internal.h — context, workspace arena, launch/profiling helpers shared by the
// HIP translation units behind include/tbk.h.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/tbk.h"

// error bits raised by kernels (device word tbk_ctx::d_err)
enum : uint32_t {
  TBK_DERR_UNSORTED = 1u << 0,
  TBK_DERR_COLLISION = 1u << 1,
  TBK_DERR_FATALOP = 1u << 2,
  TBK_DERR_SPAN = 1u << 3,      // read span >= 2^30
  TBK_DERR_NCIGAR = 1u << 4,    // tiecov: n_cigar >= 256 never terminates in the reference
  TBK_DERR_FRACTIONAL = 1u << 5, // non-integral YC met by an integer-only kernel
  TBK_DERR_OVERFLOW = 1u << 6,
  TBK_DERR_INTERNAL = 1u << 7,
  TBK_DERR_BIGBUCKET = 1u << 8, // not an error: the run sort met a (tid,start) bucket longer than its window -> radix fallback
  TBK_DERR_RAWORDER = 1u << 9   // not an error: the raw window path met an input it does not take (wgroup.hip) -> general path
};

struct KTime {
  const char* name;
  hipEvent_t a, b;
};

// A helper thread that lives as long as its context: side stages (deferred YD, the junction branch) are posted to it
// instead of paying a thread creation per call.
struct TbkWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool busy = false, quit = false, started = false;
  void post(std::function<void()> j) {
    {
      std::lock_guard<std::mutex> lk(m);
      if (!started) {
        started = true;
        th = std::thread([this]() {
          std::unique_lock<std::mutex> lk2(m);
          for (;;) {
            cv.wait(lk2, [&] { return quit || (bool)job; });
            if (quit) return;
            std::function<void()> run = std::move(job);
            job = nullptr;
            lk2.unlock();
            run();
            lk2.lock();
            busy = false;
            cv.notify_all();
          }
        });
      }
      job = std::move(j);
      busy = true;
    }
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return !busy; });
  }
  ~TbkWorker() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

// Test hooks and forced path choices.  Read ONCE, when a context is created, from TBK_DEBUG ("key=value,key=value"), and replaced as a
// whole by tbk_set_debug (include/tbk.h): no call path reads the environment.  Every field's default is the production behaviour.
struct TbkDebug {
  int path = 0;               // path=sort | window: the collapse's sort path / window path whatever the tile's size
  int raw = -1;               // raw=0: the window path with its key pass, effective-end scan and compaction as separate kernels
  int sort = 0;               // sort=radix | runs: one ordering path whatever the shape (1 / 2)
  int scan = 0;               // scan=lookback | 3pass: one form for every scan_op_run (1 / 2)
  uint32_t hash_mask = 0xFFFFFFFFu;  // hash_mask=0x..: bits of the strategy hash that survive (forces key collisions)
  uint64_t qhash_mask = ~0ull;       // qhash_mask=0x..: bits of the read-name hash (-A)
  uint32_t yd_wave_min = 0;   // yd_wave_min=N: chains of N items and more go to yd_wave_k (0: the default split)
  uint32_t yd_bgrid = 0;      // yd_bgrid=N: blocks of the chain bucketing (0: default)
  bool yd_radix = false;      // yd_radix=1: the YD items through the stable radix split for any tile
  bool yd_literal = false;    // yd_literal=1: every chain through yd_run_k, the literal list machine (normally the lists that outgrow the others)
  bool yd_own_arena = false;  // yd_own_arena=1: a deferred YD stage never borrows the main arena
  bool wg_dense_verify = false, wg_rank_merge = false;  // window path: per-record verification form; merge-sort ranking of a window's groups
  bool cov_legacy = false, cov_bundle_scan = false, cov_prep = false, junc_radix = false, no_junc_agg = false;
  uint64_t cov_tile_cap = 0;  // cov_tile_cap=N: capacity of the lean chain's tile tables (0: by the input)
  uint32_t jh_cap = 0;        // jh_cap=N: items a junction home may hold (0: JH_CAP)
  bool index_chain = false;   // index_chain=1: the record index by the per-file chain kernel whatever the files look like
  bool no_register = false;   // no_register=1: large host buffers are not page-locked for a call's copies
  bool no_bounce = false;     // no_bounce=1: results of moderate size are copied straight into the caller's (registered) arrays
  bool phases = false;        // phases=1: tbk_collapse_tile prints where its wall time went (copies in, grouping, YD stage, results) to stderr
};
void tbk_debug_parse(const char* spec, TbkDebug* out);

struct tbk_ctx {
  int device = 0;
  TbkDebug dbg;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t aux = nullptr;     // second stream for independent kernels inside one stage (created on first use)
  hipEvent_t aux_done = nullptr;
  // device workspace (bump allocator, reset at the start of each API call)
  char* ws = nullptr;
  size_t ws_cap = 0, ws_off = 0;
  // two-ended: while ws_top_mode is set, allocations come from the END of the arena, downwards (ws_top bytes in use there).  A stage
  // puts its temporaries there and what outlives it at the bottom; when its kernels are queued it sets ws_top back to 0 — the
  // temporaries are dead in stream order — and only the bottom has to stay for a deferred YD stage (ws_base_off)
  size_t ws_top = 0;
  bool ws_top_mode = false;
  bool ws_borrowed = false;      // `ws` is a range of another context's arena (deferred YD stage): never freed or regrown here
  std::vector<std::pair<char*, size_t>> ws_overflow;
  size_t ws_over_used = 0;
  // small persistent device words + pinned mirror
  uint32_t* d_err = nullptr;    // error bits: the low word of d_scalars[15] (cleared and read back with the counters)
  uint64_t* d_scalars = nullptr; // [64] misc device scalars (counts)
  uint64_t* h_scalars = nullptr; // pinned [64+4096]
  hipEvent_t stage_ev = nullptr; // recorded behind the last asynchronous upload out of the pinned staging block (h_scalars + 64)
  // view storage for tbk_groups_to_cov_in
  char* d_view = nullptr;
  size_t d_view_cap = 0;
  // what tbk_coverage_tile's first pass (cov_prep_k) would compute from the view, left behind by the view builder that had the
  // CIGAR words in registers anyway (valid: the view of `cig` / `n` records is the context's current one and was built from keys)
  struct ViewPrep {
    bool valid = false;
    const void* cig = nullptr;
    uint32_t n = 0;
    int32_t *start = nullptr, *end = nullptr, *yi = nullptr;
    uint32_t *jcnt = nullptr, *ridx = nullptr;
    uint64_t n_bases = 0, sum_abs = 0, n_junc = 0;  // n_junc: junction items (sum of jcnt)
    uint32_t err = 0;  // TBK_DERR_FATALOP / _NCIGAR (raised only when intervals are wanted) / _FRACTIONAL
  } view_prep;
  bool profiling = false;
  std::vector<KTime> ktimes;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<tbk_kernel_time> last_times;
  std::string last_error;
  int num_cu = 256;
  // deferred YD stage (tbk_collapse_opts.defer_yd): a private side context + helper thread
  void* yd_job = nullptr;        // prepared by tbk_collapse_device, consumed by tbk_collapse_yd_run
  tbk_ctx* yd_ctx = nullptr;
  TbkWorker* yd_worker = nullptr;  // created with the side context
  bool yd_pending = false;        // a deferred YD stage is running (or finished and not yet collected)
  int yd_rc = 0;
  size_t ws_base_off = 0;        // arena bytes pinned while a deferred YD stage still reads the main stage's arrays
  // side context for a branch that runs beside the main stream inside one call (tiecov junctions)
  tbk_ctx* side_ctx = nullptr;
  TbkWorker* side_worker = nullptr;
  bool side_times_pending = false;
  std::vector<void*> registered; // caller's host ranges page-locked for the copies of the current call (tbk_api.hip: host_register)
  char* d_unpack = nullptr;      // the tile tbk_unpack_tile rebuilt from its packed wire form (pack.hip)
  size_t d_unpack_cap = 0;
  std::vector<uint8_t> unpack_tbm;
  void* bam_dev = nullptr;       // device-decoded BAM input (bamdev.hip): inflated streams + record index + the SoA tile's arrays
  void* enc = nullptr;           // the encoder's device buffers (bgzdef.hip)
  void* stager = nullptr;        // pinned ring + upload stream of the staged host -> device copies (bamdev.hip)
  char* bounce = nullptr;        // a few megabytes of page-locked memory: results of moderate size come back through it (tbk_api.hip: d2h)
  // the last collapse's results, kept for the output side (tbk_collapse_opts.keep_results): one allocation, four columns of kept_n groups
  char* kept = nullptr;
  size_t kept_cap = 0;
  uint32_t kept_n = 0;
  uint32_t* kept_rep = nullptr;
  double* kept_yc = nullptr;
  int64_t* kept_yx = nullptr;
  int32_t* kept_yd = nullptr;
};
void tbk_stager_free(tbk_ctx* ctx);
void tbk_enc_free(tbk_ctx* ctx);
// the tile tbk_bam_decode left on the context: inflated streams, record offsets, record count (false: there is none)
bool tbk_bam_dev_records(tbk_ctx* ctx, const uint8_t** inf, const uint64_t** rec, uint32_t* n);

// side context (created on first use; nullptr if that fails -> the caller runs the branch inline), and the call
// bracket a branch thread puts around its work on it
hipStream_t tbk_aux_stream(tbk_ctx* ctx);  // nullptr if it cannot be created -> the caller stays on one stream
tbk_ctx* tbk_side_ctx(tbk_ctx* ctx);
int tbk_side_begin(tbk_ctx* side, size_t arena_hint);
void tbk_side_end(tbk_ctx* side);

#define TBK_HIP(call)                                                                            \
  do {                                                                                           \
    hipError_t _e = (call);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      char _b[512];                                                                              \
      snprintf(_b, sizeof(_b), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      ctx->last_error = _b;                                                                      \
      return TBK_EHIP;                                                                           \
    }                                                                                            \
  } while (0)

#define TBK_TRY(expr)        \
  do {                       \
    int _rc = (expr);        \
    if (_rc != 0) return _rc; \
  } while (0)

// ---- workspace ------------------------------------------------------------------
int tbk_ws_reserve(tbk_ctx* ctx, size_t bytes);  // grow (frees + reallocates) if needed; resets the bump pointer
int tbk_ws_presize(tbk_ctx* ctx, size_t bytes);  // best effort: a larger arena if one can be had, the old one kept otherwise; always 0
void* tbk_ws_alloc_raw(tbk_ctx* ctx, size_t bytes);
template <class T>
static inline T* ws_alloc(tbk_ctx* ctx, size_t n) {
  return (T*)tbk_ws_alloc_raw(ctx, n * sizeof(T));
}

// ---- launches --------------------------------------------------------------------
hipEvent_t tbk_event(tbk_ctx* ctx);
static inline void tbk_prof_begin(tbk_ctx* ctx, const char* name) {
  if (!ctx->profiling) return;
  KTime k{name, tbk_event(ctx), tbk_event(ctx)};
  (void)hipEventRecord(k.a, ctx->stream);
  ctx->ktimes.push_back(k);
}
static inline void tbk_prof_end(tbk_ctx* ctx) {
  if (!ctx->profiling) return;
  (void)hipEventRecord(ctx->ktimes.back().b, ctx->stream);
}

#define TBK_LAUNCH(ctx, name, kern, grid, block, shmem, ...)                          \
  do {                                                                                \
    tbk_prof_begin(ctx, name);                                                        \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), shmem, (ctx)->stream, __VA_ARGS__); \
    tbk_prof_end(ctx);                                                                \
  } while (0)

void tbk_prof_begin_call(tbk_ctx* ctx);  // brackets of one API call for the per-kernel event timing (tbk_set_profiling)
void tbk_prof_end_call(tbk_ctx* ctx);
int tbk_check_launch(tbk_ctx* ctx, const char* what);  // hipGetLastError -> TBK_EHIP
extern "C" void tbk_bam_release(tbk_ctx* ctx);
// small host tables go to the device through the context's pinned staging block: tbk_stage_acquire waits until the previous
// upload out of it has run (a no-op almost always) and returns the block, tbk_stage_release marks the upload just queued
void* tbk_stage_acquire(tbk_ctx* ctx);
void tbk_stage_release(tbk_ctx* ctx);
int tbk_sync_err(tbk_ctx* ctx, uint32_t* err_bits);    // d_scalars[0..15] -> h_scalars, stream sync, error bits
int tbk_derr_to_status(tbk_ctx* ctx, uint32_t bits);

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ---- primitives implemented in prims.hip -------------------------------------------
// exclusive prefix sum of u32 -> u32 (n may be 0); total (u64) written to d_total if non-null
int tbk_exscan_u32(tbk_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t n, uint64_t* d_total);
// exclusive prefix sum of u32 -> u64
int tbk_exscan_u32_u64(tbk_ctx* ctx, const uint32_t* in, uint64_t* out, uint32_t n, uint64_t* d_total);

// LSD radix sort of (hi,lo) 128-bit keys with a u32 payload; result left in *hi/*lo/*val
// (the routine ping-pongs between the given buffers and swaps the pointers for the caller).
struct SortBufs {
  uint64_t *hi, *lo;
  uint32_t* val;
  uint64_t *hi2, *lo2;
  uint32_t* val2;
};
// only_hi / only_lo: bits that take part in the ordering (the rest are payload that must not reorder equal keys).
// masks_are_exact: the caller knows which bits can differ (e.g. a small id range) — no scan of the keys, no read-back
int tbk_radix_sort128(tbk_ctx* ctx, SortBufs* b, uint32_t n, uint64_t only_hi = ~0ull, uint64_t only_lo = ~0ull,
                      bool masks_are_exact = false);
size_t tbk_radix_ws_bytes(uint32_t n);
// stable sort of 64-bit words by the bits of `mask` (mask_is_exact: no scan for the bits that really vary, no read-back); result in
// *w (swapped with *w2 as the passes go)
int tbk_radix_sort_w64(tbk_ctx* ctx, uint64_t** w, uint64_t** w2, uint32_t n, uint64_t mask, bool mask_is_exact);
// same result for an input made of `nruns` position-sorted runs (msort.hip)
int tbk_sort_runs(tbk_ctx* ctx, SortBufs* b, uint32_t n_hi, const uint32_t* d_run_off, uint32_t nruns, uint32_t* err,
                  uint32_t* nbig_zeroed);

// ---- pipelines --------------------------------------------------------------------
int tbk_collapse_device(tbk_ctx* ctx, const tbk_collapse_opts* o, const tbk_soa_in* in, tbk_groups_out* out);
int tbk_collapse_warm(tbk_ctx* ctx);  // tbk_warmup: the first-use costs of the YD stage's machines
int tbk_collapse_yd_run(tbk_ctx* run_on, void* job);  // consumes ctx->yd_job (prepared by tbk_collapse_device)
int tbk_coverage_device(tbk_ctx* ctx, const tbk_cov_in* in, tbk_cov_out* out);
// tiecov's input view of ng representatives (tbk_groups_to_cov_in's body; the arena must be reserved by the caller)
int tbk_cov_view_build(tbk_ctx* ctx, const int32_t* r_tid, const int32_t* r_pos, const uint8_t* r_strand, const uint32_t* r_cig_off,
                       const uint32_t* r_cig, const uint32_t* g_rep, const double* g_yc, const int64_t* g_yx, uint32_t ng, tbk_cov_in* view,
                       const uint64_t* g_key = nullptr);
int tbk_sample_device(tbk_ctx* ctx, const tbk_cov_in* in, int32_t num_samples, tbk_sample_out* out);
