// tbk_api.hip — the extern "C" boundary declared in include/tbk.h: context lifetime, workspace
// arena, host<->device staging for TBK_MEM_HOST callers, error mapping.  The pipelines live in
// collapse.hip / cov.hip; primitives in prims.hip.
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>
#include <new>

#include "tbk_internal.h"

// ---- workspace arena ------------------------------------------------------------------------------
// One big hipMalloc'd block, bump-allocated per API call.  If a call outgrows it, overflow chunks are
// chained for that call and the arena is re-created at the combined size at the start of the next
// call, so a steady-state loop performs no allocation at all.
// Host ranges page-locked for the copies of a call are shared by the contexts of the process: two contexts (or threads) copying
// from the same pageable arrays take one registration with a count, and the range is unpinned when the last user's copies have
// drained — a context never relies on a pin another context may drop under its copy.
namespace {
std::mutex g_reg_m;
struct RegEntry {
  void* p;
  int users;
};
std::vector<RegEntry> g_reg;
}  // namespace

static void release_registered(tbk_ctx* ctx) {
  if (ctx->registered.empty()) return;
  (void)hipStreamSynchronize(ctx->stream);  // the copies out of / into the ranges have to be done first
  const auto r0 = std::chrono::steady_clock::now();
  struct RP {
    tbk_ctx* c;
    std::chrono::steady_clock::time_point t;
    size_t n;
    ~RP() {
      if (c->dbg.phases) fprintf(stderr, "collapse phases: %zu ranges unregistered in %.1f ms\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count());
    }
  } rp{ctx, r0, ctx->registered.size()};
  std::lock_guard<std::mutex> lk(g_reg_m);
  for (void* p : ctx->registered)
    for (size_t i = 0; i < g_reg.size(); ++i)
      if (g_reg[i].p == p) {
        if (--g_reg[i].users == 0) {
          (void)hipHostUnregister(p);
          g_reg[i] = g_reg.back();
          g_reg.pop_back();
        }
        break;
      }
  ctx->registered.clear();
}
// every exit path of an API call that may have registered host ranges gives them back (a call that returns early through
// TBK_TRY must not leave the caller's memory pinned: the caller may free it before this context's next call)
struct RegGuard {
  tbk_ctx* ctx;
  ~RegGuard() { release_registered(ctx); }
};

static int ws_begin_call(tbk_ctx* ctx, size_t hint) {
  release_registered(ctx);  // (nothing to do unless a side path left something behind)
  ctx->ws_top = 0;
  ctx->ws_top_mode = false;
  if (ctx->ws_borrowed) {  // a range of another context's arena: what does not fit goes to overflow chunks, the range stays
    for (auto& c : ctx->ws_overflow) (void)hipFree(c.first);
    ctx->ws_overflow.clear();
    ctx->ws_off = 0;
    ctx->ws_over_used = 0;
    return 0;
  }
  size_t want = std::max(ctx->ws_cap, hint);
  if (ctx->yd_pending) {
    // a deferred YD stage still reads arrays in [0, ws_base_off): never move the arena now; this call bump-allocates
    // behind the pinned prefix and spills into overflow chunks if it must
    for (auto& c : ctx->ws_overflow) (void)hipFree(c.first);
    ctx->ws_overflow.clear();
    ctx->ws_off = ctx->ws_base_off;
    ctx->ws_over_used = 0;
    return 0;
  }
  if (!ctx->ws_overflow.empty()) {
    size_t tot = ctx->ws_cap;
    for (auto& c : ctx->ws_overflow) {
      tot += c.second;
      (void)hipFree(c.first);
    }
    ctx->ws_overflow.clear();
    want = std::max(want, tot + tot / 8);
  }
  if (want > ctx->ws_cap) {
    if (ctx->ws) (void)hipFree(ctx->ws);
    ctx->ws = nullptr;
    ctx->ws_cap = 0;
    TBK_HIP(hipMalloc((void**)&ctx->ws, want));
    ctx->ws_cap = want;
  }
  ctx->ws_off = ctx->ws_base_off;
  ctx->ws_over_used = 0;
  return 0;
}

int tbk_ws_reserve(tbk_ctx* ctx, size_t bytes) { return ws_begin_call(ctx, bytes); }

// tbk_ws_presize — best effort (tbk_reserve_tile is "purely an optimisation"): the larger arena is allocated FIRST and the old one freed only when that
// worked, so a reserve that does not fit leaves the context exactly as it was — no sticky HIP error, the old arena still there — and
// the collapse that follows sizes (or spills) by its own rules.  Nothing is resized while a deferred YD stage or a borrowed range pins
// the arena.
int tbk_ws_presize(tbk_ctx* ctx, size_t bytes) {
  if (ctx->dbg.phases)
    fprintf(stderr, "collapse phases: arena presize to %.2f GB asked (now %.2f GB; borrowed %d, YD pending %d, overflow chunks %zu)\n", bytes / 1e9, ctx->ws_cap / 1e9,
            (int)ctx->ws_borrowed, (int)ctx->yd_pending, ctx->ws_overflow.size());
  if (ctx->ws_borrowed || ctx->yd_pending) return 0;
  // (overflow chunks are the previous call's, which has returned: they go now — ws_begin_call would free them at the next call anyway —
  // and count towards what the arena has learnt it needs.  A call that spilled used to leave the reserve a no-op: the device decode of
  // the hybrid path does, and the collapse behind it then sized the arena inside the call)
  if (!ctx->ws_overflow.empty()) {
    size_t tot = ctx->ws_cap;
    for (auto& c : ctx->ws_overflow) {
      tot += c.second;
      (void)hipFree(c.first);
    }
    ctx->ws_overflow.clear();
    ctx->ws_over_used = 0;
    bytes = std::max(bytes, tot + tot / 8);
  }
  if (bytes <= ctx->ws_cap) return 0;
  char* bigger = nullptr;
  const auto a0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMalloc((void**)&bigger, bytes);
  if (ctx->dbg.phases)
    fprintf(stderr, "collapse phases: arena presize: hipMalloc %s in %.1f ms\n", e == hipSuccess ? "ok" : hipGetErrorString(e),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count());
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (ctx->ws) (void)hipFree(ctx->ws);
  ctx->ws = bigger;
  ctx->ws_cap = bytes;
  return 0;
}

void* tbk_ws_alloc_raw(tbk_ctx* ctx, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  const size_t top_end = ctx->ws_cap & ~(size_t)255;  // (the arena's size is whatever a call asked for: the end that allocations count from is aligned)
  if (ctx->ws_overflow.empty() && ctx->ws_off + ctx->ws_top + bytes <= top_end) {
    if (ctx->ws_top_mode) {  // a temporary of the current stage: from the end of the arena
      ctx->ws_top += bytes;
      return ctx->ws + (top_end - ctx->ws_top);
    }
    void* p = ctx->ws + ctx->ws_off;
    ctx->ws_off += bytes;
    return p;
  }
  // overflow chunk (only while the arena is still learning its steady-state size)
  if (!ctx->ws_overflow.empty()) {
    auto& c = ctx->ws_overflow.back();
    if (ctx->ws_over_used + bytes <= c.second) {
      void* p = c.first + ctx->ws_over_used;
      ctx->ws_over_used += bytes;
      return p;
    }
  }
  size_t cap = std::max(bytes, (size_t)64 << 20);
  char* p = nullptr;
  if (hipMalloc((void**)&p, cap) != hipSuccess) {
    ctx->last_error = "workspace hipMalloc failed";
    return nullptr;
  }
  ctx->ws_overflow.push_back({p, cap});
  ctx->ws_over_used = bytes;
  return p;
}

hipEvent_t tbk_event(tbk_ctx* ctx) {
  if (ctx->ev_used == ctx->ev_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    ctx->ev_pool.push_back(e);
  }
  return ctx->ev_pool[ctx->ev_used++];
}

void* tbk_stage_acquire(tbk_ctx* ctx) {
  if (ctx->stage_ev) (void)hipEventSynchronize(ctx->stage_ev);
  return ctx->h_scalars + 64;
}
void tbk_stage_release(tbk_ctx* ctx) {
  if (!ctx->stage_ev && hipEventCreateWithFlags(&ctx->stage_ev, hipEventDisableTiming) != hipSuccess) {
    ctx->stage_ev = nullptr;
    (void)hipStreamSynchronize(ctx->stream);  // no event: the upload has run before the block is touched again
    return;
  }
  (void)hipEventRecord(ctx->stage_ev, ctx->stream);
}

int tbk_check_launch(tbk_ctx* ctx, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return TBK_EHIP;
  }
  return 0;
}

int tbk_sync_err(tbk_ctx* ctx, uint32_t* err_bits) {
  // the error word is scalar 15: one copy brings the counters of the stage and the error bits
  TBK_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  *err_bits = (uint32_t)ctx->h_scalars[15];
  return 0;
}

int tbk_derr_to_status(tbk_ctx* ctx, uint32_t bits) {
  char b[128];
  snprintf(b, sizeof(b), "device error bits 0x%x", bits);
  ctx->last_error = b;
  if (bits & TBK_DERR_UNSORTED) return TBK_EUNSORTED;
  if (bits & TBK_DERR_FATALOP) return TBK_EFATALOP;
  if (bits & TBK_DERR_NCIGAR) return TBK_EUNSUPPORTED;
  if (bits & TBK_DERR_SPAN) return TBK_EINVAL;
  if (bits & TBK_DERR_COLLISION) return TBK_ECOLLISION;
  if (bits & TBK_DERR_OVERFLOW) return TBK_E2BIG;
  return TBK_EHIP;
}

static void merge_side_times(tbk_ctx* ctx);
void tbk_prof_begin_call(tbk_ctx* ctx) {
  ctx->ktimes.clear();
  ctx->ev_used = 0;
}
void tbk_prof_end_call(tbk_ctx* ctx) {
  if (false) {
    size_t over = 0;
    for (auto& c : ctx->ws_overflow) over += c.second;
    fprintf(stderr, "tbk arena %p: used %.2f GB of %.2f GB (+ %.2f GB in overflow chunks)\n", (void*)ctx, ctx->ws_off / 1e9, ctx->ws_cap / 1e9, over / 1e9);
  }
  release_registered(ctx);  // host ranges page-locked for this call's copies (host_register)
  ctx->last_times.clear();
  if (!ctx->profiling) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& k : ctx->ktimes) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, k.a, k.b) != hipSuccess) continue;
    bool found = false;
    for (auto& t : ctx->last_times)
      if (strcmp(t.name, k.name) == 0) {
        t.ms += ms;
        t.launches++;
        found = true;
        break;
      }
    if (!found) ctx->last_times.push_back({k.name, ms, 1});
  }
  merge_side_times(ctx);
}

// ---- staging helpers for TBK_MEM_HOST ---------------------------------------------------------------
// A large pageable source is registered (page-locked) for the duration of the call: the copy is then one DMA at the link's rate
// instead of a trip through the runtime's bounce buffers, and nothing of the caller's memory stays pinned behind the call
// (TBK_NO_REGISTER: test hook, plain copies).  tbk_host_unregister_all releases the registrations once the stream has drained.
static void host_register(tbk_ctx* ctx, const void* p, size_t bytes) {
  if (ctx->dbg.no_register || bytes < ((size_t)16 << 20)) return;
  for (void* q : ctx->registered)
    if (q == p) return;  // this call holds it already
  std::lock_guard<std::mutex> lk(g_reg_m);
  for (auto& e : g_reg)
    if (e.p == p) {  // registered by another context of this process: share it, it stays until the last user lets go
      ++e.users;
      ctx->registered.push_back(const_cast<void*>(p));
      return;
    }
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type != hipMemoryTypeUnregistered) return;  // pinned by its owner (tbk_host_alloc, the caller)
  (void)hipGetLastError();
  if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess) {
    g_reg.push_back({const_cast<void*>(p), 1});
    ctx->registered.push_back(const_cast<void*>(p));
  } else {
    (void)hipGetLastError();
  }
}
template <class T>
static int h2d(tbk_ctx* ctx, const T* src, size_t n, const T** dst) {
  *dst = nullptr;
  if (!src) return 0;
  T* d = ws_alloc<T>(ctx, n ? n : 1);
  if (!d) return TBK_ENOMEM;
  if (n) {
    host_register(ctx, src, n * sizeof(T));
    TBK_HIP(hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  }
  *dst = d;
  return 0;
}
template <class T>
static int dalloc(tbk_ctx* ctx, T* host, size_t n, T** dst) {
  *dst = nullptr;
  if (!host) return 0;
  T* d = ws_alloc<T>(ctx, n ? n : 1);
  if (!d) return TBK_ENOMEM;
  *dst = d;
  return 0;
}
// Results of moderate size come back through a small page-locked buffer of the context and a copy by the core.  Page-locking the
// caller's fresh array for one copy looks cheaper (0.04 ms per MB) but is not reliably so: late in a process that holds gigabytes the
// copy call itself was measured to block 10-20 ms for 20 MB (and 100 ms and more now and then), registered or not — the buffer's pages
// are pinned once, when the context is young.
constexpr size_t kBounce = (size_t)8 << 20;
static bool bounce_ready(tbk_ctx* ctx) {
  if (!ctx->bounce && hipHostMalloc((void**)&ctx->bounce, kBounce) != hipSuccess) {
    (void)hipGetLastError();
    ctx->bounce = nullptr;
  }
  return ctx->bounce != nullptr;
}
static int d2h_bounce(tbk_ctx* ctx, char* host, const char* dev, size_t bytes) {
  // One chunk at a time: a second copy queued while the first is in flight is given ANOTHER copy engine, whose first use in the process
  // costs 7-8 ms (measured: HIP's log of the command line) — more than the overlap of 80 us of DMA with 150 us of memcpy can save.
  double t_q = 0, t_c = 0;  // (phases=1: the copies, the core's part)
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (size_t off = 0; off < bytes; off += kBounce) {
    const size_t len = std::min(kBounce, bytes - off);
    const auto a = now();
    TBK_HIP(hipMemcpyAsync(ctx->bounce, dev + off, len, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    const auto b = now();
    memcpy(host + off, ctx->bounce, len);
    t_q += ms(a, b), t_c += ms(b, now());
  }
  if (ctx->dbg.phases) fprintf(stderr, "collapse phases: %.1f MB through the staging buffer: copies %.1f | the core's part %.1f ms\n", bytes / 1e6, t_q, t_c);
  return 0;
}
template <class T>
static int d2h(tbk_ctx* ctx, T* host, const T* dev, size_t n) {
  if (!host || !dev || !n) return 0;
  const size_t bytes = n * sizeof(T);
  if (bytes >= ((size_t)1 << 20) && bytes <= ((size_t)64 << 20) && !ctx->dbg.no_bounce) {
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, host) == hipSuccess && at.type != hipMemoryTypeUnregistered;
    (void)hipGetLastError();
    if (!pinned && bounce_ready(ctx)) return d2h_bounce(ctx, (char*)host, (const char*)dev, bytes);
  }
  host_register(ctx, host, n * sizeof(T));
  TBK_HIP(hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}

template <class T>
static int dalloc_if(tbk_ctx* ctx, bool want, size_t n, T** dst) {
  *dst = nullptr;
  if (!want) return 0;
  T* d = ws_alloc<T>(ctx, n ? n : 1);
  if (!d) return TBK_ENOMEM;
  *dst = d;
  return 0;
}
// the device arrays a TBK_MEM_HOST caller's results are computed into (keep: all four result columns, whatever the caller takes back)
static int out_dalloc(tbk_ctx* ctx, const tbk_groups_out* out, size_t cap, size_t n, bool keep, tbk_groups_out* dout) {
  dout->mem = TBK_MEM_DEVICE;
  TBK_TRY(dalloc_if(ctx, keep || out->rep, cap, &dout->rep));
  TBK_TRY(dalloc_if(ctx, keep || out->yc, cap, &dout->yc));
  TBK_TRY(dalloc_if(ctx, keep || out->yx, cap, &dout->yx));
  TBK_TRY(dalloc_if(ctx, keep || out->yd, cap, &dout->yd));
  TBK_TRY(dalloc(ctx, out->g_start, cap, &dout->g_start));
  TBK_TRY(dalloc(ctx, out->g_end, cap, &dout->g_end));
  TBK_TRY(dalloc(ctx, out->rec_group, n, &dout->rec_group));
  TBK_TRY(dalloc(ctx, out->rep_effend, cap, &dout->rep_effend));
  TBK_TRY(dalloc(ctx, out->g_key, 2 * cap, &dout->g_key));
  return 0;
}
static int out_d2h(tbk_ctx* ctx, tbk_groups_out* out, const tbk_groups_out* dout, size_t n) {
  const size_t g = dout->n_groups;

  TBK_TRY(d2h(ctx, out->rep, dout->rep, g));
  TBK_TRY(d2h(ctx, out->yc, dout->yc, g));
  TBK_TRY(d2h(ctx, out->yx, dout->yx, g));
  TBK_TRY(d2h(ctx, out->yd, dout->yd, g));
  TBK_TRY(d2h(ctx, out->g_start, dout->g_start, g));
  TBK_TRY(d2h(ctx, out->g_end, dout->g_end, g));
  TBK_TRY(d2h(ctx, out->rec_group, dout->rec_group, n));
  TBK_TRY(d2h(ctx, out->rep_effend, dout->rep_effend, g));
  TBK_TRY(d2h(ctx, out->g_key, dout->g_key, 2 * g));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
// tbk_collapse_opts.keep_results: the call's rep / yc / yx / yd (device arrays, final) copied into the context's own allocation — 24 bytes
// per group, device to device — where tbk_bam_encode and tbk_kept_results find them until the next collapse
static int keep_results(tbk_ctx* ctx, const tbk_groups_out* d) {
  const size_t g = d->n_groups;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t need = 2 * al(g * 8) + 2 * al(g * 4) + 256;
  if (need > ctx->kept_cap) {
    if (ctx->kept) (void)hipFree(ctx->kept);
    ctx->kept = nullptr;
    ctx->kept_cap = 0;
    if (hipMalloc((void**)&ctx->kept, need + need / 8) != hipSuccess) {
      (void)hipGetLastError();
      ctx->last_error = "keep_results: hipMalloc failed";
      return TBK_ENOMEM;
    }
    ctx->kept_cap = need + need / 8;
  }
  char* p = ctx->kept;
  ctx->kept_yc = (double*)p, p += al(g * 8);
  ctx->kept_yx = (int64_t*)p, p += al(g * 8);
  ctx->kept_rep = (uint32_t*)p, p += al(g * 4);
  ctx->kept_yd = (int32_t*)p;
  if (g) {
    TBK_HIP(hipMemcpyAsync(ctx->kept_yc, d->yc, g * 8, hipMemcpyDeviceToDevice, ctx->stream));
    TBK_HIP(hipMemcpyAsync(ctx->kept_yx, d->yx, g * 8, hipMemcpyDeviceToDevice, ctx->stream));
    TBK_HIP(hipMemcpyAsync(ctx->kept_rep, d->rep, g * 4, hipMemcpyDeviceToDevice, ctx->stream));
    TBK_HIP(hipMemcpyAsync(ctx->kept_yd, d->yd, g * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  ctx->kept_n = (uint32_t)g;
  return 0;
}

hipStream_t tbk_aux_stream(tbk_ctx* ctx) {
  if (!ctx->aux) {
    if (hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess) ctx->aux = nullptr;
    if (ctx->aux && hipEventCreateWithFlags(&ctx->aux_done, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(ctx->aux);
      ctx->aux = nullptr;
    }
  }
  return ctx->aux;
}

// ---- side context ----------------------------------------------------------------------------------------
tbk_ctx* tbk_side_ctx(tbk_ctx* ctx) {
  if (!ctx->side_ctx && tbk_create(ctx->device, &ctx->side_ctx) != 0) ctx->side_ctx = nullptr;
  if (ctx->side_ctx) {
    ctx->side_ctx->profiling = ctx->profiling;
    ctx->side_ctx->dbg = ctx->dbg;
  }
  return ctx->side_ctx;
}
int tbk_side_begin(tbk_ctx* side, size_t arena_hint) {
  if (hipSetDevice(side->device) != hipSuccess) return TBK_EHIP;
  tbk_prof_begin_call(side);
  return ws_begin_call(side, arena_hint);
}
void tbk_side_end(tbk_ctx* side) { tbk_prof_end_call(side); }

// kernel times of a side branch are reported with the call that ran it
static void merge_side_times(tbk_ctx* ctx) {
  if (!ctx->side_times_pending || !ctx->side_ctx) return;
  ctx->side_times_pending = false;
  for (auto& t : ctx->side_ctx->last_times) {
    bool found = false;
    for (auto& u : ctx->last_times)
      if (strcmp(u.name, t.name) == 0) {
        u.ms += t.ms;
        u.launches += t.launches;
        found = true;
        break;
      }
    if (!found) ctx->last_times.push_back(t);
  }
}

// ---- deferred YD stage -------------------------------------------------------------------------------
static int yd_stage_on(tbk_ctx* run_on, void* job, size_t hint) {
  if (hipSetDevice(run_on->device) != hipSuccess) return TBK_EHIP;
  tbk_prof_begin_call(run_on);
  int rc = ws_begin_call(run_on, hint);
  if (rc == 0) rc = tbk_collapse_yd_run(run_on, job);
  tbk_prof_end_call(run_on);
  return rc;
}

static int finish_yd(tbk_ctx* ctx) {
  if (!ctx->yd_pending) return 0;
  ctx->yd_worker->wait();
  ctx->yd_pending = false;
  ctx->ws_base_off = 0;
  if (ctx->yd_ctx) {
    ctx->last_times = ctx->yd_ctx->last_times;  // tbk_kernel_times() right after the wait = the YD stage
    if (ctx->yd_rc != 0) ctx->last_error = ctx->yd_ctx->last_error;
  }
  return ctx->yd_rc;
}

// ---- C ABI -----------------------------------------------------------------------------------------
void tbk_debug_parse(const char* spec, TbkDebug* out) {
  *out = TbkDebug();
  if (!spec) return;
  std::string s(spec);
  size_t at = 0;
  while (at < s.size()) {
    size_t end = s.find(',', at);
    if (end == std::string::npos) end = s.size();
    const std::string kv = s.substr(at, end - at);
    at = end + 1;
    const size_t eq = kv.find('=');
    const std::string k = kv.substr(0, eq), v = eq == std::string::npos ? std::string("1") : kv.substr(eq + 1);
    const unsigned long long u = strtoull(v.c_str(), nullptr, 0);
    const bool on = v != "0";
    if (k == "path") out->path = v == "sort" ? 1 : (v == "window" ? 2 : 0);
    else if (k == "raw") out->raw = on ? 1 : 0;
    else if (k == "sort") out->sort = v == "radix" ? 1 : (v == "runs" ? 2 : 0);
    else if (k == "scan") out->scan = v == "lookback" ? 1 : (v == "3pass" ? 2 : 0);
    else if (k == "hash_mask") out->hash_mask = (uint32_t)u;
    else if (k == "qhash_mask") out->qhash_mask = u;
    else if (k == "yd_wave_min") out->yd_wave_min = (uint32_t)u;
    else if (k == "yd_bgrid") out->yd_bgrid = (uint32_t)u;
    else if (k == "yd_radix") out->yd_radix = on;
    else if (k == "yd_literal") out->yd_literal = on;
    else if (k == "yd_own_arena") out->yd_own_arena = on;
    else if (k == "wg_dense_verify") out->wg_dense_verify = on;
    else if (k == "wg_rank_merge") out->wg_rank_merge = on;
    else if (k == "cov_legacy") out->cov_legacy = on;
    else if (k == "cov_bundle_scan") out->cov_bundle_scan = on;
    else if (k == "cov_prep") out->cov_prep = on;
    else if (k == "cov_tile_cap") out->cov_tile_cap = u;
    else if (k == "junc_radix") out->junc_radix = on;
    else if (k == "no_junc_agg") out->no_junc_agg = on;
    else if (k == "jh_cap") out->jh_cap = (uint32_t)u;
    else if (k == "index_chain") out->index_chain = on;
    else if (k == "no_register") out->no_register = on;
    else if (k == "phases") out->phases = on;
    else if (k == "no_bounce") out->no_bounce = on;
  }
}

extern "C" {

int tbk_abi_version(void) { return TBK_ABI_VERSION; }

int tbk_set_debug(tbk_ctx* ctx, const char* spec) {
  if (!ctx) return TBK_EINVAL;
  tbk_debug_parse(spec, &ctx->dbg);
  if (ctx->yd_ctx) ctx->yd_ctx->dbg = ctx->dbg;
  if (ctx->side_ctx) ctx->side_ctx->dbg = ctx->dbg;
  return 0;
}

const char* tbk_strerror(int s) {
  switch (s) {
    case TBK_OK: return "ok";
    case TBK_EINVAL: return "invalid argument";
    case TBK_ENOMEM: return "out of memory";
    case TBK_EHIP: return "HIP runtime error";
    case TBK_E2BIG: return "output capacity too small";
    case TBK_EUNSUPPORTED: return "option or input outside the supported/pinned semantics";
    case TBK_EUNSORTED: return "input not coordinate-sorted";
    case TBK_EFATALOP: return "unknown opcode (tiecov accepts only M/I/D/N/S)";
    case TBK_ECOLLISION: return "key hash collision survived all reseeds";
    case TBK_ENODEVICE: return "no usable gfx950 device";
  }
  return "unknown status";
}

int tbk_create(int device_ordinal, tbk_ctx** out) {
  if (!out) return TBK_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_ordinal < 0 || device_ordinal >= ndev) return TBK_ENODEVICE;
  tbk_ctx* ctx = new (std::nothrow) tbk_ctx();
  if (!ctx) return TBK_ENOMEM;
  ctx->device = device_ordinal;
  tbk_debug_parse(getenv("TBK_DEBUG"), &ctx->dbg);  // (the one place the library looks at the environment)
  bool ok = hipSetDevice(device_ordinal) == hipSuccess;
  hipDeviceProp_t prop;
  if (ok && hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  ok = ok && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) == hipSuccess;
  ctx->stream = ctx->own_stream;
  ok = ok && hipMalloc((void**)&ctx->d_scalars, 64 * sizeof(uint64_t)) == hipSuccess;
  if (ok) ctx->d_err = (uint32_t*)(ctx->d_scalars + 15);
  ok = ok && hipHostMalloc((void**)&ctx->h_scalars, (64 + 4096) * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
  if (!ok) {
    tbk_destroy(ctx);
    return TBK_EHIP;
  }
  *out = ctx;
  return 0;
}

int tbk_collapse_finish_yd(tbk_ctx* ctx) {
  if (!ctx) return TBK_EINVAL;
  return finish_yd(ctx);
}

void tbk_destroy(tbk_ctx* ctx) {
  if (!ctx) return;
  (void)finish_yd(ctx);
  tbk_bam_release(ctx);
  delete ctx->yd_worker;
  ctx->yd_worker = nullptr;
  delete ctx->side_worker;
  ctx->side_worker = nullptr;
  if (ctx->yd_ctx) {
    tbk_destroy(ctx->yd_ctx);
    ctx->yd_ctx = nullptr;
  }
  if (ctx->side_ctx) {
    tbk_destroy(ctx->side_ctx);
    ctx->side_ctx = nullptr;
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  release_registered(ctx);
  for (auto& c : ctx->ws_overflow) (void)hipFree(c.first);
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->ws && !ctx->ws_borrowed) (void)hipFree(ctx->ws);
  if (ctx->d_view) (void)hipFree(ctx->d_view);
  if (ctx->d_unpack) (void)hipFree(ctx->d_unpack);
  if (ctx->kept) (void)hipFree(ctx->kept);
  if (ctx->bounce) (void)hipHostFree(ctx->bounce);
  tbk_enc_free(ctx);
  tbk_stager_free(ctx);
  if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
  if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
  if (ctx->aux_done) (void)hipEventDestroy(ctx->aux_done);
  if (ctx->stage_ev) (void)hipEventDestroy(ctx->stage_ev);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* tbk_last_error(const tbk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int tbk_set_stream(tbk_ctx* ctx, void* s) {
  if (!ctx) return TBK_EINVAL;
  // (TBK_STREAM_DEFAULT: the device's default stream, whose handle is the null pointer that otherwise says "the context's own")
  ctx->stream = s == TBK_STREAM_DEFAULT ? (hipStream_t) nullptr : (s ? (hipStream_t)s : ctx->own_stream);
  return 0;
}
void* tbk_get_stream(tbk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int tbk_set_profiling(tbk_ctx* ctx, int enabled) {
  if (!ctx) return TBK_EINVAL;
  ctx->profiling = enabled != 0;
  return 0;
}

int tbk_kernel_times(tbk_ctx* ctx, tbk_kernel_time* out, int cap) {
  if (!ctx) return TBK_EINVAL;
  int n = (int)ctx->last_times.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = ctx->last_times[i];
  return n;
}

// Page-locked host memory.  hipHostMalloc costs 0.18 ms per MB to pin and — what a short run pays AFTER its last instruction, while its
// caller waits for the process to go — 0.17 ms per MB to tear down at exit (0.13 through hipHostFree): the 400 MB of staging a `tiebrush`
// run holds were 0.08 s of its 0.8 s (tools/micro/exit_cost.hip, measured on the MI355X box).  Anonymous memory on transparent huge
// pages, registered with hipHostRegister, is the same to the copy engines and costs 0.04 ms per MB either way.  Blocks of 4 MB and more
// take that form (hipHostMalloc when the registration is refused); the table remembers which.
namespace {
struct PinnedBlk {
  void* raw;
  size_t raw_bytes;
};
std::mutex g_pin_m;
std::vector<std::pair<void*, PinnedBlk>> g_pin;
constexpr size_t PIN_HUGE = (size_t)2 << 20;
}  // namespace

int tbk_host_alloc(size_t bytes, void** out) {
  if (!out) return TBK_EINVAL;
  *out = nullptr;
  if (bytes >= 2 * PIN_HUGE) {
    const size_t n = (bytes + PIN_HUGE - 1) & ~(PIN_HUGE - 1);
    void* raw = mmap(nullptr, n + PIN_HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw != MAP_FAILED) {
      void* p = (void*)(((uintptr_t)raw + PIN_HUGE - 1) & ~(uintptr_t)(PIN_HUGE - 1));
      (void)madvise(p, n, MADV_HUGEPAGE);
      if (hipHostRegister(p, n, hipHostRegisterDefault) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pin_m);
        g_pin.push_back({p, PinnedBlk{raw, n + PIN_HUGE}});
        *out = p;
        return 0;
      }
      (void)hipGetLastError();
      (void)munmap(raw, n + PIN_HUGE);
    }
  }
  return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? 0 : TBK_ENOMEM;
}
void tbk_host_free(void* p) {
  if (!p) return;
  PinnedBlk b{nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(g_pin_m);
    for (size_t i = 0; i < g_pin.size(); ++i)
      if (g_pin[i].first == p) {
        b = g_pin[i].second;
        g_pin[i] = g_pin.back();
        g_pin.pop_back();
        break;
      }
  }
  if (b.raw) {
    (void)hipHostUnregister(p);
    (void)munmap(b.raw, b.raw_bytes);
  } else {
    (void)hipHostFree(p);
  }
}

void tbk_collapse_opts_default(tbk_collapse_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->strategy = TBK_STRAT_CIGAR;
  o->max_nh = INT32_MAX;
  o->min_qual = -1;
}

int tbk_coverage_tile(tbk_ctx* ctx, const tbk_cov_in* in, tbk_cov_out* out) {
  if (!ctx || !in || !out) return TBK_EINVAL;
  if (in->n_records && (!in->tid || !in->pos || !in->cig_off || (in->n_cigar_ops && !in->cig))) return TBK_EINVAL;  // flag may be NULL
  if (out->cap_intervals && (!out->iv_tid || !out->iv_start || !out->iv_end || !out->iv_val)) return TBK_EINVAL;
  if (out->cap_junctions && (!out->j_tid || !out->j_start || !out->j_end || !out->j_strand || !out->j_val)) return TBK_EINVAL;
  if (in->mem != out->mem) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  size_t hint = (size_t)in->n_records * 96 + (size_t)in->n_cigar_ops * 64 + ((size_t)32 << 20);  // (the lean chain's tile tables: 12 MB at least)
  TBK_TRY(ws_begin_call(ctx, hint));
  RegGuard reg_guard{ctx};
  int rc;
  if (in->mem == TBK_MEM_DEVICE) {
    rc = tbk_coverage_device(ctx, in, out);
  } else {
    tbk_cov_in din = *in;
    tbk_cov_out dout = *out;
    din.mem = dout.mem = TBK_MEM_DEVICE;
    size_t n = in->n_records;
    TBK_TRY(h2d(ctx, in->tid, n, &din.tid));
    TBK_TRY(h2d(ctx, in->pos, n, &din.pos));
    TBK_TRY(h2d(ctx, in->flag, n, &din.flag));
    TBK_TRY(h2d(ctx, in->cig_off, n + 1, &din.cig_off));
    TBK_TRY(h2d(ctx, in->cig, (size_t)in->n_cigar_ops, &din.cig));
    TBK_TRY(h2d(ctx, in->yc, n, &din.yc));
    TBK_TRY(h2d(ctx, in->strand, n, &din.strand));
    din.yx = nullptr;
    TBK_TRY(dalloc(ctx, out->iv_tid, out->cap_intervals, &dout.iv_tid));
    TBK_TRY(dalloc(ctx, out->iv_start, out->cap_intervals, &dout.iv_start));
    TBK_TRY(dalloc(ctx, out->iv_end, out->cap_intervals, &dout.iv_end));
    TBK_TRY(dalloc(ctx, out->iv_val, out->cap_intervals, &dout.iv_val));
    TBK_TRY(dalloc(ctx, out->j_tid, out->cap_junctions, &dout.j_tid));
    TBK_TRY(dalloc(ctx, out->j_start, out->cap_junctions, &dout.j_start));
    TBK_TRY(dalloc(ctx, out->j_end, out->cap_junctions, &dout.j_end));
    TBK_TRY(dalloc(ctx, out->j_strand, out->cap_junctions, &dout.j_strand));
    TBK_TRY(dalloc(ctx, out->j_val, out->cap_junctions, &dout.j_val));
    rc = tbk_coverage_device(ctx, &din, &dout);
    out->n_intervals = dout.n_intervals;
    out->n_junctions = dout.n_junctions;
    out->n_bases = dout.n_bases;
    out->span_bases = dout.span_bases;
    if (rc == 0) {
      TBK_TRY(d2h(ctx, out->iv_tid, dout.iv_tid, dout.n_intervals));
      TBK_TRY(d2h(ctx, out->iv_start, dout.iv_start, dout.n_intervals));
      TBK_TRY(d2h(ctx, out->iv_end, dout.iv_end, dout.n_intervals));
      TBK_TRY(d2h(ctx, out->iv_val, dout.iv_val, dout.n_intervals));
      TBK_TRY(d2h(ctx, out->j_tid, dout.j_tid, dout.n_junctions));
      TBK_TRY(d2h(ctx, out->j_start, dout.j_start, dout.n_junctions));
      TBK_TRY(d2h(ctx, out->j_end, dout.j_end, dout.n_junctions));
      TBK_TRY(d2h(ctx, out->j_strand, dout.j_strand, dout.n_junctions));
      TBK_TRY(d2h(ctx, out->j_val, dout.j_val, dout.n_junctions));
      TBK_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  tbk_prof_end_call(ctx);
  return rc;
}

int tbk_collapse_tile(tbk_ctx* ctx, const tbk_collapse_opts* opts, const tbk_soa_in* in, tbk_groups_out* out) {
  if (!ctx || !opts || !in || !out) return TBK_EINVAL;
  if (opts->flags_mask != 0 || opts->keep_unmapped) return TBK_EUNSUPPORTED;
  if (opts->strategy < 0 || opts->strategy > 3) return TBK_EINVAL;
  if (in->n_files == 0 || in->n_files > 65535 || !in->file_off) return TBK_EINVAL;
  if (in->file_off[0] != 0 || in->file_off[in->n_files] != in->n_records) return TBK_EINVAL;
  if (in->n_records &&
      (!in->tid || !in->pos || !in->flag || !in->mapq || !in->strand || !in->nh || !in->cig_off || (in->n_cigar_ops && !in->cig)))
    return TBK_EINVAL;
  // (keep_results: the results may stay on the device — a host caller takes what it needs, when it knows how many groups there are)
  const bool host_tags_optional = opts->keep_results && out->mem == TBK_MEM_HOST;
  if (!host_tags_optional && (!out->rep || !out->yc || !out->yx || !out->yd)) return TBK_EINVAL;
  if (opts->keep_results && opts->defer_yd) return TBK_EINVAL;
  if (opts->strategy == TBK_STRAT_FULL && in->n_records && (!in->md_off || !in->md_has)) return TBK_EINVAL;
  if (opts->collapse_same && (!in->qname_hash || !in->qname_off || (in->n_records && !in->qname))) return TBK_EINVAL;  // -A compares names
  if ((in->prio_hi == nullptr) != (in->prio_lo == nullptr)) return TBK_EINVAL;
  bool any_tb = false;
  if (in->tbmerged)
    for (uint32_t f = 0; f < in->n_files; ++f) any_tb |= in->tbmerged[f] != 0;
  if (in->n_records && any_tb && (!in->yc_in || !in->yx_in || !in->yd_in)) return TBK_EINVAL;
  // a device-resident tile (e.g. tbk_bam_decode) may deliver its groups to host arrays; host input needs host output
  if (in->mem != out->mem && !(in->mem == TBK_MEM_DEVICE && out->mem == TBK_MEM_HOST)) return TBK_EINVAL;
  if (opts->defer_yd && (in->mem != TBK_MEM_DEVICE || out->mem != TBK_MEM_DEVICE)) return TBK_EINVAL;
  TBK_TRY(finish_yd(ctx));  // a still-pending YD stage of the previous tile owns part of the arena
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  // Arena hints.  The sort path keeps keys, sort buffers and per-record scratch (160 B per record); the window path in its raw form —
  // large plain tiles, the predicate of tbk_collapse_device restated — works on the input where it lies and needs half of that
  // (config 3: 25 GB for 320 M records — scratch at the far end of the arena, the group arrays at the bottom — and 9.6 GB for its
  // YD stage, which borrows the range behind the group arrays when it is deferred).  A hint that proves too small costs one call with overflow
  // chunks, after which the arena has learnt its size.
  bool lean = in->mem == TBK_MEM_DEVICE && !opts->store_frac && !opts->collapse_same && !in->prio_hi && ctx->dbg.path == 0 && ctx->dbg.raw < 0 &&
              (in->n_records >= (8u << 20) || (in->n_files > 64 && in->n_records >= 65536));
  if (lean && in->tbmerged)
    for (uint32_t f = 0; f < in->n_files; ++f) lean = lean && in->tbmerged[f] == 0;
  size_t hint = (size_t)in->n_records * (lean ? 84 : 160) + (size_t)in->n_cigar_ops * 8 + ((size_t)8 << 20);
  const auto pa = std::chrono::steady_clock::now();
  TBK_TRY(ws_begin_call(ctx, hint));
  if (ctx->dbg.phases) fprintf(stderr, "collapse phases: arena ready in %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pa).count());
  RegGuard reg_guard{ctx};
  int rc;
  ctx->yd_job = nullptr;
  const size_t yd_hint = (size_t)in->n_records * (lean ? 40 : 96) + ((size_t)8 << 20);
  // (phases=1: where the call's wall time goes; the marks wait for the stream, so the figures are of a serialised call)
  const bool ph = ctx->dbg.phases;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto mark = [&]() {
    if (ph) (void)hipStreamSynchronize(ctx->stream);
    return now();
  };
  const auto p0 = now();
  auto p_in = p0, p_grp = p0, p_yd = p0;
  ctx->kept_n = 0;
  if (in->mem == TBK_MEM_DEVICE && out->mem == TBK_MEM_HOST) {
    tbk_groups_out dout = *out;
    const size_t cap = out->cap_groups, n = in->n_records;
    TBK_TRY(out_dalloc(ctx, out, cap, n, opts->keep_results != 0, &dout));
    p_in = mark();
    rc = tbk_collapse_device(ctx, opts, in, &dout);
    p_grp = mark();
    if (rc == 0 && ctx->yd_job) {
      void* job = ctx->yd_job;
      ctx->yd_job = nullptr;
      rc = tbk_collapse_yd_run(ctx, job);
    }
    p_yd = mark();
    out->n_groups = dout.n_groups;
    out->n_passed = dout.n_passed;
    if (rc == 0 && opts->keep_results) rc = keep_results(ctx, &dout);
    if (rc == 0) rc = out_d2h(ctx, out, &dout, n);
  } else if (in->mem == TBK_MEM_DEVICE) {
    rc = tbk_collapse_device(ctx, opts, in, out);
    if (rc == 0 && ctx->yd_job) {
      void* job = ctx->yd_job;
      ctx->yd_job = nullptr;
      bool defer = opts->defer_yd && ctx->ws_overflow.empty();  // arena must be in steady state to pin a prefix of it
      if (defer && !ctx->yd_ctx && tbk_create(ctx->device, &ctx->yd_ctx) != 0) defer = false;
      if (defer) {
        ctx->yd_ctx->profiling = ctx->profiling;
        ctx->yd_ctx->dbg = ctx->dbg;
        ctx->ws_base_off = (ctx->ws_off + 255) & ~(size_t)255;
        ctx->yd_rc = 0;
        tbk_ctx* side = ctx->yd_ctx;
        // The YD stage works in a range of THIS arena, right behind what it reads (the window path left its scratch at the other
        // end, dead by now): no second arena of 40 bytes per record per context.  The calls that follow on this context start behind
        // the range; when the arena could not hold them as well, the stage keeps an arena of its own.
        const size_t yd_bytes = (yd_hint + 255) & ~(size_t)255;
        const bool borrow = !ctx->dbg.yd_own_arena && ctx->ws_base_off + yd_bytes + ctx->ws_cap / 4 <= ctx->ws_cap;
        if (borrow) {
          if (side->ws && !side->ws_borrowed) (void)hipFree(side->ws);
          side->ws = ctx->ws + ctx->ws_base_off;
          side->ws_cap = yd_bytes;
          side->ws_borrowed = true;
          ctx->ws_base_off += yd_bytes;
        } else if (side->ws_borrowed) {
          side->ws = nullptr;
          side->ws_cap = 0;
          side->ws_borrowed = false;
        }
        if (!ctx->yd_worker) ctx->yd_worker = new TbkWorker();
        ctx->yd_pending = true;
        ctx->yd_worker->post([ctx, side, job, yd_hint]() { ctx->yd_rc = yd_stage_on(side, job, yd_hint); });
      } else {
        rc = tbk_collapse_yd_run(ctx, job);
      }
    }
    if (rc == 0 && opts->keep_results) rc = keep_results(ctx, out);
  } else {
    tbk_soa_in din = *in;
    tbk_groups_out dout = *out;
    din.mem = dout.mem = TBK_MEM_DEVICE;
    size_t n = in->n_records;
    TBK_TRY(h2d(ctx, in->tid, n, &din.tid));
    TBK_TRY(h2d(ctx, in->pos, n, &din.pos));
    TBK_TRY(h2d(ctx, in->flag, n, &din.flag));
    TBK_TRY(h2d(ctx, in->mapq, n, &din.mapq));
    TBK_TRY(h2d(ctx, in->strand, n, &din.strand));
    TBK_TRY(h2d(ctx, in->nh, n, &din.nh));
    TBK_TRY(h2d(ctx, in->cig_off, n + 1, &din.cig_off));
    TBK_TRY(h2d(ctx, in->cig, (size_t)in->n_cigar_ops, &din.cig));
    TBK_TRY(h2d(ctx, in->yc_in, n, &din.yc_in));
    TBK_TRY(h2d(ctx, in->yx_in, n, &din.yx_in));
    TBK_TRY(h2d(ctx, in->yd_in, n, &din.yd_in));
    TBK_TRY(h2d(ctx, in->md_off, in->md_off ? n + 1 : 0, &din.md_off));
    TBK_TRY(h2d(ctx, in->md, in->md_off ? (size_t)in->md_off[n] : 0, &din.md));
    TBK_TRY(h2d(ctx, in->md_has, n, &din.md_has));
    TBK_TRY(h2d(ctx, in->qname_hash, n, &din.qname_hash));
    TBK_TRY(h2d(ctx, in->qname_off, in->qname_off ? n + 1 : 0, &din.qname_off));
    TBK_TRY(h2d(ctx, in->qname, in->qname_off ? (size_t)in->qname_off[n] : 0, &din.qname));
    TBK_TRY(h2d(ctx, in->prio_hi, n, &din.prio_hi));
    TBK_TRY(h2d(ctx, in->prio_lo, n, &din.prio_lo));
    size_t cap = out->cap_groups;
    TBK_TRY(out_dalloc(ctx, out, cap, n, opts->keep_results != 0, &dout));
    p_in = mark();
    rc = tbk_collapse_device(ctx, opts, &din, &dout);
    p_grp = mark();
    if (rc == 0 && ctx->yd_job) {
      void* job = ctx->yd_job;
      ctx->yd_job = nullptr;
      rc = tbk_collapse_yd_run(ctx, job);
    }
    p_yd = mark();
    out->n_groups = dout.n_groups;
    out->n_passed = dout.n_passed;
    if (rc == 0 && opts->keep_results) rc = keep_results(ctx, &dout);
    if (rc == 0) rc = out_d2h(ctx, out, &dout, n);
  }
  if (ph && out->mem == TBK_MEM_HOST)
    fprintf(stderr, "collapse phases ms: inputs %.1f | grouping %.1f | YD stage %.1f | results %.1f | %u groups%s\n", ms(p0, p_in), ms(p_in, p_grp), ms(p_grp, p_yd),
            ms(p_yd, now()), out->n_groups, opts->keep_results ? " (kept on the device)" : "");
  tbk_prof_end_call(ctx);
  return rc;
}

int tbk_warmup(tbk_ctx* ctx) {
  if (!ctx) return TBK_EINVAL;
  TBK_TRY(finish_yd(ctx));
  TBK_TRY(tbk_collapse_warm(ctx));
  (void)bounce_ready(ctx);
  // the copy engines: the process's first large copy in a direction sets that engine up inside the call (measured: 10-20 ms in the
  // hipMemcpyAsync that brings a first tile's results back; small copies go through a kernel and do not count)
  constexpr size_t kWarm = (size_t)4 << 20;
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, kWarm) == hipSuccess && hipMalloc(&d, kWarm) == hipSuccess) {
    (void)hipMemcpyAsync(d, h, kWarm, hipMemcpyHostToDevice, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    // (the way back on the OTHER stream: a stream sends a copy to the engine its last copy used, and the engine the device -> host
    // direction gets when the runtime chooses afresh is another one)
    hipStream_t back = tbk_aux_stream(ctx) ? tbk_aux_stream(ctx) : ctx->stream;
    (void)hipMemcpyAsync(h, d, kWarm, hipMemcpyDeviceToHost, back);
    (void)hipStreamSynchronize(back);
  }
  if (d) (void)hipFree(d);
  if (h) (void)hipHostFree(h);
  (void)hipGetLastError();
  return 0;
}

int tbk_kept_results(tbk_ctx* ctx, uint32_t first, uint32_t n, uint32_t* rep, double* yc, int64_t* yx, int32_t* yd) {
  if (!ctx) return TBK_EINVAL;
  if (!ctx->kept || (uint64_t)first + n > ctx->kept_n) return TBK_EINVAL;
  if (n == 0) return 0;
  TBK_HIP(hipSetDevice(ctx->device));
  RegGuard reg_guard{ctx};
  TBK_TRY(d2h(ctx, rep, ctx->kept_rep + first, n));
  TBK_TRY(d2h(ctx, yc, ctx->kept_yc + first, n));
  TBK_TRY(d2h(ctx, yx, ctx->kept_yx + first, n));
  TBK_TRY(d2h(ctx, yd, ctx->kept_yd + first, n));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int tbk_sample_tile(tbk_ctx* ctx, const tbk_cov_in* in, int32_t num_samples, tbk_sample_out* out) {
  if (!ctx || !in || !out || num_samples <= 0) return TBK_EINVAL;
  if (in->mem != out->mem) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  TBK_TRY(ws_begin_call(ctx, (size_t)in->n_records * 96 + ((size_t)8 << 20)));
  RegGuard reg_guard{ctx};
  int rc;
  if (in->mem == TBK_MEM_DEVICE) {
    rc = tbk_sample_device(ctx, in, num_samples, out);
  } else {
    tbk_cov_in din = *in;
    tbk_sample_out dout = *out;
    din.mem = dout.mem = TBK_MEM_DEVICE;
    size_t n = in->n_records;
    TBK_TRY(h2d(ctx, in->tid, n, &din.tid));
    TBK_TRY(h2d(ctx, in->pos, n, &din.pos));
    TBK_TRY(h2d(ctx, in->flag, n, &din.flag));
    TBK_TRY(h2d(ctx, in->cig_off, n + 1, &din.cig_off));
    TBK_TRY(h2d(ctx, in->cig, (size_t)in->n_cigar_ops, &din.cig));
    TBK_TRY(h2d(ctx, in->yx, n, &din.yx));
    din.yc = nullptr;
    din.strand = nullptr;
    TBK_TRY(dalloc(ctx, out->iv_tid, out->cap_intervals, &dout.iv_tid));
    TBK_TRY(dalloc(ctx, out->iv_start, out->cap_intervals, &dout.iv_start));
    TBK_TRY(dalloc(ctx, out->iv_end, out->cap_intervals, &dout.iv_end));
    TBK_TRY(dalloc(ctx, out->iv_count, out->cap_intervals, &dout.iv_count));
    TBK_TRY(dalloc(ctx, out->iv_heat, out->cap_intervals, &dout.iv_heat));
    rc = tbk_sample_device(ctx, &din, num_samples, &dout);
    out->n_intervals = dout.n_intervals;
    if (rc == 0) {
      size_t g = dout.n_intervals;
      TBK_TRY(d2h(ctx, out->iv_tid, dout.iv_tid, g));
      TBK_TRY(d2h(ctx, out->iv_start, dout.iv_start, g));
      TBK_TRY(d2h(ctx, out->iv_end, dout.iv_end, g));
      TBK_TRY(d2h(ctx, out->iv_count, dout.iv_count, g));
      TBK_TRY(d2h(ctx, out->iv_heat, dout.iv_heat, g));
      TBK_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  tbk_prof_end_call(ctx);
  return rc;
}

}  // extern "C"
