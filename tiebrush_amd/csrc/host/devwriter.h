// devwriter.h — the output side of the tiebrush command line on the device: flushPData's tagging (/root/reference/src/tiebrush.cpp:
// 506-525) and GSamWriter::write -> sam_write1 (GSam.h:648-653) of the collapsed groups, through tbk_bam_encode (include/tbk.h).
//
// The groups go out in chunks through three stages that run side by side:
//   gather   the representatives the HOST decoded are copied — by every core — into a pinned staging buffer (records the device decoded
//            are read where they lie: nothing to gather);
//   encode   one tbk_bam_encode per chunk: H2D of the staged records and the chunk's YC / YX / YD, tags, member cuts, deflate, D2H of the
//            finished BGZF members into a pinned buffer (a thread of its own: the context is busy for the length of the call; with a
//            second context — set_second — two threads, the even chunks on one and the odd ones on the other: a chunk's copies run
//            under the other chunk's deflate, tbk_enc_in.from);
//   write    the members appended to the output file (a thread of its own: the write of chunk k runs beside the encode of k + 1).
// Three slots of buffers, so every stage works on a chunk of its own.  The record stream is the host writer's byte for byte
// (tests/test_gpu_cli.py); the members differ (another deflate parse, members cut at record boundaries).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "GSam.h"
#include "bam.h"
#include "tbk_dl.h"

class DeviceWriter {
 public:
  static constexpr uint32_t kChunkGroups = 256u << 10;  // ~ 64 MB of records with SEQ / QUAL per chunk (512 K: the encode stage 10 % faster, the pinned staging twice as dear — no gain)
  static constexpr int kSlots = 3;  // (a fourth — gather, two encodes, write: a chunk each — measured no faster than three)

  DeviceWriter(TbkApi& api, int nthreads) : api_(api), nt_(nthreads < 1 ? 1 : nthreads) {
    if (const char* e = getenv("TBK_DW_SLOTS")) nslots_ = std::max(2, std::min((int)kSlots, atoi(e)));
    if (const char* e = getenv("TBK_DW_CHUNK_GROUPS")) {  // test hook: small chunks, so that a small input has a "later chunk"
      const long v = atol(e);
      if (v >= 64 && v <= (long)kChunkGroups) chunk_ = (uint32_t)v, chunk_forced_ = true;
    }
  }
  ~DeviceWriter() {
    for (auto& s : slot_) {
      if (s.blob) api_.host_free(s.blob);
      if (s.z) api_.host_free(s.z);
    }
  }
  // a second context (same device) for a second encode thread; nullptr: one encoder
  void set_second(tbk_ctx* c) { ctx2_ = c; }
  // pinned staging, sized for a chunk of typical records; grown when a chunk needs more.  Page-locking hundreds of megabytes takes tens
  // of milliseconds: the command line calls this on its helper thread while the inputs are decoded.
  bool reserve(size_t blob_bytes = (size_t)kChunkGroups * 272, size_t z_bytes = (size_t)kChunkGroups * 112) {
    for (int i = 0; i < nslots_; ++i) {
      Slot& s = slot_[i];
      if (!grow(s.blob, s.blob_cap, blob_bytes) || !grow(s.z, s.z_cap, z_bytes)) return false;
    }
    return true;
  }

  // Groups [0, ng) of a collapse (rep / yc / yx / yd: HOST arrays in output order; yc == nullptr: the collapse ran with keep_results and the
  // tags' values are read on the device, where the context kept them — `rep` is still the host's copy).  Representatives with rep < n_dev are records of
  // the tile tbk_bam_decode left on `ctx`; host_record(g) hands out the others.  Returns false when the device cannot take the
  // output — TBK_EUNSUPPORTED (a record too long for a BGZF member of its own: bgzdef.hip) or TBK_ENOMEM / no pinned memory, on ANY
  // chunk: the chunks before the refused one are written in full and *groups_done says how many groups that was, so the caller's host
  // writer takes the groups from there on (a long read far into the output does not cost the run).  Any other failure is fatal
  // (GError).  *payload / *zbytes: bytes of tagged records / of BGZF members written.
  bool write(tbk_ctx* ctx, GSamWriter& out, uint32_t ng, const uint32_t* rep, const double* yc, const int64_t* yx, const int32_t* yd, uint32_t n_dev,
             const std::function<tbh::RecView(uint32_t)>& host_record, uint64_t* payload, uint64_t* zbytes, std::string& why, uint32_t* groups_done,
             const std::function<void(uint32_t, int)>& host_prefetch = nullptr) {
    *payload = *zbytes = 0;
    *groups_done = 0;
    if (ng == 0) return true;
    if (!reserve()) {
      why = "pinned staging memory";
      return false;
    }
    // The encoder deflates 512 members at a time (bgz_deflate_k: two workgroups a CU, 256 CUs, each taking the next member as it comes
    // free): a chunk of 1047 members — 256 K records of ~ 260 bytes — runs THREE rounds for what two nearly hold (4.1 ms instead of 2.8
    // every chunk).  The chunk is cut a little below a whole number of rounds, by the records' mean length in a sample of the groups
    // (the host's representatives; the device's look like them); the slots' size is the cap, the test hook's size is taken as it is.
    uint32_t kChunkGroups = chunk_;  // (shadows the constant)
    if (!chunk_forced_ && ng > 65536) {
      uint64_t bytes = 0, cnt = 0;
      uint32_t longest = 0;
      const uint32_t step = std::max<uint32_t>(1, ng / 2048);
      for (uint32_t g = 0; g < ng; g += step)
        if (rep[g] >= n_dev) {
          const uint32_t l = host_record(g).len;
          bytes += l, ++cnt, longest = std::max(longest, l);
        }
      if (cnt >= 64 && longest + 1024 < 0xff00u) {
        const double tagged = (double)bytes / (double)cnt + 4 + 16;              // block_size, YC:f + YX (+ YD)
        const double member = (double)(0xff00u - longest - 256);                 // what a member holds (bgzdef.hip: B)
        for (int rounds = 4; rounds >= 1; --rounds) {                              // the most rounds the slots hold
          const double groups = 0.97 * 512.0 * rounds * member / tagged;
          if (groups <= (double)chunk_) {
            kChunkGroups = std::max<uint32_t>(65536, (uint32_t)groups & ~1023u);
            break;
          }
        }
      }
    }
    const uint32_t nchunk = (ng + kChunkGroups - 1) / kChunkGroups;
    Pool pool;
    if (nt_ > 1 && ng >= 8192) {
      pool.start(nt_ - 1);
      pool_ = &pool;
    }
    struct PoolOff {
      Pool** p;
      ~PoolOff() { *p = nullptr; }
    } pool_off{&pool_};
    // stage hand-offs: state[k] counts how far chunk k has come (1 gathered, 2 encoded, 3 written)
    std::mutex m;
    std::condition_variable cv;
    std::vector<int> state(nchunk, 0);
    // a failure names the chunk it stopped at: every stage still finishes the chunks before it (they are written), none touches that
    // chunk or a later one.  fail / fail_at / fail_msg change under `m` only (a waiter checks them under `m`: no lost wake-up).
    int fail = 0;  // 1: chunk fail_at refused (the host writer takes over from there), 2: fatal
    uint32_t fail_at = 0xFFFFFFFFu;
    std::string fail_msg;
    auto set_state = [&](uint32_t k, int v) {
      {
        std::lock_guard<std::mutex> lk(m);
        state[k] = v;
      }
      cv.notify_all();
    };
    auto set_fail = [&](uint32_t k, int code, const std::string& msg) {
      {
        std::lock_guard<std::mutex> lk(m);
        if (fail == 0 || k < fail_at) fail = code, fail_at = k, fail_msg = msg;
      }
      cv.notify_all();
    };
    auto wait_state = [&](uint32_t k, int v) {  // false: chunk k will never get there
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return state[k] >= v || (fail != 0 && k >= fail_at); });
      return state[k] >= v && !(fail != 0 && k >= fail_at);
    };
    auto failed = [&]() {
      std::lock_guard<std::mutex> lk(m);
      return fail != 0;
    };
    const char* force_refuse = getenv("TBK_TEST_DW_REFUSE_CHUNK");  // test hook: chunk k answers TBK_EUNSUPPORTED
    std::vector<uint64_t> zsz(nchunk, 0), psz(nchunk, 0);
    std::vector<uint32_t> nhost(nchunk, 0);
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    double busy_gather = 0, busy_encode = 0, busy_write = 0, busy_pass1 = 0;  // what each stage spent working (TBK_TIMING: which one paces the pipeline)
    const uint32_t n_enc = ctx2_ && nchunk > 1 && !getenv("TBK_DW_ONE_ENCODER") ? 2u : 1u;
    std::mutex busy_m;
    auto encoder = [&](uint32_t e) {
      tbk_ctx* const ectx = e == 0 ? ctx : ctx2_;
      for (uint32_t k = e; k < nchunk; k += n_enc) {
        if (!wait_state(k, 1)) return;
        Slot& s = slot_[k % (uint32_t)nslots_];
        const uint32_t g0 = k * kChunkGroups, g1 = std::min(ng, g0 + kChunkGroups);
        tbk_enc_in in;
        memset(&in, 0, sizeof(in));
        in.n = g1 - g0;
        if (yc) {
          in.mem = TBK_MEM_HOST;
          in.rep = rep + g0, in.yc = yc + g0, in.yx = yx + g0, in.yd = yd + g0;
        } else {
          in.mem = TBK_MEM_KEPT;
          in.first = g0;
        }
        if (ectx != ctx) in.from = ctx;  // (the kept results and the decoded tile live on the first context)
        in.n_dev = n_dev;
        in.n_host = nhost[k];
        in.host_blob = s.blob, in.host_off = s.off.data(), in.host_slot = s.slot.data();
        uint64_t zb = 0, pb = 0;
        const auto e0 = tnow();
        int rc = (force_refuse && (uint32_t)atol(force_refuse) == k) ? TBK_EUNSUPPORTED : api_.bam_encode(ectx, &in, s.z, s.z_cap, &zb, &pb);
        if (rc == TBK_E2BIG && zb > s.z_cap) {  // (the members of this chunk need a larger buffer: the call said how large)
          if (!grow(s.z, s.z_cap, zb + zb / 8)) rc = TBK_ENOMEM;
          else rc = api_.bam_encode(ectx, &in, s.z, s.z_cap, &zb, &pb);
        }
        if (rc != 0) {
          set_fail(k, (rc == TBK_EUNSUPPORTED || rc == TBK_ENOMEM) ? 1 : 2, std::string(api_.strerror_(rc)) + " (" + api_.last_error(ectx) + ")");
          return;
        }
        zsz[k] = zb, psz[k] = pb;
        {
          std::lock_guard<std::mutex> lk(busy_m);
          busy_encode += tms(e0, tnow());
        }
        set_state(k, 2);
      }
    };
    std::thread enc([&]() { encoder(0); });
    std::thread enc2;
    if (n_enc > 1) enc2 = std::thread([&]() { encoder(1); });
    std::thread wr([&]() {
      for (uint32_t k = 0; k < nchunk; ++k) {
        if (!wait_state(k, 2)) return;
        Slot& s = slot_[k % (uint32_t)nslots_];
        const auto w0 = tnow();
        out.write_members(s.z, (size_t)zsz[k]);
        busy_write += tms(w0, tnow());
        set_state(k, 3);
      }
    });
    // the gather, on this thread and its workers
    for (uint32_t k = 0; k < nchunk && !failed(); ++k) {
      if (k >= (uint32_t)nslots_ && !wait_state(k - (uint32_t)nslots_, 3)) break;  // the slot's previous chunk has left the building
      Slot& s = slot_[k % (uint32_t)nslots_];
      const uint32_t g0 = k * kChunkGroups, g1 = std::min(ng, g0 + kChunkGroups), nc = g1 - g0;
      const auto g_0 = tnow();
      s.slot.resize(nc);
      // pass 1: which groups bring a host record, and how long — per slice of the chunk
      const int T = nc < 8192 ? 1 : nt_;
      std::vector<uint64_t> sl_bytes((size_t)T + 1, 0);
      std::vector<uint32_t> sl_cnt((size_t)T + 1, 0);
      auto slice = [&](int t, uint32_t* a, uint32_t* b) {
        *a = g0 + (uint32_t)((uint64_t)nc * (uint32_t)t / (uint32_t)T);
        *b = g0 + (uint32_t)((uint64_t)nc * ((uint32_t)t + 1) / (uint32_t)T);
      };
      // (the records lie all over the inflated inputs: host_prefetch(g, 0 / 1) asks for g's index entry / first line a few groups ahead,
      // and the second pass copies from what the first one found instead of looking every record up again)
      s.view.resize(nc);
      parallel(T, [&](int t) {
        uint32_t a, b;
        slice(t, &a, &b);
        uint64_t by = 0;
        uint32_t cn = 0;
        constexpr uint32_t kAhead0 = 24, kAhead1 = 10;
        for (uint32_t g = a; g < b; ++g) {
          if (host_prefetch) {
            if (g + kAhead0 < b && rep[g + kAhead0] >= n_dev) host_prefetch(g + kAhead0, 0);
            if (g + kAhead1 < b && rep[g + kAhead1] >= n_dev) host_prefetch(g + kAhead1, 1);
          }
          if (rep[g] >= n_dev) {
            const tbh::RecView v = host_record(g);
            s.view[g - g0] = v;
            by += 4 + (uint64_t)v.len, ++cn;
          }
        }
        sl_bytes[(size_t)t + 1] = by, sl_cnt[(size_t)t + 1] = cn;
      });
      busy_pass1 += tms(g_0, tnow());
      for (int t = 0; t < T; ++t) sl_bytes[(size_t)t + 1] += sl_bytes[(size_t)t], sl_cnt[(size_t)t + 1] += sl_cnt[(size_t)t];
      const uint64_t total = sl_bytes[(size_t)T];
      nhost[k] = sl_cnt[(size_t)T];
      s.off.resize((size_t)nhost[k] + 1);
      if (total + 16 > s.blob_cap && !grow(s.blob, s.blob_cap, total + total / 8 + 16)) {
        set_fail(k, 1, "pinned staging memory");
        break;
      }
      // pass 2: the copies
      parallel(T, [&](int t) {
        uint32_t a, b;
        slice(t, &a, &b);
        uint64_t o = sl_bytes[(size_t)t];
        uint32_t c = sl_cnt[(size_t)t];
        for (uint32_t g = a; g < b; ++g) {
          if (rep[g] < n_dev) {
            s.slot[g - g0] = 0;
            continue;
          }
          const tbh::RecView v = s.view[g - g0];
          if (g + 6 < b) __builtin_prefetch(s.view[g + 6 - g0].p);
          s.slot[g - g0] = c;
          s.off[c++] = o;
          memcpy(s.blob + o, &v.len, 4);
          memcpy(s.blob + o + 4, v.p, v.len);
          o += 4 + (uint64_t)v.len;
        }
      });
      s.off[nhost[k]] = total;
      busy_gather += tms(g_0, tnow());
      set_state(k, 1);
    }
    enc.join();
    if (enc2.joinable()) enc2.join();
    wr.join();
    if (getenv("TBK_TIMING"))
      fprintf(stderr, "device writer stages busy ms: gather %.1f (its first pass %.1f) | encode %.1f on %u thread%s | write %.1f (%u chunks of %u groups)\n", busy_gather, busy_pass1,
              busy_encode, n_enc, n_enc > 1 ? "s" : "", busy_write, nchunk, kChunkGroups);
    if (fail == 2) GError("Error: encoding the output on the GPU failed: %s\n", fail_msg.c_str());
    const uint32_t kdone = fail ? fail_at : nchunk;  // (the threads are gone: plain reads)
    for (uint32_t k = 0; k < kdone; ++k) *payload += psz[k], *zbytes += zsz[k];
    *groups_done = fail ? std::min<uint64_t>((uint64_t)kdone * kChunkGroups, ng) : ng;
    if (fail == 1) {
      why = fail_msg;
      return false;
    }
    return true;
  }

 private:
  struct Slot {
    uint8_t* blob = nullptr;
    size_t blob_cap = 0;
    uint8_t* z = nullptr;
    size_t z_cap = 0;
    std::vector<uint64_t> off;
    std::vector<uint32_t> slot;
    std::vector<tbh::RecView> view;  // the chunk's host records as the gather's first pass found them
  };
  bool grow(uint8_t*& p, size_t& cap, size_t want) {
    if (want <= cap) return true;
    if (p) api_.host_free(p);
    p = nullptr, cap = 0;
    void* q = nullptr;
    if (api_.host_alloc(want, &q) != 0) return false;
    p = (uint8_t*)q, cap = want;
    return true;
  }
  // The gather's workers live as long as the writer (two passes a chunk, tens of chunks: starting fifteen threads for each was 10-20 ms
  // of a 50 ms stage).  parallel(T, f) runs f(0) .. f(T - 1), f(0) on the caller; T <= the pool's size + 1.
  struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> job;
    uint64_t gen = 0;
    int want = 0, left = 0;
    bool stop = false;
    void start(int n) {
      for (int i = 0; i < n; ++i)
        th.emplace_back([this, i]() {
          uint64_t seen = 0;
          for (;;) {
            std::function<void(int)> f;
            {
              std::unique_lock<std::mutex> lk(m);
              cv_go.wait(lk, [&] { return stop || gen != seen; });
              if (stop) return;
              seen = gen;
              if (i + 1 >= want) continue;  // (this round needs fewer workers)
              f = job;
            }
            f(i + 1);
            {
              std::lock_guard<std::mutex> lk(m);
              if (--left == 0) cv_done.notify_one();
            }
          }
        });
    }
    void run(int T, const std::function<void(int)>& f) {
      {
        std::lock_guard<std::mutex> lk(m);
        job = f, want = T, left = T - 1, ++gen;
      }
      cv_go.notify_all();
      f(0);
      std::unique_lock<std::mutex> lk(m);
      cv_done.wait(lk, [&] { return left == 0; });
    }
    ~Pool() {
      {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
      }
      cv_go.notify_all();
      for (auto& x : th) x.join();
    }
  };
  Pool* pool_ = nullptr;
  template <class F>
  void parallel(int T, F f) {
    if (T <= 1 || !pool_) {
      if (T <= 1) {
        f(0);
        return;
      }
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back([&f, t]() { f(t); });
      f(0);
      for (auto& x : th) x.join();
      return;
    }
    pool_->run(T, f);
  }
  TbkApi& api_;
  tbk_ctx* ctx2_ = nullptr;
  int nt_;
  uint32_t chunk_ = kChunkGroups;
  bool chunk_forced_ = false;
  int nslots_ = kSlots;
  Slot slot_[kSlots];
};
