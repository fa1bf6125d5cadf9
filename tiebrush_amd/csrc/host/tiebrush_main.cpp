// tiebrush — drop-in command line of the reference's collapse tool (/root/reference/src/tiebrush.cpp:557-676),
// with the per-record main loop (:570-592) replaced by one call into the MI355X hot path:
//   decode every input into a SoA tile (host: BGZF inflate + record / aux scan on every core, fastload.cpp; inputs larger than
//   memory stream through TInputFiles::next_tile)  ->  tbk_collapse_tile (HIP)  ->  tag + deflate + write (host).
// libtbk.so (and with it the HIP runtime) is bound with dlopen on a helper thread while the inputs are read (tbk_dl.h).
// There is no CPU implementation of the collapse in this binary: without a usable GPU it exits with an error.
#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <memory>
#include <signal.h>
#include <spawn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <chrono>
#include <time.h>
#include <thread>
#include <vector>

#include "../../../include/tbk.h"
#include "GSam.h"
#include "args.h"
#include "bgzf.h"
#include "devwriter.h"
#include "fastload.h"
#include "tagwrite.h"
#include "tbk_dl.h"
#include "tmerge.h"

extern char** environ;

#define VERSION "0.0.7"

static const char* USAGE =
    "TieBrush v" VERSION " (MI355X build)\n"
    "Collapses identical alignments from several coordinate-sorted BAM files into one BAM.\n"
    "Every output alignment carries: YC (how many alignments it stands for), YX (in how many\n"
    "samples it was seen) and YD (upstream extent of its coverage island, omitted when 0).\n"
    "\n"
    " usage: tiebrush [options] -o OUT.bam IN1.bam [IN2.bam ...]   (or one text file listing the inputs)\n"
    "\n"
    "  -h,--help            print this text and exit\n"
    "  --version            print the version and exit\n"
    "  -o FILE              output BAM (required)\n"
    "  -L,--full            group by CIGAR and MD\n"
    "  -P,--clip            group by CIGAR after removing soft clips\n"
    "  -E,--exon            group by exon coordinates\n"
    "                       (default: group by CIGAR; the three are mutually exclusive)\n"
    "  -S,--keep-supp       keep supplementary alignments\n"
    "  --keep-secondary     keep secondary alignments\n"
    "  -M,--keep-unmap      keep unmapped reads (not available in the GPU build)\n"
    "  -N INT               drop alignments with NH above INT\n"
    "  -Q INT               drop alignments with mapping quality below INT\n"
    "  -F INT               flag bits that must agree (not available in the GPU build)\n"
    "  -A,--collapse-same   do not count the same read of the same sample twice\n"
    "  --store-frac         YC adds 1/NH per alignment (needs --keep-secondary)\n"
    "  -V,--verbose         echo the command line\n"
    "  --ranks N            shard the input files over N GPUs of this node (one process each), one output BAM\n"
    "  --writer WHICH       device (default): the output records are tagged and BGZF-compressed on the GPU; host: by the CPU cores\n";

// a buffer that is allocated, not initialised (untouched pages cost nothing), on huge pages when it is large, and not freed at
// the end: these buffers live as long as the process, which ends with _exit — returning gigabytes page by page first only
// delays that.  The contents do not survive a resize (every caller fills the buffer afresh), so a buffer that grows gives its
// old block back first and grows by half at least: the streaming path resizes per tile, and tiles come in any order of size.
template <class T>
struct RawBuf {
  T* p = nullptr;
  size_t cap = 0, len = 0;
  bool borrowed = false;
  // another buffer's pages for a while (a dead array of the input tile: resident already, nothing to fault in)
  void borrow(T* q, size_t n) {
    if (!borrowed) tbh::big_free(p);
    p = q, cap = len = n, borrowed = true;
  }
  void resize(size_t n) {
    if (borrowed) p = nullptr, cap = 0, borrowed = false;
    if (n > cap) {
      const size_t want = cap ? std::max(n, cap + cap / 2) : n;
      tbh::big_free(p);
      p = (T*)tbh::big_alloc(want * sizeof(T));
      if (!p) GError("Error: out of memory\n");
      cap = want;
    }
    len = n;
  }
  T* data() { return p; }
  size_t size() const { return len; }
  T& operator[](size_t i) { return p[i]; }
};

// an input file as the page cache holds it: the device decode uploads it through its own pinned ring (tbk_bam_decode), so a private copy
// made with read() would only be one more pass over gigabytes
struct FileMap {
  const uint8_t* p = nullptr;
  size_t n = 0;
  bool map(const std::string& path) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) {
      close(fd);
      return false;
    }
    n = (size_t)st.st_size;
    if (n) {
      void* q = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (q == MAP_FAILED) {
        close(fd);
        n = 0;
        return false;
      }
      (void)madvise(q, n, MADV_SEQUENTIAL);
      p = (const uint8_t*)q;
    }
    close(fd);
    return true;
  }
  void unmap() {
    if (p) munmap(const_cast<uint8_t*>(p), n);
    p = nullptr, n = 0;
  }
};

// `tiebrush --ranks N ...`: the multi-GPU form (tiebrush_amd/ranks.py: one process per GPU over torch.distributed / RCCL, the input
// files sharded by rank, one output BAM).  The launcher, which starts the ranks, runs as a child of this process.
static volatile pid_t g_ranks_child = 0;
static void forward_to_ranks_child(int sig) {
  if (g_ranks_child > 0) kill(g_ranks_child, sig);
}
static void maybe_exec_ranks(int argc, char* argv[]) {
  bool want = false;
  for (int i = 1; i < argc; ++i) want = want || strcmp(argv[i], "--ranks") == 0 || strncmp(argv[i], "--ranks=", 8) == 0;
  if (!want) return;
  char exe[PATH_MAX];
  const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
  if (n <= 0) GError("Error: --ranks: cannot locate the installation\n");
  exe[n] = 0;
  std::string root(exe);  // <root>/tiebrush_amd/_build/tiebrush
  for (int up = 0; up < 3; ++up) {
    const size_t sl = root.rfind('/');
    if (sl == std::string::npos) GError("Error: --ranks: cannot locate the installation\n");
    root.resize(sl);
  }
  std::string pp = root;
  if (const char* old = getenv("PYTHONPATH")) pp += std::string(":") + old;
  setenv("PYTHONPATH", pp.c_str(), 1);
  std::vector<char*> av;
  const char* py = getenv("TBK_PYTHON") ? getenv("TBK_PYTHON") : "python3";
  av.push_back(const_cast<char*>(py));
  av.push_back(const_cast<char*>("-m"));
  av.push_back(const_cast<char*>("tiebrush_amd.ranks"));
  for (int i = 1; i < argc; ++i) av.push_back(argv[i]);
  av.push_back(nullptr);
  // a CHILD, never an exec of this process: under a profiler or any launcher that preloads a GPU-initialising library this process may
  // already hold the GPU, and replacing such a process is what takes a node down.  The launcher runs as a fresh process; this one
  // waits, forwards the signals a caller's timeout would send and leaves with the child's status.
  pid_t child = 0;
  const int rc = posix_spawnp(&child, py, nullptr, nullptr, av.data(), environ);
  if (rc != 0) GError("Error: --ranks: cannot start %s (%s)\n", py, strerror(rc));
  g_ranks_child = child;
  signal(SIGTERM, forward_to_ranks_child);
  signal(SIGINT, forward_to_ranks_child);
  signal(SIGHUP, forward_to_ranks_child);
  int status = 0;
  while (waitpid(child, &status, 0) < 0)
    if (errno != EINTR) GError("Error: --ranks: waiting for the launcher failed (%s)\n", strerror(errno));
  fflush(stdout);
  fflush(stderr);
  _exit(WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0));
}

int main(int argc, char* argv[]) {
  maybe_exec_ranks(argc, argv);
  TInputFiles inRecords;
  inRecords.setup(VERSION, argc, argv);
  Args args(argc, argv, "help;debug;verbose;version;full;clip;exon;keep-supp;keep-secondary;keep-unmap;collapse-same;store-frac;writer=;SMLPEDVho:N:Q:F:A");
  if (!args.error().empty()) {
    GMessage("%s\n%s\n", USAGE, args.error().c_str());
    return 1;
  }
  if (args.getOpt('h') || args.getOpt("help")) {
    fprintf(stdout, "%s", USAGE);
    return 0;
  }
  if (args.getOpt("version")) {
    fprintf(stdout, "%s\n", VERSION);
    return 0;
  }
  if (args.startNonOpt() == 0) {
    GMessage("%s", USAGE);
    GMessage("\nError: no input provided!\n");
    return 1;
  }
  const char* outfname = args.getOpt('o');
  if (!outfname || !*outfname) {
    GMessage("%s", USAGE);
    GMessage("\nError: output filename must be provided (-o)!\n");
    return 1;
  }
  tbk_collapse_opts opt;  // (tbk_collapse_opts_default: the library is not bound yet)
  memset(&opt, 0, sizeof(opt));
  opt.strategy = TBK_STRAT_CIGAR;
  opt.max_nh = INT32_MAX;
  opt.min_qual = -1;
  if (const char* s = args.getOpt('N')) opt.max_nh = atoi(s);
  if (const char* s = args.getOpt('Q')) opt.min_qual = atoi(s);
  if (const char* s = args.getOpt('F')) opt.flags_mask = (uint32_t)atoi(s);
  opt.keep_supplementary = (args.getOpt("keep-supp") || args.getOpt('S')) ? 1 : 0;
  opt.keep_secondary = args.getOpt("keep-secondary") ? 1 : 0;
  opt.keep_unmapped = (args.getOpt("keep-unmap") || args.getOpt('M')) ? 1 : 0;
  opt.collapse_same = (args.getOpt("collapse-same") || args.getOpt('A')) ? 1 : 0;
  opt.store_frac = args.getOpt("store-frac") ? 1 : 0;
  if (opt.store_frac && !opt.keep_secondary) GError("Error: --store-frac requires --keep-secondary to be enabled.\n");
  bool stratF = args.getOpt("full") || args.getOpt('L');
  bool stratP = args.getOpt("clip") || args.getOpt('P');
  bool stratE = args.getOpt("exon") || args.getOpt('E');
  if (stratF | stratP | stratE) {
    if (!(stratF ^ stratP ^ stratE)) GError("Error: only one merging strategy can be requested.\n");
    opt.strategy = stratF ? TBK_STRAT_FULL : (stratP ? TBK_STRAT_CLIP : TBK_STRAT_EXON);
  }
  if (args.getOpt("verbose") || args.getOpt('V')) {
    fprintf(stderr, "Running TieBrush " VERSION ". Command line:\n");
    args.printCmdLine(stderr);
  }
  bool dev_writer = true;
  if (const char* w = args.getOpt("writer")) {
    if (strcmp(w, "host") == 0) dev_writer = false;
    else if (strcmp(w, "device") != 0) GError("Error: --writer takes host or device\n");
  }
  if (opt.flags_mask != 0) GError("Error: -F is not supported by the GPU build (its reference semantics are unpinned)\n");
  if (opt.keep_unmapped) GError("Error: -M/--keep-unmap is not supported by the GPU build\n");
  while (const char* ifn = args.nextNonOpt()) inRecords.addFile(tbh_realpath(ifn).c_str());

  const bool timing = getenv("TBK_TIMING") != nullptr;
  // (TBK_TIMING=2: the calls' kernels too, through the library's HIP events — two events per launch are themselves milliseconds of a
  // run this short, so the phase lines of TBK_TIMING=1, which the bench's end-to-end legs run under, come without them)
  const bool ktiming = timing && atoi(getenv("TBK_TIMING")) > 1;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  auto t_start = tnow();
  // binding libtbk.so (the HIP runtime comes with it) and bringing the device up take ~0.3 s: both happen on a helper thread
  // while the inputs are opened, inflated and parsed
  TbkApi api;
  tbk_ctx* ctx = nullptr;
  int dev = getenv("TBK_DEVICE") ? atoi(getenv("TBK_DEVICE")) : 0;
  int rc = 0;
  bool api_ok = false;
  DeviceWriter* dw = nullptr;  // (pinned staging: lives until the process ends)
  double ms_ctx_ready = 0;
  // The thread brings the device up and then warms the context (below).  A caller that can use the context the moment it exists — the
  // hybrid path's device decode — says so (ctx_urgent) and gets it unwarmed: it runs the warm-up itself when its decode is done and
  // the cores are still busy with their share.
  std::mutex ctx_m;
  std::condition_variable ctx_cv;
  bool ctx_created = false, ctx_warming = false;  // (under ctx_m)
  std::atomic<bool> ctx_urgent{false};
  bool ctx_warm = false;
  // What a first call pays beyond its kernels, paid ahead: (a) for the PROCESS — the library's code objects mapped onto the device by a
  // first launch, the copy engines' first use — with one tiny collapse and tbk_warmup on whatever context `c` is; (b) for the context
  // that will run the real call — its second queue, its list machines' first dispatch, its staging buffer: tbk_warmup(ctx).
  auto warm_process = [&](tbk_ctx* c) {
    // (65 inputs of one record each: more than 64 inputs take the window path, whose kernels — the larger part of the library's
    // device code — would otherwise be mapped by the first real collapse)
    constexpr uint32_t WK = 65;
    uint32_t fo[WK + 1], co[WK + 1], cg[WK];
    uint8_t tb0[WK], mq[WK], st[WK];
    int32_t ti[WK], po[WK], nh[WK];
    uint16_t fl[WK];
    for (uint32_t i = 0; i < WK; ++i) fo[i] = co[i] = i, cg[i] = 50u << 4, tb0[i] = 0, mq[i] = 60, st[i] = '.', ti[i] = 0, po[i] = 10, nh[i] = 1, fl[i] = 0;
    fo[WK] = co[WK] = WK;
    tbk_soa_in w;
    memset(&w, 0, sizeof(w));
    w.mem = TBK_MEM_HOST;
    w.n_files = WK, w.n_records = WK, w.n_cigar_ops = WK;
    w.file_off = fo, w.tbmerged = tb0, w.tid = ti, w.pos = po, w.flag = fl, w.mapq = mq, w.strand = st, w.nh = nh, w.cig_off = co, w.cig = cg;
    uint32_t wrep[WK];
    double wyc[WK];
    int64_t wyx[WK];
    int32_t wyd[WK];
    tbk_groups_out wo;
    memset(&wo, 0, sizeof(wo));
    wo.mem = TBK_MEM_HOST;
    wo.cap_groups = WK;
    wo.rep = wrep, wo.yc = wyc, wo.yx = wyx, wo.yd = wyd;
    tbk_collapse_opts wopt = opt;
    (void)api.collapse_tile(c, &wopt, &w, &wo);
    if (c != ctx && !getenv("TBK_NO_WARMUP2")) (void)api.warmup(c);  // (the engines; ctx gets its own call below)
  };
  auto warm_own = [&]() {
    if (ctx_warm || !api_ok || rc != 0 || getenv("TBK_NO_WARMUP") || getenv("TBK_NO_WARMUP2")) return;
    ctx_warm = true;
    (void)api.warmup(ctx);
  };
  std::thread ctx_thread([&]() {
    api_ok = api.load();
    if (api_ok) rc = api.create(dev, &ctx);
    if (api_ok && rc == 0 && dev_writer) dw = new DeviceWriter(api, tbh::cpu_budget());
    ms_ctx_ready = tms(t_start, tnow());
    bool warm_here;
    {
      std::lock_guard<std::mutex> lk(ctx_m);
      ctx_created = true;
      warm_here = !ctx_urgent.load();
      ctx_warming = warm_here;
    }
    ctx_cv.notify_all();
    // (page-locking the writer's staging buffers: ~ 12 ms, beside the decode — and behind the hand-over: a decode that waits for the
    // context does not wait for this)
    if (dw) (void)dw->reserve();
    // ... and a second context for the writer's second encode thread (its queue: ~ 10 ms to create)
    if (dw && !getenv("TBK_DW_ONE_ENCODER")) {
      tbk_ctx* c2 = nullptr;
      if (api.create(dev, &c2) == 0) dw->set_second(c2);
    }
    const bool can_warm = api_ok && rc == 0 && !getenv("TBK_NO_WARMUP");
    if (warm_here) {
      if (can_warm) warm_process(ctx), warm_own();
      {
        std::lock_guard<std::mutex> lk(ctx_m);
        ctx_warming = false;
      }
      ctx_cv.notify_all();
    } else if (can_warm) {
      // the context is in use already (the hybrid path's device decode): the process-wide part on a context of this thread's own,
      // beside that decode; the decode thread does the context's part when its call has returned
      tbk_ctx* wc = nullptr;
      if (api.create(dev, &wc) == 0) {
        warm_process(wc);
        api.destroy(wc);
      }
    }
  });
  bool ctx_ready = false;
  auto check_ctx = [&]() {
    if (!api_ok) GError("Error: cannot load libtbk.so (%s); this build has no CPU collapse path\n", api.error.c_str());
    if (rc != 0) GError("Error: cannot use GPU %d (%s); this build has no CPU collapse path\n", dev, api.strerror_(rc));
  };
  auto need_ctx = [&]() {
    if (ctx_ready) return;
    ctx_thread.join();
    ctx_ready = true;
    check_ctx();
  };
  // (the hybrid path's decode thread: the context as soon as it exists; the thread above is NOT joined here)
  auto need_ctx_now = [&]() {
    ctx_urgent.store(true);
    std::unique_lock<std::mutex> lk(ctx_m);
    ctx_cv.wait(lk, [&] { return ctx_created && !ctx_warming; });
    lk.unlock();
    check_ctx();
  };
  inRecords.start();
  auto t_ctx = tnow();
  double ms_load = 0, ms_gpu = 0, ms_tag = 0, ms_inflate = 0, ms_dev_write = 0;
  uint64_t dev_payload = 0, dev_z = 0;

  int nthreads = tbh::cpu_budget();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 128) nthreads = 128;
  if (const char* e = getenv("TBK_THREADS")) nthreads = std::max(1, atoi(e));
  // The inputs stream through in tiles of about TBK_TILE_RECORDS records, cut where no read of any input reaches across
  // (TInputFiles::next_tile): exact — nothing the collapse computes crosses such a point — and host memory holds one tile's
  // window of every input instead of the inflated files (the reference holds one record per input, tmerge.cpp:331-344).
  size_t tile_records = getenv("TBK_TILE_RECORDS") ? (size_t)atoll(getenv("TBK_TILE_RECORDS")) : ((size_t)64 << 20);
  uint64_t inCounter = 0, outCounter = 0;
  {
    GSamWriter outfile(outfname, inRecords.header(), GSamFile_BAM);
    TbkTile tile;
    // output arrays sized by the upper bound (one group per record) but never initialised: the pages a call does not write
    // are never touched (a value-initialising resize of 32 M entries x 24 B costs more than the collapse itself)
    RawBuf<uint32_t> rep;
    RawBuf<double> yc;
    RawBuf<int64_t> yx;
    RawBuf<int32_t> yd;
    TInputFiles::TilePlan plan;
    size_t n_tiles = 0;
    std::function<tbh::RecView(uint32_t)> get_record = [&](uint32_t g) { return inRecords.record(rep[g]); };
    // flushPData tagging (tiebrush.cpp:506-525): the groups are independent, so slices of them are tagged, framed and
    // deflated by worker threads into per-slice runs of BGZF members, which then go to the writer in order
    // (gfirst: the groups before it are in the file already — the device writer's chunks before the one it refused, devwriter.h)
    auto write_groups_arr = [&](uint32_t ng_all, const std::function<tbh::RecView(uint32_t)>& rec_of, const double* ycp, const int64_t* yxp, const int32_t* ydp,
                                uint32_t gfirst) {
      const uint32_t ng = ng_all > gfirst ? ng_all - gfirst : 0;  // groups to write
      const int nt = ng < 4096 ? 1 : nthreads;
      // slices of ~16 K groups, taken by the workers as they come free (a static split leaves the cores that drew sparse
      // regions idle); the calling thread appends every slice's members to the file as soon as all earlier ones are out
      const uint32_t per = nt == 1 ? (ng ? ng : 1) : (uint32_t)16384;
      const uint32_t nsl = ng ? (ng + per - 1) / per : 0;
      std::vector<std::vector<uint8_t>> runs((size_t)nsl);
      std::unique_ptr<std::atomic<int>[]> ready(new std::atomic<int>[nsl ? nsl : 1]);
      for (uint32_t i = 0; i < nsl; ++i) ready[i].store(0);
      std::atomic<uint32_t> next_slice{0};
      // the writer sleeps until the slice it needs is out (no spinning beside fully subscribed workers); a worker that fails
      // says so here and the calling thread reports it once every worker has stopped
      std::mutex ready_m;
      std::condition_variable ready_cv;
      std::atomic<bool> failed{false};
      const int level = outfile.level();
      auto tag_slice = [&](uint32_t sl, std::vector<uint8_t>& o, tbh::BamRec& rr) {
        const uint32_t g0 = gfirst + sl * per, g1 = std::min(ng_all, g0 + per);
        // flushPData's tags on every representative of the slice, then the slice deflates itself (tagwrite.h)
        if (!tbh::tag_and_deflate(g0, g1, rec_of, ycp, yxp, ydp, level, o, rr, runs[(size_t)sl])) failed.store(true);
        {
          std::lock_guard<std::mutex> lk(ready_m);
          ready[sl].store(1, std::memory_order_release);
        }
        ready_cv.notify_all();
      };
      auto worker = [&]() {
        std::vector<uint8_t> o;
        o.reserve((size_t)per * 96);
        tbh::BamRec rr;
        for (;;) {
          const uint32_t sl = next_slice.fetch_add(1);
          if (sl >= nsl) break;
          if (failed.load()) {  // (stop working, but let the writer's wait for this slice end)
            std::lock_guard<std::mutex> lk(ready_m);
            ready[sl].store(1, std::memory_order_release);
            ready_cv.notify_all();
            continue;
          }
          tag_slice(sl, o, rr);
        }
      };
      std::vector<std::thread> th;
      if (nt > 1)
        for (int t = 0; t < nt; ++t) th.emplace_back(worker);
      else
        worker();
      for (uint32_t sl = 0; sl < nsl; ++sl) {
        if (!ready[sl].load(std::memory_order_acquire)) {
          std::unique_lock<std::mutex> lk(ready_m);
          ready_cv.wait(lk, [&] { return ready[sl].load(std::memory_order_acquire) != 0; });
        }
        if (failed.load()) break;
        outfile.write_members(runs[(size_t)sl].data(), runs[(size_t)sl].size());
        std::vector<uint8_t>().swap(runs[(size_t)sl]);
      }
      for (auto& x : th) x.join();
      if (failed.load()) GError("Error: deflate failed\n");
    };
    uint32_t dev_groups_done = 0;  // groups the device writer wrote before it refused a chunk: the host writer goes on from there
    // the whole-input paths leave the tags' values on the device for the device writer (tbk_collapse_opts.keep_results): only `rep`
    // comes back with the call.  When the host writer has to write after all, it fetches the values of the groups it writes.
    bool results_kept = false;
    auto write_groups = [&](uint32_t ng) {
      const uint32_t gfirst = dev_groups_done;
      dev_groups_done = 0;
      if (results_kept && ng > gfirst) {
        yc.resize(ng), yx.resize(ng), yd.resize(ng);
        const int frc = api.kept_results(ctx, gfirst, ng - gfirst, nullptr, yc.data() + gfirst, yx.data() + gfirst, yd.data() + gfirst);
        if (frc != 0) GError("Error: fetching the collapse's results failed: %s (%s)\n", api.strerror_(frc), api.last_error(ctx));
      }
      write_groups_arr(ng, get_record, yc.data(), yx.data(), yd.data(), gfirst);
    };
    const bool keep_for_writer = dev_writer && outfile.level() != 0 && !getenv("TBK_NO_KEEP_RESULTS");
    // ... and `rep` comes back once the number of groups is known, into pages that are resident already when a dead array of the host's
    // input tile is long enough (4 bytes per record: tid) — a fresh block of tens of megabytes is faulted in at 1-2 GB/s this late in the run
    auto fetch_rep = [&](uint32_t ng, int32_t* dead, size_t dead_n) {
      if (dead && ng <= dead_n) rep.borrow((uint32_t*)dead, dead_n);
      else rep.resize(ng ? ng : 1);
      auto f0 = tnow();
      const int frc = api.kept_results(ctx, 0, ng, rep.data(), nullptr, nullptr, nullptr);
      if (frc != 0) GError("Error: fetching the collapse's results failed: %s (%s)\n", api.strerror_(frc), api.last_error(ctx));
      if (timing) fprintf(stderr, "representatives of %u groups fetched in %.1f ms (%s)\n", ng, tms(f0, tnow()), rep.borrowed ? "into the input tile's pages" : "into a new block");
    };
    // the same on the device (devwriter.h): tags, framing and BGZF deflate as kernels, the host only gathers the records it decoded
    // itself and appends the finished members.  false: the host writer above takes the groups from dev_groups_done on.
    auto write_groups_device = [&](uint32_t ng, uint32_t n_dev, const std::function<tbh::RecView(uint32_t)>& host_record,
                                   const std::function<void(uint32_t, int)>& host_prefetch = nullptr) {
      if (!dev_writer || outfile.level() == 0) return false;
      need_ctx();
      if (!dw) return false;
      auto a = tnow();
      uint64_t pb = 0, zb = 0;
      std::string why;
      uint32_t done = 0;
      const bool ok = dw->write(ctx, outfile, ng, rep.data(), results_kept ? nullptr : yc.data(), yx.data(), yd.data(), n_dev, host_record, &pb, &zb, why, &done, host_prefetch);
      if (!ok && timing) fprintf(stderr, "device writer stopped after %u of %u groups (%s): host writer\n", done, ng, why.c_str());
      ms_dev_write += tms(a, tnow()), dev_payload += pb, dev_z += zb;
      dev_groups_done = ok ? 0 : done;
      return ok;
    };
    // half of what the host may still use (MemAvailable, the cgroup's limit): what the whole-input loaders may fill with inflated inputs
    auto host_budget = []() {
      size_t budget = (size_t)8 << 30;
      if (FILE* mf = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof(line), mf))
          if (strncmp(line, "MemAvailable:", 13) == 0) budget = (size_t)atoll(line + 13) * 1024 / 2;
        fclose(mf);
      }
      if (FILE* cf = fopen("/sys/fs/cgroup/memory.max", "r")) {
        char q[64];
        if (fscanf(cf, "%63s", q) == 1 && strcmp(q, "max") != 0) budget = std::min<size_t>(budget, (size_t)atoll(q) / 2);
        fclose(cf);
      }
      return budget;
    };
    bool done_fast = false;
    bool skip_fast = false;  // (the hybrid path below loaded a part of the inputs and gave up: the streaming path takes over)
    tbh::FastTile& ft = *new tbh::FastTile();  // (gigabytes, needed until the last record is written: left to the process exit)
    // ---- hybrid decode: records with SEQ / QUAL are ~ 240 inflated bytes each and the run is BGZF on the host's cores (SURVEY.md
    // §8 f1).  The GPU inflates and decodes the first files of the list (tbk_bam_decode) WHILE the cores inflate and decode the rest
    // (fastload.cpp); tbk_tile_join makes one device tile of the two, the collapse runs on it, and a representative's raw record
    // comes from wherever its file was decoded.  TBK_HYBRID=0 / 1 switches it off / on (default: inputs of 768 MB and more),
    // TBK_HYBRID_SHARE = the device's share of the compressed bytes in per cent (default 40).
    {
      const size_t k = inRecords.freaders.size();
      const char* hy = getenv("TBK_HYBRID");
      bool eligible = k >= 2 && !(getenv("TBK_HOST_FAST") && atoi(getenv("TBK_HOST_FAST")) == 0) && !getenv("TBK_DEVICE_DECODE") && !getenv("TBK_TILE_RECORDS") &&
                      opt.strategy != TBK_STRAT_FULL && !opt.collapse_same && !(hy && atoi(hy) == 0);
      std::vector<std::string> paths(k);
      std::vector<uint64_t> fsz(k, 0);
      uint64_t total = 0;
      for (size_t f = 0; f < k && eligible; ++f) {
        paths[f] = inRecords.freaders[f]->fname;
        struct stat st;
        if (inRecords.freaders[f]->tbMerged || !tbh::bgzf_probe(paths[f]) || stat(paths[f].c_str(), &st) != 0) eligible = false;
        else fsz[f] = (uint64_t)st.st_size, total += fsz[f];
      }
      if (eligible && !(hy && atoi(hy) != 0) && total < ((uint64_t)768 << 20)) eligible = false;
      if (eligible) {
        // The device's share of the compressed bytes: both sides should end together.  The device starts late — the HIP runtime takes
        // ~ 0.1 s to come up (t_ctx = 0.15: the helper thread's warm-up beside the decode and the context's own behind it cost the
        // device's side another 0.05), the cores work alone meanwhile — and is then several times faster: with x of T bytes on the device,
        // t_ctx + x / R_dev = (T - x) / R_host.  Rates measured on an MI355X box with a 16-core quota (round 5's end-to-end legs, tools/e2e_leg.py): the
        // device side 4.7 GB/s of compressed BAM (upload, inflate, record index, SoA), a core 0.18 GB/s (inflate with the record index
        // riding along, SoA).  1.8 GB of input: 54 % (17 of 32 files; measured round 6, five runs a share: 16 files 407 ms for the
        // decode, 17 files 351-361, 18 files 374-382, 19 files 377-384); 7.1 GB: 63 %.  TBK_HYBRID_SHARE (per cent) overrides.
        const int host_threads = getenv("TBK_THREADS") ? nthreads : std::max(2, nthreads - 3);
        double share;
        if (getenv("TBK_HYBRID_SHARE")) {
          share = atof(getenv("TBK_HYBRID_SHARE")) / 100.0;
        } else {
          const double T = (double)total / 1e9, r_dev = 4.7, r_host = 0.18 * host_threads, t_ctx = 0.15;
          const double x = (T / r_host - t_ctx) / (1.0 / r_dev + 1.0 / r_host);
          share = std::min(0.9, std::max(0.2, x / T));
        }
        size_t kd = 0;
        uint64_t acc = 0;
        while (kd + 1 < k && (double)(acc + fsz[kd]) <= share * (double)total + (double)fsz[kd] / 2) acc += fsz[kd++];
        if (kd == 0) kd = 1;
        const size_t budget = host_budget();
        auto t0 = tnow();
        if (timing) fprintf(stderr, "hybrid decode starts at %.1f ms\n", tms(t_start, t0));
        // the device's share, on a thread of its own: read the files, wait for the context, decode
        tbk_soa_in in_d;
        memset(&in_d, 0, sizeof(in_d));
        std::vector<uint32_t> fo_d(kd + 1, 0);
        std::vector<uint8_t> tb_d(kd, 0);
        int rc_d = -1;
        bool read_ok = true;
        double ms_dread = 0, ms_ddec = 0, ms_dcall = 0;
        std::thread dth([&]() {
          auto d0 = tnow();
          std::vector<FileMap> comp(kd);
          for (size_t f = 0; f < kd; ++f)
            if (!comp[f].map(paths[f]) || comp[f].n != fsz[f]) read_ok = false;
          auto d1 = tnow();
          ms_dread = tms(d0, d1);
          if (!read_ok) return;
          need_ctx_now();
          auto d2 = tnow();
          std::vector<const uint8_t*> ptr(kd);
          for (size_t f = 0; f < kd; ++f) ptr[f] = comp[f].p;
          if (ktiming) (void)api.set_profiling(ctx, 1);
          rc_d = api.bam_decode(ctx, (uint32_t)kd, ptr.data(), fsz.data(), tb_d.data(), 0, 0, &in_d, fo_d.data());
          ms_dcall = tms(d2, tnow());
          if (ktiming) {  // the call's kernels (HIP events): what of its wall time the GPU was busy with
            tbk_kernel_time kt[64];
            const int nk = api.kernel_times(ctx, kt, 64);
            std::string line = "device decode kernels ms:";
            double sum = 0;
            for (int i = 0; i < nk; ++i) {
              char b[96];
              snprintf(b, sizeof(b), " %s %.1f (%u)", kt[i].name, kt[i].ms, kt[i].launches);
              line += b;
              sum += kt[i].ms;
            }
            fprintf(stderr, "%s | sum %.1f of the call's %.1f\n", line.c_str(), sum, ms_dcall);
            (void)api.set_profiling(ctx, 0);
          }
          for (auto& c : comp) c.unmap();
          if (rc_d == 0 && acc > 0) {  // the arena for the joined tile, sized while the cores are still decoding their share
            // (tbk_reserve_tile sizes for the window path AND a deferred YD stage that borrows its range of the arena: 165 bytes a record.
            // This call defers nothing — 84 bytes a record and the CIGAR words —: six tenths of the tile's size asks for what it takes.
            // An allocation of gigabytes is now and then 20-25 ms per GB of the driver's time)
            const double up = (double)total / (double)acc * 1.05 * 0.6;
            (void)api.reserve_tile(ctx, (uint64_t)((double)in_d.n_records * up), (uint64_t)((double)in_d.n_cigar_ops * up));
          }
          ms_ddec = tms(d1, tnow());
          warm_own();  // (the helper thread left the context's own part to this one: need_ctx_now)
        });
        // the cores' share
        std::vector<std::string> ph(paths.begin() + (long)kd, paths.end());
        std::vector<uint8_t> tbh_(k - kd, 0);
        bool fits = false;
        std::string err;
        // (the device's side needs cores too while it runs — the HIP start-up, then the threads that feed the upload ring —, and a
        // container's CPU quota stalls EVERY thread of the process once the sum goes over it: the loader leaves them room)
        const bool okh = tbh::fast_load(ph, tbh_, host_threads, budget, ft, &fits, err);
        auto t_host = tnow();
        dth.join();
        auto t1 = tnow();
        need_ctx();  // (the helper thread ended long ago; this only joins it)
        if (!okh) GError("Error: reading the input failed (%s)\n", err.c_str());
        if (!read_ok) GError("Error: reading the input failed\n");
        if (rc_d != 0 && rc_d != TBK_ENOMEM && rc_d != TBK_E2BIG) GError("Error: decoding the input on the GPU failed: %s (%s)\n", api.strerror_(rc_d), api.last_error(ctx));
        bool ok = fits && rc_d == 0;
        tbk_soa_in in;
        std::vector<uint32_t> fo(k + 1, 0);
        std::vector<uint8_t> tbm(k, 0);
        tbk_groups_out out;
        memset(&out, 0, sizeof(out));
        RawBuf<uint64_t> roff;
        RawBuf<uint8_t> blob;
        std::vector<uint32_t> dev_slot;
        const uint32_t n_d = in_d.n_records;
        auto t_join = t1, t_col = t1, t_rec = t1;
        if (ok) {
          tbk_soa_in in_h = ft.view();
          rc = api.tile_join(ctx, &in_d, &in_h, &in, fo.data(), tbm.data());
          t_join = tnow();
          if (rc == 0) {
            const size_t n = in.n_records;
            out.mem = TBK_MEM_HOST;
            out.cap_groups = (uint32_t)(n ? n : 1);
            tbk_collapse_opts copt = opt;
            copt.keep_results = keep_for_writer ? 1 : 0;
            if (!keep_for_writer) {
              rep.resize(n ? n : 1), yc.resize(n ? n : 1), yx.resize(n ? n : 1), yd.resize(n ? n : 1);
              out.rep = rep.data(), out.yc = yc.data(), out.yx = yx.data(), out.yd = yd.data();
            }
            if (ktiming) (void)api.set_profiling(ctx, 1);
            rc = api.collapse_tile(ctx, &copt, &in, &out);
            if (rc == 0 && getenv("TBK_TEST_WHOLE_ENOMEM")) rc = TBK_ENOMEM;  // test hook: exercise the fall-back below
            results_kept = rc == 0 && keep_for_writer;
            if (results_kept) fetch_rep(out.n_groups, ft.tid, ft.n);  // (the host part's SoA went to the device with tbk_tile_join)
            t_col = tnow();
            if (ktiming) {
              tbk_kernel_time kt[64];
              const int nk = api.kernel_times(ctx, kt, 64);
              double sum = 0;
              for (int i = 0; i < nk; ++i) sum += kt[i].ms;
              fprintf(stderr, "collapse call: %.1f ms, its kernels %.1f ms in %d kinds\n", tms(t_join, t_col), sum, nk);
              (void)api.set_profiling(ctx, 0);
            }
            if (rc == TBK_EUNSORTED) GError("Error: an input file is not coordinate-sorted!\n");
          }
          bool wrote_dev = false;
          if (rc == 0) {
            wrote_dev = write_groups_device(
                out.n_groups, n_d,
                [&](uint32_t g) {
                  tbh::RecView v;
                  v.p = ft.record(rep[g] - n_d, &v.len);
                  return v;
                },
                [&](uint32_t g, int stage) { stage == 0 ? ft.prefetch_index(rep[g] - n_d) : ft.prefetch_record(rep[g] - n_d); });
            t_rec = tnow();
          }
          if (rc == 0 && !wrote_dev) {  // the representatives the device decoded: their raw records come back from there
            std::vector<uint32_t> dev_rep;
            dev_slot.assign(out.n_groups, 0);
            for (uint32_t g = 0; g < out.n_groups; ++g)
              if (rep[g] < n_d) {
                dev_slot[g] = (uint32_t)dev_rep.size();
                dev_rep.push_back(rep[g]);
              }
            roff.resize(dev_rep.size() + 1);
            blob.resize(dev_rep.size() * 260 + 4096);
            rc = api.bam_records(ctx, dev_rep.data(), (uint32_t)dev_rep.size(), TBK_MEM_HOST, blob.data(), blob.size(), roff.data());
            if (rc == TBK_E2BIG) {
              blob.resize(roff[dev_rep.size()]);
              rc = api.bam_records(ctx, dev_rep.data(), (uint32_t)dev_rep.size(), TBK_MEM_HOST, blob.data(), blob.size(), roff.data());
            }
            t_rec = tnow();
          }
          ok = rc == 0;
          if (!ok && rc != TBK_ENOMEM && rc != TBK_E2BIG && rc != TBK_EUNSUPPORTED)
            GError("Error: GPU collapse failed: %s (%s)\n", api.strerror_(rc), api.last_error(ctx));
          if (ok && wrote_dev) {
            auto t3 = tnow();
            if (timing)
              fprintf(stderr,
                      "hybrid path ms: device %zu of %zu files (context ready at %.1f | map %.1f | decode incl. context %.1f, the call %.1f) beside host (read %.1f | inflate %.1f | index %.1f | SoA %.1f) = %.1f | "
                      "join %.1f | collapse %.1f | gather + tag + deflate (GPU) + write %.1f (%.1f MB of records -> %.1f MB)\n",
                      kd, k, ms_ctx_ready, ms_dread, ms_ddec, ms_dcall, ft.ms_read, ft.ms_inflate, ft.ms_index, ft.ms_soa, tms(t0, t1), tms(t1, t_join), tms(t_join, t_col), tms(t_col, t3),
                      dev_payload / 1e6, dev_z / 1e6);
            ms_inflate += tms(t0, t1);
            ms_gpu += tms(t1, t_col);
            ms_tag += tms(t_col, t3);
            inCounter += out.n_passed;
            outCounter += out.n_groups;
            n_tiles = 1;
            done_fast = true;
          }
        }
        {
          auto r0 = tnow();
          if (ctx_ready) api.bam_release(ctx);
          if (timing) fprintf(stderr, "bam_release %.1f ms\n", tms(r0, tnow()));
        }
        if (done_fast) {
        } else if (!ok) {
          if (timing) fprintf(stderr, "hybrid decode given up (%s): streaming host path\n", rc_d != 0 ? api.strerror_(rc_d) : (fits ? api.strerror_(rc) : "the host's share does not fit"));
          tbh::big_release_all(nthreads);
          skip_fast = true;
        } else {
          get_record = [&](uint32_t g) {
            tbh::RecView v;
            if (rep[g] < n_d) {
              const uint32_t s = dev_slot[g];
              v.p = blob.data() + roff[s] + 4;
              v.len = (uint32_t)(roff[s + 1] - roff[s] - 4);
            } else {
              v.p = ft.record(rep[g] - n_d, &v.len);
            }
            return v;
          };
          write_groups(out.n_groups);
          auto t3 = tnow();
          if (timing)
            fprintf(stderr,
                    "hybrid path ms: device %zu of %zu files (read %.1f | decode incl. context %.1f) beside host (read %.1f | inflate %.1f | index %.1f | SoA %.1f) = %.1f | "
                    "join %.1f | collapse %.1f | fetch representatives %.1f | tag+deflate+write %.1f\n",
                    kd, k, ms_dread, ms_ddec, ft.ms_read, ft.ms_inflate, ft.ms_index, ft.ms_soa, tms(t0, t1), tms(t1, t_join), tms(t_join, t_col), tms(t_col, t_rec),
                    tms(t_rec, t3));
          (void)t_host;
          ms_inflate += tms(t0, t1);
          ms_gpu += tms(t1, t_rec);
          ms_tag += tms(t_rec, t3);
          inCounter += out.n_passed;
          outCounter += out.n_groups;
          n_tiles = 1;
          done_fast = true;
        }
      }
    }
    // ---- whole-input host path: inputs that fit in memory are read, inflated and decoded into the tile in two parallel passes
    // (fastload.cpp) while the helper thread brings the device up; one collapse, one tagged output pass ----
    if (!done_fast && !skip_fast && !(getenv("TBK_HOST_FAST") && atoi(getenv("TBK_HOST_FAST")) == 0) &&
        !(getenv("TBK_DEVICE_DECODE") && atoi(getenv("TBK_DEVICE_DECODE")) != 0) && !getenv("TBK_TILE_RECORDS") && opt.strategy != TBK_STRAT_FULL &&
        !opt.collapse_same) {
      const size_t k = inRecords.freaders.size();
      bool all_bam = true;  // (SAM text inputs are converted by the streaming reader)
      std::vector<std::string> paths(k);
      std::vector<uint8_t> tb(k);
      for (size_t f = 0; f < k; ++f) {
        paths[f] = inRecords.freaders[f]->fname;
        tb[f] = inRecords.freaders[f]->tbMerged ? 1 : 0;
        all_bam = all_bam && tbh::bgzf_probe(paths[f]);
      }
      const size_t budget = host_budget();
      bool fits = false;
      std::string err;
      if (all_bam && k > 0) {
        auto t0 = tnow();
        if (!tbh::fast_load(paths, tb, nthreads, budget, ft, &fits, err)) GError("Error: reading the input failed (%s)\n", err.c_str());
        if (fits) {
          auto t1 = tnow();
          tbk_soa_in in = ft.view();
          const size_t n = ft.n;
          need_ctx();
          auto t_ctxw = tnow();
          tbk_groups_out out;
          memset(&out, 0, sizeof(out));
          out.mem = TBK_MEM_HOST;
          out.cap_groups = (uint32_t)(n ? n : 1);
          tbk_collapse_opts copt = opt;
          copt.keep_results = keep_for_writer ? 1 : 0;
          if (!keep_for_writer) {
            rep.resize(n ? n : 1), yc.resize(n ? n : 1), yx.resize(n ? n : 1), yd.resize(n ? n : 1);
            out.rep = rep.data(), out.yc = yc.data(), out.yx = yx.data(), out.yd = yd.data();
          }
          rc = api.collapse_tile(ctx, &copt, &in, &out);
          if (rc == 0 && getenv("TBK_TEST_WHOLE_ENOMEM")) rc = TBK_ENOMEM;  // test hook: exercise the fall-back below
          results_kept = rc == 0 && keep_for_writer;
          if (results_kept) fetch_rep(out.n_groups, ft.tid, ft.n);  // (the SoA went to the device with the call)
          auto t2 = tnow();
          if (rc == TBK_EUNSORTED) GError("Error: an input file is not coordinate-sorted!\n");
          if (rc == TBK_ENOMEM || rc == TBK_E2BIG) {  // one tile of everything is more than the GPU takes: the streaming path bounds it
            if (timing) fprintf(stderr, "whole-input tile not used (%s: %s): streaming host path\n", api.strerror_(rc), api.last_error(ctx));
            tbh::big_release_all(nthreads);
          } else {
            if (rc != 0) GError("Error: GPU collapse failed: %s (%s)\n", api.strerror_(rc), api.last_error(ctx));
            get_record = [&](uint32_t g) {
              tbh::RecView v;
              v.p = ft.record(rep[g], &v.len);
              return v;
            };
            if (!write_groups_device(out.n_groups, 0, get_record, [&](uint32_t g, int stage) { stage == 0 ? ft.prefetch_index(rep[g]) : ft.prefetch_record(rep[g]); }))
              write_groups(out.n_groups);
            auto t3 = tnow();
            if (timing)
              fprintf(stderr, "host path ms: read %.1f | inflate %.1f | index %.1f | SoA %.1f | wait for the device %.1f | collapse (PCIe incl.) %.1f | tag+deflate+write %.1f\n",
                      ft.ms_read, ft.ms_inflate, ft.ms_index, ft.ms_soa, tms(t1, t_ctxw), tms(t_ctxw, t2), tms(t2, t3));
            ms_inflate += tms(t0, t1);
            ms_gpu += tms(t1, t2);
            ms_tag += tms(t2, t3);
            inCounter += out.n_passed;
            outCounter += out.n_groups;
            n_tiles = 1;
            done_fast = true;
          }
        }
      }
    }
    // ---- device decode: when the inputs fit, their BGZF members go to the GPU as they are — inflate,
    // record index, aux scan and SoA happen there (tbk_bam_decode), the collapse reads the tile where it lies, and only the
    // representatives' raw records come back (tbk_bam_records) to be tagged.  Anything it cannot take falls through to the
    // streaming host path below.
    bool done_on_device = done_fast;
    if (!done_fast) {
      const char* e = getenv("TBK_DEVICE_DECODE");
      const bool want = e ? atoi(e) != 0 : false;  // (opt in: with libdeflate on every core the host inflates faster than the device path end to end)
      uint64_t total = 0;
      const size_t k = inRecords.freaders.size();
      std::vector<uint64_t> fsz(k, 0);
      for (size_t f = 0; f < k; ++f) {
        struct stat st;
        if (stat(inRecords.freaders[f]->fname.c_str(), &st) == 0) fsz[f] = (uint64_t)st.st_size;
        total += fsz[f];
      }
      const uint64_t lim = getenv("TBK_DEVICE_DECODE_MAX") ? (uint64_t)atoll(getenv("TBK_DEVICE_DECODE_MAX")) : ((uint64_t)6 << 30);
      bool all_bam = true;  // (SAM text inputs are decoded by the host)
      for (size_t f = 0; f < k; ++f) all_bam = all_bam && tbh::bgzf_probe(inRecords.freaders[f]->fname);
      if (want && all_bam && total > 0 && total <= lim) {
        auto t0 = tnow();
        std::vector<FileMap> comp(k);
        for (size_t f = 0; f < k; ++f)
          if (!comp[f].map(inRecords.freaders[f]->fname) || comp[f].n != fsz[f]) GError("Error: reading the input failed\n");
        std::vector<const uint8_t*> ptr(k);
        std::vector<uint8_t> tb(k);
        for (size_t f = 0; f < k; ++f) {
          ptr[f] = comp[f].p;
          tb[f] = inRecords.freaders[f]->tbMerged ? 1 : 0;
        }
        std::vector<uint32_t> fo(k + 1, 0);
        auto t_read = tnow();
        need_ctx();
        auto t_ctxw = tnow();
        tbk_soa_in in;
        rc = api.bam_decode(ctx, (uint32_t)k, ptr.data(), fsz.data(), tb.data(), opt.strategy == TBK_STRAT_FULL, opt.collapse_same != 0, &in, fo.data());
        auto t1 = tnow();
        for (auto& c : comp) c.unmap();
        if (rc == 0) {
          const size_t n = in.n_records;
          rep.resize(n ? n : 1);
          yc.resize(n ? n : 1);
          yx.resize(n ? n : 1);
          yd.resize(n ? n : 1);
          tbk_groups_out out;
          memset(&out, 0, sizeof(out));
          out.mem = TBK_MEM_HOST;
          out.cap_groups = (uint32_t)(n ? n : 1);
          out.rep = rep.data();
          out.yc = yc.data();
          out.yx = yx.data();
          out.yd = yd.data();
          auto t_pre = tnow();
          rc = api.collapse_tile(ctx, &opt, &in, &out);
          if (timing) fprintf(stderr, "device path: output arrays %.1f ms, collapse call %.1f ms\n", tms(t1, t_pre), tms(t_pre, tnow()));
          if (rc == 0 && getenv("TBK_TEST_WHOLE_ENOMEM")) rc = TBK_ENOMEM;  // test hook: exercise the fall-back below
          auto t_col = tnow();
          if (rc == TBK_EUNSORTED) GError("Error: an input file is not coordinate-sorted!\n");
          RawBuf<uint64_t> roff;
          RawBuf<uint8_t> blob;
          bool wrote_dev = false;
          if (rc == 0) wrote_dev = write_groups_device(out.n_groups, (uint32_t)n, [](uint32_t) { return tbh::RecView(); });
          if (rc == 0 && !wrote_dev) {
            roff.resize((size_t)out.n_groups + 1);
            blob.resize((size_t)out.n_groups * 96 + 4096);
            rc = api.bam_records(ctx, rep.data(), out.n_groups, TBK_MEM_HOST, blob.data(), blob.size(), roff.data());
            if (rc == TBK_E2BIG) {
              blob.resize(roff[out.n_groups]);
              rc = api.bam_records(ctx, rep.data(), out.n_groups, TBK_MEM_HOST, blob.data(), blob.size(), roff.data());
            }
          }
          if (rc == TBK_ENOMEM || rc == TBK_E2BIG) {  // decoded, but the whole input as one tile is more than the GPU takes:
            if (timing)                                // give the device copies back and let the streaming path bound the tile
              fprintf(stderr, "device decode given up (%s: %s): streaming host path\n", api.strerror_(rc), api.last_error(ctx));
            api.bam_release(ctx);
          } else {
            if (rc != 0) GError("Error: GPU collapse / fetching the representative records failed: %s (%s)\n", api.strerror_(rc), api.last_error(ctx));
            auto t_rec = tnow();
            api.bam_release(ctx);
            auto t2 = tnow();
            if (timing)
              fprintf(stderr, "device path ms: read files %.1f | wait for the HIP context %.1f | decode %.1f | collapse %.1f | fetch representatives %.1f | release %.1f\n",
                      tms(t0, t_read), tms(t_read, t_ctxw), tms(t_ctxw, t1), tms(t1, t_col), tms(t_col, t_rec), tms(t_rec, t2));
            get_record = [&](uint32_t g) {
              tbh::RecView v;
              v.p = blob.data() + roff[g] + 4;
              v.len = (uint32_t)(roff[g + 1] - roff[g] - 4);
              return v;
            };
            if (!wrote_dev) write_groups(out.n_groups);
            auto t3 = tnow();
            ms_inflate += tms(t0, t1);
            ms_gpu += tms(t1, t2);
            ms_tag += tms(t2, t3);
            inCounter += out.n_passed;
            outCounter += out.n_groups;
            n_tiles = 1;
            done_on_device = true;
            if (timing) fprintf(stderr, "device decode: %zu records from %llu compressed bytes\n", n, (unsigned long long)total);
          }
        } else {
          if (timing) fprintf(stderr, "device decode not used (%s: %s): streaming host path\n", api.strerror_(rc), api.last_error(ctx));
          api.bam_release(ctx);  // (whatever the failed decode left on the device goes back before the streaming path sizes its tiles)
        }
      }
    }
    // ---- streaming path: the inputs go through in tiles (TInputFiles::next_tile), and the OUTPUT side of tile i runs beside the
    // input side of tile i + 1.  A tile's representatives are copied out of the input windows right behind its collapse (the windows
    // move on with the next tile); tags, deflate and the write of the tile then belong to a writer thread with a context of its own
    // (the device writer, or every core under --writer host), while this thread inflates, decodes and collapses the next tile.
    // Two slots: a tile's arrays are free again once its members are in the file.
    struct StreamSlot {
      RawBuf<uint32_t> rep;
      RawBuf<double> yc;
      RawBuf<int64_t> yx;
      RawBuf<int32_t> yd;
      RawBuf<uint8_t> blob;   // the representatives' raw records, group after group (no block_size)
      RawBuf<uint64_t> boff;  // [ng + 1]
      uint32_t ng = 0;
      bool busy = false;
    };
    StreamSlot slots[2];
    std::mutex sm;
    std::condition_variable scv;
    std::vector<int> queue_;   // slots handed to the writer, in tile order
    bool producer_done = false;
    double ms_writer_busy = 0, ms_wait_slot = 0, ms_gather = 0;
    tbk_ctx* wctx = nullptr;   // the writer's context (the collapse of the next tile keeps `ctx` busy)
    std::thread writer_thread;
    auto writer_main = [&]() {
      for (;;) {
        int si = -1;
        {
          std::unique_lock<std::mutex> lk(sm);
          scv.wait(lk, [&] { return !queue_.empty() || producer_done; });
          if (queue_.empty()) return;
          si = queue_.front();
          queue_.erase(queue_.begin());
        }
        StreamSlot& S = slots[si];
        auto a = tnow();
        const std::function<tbh::RecView(uint32_t)> from_blob = [&S](uint32_t g) {
          tbh::RecView v;
          v.p = S.blob.data() + S.boff[g];
          v.len = (uint32_t)(S.boff[g + 1] - S.boff[g]);
          return v;
        };
        bool wrote = false;
        uint32_t wdone = 0;
        if (dev_writer && dw && wctx && outfile.level() != 0) {
          uint64_t pb = 0, zb = 0;
          std::string why;
          wrote = dw->write(wctx, outfile, S.ng, S.rep.data(), S.yc.data(), S.yx.data(), S.yd.data(), 0, from_blob, &pb, &zb, why, &wdone);
          dev_payload += pb, dev_z += zb;
          if (!wrote && timing) fprintf(stderr, "device writer stopped after %u of %u groups (%s): host writer\n", wdone, S.ng, why.c_str());
        }
        if (!wrote) write_groups_arr(S.ng, from_blob, S.yc.data(), S.yx.data(), S.yd.data(), wdone);
        const double ms = tms(a, tnow());
        ms_writer_busy += ms;
        if (wrote) ms_dev_write += ms;
        {
          std::lock_guard<std::mutex> lk(sm);
          S.busy = false;
        }
        scv.notify_all();
      }
    };
    int next_slot = 0;
    for (; !done_on_device;) {
      auto ti = tnow();
      const bool more = inRecords.next_tile(plan, tile_records, nthreads);
      ms_inflate += tms(ti, tnow());
      if (!more) break;
      ++n_tiles;
      auto t0 = tnow();
      inRecords.load_tile(tile, opt.strategy == TBK_STRAT_FULL, opt.collapse_same != 0, nthreads, &plan);
      auto t1 = tnow();
      tbk_soa_in in = tile.view();
      size_t n = tile.n();
      StreamSlot& S = slots[next_slot];
      {
        auto w0 = tnow();
        std::unique_lock<std::mutex> lk(sm);
        scv.wait(lk, [&] { return !S.busy; });
        ms_wait_slot += tms(w0, tnow());
      }
      S.rep.resize(n ? n : 1);
      S.yc.resize(n ? n : 1);
      S.yx.resize(n ? n : 1);
      S.yd.resize(n ? n : 1);
      need_ctx();
      if (!writer_thread.joinable()) {
        if (dev_writer && api.create(dev, &wctx) != 0) wctx = nullptr;  // (no second context: the host writer takes the output)
        writer_thread = std::thread(writer_main);
      }
      tbk_groups_out out;
      memset(&out, 0, sizeof(out));
      out.mem = TBK_MEM_HOST;
      out.cap_groups = (uint32_t)(n ? n : 1);
      out.rep = S.rep.data();
      out.yc = S.yc.data();
      out.yx = S.yx.data();
      out.yd = S.yd.data();
      rc = api.collapse_tile(ctx, &opt, &in, &out);
      auto t2 = tnow();
      if (rc == TBK_EUNSORTED) GError("Error: an input file is not coordinate-sorted!\n");
      if (rc != 0) GError("Error: GPU collapse failed: %s (%s)\n", api.strerror_(rc), api.last_error(ctx));
      // the representatives leave the windows: sizes per slice of groups, a prefix, the copies — every core
      {
        const uint32_t ng = out.n_groups;
        S.ng = ng;
        S.boff.resize((size_t)ng + 1);
        const int T = ng < 8192 ? 1 : nthreads;
        std::vector<uint64_t> part((size_t)T + 1, 0);
        auto slice = [&](int t, uint32_t* a, uint32_t* b) {
          *a = (uint32_t)((uint64_t)ng * (uint32_t)t / (uint32_t)T);
          *b = (uint32_t)((uint64_t)ng * ((uint32_t)t + 1) / (uint32_t)T);
        };
        auto par = [&](const std::function<void(int)>& f) {
          std::vector<std::thread> th;
          for (int t = 1; t < T; ++t) th.emplace_back(f, t);
          f(0);
          for (auto& x : th) x.join();
        };
        par([&](int t) {
          uint32_t a, b;
          slice(t, &a, &b);
          uint64_t by = 0;
          for (uint32_t g = a; g < b; ++g) by += inRecords.record(S.rep[g]).len;
          part[(size_t)t + 1] = by;
        });
        for (int t = 0; t < T; ++t) part[(size_t)t + 1] += part[(size_t)t];
        S.blob.resize((size_t)part[(size_t)T] + 16);
        par([&](int t) {
          uint32_t a, b;
          slice(t, &a, &b);
          uint64_t o = part[(size_t)t];
          for (uint32_t g = a; g < b; ++g) {
            const tbh::RecView v = inRecords.record(S.rep[g]);
            S.boff[g] = o;
            memcpy(S.blob.data() + o, v.p, v.len);
            o += v.len;
          }
        });
        S.boff[ng] = part[(size_t)T];
      }
      auto t3 = tnow();
      ms_gather += tms(t2, t3);
      inCounter += out.n_passed;
      outCounter += out.n_groups;
      inRecords.release_tile(plan);
      {
        std::lock_guard<std::mutex> lk(sm);
        S.busy = true;
        queue_.push_back(next_slot);
      }
      scv.notify_all();
      next_slot ^= 1;
      ms_load += tms(t0, t1);
      ms_gpu += tms(t1, t2);
    }
    if (writer_thread.joinable()) {
      {
        std::lock_guard<std::mutex> lk(sm);
        producer_done = true;
      }
      scv.notify_all();
      auto w0 = tnow();
      writer_thread.join();
      ms_tag += tms(w0, tnow());
      if (wctx) api.destroy(wctx);
      if (timing)
        fprintf(stderr, "streamed: %zu tiles; this thread inflate+index %.1f | SoA %.1f | collapse %.1f | gather representatives %.1f | waited for a free slot %.1f | "
                        "waited for the writer at the end %.1f; writer thread busy %.1f\n",
                n_tiles, ms_inflate, ms_load, ms_gpu, ms_gather, ms_wait_slot, tms(w0, tnow()), ms_writer_busy);
    }
    if (timing) fprintf(stderr, "tiles: %zu (at %.1f ms)\n", n_tiles, tms(t_start, tnow()));
  }
  auto t_closed = tnow();
  if (timing) fprintf(stderr, "writer closed at %.1f ms\n", tms(t_start, t_closed));
  need_ctx();
  // (no tbk_destroy / stop: the process ends below, the OS reclaims device and host memory faster than piecewise frees)
  if (timing && dev_z) fprintf(stderr, "device writer: %.1f MB of tagged records -> %.1f MB of BGZF members in %.1f ms\n", dev_payload / 1e6, dev_z / 1e6, ms_dev_write);
  if (timing)
    fprintf(stderr, "timing ms: open+context %.1f | inflate+index %.1f | SoA %.1f | collapse (PCIe incl.) %.1f | tag+queue %.1f | total to writer close %.1f\n",
            tms(t_start, t_ctx), ms_inflate, ms_load, ms_gpu, ms_tag, tms(t_start, t_closed));
  double p = 100.00 - (double)(outCounter * 100.00) / (double)inCounter;
  GMessage("%ld input records written as %ld (%.2f%% reduction)\n", (long)inCounter, (long)outCounter, p);
  fflush(stdout);
  fflush(stderr);
  // the gigabytes of the whole-input path go back in parallel: a process that just exits returns them in one thread while its
  // caller waits (measured: 0.17 s for 2.7 GB)
  {
    auto a = tnow();
    if (!getenv("TBK_NO_RELEASE")) tbh::big_release_all(nthreads);  // (TBK_NO_RELEASE: diagnosis — what the exit costs without it)
    if (timing) fprintf(stderr, "released the large buffers in %.1f ms\n", tms(a, tnow()));
  }
  if (getenv("TBK_EXIT_TIMING")) {  // (diagnosis: what is left of the process exit)
    auto a = tnow();
    if (atoi(getenv("TBK_EXIT_TIMING")) > 1) api.destroy(ctx);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    fprintf(stderr, "exit timing: tbk_destroy %.1f ms; _exit at %.3f\n", tms(a, tnow()), (double)ts.tv_sec + ts.tv_nsec * 1e-9);
    if (atoi(getenv("TBK_EXIT_TIMING")) > 2) return 0;  // (a plain return: what a profiler's exit handlers need to write their traces)
  }
  if (const char* e = getenv("TBK_EXIT_SLEEP_MS")) usleep((useconds_t)atoi(e) * 1000);  // (diagnosis)
  _exit(0);  // the output is closed and flushed: skip the runtime's teardown of a process that is done (tens of ms of hipFree / unload)
}
