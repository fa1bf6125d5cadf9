// tiecov — drop-in command line of the reference's coverage tool (/root/reference/src/tiecov.cpp:345-573).
// The per-record loop (:435-499: bundles, addCov, addJunction, addMean, flushes) is replaced by
// tbk_coverage_tile / tbk_sample_tile (HIP); BAM decode and text output stay on the host.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/tbk.h"
#include "GSam.h"
#include "bigwig.h"
#include "args.h"

#define VERSION "0.0.7"

static const char* USAGE =
    "TieCov v" VERSION " (MI355X build)\n"
    "Summarises a (TieBrush-collapsed) BAM file as BED-like tracks.\n"
    "\n"
    " usage: tiecov [-s out.sample] [-c out.coverage] [-j out.junctions] input.bam\n"
    "\n"
    "  -h,--help    print this text and exit\n"
    "  --version    print the version and exit\n"
    "  -c PREFIX    per-base coverage as bedGraph (the YC tag weighs each alignment)\n"
    "  -j PREFIX    splice junctions as BED\n"
    "  -s PREFIX    estimated number of samples per position as bedGraph (needs @CO SAMPLE: header lines)\n"
    "  -W           write the coverage (-c) as a bigWig file (PREFIX.bigwig) instead of a bedGraph\n"
    " At least one of -c / -j / -s is required.\n";

static bool ends_with(const std::string& s, const char* suf) {
  size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// Text output: the lines are independent, so slices of them are formatted by worker threads (same printf formats as the
// reference) and written in order.
template <class F>
static void emit_lines(FILE* f, uint32_t n, F fmt) {
  unsigned hw = (unsigned)tbh::cpu_budget();
  size_t nt = n < 50000 ? 1 : std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, 32));
  std::vector<std::string> parts(nt);
  auto work = [&](size_t t) {
    const uint32_t lo = (uint32_t)((uint64_t)n * t / nt), hi = (uint32_t)((uint64_t)n * (t + 1) / nt);
    std::string& o = parts[t];
    o.reserve((size_t)(hi - lo) * 40);
    char b[1024];
    for (uint32_t i = lo; i < hi; ++i) {
      int len = fmt(i, b, sizeof(b));
      if (len < 0) len = 0;
      if ((size_t)len >= sizeof(b)) {  // a very long reference name: format again into a buffer that fits
        std::vector<char> big((size_t)len + 1);
        len = fmt(i, big.data(), big.size());
        o.append(big.data(), (size_t)len);
      } else {
        o.append(b, (size_t)len);
      }
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (auto& o : parts)
    if (!o.empty() && fwrite(o.data(), 1, o.size(), f) != o.size()) GError("Error: failed to write an output line\n");
}

int main(int argc, char* argv[]) {
  Args args(argc, argv, "help;verbose;version;DVWhc:s:j:");
  if (!args.error().empty()) {
    GMessage("%s\n%s\n", USAGE, args.error().c_str());
    return 1;
  }
  if (args.getOpt('h') || args.getOpt("help")) {
    GMessage("%s", USAGE);
    return 1;  // the reference exits 1 here (tiecov.cpp:535-538)
  }
  if (args.getOpt("version")) {
    fprintf(stdout, "%s\n", VERSION);
    return 0;
  }
  if (!args.getOpt('c') && !args.getOpt('s') && !args.getOpt('j')) {
    GMessage("%s", USAGE);
    GMessage("\nError: at least one of -c/-j/-s arguments required!\n");
    return 1;
  }
  const bool bigwig = args.getOpt('W') != nullptr;
  if (args.getOpt("verbose") || args.getOpt('V')) {
    fprintf(stderr, "Running TieCov " VERSION ". Command line:\n");
    args.printCmdLine(stderr);
  }
  std::string covfname = args.getOpt('c') ? args.getOpt('c') : "";
  std::string jfname = args.getOpt('j') ? args.getOpt('j') : "";
  std::string sfname = args.getOpt('s') ? args.getOpt('s') : "";
  if (args.startNonOpt() == 0) {
    GMessage("%s", USAGE);
    GMessage("\nError: no input file provided!\n");
    return 1;
  }
  std::string infname = args.nextNonOpt();
  GSamReader samreader(infname.c_str(), SAM_QNAME | SAM_FLAG | SAM_RNAME | SAM_POS | SAM_CIGAR | SAM_AUX);
  sam_hdr_t* hdr = samreader.header();
  FILE *coutf = nullptr, *joutf = nullptr, *soutf = nullptr;
  tbh::BigWigWriter bw;
  bool cov_bw = false;
  if (!covfname.empty()) {
    if (covfname == "-" || covfname == "stdout") {
      coutf = stdout;
    } else if (bigwig) {  // tiecov.cpp:365-402
      if (!ends_with(covfname, ".bigwig")) covfname += ".bigwig";
      std::vector<std::string> names;
      std::vector<uint32_t> lens;
      for (int t = 0; t < hdr->n_targets; ++t) {
        names.push_back(hdr->target_name[t]);
        lens.push_back(hdr->target_len[t]);
      }
      std::string err;
      if (!bw.open(covfname, names, lens, err)) GError("Error creating file %s\n", covfname.c_str());
      cov_bw = true;
    } else {
      if (!ends_with(covfname, ".bedgraph")) covfname += ".bedgraph";
      coutf = fopen(covfname.c_str(), "w");
      if (!coutf) GError("Error creating file %s\n", covfname.c_str());
      fprintf(coutf, "track type=bedGraph\n");
    }
  }
  if (!jfname.empty()) {
    if (!ends_with(jfname, ".bed")) jfname += ".bed";
    joutf = fopen(jfname.c_str(), "w");
    if (!joutf) GError("Error creating file %s\n", jfname.c_str());
    fprintf(joutf, "track name=junctions\n");
  }
  if (!sfname.empty()) {
    if (!ends_with(sfname, ".bedgraph")) sfname += ".bedgraph";
    soutf = fopen(sfname.c_str(), "w");
    if (!soutf) GError("Error creating file %s\n", sfname.c_str());
    fprintf(soutf,
            "track type=bedGraph name=\"Sample Count Heatmap\" description=\"Sample Count Heatmap\" visibility=full "
            "graphType=\"heatmap\" color=200,100,0 altColor=0,100,200\n");
  }
  int num_samples = 0;
  if (soutf) {  // load_sample_info (commons.h:47-71)
    num_samples = (int)hdr->co_samples().size();
    if (num_samples == 0) GError("Error: no sample lines found in header");
  }
  // the HIP runtime takes ~0.2 s to come up: do that on a helper thread while the input is decoded
  tbk_ctx* ctx = nullptr;
  int dev = getenv("TBK_DEVICE") ? atoi(getenv("TBK_DEVICE")) : 0;
  int rc = 0;
  std::thread ctx_thread([&]() { rc = tbk_create(dev, &ctx); });
  // ---- decode to SoA (tiecov.cpp:482-485 defaults: YC absent -> 1.0, YX absent -> 1)
  tbh::BamFile* bf = samreader.file();
  size_t n = bf->n();
  std::vector<int32_t> tid(n), pos(n);
  std::vector<uint16_t> flag(n);
  std::vector<uint32_t> cig_off(n + 1, 0), cig;
  std::vector<double> yc(n, 1.0);
  std::vector<int64_t> yx(n, 1);
  std::vector<uint8_t> strand(n, '.');
  // pass 1 (serial, cheap): CIGAR offsets; pass 2 (threads): fields + one aux scan per record
  uint64_t ops = 0;
  for (size_t i = 0; i < n; ++i) {
    if (ops >= (1ull << 32)) break;
    cig_off[i] = (uint32_t)ops;
    ops += bf->rec(i).n_cigar();
  }
  if (ops >= (1ull << 32) || n >= (1ull << 32)) GError("Error: input too large for one tile\n");
  cig.resize(ops);
  uint64_t co = ops;
  {
    unsigned hw = (unsigned)tbh::cpu_budget();
    size_t nt = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, 32));
    if (n < 100000) nt = 1;
    auto work = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        tbh::RecView v = bf->rec(i);
        tid[i] = v.tid();
        pos[i] = v.pos();
        flag[i] = v.flag();
        uint32_t o = cig_off[i];
        for (uint32_t c = 0; c < v.n_cigar(); ++c) cig[o + c] = v.cigar(c);
        const uint8_t *a = v.aux_begin(), *e = v.aux_end();
        if (const uint8_t* s = tbh::aux_get(a, e, "YC")) yc[i] = tbh::aux2f(s);
        if (const uint8_t* s = tbh::aux_get(a, e, "YX")) yx[i] = tbh::aux2i(s);
        char xs = 0, ts = 0;
        if (const uint8_t* s = tbh::aux_get(a, e, "XS")) xs = (*s == 'A' || *s == 'Z') ? (char)s[1] : 0;
        if (!xs)
          if (const uint8_t* s = tbh::aux_get(a, e, "ts")) ts = (*s == 'A' || *s == 'Z') ? (char)s[1] : 0;
        char c = xs;
        if (c == 0 && (ts == '+' || ts == '-')) c = (flag[i] & 0x10) ? (ts == '+' ? '-' : '+') : ts;
        strand[i] = (uint8_t)((c == '+' || c == '-') ? c : '.');
      }
    };
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto& x : th) x.join();
  }
  cig_off[n] = (uint32_t)co;

  ctx_thread.join();
  if (rc != 0) GError("Error: cannot use GPU %d (%s); this build has no CPU coverage path\n", dev, tbk_strerror(rc));
  tbk_cov_in in;
  memset(&in, 0, sizeof(in));
  in.mem = TBK_MEM_HOST;
  in.n_records = (uint32_t)n;
  in.n_cigar_ops = (uint32_t)co;
  in.tid = tid.data();
  in.pos = pos.data();
  in.flag = flag.data();
  in.cig_off = cig_off.data();
  in.cig = cig.data();
  in.yc = yc.data();
  in.strand = strand.data();
  in.yx = yx.data();
  if (coutf || cov_bw || joutf) {
    size_t ci = (coutf || cov_bw) ? 2 * (size_t)co + 2 * n + 16 : 0, cj = joutf ? (size_t)co + 16 : 0;
    std::vector<int32_t> it(ci ? ci : 1), is(ci ? ci : 1), ie(ci ? ci : 1), jt(cj ? cj : 1), js(cj ? cj : 1), je(cj ? cj : 1);
    std::vector<double> iv(ci ? ci : 1), jv(cj ? cj : 1);
    std::vector<uint8_t> jstr(cj ? cj : 1);
    tbk_cov_out o;
    memset(&o, 0, sizeof(o));
    o.mem = TBK_MEM_HOST;
    o.cap_intervals = (uint32_t)ci;
    o.iv_tid = it.data();
    o.iv_start = is.data();
    o.iv_end = ie.data();
    o.iv_val = iv.data();
    o.cap_junctions = (uint32_t)cj;
    o.j_tid = jt.data();
    o.j_start = js.data();
    o.j_end = je.data();
    o.j_strand = jstr.data();
    o.j_val = jv.data();
    rc = tbk_coverage_tile(ctx, &in, &o);
    if (rc == TBK_EFATALOP) GError("ERROR: unknown opcode in a CIGAR string (tiecov accepts M, I, D, N, S only)\n");
    if (rc != 0) GError("Error: GPU coverage failed: %s (%s)\n", tbk_strerror(rc), tbk_last_error(ctx));
    if (coutf)  // flushCoverage, tiecov.cpp:237
      emit_lines(coutf, o.n_intervals, [&](uint32_t i, char* b, size_t cap) {
        return snprintf(b, cap, "%s\t%d\t%d\t%.3f\n", hdr->target_name[it[i]].c_str(), is[i], ie[i], iv[i]);
      });
    if (cov_bw) {  // flushCoverage(bigWigFile_t*), tiecov.cpp:243-275: the same intervals, the value as a float
      for (uint32_t i = 0; i < o.n_intervals; ++i) bw.add((uint32_t)it[i], (uint32_t)is[i], (uint32_t)ie[i], (float)iv[i]);
      std::string err;
      if (!bw.close(err)) GError("Error: writing %s failed (%s)\n", covfname.c_str(), err.c_str());
    }
    if (joutf)  // CJunc::write, tiecov.cpp:91-95
      emit_lines(joutf, o.n_junctions, [&](uint32_t i, char* b, size_t cap) {
        return snprintf(b, cap, "%s\t%d\t%d\tJUNC%08d\t%.3f\t%c\n", hdr->target_name[jt[i]].c_str(), js[i], je[i], (int)i + 1, jv[i],
                        (char)jstr[i]);
      });
  }
  if (soutf) {
    // the value of the track changes only where an M segment starts or ends: at most 2 intervals per CIGAR operation + 2 per
    // record, as for the coverage track (not one per covered base); if a call still reports TBK_E2BIG it also reports the
    // count it needs, and the call is repeated with that
    size_t cs = 2 * (size_t)co + 2 * (size_t)n + 16;
    std::vector<int32_t> st, ss, se;
    std::vector<int64_t> sc;
    std::vector<float> sh;
    tbk_sample_out so;
    for (int attempt = 0; attempt < 2; ++attempt) {
      st.resize(cs);
      ss.resize(cs);
      se.resize(cs);
      sc.resize(cs);
      sh.resize(cs);
      memset(&so, 0, sizeof(so));
      so.mem = TBK_MEM_HOST;
      so.cap_intervals = (uint32_t)cs;
      so.iv_tid = st.data();
      so.iv_start = ss.data();
      so.iv_end = se.data();
      so.iv_count = sc.data();
      so.iv_heat = sh.data();
      rc = tbk_sample_tile(ctx, &in, num_samples, &so);
      if (rc != TBK_E2BIG) break;
      cs = (size_t)so.n_intervals + 16;
    }
    if (rc == TBK_EFATALOP) GError("ERROR: unknown opcode in a CIGAR string (tiecov accepts M, I, D, N, S only)\n");
    if (rc != 0) GError("Error: GPU sample track failed: %s (%s)\n", tbk_strerror(rc), tbk_last_error(ctx));
    emit_lines(soutf, so.n_intervals, [&](uint32_t i, char* b, size_t cap) {  // flushCoverage(pair), tiecov.cpp:289
      return snprintf(b, cap, "%s\t%d\t%d\t%ld\t%f\n", hdr->target_name[st[i]].c_str(), ss[i], se[i], (long)sc[i], sh[i]);
    });
  }
  if (coutf && coutf != stdout) fclose(coutf);
  if (joutf) fclose(joutf);
  if (soutf) fclose(soutf);
  tbk_destroy(ctx);
  return 0;
}
