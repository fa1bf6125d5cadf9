// tb_cpu_e2e — files -> files on the host cores.  TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/).
//
// The reference's command line restated on the CPU for the end-to-end baseline (SURVEY.md §8d "end-to-end ... with the reference CPU
// path beside it"): the repo's host codec reads the inputs the way the reference's main loop does — TInputFiles::start() /
// TInputFiles::next() -> GSamReader::next() (one GSamRecord per record: BGZF inflate, BAM parse, setupCoordinates; /root/reference/src
// /tiebrush.cpp:557-601, GSam.h:506-516, tmerge.cpp:319-344) —, the oracle (tb_oracle.c: the literal restatement of addPData /
// flushPData / GSegList) collapses what passed, and every representative is tagged and written through GSamWriter::write
// (tiebrush.cpp:506-525, GSam.h:648-653).  Nothing here touches the GPU and nothing of the product path calls it.
//
// Threads: whatever the process is allowed to use — the host codec sizes its pools by the CPU affinity / quota (tbh::cpu_budget), so a
// caller that wants the reference's single-threaded figure pins the process to one core (bench.py does: os.sched_setaffinity).
//
//   tb_cpu_e2e [-L | -P | -E] [-S] [--keep-secondary] [-A] [-N n] [-Q q] -o OUT.bam IN1.bam IN2.bam ...
// prints one line on stderr: records in / passed / groups out and the three phase times.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <vector>

#include "../tiebrush_amd/csrc/host/GSam.h"
#include "../tiebrush_amd/csrc/host/tmerge.h"
#include "tb_oracle.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct PerFile {
  std::vector<int32_t> tid, pos, nh;
  std::vector<uint16_t> flag;
  std::vector<uint8_t> mapq, strand;
  std::vector<uint32_t> ncig, cig;
  std::vector<double> yc;
  std::vector<int64_t> yx, yd;
  std::vector<uint32_t> qlen;
  std::vector<uint8_t> qn;
  std::vector<GSamRecord*> rec;
};

int main(int argc, char** argv) {
  tbo_opts o;
  tbo_opts_default(&o);
  const char* out_path = nullptr;
  std::vector<const char*> inputs;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-o" && i + 1 < argc) out_path = argv[++i];
    else if (a == "-L" || a == "--full") o.strategy = TBO_STRAT_FULL;
    else if (a == "-P" || a == "--clip") o.strategy = TBO_STRAT_CLIP;
    else if (a == "-E" || a == "--exon") o.strategy = TBO_STRAT_EXON;
    else if (a == "-S" || a == "--keep-supp") o.keep_supplementary = 1;
    else if (a == "--keep-secondary") o.keep_secondary = 1;
    else if (a == "-A" || a == "--collapse-same") o.collapse_same = 1;
    else if (a == "-N" && i + 1 < argc) o.max_nh = atoi(argv[++i]);
    else if (a == "-Q" && i + 1 < argc) o.min_qual = atoi(argv[++i]);
    else if (!a.empty() && a[0] == '-') {
      fprintf(stderr, "tb_cpu_e2e: unknown option %s\n", a.c_str());
      return 1;
    } else inputs.push_back(argv[i]);
  }
  if (!out_path || inputs.empty()) {
    fprintf(stderr, "usage: tb_cpu_e2e [-L|-P|-E] [-S] [--keep-secondary] [-A] [-N n] [-Q q] -o OUT.bam IN.bam ...\n");
    return 1;
  }
  const double t0 = now_s();
  TInputFiles in;
  in.setup("0.0.6", argc, argv);
  for (const char* p : inputs) in.addFile(p);
  const int k = in.start();
  std::vector<PerFile> F((size_t)k);
  std::vector<uint8_t> tbm((size_t)k, 0);
  bool any_tbm = false;
  for (int f = 0; f < k; ++f) {
    tbm[(size_t)f] = in.freaders[(size_t)f]->tbMerged ? 1 : 0;
    any_tbm |= tbm[(size_t)f] != 0;
  }
  // the reference's loop: one record at a time in merge order (tiebrush.cpp:569); the oracle takes the files' own order, so the
  // fields are filed under the record's input
  size_t n_in = 0;
  while (TInputRecord* ir = in.next()) {
    GSamRecord& b = *ir->brec;
    PerFile& P = F[(size_t)ir->fidx];
    const tbh::RecView v = b.view();
    P.tid.push_back(v.tid());
    P.pos.push_back(v.pos());
    P.flag.push_back(v.flag());
    P.mapq.push_back(v.mapq());
    P.strand.push_back((uint8_t)b.spliceStrand());
    P.nh.push_back(b.find_tag("NH") ? (int32_t)b.tag_int("NH") : TBO_NH_ABSENT);
    const uint32_t nc = v.n_cigar();
    P.ncig.push_back(nc);
    for (uint32_t c = 0; c < nc; ++c) P.cig.push_back(v.cigar(c));
    if (any_tbm) {  // carried tags of TieBrush-merged inputs (tiebrush.cpp:389-395)
      P.yc.push_back(b.find_tag("YC") ? b.tag_float("YC") : 0.0);
      P.yx.push_back(b.find_tag("YX") ? b.tag_int("YX") : 1);
      P.yd.push_back(b.find_tag("YD") ? b.tag_int("YD") : 0);
    }
    if (o.collapse_same) {  // -A compares the names themselves (tiebrush.cpp:422-424)
      const char* q = b.name();
      const uint32_t l = (uint32_t)strlen(q);
      P.qlen.push_back(l);
      P.qn.insert(P.qn.end(), q, q + l);
    }
    P.rec.push_back(ir->brec);
    ir->disown();
    ++n_in;
  }
  const double t1 = now_s();
  // file-major arrays for the oracle
  std::vector<uint32_t> file_off((size_t)k + 1, 0), cig_off(n_in + 1, 0), cig;
  std::vector<int32_t> tid, pos, nh;
  std::vector<uint16_t> flag;
  std::vector<uint8_t> mapq, strand;
  std::vector<double> yc_in;
  std::vector<int64_t> yx_in, yd_in;
  std::vector<uint32_t> qn_off(1, 0);
  std::vector<uint8_t> qn;
  std::vector<GSamRecord*> recs;
  tid.reserve(n_in), pos.reserve(n_in), nh.reserve(n_in), flag.reserve(n_in), mapq.reserve(n_in), strand.reserve(n_in), recs.reserve(n_in);
  size_t at = 0;
  for (int f = 0; f < k; ++f) {
    PerFile& P = F[(size_t)f];
    tid.insert(tid.end(), P.tid.begin(), P.tid.end());
    pos.insert(pos.end(), P.pos.begin(), P.pos.end());
    nh.insert(nh.end(), P.nh.begin(), P.nh.end());
    flag.insert(flag.end(), P.flag.begin(), P.flag.end());
    mapq.insert(mapq.end(), P.mapq.begin(), P.mapq.end());
    strand.insert(strand.end(), P.strand.begin(), P.strand.end());
    cig.insert(cig.end(), P.cig.begin(), P.cig.end());
    if (any_tbm) {
      yc_in.insert(yc_in.end(), P.yc.begin(), P.yc.end());
      yx_in.insert(yx_in.end(), P.yx.begin(), P.yx.end());
      yd_in.insert(yd_in.end(), P.yd.begin(), P.yd.end());
    }
    qn.insert(qn.end(), P.qn.begin(), P.qn.end());
    for (uint32_t l : P.qlen) qn_off.push_back(qn_off.back() + l);
    recs.insert(recs.end(), P.rec.begin(), P.rec.end());
    for (uint32_t c : P.ncig) {
      cig_off[at + 1] = cig_off[at] + c;
      ++at;
    }
    file_off[(size_t)f + 1] = (uint32_t)at;
    P = PerFile();
  }
  tbo_in I;
  memset(&I, 0, sizeof(I));
  I.n_files = (uint32_t)k;
  I.n_records = (uint32_t)n_in;
  I.file_off = file_off.data();
  I.tbmerged = tbm.data();
  I.tid = tid.data();
  I.pos = pos.data();
  I.flag = flag.data();
  I.mapq = mapq.data();
  I.strand = strand.data();
  I.nh = nh.data();
  I.cig_off = cig_off.data();
  I.cig = cig.data();
  if (any_tbm) {
    I.yc_in = yc_in.data();
    I.yx_in = yx_in.data();
    I.yd_in = yd_in.data();
  }
  if (o.collapse_same) {
    I.qn_off = qn_off.data();
    I.qn = qn.data();
  }
  std::vector<uint32_t> rep(n_in ? n_in : 1);
  std::vector<double> yc(n_in ? n_in : 1);
  std::vector<int64_t> yx(n_in ? n_in : 1);
  std::vector<int32_t> yd(n_in ? n_in : 1);
  tbo_groups G;
  memset(&G, 0, sizeof(G));
  G.cap = (uint32_t)rep.size();
  G.rep = rep.data();
  G.yc = yc.data();
  G.yx = yx.data();
  G.yd = yd.data();
  const int rc = tbo_collapse(&o, &I, &G);
  if (rc != TBO_OK) {
    fprintf(stderr, "tb_cpu_e2e: the oracle refused the input (%d)\n", rc);
    return 1;
  }
  const double t2 = now_s();
  {
    GSamWriter out(out_path, in.header(), GSamFile_BAM);
    for (uint32_t g = 0; g < G.n_groups; ++g) {  // flushPData's tagging (tiebrush.cpp:506-525)
      GSamRecord* r = recs[rep[g]];
      r->add_double_tag("YC", yc[g]);
      r->add_int_tag("YX", yx[g]);
      if (yd[g] > 0)
        r->add_int_tag("YD", yd[g]);
      else
        r->remove_tag("YD");
      out.write(r);
    }
  }
  const double t3 = now_s();
  const double p = G.n_passed ? 100.0 - (double)G.n_groups * 100.0 / (double)G.n_passed : 0.0;
  fprintf(stderr, "%ld input records written as %ld (%.2f%% reduction)\n", (long)G.n_passed, (long)G.n_groups, p);
  fprintf(stderr, "tb_cpu_e2e: records_in %zu passed %u groups %u  read+parse %.3f s  collapse %.3f s  tag+deflate+write %.3f s  total %.3f s\n", n_in,
          G.n_passed, G.n_groups, t1 - t0, t2 - t1, t3 - t2, t3 - t0);
  fflush(stderr);
  _exit(0);  // (tens of millions of records: the process ends without walking their destructors, as the clock should)
}
