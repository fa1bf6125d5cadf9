// tbh_tool — small test driver for the host-side codec and API mirror (used by tests/, CPU only):
//   cat IN.bam OUT.bam            decode with GSamReader, re-encode with GSamWriter
//   mergeorder IN1.bam IN2.bam..  print "fidx idx" of every record in TInputFiles::next() order
//   soa OUTDIR IN1.bam ...        dump the SoA tile arrays (one raw little-endian file per array)
//   tags IN.bam OUT.bam SPEC...   apply tag edits to every record: YC=f:2.5  YX=i:255  YD=i:0  YD=del
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "GSam.h"
#include "tmerge.h"
#include "bigwig.h"

template <class T>
static void dump(const std::string& dir, const char* name, const std::vector<T>& v) {
  FILE* f = fopen((dir + "/" + name).c_str(), "wb");
  if (!f) GError("cannot write %s/%s\n", dir.c_str(), name);
  if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::string cmd = argv[1];
  if (cmd == "cat" && argc == 4) {
    GSamReader rd(argv[2]);
    GSamWriter wr(argv[3], rd.header());
    GSamRecord r;
    while (rd.next(r)) wr.write(&r);
    return 0;
  }
  if (cmd == "mergeorder" && argc >= 3) {
    TInputFiles in;
    in.setup("test", 0, nullptr);
    for (int i = 2; i < argc; ++i) in.addFile(argv[i]);
    int k = in.start();
    std::vector<uint32_t> idx(k, 0);
    while (TInputRecord* r = in.next()) printf("%d %u\n", r->fidx, idx[r->fidx]++);
    return 0;
  }
  if (cmd == "soa" && argc >= 4) {
    TInputFiles in;
    in.setup("test", 0, nullptr);
    for (int i = 3; i < argc; ++i) in.addFile(argv[i]);
    in.start();
    TbkTile t;
    in.load_tile(t, true, true, 4);
    std::string d = argv[2];
    dump(d, "file_off", t.file_off);
    dump(d, "tbmerged", t.tbmerged);
    dump(d, "tid", t.tid);
    dump(d, "pos", t.pos);
    dump(d, "flag", t.flag);
    dump(d, "mapq", t.mapq);
    dump(d, "strand", t.strand);
    dump(d, "nh", t.nh);
    dump(d, "cig_off", t.cig_off);
    dump(d, "cig", t.cig);
    dump(d, "yc_in", t.yc_in);
    dump(d, "yx_in", t.yx_in);
    dump(d, "yd_in", t.yd_in);
    dump(d, "md_off", t.md_off);
    dump(d, "md", t.md);
    dump(d, "md_has", t.md_has);
    dump(d, "qname_hash", t.qname_hash);
    dump(d, "qname_off", t.qname_off);
    dump(d, "qname", t.qname);
    FILE* f = fopen((d + "/header.txt").c_str(), "w");
    fputs(in.header()->text.c_str(), f);
    fclose(f);
    return 0;
  }
  if (cmd == "tiles" && argc >= 4) {  // tiles <target records> files...: one line per streamed tile = records taken from every input
    TInputFiles in;
    in.setup("test", 0, nullptr);
    for (int i = 3; i < argc; ++i) in.addFile(argv[i]);
    in.start();
    TInputFiles::TilePlan plan;
    while (in.next_tile(plan, (size_t)atoll(argv[2]), 4)) {
      size_t win = 0;
      for (auto fr : in.freaders) win += fr->samreader->file()->n();
      printf("%zu", win);  // records resident in the windows when the tile was cut
      for (size_t f = 0; f < plan.hi.size(); ++f) printf(" %zu", plan.hi[f]);
      printf("\n");
      in.release_tile(plan);
    }
    return 0;
  }
  if (cmd == "tags" && argc >= 5) {
    GSamReader rd(argv[2]);
    GSamWriter wr(argv[3], rd.header());
    GSamRecord r;
    while (rd.next(r)) {
      for (int i = 4; i < argc; ++i) {
        std::string s = argv[i];
        char tag[2] = {s[0], s[1]};
        std::string v = s.substr(3);
        if (v == "del")
          r.remove_tag(tag);
        else if (v[0] == 'f')
          r.add_double_tag(tag, atof(v.c_str() + 2));
        else
          r.add_int_tag(tag, atoll(v.c_str() + 2));
      }
      wr.write(&r);
    }
    return 0;
  }
  if (cmd == "bedgraph2bw" && argc == 5) {  // <alignment file: the chromosome list> <in.bedgraph> <out.bigwig>
    GSamReader rd(argv[2]);
    sam_hdr_t* hdr = rd.header();
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    for (int t = 0; t < hdr->n_targets; ++t) {
      names.push_back(hdr->target_name[t]);
      lens.push_back(hdr->target_len[t]);
    }
    tbh::BigWigWriter bw;
    std::string err;
    if (!bw.open(argv[4], names, lens, err)) {
      fprintf(stderr, "%s\n", err.c_str());
      return 1;
    }
    FILE* f = fopen(argv[3], "r");
    if (!f) return 1;
    char line[4096], chrom[2048];
    while (fgets(line, sizeof(line), f)) {
      unsigned long a, b;
      double v;
      if (sscanf(line, "%2047s %lu %lu %lf", chrom, &a, &b, &v) != 4) continue;  // (track line)
      int tid = hdr->name2tid(chrom);
      if (tid < 0) return 1;
      bw.add((uint32_t)tid, (uint32_t)a, (uint32_t)b, (float)v);
    }
    fclose(f);
    if (!bw.close(err)) {
      fprintf(stderr, "%s\n", err.c_str());
      return 1;
    }
    return 0;
  }
  fprintf(stderr, "usage: tbh_tool cat|mergeorder|soa|tiles|tags|bedgraph2bw ...\n");
  return 2;
}
