// tbh_tool — small test driver for the host-side codec and API mirror (used by tests/, CPU only):
//   cat IN.bam OUT.bam            decode with GSamReader, re-encode with GSamWriter
//   mergeorder IN1.bam IN2.bam..  print "fidx idx" of every record in TInputFiles::next() order
//   soa OUTDIR IN1.bam ...        dump the SoA tile arrays (one raw little-endian file per array)
//   tags IN.bam OUT.bam SPEC...   apply tag edits to every record: YC=f:2.5  YX=i:255  YD=i:0  YD=del
//   mkbam SOADIR PREFIX [LEVEL [THREADS]]   encode the raw SoA arrays of a synthetic tile (file_off, tid, pos, flag, mapq, strand,
//                                 nh, cig_off, cig as little-endian files + header.txt) as PREFIX<f>.bam, one per input file: the
//                                 records tiebrush_amd.synth.write_bams writes (SEQ '*', QNAME r<f>_<i>, NH:C / XS:A), files in parallel
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "GSam.h"
#include "bgzf.h"
#include "fastload.h"
#include "tmerge.h"
#include "bigwig.h"

template <class T>
static bool slurp(const std::string& dir, const char* name, std::vector<T>& v) {
  FILE* f = fopen((dir + "/" + name).c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)sz / sizeof(T));
  const bool ok = v.empty() || fread(v.data(), sizeof(T), v.size(), f) == v.size();
  fclose(f);
  return ok;
}
// UCSC binning scheme (SAM spec 5.3), as htslib's hts_reg2bin(beg, end, 14, 5)
static uint16_t reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}
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
  if (cmd == "fastsoa" && argc >= 4) {  // fastsoa OUTDIR IN1.bam ...: the tile of the whole-input loader (fastload.cpp), as `soa` dumps it
    std::vector<std::string> paths;
    std::vector<uint8_t> tb;
    for (int i = 3; i < argc; ++i) {
      paths.push_back(argv[i]);
      GSamReader rd(argv[i]);
      tb.push_back(rd.header()->is_tiebrush() ? 1 : 0);
    }
    tbh::FastTile t;
    bool fits = false;
    std::string err;
    if (!tbh::fast_load(paths, tb, argc > 3 ? 3 : 1, (size_t)1 << 40, t, &fits, err) || !fits) {
      fprintf(stderr, "%s\n", err.c_str());
      return 1;
    }
    fprintf(stderr, "fastsoa: %zu of %zu inputs indexed by their members' own walks; ms read %.1f inflate %.1f index %.1f SoA %.1f\n", t.n_fused, t.in.size(), t.ms_read,
            t.ms_inflate, t.ms_index, t.ms_soa);
    const std::string d = argv[2];
    auto dumpp = [&](const char* name, const void* p, size_t bytes) {
      FILE* f = fopen((d + "/" + name).c_str(), "wb");
      if (!f) GError("cannot write %s/%s\n", d.c_str(), name);
      if (bytes) fwrite(p, 1, bytes, f);
      fclose(f);
    };
    dump(d, "file_off", t.file_off);
    dump(d, "tbmerged", t.tbmerged);
    dumpp("tid", t.tid, t.n * 4);
    dumpp("pos", t.pos, t.n * 4);
    dumpp("flag", t.flag, t.n * 2);
    dumpp("mapq", t.mapq, t.n);
    dumpp("strand", t.strand, t.n);
    dumpp("nh", t.nh, t.n * 4);
    dumpp("cig_off", t.cig_off, (t.n + 1) * 4);
    dumpp("cig", t.cig, t.n_cig * 4);
    dumpp("yc_in", t.yc_in, t.yc_in ? t.n * 8 : 0);
    dumpp("yx_in", t.yx_in, t.yx_in ? t.n * 8 : 0);
    dumpp("yd_in", t.yd_in, t.yd_in ? t.n * 8 : 0);
    // the raw records behind a few tile indices: first, last, and one from the middle of every file
    std::vector<uint8_t> recs;
    for (size_t f = 0; f + 1 < t.file_off.size(); ++f)
      for (uint32_t g : {t.file_off[f], (t.file_off[f] + t.file_off[f + 1]) / 2, t.file_off[f + 1] - 1}) {
        if (t.file_off[f] == t.file_off[f + 1]) continue;
        uint32_t len;
        const uint8_t* p = t.record(g, &len);
        recs.insert(recs.end(), (const uint8_t*)&g, (const uint8_t*)&g + 4);
        recs.insert(recs.end(), (const uint8_t*)&len, (const uint8_t*)&len + 4);
        recs.insert(recs.end(), p, p + len);
      }
    dump(d, "probe_records", recs);
    return 0;
  }
  if (cmd == "mkbam" && argc >= 4) {
    const std::string d = argv[2], prefix = argv[3];
    const int level = argc > 4 ? atoi(argv[4]) : 1;
    int threads = argc > 5 && atoi(argv[5]) > 0 ? atoi(argv[5]) : tbh::cpu_budget();
    // "seq": every record carries SEQ / QUAL of its query length and the aux tags of an aligner's output, about 250 bytes per
    // 100-bp read like the reference's fixtures (test/t1/t1s0.bam) — what BGZF inflate / deflate and the tagging really move
    const bool with_seq = argc > 6 && strcmp(argv[6], "seq") == 0;
    std::vector<uint32_t> file_off, cig_off, cig;
    std::vector<int32_t> tid, pos, nh;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq, strand;
    if (!slurp(d, "file_off", file_off) || !slurp(d, "tid", tid) || !slurp(d, "pos", pos) || !slurp(d, "flag", flag) || !slurp(d, "mapq", mapq) ||
        !slurp(d, "strand", strand) || !slurp(d, "nh", nh) || !slurp(d, "cig_off", cig_off) || !slurp(d, "cig", cig) || file_off.size() < 2)
      GError("mkbam: cannot read the SoA arrays in %s\n", d.c_str());
    std::string text;
    {
      FILE* f = fopen((d + "/header.txt").c_str(), "r");
      if (!f) GError("mkbam: no header.txt in %s\n", d.c_str());
      char buf[4096];
      size_t r;
      while ((r = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, r);
      fclose(f);
    }
    tbh::BamHeader hdr;
    hdr.text = text;
    for (const std::string& l : hdr.lines()) {
      if (l.compare(0, 3, "@SQ") != 0) continue;
      const size_t a = l.find("SN:"), b = l.find("LN:");
      if (a == std::string::npos || b == std::string::npos) continue;
      hdr.target_name.push_back(l.substr(a + 3, l.find('\t', a) - a - 3));
      hdr.target_len.push_back((uint32_t)atoll(l.c_str() + b + 3));
    }
    hdr.n_targets = (int32_t)hdr.target_name.size();
    const uint32_t k = (uint32_t)file_off.size() - 1;
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > k) threads = (int)k;
    std::atomic<uint32_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
      std::vector<uint8_t> buf;
      for (;;) {
        const uint32_t f = next.fetch_add(1);
        if (f >= k) return;
        tbh::BgzfWriter w;
        if (!w.open(prefix + std::to_string(f) + ".bam", level, 1)) {
          failed = 1;
          return;
        }
        // blocks as htslib cuts them: the header in blocks of its own (bam_hdr_write ends with a flush), and a new block whenever
        // the next record does not fit into the current one (bgzf_flush_try in bam_write1) — every block begins with a record
        std::vector<uint8_t> z;
        auto flush_block = [&]() {
          if (buf.empty()) return;
          z.clear();
          if (!tbh::bgzf_deflate_members(buf.data(), buf.size(), level, z) || !w.write_members(z.data(), z.size())) failed = 1;
          buf.clear();
        };
        buf.clear();
        hdr.serialize(buf);
        flush_block();
        for (uint32_t i = file_off[f]; i < file_off[f + 1]; ++i) {
          char nm[40];
          const int nl = snprintf(nm, sizeof(nm), "r%u_%u", f, i - file_off[f]) + 1;
          const uint32_t c0 = cig_off[i], nc = cig_off[i + 1] - c0;
          int64_t rl = 0, ql = 0;
          for (uint32_t q = 0; q < nc; ++q) {
            const uint32_t op = cig[c0 + q] & 0xF;
            if ((0x18Du >> op) & 1u) rl += cig[c0 + q] >> 4;
            if ((0x193u >> op) & 1u) ql += cig[c0 + q] >> 4;  // M I S = X consume the query
          }
          const bool has_nh = nh[i] != INT32_MIN, has_xs = strand[i] == '+' || strand[i] == '-';
          const uint32_t lseq = with_seq ? (uint32_t)ql : 0u;
          // AS XN XM XO XG NM as C-typed zeros, MD:Z:<ql>, YT:Z:UU — the tag set of the fixtures' aligner
          char md[16];
          const int mdl = with_seq ? snprintf(md, sizeof(md), "%lld", (long long)ql) + 1 : 0;
          const uint32_t extra = with_seq ? 6u * 4u + 3u + (uint32_t)mdl + 6u : 0u;
          const uint32_t body = 32 + (uint32_t)nl + 4 * nc + (lseq + 1) / 2 + lseq + extra + (has_nh ? 4u : 0u) + (has_xs ? 4u : 0u);
          if (buf.size() + 4 + (size_t)body > 0xff00) flush_block();
          const size_t o = buf.size();
          buf.resize(o + 4 + body);
          uint8_t* p = buf.data() + o;
          auto w32 = [&](size_t at, uint32_t v) { memcpy(p + at, &v, 4); };
          auto w16 = [&](size_t at, uint16_t v) { memcpy(p + at, &v, 2); };
          w32(0, body);
          w32(4, (uint32_t)tid[i]);
          w32(8, (uint32_t)pos[i]);
          p[12] = (uint8_t)nl;
          p[13] = mapq[i];
          w16(14, pos[i] >= 0 ? reg2bin(pos[i], pos[i] + (rl > 1 ? rl : 1)) : (uint16_t)4680);
          w16(16, (uint16_t)nc);
          w16(18, flag[i]);
          w32(20, lseq);                // l_seq
          w32(24, 0xFFFFFFFFu);         // next refID
          w32(28, 0xFFFFFFFFu);         // next pos
          w32(32, 0);                   // tlen
          memcpy(p + 36, nm, (size_t)nl);
          memcpy(p + 36 + nl, cig.data() + c0, 4 * (size_t)nc);
          uint8_t* a = p + 36 + nl + 4 * nc;
          if (with_seq) {
            uint64_t x = ((uint64_t)f << 40) ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull;  // (deterministic per record)
            auto rnd = [&]() {
              x ^= x << 13, x ^= x >> 7, x ^= x << 17;
              return x;
            };
            static const uint8_t nt[4] = {1, 2, 4, 8};  // A C G T
            for (uint32_t b = 0; b < (lseq + 1) / 2; ++b) {
              const uint64_t r = rnd();
              a[b] = (uint8_t)((nt[r & 3] << 4) | ((2 * b + 1 < lseq) ? nt[(r >> 2) & 3] : 0));
            }
            a += (lseq + 1) / 2;
            for (uint32_t b = 0; b < lseq; ++b) {  // Phred 40 with a dip every so often (a quality string deflates well, not to nothing)
              const uint64_t r = rnd();
              a[b] = (r & 15) == 0 ? (uint8_t)(20 + ((r >> 4) & 15)) : (uint8_t)40;
            }
            a += lseq;
            static const char* zt[6] = {"AS", "XN", "XM", "XO", "XG", "NM"};
            for (int z = 0; z < 6; ++z) {
              a[0] = (uint8_t)zt[z][0], a[1] = (uint8_t)zt[z][1], a[2] = 'C', a[3] = 0;
              a += 4;
            }
            a[0] = 'M', a[1] = 'D', a[2] = 'Z';
            memcpy(a + 3, md, (size_t)mdl);
            a += 3 + mdl;
            memcpy(a, "YTZUU", 6);
            a += 6;
          }
          if (has_nh) {
            a[0] = 'N', a[1] = 'H', a[2] = 'C', a[3] = (uint8_t)nh[i];
            a += 4;
          }
          if (has_xs) a[0] = 'X', a[1] = 'S', a[2] = 'A', a[3] = strand[i];
        }
        flush_block();
        if (!w.close()) failed = 1;
      }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
    return failed ? 1 : 0;
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
  fprintf(stderr, "usage: tbh_tool cat|mergeorder|soa|fastsoa|mkbam|tiles|tags|bedgraph2bw ...\n");
  return 2;
}
