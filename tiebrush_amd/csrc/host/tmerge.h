// tmerge.h — host-side mirror of the reference's TInputFiles / TInputRecord surface
// (/root/reference/src/tmerge.h:13-117, tmerge.cpp:48-350): same method names and meaning.
// `next()` still hands out one record at a time in the reference's merge order for callers that
// iterate; the accelerated tools do not iterate — they call `load_tile()` and hand the whole tile
// to tbk_collapse_tile (include/tbk.h).
#pragma once
#include <string>
#include <vector>

#include "../../../include/tbk.h"
#include "GSam.h"

struct TSamReader {
  std::string fname;
  GSamReader* samreader = nullptr;
  bool tbMerged = false;
  explicit TSamReader(const char* fn = nullptr) : fname(fn ? fn : "") {}
  ~TSamReader() { delete samreader; }
};

struct TInputRecord {
  GSamRecord* brec;
  int fidx;
  bool tbMerged;
  // "decreasing location sort" of the reference (tmerge.h:28-50): a < b means a pops later
  bool operator<(TInputRecord& o);
  void disown() { brec = nullptr; }
  TInputRecord(GSamRecord* b = nullptr, int i = 0, bool tb = false) : brec(b), fidx(i), tbMerged(tb) {}
  ~TInputRecord() { delete brec; }
};

// Structure-of-arrays tile of every input file (file-major), the layout of tbk_soa_in.
struct TbkTile {
  std::vector<uint32_t> file_off;
  std::vector<uint8_t> tbmerged;
  std::vector<int32_t> tid, pos, nh;
  std::vector<uint16_t> flag;
  std::vector<uint8_t> mapq, strand;
  std::vector<uint32_t> cig_off, cig;
  std::vector<double> yc_in;
  std::vector<int64_t> yx_in, yd_in;
  std::vector<uint32_t> md_off;
  std::vector<uint8_t> md, md_has;
  std::vector<uint64_t> qname_hash;
  std::vector<uint32_t> qname_off;  // -A: the names themselves (CSR, no NUL): the device confirms equal hashes on the bytes
  std::vector<uint8_t> qname;
  tbk_soa_in view() const;
  size_t n() const { return tid.size(); }
};

struct TInputFiles {
 protected:
  TInputRecord* crec = nullptr;
  sam_hdr_t* mHdr = nullptr;
  std::string pg_ver, pg_args;
  std::vector<size_t> cursor_;  // next record of each file for next()

 public:
  std::vector<TSamReader*> freaders;
  std::vector<TInputRecord*> recs;  // one head record per file, kept sorted like the reference's GList
  std::string headerfilename;
  bool headerfiletbMerged = false;

  ~TInputFiles();
  sam_hdr_t* header() { return mHdr; }
  void setup(const char* ver, int argc, char** argv);
  void addFile(const char* fn);
  bool addSam(GSamReader* r, int fidx);
  int count() { return (int)freaders.size(); }
  int start();            // opens every input, merges headers, primes next()
  TInputRecord* next();   // valid until the following next() (tmerge.cpp:331-344)
  void stop();
  // accelerated path.  Everything the collapse does is confined to one reference sequence (buckets, groups and tiecov
  // bundles never span a tid; the YD lists are reset at every tid change, tiebrush.cpp:586-589), so the input can be cut
  // into tiles of whole reference sequences without changing a single output byte.  plan_tiles() groups consecutive tids
  // up to `max_records` per tile; load_tile() decodes one tile into SoA form (multi-threaded aux scan).
  struct TilePlan {
    int32_t tid_lo = 0, tid_hi = 0;  // [tid_lo, tid_hi)
    std::vector<size_t> lo, hi;      // per input file: record range
    size_t n = 0;
  };
  std::vector<TilePlan> plan_tiles(size_t max_records);   // (reads every input to its end first: small inputs, tests)
  // Streaming form: the next tile of about target_records records, cut at a GLOBAL bundle boundary — a coordinate no read
  // of any input reaches across — or at a reference change: buckets and groups are whole, the per-sample YD lists provably
  // clear at such a point (every node ends before the next read starts, tiebrush.cpp:230-241), so tiles collapse
  // independently and the output is byte-identical to the one-tile run.  Only the windows of the inputs up to the cut are
  // in memory; release_tile() drops them.  Returns false when every input is exhausted.
  bool next_tile(TilePlan& plan, size_t target_records, int threads);
  void release_tile(const TilePlan& plan);
  void load_tile(TbkTile& t, bool want_md, bool want_qname_hash, int threads, const TilePlan* plan = nullptr);
  tbh::RecView record(uint32_t global_index) const;  // raw record behind tile index i
  std::vector<uint32_t> tile_off_;                   // file_off of the last load_tile()
  std::vector<size_t> tile_lo_;                      // first record of each file inside the last tile
};

uint64_t tbh_qname_hash(const char* name, int pair_order);
std::string tbh_realpath(const std::string& p);
