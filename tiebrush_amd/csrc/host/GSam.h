// GSam.h — host-side mirror of the reference's record/IO facade (same class and method names and
// argument meaning as /root/reference/src/GSam.h:23-659), implemented over the zlib-only BAM codec
// in bam.h instead of htslib.  BAM only: SAM/CRAM input is out of scope.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <memory>
#include <string>
#include <vector>

#include "bam.h"
#include "bgzf.h"

typedef unsigned int uint;
typedef tbh::BamHeader sam_hdr_t;

enum GSamFileType { GSamFile_SAM = 1, GSamFile_UBAM, GSamFile_BAM, GSamFile_CRAM };

// required_fields bits accepted for source compatibility (only meaningful for CRAM in the reference, GSam.h:380)
enum { SAM_QNAME = 1, SAM_FLAG = 2, SAM_RNAME = 4, SAM_POS = 8, SAM_MAPQ = 16, SAM_CIGAR = 32, SAM_RNEXT = 64,
       SAM_PNEXT = 128, SAM_TLEN = 256, SAM_SEQ = 512, SAM_QUAL = 1024, SAM_AUX = 2048 };

[[noreturn]] void GError(const char* fmt, ...);
void GMessage(const char* fmt, ...);

struct GSeg {
  uint start = 0, end = 0;
  GSeg(uint s = 0, uint e = 0) : start(s), end(e) {}
  int len() const { return (int)(end - start + 1); }
};

class GSamRecord : public GSeg {
  friend class GSamReader;
  friend class GSamWriter;
  tbh::BamRec b;
  sam_hdr_t* b_hdr = nullptr;
  bool hard_Clipped = false, soft_Clipped = false, has_Introns = false;

 public:
  std::vector<GSeg> exons;  // 1-based
  int clipL = 0, clipR = 0, mapped_len = 0;

  GSamRecord() {}
  GSamRecord(const tbh::RecView& v, sam_hdr_t* hdr) : b_hdr(hdr) {
    b.d.assign(v.p, v.p + v.len);
    setupCoordinates();
  }
  void init(const tbh::RecView& v, sam_hdr_t* hdr) {
    clear();
    b_hdr = hdr;
    b.d.assign(v.p, v.p + v.len);
    setupCoordinates();
  }
  void clear() {
    b.d.clear();
    exons.clear();
    start = end = 0;
    mapped_len = clipL = clipR = 0;
    hard_Clipped = soft_Clipped = has_Introns = false;
    b_hdr = nullptr;
  }
  void setupCoordinates();  // GSam.cpp:351-417
  bool isHardClipped() { return hard_Clipped; }
  bool isSoftClipped() { return soft_Clipped; }
  bool hasIntrons() { return has_Introns; }
  tbh::BamRec* get_b() { return &b; }
  tbh::RecView view() const { return b.view(); }

  int add_int_tag(const char tag[2], int64_t val) { return b.update_int(tag, val); }
  int add_double_tag(const char tag[2], double val) { return b.update_float(tag, (float)val); }
  int remove_tag(const char tag[2]) { return b.del(tag); }
  int delete_tag(const char tag[2]) { return remove_tag(tag); }

  uint32_t flags() { return view().flag(); }
  bool isUnmapped() { return (flags() & 0x4) != 0; }
  bool isMapped() { return (flags() & 0x4) == 0; }
  bool isPaired() { return (flags() & 0x1) != 0; }
  const char* name() { return view().qname(); }
  int pairOrder() {
    uint32_t f = flags();
    if (f & 0x40) return 1;
    if (f & 0x80) return 2;
    return 0;
  }
  bool revStrand() { return (flags() & 0x10) != 0; }
  char alnStrand() { return (flags() & 0x10) ? '-' : '+'; }
  bool isPrimary() { return !(flags() & 0x100); }
  const char* refName() {
    if (!b_hdr) return nullptr;
    int32_t t = view().tid();
    return t < 0 ? "*" : b_hdr->target_name[t].c_str();
  }
  int32_t refId() { return view().tid(); }
  uint8_t mapq() { return view().mapq(); }
  const uint8_t* find_tag(const char tag[2]) {
    tbh::RecView v = view();
    return tbh::aux_get(v.aux_begin(), v.aux_end(), tag);
  }
  char* tag_str(const char tag[2]) {
    const uint8_t* s = find_tag(tag);
    return s ? (char*)tbh::aux2Z(s) : nullptr;
  }
  int64_t tag_int(const char tag[2], int nfval = 0) {
    const uint8_t* s = find_tag(tag);
    return s ? tbh::aux2i(s) : nfval;
  }
  double tag_float(const char tag[2]) {
    const uint8_t* s = find_tag(tag);
    return s ? tbh::aux2f(s) : 0;
  }
  char tag_char(const char tag[2]) {
    const uint8_t* s = find_tag(tag);
    return s ? tbh::aux2A(s) : 0;
  }
  char tag_char1(const char tag[2]) {  // GSam.cpp:436-444
    const uint8_t* s = find_tag(tag);
    if (!s) return 0;
    return (*s == 'A' || *s == 'Z') ? (char)s[1] : 0;
  }
  char spliceStrand();  // GSam.cpp:464-475
  std::string cigar();  // text form
};

class GSamReader {
  std::shared_ptr<tbh::BamFile> f_;
  size_t next_ = 0;
  std::string fname_;
  int threads_ = 1;
  bool more();

 public:
  GSamReader(const char* fn, int32_t required_fields = 0, const char* cram_ref = nullptr, int inflate_threads = 4) {
    bopen(fn, required_fields, cram_ref, inflate_threads);
  }
  void bopen(const char* filename, int32_t = 0, const char* = nullptr, int inflate_threads = 4);
  void bclose() { f_.reset(); }
  sam_hdr_t* header() { return f_ ? &f_->hdr : nullptr; }
  const char* fileName() { return fname_.c_str(); }
  const char* refName(int tid) {
    if (!f_ || tid < 0 || tid >= f_->hdr.n_targets) return nullptr;
    return f_->hdr.target_name[tid].c_str();
  }
  void rewind() { next_ = 0; }
  GSamRecord* next();           // caller frees (GSam.h:506-516)
  bool next(GSamRecord& rec);   // record reuse (GSam.h:518-527)
  // batch access for the accelerated path
  tbh::BamFile* file() { return f_.get(); }
};

class GSamWriter {
  tbh::BgzfWriter w_;
  sam_hdr_t hdr_;

 public:
  GSamWriter(const char* fname, sam_hdr_t* bh, GSamFileType ftype = GSamFile_BAM);
  ~GSamWriter() { w_.close(); }
  sam_hdr_t* header() { return &hdr_; }
  void write(GSamRecord* brec);
  void write_raw(const tbh::BamRec& r);
  // a run of already framed records (each: little-endian block_size, then the record)
  void write_framed(const uint8_t* p, size_t n);
  // the same run already deflated into BGZF members by the caller's worker threads (tbh::bgzf_deflate_members)
  void write_members(const uint8_t* z, size_t n);
  int level() const { return w_.level(); }
};
