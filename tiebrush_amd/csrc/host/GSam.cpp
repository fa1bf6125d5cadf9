#include "GSam.h"
#include "sam.h"

#include <stdlib.h>

#include <algorithm>

#include <stdarg.h>
#include <string.h>

void GError(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  exit(1);
}
void GMessage(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
}

// literal walk of the reference's coordinate setup (GSam.cpp:351-417)
void GSamRecord::setupCoordinates() {
  exons.clear();
  start = end = 0;
  mapped_len = clipL = clipR = 0;
  hard_Clipped = soft_Clipped = has_Introns = false;
  if (b.d.empty()) return;
  tbh::RecView v = view();
  if (v.flag() & 0x4) return;
  int l = 0;
  int pos = v.pos();
  start = (uint)(pos + 1);
  int exstart = pos;
  bool intron = false, ins = false;
  uint32_t n = v.n_cigar();
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t c = v.cigar(i);
    uint32_t op = c & 0xF, len = c >> 4;
    switch (op) {
      case 7: case 8: case 0: case 2:
        l += (int)len;
        intron = false;
        ins = false;
        break;
      case 3:
        if (!ins || !intron) {
          GSeg ex((uint)(exstart + 1), (uint)(pos + l));
          exons.push_back(ex);
          mapped_len += ex.len();
        }
        has_Introns = true;
        l += (int)len;
        exstart = pos + l;
        intron = true;
        break;
      case 4:
        soft_Clipped = true;
        if (l) clipR = (int)len; else clipL = (int)len;
        intron = false;
        ins = false;
        break;
      case 5:
        hard_Clipped = true;
        intron = false;
        ins = false;
        break;
      case 1:
        ins = true;
        break;
      case 6:
        break;
      default:
        fprintf(stderr, "Unhandled CIGAR operation %d:%d\n", (int)op, (int)len);
    }
  }
  GSeg ex((uint)(exstart + 1), (uint)(pos + l));
  exons.push_back(ex);
  mapped_len += ex.len();
  end = ex.end;
}

char GSamRecord::spliceStrand() {
  char c = tag_char1("XS");
  if (c == 0) {
    char m = tag_char1("ts");
    if (m == '+' || m == '-') {
      if (flags() & 0x10) c = (m == '+') ? '-' : '+';
      else c = m;
    }
  }
  return (c == '+' || c == '-') ? c : '.';
}

std::string GSamRecord::cigar() {
  tbh::RecView v = view();
  if (v.n_cigar() == 0) return "*";
  std::string s;
  for (uint32_t i = 0; i < v.n_cigar(); ++i) {
    uint32_t c = v.cigar(i);
    s += std::to_string(c >> 4);
    s += "MIDNSHP=XB"[c & 0xF];
  }
  return s;
}

void GSamReader::bopen(const char* filename, int32_t, const char*, int inflate_threads) {
  fname_ = filename;
  f_ = std::make_shared<tbh::BamFile>();
  std::string err;
  if (tbh::cram_probe(fname_)) GError("Error: could not open alignment file %s (CRAM input is not supported by this build)\n", filename);
  if (!tbh::bgzf_probe(fname_) && !tbh::sam_probe(fname_))
    GError("Error: could not open alignment file %s (neither BAM nor SAM)\n", filename);
  threads_ = inflate_threads < 1 ? 1 : inflate_threads;
  // header only: records are inflated on demand (GSamReader::next / TInputFiles::next_tile), the window slides
  if (!f_->open(fname_, err, threads_)) GError("Error: could not open alignment file %s (%s)\n", filename, err.c_str());
  next_ = 0;
}

// is there a record at next_?  Slides the window: what was handed out is dropped once it is a large share of the window
bool GSamReader::more() {
  if (next_ < f_->n()) return true;
  if (next_ > 0) {
    f_->consume(next_);
    next_ = 0;
  }
  std::string err;
  if (!f_->at_eof() && !f_->fill(1, err, threads_)) GError("Error: reading %s failed (%s)\n", fname_.c_str(), err.c_str());
  return next_ < f_->n();
}

GSamRecord* GSamReader::next() {
  if (!f_) GError("Warning: GSamReader::next() called with no open file.\n");
  if (!more()) return nullptr;
  return new GSamRecord(f_->rec(next_++), &f_->hdr);
}

bool GSamReader::next(GSamRecord& rec) {
  if (!f_) GError("Warning: GSamReader::next() called with no open file.\n");
  if (!more()) return false;
  rec.init(f_->rec(next_++), &f_->hdr);
  return true;
}

GSamWriter::GSamWriter(const char* fname, sam_hdr_t* bh, GSamFileType ftype) {
  if (!bh) GError("Error: no header data provided for GSamWriter::create()!\n");
  if (ftype != GSamFile_BAM) GError("Error: only BAM output is supported\n");
  hdr_ = *bh;
  // zlib level of the BGZF members: 6 as htslib's default; TBK_BAM_LEVEL overrides (the records are the same at any level)
  const int level = getenv("TBK_BAM_LEVEL") ? std::max(0, std::min(9, atoi(getenv("TBK_BAM_LEVEL")))) : 6;
  if (!w_.open(fname, level)) GError("Error: could not create output file %s\n", fname);
  std::vector<uint8_t> h;
  hdr_.serialize(h);
  if (!w_.write(h.data(), h.size())) GError("Error writing the header to %s\n", fname);
}

void GSamWriter::write_raw(const tbh::BamRec& r) {
  uint32_t bs = (uint32_t)r.d.size();
  uint8_t le[4] = {(uint8_t)bs, (uint8_t)(bs >> 8), (uint8_t)(bs >> 16), (uint8_t)(bs >> 24)};
  if (!w_.write(le, 4) || !w_.write(r.d.data(), r.d.size())) GError("Error: failed to write an alignment record\n");
}

void GSamWriter::write_framed(const uint8_t* p, size_t n) {
  if (n && !w_.write(p, n)) GError("Error: failed to write alignment records\n");
}

void GSamWriter::write_members(const uint8_t* z, size_t n) {
  if (!w_.write_members(z, n)) GError("Error: failed to write alignment records\n");
}

void GSamWriter::write(GSamRecord* brec) {
  if (brec) write_raw(*brec->get_b());
}
