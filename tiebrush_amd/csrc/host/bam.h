// bam.h — BAM header / record access and aux-tag editing with htslib 1.18's observable rules
// (SURVEY.md §3.2): what the reference reaches through sam_hdr_*, bam_aux_* and sam_read1/sam_write1.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace tbh {

struct BamHeader {
  std::string text;  // '\n'-terminated lines, no trailing NUL
  std::vector<std::string> target_name;
  std::vector<uint32_t> target_len;
  int32_t n_targets = 0;

  bool parse(const std::vector<uint8_t>& data, size_t* rec_begin, std::string& err);
  void serialize(std::vector<uint8_t>& out) const;
  std::vector<std::string> lines() const;
  // "@HD ... SO:coordinate" (tmerge.cpp:64-67)
  bool sorted_by_coordinate() const;
  // a @PG line with PN:TieBrush and a VN tag (tmerge.cpp:70-77)
  bool is_tiebrush() const;
  // payloads of "@CO\tSAMPLE:<x>" lines (commons.h:23-71, tmerge.cpp:193-214)
  std::vector<std::string> co_samples() const;
  int name2tid(const std::string& name) const;
  // sam_hdr_add_line(hdr,"CO","SAMPLE:<x>"): after the last @CO line, else at the end
  void add_co(const std::string& payload);
  // sam_hdr_add_pg(hdr,"TieBrush","VN",ver,"CL",cl): unique ID (TieBrush, TieBrush.1, ...), PP = last @PG ID,
  // placed after the last @PG line (before the @CO block) as in the reference's output
  void add_pg(const std::string& name, const std::string& ver, const std::string& cl);
};

// Offsets inside one BAM record (pointing at the refID field, i.e. after block_size)
struct RecView {
  const uint8_t* p = nullptr;  // core start
  uint32_t len = 0;            // block_size
  int32_t tid() const;
  int32_t pos() const;
  uint8_t l_read_name() const { return p[8]; }
  uint8_t mapq() const { return p[9]; }
  uint16_t n_cigar() const;
  uint16_t flag() const;
  int32_t l_seq() const;
  const char* qname() const { return (const char*)(p + 32); }
  const uint8_t* cigar_bytes() const { return p + 32 + l_read_name(); }
  uint32_t cigar(uint32_t i) const;
  const uint8_t* aux_begin() const;
  const uint8_t* aux_end() const { return p + len; }
};

// bam_aux_get: pointer to the TYPE byte of the first occurrence of `tag`, or nullptr
const uint8_t* aux_get(const uint8_t* aux, const uint8_t* end, const char tag[2]);
// size in bytes of the field starting at its 2-byte tag (tag+type+value), 0 on malformed data
size_t aux_field_size(const uint8_t* field, const uint8_t* end);
int64_t aux2i(const uint8_t* s);  // bam_aux2i: integer types, else 0
double aux2f(const uint8_t* s);   // bam_aux2f: d, f, integers, else 0
char aux2A(const uint8_t* s);     // bam_aux2A: 'A' else 0
const char* aux2Z(const uint8_t* s);  // bam_aux2Z: Z/H else NULL

// An owned, editable record (the bam1_t the reference mutates before sam_write1)
struct BamRec {
  std::vector<uint8_t> d;  // core + variable part, WITHOUT the leading block_size
  RecView view() const {
    RecView v;
    v.p = d.data();
    v.len = (uint32_t)d.size();
    return v;
  }
  int update_int(const char tag[2], int64_t val);   // bam_aux_update_int
  int update_float(const char tag[2], float val);   // bam_aux_update_float
  int del(const char tag[2]);                       // bam_aux_del(bam_aux_get())
};

// A BAM file read as a sliding window: open() reads the header, fill() inflates further BGZF members and indexes the
// records they complete, consume() drops records from the front.  Host memory holds the window, not the file; the file
// descriptor is only open during a fill() (any number of inputs, whatever RLIMIT_NOFILE says).  Records handed out by
// rec() are invalidated by the next fill() / consume().
struct BamFile {
  BamHeader hdr;
  std::vector<uint8_t> data;      // inflated window (the first record starts at rec_off[0])
  std::vector<uint64_t> rec_off;  // offset (in data) of every indexed record's block_size field
  std::vector<int64_t> key_start; // per indexed record: sort key (tid + 1) << 32 | (pos + 1); refID -1 -> INT64_MAX
  std::vector<int64_t> pmax_end;  // per indexed record: running max over the window of (tid + 1) << 32 | end + 1 (end 1-based inclusive)
  std::string path;
  bool open(const std::string& path, std::string& err, int threads = 1);
  // inflate until the window holds at least min_records records (or the file ends); false on malformed input
  bool fill(size_t min_records, std::string& err, int threads = 1, size_t chunk_bytes = (size_t)8 << 20);
  void consume(size_t n_records);  // drop the first n records of the window
  bool at_eof() const { return eof_ && parsed_ == data.size(); }
  bool load(const std::string& path, std::string& err, int threads = 1);  // open + everything (small files, tests)
  size_t n() const { return rec_off.size(); }
  RecView rec(size_t i) const;
  uint64_t first_index() const { return consumed_; }  // file-wide index of the window's first record

 private:
  bool index_records(std::string& err);
  uint64_t file_pos_ = 0;   // next compressed byte to read
  size_t parsed_ = 0;       // bytes of `data` covered by the header and the indexed records
  uint64_t consumed_ = 0;   // records dropped so far
  bool eof_ = false;
  bool header_done_ = false;
};

}  // namespace tbh
