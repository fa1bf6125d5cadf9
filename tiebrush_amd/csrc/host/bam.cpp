#include "bam.h"
#include "sam.h"

#include <stdio.h>
#include <string.h>

#include <sstream>

#include "bgzf.h"

namespace tbh {

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline void wr32(std::vector<uint8_t>& o, uint32_t v) {
  for (int k = 0; k < 4; ++k) o.push_back((uint8_t)(v >> (8 * k)));
}

// ---- header ---------------------------------------------------------------------------------------
bool BamHeader::parse(const std::vector<uint8_t>& data, size_t* rec_begin, std::string& err) {
  if (data.size() < 12 || memcmp(data.data(), "BAM\1", 4) != 0) {
    err = "not a BAM stream";
    return false;
  }
  uint32_t l_text = rd32(&data[4]);
  if (8 + (size_t)l_text + 4 > data.size()) {
    err = "truncated BAM header";
    return false;
  }
  text.assign((const char*)&data[8], l_text);
  size_t z = text.find('\0');
  if (z != std::string::npos) text.resize(z);
  if (!text.empty() && text.back() != '\n') text.push_back('\n');
  size_t p = 8 + l_text;
  n_targets = (int32_t)rd32(&data[p]);
  p += 4;
  target_name.clear();
  target_len.clear();
  for (int32_t i = 0; i < n_targets; ++i) {
    if (p + 4 > data.size()) {
      err = "truncated reference list";
      return false;
    }
    uint32_t l_name = rd32(&data[p]);
    if (p + 8 + l_name > data.size()) {
      err = "truncated reference list";
      return false;
    }
    target_name.emplace_back((const char*)&data[p + 4], l_name ? l_name - 1 : 0);
    target_len.push_back(rd32(&data[p + 4 + l_name]));
    p += 8 + l_name;
  }
  *rec_begin = p;
  return true;
}

void BamHeader::serialize(std::vector<uint8_t>& out) const {
  out.insert(out.end(), {'B', 'A', 'M', 1});
  wr32(out, (uint32_t)text.size());
  out.insert(out.end(), text.begin(), text.end());
  wr32(out, (uint32_t)n_targets);
  for (int32_t i = 0; i < n_targets; ++i) {
    wr32(out, (uint32_t)target_name[i].size() + 1);
    out.insert(out.end(), target_name[i].begin(), target_name[i].end());
    out.push_back(0);
    wr32(out, target_len[i]);
  }
}

std::vector<std::string> BamHeader::lines() const {
  std::vector<std::string> r;
  size_t s = 0;
  while (s < text.size()) {
    size_t e = text.find('\n', s);
    if (e == std::string::npos) e = text.size();
    if (e > s) r.push_back(text.substr(s, e - s));
    s = e + 1;
  }
  return r;
}

static std::vector<std::string> split_tabs(const std::string& l) {
  std::vector<std::string> f;
  size_t s = 0;
  for (;;) {
    size_t e = l.find('\t', s);
    f.push_back(l.substr(s, e == std::string::npos ? std::string::npos : e - s));
    if (e == std::string::npos) break;
    s = e + 1;
  }
  return f;
}

bool BamHeader::sorted_by_coordinate() const {
  for (auto& l : lines())
    if (l.compare(0, 3, "@HD") == 0)
      for (auto& f : split_tabs(l))
        if (f == "SO:coordinate") return true;
  return false;
}

bool BamHeader::is_tiebrush() const {
  for (auto& l : lines()) {
    if (l.compare(0, 3, "@PG") != 0) continue;
    bool pn = false, vn = false;
    for (auto& f : split_tabs(l)) {
      if (f == "PN:TieBrush") pn = true;
      if (f.compare(0, 3, "VN:") == 0) vn = true;
    }
    if (pn && vn) return true;
  }
  return false;
}

std::vector<std::string> BamHeader::co_samples() const {
  std::vector<std::string> r;
  for (auto& l : lines())
    if (l.compare(0, 11, "@CO\tSAMPLE:") == 0) {
      std::string v = l.substr(11);
      size_t t = v.find('\t');  // tmerge.cpp:211 reads up to the next tab
      if (t != std::string::npos) v.resize(t);
      r.push_back(v);
    }
  return r;
}

int BamHeader::name2tid(const std::string& name) const {
  for (int32_t i = 0; i < n_targets; ++i)
    if (target_name[i] == name) return i;
  return -1;
}

static void insert_after_last(std::string& text, const char* type3, const std::string& line) {
  // position just after the last line starting with type3 ("@CO"/"@PG"); end of text when none
  size_t best = std::string::npos, s = 0;
  while (s < text.size()) {
    size_t e = text.find('\n', s);
    if (e == std::string::npos) e = text.size() - 1;
    if (text.compare(s, 3, type3) == 0) best = e + 1;
    s = e + 1;
  }
  if (best == std::string::npos) best = text.size();
  text.insert(best, line + "\n");
}

void BamHeader::add_co(const std::string& payload) { insert_after_last(text, "@CO", "@CO\t" + payload); }

void BamHeader::add_pg(const std::string& name, const std::string& ver, const std::string& cl) {
  std::vector<std::string> ids;
  std::string last_id;
  for (auto& l : lines())
    if (l.compare(0, 3, "@PG") == 0)
      for (auto& f : split_tabs(l))
        if (f.compare(0, 3, "ID:") == 0) {
          ids.push_back(f.substr(3));
          last_id = f.substr(3);
        }
  std::string id = name;
  for (int k = 1;; ++k) {
    bool clash = false;
    for (auto& x : ids) clash |= (x == id);
    if (!clash) break;
    id = name + "." + std::to_string(k);
  }
  std::string line = "@PG\tID:" + id + "\tPN:" + name;
  if (!last_id.empty()) line += "\tPP:" + last_id;
  line += "\tVN:" + ver + "\tCL:" + cl;
  // after the last @PG; when there is none, before any @CO block (htslib keeps @CO last)
  bool has_pg = !ids.empty();
  if (has_pg) {
    insert_after_last(text, "@PG", line);
  } else {
    size_t co = std::string::npos, s = 0;
    while (s < text.size()) {
      size_t e = text.find('\n', s);
      if (e == std::string::npos) e = text.size() - 1;
      if (text.compare(s, 3, "@CO") == 0) {
        co = s;
        break;
      }
      s = e + 1;
    }
    if (co == std::string::npos) co = text.size();
    text.insert(co, line + "\n");
  }
}

// ---- records -----------------------------------------------------------------------------------------
int32_t RecView::tid() const { return (int32_t)rd32(p); }
int32_t RecView::pos() const { return (int32_t)rd32(p + 4); }
uint16_t RecView::n_cigar() const { return rd16(p + 12); }
uint16_t RecView::flag() const { return rd16(p + 14); }
int32_t RecView::l_seq() const { return (int32_t)rd32(p + 16); }
uint32_t RecView::cigar(uint32_t i) const { return rd32(cigar_bytes() + 4 * i); }
const uint8_t* RecView::aux_begin() const {
  int32_t ls = l_seq();
  return cigar_bytes() + 4 * (size_t)n_cigar() + (size_t)((ls + 1) / 2) + (size_t)ls;
}

static size_t type_size(uint8_t t) {
  switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
  }
  return 0;
}

size_t aux_field_size(const uint8_t* f, const uint8_t* end) {
  if (f + 3 > end) return 0;
  uint8_t t = f[2];
  size_t fs = type_size(t);
  if (fs) return f + 3 + fs <= end ? 3 + fs : 0;
  if (t == 'Z' || t == 'H') {
    const uint8_t* z = (const uint8_t*)memchr(f + 3, 0, (size_t)(end - (f + 3)));
    return z ? (size_t)(z - f) + 1 : 0;
  }
  if (t == 'B') {
    if (f + 8 > end) return 0;
    size_t es = type_size(f[3]);
    if (!es || f[3] == 'A' || f[3] == 'd') return 0;
    size_t n = rd32(f + 4);
    return f + 8 + n * es <= end ? 8 + n * es : 0;
  }
  return 0;
}

const uint8_t* aux_get(const uint8_t* aux, const uint8_t* end, const char tag[2]) {
  const uint8_t* f = aux;
  while (f + 3 <= end) {
    size_t sz = aux_field_size(f, end);
    if (!sz) return nullptr;
    if (f[0] == (uint8_t)tag[0] && f[1] == (uint8_t)tag[1]) return f + 2;
    f += sz;
  }
  return nullptr;
}

int64_t aux2i(const uint8_t* s) {
  switch (*s) {
    case 'c': return (int8_t)s[1];
    case 'C': return s[1];
    case 's': return (int16_t)rd16(s + 1);
    case 'S': return rd16(s + 1);
    case 'i': return (int32_t)rd32(s + 1);
    case 'I': return rd32(s + 1);
  }
  return 0;
}
double aux2f(const uint8_t* s) {
  if (*s == 'd') {
    double v;
    memcpy(&v, s + 1, 8);
    return v;
  }
  if (*s == 'f') {
    float v;
    memcpy(&v, s + 1, 4);
    return v;
  }
  return (double)aux2i(s);
}
char aux2A(const uint8_t* s) { return *s == 'A' ? (char)s[1] : 0; }
const char* aux2Z(const uint8_t* s) { return (*s == 'Z' || *s == 'H') ? (const char*)(s + 1) : nullptr; }

int BamRec::update_int(const char tag[2], int64_t val) {
  if (val < INT32_MIN || val > (int64_t)UINT32_MAX) return -1;
  uint8_t type;
  uint32_t sz;
  if (val < INT16_MIN) { type = 'i'; sz = 4; }
  else if (val < INT8_MIN) { type = 's'; sz = 2; }
  else if (val < 0) { type = 'c'; sz = 1; }
  else if (val < UINT8_MAX) { type = 'C'; sz = 1; }      // 255 does NOT fit 'C' (strict <)
  else if (val < UINT16_MAX) { type = 'S'; sz = 2; }
  else { type = 'I'; sz = 4; }
  RecView v = view();
  const uint8_t* s = aux_get(v.aux_begin(), v.aux_end(), tag);
  uint8_t le[4];
  uint32_t uv = (uint32_t)val;
  for (int k = 0; k < 4; ++k) le[k] = (uint8_t)(uv >> (8 * k));
  if (!s) {  // append tag,type,value
    d.push_back((uint8_t)tag[0]);
    d.push_back((uint8_t)tag[1]);
    d.push_back(type);
    d.insert(d.end(), le, le + sz);
    return 0;
  }
  uint32_t old_sz;
  switch (*s) {
    case 'c': case 'C': old_sz = 1; break;
    case 's': case 'S': old_sz = 2; break;
    case 'i': case 'I': old_sz = 4; break;
    default: return -1;  // EINVAL: not an integer tag, stale value survives
  }
  size_t off = (size_t)(s - d.data());
  if (old_sz < sz) {
    d.insert(d.begin() + off + 1 + old_sz, sz - old_sz, 0);
  } else {  // reuse the old width; only the sign class of the type letter follows the value
    sz = old_sz;
    type = (uint8_t)((val < 0 ? "\0cs\0i" : "\0CS\0I")[old_sz]);
  }
  d[off] = type;
  memcpy(&d[off + 1], le, sz);
  return 0;
}

int BamRec::update_float(const char tag[2], float val) {
  RecView v = view();
  const uint8_t* s = aux_get(v.aux_begin(), v.aux_end(), tag);
  uint8_t le[4];
  memcpy(le, &val, 4);
  if (!s) {
    d.push_back((uint8_t)tag[0]);
    d.push_back((uint8_t)tag[1]);
    d.push_back('f');
    d.insert(d.end(), le, le + 4);
    return 0;
  }
  size_t off = (size_t)(s - d.data());
  if (*s == 'd') {
    d.erase(d.begin() + off + 5, d.begin() + off + 9);
  } else if (*s != 'f') {
    return -1;  // EINVAL: the reference ignores the return value (GSam.h:303-305)
  }
  d[off] = 'f';
  memcpy(&d[off + 1], le, 4);
  return 0;
}

int BamRec::del(const char tag[2]) {
  RecView v = view();
  const uint8_t* s = aux_get(v.aux_begin(), v.aux_end(), tag);
  if (!s) return 0;
  size_t off = (size_t)(s - 2 - d.data());
  size_t sz = aux_field_size(d.data() + off, d.data() + d.size());
  if (!sz) return -1;
  d.erase(d.begin() + off, d.begin() + off + sz);
  return 0;
}


bool BamFile::open(const std::string& p, std::string& err, int threads) {
  path = p;
  data.clear();
  rec_off.clear();
  key_start.clear();
  pmax_end.clear();
  file_pos_ = 0;
  parsed_ = 0;
  consumed_ = 0;
  eof_ = false;
  header_done_ = false;
  // the header may span several members: inflate until it parses
  for (;;) {
    if (!fill(0, err, threads, (size_t)256 << 10)) return false;  // (a header-sized read: the records are read when a tile asks for them)
    if (header_done_) return true;
    if (eof_) {
      err = "truncated BAM header (" + p + ")";
      return false;
    }
  }
}

bool BamFile::load(const std::string& p, std::string& err, int threads) {
  if (!open(p, err, threads)) return false;
  while (!at_eof())
    if (!fill(rec_off.size() + ((size_t)1 << 20), err, threads)) return false;
  return true;
}

// index the records that are complete in data[parsed_, ...)
bool BamFile::index_records(std::string& err) {
  if (!header_done_) {
    size_t off = 0;
    std::string herr;
    if (!hdr.parse(data, &off, herr)) {
      if (!eof_) return true;  // not all of it is here yet
      err = herr + " (" + path + ")";
      return false;
    }
    header_done_ = true;
    parsed_ = off;
  }
  size_t off = parsed_;
  int64_t run = pmax_end.empty() ? INT64_MIN : pmax_end.back();
  while (off + 4 <= data.size()) {
    uint32_t bs = rd32(&data[off]);
    if (bs < 32) {
      err = "corrupt record in " + path;
      return false;
    }
    if (off + 4 + (size_t)bs > data.size()) break;  // the record continues in the next member
    // the variable-length fields must lie inside the record and refID inside the header (htslib's bam_read1 rejects the same):
    // RecView::qname / cigar / aux_begin index by these lengths without further checks
    const uint8_t* r = &data[off + 4];
    const int32_t tid = (int32_t)rd32(r);
    const int32_t pos = (int32_t)rd32(r + 4);
    const uint32_t l_read_name = r[8];
    const uint32_t n_cigar = (uint32_t)r[12] | ((uint32_t)r[13] << 8);
    const uint32_t flag = (uint32_t)r[14] | ((uint32_t)r[15] << 8);
    const int32_t l_seq = (int32_t)rd32(r + 16);
    const int32_t mtid = (int32_t)rd32(r + 20);
    const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + (l_seq < 0 ? 0ull : ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq);
    const bool name_ok = l_read_name >= 1 && 32 + l_read_name <= bs && r[32 + l_read_name - 1] == 0;
    if (l_seq < 0 || need > bs || !name_ok || tid < -1 || tid >= hdr.n_targets || mtid < -1 || mtid >= hdr.n_targets) {
      err = "malformed record " + std::to_string(consumed_ + rec_off.size()) + " in " + path +
            " (field lengths / reference id outside the record / header)";
      return false;
    }
    // sort key and read end (reference length of the CIGAR: M, D, N, =, X) for the tile cuts of the streaming driver
    int64_t ks = tid < 0 ? INT64_MAX : (((int64_t)tid + 1) << 32) | (int64_t)(uint32_t)(pos + 1);
    int64_t ke = ks;
    if (tid >= 0 && !(flag & 0x4)) {
      int64_t l = 0;
      const uint8_t* c = r + 32 + l_read_name;
      for (uint32_t i = 0; i < n_cigar; ++i) {
        const uint32_t w = rd32(c + 4 * i);
        if ((0x18Du >> (w & 0xF)) & 1u) l += w >> 4;
      }
      // (+ 1: the YD lists may hold a node that starts one base behind a read's end — a CIGAR ending in an intron — so a tile
      // may only be cut where the next read starts beyond end + 1)
      ke = (((int64_t)tid + 1) << 32) | (int64_t)(uint32_t)(pos + (l > 0 ? l : 1) + 1);
    }
    if (ke > run) run = ke;
    rec_off.push_back(off);
    key_start.push_back(ks);
    pmax_end.push_back(run);
    off += 4 + (size_t)bs;
  }
  parsed_ = off;
  if (eof_ && parsed_ != data.size()) {
    err = "truncated record at the end of " + path;
    return false;
  }
  return true;
}

bool BamFile::fill(size_t min_records, std::string& err, int threads, size_t chunk_bytes) {
  const size_t kFillBytes = chunk_bytes < ((size_t)128 << 10) ? ((size_t)128 << 10) : chunk_bytes;  // compressed bytes per read (>= 2 members)
  std::vector<uint8_t> raw;
  bool first = true;
  if (file_pos_ == 0 && !eof_ && !header_done_ && !bgzf_probe(path) && sam_probe(path)) {
    // SAM text: converted whole into the byte stream a BAM file inflates to (sam.cpp); no window — the file is in memory
    std::vector<uint8_t> all;
    if (!sam_to_bam(path, all, err)) return false;
    data.insert(data.end(), all.begin(), all.end());
    eof_ = true;
    return index_records(err);
  }
  while (first || (rec_off.size() < min_records && !eof_) || (!header_done_ && !eof_)) {
    first = false;
    if (eof_) break;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
      err = "cannot open " + path;
      return false;
    }
    raw.resize(kFillBytes);
    if (fseeko(f, (off_t)file_pos_, SEEK_SET) != 0) {
      fclose(f);
      err = "cannot seek in " + path;
      return false;
    }
    const size_t got = fread(raw.data(), 1, raw.size(), f);
    const bool hit_eof = got < raw.size();
    fclose(f);
    size_t used = 0;
    if (!bgzf_inflate_chunk(raw.data(), got, hit_eof, data, &used, err, threads, path)) return false;
    file_pos_ += used;
    if (hit_eof) eof_ = true;
    if (used == 0 && !hit_eof) {
      err = "BGZF member larger than the read buffer in " + path;
      return false;
    }
    if (!index_records(err)) return false;
    if (min_records == 0 && header_done_) break;
  }
  if (eof_ && !index_records(err)) return false;
  return true;
}

void BamFile::consume(size_t nrec) {
  if (nrec == 0) return;
  if (nrec > rec_off.size()) nrec = rec_off.size();
  const size_t cut = nrec == rec_off.size() ? parsed_ : (size_t)rec_off[nrec];
  consumed_ += nrec;
  rec_off.erase(rec_off.begin(), rec_off.begin() + (long)nrec);
  key_start.erase(key_start.begin(), key_start.begin() + (long)nrec);
  pmax_end.erase(pmax_end.begin(), pmax_end.begin() + (long)nrec);
  // (pmax_end keeps counting the dropped records: they end before every later cut, so they never decide one)
  data.erase(data.begin(), data.begin() + (long)cut);
  for (auto& o : rec_off) o -= cut;
  parsed_ -= cut;
}

RecView BamFile::rec(size_t i) const {
  RecView v;
  v.p = data.data() + rec_off[i] + 4;
  v.len = rd32(data.data() + rec_off[i]);
  return v;
}

}  // namespace tbh
