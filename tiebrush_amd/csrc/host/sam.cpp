// sam.cpp — SAM text input: the file is turned into the uncompressed BAM byte stream the rest of the host code reads
// (header block + records), so a SAM file is just another input of tiebrush / tiecov.  The reference gets this from htslib
// (GSamReader opens SAM, BAM and CRAM alike: GSam.h:371-401, sam_read1); the encoding rules below are the SAM specification's
// and htslib 1.18's sam_parse1 choices where the two leave room (integer tags take the smallest type that holds the value,
// 'bin' by reg2bin over the alignment's reference span).  CRAM is not read (it needs the reference sequences and the CRAM
// codecs; refused with a message).  A SAM input is held in memory whole — the streaming window of BamFile applies to BAM.
#include <ctype.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>

#include "bam.h"
#include "sam.h"

namespace tbh {

namespace {
void put32(std::vector<uint8_t>& o, uint32_t v) {
  for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i)));
}
void put16(std::vector<uint8_t>& o, uint32_t v) {
  o.push_back((uint8_t)v);
  o.push_back((uint8_t)(v >> 8));
}
int reg2bin(int64_t beg, int64_t end) {  // SAM specification, section 5.3
  --end;
  if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}
bool parse_i64(const std::string& s, int64_t* v) {
  if (s.empty()) return false;
  errno = 0;
  char* e = nullptr;
  long long x = strtoll(s.c_str(), &e, 10);
  if (errno || *e) return false;
  *v = x;
  return true;
}
std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    size_t e = s.find(sep, b);
    if (e == std::string::npos) {
      out.push_back(s.substr(b));
      break;
    }
    out.push_back(s.substr(b, e - b));
    b = e + 1;
  }
  return out;
}
// one optional field "TG:T:VALUE" -> BAM aux bytes
bool put_aux(const std::string& f, std::vector<uint8_t>& o) {
  if (f.size() < 5 || f[2] != ':' || f[4] != ':') return false;
  const char t = f[3];
  const std::string v = f.substr(5);
  o.push_back((uint8_t)f[0]);
  o.push_back((uint8_t)f[1]);
  switch (t) {
    case 'A':
      if (v.size() != 1) return false;
      o.push_back('A');
      o.push_back((uint8_t)v[0]);
      return true;
    case 'i': {
      int64_t x;
      if (!parse_i64(v, &x)) return false;
      if (x < 0) {
        if (x >= -128) {
          o.push_back('c');
          o.push_back((uint8_t)(int8_t)x);
        } else if (x >= -32768) {
          o.push_back('s');
          put16(o, (uint32_t)(uint16_t)(int16_t)x);
        } else if (x >= INT32_MIN) {
          o.push_back('i');
          put32(o, (uint32_t)(int32_t)x);
        } else {
          return false;
        }
      } else {
        if (x <= 255) {
          o.push_back('C');
          o.push_back((uint8_t)x);
        } else if (x <= 65535) {
          o.push_back('S');
          put16(o, (uint32_t)x);
        } else if (x <= (int64_t)UINT32_MAX) {
          o.push_back('I');
          put32(o, (uint32_t)x);
        } else {
          return false;
        }
      }
      return true;
    }
    case 'f': {
      errno = 0;
      char* e = nullptr;
      float x = strtof(v.c_str(), &e);
      if (v.empty() || *e) return false;
      uint32_t b;
      memcpy(&b, &x, 4);
      o.push_back('f');
      put32(o, b);
      return true;
    }
    case 'Z':
    case 'H':
      o.push_back((uint8_t)t);
      o.insert(o.end(), v.begin(), v.end());
      o.push_back(0);
      return true;
    case 'B': {
      if (v.size() < 1) return false;
      const char st = v[0];
      if (!strchr("cCsSiIf", st)) return false;
      std::vector<std::string> items;
      if (v.size() > 2) items = split(v.substr(2), ',');
      if (v.size() >= 2 && v[1] != ',') return false;
      o.push_back('B');
      o.push_back((uint8_t)st);
      put32(o, (uint32_t)items.size());
      for (auto& it : items) {
        if (st == 'f') {
          char* e = nullptr;
          float x = strtof(it.c_str(), &e);
          if (it.empty() || *e) return false;
          uint32_t b;
          memcpy(&b, &x, 4);
          put32(o, b);
        } else {
          int64_t x;
          if (!parse_i64(it, &x)) return false;
          if (st == 'c' || st == 'C')
            o.push_back((uint8_t)x);
          else if (st == 's' || st == 'S')
            put16(o, (uint32_t)(uint16_t)x);
          else
            put32(o, (uint32_t)x);
        }
      }
      return true;
    }
    default:
      return false;
  }
}
}  // namespace

bool sam_probe(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char line[4096];
  bool ok = false;
  if (fgets(line, sizeof(line), f)) {
    if (line[0] == '@' && isupper((unsigned char)line[1]) && isupper((unsigned char)line[2]) && (line[3] == '\t' || line[3] == '\n')) {
      ok = true;  // a header line
    } else {
      int tabs = 0;
      for (const char* p = line; *p; ++p) tabs += *p == '\t';
      const char* p = strchr(line, '\t');
      ok = tabs >= 10 && p && isdigit((unsigned char)p[1]);  // eleven mandatory fields, FLAG a number
    }
  }
  fclose(f);
  return ok;
}

bool cram_probe(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char m[4] = {0, 0, 0, 0};
  const size_t n = fread(m, 1, 4, f);
  fclose(f);
  return n == 4 && memcmp(m, "CRAM", 4) == 0;
}

bool sam_to_bam(const std::string& path, std::vector<uint8_t>& out, std::string& err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  std::string text;
  std::vector<std::string> names;
  std::vector<uint32_t> lens;
  std::map<std::string, int32_t> name2tid;
  std::vector<uint8_t> recs;
  std::string line;
  char buf[1 << 16];
  bool in_header = true;
  uint64_t lineno = 0;
  auto fail = [&](const char* what) {
    err = std::string(what) + " at line " + std::to_string(lineno) + " of " + path;
    fclose(f);
    return false;
  };
  for (;;) {
    line.clear();
    bool got = false;
    while (fgets(buf, sizeof(buf), f)) {
      got = true;
      line += buf;
      if (!line.empty() && line.back() == '\n') break;
    }
    if (!got) break;
    ++lineno;
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
    if (line.empty()) continue;
    if (line[0] == '@') {
      if (!in_header) return fail("header line after the first alignment");
      text += line;
      text += '\n';
      if (line.compare(0, 4, "@SQ\t") == 0) {
        std::string sn;
        int64_t ln = -1;
        for (auto& fld : split(line.substr(4), '\t')) {
          if (fld.compare(0, 3, "SN:") == 0) sn = fld.substr(3);
          if (fld.compare(0, 3, "LN:") == 0 && !parse_i64(fld.substr(3), &ln)) return fail("bad @SQ LN");
        }
        if (sn.empty() || ln < 0 || ln > INT32_MAX) return fail("bad @SQ line");
        if (name2tid.count(sn)) return fail("duplicate @SQ SN");
        name2tid[sn] = (int32_t)names.size();
        names.push_back(sn);
        lens.push_back((uint32_t)ln);
      }
      continue;
    }
    in_header = false;
    std::vector<std::string> fld = split(line, '\t');
    if (fld.size() < 11) return fail("fewer than 11 fields");
    int64_t flag, pos, mapq, pnext, tlen;
    if (!parse_i64(fld[1], &flag) || flag < 0 || flag > 65535) return fail("bad FLAG");
    if (!parse_i64(fld[3], &pos) || pos < 0 || pos > INT32_MAX) return fail("bad POS");
    if (!parse_i64(fld[4], &mapq) || mapq < 0 || mapq > 255) return fail("bad MAPQ");
    if (!parse_i64(fld[7], &pnext) || pnext < 0 || pnext > INT32_MAX) return fail("bad PNEXT");
    if (!parse_i64(fld[8], &tlen) || tlen < INT32_MIN || tlen > INT32_MAX) return fail("bad TLEN");
    int32_t tid = -1, mtid = -1;
    if (fld[2] != "*") {
      auto it = name2tid.find(fld[2]);
      if (it == name2tid.end()) return fail("RNAME not in the header");
      tid = it->second;
    }
    if (fld[6] == "=")
      mtid = tid;
    else if (fld[6] != "*") {
      auto it = name2tid.find(fld[6]);
      if (it == name2tid.end()) return fail("RNEXT not in the header");
      mtid = it->second;
    }
    const std::string& qn = fld[0];
    if (qn.empty() || qn.size() > 254) return fail("bad QNAME");
    // CIGAR
    std::vector<uint32_t> cig;
    int64_t reflen = 0;
    if (fld[5] != "*") {
      const char* p = fld[5].c_str();
      while (*p) {
        if (!isdigit((unsigned char)*p)) return fail("bad CIGAR");
        char* e = nullptr;
        unsigned long n = strtoul(p, &e, 10);
        const char* ops = "MIDNSHP=XB";
        const char* q = *e ? strchr(ops, *e) : nullptr;
        if (!q || n >= (1ul << 28)) return fail("bad CIGAR");
        const uint32_t op = (uint32_t)(q - ops);
        cig.push_back((uint32_t)(n << 4) | op);
        if ((0x18Du >> op) & 1u) reflen += (int64_t)n;
        p = e + 1;
      }
      if (cig.size() > 65535) return fail("more than 65535 CIGAR operations");
    }
    // SEQ / QUAL
    std::vector<uint8_t> seq, qual;
    uint32_t l_seq = 0;
    if (fld[9] != "*") {
      static const char* code = "=ACMGRSVTWYHKDBN";
      l_seq = (uint32_t)fld[9].size();
      seq.assign((l_seq + 1) / 2, 0);
      for (uint32_t i = 0; i < l_seq; ++i) {
        const char c = (char)toupper((unsigned char)fld[9][i]);
        const char* q = strchr(code, c);
        const uint8_t v = (q && c) ? (uint8_t)(q - code) : 15;
        seq[i >> 1] |= (i & 1) ? v : (uint8_t)(v << 4);
      }
      if (fld[10] == "*") {
        qual.assign(l_seq, 0xFF);
      } else {
        if (fld[10].size() != l_seq) return fail("SEQ and QUAL differ in length");
        qual.resize(l_seq);
        for (uint32_t i = 0; i < l_seq; ++i) qual[i] = (uint8_t)(fld[10][i] - 33);
      }
    }
    std::vector<uint8_t> aux;
    for (size_t a = 11; a < fld.size(); ++a)
      if (!put_aux(fld[a], aux)) return fail("bad optional field");
    const int64_t p0 = pos - 1;  // 0-based, -1 when POS is 0
    const int64_t end0 = ((flag & 0x4) || reflen == 0) ? p0 + 1 : p0 + reflen;
    const int bin = reg2bin(p0 < 0 ? -1 : p0, end0 < 0 ? 0 : end0);
    const size_t bs = 32 + qn.size() + 1 + 4 * cig.size() + seq.size() + qual.size() + aux.size();
    put32(recs, (uint32_t)bs);
    put32(recs, (uint32_t)tid);
    put32(recs, (uint32_t)(int32_t)p0);
    recs.push_back((uint8_t)(qn.size() + 1));
    recs.push_back((uint8_t)mapq);
    put16(recs, (uint32_t)bin);
    put16(recs, (uint32_t)cig.size());
    put16(recs, (uint32_t)flag);
    put32(recs, l_seq);
    put32(recs, (uint32_t)mtid);
    put32(recs, (uint32_t)(int32_t)(pnext - 1));
    put32(recs, (uint32_t)(int32_t)tlen);
    recs.insert(recs.end(), qn.begin(), qn.end());
    recs.push_back(0);
    for (uint32_t c : cig) put32(recs, c);
    recs.insert(recs.end(), seq.begin(), seq.end());
    recs.insert(recs.end(), qual.begin(), qual.end());
    recs.insert(recs.end(), aux.begin(), aux.end());
  }
  fclose(f);
  out.clear();
  for (uint8_t c : {(uint8_t)'B', (uint8_t)'A', (uint8_t)'M', (uint8_t)1}) out.push_back(c);
  put32(out, (uint32_t)text.size());
  out.insert(out.end(), text.begin(), text.end());
  put32(out, (uint32_t)names.size());
  for (size_t i = 0; i < names.size(); ++i) {
    put32(out, (uint32_t)names[i].size() + 1);
    out.insert(out.end(), names[i].begin(), names[i].end());
    out.push_back(0);
    put32(out, lens[i]);
  }
  out.insert(out.end(), recs.begin(), recs.end());
  return true;
}

}  // namespace tbh
