#include "bgzf.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <thread>

namespace tbh {

// ---- raw deflate codec ------------------------------------------------------------------------------------------------
// BGZF members are whole, independent raw-deflate streams of <= 64 KiB: exactly what libdeflate's one-shot API is made for (it
// is what htslib itself links when it can, 2-3x zlib's speed at the same levels).  The image ships libdeflate.so.0 without
// headers, so the handful of entry points is resolved with dlopen and declared here; without the library (or with
// TBK_NO_LIBDEFLATE set) everything goes through zlib.  Either way the inflated bytes are the same bytes; the deflated stream
// of a member may differ byte for byte between the two (both are valid deflate of the same payload).
namespace {
struct LibDeflate {
  void* (*alloc_dec)() = nullptr;
  int (*dec)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
  void (*free_dec)(void*) = nullptr;
  void* (*alloc_comp)(int) = nullptr;
  size_t (*comp)(void*, const void*, size_t, void*, size_t) = nullptr;
  void (*free_comp)(void*) = nullptr;
  uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
  bool ok = false;
  LibDeflate() {
    if (getenv("TBK_NO_LIBDEFLATE")) return;
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc_dec = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
    dec = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
    free_dec = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
    alloc_comp = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
    comp = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
    free_comp = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
    crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
    ok = alloc_dec && dec && free_dec && alloc_comp && comp && free_comp && crc;
  }
};
const LibDeflate& ld() {
  static const LibDeflate L;
  return L;
}
struct TlsCodec {  // one decompressor and one compressor per thread (and level)
  void* d = nullptr;
  void* c = nullptr;
  int c_level = -100;
  ~TlsCodec() {
    if (d) ld().free_dec(d);
    if (c) ld().free_comp(c);
  }
};
thread_local TlsCodec tls;
}  // namespace

uint32_t bgzf_crc32(const uint8_t* p, size_t n) {
  if (ld().ok) return ld().crc(0, p, n);
  return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}

// inflate one raw deflate stream that must yield exactly out_n bytes
static bool inflate_raw(const uint8_t* in, size_t n, uint8_t* out, size_t out_n);
bool bgzf_inflate_member(const uint8_t* cdata, size_t clen, uint8_t* out, uint32_t isize, uint32_t crc) {
  if (!inflate_raw(cdata, clen, out, isize)) return false;
  return bgzf_crc32(out, isize) == crc;  // the member's CRC32 covers the uncompressed bytes (RFC 1952); htslib rejects a mismatch
}
static bool inflate_raw(const uint8_t* in, size_t n, uint8_t* out, size_t out_n) {
  if (ld().ok) {
    if (!tls.d) tls.d = ld().alloc_dec();
    size_t got = 0;
    return tls.d && ld().dec(tls.d, in, n, out, out_n, &got) == 0 && got == out_n;
  }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<uint8_t*>(in);
  zs.avail_in = (uInt)n;
  zs.next_out = out;
  zs.avail_out = (uInt)out_n;
  int rc = inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  return rc == Z_STREAM_END && zs.avail_out == 0;
}
// deflate n bytes into out[0, cap): the stream's length, 0 when it does not fit or the codec fails
static size_t deflate_raw(const uint8_t* src, size_t n, int level, uint8_t* out, size_t cap) {
  if (ld().ok) {
    const int lv = level < 0 ? 6 : (level > 12 ? 12 : level);
    if (!tls.c || tls.c_level != lv) {
      if (tls.c) ld().free_comp(tls.c);
      tls.c = ld().alloc_comp(lv);
      tls.c_level = lv;
    }
    return tls.c ? ld().comp(tls.c, src, n, out, cap) : 0;
  }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
  zs.next_in = const_cast<uint8_t*>(src);
  zs.avail_in = (uInt)n;
  zs.next_out = out;
  zs.avail_out = (uInt)cap;
  int rc = deflate(&zs, Z_FINISH);
  size_t clen = zs.total_out;
  deflateEnd(&zs);
  return rc == Z_STREAM_END ? clen : 0;
}

static const uint8_t kEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static constexpr size_t kBlock = 0xff00;

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

int cpu_budget() {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      long quota = atol(q);
      int c = (int)((quota + period - 1) / period);
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  }
  return n;
}

bool bgzf_probe(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  uint8_t h[18];
  size_t n = fread(h, 1, sizeof(h), f);
  fclose(f);
  return n == sizeof(h) && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
}

struct Member {
  size_t off;    // start of the member in the file
  size_t cdata;  // start of the deflate stream
  size_t clen;
  uint32_t isize;
  uint32_t crc;  // CRC32 of the uncompressed bytes (RFC 1952 trailer)
  size_t out_off;
};

// Inflate the whole BGZF members found in raw[0, n) and append their payload to `out`; *consumed = bytes of `raw` they
// occupied (a member cut off by the end of the buffer is left for the next call; with at_eof it is an error).
bool bgzf_inflate_chunk(const uint8_t* raw, size_t n, bool at_eof, std::vector<uint8_t>& out, size_t* consumed, std::string& err,
                        int threads, const std::string& path) {
  std::vector<Member> mem;
  size_t off = 0, total = out.size();
  *consumed = 0;
  while (off < n) {
    if (off + 18 > n) break;  // not even a header: wait for more
    if (raw[off] != 0x1f || raw[off + 1] != 0x8b || raw[off + 2] != 8 || !(raw[off + 3] & 4)) {
      err = "not a BGZF member in " + path;
      return false;
    }
    uint16_t xlen = rd16(&raw[off + 10]);
    size_t p = off + 12, end = p + xlen;
    if (end > n) break;
    int bsize = -1;
    while (p + 4 <= end) {
      uint16_t slen = rd16(&raw[p + 2]);
      if (p + 4 + slen > end) break;  // (a subfield that runs past XLEN: the member is rejected below unless BC was found before)
      if (raw[p] == 'B' && raw[p + 1] == 'C' && slen == 2) bsize = rd16(&raw[p + 4]);
      p += 4 + slen;
    }
    // member = 12 + xlen bytes of header, the deflate stream, CRC32 + ISIZE (8 bytes): a BSIZE too small for that is corrupt
    if (bsize < 0 || (size_t)bsize + 1 < (size_t)12 + xlen + 8) {
      err = "corrupt BGZF member in " + path;
      return false;
    }
    if (off + (size_t)bsize + 1 > n) break;  // the member continues beyond the buffer
    Member m;
    m.off = off;
    m.cdata = off + 12 + xlen;
    m.clen = (size_t)bsize + 1 - 8 - (12 + xlen);
    m.isize = rd32(&raw[off + bsize + 1 - 4]);
    m.crc = rd32(&raw[off + bsize + 1 - 8]);
    if (m.isize > 65536) {  // a BGZF block holds at most 64 KiB (SAM spec 4.1)
      err = "BGZF member with ISIZE > 64 KiB in " + path;
      return false;
    }
    m.out_off = total;
    total += m.isize;
    mem.push_back(m);
    off += (size_t)bsize + 1;
  }
  if (at_eof && off != n) {
    err = "truncated BGZF member at the end of " + path;
    return false;
  }
  *consumed = off;
  out.resize(total);
  std::atomic<size_t> next{0};
  std::atomic<bool> ok{true};
  auto work = [&]() {
    for (;;) {
      size_t i = next.fetch_add(8);  // (a few members per claim: the counter is shared by every worker)
      if (i >= mem.size() || !ok.load()) break;
      for (size_t j = i; j < i + 8 && j < mem.size(); ++j) {
        const Member& m = mem[j];
        if (m.isize == 0) continue;
        if (!inflate_raw(raw + m.cdata, m.clen, out.data() + m.out_off, m.isize)) ok = false;
        // the member's CRC32 covers the uncompressed bytes (RFC 1952); htslib rejects a mismatch, so do we
        else if (bgzf_crc32(out.data() + m.out_off, m.isize) != m.crc) ok = false;
      }
    }
  };
  if (threads <= 1 || mem.size() < 8) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
  if (!ok) {
    err = "inflate failed or CRC32 mismatch in " + path;
    return false;
  }
  return true;
}

bool bgzf_read_file(const std::string& path, std::vector<uint8_t>& out, std::string& err, int threads) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  std::vector<uint8_t> raw;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  raw.resize(sz > 0 ? (size_t)sz : 0);
  if (sz > 0 && fread(raw.data(), 1, raw.size(), f) != raw.size()) {
    fclose(f);
    err = "short read on " + path;
    return false;
  }
  fclose(f);
  out.clear();
  size_t used = 0;
  return bgzf_inflate_chunk(raw.data(), raw.size(), true, out, &used, err, threads, path);
}

BgzfWriter::~BgzfWriter() {
  if (f_) close();
}

bool BgzfWriter::open(const std::string& path, int level, int threads) {
  level_ = level;
  if (threads <= 0) threads = cpu_budget();
  threads_ = threads < 1 ? 1 : (threads > 128 ? 128 : threads);
  if (path == "-") {
    f_ = stdout;
    own_ = false;
  } else {
    f_ = fopen(path.c_str(), "wb");
    own_ = true;
  }
  if (!f_) {
    err_ = "cannot create " + path;
    return false;
  }
  return true;
}

// deflate one <=0xff00-byte block into a complete BGZF member
static bool deflate_member(const uint8_t* src, size_t n, int level, std::vector<uint8_t>& out) {
  out.resize(0x10000 + 64);
  const size_t clen = deflate_raw(src, n, level, out.data() + 18, out.size() - 18 - 8);
  if (clen == 0) return false;
  size_t bsize = clen + 25;  // total member length - 1
  const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 0xff), (uint8_t)(bsize >> 8)};
  memcpy(out.data(), hdr, 18);
  uint32_t crc = bgzf_crc32(src, n);
  uint32_t isz = (uint32_t)n;
  uint8_t* t = out.data() + 18 + clen;
  for (int k = 0; k < 4; ++k) t[k] = (uint8_t)(crc >> (8 * k));
  for (int k = 0; k < 4; ++k) t[4 + k] = (uint8_t)(isz >> (8 * k));
  out.resize(18 + clen + 8);
  return true;
}

// src -> whole BGZF members appended to `out` (any byte run may be cut into members anywhere: they are independent streams)
bool bgzf_deflate_members(const uint8_t* src, size_t n, int level, std::vector<uint8_t>& out) {
  std::vector<uint8_t> m;
  for (size_t off = 0; off < n; off += kBlock) {
    const size_t len = n - off < kBlock ? n - off : kBlock;
    if (!deflate_member(src + off, len, level, m)) return false;
    out.insert(out.end(), m.begin(), m.end());
  }
  return true;
}

bool BgzfWriter::write_members(const uint8_t* z, size_t n) {
  if (!flush_chunk() || !wait_bg()) return false;  // what was written uncompressed so far goes out first, as its own members
  if (n && fwrite(z, 1, n, f_) != n) {
    err_ = "write failed";
    return false;
  }
  return true;
}

bool BgzfWriter::wait_bg() {
  if (bg_) {
    bg_->join();
    delete bg_;
    bg_ = nullptr;
  }
  if (!bg_ok_ && err_.empty()) err_ = "deflate or write failed";
  return bg_ok_;
}

// The chunk is deflated block-parallel and written by a background thread while the caller keeps producing: one chunk
// in flight (the member sequence is the same as with synchronous flushing).
bool BgzfWriter::flush_chunk() {
  if (!wait_bg()) return false;
  if (buf_.empty()) return true;
  bg_buf_.swap(buf_);
  buf_.clear();
  bg_ = new std::thread([this]() {
    const std::vector<uint8_t>& src = bg_buf_;
    size_t nblk = (src.size() + kBlock - 1) / kBlock;
    std::vector<std::vector<uint8_t>> outs(nblk);
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= nblk) break;
        size_t off = i * kBlock;
        size_t n = src.size() - off < kBlock ? src.size() - off : kBlock;
        if (!deflate_member(src.data() + off, n, level_, outs[i])) ok = false;
      }
    };
    int nt = (int)(nblk < (size_t)threads_ ? nblk : (size_t)threads_);
    if (nt <= 1) {
      work();
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back(work);
      for (auto& t : th) t.join();
    }
    bool good = ok;
    for (auto& o : outs)
      if (good && fwrite(o.data(), 1, o.size(), f_) != o.size()) good = false;
    if (!good) bg_ok_ = false;
    bg_buf_.clear();
  });
  return true;
}

bool BgzfWriter::write(const void* p, size_t n) {
  const uint8_t* s = (const uint8_t*)p;
  buf_.insert(buf_.end(), s, s + n);
  // cut on a block boundary so that the member sequence does not depend on the chunking
  if (buf_.size() >= chunk_) {
    size_t whole = (buf_.size() / kBlock) * kBlock;
    std::vector<uint8_t> tail(buf_.begin() + whole, buf_.end());
    buf_.resize(whole);
    if (!flush_chunk()) return false;
    buf_.swap(tail);
  }
  return true;
}

bool BgzfWriter::close() {
  if (!f_) return true;
  bool ok = flush_chunk();
  ok = wait_bg() && ok;
  ok = ok && fwrite(kEof, 1, sizeof(kEof), f_) == sizeof(kEof);
  if (own_)
    ok = (fclose(f_) == 0) && ok;
  else
    fflush(f_);
  f_ = nullptr;
  return ok;
}

}  // namespace tbh
