// bgzf.h — minimal BGZF codec over zlib (htslib is not available in this image).
// Host side only: BGZF inflate/deflate stays on the CPU by design (north_star).
#pragma once
#include <stdint.h>

#include <string>
#include <thread>
#include <vector>

namespace tbh {

// Inflate a whole BGZF file into `out`.  Blocks are independent, so they are inflated by
// `threads` workers.  Returns false and fills `err` on malformed input.
bool bgzf_read_file(const std::string& path, std::vector<uint8_t>& out, std::string& err, int threads = 1);

// Streaming form: inflate the whole members inside raw[0, n), appending to `out`; *consumed = compressed bytes used (a member
// cut off by the end of the buffer waits for the next call unless at_eof).
bool bgzf_inflate_chunk(const uint8_t* raw, size_t n, bool at_eof, std::vector<uint8_t>& out, size_t* consumed, std::string& err,
                        int threads, const std::string& path);

// One member's deflate stream (cdata[0, clen)) -> exactly isize bytes at `out`, CRC32 checked against the member's trailer
bool bgzf_inflate_member(const uint8_t* cdata, size_t clen, uint8_t* out, uint32_t isize, uint32_t crc);

// CRC32 (RFC 1952) of a byte run
uint32_t bgzf_crc32(const uint8_t* p, size_t n);

// Deflate a byte run into whole BGZF members (<= 0xff00 payload bytes each), appended to `out`.
bool bgzf_deflate_members(const uint8_t* src, size_t n, int level, std::vector<uint8_t>& out);

// Worker threads worth starting: the hardware concurrency, cut down to the cgroup CPU quota when there is one (a container
// may show 256 cores and be allowed 16: more threads than that only add throttling).
int cpu_budget();

// True when the file starts with a BGZF member (gzip magic + BC extra field).
bool bgzf_probe(const std::string& path);

// BGZF members are independent deflate streams: the writer buffers up to `chunk` bytes of payload, then deflates the
// 0xff00-byte blocks of that chunk on `threads` workers and writes them out in order.
class BgzfWriter {
 public:
  BgzfWriter() = default;
  ~BgzfWriter();
  bool open(const std::string& path, int level = 6, int threads = 0 /*0 = hardware concurrency, capped at 32*/);
  bool write(const void* p, size_t n);
  // already deflated members (bgzf_deflate_members), e.g. produced by worker threads: written as they are, in call order
  bool write_members(const uint8_t* z, size_t n);
  int level() const { return level_; }
  bool close();  // flushes and appends the 28-byte EOF block
  const std::string& error() const { return err_; }

 private:
  bool flush_chunk();   // hands buf_ to the background compressor (after waiting for the previous hand-off)
  bool wait_bg();       // joins the background compressor; false if it failed
  FILE* f_ = nullptr;
  bool own_ = false;
  int level_ = 6;
  int threads_ = 1;
  size_t chunk_ = (size_t)16 << 20;
  std::vector<uint8_t> buf_;
  std::vector<uint8_t> bg_buf_;   // chunk being deflated + written while the caller fills buf_ again
  std::thread* bg_ = nullptr;
  bool bg_ok_ = true;
  std::string err_;
};

}  // namespace tbh
