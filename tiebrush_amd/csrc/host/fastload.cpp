#include "fastload.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include "bam.h"
#include "bgzf.h"

namespace tbh {

namespace {
inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct Member {
  uint32_t f;
  size_t cdata, clen;
  uint32_t isize, crc;
  size_t out_off;
  // the member's own record walk, made right behind its inflate (below): records kept, their CIGAR operations, where the
  // member's offsets wait in the file's slot array, what the walk met
  uint32_t nrec = 0, flags = 0;
  uint64_t ncig = 0, slot0 = 0;
  int32_t max_tid = -1;
};
enum : uint32_t { MW_BAD = 1u, MW_UNPLACED = 2u, MW_PLACED = 4u, MW_PLACED_AFTER_UNPLACED = 8u };

template <class F>
void parallel(int threads, F f) {
  if (threads <= 1) {
    f();
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t) th.emplace_back(f);
  for (auto& x : th) x.join();
}
double ms_since(std::chrono::steady_clock::time_point a) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
}
}  // namespace

namespace {
std::mutex g_big_m;
std::vector<std::pair<char*, size_t>> g_big;  // the large blocks handed out (never freed one by one)
}  // namespace

void* big_alloc(size_t bytes) {
  if (bytes < ((size_t)4 << 20)) return malloc(bytes ? bytes : 1);
  void* p = nullptr;
  const size_t sz = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  if (posix_memalign(&p, (size_t)2 << 20, sz) != 0) return nullptr;
  (void)madvise(p, sz, MADV_HUGEPAGE);
  std::lock_guard<std::mutex> lk(g_big_m);
  g_big.emplace_back((char*)p, sz);
  return p;
}

void big_free(void* q) {
  if (!q) return;
  {
    std::lock_guard<std::mutex> lk(g_big_m);
    for (size_t i = 0; i < g_big.size(); ++i)
      if (g_big[i].first == (char*)q) {
        g_big[i] = g_big.back();
        g_big.pop_back();
        break;
      }
  }
  free(q);  // (both kinds of block come from the C allocator: malloc / posix_memalign)
}

void big_release_all(int threads) {
  std::vector<std::pair<char*, size_t>> work;
  {
    std::lock_guard<std::mutex> lk(g_big_m);
    constexpr size_t kSlice = (size_t)32 << 20;
    for (auto& b : g_big)
      for (size_t o = 0; o < b.second; o += kSlice) work.emplace_back(b.first + o, std::min(kSlice, b.second - o));
    g_big.clear();
  }
  std::atomic<size_t> next{0};
  parallel(std::max(1, threads), [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= work.size()) break;
      (void)madvise(work[i].first, work[i].second, MADV_DONTNEED);
    }
  });
}

FastTile::~FastTile() {}  // (big_alloc blocks: released all at once, big_release_all, or with the process)

tbk_soa_in FastTile::view() const {
  tbk_soa_in v;
  memset(&v, 0, sizeof(v));
  v.mem = TBK_MEM_HOST;
  v.n_files = (uint32_t)in.size();
  v.n_records = (uint32_t)n;
  v.n_cigar_ops = (uint32_t)n_cig;
  v.file_off = file_off.data();
  v.tbmerged = tbmerged.data();
  v.tid = tid;
  v.pos = pos;
  v.flag = flag;
  v.mapq = mapq;
  v.strand = strand;
  v.nh = nh;
  v.cig_off = cig_off;
  v.cig = cig;
  v.yc_in = yc_in;
  v.yx_in = yx_in;
  v.yd_in = yd_in;
  return v;
}

const uint8_t* FastTile::record(uint32_t g, uint32_t* len) const {
  const size_t f = (size_t)(std::upper_bound(file_off.begin(), file_off.end(), g) - file_off.begin()) - 1;
  const uint8_t* p = in[f].data + in[f].rec_off[g - file_off[f]];
  *len = rd32(p);
  return p + 4;
}

bool fast_load(const std::vector<std::string>& paths, const std::vector<uint8_t>& tbm, int threads, size_t mem_budget, FastTile& t, bool* fits,
               std::string& err) {
  const size_t k = paths.size();
  *fits = true;
  t.in.resize(k);
  t.tbmerged = tbm;
  std::mutex em;
  std::atomic<bool> ok{true};
  auto fail = [&](const std::string& e) {
    std::lock_guard<std::mutex> lk(em);
    if (ok.exchange(false)) err = e;
  };
  if (threads < 1) threads = 1;
  auto t0 = std::chrono::steady_clock::now();
  // ---- 1. the files as they lie on disk; their BGZF members (a walk over the member headers) ----
  std::vector<std::vector<Member>> mem(k);
  {
    std::atomic<size_t> nf{0};
    parallel(std::min<int>(threads, (int)k), [&]() {
      for (;;) {
        const size_t f = nf.fetch_add(1);
        if (f >= k || !ok) break;
        FastTile::In& I = t.in[f];
        I.path = paths[f];
        // mapped, not read: the inflate tasks take the compressed bytes from the page cache where they lie (a read() into a buffer of
        // the process's own is one more pass over the file, on the loader's critical path)
        {
          const int fd = open(paths[f].c_str(), O_RDONLY);
          struct stat st;
          if (fd < 0 || fstat(fd, &st) != 0) {
            if (fd >= 0) close(fd);
            fail("cannot open " + paths[f]);
            break;
          }
          I.comp_n = (size_t)st.st_size;
          I.comp = nullptr;
          if (I.comp_n) {
            void* q = mmap(nullptr, I.comp_n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (q == MAP_FAILED) {
              close(fd);
              fail("cannot map " + paths[f]);
              break;
            }
            (void)madvise(q, I.comp_n, MADV_SEQUENTIAL);
            I.comp = (const uint8_t*)q;
          }
          close(fd);
        }
        const uint8_t* raw = I.comp;
        const size_t n = I.comp_n;
        size_t off = 0, total = 0;
        while (off < n) {
          if (off + 18 > n || raw[off] != 0x1f || raw[off + 1] != 0x8b || raw[off + 2] != 8 || !(raw[off + 3] & 4)) {
            fail("not a BGZF member in " + paths[f]);
            break;
          }
          const uint16_t xlen = rd16(&raw[off + 10]);
          size_t p = off + 12;
          const size_t end = p + xlen;
          if (end > n) {
            fail("truncated BGZF member at the end of " + paths[f]);
            break;
          }
          int bsize = -1;
          while (p + 4 <= end) {
            const uint16_t slen = rd16(&raw[p + 2]);
            if (p + 4 + slen > end) break;
            if (raw[p] == 'B' && raw[p + 1] == 'C' && slen == 2) bsize = rd16(&raw[p + 4]);
            p += 4 + slen;
          }
          if (bsize < 0 || (size_t)bsize + 1 < (size_t)12 + xlen + 8 || off + (size_t)bsize + 1 > n) {
            fail("corrupt or truncated BGZF member in " + paths[f]);
            break;
          }
          Member m;
          m.f = (uint32_t)f;
          m.cdata = off + 12 + xlen;
          m.clen = (size_t)bsize + 1 - 8 - (12 + xlen);
          m.isize = rd32(&raw[off + bsize + 1 - 4]);
          m.crc = rd32(&raw[off + bsize + 1 - 8]);
          if (m.isize > 65536) {
            fail("BGZF member with ISIZE > 64 KiB in " + paths[f]);
            break;
          }
          m.out_off = total;
          total += m.isize;
          mem[f].push_back(m);
          off += (size_t)bsize + 1;
        }
        I.data_n = total;
      }
    });
    if (!ok) return false;
  }
  size_t total = 0, nmem = 0;
  for (size_t f = 0; f < k; ++f) {
    total += t.in[f].data_n;
    nmem += mem[f].size();
  }
  if (total + total / 2 > mem_budget) {  // the inflated streams + the tile + the output would not fit: the streaming path takes it
    *fits = false;
    for (auto& I : t.in)
      if (I.comp) munmap(const_cast<uint8_t*>(I.comp), I.comp_n);
    t.in.clear();
    return true;
  }
  t.ms_read = ms_since(t0);
  t0 = std::chrono::steady_clock::now();
  // ---- 2. inflate: every member of every input is a task of its own (any number of inputs keeps every worker busy) ----
  // The record index rides with the inflate.  htslib's writers start a new BGZF member whenever the next record does not fit
  // (bgzf_flush_try in bam_write1) and end the header with a flush, so in nearly every file member 0 is the header and every other
  // member is a whole number of records: the task that has just inflated a member walks its records while they are in cache — a walk over
  // a whole file later is a chain of cache misses, one per record — and leaves their offsets in the member's own stretch of a slot
  // array.  A file where a walk does not end exactly at its member's end (or whose header is not member 0) takes the file-wide walk below,
  // which also decides what is an error.
  std::vector<Member*> all;
  all.reserve(nmem);
  std::vector<uint64_t*> slots(k, nullptr);
  for (size_t f = 0; f < k; ++f) {
    t.in[f].data = (uint8_t*)big_alloc(t.in[f].data_n);
    if (!t.in[f].data) {
      err = "out of memory";
      return false;
    }
    size_t mi = 0;
    for (Member& m : mem[f]) {
      m.slot0 = m.out_off / 36 + mi++;  // (a record takes 36 bytes at least: the stretches cannot meet)
      all.push_back(&m);
    }
    slots[f] = (uint64_t*)big_alloc((t.in[f].data_n / 36 + mem[f].size() + 1) * sizeof(uint64_t));
    if (!slots[f]) {
      err = "out of memory";
      return false;
    }
  }
  auto walk_member = [&](Member& m) {
    if (m.out_off == 0) return;  // (member 0: the header, looked at below)
    const FastTile::In& I = t.in[m.f];
    const uint8_t* d = I.data + m.out_off;
    const size_t n = m.isize;
    uint64_t* out = slots[m.f] + m.slot0;
    size_t off = 0;
    uint32_t nrec = 0, flags = 0;
    uint64_t ncig = 0;
    int32_t mx = -1;
    while (off + 4 <= n) {
      const uint32_t bs = rd32(d + off);
      if (bs < 32 || off + 4 + (size_t)bs > n) {
        flags |= MW_BAD;
        break;
      }
      const uint8_t* r = d + off + 4;
      const int32_t tid = (int32_t)rd32(r);
      const uint32_t l_read_name = r[8];
      const uint32_t n_cigar = (uint32_t)r[12] | ((uint32_t)r[13] << 8);
      const int32_t l_seq = (int32_t)rd32(r + 16);
      const int32_t mtid = (int32_t)rd32(r + 20);
      const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + (l_seq < 0 ? 0ull : ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq);
      const bool name_ok = l_read_name >= 1 && 32 + l_read_name <= bs && r[32 + l_read_name - 1] == 0;
      if (l_seq < 0 || need > bs || !name_ok || tid < -1 || mtid < -1) {
        flags |= MW_BAD;  // (the file-wide walk names the record)
        break;
      }
      mx = std::max(mx, std::max(tid, mtid));
      if (tid < 0) {
        flags |= MW_UNPLACED;
      } else {
        flags |= MW_PLACED | ((flags & MW_UNPLACED) ? MW_PLACED_AFTER_UNPLACED : 0u);
        out[nrec++] = m.out_off + off;
        ncig += n_cigar;
      }
      off += 4 + (size_t)bs;
    }
    if (off != n) flags |= MW_BAD;
    m.nrec = nrec, m.ncig = ncig, m.flags = flags, m.max_tid = mx;
  };
  {
    std::atomic<size_t> next{0};
    parallel(threads, [&]() {
      for (;;) {
        const size_t i = next.fetch_add(16);
        if (i >= all.size() || !ok) break;
        for (size_t j = i; j < i + 16 && j < all.size(); ++j) {
          Member& m = *all[j];
          if (!m.isize) continue;
          FastTile::In& I = t.in[m.f];
          if (!bgzf_inflate_member(I.comp + m.cdata, m.clen, I.data + m.out_off, m.isize, m.crc)) fail("inflate failed or CRC32 mismatch in " + I.path);
          else walk_member(m);
        }
      }
    });
    if (!ok) return false;
  }
  for (size_t f = 0; f < k; ++f) {
    if (t.in[f].comp) munmap(const_cast<uint8_t*>(t.in[f].comp), t.in[f].comp_n);
    t.in[f].comp = nullptr;
  }
  t.ms_inflate = ms_since(t0);
  t0 = std::chrono::steady_clock::now();
  // ---- 3. index: the records of every input (field lengths checked as bam_read1 checks them), unplaced reads dropped ----
  std::atomic<size_t> n_fused{0};
  std::vector<uint8_t> fused_file(k, 0);
  {
    std::atomic<size_t> nf{0};
    parallel(std::min<int>(threads, (int)k), [&]() {
      for (;;) {
        const size_t f = nf.fetch_add(1);
        if (f >= k || !ok) break;
        FastTile::In& I = t.in[f];
        const uint8_t* d = I.data;
        const size_t n = I.data_n;
        if (n < 12 || memcmp(d, "BAM\1", 4) != 0) {
          fail("not a BAM stream (" + I.path + ")");
          break;
        }
        const uint32_t l_text = rd32(d + 4);
        size_t p = 8 + (size_t)l_text;
        if (p + 4 > n) {
          fail("truncated BAM header (" + I.path + ")");
          break;
        }
        const int32_t n_targets = (int32_t)rd32(d + p);
        p += 4;
        bool hdr_ok = true;
        for (int32_t i = 0; i < n_targets && hdr_ok; ++i) {
          if (p + 4 > n) hdr_ok = false;
          else {
            const uint32_t l_name = rd32(d + p);
            if (p + 8 + l_name > n) hdr_ok = false;
            p += 8 + (size_t)l_name;
          }
        }
        if (!hdr_ok) {
          fail("truncated reference list (" + I.path + ")");
          break;
        }
        // the members' own walks (phase 2) are the file's index when member 0 is exactly the header and every other walk ended at its
        // member's end with well-formed records, the reference ids inside the header's table and no placed read behind an unplaced one
        {
          std::vector<Member>& M = mem[f];
          bool fused = !M.empty() && M[0].out_off == 0 && (size_t)M[0].isize == p;
          bool seen_unplaced = false;
          uint64_t nrec = 0, ncig = 0;
          for (size_t i = 1; fused && i < M.size(); ++i) {
            const Member& m = M[i];
            if ((m.flags & (MW_BAD | MW_PLACED_AFTER_UNPLACED)) || m.max_tid >= n_targets || (seen_unplaced && (m.flags & MW_PLACED))) fused = false;
            seen_unplaced = seen_unplaced || (m.flags & MW_UNPLACED);
            nrec += m.nrec, ncig += m.ncig;
          }
          if (fused) {
            I.rec_off = (uint64_t*)big_alloc((nrec + 1) * sizeof(uint64_t));
            if (!I.rec_off) {
              fail("out of memory");
              break;
            }
            uint64_t at = 0;
            for (size_t i = 1; i < M.size(); ++i) {  // (where the member's offsets go: the copies are made below, by every worker)
              M[i].ncig = at;
              at += M[i].nrec;
            }
            I.n_rec = (size_t)nrec;
            I.n_cig = ncig;
            n_fused.fetch_add(1);
            fused_file[f] = 1;
            continue;
          }
        }
        // (a record takes at least 36 bytes: the index can be sized before the walk)
        I.rec_off = (uint64_t*)big_alloc(((n - p) / 36 + 1) * sizeof(uint64_t));
        if (!I.rec_off) {
          fail("out of memory");
          break;
        }
        size_t nrec = 0;
        uint64_t ncig = 0;
        bool unplaced = false;
        size_t off = p, idx = 0;
        while (off + 4 <= n) {
          const uint32_t bs = rd32(d + off);
          if (bs < 32 || off + 4 + (size_t)bs > n) {
            fail((bs < 32 ? "corrupt record in " : "truncated record at the end of ") + I.path);
            break;
          }
          const uint8_t* r = d + off + 4;
          const int32_t tid = (int32_t)rd32(r);
          const uint32_t l_read_name = r[8];
          const uint32_t n_cigar = (uint32_t)r[12] | ((uint32_t)r[13] << 8);
          const int32_t l_seq = (int32_t)rd32(r + 16);
          const int32_t mtid = (int32_t)rd32(r + 20);
          const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + (l_seq < 0 ? 0ull : ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq);
          const bool name_ok = l_read_name >= 1 && 32 + l_read_name <= bs && r[32 + l_read_name - 1] == 0;
          if (l_seq < 0 || need > bs || !name_ok || tid < -1 || tid >= n_targets || mtid < -1 || mtid >= n_targets) {
            fail("malformed record " + std::to_string(idx) + " in " + I.path + " (field lengths / reference id outside the record / header)");
            break;
          }
          // unplaced reads (refID -1) sort last and belong to no tile (the GPU build always drops them, tiebrush.cpp:535)
          if (tid < 0) {
            unplaced = true;
          } else {
            if (unplaced) {
              fail(I.path + " file not coordinate-sorted!");
              break;
            }
            I.rec_off[nrec++] = off;
            ncig += n_cigar;
          }
          off += 4 + (size_t)bs;
          ++idx;
        }
        if (ok && off != n) fail("truncated record at the end of " + I.path);
        I.n_cig = ncig;
        I.n_rec = nrec;
      }
    });
    if (!ok) return false;
  }
  {  // the members' offsets into their files' index: tasks of 64 members, whatever the number of inputs (the slot arrays are left to
     // the process's end like the other large blocks: returning tens of megabytes per input page by page is not worth a core's time here)
    std::vector<std::pair<uint32_t, uint32_t>> tasks;  // (file, first member)
    for (size_t f = 0; f < k; ++f)
      if (fused_file[f])
        for (size_t i = 1; i < mem[f].size(); i += 64) tasks.emplace_back((uint32_t)f, (uint32_t)i);
    std::atomic<size_t> nt{0};
    parallel(threads, [&]() {
      for (;;) {
        const size_t q = nt.fetch_add(1);
        if (q >= tasks.size()) break;
        const uint32_t f = tasks[q].first;
        const std::vector<Member>& M = mem[f];
        for (size_t i = tasks[q].second; i < M.size() && i < (size_t)tasks[q].second + 64; ++i)
          memcpy(t.in[f].rec_off + M[i].ncig, slots[f] + M[i].slot0, (size_t)M[i].nrec * sizeof(uint64_t));
      }
    });
  }
  t.file_off.assign(k + 1, 0);
  std::vector<uint64_t> cig_base(k + 1, 0);
  for (size_t f = 0; f < k; ++f) {
    if ((uint64_t)t.file_off[f] + t.in[f].n_rec >= (1ull << 32)) {
      err = "more than 2^32 records in one tile";
      return false;
    }
    t.file_off[f + 1] = t.file_off[f] + (uint32_t)t.in[f].n_rec;
    cig_base[f + 1] = cig_base[f] + t.in[f].n_cig;
  }
  if (cig_base[k] >= (1ull << 32)) {
    err = "more than 2^32 CIGAR operations in one tile";
    return false;
  }
  t.n = t.file_off[k];
  t.n_cig = (size_t)cig_base[k];
  t.ms_index = ms_since(t0);
  t.n_fused = n_fused.load();
  t0 = std::chrono::steady_clock::now();
  // ---- 4. the tile: tasks of consecutive records of one input, every worker busy whatever the number of inputs ----
  const size_t n = t.n;
  bool any_tb = false;
  for (size_t f = 0; f < k; ++f) any_tb |= tbm[f] != 0;
  auto alloc = [&](size_t bytes) { return big_alloc(bytes); };
  t.tid = (int32_t*)alloc(n * 4);
  t.pos = (int32_t*)alloc(n * 4);
  t.nh = (int32_t*)alloc(n * 4);
  t.flag = (uint16_t*)alloc(n * 2);
  t.mapq = (uint8_t*)alloc(n);
  t.strand = (uint8_t*)alloc(n);
  t.cig_off = (uint32_t*)alloc((n + 1) * 4);
  t.cig = (uint32_t*)alloc(t.n_cig * 4);
  if (any_tb) {
    t.yc_in = (double*)alloc(n * 8);
    t.yx_in = (int64_t*)alloc(n * 8);
    t.yd_in = (int64_t*)alloc(n * 8);
  }
  if (!t.tid || !t.pos || !t.nh || !t.flag || !t.mapq || !t.strand || !t.cig_off || !t.cig || (any_tb && (!t.yc_in || !t.yx_in || !t.yd_in))) {
    err = "out of memory";
    return false;
  }
  t.cig_off[n] = (uint32_t)t.n_cig;
  constexpr size_t kTask = (size_t)1 << 15;
  struct Task {
    uint32_t f;
    size_t a, b;
  };
  std::vector<Task> tasks;
  for (size_t f = 0; f < k; ++f)
    for (size_t a = 0; a < t.in[f].n_rec; a += kTask) tasks.push_back(Task{(uint32_t)f, a, std::min(t.in[f].n_rec, a + kTask)});
  // CIGAR base of every task: the n_cigar fields of the task's records (a pass over one cache line per record)
  std::vector<uint64_t> tcig(tasks.size() + 1, 0);
  {
    std::atomic<size_t> nt{0};
    parallel(threads, [&]() {
      for (;;) {
        const size_t ti = nt.fetch_add(1);
        if (ti >= tasks.size()) break;
        const Task& T = tasks[ti];
        const FastTile::In& I = t.in[T.f];
        uint64_t c = 0;
        for (size_t i = T.a; i < T.b; ++i) {
          const uint8_t* r = I.data + I.rec_off[i] + 4;
          c += (uint32_t)r[12] | ((uint32_t)r[13] << 8);
        }
        tcig[ti + 1] = c;
      }
    });
    for (size_t ti = 0; ti < tasks.size(); ++ti) tcig[ti + 1] += tcig[ti];
  }
  {
    std::atomic<size_t> nt{0};
    parallel(threads, [&]() {
      for (;;) {
        const size_t ti = nt.fetch_add(1);
        if (ti >= tasks.size()) break;
        const Task& T = tasks[ti];
        const FastTile::In& I = t.in[T.f];
        const bool tb = tbm[T.f] != 0;
        uint64_t co = tcig[ti];
        size_t g = t.file_off[T.f] + T.a;
        for (size_t i = T.a; i < T.b; ++i, ++g) {
          const uint8_t* p0 = I.data + I.rec_off[i];
          RecView v;
          v.p = p0 + 4;
          v.len = rd32(p0);
          t.tid[g] = v.tid();
          t.pos[g] = v.pos();
          const uint16_t fl = v.flag();
          t.flag[g] = fl;
          t.mapq[g] = v.mapq();
          t.cig_off[g] = (uint32_t)co;
          const uint32_t nc = v.n_cigar();
          memcpy(t.cig + co, v.cigar_bytes(), (size_t)nc * 4);
          co += nc;
          // one aux scan; bam_aux_get semantics = first occurrence of each tag
          char xs = 0, ts = 0;
          int32_t nh = TBK_NH_ABSENT;
          unsigned seen = 0;
          const uint8_t* a = v.aux_begin();
          const uint8_t* e = v.aux_end();
          if (tb) {
            t.yc_in[g] = 0.0;
            t.yx_in[g] = 1;
            t.yd_in[g] = 0;
          }
          while (a + 3 <= e) {
            const size_t sz = aux_field_size(a, e);
            if (!sz) break;
            const uint8_t* s = a + 2;
            if (a[0] == 'N' && a[1] == 'H' && !(seen & 1)) {
              seen |= 1;
              nh = (int32_t)aux2i(s);
            } else if (a[0] == 'X' && a[1] == 'S' && !(seen & 2)) {
              seen |= 2;
              xs = (*s == 'A' || *s == 'Z') ? (char)s[1] : 0;
            } else if (a[0] == 't' && a[1] == 's' && !(seen & 4)) {
              seen |= 4;
              ts = (*s == 'A' || *s == 'Z') ? (char)s[1] : 0;
            } else if (tb && a[0] == 'Y' && a[1] == 'C' && !(seen & 8)) {
              seen |= 8;
              t.yc_in[g] = aux2f(s);
            } else if (tb && a[0] == 'Y' && a[1] == 'X' && !(seen & 16)) {
              seen |= 16;
              t.yx_in[g] = aux2i(s);
            } else if (tb && a[0] == 'Y' && a[1] == 'D' && !(seen & 32)) {
              seen |= 32;
              t.yd_in[g] = aux2i(s);
            }
            a += sz;
          }
          t.nh[g] = nh;
          char c = xs;  // GSamRecord::spliceStrand (GSam.cpp:464-475)
          if (c == 0 && (ts == '+' || ts == '-')) c = (fl & 0x10) ? (ts == '+' ? '-' : '+') : ts;
          t.strand[g] = (uint8_t)((c == '+' || c == '-') ? c : '.');
        }
      }
    });
  }
  t.ms_soa = ms_since(t0);
  return true;
}

}  // namespace tbh
