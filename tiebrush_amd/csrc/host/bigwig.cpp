// bigwig.cpp — see bigwig.h.  All integers little-endian (the writer's magic says so).
#include "bigwig.h"

#include <math.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>

namespace tbh {

namespace {
constexpr uint32_t BW_MAGIC = 0x888FFC26u, BPT_MAGIC = 0x78CA8C91u, CIR_MAGIC = 0x2468ACE0u;
constexpr uint32_t ITEMS_PER_SLOT = 1024, BLOCK = 256;
constexpr int MAX_ZOOM = 10;

struct Buf {
  std::vector<uint8_t> d;
  void u8(uint32_t v) { d.push_back((uint8_t)v); }
  void u16(uint32_t v) {
    d.push_back((uint8_t)v);
    d.push_back((uint8_t)(v >> 8));
  }
  void u32(uint32_t v) {
    for (int i = 0; i < 4; ++i) d.push_back((uint8_t)(v >> (8 * i)));
  }
  void u64(uint64_t v) {
    for (int i = 0; i < 8; ++i) d.push_back((uint8_t)(v >> (8 * i)));
  }
  void f32(float v) {
    uint32_t b;
    memcpy(&b, &v, 4);
    u32(b);
  }
  void f64(double v) {
    uint64_t b;
    memcpy(&b, &v, 8);
    u64(b);
  }
};
bool put(FILE* f, const std::vector<uint8_t>& d) { return d.empty() || fwrite(d.data(), 1, d.size(), f) == d.size(); }
}  // namespace

bool BigWigWriter::open(const std::string& path, const std::vector<std::string>& names, const std::vector<uint32_t>& lens, std::string& err) {
  path_ = path;
  names_ = names;
  lens_ = lens;
  f_ = fopen(path.c_str(), "wb");
  if (!f_) {
    err = "cannot create " + path;
    return false;
  }
  return true;
}

void BigWigWriter::add(uint32_t chrom, uint32_t start, uint32_t end, float value) { items_.push_back(Item{chrom, start, end, value}); }

// one compressed section; remembers its place for the index
bool BigWigWriter::write_sections(const std::vector<uint8_t>& payload, uint32_t chrom, uint32_t start, uint32_t end, std::vector<Block>& blocks) {
  uLongf clen = compressBound((uLong)payload.size());
  std::vector<uint8_t> z(clen);
  if (compress2(z.data(), &clen, payload.data(), (uLong)payload.size(), Z_DEFAULT_COMPRESSION) != Z_OK) return false;
  const uint64_t off = (uint64_t)ftello(f_);
  if (fwrite(z.data(), 1, clen, f_) != clen) return false;
  blocks.push_back(Block{chrom, start, end, off, (uint64_t)clen});
  if (payload.size() > max_uncompressed_) max_uncompressed_ = (uint32_t)payload.size();
  return true;
}

// R-tree over the sections, bottom-up: leaves of up to BLOCK sections, parents of up to BLOCK children; written root first
bool BigWigWriter::write_index(const std::vector<Block>& blocks, uint64_t* index_off) {
  struct Node {
    uint32_t c0, s0, c1, e1;  // bounding range: (start chrom, start base) .. (end chrom, end base)
    size_t first, count;      // children (index into the level below, or into blocks for leaves)
    uint64_t off = 0;
  };
  auto bound = [](Node& n, uint32_t c0, uint32_t s0, uint32_t c1, uint32_t e1, bool first) {
    if (first) {
      n.c0 = c0;
      n.s0 = s0;
      n.c1 = c1;
      n.e1 = e1;
      return;
    }
    if (c0 < n.c0 || (c0 == n.c0 && s0 < n.s0)) {
      n.c0 = c0;
      n.s0 = s0;
    }
    if (c1 > n.c1 || (c1 == n.c1 && e1 > n.e1)) {
      n.c1 = c1;
      n.e1 = e1;
    }
  };
  std::vector<std::vector<Node>> levels;  // levels[0] = leaves
  {
    std::vector<Node> leaves;
    for (size_t i = 0; i < blocks.size(); i += BLOCK) {
      Node n{};
      n.first = i;
      n.count = std::min<size_t>(BLOCK, blocks.size() - i);
      for (size_t j = 0; j < n.count; ++j) bound(n, blocks[i + j].chrom, blocks[i + j].start, blocks[i + j].chrom, blocks[i + j].end, j == 0);
      leaves.push_back(n);
    }
    if (leaves.empty()) {
      Node n{};
      leaves.push_back(n);
    }
    levels.push_back(leaves);
  }
  while (levels.back().size() > 1) {
    const std::vector<Node>& below = levels.back();
    std::vector<Node> up;
    for (size_t i = 0; i < below.size(); i += BLOCK) {
      Node n{};
      n.first = i;
      n.count = std::min<size_t>(BLOCK, below.size() - i);
      for (size_t j = 0; j < n.count; ++j) bound(n, below[i + j].c0, below[i + j].s0, below[i + j].c1, below[i + j].e1, j == 0);
      up.push_back(n);
    }
    levels.push_back(up);
  }
  *index_off = (uint64_t)ftello(f_);
  const Node& root = levels.back()[0];
  // node offsets: root first, then each level below in order
  uint64_t off = *index_off + 48;
  for (size_t L = levels.size(); L-- > 0;) {
    for (Node& n : levels[L]) {
      n.off = off;
      off += 4 + (uint64_t)n.count * (L == 0 ? 32 : 24);
    }
  }
  Buf h;
  h.u32(CIR_MAGIC);
  h.u32(BLOCK);
  h.u64(blocks.size());
  h.u32(root.c0);
  h.u32(root.s0);
  h.u32(root.c1);
  h.u32(root.e1);
  h.u64(*index_off);  // endFileOffset: where the indexed data ends (the index follows it)
  h.u32(ITEMS_PER_SLOT);
  h.u32(0);
  if (!put(f_, h.d)) return false;
  for (size_t L = levels.size(); L-- > 0;) {
    for (const Node& n : levels[L]) {
      Buf b;
      b.u8(L == 0 ? 1 : 0);
      b.u8(0);
      b.u16((uint32_t)n.count);
      for (size_t j = 0; j < n.count; ++j) {
        if (L == 0) {
          const Block& k = blocks[n.first + j];
          b.u32(k.chrom);
          b.u32(k.start);
          b.u32(k.chrom);
          b.u32(k.end);
          b.u64(k.off);
          b.u64(k.size);
        } else {
          const Node& c = levels[L - 1][n.first + j];
          b.u32(c.c0);
          b.u32(c.s0);
          b.u32(c.c1);
          b.u32(c.e1);
          b.u64(c.off);
        }
      }
      if (!put(f_, b.d)) return false;
    }
  }
  return true;
}

bool BigWigWriter::close(std::string& err) {
  auto fail = [&](const char* what) {
    err = std::string(what) + " (" + path_ + ")";
    if (f_) fclose(f_);
    f_ = nullptr;
    return false;
  };
  if (!f_) return fail("bigWig file is not open");
  // ---- zoom plan: reductions of 4x from a first level a few times the mean interval, while a level still has many records
  uint64_t covered = 0;
  double vmin = 0, vmax = 0, vsum = 0, vsq = 0;
  for (size_t i = 0; i < items_.size(); ++i) {
    const Item& it = items_[i];
    const double w = (double)(it.end - it.start), v = (double)it.val;
    covered += it.end - it.start;
    if (i == 0 || v < vmin) vmin = v;
    if (i == 0 || v > vmax) vmax = v;
    vsum += v * w;
    vsq += v * v * w;
  }
  struct ZRec {
    uint32_t chrom, start, end, valid;
    float mn, mx, sum, sq;
  };
  std::vector<uint32_t> reductions;
  std::vector<std::vector<ZRec>> zoom;
  if (!items_.empty()) {
    uint64_t red = std::max<uint64_t>(32, 4 * (covered / items_.size() + 1));
    for (int z = 0; z < MAX_ZOOM && red < (1ull << 31); ++z, red *= 4) {
      std::vector<ZRec> recs;
      for (const Item& it : items_) {  // bins aligned to multiples of the reduction; an interval feeds every bin it overlaps
        for (uint64_t b = it.start / red; b * red < it.end; ++b) {
          const uint32_t bs = (uint32_t)(b * red), be = (uint32_t)std::min<uint64_t>((b + 1) * red, UINT32_MAX);
          const uint32_t lo = std::max(bs, it.start), hi = std::min(be, it.end);
          if (recs.empty() || recs.back().chrom != it.chrom || recs.back().start != bs) {
            const uint32_t clen = it.chrom < lens_.size() ? lens_[it.chrom] : be;
            recs.push_back(ZRec{it.chrom, bs, std::min(be, std::max(clen, hi)), 0, it.val, it.val, 0.f, 0.f});
          }
          ZRec& r = recs.back();
          const float w = (float)(hi - lo);
          r.valid += hi - lo;
          r.mn = std::min(r.mn, it.val);
          r.mx = std::max(r.mx, it.val);
          r.sum += it.val * w;
          r.sq += it.val * it.val * w;
        }
      }
      if (z > 0 && recs.size() * 2 > zoom.back().size()) break;  // no longer shrinking
      reductions.push_back((uint32_t)red);
      zoom.push_back(std::move(recs));
      if (zoom.back().size() < 1000) break;
    }
  }
  const uint32_t nz = (uint32_t)zoom.size();
  // ---- fixed part (rewritten at the end with the offsets), chromosome tree
  const uint64_t summary_off = 64 + 24ull * nz;
  const uint64_t tree_off = summary_off + 40;
  {
    std::vector<uint8_t> zero((size_t)tree_off, 0);
    if (!put(f_, zero)) return fail("write failed");
  }
  uint32_t key = 1;
  for (auto& n : names_) key = std::max<uint32_t>(key, (uint32_t)n.size());
  {
    // leaves of up to BLOCK chromosomes (sorted by name, as a B+ tree needs), parents of up to BLOCK children; every node is
    // padded to BLOCK items so that the offsets are known before anything is written
    std::vector<uint32_t> order(names_.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return names_[a] < names_[b]; });
    std::vector<size_t> level_nodes;  // nodes per level, leaves first
    size_t cnt = std::max<size_t>(1, (order.size() + BLOCK - 1) / BLOCK);
    level_nodes.push_back(cnt);
    while (cnt > 1) {
      cnt = (cnt + BLOCK - 1) / BLOCK;
      level_nodes.push_back(cnt);
    }
    const uint64_t node_bytes = 4 + (uint64_t)BLOCK * (key + 8);
    std::vector<uint64_t> level_off(level_nodes.size());
    uint64_t off = tree_off + 32;
    for (size_t L = level_nodes.size(); L-- > 0;) {
      level_off[L] = off;
      off += level_nodes[L] * node_bytes;
    }
    Buf b;
    b.u32(BPT_MAGIC);
    b.u32(BLOCK);
    b.u32(key);
    b.u32(8);
    b.u64(order.size());
    b.u64(0);
    auto put_key = [&](const std::string& s) {
      for (uint32_t i = 0; i < key; ++i) b.u8(i < s.size() ? (uint8_t)s[i] : 0);
    };
    // first key under node j of level L
    auto first_key = [&](size_t L, size_t j) -> const std::string& {
      size_t span = 1;
      for (size_t l = 0; l < L; ++l) span *= BLOCK;
      return names_[order[std::min(order.size() - 1, j * span * BLOCK)]];
    };
    for (size_t L = level_nodes.size(); L-- > 0;) {
      for (size_t j = 0; j < level_nodes[L]; ++j) {
        const size_t below = L == 0 ? order.size() : level_nodes[L - 1];
        const size_t first = j * BLOCK, n = below > first ? std::min<size_t>(BLOCK, below - first) : 0;
        b.u8(L == 0 ? 1 : 0);
        b.u8(0);
        b.u16((uint32_t)n);
        for (size_t i = 0; i < BLOCK; ++i) {
          if (i < n && L == 0) {
            const uint32_t id = order[first + i];
            put_key(names_[id]);
            b.u32(id);
            b.u32(lens_[id]);
          } else if (i < n) {
            put_key(first_key(L - 1, first + i));
            b.u64(level_off[L - 1] + (first + i) * node_bytes);
          } else {
            for (uint32_t z = 0; z < key + 8; ++z) b.u8(0);
          }
        }
      }
    }
    if (!put(f_, b.d)) return fail("write failed");
  }
  // ---- data sections
  const uint64_t data_off = (uint64_t)ftello(f_);
  std::vector<Block> blocks;
  {
    Buf c;
    c.u64(0);  // section count, patched below
    if (!put(f_, c.d)) return fail("write failed");
    size_t i = 0;
    while (i < items_.size()) {
      size_t j = i;
      while (j < items_.size() && j - i < ITEMS_PER_SLOT && items_[j].chrom == items_[i].chrom) ++j;
      Buf s;
      s.u32(items_[i].chrom);
      s.u32(items_[i].start);
      s.u32(items_[j - 1].end);
      s.u32(0);  // itemStep
      s.u32(0);  // itemSpan
      s.u8(1);   // bedGraph
      s.u8(0);
      s.u16((uint32_t)(j - i));
      for (size_t q = i; q < j; ++q) {
        s.u32(items_[q].start);
        s.u32(items_[q].end);
        s.f32(items_[q].val);
      }
      if (!write_sections(s.d, items_[i].chrom, items_[i].start, items_[j - 1].end, blocks)) return fail("write failed");
      i = j;
    }
  }
  uint64_t index_off = 0;
  if (!write_index(blocks, &index_off)) return fail("write failed");
  // ---- zoom levels
  std::vector<uint64_t> zdata(nz), zindex(nz);
  for (uint32_t z = 0; z < nz; ++z) {
    zdata[z] = (uint64_t)ftello(f_);
    Buf c;
    c.u32((uint32_t)zoom[z].size());
    if (!put(f_, c.d)) return fail("write failed");
    std::vector<Block> zb;
    size_t i = 0;
    while (i < zoom[z].size()) {
      size_t j = i;
      while (j < zoom[z].size() && j - i < ITEMS_PER_SLOT && zoom[z][j].chrom == zoom[z][i].chrom) ++j;
      Buf s;
      for (size_t q = i; q < j; ++q) {
        const ZRec& r = zoom[z][q];
        s.u32(r.chrom);
        s.u32(r.start);
        s.u32(r.end);
        s.u32(r.valid);
        s.f32(r.mn);
        s.f32(r.mx);
        s.f32(r.sum);
        s.f32(r.sq);
      }
      if (!write_sections(s.d, zoom[z][i].chrom, zoom[z][i].start, zoom[z][j - 1].end, zb)) return fail("write failed");
      i = j;
    }
    if (!write_index(zb, &zindex[z])) return fail("write failed");
  }
  {
    Buf m;
    m.u32(BW_MAGIC);  // the file ends with the magic again
    if (!put(f_, m.d)) return fail("write failed");
  }
  // ---- header, zoom headers, summary, section count
  Buf h;
  h.u32(BW_MAGIC);
  h.u16(4);
  h.u16(nz);
  h.u64(tree_off);
  h.u64(data_off);
  h.u64(index_off);
  h.u16(0);
  h.u16(0);
  h.u64(0);
  h.u64(summary_off);
  h.u32(max_uncompressed_);
  h.u64(0);
  for (uint32_t z = 0; z < nz; ++z) {
    h.u32(reductions[z]);
    h.u32(0);
    h.u64(zdata[z]);
    h.u64(zindex[z]);
  }
  h.u64(covered);
  h.f64(vmin);
  h.f64(vmax);
  h.f64(vsum);
  h.f64(vsq);
  if (fseeko(f_, 0, SEEK_SET) != 0 || !put(f_, h.d)) return fail("write failed");
  Buf c;
  c.u64(blocks.size());
  if (fseeko(f_, (off_t)data_off, SEEK_SET) != 0 || !put(f_, c.d)) return fail("write failed");
  if (fclose(f_) != 0) {
    f_ = nullptr;
    return fail("write failed");
  }
  f_ = nullptr;
  return true;
}

}  // namespace tbh
