// deflate_model.cpp — TEST INFRASTRUCTURE (never linked into the product): a host-compiled model of the device BGZF encoder
// (tiebrush_amd/csrc/bgzdef.hip).  It runs the same parse — the latest earlier position with the same 4-byte hash as the one
// candidate per table, greedy / lazy token choice, one dynamic-Huffman block per member — and the very same serial code
// (tiebrush_amd/csrc/deflate_codes.h) for code lengths, canonical codes, the block header and the token bits, so that those are pinned
// against zlib on this CPU-only container: every member it writes must inflate with zlib to its payload.  It also reports the sizes
// zlib's levels reach on the same members, which is how the parse was chosen.
//
//   deflate_model selftest
//   deflate_model <file> [hbits=11] [lazy=1] [tables=2] [min4_far] [good=16] [member=65280] [h8bits=10]   (defaults = the kernel's)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <vector>

#include "../../tiebrush_amd/csrc/deflate_codes.h"

struct BitOut {
  std::vector<uint8_t> b;
  uint64_t acc = 0;
  int n = 0;
  void put(uint32_t v, int bits) {
    acc |= (uint64_t)v << n;
    n += bits;
    while (n >= 8) b.push_back((uint8_t)acc), acc >>= 8, n -= 8;
  }
  void flush() {
    if (n > 0) b.push_back((uint8_t)acc);
    acc = 0, n = 0;
  }
};

struct Params {
  int hbits = 11, lazy = 1, tables = 2, batch = 256, min4_far = 32768, good = 16, member = 0xff00, h8bits = 10, shortcap = 0, tie8 = 1;
};

static inline uint32_t ld32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static inline uint64_t ld64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

// one member: tokens (lit: byte; match: 0x80000000 | len << 16 | dist - 1)
static void parse(const uint8_t* s, uint32_t n, const Params& P, std::vector<uint32_t>& tok) {
  const int h8b = P.h8bits ? P.h8bits : P.hbits;
  std::vector<uint32_t> t4((size_t)1 << P.hbits, 0xFFFFFFFFu), t8((size_t)1 << h8b, 0xFFFFFFFFu);
  std::vector<uint16_t> mlen(n + 1, 0), mdist(n + 1, 0);
  for (uint32_t p = 0; p < n; ++p) {
    uint32_t best = 0, bd = 0;
    if (p + 4 <= n) {
      const uint32_t h = (ld32(s + p) * 0x9E3779B1u) >> (32 - P.hbits);
      uint32_t c[2] = {t4[h], 0xFFFFFFFFu};
      t4[h] = p;
      if (P.tables > 1 && p + 8 <= n) {
        const uint32_t h8 = (ld32(s + p) * 0x9E3779B1u + ld32(s + p + 4) * 0x85EBCA77u) >> (32 - h8b);   // (the kernel's hash)
        c[1] = t8[h8];
        t8[h8] = p;
      }
      uint32_t sl[2] = {0, 0}, fl[2] = {0, 0};
      for (int k = 0; k < 2; ++k) {
        const uint32_t q = c[k];
        if (q == 0xFFFFFFFFu || p - q > 32768) continue;
        uint32_t l = 0;
        const uint32_t lim = std::min<uint32_t>(258, n - p);
        while (l < lim && s[q + l] == s[p + l]) ++l;
        if (l < 4) continue;
        fl[k] = l;
        sl[k] = P.shortcap ? std::min<uint32_t>(l, (uint32_t)P.shortcap) : l;
      }
      // the kernel keeps ONE candidate per position, chosen on the short lengths (the first `shortcap` bytes); its full length is
      // found when the parse takes it
      int pick = -1;
      if (sl[0] || sl[1]) pick = sl[1] > sl[0] ? 1 : (sl[0] > sl[1] ? 0 : ((P.tie8 && sl[1]) ? 1 : 0));
      if (pick >= 0 && !sl[pick]) pick ^= 1;
      if (pick >= 0) best = fl[pick], bd = p - c[pick];
      if (best == 4 && bd > (uint32_t)P.min4_far) best = 0, bd = 0;
    }
    mlen[p] = (uint16_t)best;
    mdist[p] = (uint16_t)(bd ? bd - 1 : 0);
  }
  for (uint32_t p = 0; p < n;) {
    uint32_t l = mlen[p];
    if (l >= 4) {
      if (P.lazy && l < (uint32_t)P.good && p + 1 < n && mlen[p + 1] > l) {
        tok.push_back(s[p]);
        ++p;
        continue;
      }
      tok.push_back(0x80000000u | (l << 16) | mdist[p]);
      p += l;
    } else {
      tok.push_back(s[p]);
      ++p;
    }
  }
}

struct Codes {
  uint8_t len[DFL_NLIT + 2];
  uint16_t code[DFL_NLIT + 2];
};

// lengths (<= maxbits) of the n symbols with frequencies f[]; at least two symbols get a code, as zlib arranges it
static void build(const uint32_t* f_in, int n, int maxbits, uint8_t* len, uint16_t* code) {
  std::vector<uint32_t> f(f_in, f_in + n);
  int nz = 0;
  for (int s = 0; s < n; ++s) nz += f[s] != 0;
  for (int s = 0; nz < 2 && s < n; ++s)
    if (f[s] == 0) f[s] = 1, ++nz;
  std::vector<uint32_t> key;
  for (int s = 0; s < n; ++s)
    if (f[s]) key.push_back(f[s] << 9 | (uint32_t)s);   // (f < 2^17 in a 64 KiB member)
  std::sort(key.begin(), key.end());
  std::vector<uint32_t> a(key.size());
  for (size_t i = 0; i < key.size(); ++i) a[i] = key[i] >> 9;
  uint32_t blc[18];
  dfl_code_lengths(a.data(), (int)a.size(), maxbits, blc);
  memset(len, 0, (size_t)n);
  for (size_t i = 0; i < key.size(); ++i) len[key[i] & 511] = (uint8_t)a[i];
  dfl_canonical_codes(len, n, maxbits, code, blc);
}

static void encode(const std::vector<uint32_t>& tok, const uint8_t* s, uint32_t n, BitOut& o) {
  uint32_t lf[DFL_NLIT + 2] = {0}, df[DFL_NDIST + 2] = {0};
  uint64_t extra = 0;
  for (uint32_t t : tok) {
    if (t & 0x80000000u) {
      uint32_t c, e, v;
      dfl_len_code((t >> 16) & 0x1FF, &c, &e, &v);
      ++lf[257 + c];
      extra += e;
      dfl_dist_code((t & 0xFFFF) + 1, &c, &e, &v);
      ++df[c];
      extra += e;
    } else {
      ++lf[t];
    }
  }
  ++lf[256];
  Codes L, D, C;
  build(lf, DFL_NLIT, 15, L.len, L.code);
  build(df, DFL_NDIST, 15, D.len, D.code);
  int hlit = DFL_NLIT, hdist = DFL_NDIST;
  while (hlit > 257 && L.len[hlit - 1] == 0) --hlit;
  while (hdist > 1 && D.len[hdist - 1] == 0) --hdist;
  uint8_t all[DFL_NLIT + DFL_NDIST];
  memcpy(all, L.len, (size_t)hlit);
  memcpy(all + hlit, D.len, (size_t)hdist);
  uint16_t rle[DFL_NLIT + DFL_NDIST];
  uint32_t cf[DFL_NCL] = {0};
  const int nr = dfl_rle_lengths(all, hlit + hdist, rle, cf);
  build(cf, DFL_NCL, 7, C.len, C.code);
  int hclen = DFL_NCL;
  while (hclen > 4 && C.len[dfl_cl_order(hclen - 1)] == 0) --hclen;
  uint64_t dyn_bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen + extra;
  for (int i = 0; i < nr; ++i) {
    const int sy = rle[i] & 0xFF;
    dyn_bits += C.len[sy] + (sy == 16 ? 2 : (sy == 17 ? 3 : (sy == 18 ? 7 : 0)));
  }
  for (int sy = 0; sy < DFL_NLIT; ++sy) dyn_bits += (uint64_t)lf[sy] * L.len[sy];
  for (int sy = 0; sy < DFL_NDIST; ++sy) dyn_bits += (uint64_t)df[sy] * D.len[sy];
  if (dyn_bits > (uint64_t)n * 8 + 40) {  // stored
    o.put(1, 1);
    o.put(0, 2);
    o.flush();
    o.put(n & 0xFFFF, 16);
    o.put(~n & 0xFFFF, 16);
    for (uint32_t i = 0; i < n; ++i) o.put(s[i], 8);
    return;
  }
  o.put(1, 1);
  o.put(2, 2);
  o.put((uint32_t)hlit - 257, 5);
  o.put((uint32_t)hdist - 1, 5);
  o.put((uint32_t)hclen - 4, 4);
  for (int i = 0; i < hclen; ++i) o.put(C.len[dfl_cl_order(i)], 3);
  for (int i = 0; i < nr; ++i) {
    const int sy = rle[i] & 0xFF;
    o.put(C.code[sy], C.len[sy]);
    if (sy >= 16) o.put((uint32_t)rle[i] >> 8, sy == 16 ? 2 : (sy == 17 ? 3 : 7));
  }
  for (uint32_t t : tok) {
    if (t & 0x80000000u) {
      uint32_t c, e, v;
      dfl_len_code((t >> 16) & 0x1FF, &c, &e, &v);
      o.put(L.code[257 + c], L.len[257 + c]);
      if (e) o.put(v, (int)e);
      dfl_dist_code((t & 0xFFFF) + 1, &c, &e, &v);
      o.put(D.code[c], D.len[c]);
      if (e) o.put(v, (int)e);
    } else {
      o.put(L.code[t], L.len[t]);
    }
  }
  o.put(L.code[256], L.len[256]);
  o.flush();
}

static size_t zsize(const uint8_t* s, uint32_t n, int level) {
  z_stream z;
  memset(&z, 0, sizeof(z));
  deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  std::vector<uint8_t> o(deflateBound(&z, n) + 64);
  z.next_in = const_cast<uint8_t*>(s), z.avail_in = n, z.next_out = o.data(), z.avail_out = (uInt)o.size();
  deflate(&z, Z_FINISH);
  const size_t r = z.total_out;
  deflateEnd(&z);
  return r;
}

// deflate_codes.h on its own: code lengths of adversarial frequency sets must be complete (Kraft sum exactly 1), within the limit,
// never shorter for a rarer symbol, and — where the limit does not bind — as cheap as an independent Huffman construction
static int selftest() {
  uint64_t x = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    return x;
  };
  int cases = 0;
  for (int kind = 0; kind < 6; ++kind)
    for (int maxbits : {7, 15})
      for (int m = 2; m <= (maxbits == 7 ? 19 : 286); m += (m < 40 ? 1 : 17)) {
        std::vector<uint32_t> f((size_t)m);
        for (int i = 0; i < m; ++i) {
          switch (kind) {
            case 0: f[i] = 1; break;
            case 1: f[i] = (uint32_t)(rnd() % 1000) + 1; break;
            case 2: f[i] = i < 2 ? 1u : f[i - 1] + f[i - 2]; if (f[i] > 60000) f[i] = 60000; break;  // Fibonacci: the deepest trees
            case 3: f[i] = 1u << (i < 16 ? i : 16); break;
            case 4: f[i] = (uint32_t)(rnd() % 3) == 0 ? 50000u : 1u; break;
            default: f[i] = (uint32_t)(rnd() % 65000) + 1; break;
          }
        }
        std::sort(f.begin(), f.end());
        std::vector<uint32_t> a(f);
        uint32_t blc[18];
        dfl_code_lengths(a.data(), m, maxbits, blc);
        uint64_t kraft = 0, cost = 0;
        for (int i = 0; i < m; ++i) {
          if (a[i] < 1 || (int)a[i] > maxbits) return fprintf(stderr, "selftest: length %u outside 1..%d (kind %d, m %d)\n", a[i], maxbits, kind, m), 1;
          if (i && a[i] > a[i - 1]) return fprintf(stderr, "selftest: a rarer symbol got the shorter code (kind %d, m %d)\n", kind, m), 1;
          kraft += (uint64_t)1 << (maxbits - a[i]);
          cost += (uint64_t)a[i] * f[i];
        }
        if (kraft != (uint64_t)1 << maxbits) return fprintf(stderr, "selftest: Kraft sum off (kind %d, m %d, maxbits %d)\n", kind, m, maxbits), 1;
        // independent construction: repeated merging of the two lightest nodes (O(m^2), fine here)
        std::vector<uint64_t> w(f.begin(), f.end());
        uint64_t hcost = 0;
        std::vector<uint64_t> heap(w);
        int depth_bound = 0;
        while (heap.size() > 1) {
          std::sort(heap.begin(), heap.end());
          const uint64_t s2 = heap[0] + heap[1];
          hcost += s2;
          heap.erase(heap.begin(), heap.begin() + 2);
          heap.push_back(s2);
          ++depth_bound;
        }
        if (cost < hcost) return fprintf(stderr, "selftest: cheaper than a Huffman code?! (kind %d, m %d)\n", kind, m), 1;
        // when the unlimited optimum already fits (cost equal) nothing was lost; otherwise the limit may cost something, never a lot
        if (cost > hcost + hcost / 8 + 64) return fprintf(stderr, "selftest: limited code far from the optimum: %llu vs %llu (kind %d, m %d, maxbits %d)\n",
                                                          (unsigned long long)cost, (unsigned long long)hcost, kind, m, maxbits), 1;
        uint8_t len[288];
        uint16_t code[288];
        for (int i = 0; i < m; ++i) len[i] = (uint8_t)a[i];
        dfl_canonical_codes(len, m, maxbits, code, blc);
        for (int i = 0; i < m; ++i)      // prefix-free: no code is the bit-reversed prefix of another
          for (int j = 0; j < m; ++j)
            if (i != j && len[i] <= len[j] && (code[j] & ((1u << len[i]) - 1u)) == code[i])
              return fprintf(stderr, "selftest: code %d is a prefix of code %d (kind %d, m %d)\n", i, j, kind, m), 1;
        ++cases;
      }
  // the arithmetic symbol maps against RFC 1951's tables
  static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  for (uint32_t l = 3; l <= 258; ++l) {
    uint32_t c, e, v;
    dfl_len_code(l, &c, &e, &v);
    if (c > 28 || e != lext[c] || lbase[c] + v != l || v >= (1u << e) + (e == 0)) return fprintf(stderr, "selftest: length %u -> code %u\n", l, c), 1;
  }
  for (uint32_t d = 1; d <= 32768; ++d) {
    uint32_t c, e, v;
    dfl_dist_code(d, &c, &e, &v);
    if (c > 29 || e != dext[c] || dbase[c] + v != d || v >= (1u << e) + (e == 0)) return fprintf(stderr, "selftest: distance %u -> code %u\n", d, c), 1;
  }
  printf("{\"selftest\": \"ok\", \"cases\": %d}\n", cases);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && strcmp(argv[1], "selftest") == 0) return selftest();
  if (argc < 2) return 2;
  Params P;
  if (argc > 2) P.hbits = atoi(argv[2]);
  if (argc > 3) P.lazy = atoi(argv[3]);
  if (argc > 4) P.tables = atoi(argv[4]);
  if (argc > 5) P.min4_far = atoi(argv[5]);
  if (argc > 6) P.good = atoi(argv[6]);
  if (argc > 7) P.member = atoi(argv[7]);
  if (argc > 8) P.h8bits = atoi(argv[8]);
  if (argc > 9) P.shortcap = atoi(argv[9]);
  if (argc > 10) P.tie8 = atoi(argv[10]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> d;
  uint8_t buf[1 << 16];
  size_t r;
  while ((r = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + r);
  fclose(f);
  size_t tot = 0, z1 = 0, z6 = 0, z9 = 0, ntok = 0, nmem = 0;
  for (size_t off = 0; off < d.size(); off += (size_t)P.member) {
    const uint32_t n = (uint32_t)std::min<size_t>((size_t)P.member, d.size() - off);
    std::vector<uint32_t> tok;
    parse(d.data() + off, n, P, tok);
    BitOut o;
    encode(tok, d.data() + off, n, o);
    // every member must inflate with zlib to its payload
    std::vector<uint8_t> back(n + 16);
    z_stream z;
    memset(&z, 0, sizeof(z));
    inflateInit2(&z, -15);
    z.next_in = o.b.data(), z.avail_in = (uInt)o.b.size(), z.next_out = back.data(), z.avail_out = (uInt)back.size();
    const int rc = inflate(&z, Z_FINISH);
    if (rc != Z_STREAM_END || z.total_out != n || memcmp(back.data(), d.data() + off, n) != 0) {
      fprintf(stderr, "member at %zu: zlib does not give the payload back (rc %d, %lu of %u bytes)\n", off, rc, z.total_out, n);
      return 1;
    }
    inflateEnd(&z);
    tot += o.b.size(), ntok += tok.size(), ++nmem;
    z1 += zsize(d.data() + off, n, 1), z6 += zsize(d.data() + off, n, 6), z9 += zsize(d.data() + off, n, 9);
  }
  printf("{\"bytes\": %zu, \"members\": %zu, \"model\": %zu, \"zlib1\": %zu, \"zlib6\": %zu, \"zlib9\": %zu, \"tokens\": %zu, \"model_over_zlib6\": %.4f}\n", d.size(), nmem, tot,
         z1, z6, z9, ntok, (double)tot / (double)z6);
  return 0;
}
