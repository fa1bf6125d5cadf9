// deflate_codes.h — the serial pieces of an RFC 1951 encoder, written once for the device (bgzdef.hip: one lane of the member's
// workgroup runs them on arrays in LDS) and for the host-compiled model of the encoder that the tests and tools/deflate_model use to
// pin them against zlib without a GPU.  Nothing here allocates, recurses or calls a library.
//
//   dfl_code_lengths     minimum-redundancy code lengths of a symbol set, limited to `maxbits` (Moffat & Katajainen's in-place
//                        algorithm on the sorted frequencies, then zlib's overflow repair for the limit)
//   dfl_canonical_codes  canonical codes of RFC 1951 §3.2.2 from the lengths, bit-reversed (deflate sends Huffman codes MSB first
//                        inside an LSB-first bit stream)
//   dfl_rle_lengths      the code-length sequence of a dynamic block as symbols 0..18 with their extra bits (§3.2.7)
//   dfl_len_code / dfl_dist_code   length / distance -> symbol, extra bits (arithmetic, no tables)
//
// What GSamWriter::write -> sam_write1 -> bgzf_write does through zlib's deflate() (/root/reference/src/GSam.h:648-653).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DFL_FN __host__ __device__ inline
#else
#define DFL_FN static inline
#endif

#define DFL_NLIT 286   // literal / length symbols that can be sent (0..285)
#define DFL_NDIST 30
#define DFL_NCL 19
#define DFL_MIN_MATCH 3
#define DFL_MAX_MATCH 258

// length 3..258 -> (symbol - 257, number of extra bits, extra value)
DFL_FN void dfl_len_code(uint32_t len, uint32_t* code, uint32_t* ebits, uint32_t* eval) {
  const uint32_t l = len - 3;
  if (len == 258) {
    *code = 28, *ebits = 0, *eval = 0;
  } else if (l < 8) {
    *code = l, *ebits = 0, *eval = 0;
  } else {
    const uint32_t e = (31u - (uint32_t)__builtin_clz(l)) - 2u;
    *code = 4 * e + 4 + ((l >> e) & 3u);
    *ebits = e;
    *eval = l & ((1u << e) - 1u);
  }
}
// distance 1..32768 -> (symbol, number of extra bits, extra value)
DFL_FN void dfl_dist_code(uint32_t dist, uint32_t* code, uint32_t* ebits, uint32_t* eval) {
  const uint32_t d = dist - 1;
  if (d < 4) {
    *code = d, *ebits = 0, *eval = 0;
  } else {
    const uint32_t e = (31u - (uint32_t)__builtin_clz(d)) - 1u;
    *code = 2 * e + 2 + ((d >> e) & 1u);
    *ebits = e;
    *eval = d & ((1u << e) - 1u);
  }
}

// Code lengths of the m symbols whose frequencies stand SORTED ASCENDING in a[0, m) (m >= 2, every frequency > 0): on return a[i] is
// the length of the i-th rarest symbol (non-increasing in i), never above maxbits, and the set is complete (Kraft sum exactly 1).
// blc[0 .. maxbits] is scratch.
DFL_FN void dfl_code_lengths(uint32_t* a, int m, int maxbits, uint32_t* blc) {
  if (m == 2) {
    a[0] = a[1] = 1;
    return;
  }
  // Moffat & Katajainen, "In-place calculation of minimum-redundancy codes": phase 1 builds the tree into a[] (internal node
  // weights, then parent indices), phase 2 turns parents into internal depths, phase 3 leaves the leaf depths
  a[0] += a[1];
  int root = 0, leaf = 2;
  for (int next = 1; next < m - 1; ++next) {
    if (leaf >= m || a[root] < a[leaf]) {
      a[next] = a[root];
      a[root++] = (uint32_t)next;
    } else {
      a[next] = a[leaf++];
    }
    if (leaf >= m || (root < next && a[root] < a[leaf])) {
      a[next] += a[root];
      a[root++] = (uint32_t)next;
    } else {
      a[next] += a[leaf++];
    }
  }
  a[m - 2] = 0;
  for (int next = m - 3; next >= 0; --next) a[next] = a[a[next]] + 1;
  int avbl = 1, used = 0, dpth = 0, next = m - 1;
  root = m - 2;
  while (avbl > 0) {
    while (root >= 0 && (int)a[root] == dpth) {
      ++used;
      --root;
    }
    while (avbl > used) {
      a[next--] = (uint32_t)dpth;
      --avbl;
    }
    avbl = 2 * used;
    ++dpth;
    used = 0;
  }
  if ((int)a[0] <= maxbits) return;
  // the limit (zlib's gen_bitlen): clamp, then repair the Kraft sum by moving one leaf down from the deepest level that still has
  // one above the limit's row, and deal the lengths out again, the longest to the rarest
  for (int b = 0; b <= maxbits; ++b) blc[b] = 0;
  int overflow = 0;
  for (int i = 0; i < m; ++i) {
    int l = (int)a[i];
    if (l > maxbits) l = maxbits, ++overflow;
    ++blc[l];
  }
  // every clamped leaf claims 2^-maxbits more than it is entitled to... the sum over the clamped leaves of (2^-maxbits - 2^-l):
  // recompute the excess exactly in units of 2^-maxbits instead of trusting a count
  {
    uint64_t kraft = 0;  // in units of 2^-maxbits
    for (int b = 1; b <= maxbits; ++b) kraft += (uint64_t)blc[b] << (maxbits - b);
    uint64_t over = kraft - ((uint64_t)1 << maxbits);
    while (over > 0) {
      int bits = maxbits - 1;
      while (blc[bits] == 0) --bits;
      // one leaf of `bits` becomes an internal node with two children at bits + 1, one of which takes a leaf from the last row:
      // the sum drops by exactly one unit of 2^-maxbits
      --blc[bits];
      blc[bits + 1] += 2;
      --blc[maxbits];
      --over;
    }
  }
  int i = 0;
  for (int b = maxbits; b >= 1; --b)
    for (uint32_t c = blc[b]; c > 0; --c) a[i++] = (uint32_t)b;
}

DFL_FN uint32_t dfl_bitrev(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
  return r;
}

// canonical codes (RFC 1951 §3.2.2) of n symbols from len[], bit-reversed; code[s] is meaningless where len[s] == 0
DFL_FN void dfl_canonical_codes(const uint8_t* len, int n, int maxbits, uint16_t* code, uint32_t* blc /* [maxbits + 2] scratch */) {
  for (int b = 0; b <= maxbits + 1; ++b) blc[b] = 0;
  for (int s = 0; s < n; ++s) ++blc[len[s]];
  blc[0] = 0;
  uint32_t c = 0, prev = 0;
  for (int b = 1; b <= maxbits; ++b) {  // blc[b] becomes the first code of length b
    c = (c + prev) << 1;
    prev = blc[b];
    blc[b] = c;
  }
  for (int s = 0; s < n; ++s)
    if (len[s]) code[s] = (uint16_t)dfl_bitrev(blc[len[s]]++, len[s]);
}

// The code lengths lens[0, n) as the symbols of §3.2.7: out[i] = symbol | extra value << 8 (16: copy the previous length 3..6 times, 2
// extra bits; 17: 3..10 zeros, 3 bits; 18: 11..138 zeros, 7 bits).  Returns the number of symbols; clfreq[19] is incremented.
DFL_FN int dfl_rle_lengths(const uint8_t* lens, int n, uint16_t* out, uint32_t* clfreq) {
  int no = 0;
  for (int i = 0; i < n;) {
    const uint8_t v = lens[i];
    int run = 1;
    while (i + run < n && lens[i + run] == v) ++run;
    i += run;
    if (v == 0) {
      while (run >= 11) {
        const int r = run > 138 ? 138 : run;
        out[no++] = (uint16_t)(18 | ((r - 11) << 8));
        ++clfreq[18];
        run -= r;
      }
      if (run >= 3) {
        out[no++] = (uint16_t)(17 | ((run - 3) << 8));
        ++clfreq[17];
        run = 0;
      }
      for (; run > 0; --run) out[no++] = 0, ++clfreq[0];
    } else {
      out[no++] = v, ++clfreq[v];
      --run;
      while (run >= 3) {
        const int r = run > 6 ? 6 : run;
        out[no++] = (uint16_t)(16 | ((r - 3) << 8));
        ++clfreq[16];
        run -= r;
      }
      for (; run > 0; --run) out[no++] = v, ++clfreq[v];
    }
  }
  return no;
}

// order in which the code-length code's own lengths are sent (§3.2.7)
DFL_FN int dfl_cl_order(int i) {
  const uint8_t o[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  return o[i];
}

// fixed codes of §3.2.6
DFL_FN int dfl_fixed_litlen_bits(int s) { return s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)); }
