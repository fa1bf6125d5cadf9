// strategy.hpp — the record view of the collapse stage and the strategy key of a record (what "the same alignment" means
// under the four merge strategies): seeded hash, exact equality, and the reference's three-way compare.
// Reference: cmpCigar / cmpCigarClip / cmpExons / cmpFull, /root/reference/src/tiebrush.cpp:285-345.
// Shared by collapse.hip (sort path) and wgroup.hip (window path).
#pragma once
#include "dev_common.hpp"
#include "tbk_internal.h"

namespace tbkd {

// seed of the first grouping attempt (collapse.hip reseeds on a collision) and of the key word that travels in a group partial
// (shard.hip, tbk_partial_pack: every rank hashes with this one, so equal alignments carry equal words)
constexpr uint64_t TBK_KEY_SEED0 = 0x71EB5EEDull;

struct ColIn {
  uint32_t n, k;
  const uint32_t* file_off;  // device copy [k+1]
  const uint8_t* tbm;        // device copy [k]
  const int32_t *tid, *pos;
  const uint16_t* flag;
  const uint8_t *mapq, *strand;
  const int32_t* nh;
  const uint32_t *cig_off, *cig;
  const double* yc_in;
  const int64_t *yx_in, *yd_in;
  const uint32_t* md_off;
  const uint8_t *md, *md_has;
  const uint64_t* qh;
  const uint32_t* qn_off;  // read names (CSR), -A only
  const uint8_t* qn;
  uint64_t qh_mask;        // ~0 in production; TBK_DEBUG_QHASH_MASK narrows the name-hash filter to provoke collisions in tests
  const uint64_t *prio_hi, *prio_lo;
};

struct ColOpt {
  int strategy;
  int max_nh, min_qual;
  int keep_supp, keep_sec, collapse_same, store_frac;
  uint64_t seed;
  uint32_t hash_mask;  // 0xFFFFFFFF in production; TBK_DEBUG_HASH_MASK narrows it to provoke collisions in tests
};

// -A: "the same read" = same QNAME and same pairOrder (GSamRecord::pairOrder: 1 first, 2 second, 0 neither; tiebrush.cpp:422-424).
// The 64-bit hash of (name, pairOrder) is a filter; equality is decided on the bytes.
__device__ inline bool same_read(const ColIn& I, uint32_t a, uint32_t b) {
  if ((I.qh[a] & I.qh_mask) != (I.qh[b] & I.qh_mask)) return false;
  const uint32_t fa = I.flag[a], fb = I.flag[b];
  const uint32_t pa = (fa & 0x40) ? 1u : ((fa & 0x80) ? 2u : 0u), pb = (fb & 0x40) ? 1u : ((fb & 0x80) ? 2u : 0u);
  if (pa != pb) return false;
  const uint32_t a0 = I.qn_off[a], a1 = I.qn_off[a + 1], b0 = I.qn_off[b], b1 = I.qn_off[b + 1];
  if (a1 - a0 != b1 - b0) return false;
  for (uint32_t k = 0; k < a1 - a0; ++k)
    if (I.qn[a0 + k] != I.qn[b0 + k]) return false;
  return true;
}

__device__ __forceinline__ uint32_t strand_code(uint8_t s) { return s == '+' ? 0u : (s == '-' ? 1u : 2u); }

// clipped CIGAR view (cmpCigarClip tiebrush.cpp:312-332)
template <class C>
__device__ __forceinline__ void clip_view(C c, uint32_t n, uint32_t* b, uint32_t* e) {
  uint32_t lead = 0, last = 0;  // leading S operations; index behind the last operation that is not S
  bool leading = true;
  cig_for_each(c, n, [&](uint32_t k, uint32_t w) {
    const bool is_s = cig_op(w) == C_S;
    leading = leading && is_s;
    lead += leading ? 1u : 0u;
    last = is_s ? last : k + 1;
  });
  *b = lead;
  *e = last > lead ? last : lead;  // (all S: the empty view at n)
}

// (c, n): the record's CIGAR words, I.cig + I.cig_off[i] and their count
template <class C>
__device__ inline uint64_t strategy_hash(const ColIn& I, const ColOpt& O, int strategy, uint32_t i, C c, uint32_t n) {
  uint64_t h = O.seed;
  switch (strategy) {
    case TBK_STRAT_CIGAR:
    case TBK_STRAT_FULL: {
      h = hash_step(h, n);
      cig_for_each(c, n, [&](uint32_t, uint32_t w) { h = hash_step(h, w); });
      if (strategy == TBK_STRAT_FULL) {
        uint32_t has = I.md_has[i];
        h = hash_step(h, has);
        if (has) {
          uint32_t m0 = I.md_off[i], m1 = I.md_off[i + 1];
          h = hash_step(h, m1 - m0);
          for (uint32_t k = m0; k < m1; ++k) h = hash_step(h, I.md[k]);
        }
      }
      break;
    }
    case TBK_STRAT_CLIP: {
      uint32_t b, e;
      clip_view(c, n, &b, &e);
      h = hash_step(h, e - b);
      cig_for_each(c, n, [&](uint32_t k, uint32_t w) {
        if (k >= b && k < e) h = hash_step(h, w);
      });
      break;
    }
    case TBK_STRAT_EXON: {
      int nex = 0;
      walk_exons(I.pos[i], c, n, [&](int s, int e) { h = hash_step(h, ((uint64_t)(uint32_t)s << 32) | (uint32_t)e); }, &nex);
      h = hash_step(h, (uint64_t)nex);
      break;
    }
  }
  return h;
}

__device__ inline uint64_t strategy_hash(const ColIn& I, const ColOpt& O, uint32_t i) {
  return strategy_hash(I, O, O.strategy, i, I.cig + I.cig_off[i], I.cig_off[i + 1] - I.cig_off[i]);
}

// The 128-bit group key of one record from its raw fields (shared by col_keys_k and the raw window path, wgroup.hip):
//   hi = tid+1 : 31 | start : 31 | strand code : 2        lo = span : 32 | h32
// start / end as GSamRecord::setupCoordinates leaves them (GSam.cpp:351-417; unmapped: 0 / 0), `pass` = passes_options
// (tiebrush.cpp:532-541).  h32 = 31 bits of the seeded strategy hash, or — bit 31 set — an EXACT code: with (tid, start,
// strand, span) in the key, a strategy key that is a single reference-consuming operation (after clip stripping under -P),
// a single exon under -E, or the spliced shape M N M / two exons with a first block < 2^10 and a gap < 2^20 is identified
// by the code alone, so equal keys are equal alignments and need no comparison of the CIGARs (nearly every read of an
// RNA-seq sample).  The order inside a (strand, end) tie set never depends on this word (col_tie_sort_k).
struct RecKey {
  uint64_t hi, lo;
  int32_t end;
  bool pass;
  uint32_t err;  // TBK_DERR_* bits met (the caller raises them)
};
// One pass over the CIGAR for the key: reference length, the clipped view [b, e) (cmpCigarClip; the whole CIGAR when `clip` is
// false) and the first three words of that view — no indexed access afterwards.
struct CigScan {
  int l;
  uint32_t b, e, v0, v1, v2;
};
template <class C>
__device__ __forceinline__ CigScan cig_scan(C c, uint32_t n, bool clip) {
  CigScan r;
  r.l = 0;
  r.v0 = r.v1 = r.v2 = 0;
  uint32_t lead = 0, last = 0, nv = 0;
  bool leading = clip;
  cig_for_each(c, n, [&](uint32_t k, uint32_t w) {
    const uint32_t op = cig_op(w);
    r.l += ((0x18Du >> op) & 1u) ? (int)cig_len(w) : 0;  // M,=,X,D,N consume the reference
    const bool is_s = clip && op == C_S;
    leading = leading && is_s;
    lead += leading ? 1u : 0u;
    last = is_s ? last : k + 1;
    if (!leading) {
      r.v0 = nv == 0 ? w : r.v0;
      r.v1 = nv == 1 ? w : r.v1;
      r.v2 = nv == 2 ? w : r.v2;
      ++nv;
    }
  });
  r.b = lead;
  r.e = clip ? (last > lead ? last : lead) : n;
  return r;
}

// The strategy hash of the CIGAR / soft-clip-stripped strategies from the scan above: the same value as strategy_hash (length of the
// view, then its words in order), without a second pass over the CIGAR for the view's bounds — the raw window kernel runs this in
// nearly every wave (one record in thirty needs it, a wave holds 64) and is bound by its vector instructions.
__device__ __forceinline__ uint32_t cig_word_at(const uint32_t* c, uint32_t k) { return c[k]; }
__device__ __forceinline__ uint32_t cig_word_at(const CigView& c, uint32_t k) { return c.p[k]; }  // (k >= 3)
template <class C>
__device__ __forceinline__ uint64_t view_hash(uint64_t seed, const CigScan& v, C c) {
  const uint32_t vn = v.e - v.b;
  uint64_t h = hash_step(seed, vn);
  if (vn > 0) h = hash_step(h, v.v0);
  if (vn > 1) h = hash_step(h, v.v1);
  if (vn > 2) h = hash_step(h, v.v2);
  for (uint32_t k = v.b + 3; k < v.e; ++k) h = hash_step(h, cig_word_at(c, k));
  return h;
}

// (ST: the strategy as a compile-time constant, or -1: O.strategy)
template <int ST = -1, class C>
__device__ __forceinline__ RecKey record_key(const ColIn& I, const ColOpt& O, uint32_t i, uint32_t fl, int pos, int tidv, int mq, int32_t nhv,
                                             uint32_t sc, C c, uint32_t nc) {
  const int strategy = ST >= 0 ? ST : O.strategy;
  bool pass = true;  // passes_options, tiebrush.cpp:532-541
  if (!O.keep_supp && (fl & 0x800)) pass = false;
  if (!O.keep_sec && (fl & 0x100)) pass = false;
  if (fl & 0x4) pass = false;  // keep_unmapped is rejected at the ABI
  if (mq < O.min_qual) pass = false;
  const int nh = nhv == TBK_NH_ABSENT ? 0 : nhv;
  if (nh > O.max_nh) pass = false;
  const bool mapped = !(fl & 0x4);
  int start = 0, end = 0;
  uint32_t h32 = 0;
  bool exact = false, hashed = false;
  if (strategy == TBK_STRAT_EXON) {
    int nex = 0, e1 = 0, s2 = 0, ix = 0;
    const int l = walk_exons(pos, c, nc,
                             [&](int s, int e) {
                               e1 = ix == 0 ? e : e1;
                               s2 = ix == 1 ? s : s2;
                               ++ix;
                             },
                             &nex);
    if (mapped) {
      start = pos + 1;
      end = pos + l;
    }
    if (nex == 1) {
      h32 = 0x8000000Fu;
      exact = true;
    }
    if (nex == 2) {
      const uint32_t a = (uint32_t)(e1 - pos), g = (uint32_t)(s2 - e1 - 1);  // first exon length, gap
      if (a < (1u << 10) && g < (1u << 20)) {
        h32 = 0xC0000000u | (a << 20) | g;
        exact = true;
      }
    }
  } else {
    const CigScan v = cig_scan(c, nc, strategy == TBK_STRAT_CLIP);
    if (mapped) {
      start = pos + 1;
      end = pos + v.l;
    }
    if (strategy != TBK_STRAT_FULL) {
      const uint32_t vn = v.e - v.b;
      if (vn == 1 && ((0x18Du >> cig_op(v.v0)) & 1u)) {
        h32 = 0x80000000u | cig_op(v.v0);
        exact = true;
      }
      if (vn == 3 && cig_op(v.v0) == C_M && cig_op(v.v1) == C_N && cig_op(v.v2) == C_M && cig_len(v.v0) < (1u << 10) &&
          cig_len(v.v1) < (1u << 20)) {
        h32 = 0xC0000000u | (cig_len(v.v0) << 20) | cig_len(v.v1);
        exact = true;
      }
      if (pass && !exact) h32 = (uint32_t)(view_hash(O.seed, v, c) >> 32) & O.hash_mask & 0x7FFFFFFFu;
      hashed = true;
    }
  }
  if (pass && !exact && !hashed) h32 = (uint32_t)(strategy_hash(I, O, strategy, i, c, nc) >> 32) & O.hash_mask & 0x7FFFFFFFu;
  if (!pass) h32 = 0;
  int64_t span = (int64_t)end - (int64_t)start + 1;
  RecKey K;
  K.err = 0;
  if (pass && (span < 0 || span >= (1ll << 30) || start < 0 || tidv < -1)) {  // key fields: tid+1 and start need 31 bits
    K.err = TBK_DERR_SPAN;
    span = 0;
  }
  K.hi = ((uint64_t)(uint32_t)(tidv + 1) << 33) | ((uint64_t)(uint32_t)start << 2) | sc;
  K.lo = ((uint64_t)span << 32) | h32;
  K.end = end;
  K.pass = pass;
  return K;
}

// exact equality of the strategy keys of two records (start/end/strand are already equal)
__device__ inline bool strategy_equal(const ColIn& I, int strategy, uint32_t a, uint32_t b) {
  const uint32_t* ca = I.cig + I.cig_off[a];
  const uint32_t* cb = I.cig + I.cig_off[b];
  uint32_t na = I.cig_off[a + 1] - I.cig_off[a], nb = I.cig_off[b + 1] - I.cig_off[b];
  switch (strategy) {
    case TBK_STRAT_CIGAR:
    case TBK_STRAT_FULL: {
      if (na != nb) return false;
      for (uint32_t k = 0; k < na; ++k)
        if (ca[k] != cb[k]) return false;
      if (strategy == TBK_STRAT_FULL) {
        uint32_t ha = I.md_has[a], hb = I.md_has[b];
        if (ha != hb) return false;
        if (ha) {
          uint32_t la = I.md_off[a + 1] - I.md_off[a], lb = I.md_off[b + 1] - I.md_off[b];
          if (la != lb) return false;
          for (uint32_t k = 0; k < la; ++k)
            if (I.md[I.md_off[a] + k] != I.md[I.md_off[b] + k]) return false;
        }
      }
      return true;
    }
    case TBK_STRAT_CLIP: {
      uint32_t ba, ea, bb, eb;
      clip_view(ca, na, &ba, &ea);
      clip_view(cb, nb, &bb, &eb);
      if (ea - ba != eb - bb) return false;
      for (uint32_t k = 0; k < ea - ba; ++k)
        if (ca[ba + k] != cb[bb + k]) return false;
      return true;
    }
    case TBK_STRAT_EXON: {
      // same exon list: hash both walks with two independent seeds and compare element-wise through
      // a lock-step re-walk: exon lists are short, so walk b for every exon index of a
      int nxa = 0, nxb = 0;
      bool eq = true;
      int ia = 0;
      walk_exons(I.pos[a], ca, na,
                 [&](int s, int e) {
                   int ib = 0, cnt = 0;
                   bool found = false;
                   walk_exons(I.pos[b], cb, nb,
                              [&](int s2, int e2) {
                                if (ib == ia) found = (s2 == s && e2 == e);
                                ++ib;
                              },
                              &cnt);
                   if (!found) eq = false;
                   ++ia;
                 },
                 &nxa);
      walk_exons(I.pos[b], cb, nb, [](int, int) {}, &nxb);
      return eq && nxa == nxb;
    }
  }
  return false;
}

// three-way compare of the strategy keys in the reference's order (cmpCigar & co, tiebrush.cpp:285-345)
__device__ inline int strategy_cmp(const ColIn& I, int strategy, uint32_t a, uint32_t b) {
  const uint32_t* ca = I.cig + I.cig_off[a];
  const uint32_t* cb = I.cig + I.cig_off[b];
  uint32_t na = I.cig_off[a + 1] - I.cig_off[a], nb = I.cig_off[b + 1] - I.cig_off[b];
  auto memcmp_u32 = [](const uint32_t* x, const uint32_t* y, uint32_t n) -> int {
    for (uint32_t k = 0; k < n; ++k) {
      if (x[k] != y[k]) {  // memcmp over little-endian words: lowest byte first
        uint32_t xs = __builtin_bswap32(x[k]), ys = __builtin_bswap32(y[k]);
        return xs < ys ? -1 : 1;
      }
    }
    return 0;
  };
  switch (strategy) {
    case TBK_STRAT_CIGAR:
    case TBK_STRAT_FULL: {
      if (na != nb) return (int)na - (int)nb;
      int c = memcmp_u32(ca, cb, na);
      if (c != 0 || strategy == TBK_STRAT_CIGAR) return c;
      uint32_t ha = I.md_has[a], hb = I.md_has[b];
      if (!ha || !hb) {
        if (ha == hb) return 0;
        return ha ? 1 : -1;
      }
      uint32_t la = I.md_off[a + 1] - I.md_off[a], lb = I.md_off[b + 1] - I.md_off[b];
      uint32_t m = la < lb ? la : lb;
      for (uint32_t k = 0; k < m; ++k) {
        uint8_t x = I.md[I.md_off[a] + k], y = I.md[I.md_off[b] + k];
        if (x != y) return x < y ? -1 : 1;
      }
      if (la == lb) return 0;
      return la < lb ? -1 : 1;
    }
    case TBK_STRAT_CLIP: {
      uint32_t ba, ea, bb, eb;
      clip_view(ca, na, &ba, &ea);
      clip_view(cb, nb, &bb, &eb);
      if (ea - ba != eb - bb) return (int)(ea - ba) - (int)(eb - bb);
      return memcmp_u32(ca + ba, cb + bb, ea - ba);
    }
    case TBK_STRAT_EXON: {
      int nxa = 0, nxb = 0;
      walk_exons(I.pos[a], ca, na, [](int, int) {}, &nxa);
      walk_exons(I.pos[b], cb, nb, [](int, int) {}, &nxb);
      if (nxa != nxb) return nxa - nxb;
      int res = 0, ia = 0;
      walk_exons(I.pos[a], ca, na,
                 [&](int s, int e) {
                   if (res == 0) {
                     int ib = 0, cnt = 0;
                     walk_exons(I.pos[b], cb, nb,
                                [&](int s2, int e2) {
                                  if (ib == ia && res == 0) {
                                    if (s != s2)
                                      res = s - s2;
                                    else if (e != e2)
                                      res = e - e2;
                                  }
                                  ++ib;
                                },
                                &cnt);
                   }
                   ++ia;
                 },
                 &nxa);
      return res;
    }
  }
  return 0;
}

}  // namespace tbkd
using namespace tbkd;
