// collapse.hip — tiebrush hot path on gfx950: k-way coordinate merge order, duplicate-group
// detection, YC/YX reduction, representative selection, bucket ordering and the YD machine.
//
// Reference semantics (paths relative to /root/reference/src):
//   merge order      tmerge.h:28-50, tmerge.cpp:331-344: greedy k-way merge on (tid,start,end,fidx)
//                    == sort by (per-file prefix-max of (tid,start,end), fidx, idx)   [SURVEY.md §3.1]
//   filter           tiebrush.cpp:532-541 (passes_options) — applied AFTER the order is fixed
//   bucket / group   tiebrush.cpp:477-499, :438-457: same (tid,start) bucket; group key
//                    (tstrand,end,strategy key); list order = (strand char, end, strategy compare)
//   accumulate       tiebrush.cpp:378-436 (settle/dupAdd), flush :501-530
//   YD               tiebrush.cpp:111-250 (GSegList) called per sample & strand list at flush
//
// GPU formulation.  Every record gets a 128-bit sort key
//      hi = (tid+1 : 31 | start : 31 | strand code : 2) -> bucket order, then strand ('+' < '-' < '.')
//      lo = (span : 32 | h32)                            -> end order, then a 32-bit slice of a seeded
//                                                           64-bit hash of the strategy key
// and the passing records are LSD-radix sorted (stable, so equal keys stay in file-major order).
// Groups are runs of equal keys; adjacent members are verified against the full strategy key, a
// hash collision raises TBK_DERR_COLLISION and the host retries with another seed.  Per group a
// wave-segmented reduction + one atomic per (wave, group) gives YC, |samples|, sum YX, max YD and
// the representative = argmin (effend, record index), where effend is the per-file running max of
// `end` (the merge-order key).  Groups that tie on (bucket,strand,end) are re-ordered by the
// reference comparator (n_cigar, memcmp, ...).  The YD list machine is sequential per
// (sample,strand) list; it is cut at provable renewal points (read start beyond every earlier end
// of that list, or a chromosome change) into independent chains, one GPU thread each.
#include <stdlib.h>

#include <chrono>

#include "dev_common.hpp"
#include "rx_w64.hpp"
#include "scan_op.hpp"
#include "strategy.hpp"
#include "tbk_internal.h"
#include "wgroup.h"

namespace {

// ---- K1: keys --------------------------------------------------------------------------------------
struct EffKey {  // scan element: lexicographic running max of (hi, end) per file + count of passing records (all 32-bit words)
  uint32_t khi_h, khi_l;
  int32_t kend;
  uint32_t flag_cnt;  // bit 31 = a file head lies in the covered span, bits 0..30 = passing records
};
struct EffOp {
  __device__ __forceinline__ EffKey operator()(const EffKey& a, const EffKey& b) const {
    uint64_t ak = ((uint64_t)a.khi_h << 32) | a.khi_l, bk = ((uint64_t)b.khi_h << 32) | b.khi_l;
    bool take_b = (b.flag_cnt >> 31) || bk > ak || (bk == ak && b.kend > a.kend);
    EffKey r;
    r.khi_h = take_b ? b.khi_h : a.khi_h;
    r.khi_l = take_b ? b.khi_l : a.khi_l;
    r.kend = take_b ? b.kend : a.kend;
    r.flag_cnt = ((a.flag_cnt | b.flag_cnt) & 0x80000000u) | ((a.flag_cnt + b.flag_cnt) & 0x7FFFFFFFu);
    return r;
  }
};

__global__ __launch_bounds__(256) void col_keys_k(ColIn I, ColOpt O, uint64_t* __restrict__ khi, uint64_t* __restrict__ klo, int32_t* __restrict__ kend,
                                                  uint8_t* __restrict__ kflags /*bit0 pass, bit1 file head*/, uint16_t* __restrict__ fidx,
                                                  uint32_t* __restrict__ err) {
  // the file of the block's first record by one bisection; every thread walks on from there (a block rarely spans two files)
  __shared__ uint32_t s_f;
  const uint32_t i0 = blockIdx.x * blockDim.x;
  if (threadIdx.x == 0) {
    uint32_t lo = 0, hi = I.k;  // last f with file_off[f] <= i0
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (I.file_off[mid] <= i0)
        lo = mid;
      else
        hi = mid;
    }
    s_f = lo;
  }
  const uint32_t i = i0 + threadIdx.x;
  const bool in = i < I.n;
  // the record's fields: independent loads, all in flight before the first use
  const uint16_t fl = in ? I.flag[i] : (uint16_t)0x4;
  const int pos = in ? I.pos[i] : 0;
  const int tidv = in ? I.tid[i] : 0;
  const int mq = in ? (int)I.mapq[i] : 0;
  const auto nhv = in ? I.nh[i] : TBK_NH_ABSENT;
  const uint32_t sc = in ? strand_code(I.strand[i]) : 0u;
  const uint32_t c0 = in ? I.cig_off[i] : 0u, c1 = in ? I.cig_off[i + 1] : 0u;
  __syncthreads();
  if (!in) return;
  uint32_t f = s_f;
  while (f + 1 < I.k && I.file_off[f + 1] <= i) ++f;
  const RecKey K = record_key(I, O, i, fl, pos, tidv, mq, nhv, sc, I.cig + c0, c1 - c0);
  if (K.err) atomicOr(err, K.err);
  const bool pass = K.pass;
  khi[i] = K.hi;
  klo[i] = K.lo;
  kend[i] = K.end;
  kflags[i] = (pass ? 1u : 0u) | (i == I.file_off[f] ? 2u : 0u);
  fidx[i] = (uint16_t)f;
}

struct EffLoad {
  const uint64_t* khi;
  const int32_t* kend;
  const uint8_t* kflags;
  __device__ __forceinline__ EffKey operator()(uint32_t i) const {
    EffKey e;
    uint64_t k = khi[i] >> 2;  // (tid,start) only: the strand bits are not part of the merge key
    e.khi_h = (uint32_t)(k >> 32);
    e.khi_l = (uint32_t)k;
    e.kend = kend[i];
    uint8_t f = kflags[i];
    e.flag_cnt = ((uint32_t)((f >> 1) & 1u) << 31) | (f & 1u);
    return e;
  }
};
struct EffStore {
  const uint64_t* khi;
  const uint64_t* klo;
  const uint8_t* kflags;
  int32_t* effend;
  uint64_t* chi;
  uint64_t* clo;
  uint32_t* cval;
  uint64_t* n_pass;
  uint32_t n;
  uint32_t* err;
  const uint16_t* fidx;
  uint32_t* head_off;  // [file] compacted offset of the file's first record (files without records: untouched)
  uint32_t* ceff;      // optional (window path): effective end — or the explicit merge priority — in compacted order
  const uint64_t* prio;
  __device__ __forceinline__ void operator()(uint32_t i, const EffKey&, const EffKey& inc, const EffKey& ex) const {
    if (kflags[i] & 2u) head_off[fidx[i]] = ex.flag_cnt & 0x7FFFFFFFu;
    if (kflags[i] & 1u) {
      uint64_t ik = ((uint64_t)inc.khi_h << 32) | inc.khi_l;
      if (ik != (khi[i] >> 2)) atomicOr(err, TBK_DERR_UNSORTED);  // an earlier record of the file has a larger (tid,start)
      effend[i] = inc.kend;
      uint32_t d = ex.flag_cnt & 0x7FFFFFFFu;
      chi[d] = khi[i];
      clo[d] = klo[i];
      cval[d] = i;
      if (ceff) ceff[d] = prio ? (uint32_t)prio[i] : (uint32_t)inc.kend;
    }
    if (i + 1 == n) *n_pass = inc.flag_cnt & 0x7FFFFFFFu;
  }
};

// run_off[f] = compacted offset where file f's passing records begin (f == k: their total); empty files take the
// offset of the next file that has records
__global__ void col_runs_k(uint32_t k, const uint32_t* __restrict__ file_off, const uint32_t* __restrict__ head_off,
                           const uint64_t* __restrict__ n_pass, uint32_t* __restrict__ run_off) {
  uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f > k) return;
  uint32_t g = f;
  while (g < k && file_off[g] == file_off[g + 1]) ++g;
  run_off[f] = g < k ? head_off[g] : (uint32_t)*n_pass;
}

// ---- K4: heads -------------------------------------------------------------------------------------
// flags: bit0 group head, bit1 tie-set head, bit2 file head (first record of its file in the group)
__global__ void col_heads_k(ColIn I, int strategy, const uint64_t* __restrict__ pm, uint32_t m_hi, const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo,
                            const uint32_t* __restrict__ val, const uint16_t* __restrict__ fidx, uint8_t* __restrict__ flags,
                            uint32_t* __restrict__ ghead, uint32_t* __restrict__ err) {
  const uint32_t m = (uint32_t)*pm;
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) {  // (the grid covers the host's upper bound m_hi: the scan behind this kernel reads zeros there)
    if (q < m_hi) ghead[q] = 0u;
    return;
  }
  uint32_t gi = val[q];
  bool bucket_head = true, group_head = true, tie_head = true, file_head = true;
  if (q > 0) {
    uint32_t pv = val[q - 1];
    bucket_head = (hi[q] >> 2) != (hi[q - 1] >> 2);
    group_head = hi[q] != hi[q - 1] || lo[q] != lo[q - 1];
    tie_head = hi[q] != hi[q - 1] || (lo[q] >> 32) != (lo[q - 1] >> 32);
    (void)bucket_head;
    if (!group_head && !strategy_equal(I, strategy, gi, pv)) atomicOr(err, TBK_DERR_COLLISION);
    file_head = group_head || fidx[gi] != fidx[pv];
  }
  flags[q] = (group_head ? 1u : 0u) | (tie_head ? 2u : 0u) | (file_head ? 4u : 0u);
  ghead[q] = group_head ? 1u : 0u;
}

struct GroupAcc {
  double* yc;
  uint32_t* ns;
  long long* yxin;
  long long* ydin;
  unsigned long long* rep;
  uint32_t* first;  // sorted position of the group head
  uint8_t* tie;
};

// wave-segmented reductions over lanes holding equal (sorted) keys
template <class T, class Op>
__device__ __forceinline__ T seg_reduce(T v, uint32_t key, Op op) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    uint32_t ok = __shfl_up(key, d, 64);
    if ((int)lane_id() >= d && ok == key) v = op(v, o);
  }
  return v;
}

__global__ void col_reduce_k(ColIn I, ColOpt O, const uint64_t* __restrict__ pm, const uint32_t* __restrict__ val, const uint8_t* __restrict__ flags,
                             const uint32_t* __restrict__ gex, const uint16_t* __restrict__ fidx,
                             const int32_t* __restrict__ effend, GroupAcc G, uint32_t* __restrict__ sgid, uint32_t* __restrict__ err) {
  const uint32_t m = (uint32_t)*pm;
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  bool act = q < m;
  uint32_t sg = 0xFFFFFFFFu;
  double yc = 0.0;
  uint32_t ns = 0;
  long long yxin = 0, ydin = 0;
  unsigned long long rep = ~0ull;
  if (act) {
    uint8_t fl = flags[q];
    sg = gex[q] + (fl & 1u) - 1u;
    sgid[q] = sg;
    uint32_t gi = val[q];
    uint32_t f = fidx[gi];
    if (I.tbm[f]) {  // settle/dupAdd, TieBrush input (tiebrush.cpp:389-395, :412-419)
      yc = I.yc_in[gi];
      if (yc == 0.0) yc = 1.0;
      if (yc != rint(yc) || fabs(yc) > 9.0e15) atomicOr(err, TBK_DERR_FRACTIONAL);
      yxin = I.yx_in[gi];
      ydin = I.yd_in[gi];
    } else {
      if (O.store_frac) {
        int nh = I.nh[gi] == TBK_NH_ABSENT ? 1 : I.nh[gi];
        yc = 1.0 / nh;
      } else {
        yc = 1.0;
      }
      ns = (fl & 4u) ? 1u : 0u;
    }
    rep = ((unsigned long long)(uint32_t)effend[gi] << 32) | gi;
    if (fl & 1u) {
      G.first[sg] = q;
      G.tie[sg] = (fl >> 1) & 1u;
    }
  }
  yc = seg_reduce(yc, sg, [](double a, double b) { return a + b; });
  ns = seg_reduce(ns, sg, [](uint32_t a, uint32_t b) { return a + b; });
  yxin = seg_reduce(yxin, sg, [](long long a, long long b) { return a + b; });
  ydin = seg_reduce(ydin, sg, [](long long a, long long b) { return a > b ? a : b; });
  rep = seg_reduce(rep, sg, [](unsigned long long a, unsigned long long b) { return a < b ? a : b; });
  uint32_t nxt = __shfl_down(sg, 1, 64);
  bool last = act && (lane_id() == 63 || nxt != sg);
  if (last) {
    atomicAdd(&G.yc[sg], yc);
    if (ns) atomicAdd(&G.ns[sg], ns);
    if (yxin) atomicAdd((unsigned long long*)&G.yxin[sg], (unsigned long long)yxin);
    if (ydin > 0) __hip_atomic_fetch_max(&G.ydin[sg], ydin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicMin(&G.rep[sg], rep);
  }
}

// -A (collapse_same): a non-first record of its file whose (qname,pairOrder) equals the representative's
// is not counted (tiebrush.cpp:422-424)
__global__ void col_same_k(ColIn I, ColOpt O, const uint64_t* __restrict__ pm, const uint32_t* __restrict__ val, const uint8_t* __restrict__ flags,
                           const uint32_t* __restrict__ sgid, const uint16_t* __restrict__ fidx, GroupAcc G) {
  const uint32_t m = (uint32_t)*pm;
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  uint32_t gi = val[q];
  if (I.tbm[fidx[gi]] || (flags[q] & 4u)) return;
  uint32_t sg = sgid[q];
  uint32_t r = (uint32_t)(G.rep[sg] & 0xFFFFFFFFull);
  if (same_read(I, gi, r)) atomicAdd(&G.yc[sg], -1.0);
}

// ---- ordered YC accumulation (--store-frac, fractional carried YC) ------------------------------------------------
// accYC is a double accumulated in MERGE order (settle, then dupAdd in pop order: tiebrush.cpp:378-436).  Integer
// counts are exact in any order; fractional terms are not, so the members of every group are re-sorted by
// (group, effend) — stable, hence (effend, record index) = merge order — and summed by one thread per group.
__global__ void ord_fill_k(uint32_t m, const uint32_t* __restrict__ val, const uint32_t* __restrict__ sgid,
                           const int32_t* __restrict__ effend, const uint64_t* __restrict__ prio_hi, const uint8_t* __restrict__ flags,
                           uint64_t* __restrict__ hi, uint64_t* __restrict__ lo, uint32_t* __restrict__ v, uint8_t* __restrict__ fh_by_rec) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  uint32_t gi = val[q];
  hi[q] = sgid[q];
  // (cross-rank tiles: the merge order was fixed where the files live — the tile lacks the filtered records the scan would need)
  lo[q] = prio_hi ? (uint32_t)prio_hi[gi] : (uint32_t)effend[gi];
  v[q] = gi;
  fh_by_rec[gi] = (flags[q] >> 2) & 1u;  // first record of its file inside the group
}
__global__ void ord_sum_k(ColIn I, ColOpt O, uint32_t ng, uint32_t m, const uint32_t* __restrict__ v, const uint16_t* __restrict__ fidx,
                          const uint8_t* __restrict__ fh_by_rec, GroupAcc G) {
  uint32_t sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= ng) return;
  uint32_t q0 = G.first[sg], q1 = (sg + 1 < ng) ? G.first[sg + 1] : m;
  uint32_t rep = (uint32_t)(G.rep[sg] & 0xFFFFFFFFull);
  double acc = 0.0;
  for (uint32_t q = q0; q < q1; ++q) {
    uint32_t gi = v[q];
    double y;
    if (I.tbm[fidx[gi]]) {
      y = I.yc_in[gi];
      if (y == 0.0) y = 1.0;
    } else if (O.collapse_same && !fh_by_rec[gi] && same_read(I, gi, rep)) {
      continue;  // -A: same read of the same sample is not counted again (tiebrush.cpp:422-424)
    } else if (O.store_frac) {
      int nh = I.nh[gi] == TBK_NH_ABSENT ? 1 : I.nh[gi];
      y = 1.0 / nh;
    } else {
      y = 1.0;
    }
    acc = (q == q0) ? y : acc + y;
  }
  G.yc[sg] = acc;
}

// cross-rank stitch: the representative is the member with the smallest explicit priority (groups have at most one
// member per rank, so a serial scan of the group's contiguous members is cheap)
__global__ void col_rep_prio_k(ColIn I, const uint64_t* __restrict__ png, const uint64_t* __restrict__ pm, const uint32_t* __restrict__ val, GroupAcc G) {
  const uint32_t ng = (uint32_t)*png, m = (uint32_t)*pm;
  uint32_t sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= ng) return;
  uint32_t q0 = G.first[sg], q1 = (sg + 1 < ng) ? G.first[sg + 1] : m;
  uint32_t best = val[q0];
  for (uint32_t q = q0 + 1; q < q1; ++q) {
    uint32_t gi = val[q];
    if (I.prio_hi[gi] < I.prio_hi[best] || (I.prio_hi[gi] == I.prio_hi[best] && I.prio_lo[gi] < I.prio_lo[best])) best = gi;
  }
  G.rep[sg] = (G.rep[sg] & 0xFFFFFFFF00000000ull) | best;
}

// ---- tie sets: order groups that share (bucket,strand,end) by the reference comparator -----------------
// One thread per tie set (its head): identity order for the set, insertion sort by the reference comparator on any
// member (the group head) when the set has more than one group, then the inverse permutation of the set's range.
__global__ void col_tie_sort_k(ColIn I, int strategy, const uint64_t* __restrict__ png, const uint32_t* __restrict__ val, GroupAcc G,
                               uint32_t* __restrict__ gperm, uint32_t* __restrict__ ginv) {
  const uint32_t ng = (uint32_t)*png;
  uint32_t sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= ng || !G.tie[sg]) return;
  uint32_t e = 1;
  while (sg + e < ng && !G.tie[sg + e]) ++e;
  for (uint32_t a = 0; a < e; ++a) gperm[sg + a] = sg + a;
  for (uint32_t a = 1; a < e; ++a) {
    uint32_t x = gperm[sg + a];
    uint32_t rx = val[G.first[x]];
    uint32_t b = a;
    while (b > 0) {
      uint32_t y = gperm[sg + b - 1];
      if (strategy_cmp(I, strategy, rx, val[G.first[y]]) < 0) {
        gperm[sg + b] = y;
        --b;
      } else {
        break;
      }
    }
    gperm[sg + b] = x;
  }
  for (uint32_t a = 0; a < e; ++a) ginv[gperm[sg + a]] = sg + a;
}

// ---- YD ----------------------------------------------------------------------------------------------------
// incidence items: one per (group, sample list) the flush touches (tiebrush.cpp:511-521)
__global__ void yd_count_k(ColIn I, uint32_t m, const uint32_t* __restrict__ val, const uint8_t* __restrict__ flags,
                           const uint16_t* __restrict__ fidx, uint32_t* __restrict__ cnt) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  uint32_t gi = val[q];
  uint32_t c = 0;
  if ((flags[q] & 4u) && !I.tbm[fidx[gi]]) c = (I.strand[gi] == '+' || I.strand[gi] == '-') ? 1u : 2u;
  cnt[q] = c;
}
__global__ void yd_fill_k(ColIn I, uint32_t m, const uint32_t* __restrict__ val, const uint8_t* __restrict__ flags,
                          const uint16_t* __restrict__ fidx, const uint32_t* __restrict__ sgid, const uint32_t* __restrict__ ginv,
                          const uint32_t* __restrict__ off, const uint32_t* __restrict__ ooff, GroupAcc G, uint64_t* __restrict__ item) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  uint32_t gi = val[q];
  uint32_t f = fidx[gi];
  if (!(flags[q] & 4u) || I.tbm[f]) return;
  uint32_t sg = sgid[q];
  uint32_t o = ginv[sg];
  uint32_t p = ooff[o] + (off[q] - off[G.first[sg]]);  // position in output order
  uint8_t s = I.strand[gi];
  if (s != '-') {  // '+' or '.': fsegs[f]
    item[p] = ((uint64_t)(f * 2) << 32) | o;
    ++p;
  }
  if (s != '+') {  // '-' or '.': rsegs[f]
    item[p] = ((uint64_t)(f * 2 + 1) << 32) | o;
  }
}

// items are generated directly in OUTPUT order (group o, then file order inside the group): the later sort then only
// has to split them by list id (stable), not order them by group
__global__ void yd_gcount_k(uint32_t ng, uint32_t m, const uint32_t* __restrict__ gperm, GroupAcc G, const uint32_t* __restrict__ ioff,
                            const uint32_t* __restrict__ icnt, uint32_t* __restrict__ ocnt) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  uint32_t sg = gperm[o];
  uint32_t q0 = G.first[sg];
  uint32_t end = (sg + 1 < ng) ? ioff[G.first[sg + 1]] : (ioff[m - 1] + icnt[m - 1]);
  ocnt[o] = end - ioff[q0];
}

// window path: the (group, sample) incidences exist already (WgOut::pfile / pgrp, a group's incidences in file order)
__global__ void yd_gcount_w_k(uint32_t ng, const uint32_t* __restrict__ gperm, GroupAcc G, const uint64_t* __restrict__ ghi,
                              uint32_t* __restrict__ ocnt) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  const uint32_t sg = gperm[o];
  ocnt[o] = G.ns[sg] * (((uint32_t)ghi[sg] & 3u) == 2u ? 2u : 1u);  // '.' feeds both lists (tiebrush.cpp:515-520)
}
__global__ void yd_fill_w_k(uint32_t np, const uint16_t* __restrict__ pfile, const uint32_t* __restrict__ pgrp,
                            const uint32_t* __restrict__ gpoff, const uint32_t* __restrict__ ginv, const uint32_t* __restrict__ ooff,
                            const uint64_t* __restrict__ ghi, uint64_t* __restrict__ item) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const uint32_t sg = pgrp[p];
  const uint32_t o = ginv[sg];
  const uint32_t c = (uint32_t)ghi[sg] & 3u;  // strand code: 0 '+', 1 '-', 2 '.'
  uint32_t pos = ooff[o] + (p - gpoff[sg]) * (c == 2u ? 2u : 1u);
  const uint32_t f = pfile[p];
  if (c != 1u) {  // '+' or '.': fsegs[f]
    item[pos] = ((uint64_t)(f * 2) << 32) | o;
    ++pos;
  }
  if (c != 0u) {  // '-' or '.': rsegs[f]
    item[pos] = ((uint64_t)(f * 2 + 1) << 32) | o;
  }
}
__global__ void col_recgroup_w_k(uint32_t n, const uint32_t* __restrict__ rec_sg, const uint32_t* __restrict__ ginv, int32_t* __restrict__ rec_group) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t sg = rec_sg[i];
  rec_group[i] = sg == 0xFFFFFFFFu ? -1 : (int32_t)ginv[sg];
}


// ---- items by list without a sort (window path, <= 64 input files: <= 128 lists) ---------------------------------------------
// An item is a (group, list) pair, a list a (file, strand list) pair, and the chain kernels want the items of a list together, in
// group order.  Round 2 wrote one 8-byte word per item in group order and split the 178 M words of config 3 by list with a stable
// radix pass (count 0.1 + fill 0.7 + histogram 0.7 + scatter 2.8 ms).  But a group holds at most one item per list: the lists of
// the 64 groups of a wave form a 64 x 128 bit matrix whose column c, read off with one ballot, gives every group of the wave its
// rank in list c.  So: one pass counts the items per (list, tile of 256 groups), a row scan turns the counts into offsets, and
// the second pass places every item — coordinates, exon offset and item word — where its list wants it.  No item array is
// written twice, none is sorted.
constexpr uint32_t YS_NT = 1024;  // groups per tile (one thread each)
constexpr uint32_t YS_NL = 128;   // lists
struct YsIn {
  uint32_t ng;
  const uint32_t* gperm;  // output order -> group
  const uint32_t* ns;     // samples of a group
  const uint32_t* gpoff;  // first incidence of a group (its incidences are in file order)
  const uint16_t* pfile;
  const uint64_t* ghi;    // group key: strand code in the low two bits
  const uint64_t* gfmask; // the group's files as a bit mask (WgOut::gfmask), or null: walk the incidences
};
// the lists of a group as a 128-bit mask, from the 64-bit set of its files and its strand code ('.' feeds both lists of a file,
// tiebrush.cpp:515-520)
__device__ __forceinline__ uint64_t ys_spread32(uint64_t x) {  // bit f of the low 32 -> bit 2 f
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}
__device__ __forceinline__ void ys_mask_from_files(uint64_t files, uint32_t c, uint64_t* lo, uint64_t* hi) {
  const uint64_t a = ys_spread32(files & 0xFFFFFFFFull), b = ys_spread32(files >> 32);
  *lo = c == 0u ? a : (c == 1u ? a << 1 : (a | (a << 1)));
  *hi = c == 0u ? b : (c == 1u ? b << 1 : (b | (b << 1)));
}
// 64 x 64 bit matrix, one row per lane, transposed in six exchange steps: afterwards lane c holds column c (bit r = row r's bit c)
__device__ __forceinline__ uint64_t wave_bit_transpose(uint64_t x) {
  const uint32_t lane = lane_id();
#define YS_TSTEP(J, M)                                                                     \
  {                                                                                        \
    const uint64_t tt = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), J, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)x, J, 64); \
    x = (lane & J) ? (((tt >> J) & M) | (x & ~M)) : ((x & M) | ((tt & M) << J));           \
  }
  YS_TSTEP(32, 0x00000000FFFFFFFFull)
  YS_TSTEP(16, 0x0000FFFF0000FFFFull)
  YS_TSTEP(8, 0x00FF00FF00FF00FFull)
  YS_TSTEP(4, 0x0F0F0F0F0F0F0F0Full)
  YS_TSTEP(2, 0x3333333333333333ull)
  YS_TSTEP(1, 0x5555555555555555ull)
#undef YS_TSTEP
  return x;
}
// The chain heads come with the placement.  An item opens a chain when it is the first of its list, when its reference sequence
// differs from the item before it in the list (rspacing.reset(), tiebrush.cpp:586-589), or when it starts beyond the running maximum
// of end + 1 of the items before it on that reference (the list renews itself, processRead :230-241; end + 1: a CIGAR that ends in an
// intron leaves a last exon (end + 1, end), GSam.cpp:351-417, a node a read starting at end + 1 does not clear).  What a tile of groups
// has to know of the items before it is, per list, an aggregate of this monoid — so the first pass also folds every (list, tile)
// cell's items into one, a scan along each list's row gives every cell the aggregate of the cells before it, and the second pass
// walks a cell's items once more, in order, to flag its heads.  (Round 2 and the radix path find the heads afterwards, in a scan over
// all items: 2.2 ms for 178 M items on config 3.)
struct YsAgg {
  uint32_t first_tid, last_tid;  // tid + 1 of the first / last item; last_tid == 0: no item
  int32_t mx;                    // maximum of end + 1 over the trailing items that share last_tid
  uint32_t whole;                // every item shares one tid
};
__device__ __forceinline__ YsAgg ys_combine(const YsAgg& a, const YsAgg& b) {
  if (b.last_tid == 0u) return a;
  if (a.last_tid == 0u) return b;
  YsAgg r;
  r.first_tid = a.first_tid;
  r.last_tid = b.last_tid;
  const bool joins = b.whole && b.first_tid == a.last_tid;
  r.mx = joins ? (a.mx > b.mx ? a.mx : b.mx) : b.mx;
  r.whole = joins ? a.whole : 0u;
  return r;
}
// A tile's items of list c are the set bits of column c, wave by wave.  The walk is two-level: every (wave, list) cell — two per
// thread — folds its few items (three or four on average), thread c chains the sixteen cells of its list (ys_chain_cells), and the
// cells are walked once more where the heads are wanted.  Aggregates live in LDS as three words (the `whole` bit rides in bit 31
// of first_tid: tid + 1 < 2^31).
struct YsCells {
  uint32_t (*a)[YS_NL];  // first_tid | whole << 31
  uint32_t (*b)[YS_NL];  // last_tid
  int32_t (*m)[YS_NL];   // mx
  __device__ __forceinline__ YsAgg get(uint32_t x, uint32_t c) const { return YsAgg{a[x][c] & 0x7FFFFFFFu, b[x][c], m[x][c], a[x][c] >> 31}; }
  __device__ __forceinline__ void put(uint32_t x, uint32_t c, const YsAgg& v) const {
    a[x][c] = v.first_tid | (v.whole << 31);
    b[x][c] = v.last_tid;
    m[x][c] = v.mx;
  }
};
// cells -> the aggregate of the list's items before each cell (start: before the tile); returns the aggregate behind the last cell
__device__ __forceinline__ YsAgg ys_chain_cells(const YsCells& P, uint32_t c, YsAgg run) {
  for (uint32_t x = 0; x < YS_NT / 64; ++x) {
    const YsAgg mine = P.get(x, c);
    P.put(x, c, run);
    run = ys_combine(run, mine);
  }
  return run;
}
__global__ __launch_bounds__(YS_NT) void yd_lcount_k(YsIn S, const uint4* __restrict__ gpk /* YdGroups::pk */, uint32_t ntiles,
                                                     uint32_t* __restrict__ table /* [YS_NL][ntiles] */, uint4* __restrict__ agg /* [YS_NL][ntiles] */,
                                                     uint64_t* __restrict__ gfiles /* [2 ng]: the files of output group o, its strand code */) {
  __shared__ uint64_t wb[YS_NT / 64][YS_NL];
  __shared__ uint32_t g_tid[YS_NT];
  __shared__ int32_t g_e1[YS_NT];
  __shared__ uint32_t pa[YS_NT / 64][YS_NL], pb[YS_NT / 64][YS_NL];
  __shared__ int32_t pm[YS_NT / 64][YS_NL];
  const YsCells P{pa, pb, pm};
  uint64_t lo = 0, hi = 0;
  {
    const uint32_t o = blockIdx.x * YS_NT + threadIdx.x;
    if (o < S.ng) {
      const uint32_t sg = S.gperm[o];
      const uint32_t c = (uint32_t)S.ghi[sg] & 3u;
      const uint32_t n = S.gfmask ? 0u : S.ns[sg], p0 = S.gfmask ? 0u : S.gpoff[sg];
      uint64_t files = 0;
      if (S.gfmask)
        files = S.gfmask[sg];
      else
        for (uint32_t i = 0; i < n; ++i) files |= 1ull << S.pfile[p0 + i];
      gfiles[2 * (size_t)o] = files;
      gfiles[2 * (size_t)o + 1] = c;
      ys_mask_from_files(files, c, &lo, &hi);
      const uint4 g = gpk[o];
      g_tid[threadIdx.x] = g.x;
      g_e1[threadIdx.x] = (int32_t)g.z + 1;
    }
  }
  wb[threadIdx.x >> 6][lane_id()] = wave_bit_transpose(lo);  // lane c: the groups of this wave in lists c and 64 + c
  wb[threadIdx.x >> 6][64u + lane_id()] = wave_bit_transpose(hi);
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < (YS_NT / 64) * YS_NL; q += YS_NT) {
    const uint32_t x = q / YS_NL, c = q % YS_NL;
    YsAgg A{0u, 0u, INT32_MIN, 1u};
    for (uint64_t m = wb[x][c]; m; m &= m - 1) {
      const uint32_t g = x * 64u + (uint32_t)__builtin_ctzll(m);
      A = ys_combine(A, YsAgg{g_tid[g], g_tid[g], g_e1[g], 1u});
    }
    P.put(x, c, A);
  }
  __syncthreads();
  if (threadIdx.x < YS_NL) {
    const uint32_t c = threadIdx.x;
    uint32_t n = 0;
    for (uint32_t x = 0; x < YS_NT / 64; ++x) n += (uint32_t)__builtin_popcountll(wb[x][c]);
    const YsAgg A = ys_chain_cells(P, c, YsAgg{0u, 0u, INT32_MIN, 1u});
    table[(size_t)c * ntiles + blockIdx.x] = n;
    agg[(size_t)c * ntiles + blockIdx.x] = make_uint4(A.first_tid, A.last_tid, (uint32_t)A.mx, A.whole);
  }
}
// block l: row l of the aggregates -> exclusive prefix in place (the aggregate of the list's items in the tiles before)
__global__ __launch_bounds__(256) void yd_lagg_scan_k(uint4* __restrict__ agg, uint32_t ntiles) {
  __shared__ uint4 part[256];
  uint4* row = agg + (size_t)blockIdx.x * ntiles;
  const uint32_t per = (ntiles + 255u) / 256u, i0 = threadIdx.x * per, i1 = i0 + per < ntiles ? i0 + per : ntiles;
  auto un = [](const uint4& v) { return YsAgg{v.x, v.y, (int32_t)v.z, v.w}; };
  auto pk = [](const YsAgg& a) { return make_uint4(a.first_tid, a.last_tid, (uint32_t)a.mx, a.whole); };
  YsAgg A{0u, 0u, INT32_MIN, 1u};
  for (uint32_t i = i0; i < i1; ++i) A = ys_combine(A, un(row[i]));
  part[threadIdx.x] = pk(A);
  __syncthreads();
  if (threadIdx.x == 0) {  // 256 partial aggregates: a serial exclusive scan
    YsAgg run{0u, 0u, INT32_MIN, 1u};
    for (uint32_t q = 0; q < 256; ++q) {
      const YsAgg mine = un(part[q]);
      part[q] = pk(run);
      run = ys_combine(run, mine);
    }
  }
  __syncthreads();
  YsAgg run = un(part[threadIdx.x]);
  for (uint32_t i = i0; i < i1; ++i) {
    const YsAgg mine = un(row[i]);
    row[i] = pk(run);
    run = ys_combine(run, mine);
  }
}
// block l: row l of the table -> exclusive prefix in place, row total -> totals[l]
__global__ __launch_bounds__(1024) void yd_lscan_k(uint32_t* __restrict__ table, uint32_t ntiles, uint64_t* __restrict__ totals,
                                                   unsigned long long* __restrict__ nit /* zeroed: the items of all lists */) {
  __shared__ uint32_t sm[16];
  __shared__ uint32_t carry_s;
  uint32_t* row = table + (size_t)blockIdx.x * ntiles;
  constexpr uint32_t E = 8;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < ntiles; base += 1024 * E) {
    const uint32_t i0 = base + threadIdx.x * E;
    uint32_t v[E], s = 0;
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      v[e] = i0 + e < ntiles ? row[i0 + e] : 0u;
      s += v[e];
    }
    uint32_t tot;
    uint32_t ex = carry_s + block_excl_sum<uint32_t, 1024>(s, sm, &tot);
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      if (i0 + e < ntiles) row[i0 + e] = ex;
      ex += v[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    totals[blockIdx.x] = carry_s;
    atomicAdd(nit, (unsigned long long)carry_s);  // (128 blocks: the sum that was a launch of its own)
  }
}

// The item -> group word: the low half of the 64-bit item words of the radix split, or (items placed by list) an array of its own.
struct YdWords {
  const uint64_t* w64;
  const uint32_t* w32;
  __device__ __forceinline__ uint32_t group(uint32_t t) const { return w32 ? w32[t] : (uint32_t)w64[t]; }
};
struct YdItems {
  uint4* pk;      // (tid + 1, start, end, offset of the item's exon list in the per-group exon arrays): written with one store
  uint32_t* nex;  // exon count (contiguous: the input of the node-offset scan); items placed by list: bit 31 = the item opens a chain
  // Items placed by list (<= 64 inputs) are lean: (start, end) only — 16 bytes an item with the exon count and the group word
  // instead of 24.  The reference id is not needed behind the placement (the heads are flagged there), and the exon offset, which
  // only the spliced items use (one in twelve on config 3), is read from the item's group.
  uint2* se = nullptr;
  const uint32_t* gxoff = nullptr;  // per group: offset of its exon list (YdGroups::xoff)
  __device__ __forceinline__ uint4 rec(uint32_t t) const {
    if (se) {
      const uint2 a = se[t];
      return make_uint4(0u, a.x, a.y, 0u);
    }
    return pk[t];
  }
  __device__ __forceinline__ uint32_t xo_of(const uint4& it, uint32_t group) const { return se ? gxoff[group] : it.w; }
  __device__ __forceinline__ uint32_t tidp1(uint32_t t) const { return reinterpret_cast<const uint32_t*>(pk)[4 * (size_t)t]; }
  __device__ __forceinline__ int32_t start(uint32_t t) const { return (int32_t) reinterpret_cast<const uint32_t*>(pk)[4 * (size_t)t + 1]; }
  __device__ __forceinline__ int32_t end(uint32_t t) const { return (int32_t) reinterpret_cast<const uint32_t*>(pk)[4 * (size_t)t + 2]; }
  __device__ __forceinline__ uint32_t xo(uint32_t t) const { return reinterpret_cast<const uint32_t*>(pk)[4 * (size_t)t + 3]; }
};

// per output group (computed once, every item of the group reuses it): coordinates, exon count, exon list
struct YdGroups {
  uint4* pk;       // (tid + 1, start, end, exon word): one 16-byte gather per item instead of four
  uint32_t* nex;   // the exon counts, contiguous, as the input of the offset scan
  uint32_t* xoff;
};
// The exon word of a group / an item (YdGroups::pk.w, YdItems::nex): the exon count — or, with bit 30 set, the two exons themselves:
// a group whose key word is the exact code of the shape M N M carries the first block a : 10 and the gap g : 20 (strategy.hpp), so
// its exons are (start, start + a - 1), (start + a + g, end) and the chain kernels fetch nothing for such an item — neither its group
// word nor the three sectors of the exon arrays (one item in twelve on config 3, nearly all of them of this shape; 2.6 GB of gathers
// per launch of yd_wave_k).  Bit 31 of an item's word is the head flag of the items placed by list.
constexpr uint32_t YD_X2 = 1u << 30;
__device__ __forceinline__ uint32_t yd_nex_count(uint32_t w) { return (w & YD_X2) ? 2u : (w & 0x3FFFFFFFu); }
__device__ __forceinline__ bool yd_nex_x2(uint32_t w) { return (w & YD_X2) != 0u; }
__device__ __forceinline__ uint32_t yd_x2_a(uint32_t w) { return (w >> 20) & 0x3FFu; }
__device__ __forceinline__ uint32_t yd_x2_g(uint32_t w) { return w & 0xFFFFFu; }
__device__ __forceinline__ bool yd_nex_far(uint32_t w) { return !(w & YD_X2) && (w & 0x3FFFFFFFu) > 1u; }  // its exons lie in the groups' arrays

// The exons of a group whose key word is an exact code (strategy.hpp: record_key) follow from the key alone — one
// reference-consuming operation: one exon (start, end); M N M / two exons with first block a and gap g: (start, start + a - 1),
// (start + a + g, end) — so the representative's CIGAR, a random access per group, is only walked for the other groups (a few
// per cent of an RNA-seq sample).  Returns the exon count, 0 when the CIGAR has to be walked.
__device__ __forceinline__ uint32_t yd_exons_from_key(uint64_t lo, uint32_t st, uint32_t en, uint32_t* e0, uint32_t* s1) {
  const uint32_t h32 = (uint32_t)lo;
  *e0 = en;
  *s1 = 0;
  if ((h32 >> 30) == 2u) {  // one operation / one exon
    const uint32_t op = h32 & 0xFu;
    return (op == C_M || op == C_D || op == 7u || op == 8u || op == 0xFu) ? 1u : 0u;  // (M D = X, or -E's one-exon code; an N alone is two exons)
  }
  if ((h32 >> 30) == 3u) {
    const uint32_t a = (h32 >> 20) & 0x3FFu, g = h32 & 0xFFFFFu;
    *e0 = st + a - 1u;
    *s1 = st + a + g;
    return 2u;
  }
  return 0u;
}

__global__ void yd_groups_k(ColIn I, uint32_t ng, const uint32_t* __restrict__ gperm, GroupAcc G, const uint64_t* __restrict__ shi,
                            const uint64_t* __restrict__ slo, YdGroups Q) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  uint32_t sg = gperm[o];
  uint32_t q = G.first[sg];
  uint64_t h = shi[q], l = slo[q];
  int32_t st = (int32_t)(uint32_t)((h >> 2) & 0x7FFFFFFFull);
  const uint32_t tidp1 = (uint32_t)(h >> 33);
  const uint32_t en = (uint32_t)(st + (int32_t)(uint32_t)(l >> 32) - 1);
  uint32_t e0, s1;
  const uint32_t nk = yd_exons_from_key(l, (uint32_t)st, en, &e0, &s1);
  int nex = (int)nk;
  if (nex == 0) {
    const uint32_t r = (uint32_t)(G.rep[sg] & 0xFFFFFFFFull);
    walk_exons(I.pos[r], I.cig + I.cig_off[r], I.cig_off[r + 1] - I.cig_off[r], [](int, int) {}, &nex);
  }
  uint32_t xw = (uint32_t)nex;
  if (nk == 2u) {  // (from the key: the shape's two lengths, both inside their fields by the code's definition)
    const uint32_t a = e0 - (uint32_t)st + 1u, g = s1 - e0 - 1u;
    if (a < (1u << 10) && g < (1u << 20)) xw = YD_X2 | (a << 20) | g;
  }
  // the exon arrays hold the groups whose exons no item word can say: several exons, not the exact two-exon shape (a single exon is
  // (start, end); yd_gexons_k walks only those — a few per cent of an RNA-seq sample — and the offset scan counts only them)
  Q.nex[o] = yd_nex_far(xw) ? (uint32_t)nex : 0u;
  Q.pk[o] = make_uint4(tidp1, (uint32_t)st, en, xw);
}

__global__ void yd_gexons_k(ColIn I, uint32_t ng, const uint32_t* __restrict__ gperm, GroupAcc G, const uint64_t* __restrict__ slo, YdGroups Q,
                            uint32_t* __restrict__ ex_s, uint32_t* __restrict__ ex_e) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  const uint4 pk = Q.pk[o];
  if (!yd_nex_far(pk.w)) return;  // (nothing reads the arrays for it: yd_groups_k counted no slot)
  const uint32_t sg = gperm[o];
  uint32_t w = Q.xoff[o];
  {
    uint32_t e0, s1;
    const uint32_t nk = yd_exons_from_key(slo[G.first[sg]], pk.y, pk.z, &e0, &s1);
    if (nk) {
      ex_s[w] = pk.y;
      ex_e[w] = e0;
      if (nk == 2u) {
        ex_s[w + 1] = s1;
        ex_e[w + 1] = pk.z;
      }
      return;
    }
  }
  uint32_t r = (uint32_t)(G.rep[sg] & 0xFFFFFFFFull);
  int nex = 0;
  walk_exons(I.pos[r], I.cig + I.cig_off[r], I.cig_off[r + 1] - I.cig_off[r],
             [&](int es, int ee) {
               ex_s[w] = (uint32_t)es;
               ex_e[w] = (uint32_t)ee;
               ++w;
             },
             &nex);
}

// the coordinates of an item's group, written where the list sort drops the item (the sort's last scatter pass walks the items in
// group order, so the group table is read nearly sequentially; a pass over the sorted items would touch a line per item)
struct YdEmit {
  YdGroups Q;
  YdItems Y;
  __device__ __forceinline__ void operator()(uint32_t t, uint64_t w) const {
    const uint32_t o = (uint32_t)w;
    const uint4 g = Q.pk[o];
    Y.pk[t] = make_uint4(g.x, g.y, g.z, Q.xoff[o]);
    Y.nex[t] = g.w;
  }
};

// the second pass: every item dropped at its place in its list (the same record YdEmit leaves behind the radix split).  The items of
// the tile are walked list by list — thread k takes the k-th, k + 256-th, ... item of that order, finds its list by a bisection in
// the tile's per-list prefix and its group by selecting the r-th set bit of the list's 256-bit column — so that consecutive
// threads write consecutive positions of one list.
__device__ __forceinline__ uint32_t ys_select(uint64_t v, uint32_t r) {  // position of the r-th (0-based) set bit of v
  // (the half first, then five 32-bit steps of bit-field extract + count: the placement loop is bound by its vector instructions)
  const uint32_t lo = (uint32_t)v, cl = (uint32_t)__builtin_popcount(lo);
  const bool up = r >= cl;
  const uint32_t w = up ? (uint32_t)(v >> 32) : lo;
  r = up ? r - cl : r;
  uint32_t pos = 0;
#pragma unroll
  for (uint32_t st = 16; st >= 1; st >>= 1) {
    const uint32_t c = (uint32_t)__builtin_popcount((w >> pos) & ((1u << st) - 1u));
    const bool go = r >= c;
    r = go ? r - c : r;
    pos = go ? pos + st : pos;
  }
  return pos + (up ? 32u : 0u);
}
__global__ __launch_bounds__(YS_NT) void yd_lscatter_k(YsIn S, uint32_t ntiles, const uint32_t* __restrict__ table, const uint64_t* __restrict__ totals,
                                                       const uint4* __restrict__ agg, const uint64_t* __restrict__ gfiles, YdGroups Q, YdItems Y,
                                                       uint32_t* __restrict__ item) {
  __shared__ uint32_t base[YS_NL];            // first position of this tile's items of list c
  __shared__ uint32_t pre[YS_NL + 1];         // the tile's items in lists before c
  __shared__ uint64_t wb[YS_NT / 64][YS_NL];  // column c of wave w's bit matrix
  __shared__ uint64_t hb[YS_NT / 64][YS_NL];  // ... and which of those items open a chain
  __shared__ uint32_t ltot[YS_NL];            // items of list c in the whole call
  __shared__ uint32_t lsum[YS_NL];            // ... and in this tile
  __shared__ uint4 grec[YS_NT];
  __shared__ uint32_t gnex[YS_NT];
  __shared__ uint32_t pa[YS_NT / 64][YS_NL], pb[YS_NT / 64][YS_NL];
  __shared__ int32_t pm[YS_NT / 64][YS_NL];
  const YsCells P{pa, pb, pm};
  const uint32_t t = threadIdx.x, w = t >> 6;
  if (t < YS_NL) ltot[t] = (uint32_t)totals[t];
  const uint32_t o = blockIdx.x * YS_NT + t;
  uint64_t lo = 0, hi = 0;
  if (o < S.ng) ys_mask_from_files(gfiles[2 * (size_t)o], (uint32_t)gfiles[2 * (size_t)o + 1], &lo, &hi);
  wb[w][lane_id()] = wave_bit_transpose(lo);
  wb[w][64u + lane_id()] = wave_bit_transpose(hi);
  if (o < S.ng) {
    const uint4 g = Q.pk[o];
    grec[t] = make_uint4(g.x, g.y, g.z, Q.xoff[o]);
    gnex[t] = g.w;
  }
  __syncthreads();
  for (uint32_t q = t; q < (YS_NT / 64) * YS_NL; q += YS_NT) {  // every cell folds its items
    const uint32_t x = q / YS_NL, c = q % YS_NL;
    YsAgg A{0u, 0u, INT32_MIN, 1u};
    for (uint64_t m = wb[x][c]; m; m &= m - 1) {
      const uint4 g = grec[x * 64u + (uint32_t)__builtin_ctzll(m)];
      A = ys_combine(A, YsAgg{g.x, g.x, (int32_t)g.z + 1, 1u});
    }
    P.put(x, c, A);
  }
  __syncthreads();
  if (t < YS_NL) {
    uint32_t b = 0;  // list base: the totals of the lists before it (128 values: a serial sum per thread is cheap enough)
    for (uint32_t c = 0; c < t; ++c) b += ltot[c];
    base[t] = b + table[(size_t)t * ntiles + blockIdx.x];
    uint32_t n = 0;
    for (uint32_t x = 0; x < YS_NT / 64; ++x) n += (uint32_t)__builtin_popcountll(wb[x][t]);
    lsum[t] = n;
    // every cell of the list learns the aggregate of the list's items before it, those of the tiles before included
    const uint4 av = agg[(size_t)t * ntiles + blockIdx.x];
    (void)ys_chain_cells(P, t, YsAgg{av.x, av.y, (int32_t)av.z, av.w});
  }
  __syncthreads();
  for (uint32_t q = t; q < (YS_NT / 64) * YS_NL; q += YS_NT) {  // ... and flags its heads
    const uint32_t x = q / YS_NL, c = q % YS_NL;
    YsAgg A = P.get(x, c);
    uint64_t heads = 0;
    for (uint64_t m = wb[x][c]; m; m &= m - 1) {
      const uint32_t bit = (uint32_t)__builtin_ctzll(m);
      const uint4 g = grec[x * 64u + bit];
      if (A.last_tid == 0u || g.x != A.last_tid || (int32_t)g.y > A.mx) heads |= 1ull << bit;
      A = ys_combine(A, YsAgg{g.x, g.x, (int32_t)g.z + 1, 1u});
    }
    hb[x][c] = heads;
  }
  __syncthreads();
  if (t < 64) {  // exclusive prefix of the 128 per-list counts: one wave, two lists per lane
    const uint32_t a = lsum[2 * t], b = lsum[2 * t + 1];
    const uint32_t inc = wave_incl_sum(a + b);
    pre[2 * t] = inc - a - b;
    pre[2 * t + 1] = inc - b;
    if (t == 63) pre[YS_NL] = inc;
  } else if (t < 64 + YS_NL) {  // ... and of every list's items wave by wave (the cells' aggregates are dead: pa holds it)
    const uint32_t c = t - 64;
    uint32_t run = 0;
    for (uint32_t x = 0; x < YS_NT / 64; ++x) {
      pa[x][c] = run;
      run += (uint32_t)__builtin_popcountll(wb[x][c]);
    }
  }
  __syncthreads();
  const uint32_t T = pre[YS_NL];
  for (uint32_t idx = t; idx < T; idx += YS_NT) {
    uint32_t c = 0;  // last list with pre[c] <= idx
#pragma unroll
    for (uint32_t st = 64; st >= 1; st >>= 1) c = pre[c + st] <= idx ? c + st : c;
    uint32_t r = idx - pre[c];
    const uint32_t rank = r;
    // the wave whose column holds the r-th group: the last one with at most r of the list's items before it (it holds an item:
    // r < the list's count).  A walk over the columns ran as long as the slowest lane of the wave: sixteen rounds, not eight.
    uint32_t x = 0;
#pragma unroll
    for (uint32_t st = YS_NT / 128; st >= 1; st >>= 1) x = pa[x + st][c] <= r ? x + st : x;
    r -= pa[x][c];
    const uint32_t bit = ys_select(wb[x][c], r);
    const uint32_t g = x * 64u + bit;
    const uint32_t pos = base[c] + rank;
    if (Y.se)
      Y.se[pos] = make_uint2(grec[g].y, grec[g].z);
    else
      Y.pk[pos] = grec[g];
    Y.nex[pos] = gnex[g] | ((uint32_t)((hb[x][c] >> bit) & 1ull) << 31);  // (bit 31: the item opens a chain — read by yd_number)
    item[pos] = blockIdx.x * YS_NT + g;
  }
}

struct SegMaxY {
  int32_t mx;
  uint32_t flag;
};
struct SegMaxYOp {
  __device__ __forceinline__ SegMaxY operator()(const SegMaxY& a, const SegMaxY& b) const {
    SegMaxY r;
    r.mx = b.flag ? b.mx : (a.mx > b.mx ? a.mx : b.mx);
    r.flag = a.flag | b.flag;
    return r;
  }
};
struct YdLoad {
  const uint64_t* list;  // item words: list id in the high half
  YdItems Y;
  __device__ __forceinline__ bool list_head(uint32_t t) const {  // new list, or rspacing.reset() (:586-589)
    return t == 0 || (list[t] >> 32) != (list[t - 1] >> 32) || Y.tidp1(t) != Y.tidp1(t - 1);
  }
  __device__ __forceinline__ SegMaxY operator()(uint32_t t) const {
    SegMaxY s;
    // end + 1: a CIGAR that ends in an intron leaves a last exon (end + 1, end) in the list (GSam.cpp:351-417) — a node that
    // a read starting at end + 1 does not clear; only a start beyond end + 1 renews the list
    s.mx = Y.end(t) + 1;
    s.flag = list_head(t) ? 1u : 0u;
    return s;
  }
};
// chain heads and exon counts summed together; a head's exclusive sums are its chain's number and the first node of its arena
struct HeadNex {
  uint32_t h, n;
};
struct HeadNexOp {
  __device__ __forceinline__ HeadNex operator()(const HeadNex& a, const HeadNex& b) const { return HeadNex{a.h + b.h, a.n + b.n}; }
};
struct YdAux {  // (start, exon count) of the item
  YdItems Y;
  __device__ __forceinline__ int2 operator()(uint32_t t) const { return make_int2(Y.start(t), (int)yd_nex_count(Y.nex[t])); }
};
struct YdHead {
  __device__ __forceinline__ HeadNex operator()(uint32_t, const SegMaxY& v, const SegMaxY&, const SegMaxY& ex, const int2& a) const {
    // renewal: the read starts beyond every earlier end of this list => every node is cleared (processRead :230-241)
    return HeadNex{(v.flag || a.x > ex.mx) ? 1u : 0u, (uint32_t)a.y};
  }
};
// chains numbered in item order, each with its first item and the first node of its arena
struct YdStore {
  YdItems Y;
  uint32_t *chain_first, *chain_noff;
  uint64_t* totals;  // [0] chains, [1] nodes
  uint32_t nit;
  __device__ __forceinline__ void operator()(uint32_t t, const SegMaxY&, const SegMaxY&, const SegMaxY&, const HeadNex& v, const HeadNex& ex,
                                             const int2&) const {
    if (v.h) {
      chain_first[ex.h] = t;
      chain_noff[ex.h] = ex.n;
    }
    if (t + 1 == nit) {
      totals[0] = ex.h + v.h;
      totals[1] = ex.n + v.n;
    }
  }
};

// Items placed by list carry their head flag in bit 31 of the exon-count word: numbering the chains (and summing the exon counts
// before each head, its node-arena base) is a scan of that one array — 4 bytes per item, four items per lane and load, thread order =
// item order: tile sums, a scan of the 43 k tile sums, and a second pass that writes at the heads (3 % of the items).
constexpr uint32_t YN_NT = 256, YN_ROWS = 4, YN_TILE = YN_NT * 4 * YN_ROWS;  // 4096 items per block
__device__ __forceinline__ uint4 yn_load(const uint32_t* __restrict__ w, uint64_t i, uint32_t nit) {  // items i .. i + 3 (0 beyond the end)
  if (i + 3 < nit) return *reinterpret_cast<const uint4*>(w + i);  // (i is a multiple of 4, the array 256-byte aligned)
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (i < nit) v.x = w[i];
  if (i + 1 < nit) v.y = w[i + 1];
  if (i + 2 < nit) v.z = w[i + 2];
  return v;
}
__global__ __launch_bounds__(YN_NT) void yn_reduce_k(const uint32_t* __restrict__ w, uint32_t nit, unsigned long long* __restrict__ part) {
  __shared__ unsigned long long sm[YN_NT / 64];
  unsigned long long acc = 0;  // heads : 32 | exon counts : 32 (nodes < 2^31: checked by the caller)
#pragma unroll
  for (uint32_t r = 0; r < YN_ROWS; ++r) {
    const uint4 v = yn_load(w, (uint64_t)blockIdx.x * YN_TILE + ((uint64_t)r * YN_NT + threadIdx.x) * 4u, nit);
    const uint32_t h = (v.x >> 31) + (v.y >> 31) + (v.z >> 31) + (v.w >> 31);
    const uint32_t n = yd_nex_count(v.x) + yd_nex_count(v.y) + yd_nex_count(v.z) + yd_nex_count(v.w);
    acc += ((unsigned long long)h << 32) + n;
  }
  acc = wave_sum(acc);
  if (lane_id() == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
// single block: exclusive prefix of the tile sums in place; the grand totals -> totals[0] (chains), totals[1] (nodes)
__global__ __launch_bounds__(1024) void yn_spine_k(unsigned long long* __restrict__ part, uint32_t nb, uint64_t* __restrict__ totals) {
  __shared__ unsigned long long sm[16];
  __shared__ unsigned long long carry_s, nodes_s;
  constexpr uint32_t E = 16;
  if (threadIdx.x == 0) carry_s = nodes_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 1024 * E) {
    const uint32_t i0 = base + threadIdx.x * E;
    unsigned long long v[E], s2 = 0, n2 = 0;
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      v[e] = i0 + e < nb ? part[i0 + e] : 0ull;
      s2 += v[e];
      n2 += v[e] & 0xFFFFFFFFull;  // (a tile's exon counts are far below 2^32; their total need not be: counted on its own)
    }
    n2 = wave_sum(n2);
    if (lane_id() == 0 && n2) atomicAdd(&nodes_s, n2);
    unsigned long long tot;
    unsigned long long ex = carry_s + block_excl_sum<unsigned long long, 1024>(s2, sm, &tot);
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      if (i0 + e < nb) part[i0 + e] = ex;
      ex += v[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    totals[0] = carry_s >> 32;  // (meaningless when the nodes overflow 32 bits: the caller refuses the tile on totals[1])
    totals[1] = nodes_s;
  }
}
__global__ __launch_bounds__(YN_NT) void yn_emit_k(const uint32_t* __restrict__ w, uint32_t nit, const unsigned long long* __restrict__ part,
                                                   uint32_t* __restrict__ chain_first, uint32_t* __restrict__ chain_noff) {
  __shared__ unsigned long long sm[YN_NT / 64 + 4];
  unsigned long long carry = part[blockIdx.x];
#pragma unroll
  for (uint32_t r = 0; r < YN_ROWS; ++r) {
    const uint64_t i = (uint64_t)blockIdx.x * YN_TILE + ((uint64_t)r * YN_NT + threadIdx.x) * 4u;
    const uint4 v = yn_load(w, i, nit);
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
    unsigned long long mine = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) mine += ((unsigned long long)(wv[e] >> 31) << 32) + yd_nex_count(wv[e]);
    unsigned long long tot;
    unsigned long long ex = carry + block_excl_sum<unsigned long long, YN_NT>(mine, sm, &tot);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if ((wv[e] >> 31) && (ex >> 32) < nit) {  // chains numbered in item order, each with its first item and the first node of its arena
        chain_first[(uint32_t)(ex >> 32)] = (uint32_t)(i + (uint64_t)e);  // (the bound only matters when the node count overflows)
        chain_noff[(uint32_t)(ex >> 32)] = (uint32_t)ex;
      }
      ex += ((unsigned long long)(wv[e] >> 31) << 32) + yd_nex_count(wv[e]);
    }
    carry += tot;
  }
}

constexpr uint32_t YD_LONG = 24;  // chains at least this long get a whole wave
constexpr int YD_FAST_NODES = 12;   // the spliced-read fast path of yd_wave_k searches at most this many nodes per lane

// ids[0..] = short chains (thread each), ids2 = long chains (wave each); counts in cnt[0], cnt[1]
__global__ void yd_classify_k(uint32_t nchains, uint32_t nit, const uint32_t* __restrict__ chain_first, uint32_t* __restrict__ ids_short,
                              uint32_t* __restrict__ ids_long, uint32_t* __restrict__ cnt) {
  // block-aggregated append: two global atomics per 256 chains
  __shared__ uint32_t s_cnt[2], s_base[2];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  bool act = c < nchains;
  bool is_long = false;
  uint32_t slot = 0;
  if (act) {
    uint32_t t0 = chain_first[c];
    uint32_t t1 = (c + 1 < nchains) ? chain_first[c + 1] : nit;
    is_long = (t1 - t0) >= YD_LONG;
    slot = atomicAdd(&s_cnt[is_long ? 1 : 0], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 2) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&cnt[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
  __syncthreads();
  if (act) {
    if (is_long)
      ids_long[s_base[1] + slot] = c;
    else
      ids_short[s_base[0] + slot] = c;
  }
}

// GSegList (tiebrush.cpp:111-250) with node indices into a per-chain arena; literal, including the
// mergeRead tail drop.  One thread runs one chain.
struct SegNodes {
  uint32_t* s;
  uint32_t* e;
  int32_t* nx;
};

__device__ void yd_merge_read(const uint32_t* __restrict__ xs, const uint32_t* __restrict__ xe, uint32_t nex, SegNodes N, int32_t& head,
                              uint32_t& alloc) {
  if (head < 0) {  // :168-177
    int32_t cn = -1;
    for (uint32_t k = 0; k < nex; ++k) {
      uint32_t nw = alloc++;
      N.s[nw] = xs[k];
      N.e[nw] = xe[k];
      N.nx[nw] = -1;
      if (cn < 0)
        head = (int32_t)nw;
      else
        N.nx[cn] = (int32_t)nw;
      cn = (int32_t)nw;
    }
    return;
  }
  int32_t cur = head, prev = -1;
  for (uint32_t k = 0; k < nex; ++k) {
    uint32_t es = xs[k], ee = xe[k];
    while (cur >= 0) {
      if (ee < N.s[cur]) {  // insert before cur :182-191
        uint32_t nw = alloc++;
        N.s[nw] = es;
        N.e[nw] = ee;
        N.nx[nw] = cur;
        if (cur == head)
          head = (int32_t)nw;
        else
          N.nx[prev] = (int32_t)nw;
        prev = (int32_t)nw;
        break;
      }
      if (es <= N.e[cur]) {  // overlap :194-212
        if (es < N.s[cur]) N.s[cur] = es;
        if (ee > N.e[cur]) N.e[cur] = ee;
        int32_t nx = N.nx[cur];
        while (nx >= 0 && N.s[nx] <= N.e[cur]) {
          uint32_t nend = N.e[nx];
          N.nx[cur] = N.nx[nx];
          nx = N.nx[cur];
          if (nend > N.e[cur]) {
            N.e[cur] = nend;
            break;
          }
        }
        break;
      }
      prev = cur;  // :214-216
      cur = N.nx[cur];
    }
    if (cur < 0) break;  // this exon and all later ones are dropped (reference behaviour)
  }
}

// thread per chain (short chains, and long chains whose list outgrew the 64 lanes of yd_wave_k)
__global__ void yd_run_k(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ nids, uint32_t nchains, uint32_t nit,
                         const uint32_t* __restrict__ chain_first, YdItems Y, YdWords v,
                         const uint32_t* __restrict__ noff, const uint32_t* __restrict__ ex_s, const uint32_t* __restrict__ ex_e,
                         SegNodes N, int32_t* __restrict__ g_yd, uint32_t* __restrict__ yd_d) {
  uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= *nids) return;
  uint32_t c = ids[x];
  uint32_t t0 = chain_first[c];
  uint32_t t1 = (c + 1 < nchains) ? chain_first[c + 1] : nit;
  int32_t head = -1;
  uint32_t last_pos = 0;
  int last_dist = -1;
  uint32_t alloc = noff[c];  // (per chain)
  for (uint32_t t = t0; t < t1; ++t) {
    const uint4 it = Y.rec(t);
    uint32_t rstart = it.y;
    const uint32_t xw = Y.nex[t], nex = yd_nex_count(xw);
    // the item's exons: in the groups' arrays, or (one exon; the exact two-exon shape) said by the item itself
    uint32_t ls[2] = {it.y, 0u}, le[2] = {it.z, it.z};
    if (yd_nex_x2(xw)) {
      le[0] = it.y + yd_x2_a(xw) - 1u;
      ls[1] = le[0] + yd_x2_g(xw) + 1u;
    }
    const bool far = yd_nex_far(xw);
    const uint32_t xo = far ? Y.xo_of(it, v.group(t)) : 0u;
    const uint32_t* const pxs = far ? ex_s + xo : ls;
    const uint32_t* const pxe = far ? ex_e + xo : le;
    int d;
    if (last_pos == rstart) {  // :225-228
      yd_merge_read(pxs, pxe, nex, N, head, alloc);
      d = last_dist;
    } else {
      d = 0;
      int32_t node = head, prev = -1;
      while (node >= 0 && N.s[node] < rstart) {
        prev = node;
        node = N.nx[node];
      }
      if (prev >= 0) {
        if (N.e[prev] >= rstart) d = (int)(rstart - N.s[prev]);
        if (d == 0) head = N.nx[prev];  // clearTo(prev)
      }
      last_pos = rstart;
      last_dist = d;
      yd_merge_read(pxs, pxe, nex, N, head, alloc);
    }
    if (yd_d)
      yd_d[t] = d > 0 ? (uint32_t)d : 0u;
    else if (d > 0)
      atomicMax(&g_yd[v.group(t)], d);
  }
}

// ---- thread per chain, the list in registers -------------------------------------------------------------------------------------
// Short chains (the bulk of the chains, a minority of the items) run one per lane.  What made the first thread-per-chain kernel
// (yd_run_k) slow was the list in global memory — a pointer chase per node — and every lane streaming its own items (64 cache
// lines per load instruction).  Here the sorted node list of a lane is YL_CAP (start, end) pairs in REGISTERS (every index is a
// compile-time constant after unrolling; a dynamic position is a chain of selects: an LDS-resident list was measured 4x slower,
// its shifts are chains of dependent LDS round trips), and the items reach the lanes through LDS: R consecutive lanes load R
// consecutive items of one chain, so a load instruction brings in whole segments, and every lane then reads its own R staged
// items.  Lanes of a wave take chains of similar length (bucketed by log2 of the length, longest first).  A list that outgrows
// YL_CAP hands the chain to yd_run_k (ids_over: the outputs are maxima, a rerun from the chain's start is harmless).  Chains of
// YD_WAVE_MIN_DEFAULT items and more keep a wave to themselves (yd_wave_k): a lane steps through an item in thousands of cycles
// of shared issue time, so a long chain in a lane is a long tail — measured on config 3, lane + wave kernels: threshold 24:
// 1.8 + 6.1 ms, 128: 4.8 + 6.1, 512: 4.4 + 3.5, 2048: 8.4 + 2.6 (before: yd_run_k 2.3 + yd_wave_k 7.7).
constexpr int YL_CAP = 8;
constexpr uint32_t YD_WAVE_MIN_DEFAULT = 24;

constexpr int YD_NB = 32;  // length buckets (log2)

__device__ __forceinline__ uint32_t yd_bucket(uint32_t len) { return 31u - (uint32_t)__builtin_clz(len | 1u); }

// cnt[b] = chains of length bucket b that go to the lane kernel, cnt[YD_NB + b] = those that get a wave
// (both bucket kernels walk the chains with a fixed grid and touch the global counters once per block and bucket: returning
// atomics on one word take ~ 12 ns each, and one per 256 chains and bucket was most of the 0.3 ms these kernels took on config 3)
constexpr uint32_t YD_BGRID = 1024;
__global__ void yd_bucket_count_k(uint32_t nchains, uint32_t nit, uint32_t wave_min, const uint32_t* __restrict__ chain_first,
                                  uint32_t* __restrict__ cnt) {
  __shared__ uint32_t s_cnt[2 * YD_NB];
  if (threadIdx.x < 2 * YD_NB) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nchains; c += gridDim.x * blockDim.x) {
    const uint32_t len = ((c + 1 < nchains) ? chain_first[c + 1] : nit) - chain_first[c];
    atomicAdd(&s_cnt[(len >= wave_min ? YD_NB : 0) + yd_bucket(len)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 2 * YD_NB && s_cnt[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], s_cnt[threadIdx.x]);
}
// cursors: the buckets of either kind laid out longest first (a wave that starts a long chain late is the kernel's tail);
// cur[2 * YD_NB] = the number of chains that get a wave
__global__ void yd_bucket_off_k(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ cur) {
  for (int kind = 0; kind < 2; ++kind) {
    uint32_t o = 0;
    for (int b = YD_NB - 1; b >= 0; --b) {
      cur[kind * YD_NB + b] = o;
      o += cnt[kind * YD_NB + b];
    }
    if (kind == 1) cur[2 * YD_NB] = o;
  }
}
__global__ void yd_bucket_fill_k(uint32_t nchains, uint32_t nit, uint32_t wave_min, const uint32_t* __restrict__ chain_first, uint32_t* __restrict__ cur,
                                 uint32_t* __restrict__ ids_lane, uint32_t* __restrict__ ids_wave) {
  __shared__ uint32_t s_cnt[2 * YD_NB], s_next[2 * YD_NB];
  if (threadIdx.x < 2 * YD_NB) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  auto bucket_of = [&](uint32_t c) {
    const uint32_t len = ((c + 1 < nchains) ? chain_first[c + 1] : nit) - chain_first[c];
    return (len >= wave_min ? (uint32_t)YD_NB : 0u) + yd_bucket(len);
  };
  // the block's chains per bucket, one reservation per bucket, then the same walk hands out the places
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nchains; c += gridDim.x * blockDim.x) atomicAdd(&s_cnt[bucket_of(c)], 1u);
  __syncthreads();
  if (threadIdx.x < 2 * YD_NB) s_next[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&cur[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
  __syncthreads();
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nchains; c += gridDim.x * blockDim.x) {
    const uint32_t b = bucket_of(c);
    (b >= (uint32_t)YD_NB ? ids_wave : ids_lane)[atomicAdd(&s_next[b], 1u)] = c;
  }
}

template <int R /* items per lane and refill: the items of R consecutive lanes' worth are one coalesced segment */>
__global__ __launch_bounds__(64) void yd_lane_k(const uint32_t* __restrict__ ids, uint32_t nids, uint32_t nchains, uint32_t nit,
                                                const uint32_t* __restrict__ chain_first, YdItems Y, YdWords v,
                                                const uint32_t* __restrict__ ex_s, const uint32_t* __restrict__ ex_e,
                                                int32_t* __restrict__ g_yd, uint32_t* __restrict__ yd_d, uint32_t* __restrict__ ids_over,
                                                uint32_t* __restrict__ n_over) {
  constexpr int RS = R + 1;  // row stride of the staged items (odd: the lanes' own reads fall on different banks)
  __shared__ uint32_t S_start[64 * RS], S_xo[64 * RS], S_e0[64 * RS], S_s1[64 * RS], S_e1[64 * RS], S_nex[64 * RS], S_o[64 * RS];
  const uint32_t x = blockIdx.x * 64 + threadIdx.x;
  const uint32_t l = threadIdx.x;
  const bool mine = x < nids;
  const uint32_t c = mine ? ids[x] : 0u;
  uint32_t t = mine ? chain_first[c] : 0u;
  const uint32_t t1 = mine ? ((c + 1 < nchains) ? chain_first[c + 1] : nit) : 0u;
  // the lane's sorted node list: YL_CAP (start, end) pairs in registers — every index below is a compile-time constant after
  // unrolling, a dynamic position is a chain of selects (no scratch, no LDS round trip on the machine's critical path)
  uint32_t ns[YL_CAP], ne[YL_CAP];
#pragma unroll
  for (int j = 0; j < YL_CAP; ++j) ns[j] = ne[j] = 0;
  int cnt = 0;
  uint32_t last_pos = 0;
  int last_dist = -1;
  bool over = false;
#define YL_GET(dst, a, k)                                         \
  {                                                               \
    dst = a[0];                                                   \
    _Pragma("unroll") for (int j_ = 1; j_ < YL_CAP; ++j_) dst = (k) == j_ ? a[j_] : dst; \
  }
#define YL_PUT(a, k, val)                                         \
  {                                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < YL_CAP; ++j_) a[j_] = (k) == j_ ? (val) : a[j_]; \
  }
#define YL_REMOVE1(at) /* drop node `at` */                        \
  {                                                               \
    _Pragma("unroll") for (int j_ = 0; j_ + 1 < YL_CAP; ++j_) {   \
      ns[j_] = j_ >= (at) ? ns[j_ + 1] : ns[j_];                  \
      ne[j_] = j_ >= (at) ? ne[j_ + 1] : ne[j_];                  \
    }                                                             \
    --cnt;                                                        \
  }
  constexpr int LG = 64 / R;  // chains whose next R items one load instruction brings in
  const uint32_t g = l / R, i0 = l % R;
  while (__any(t < t1 && !over)) {
    // ---- refill: the next R items of every lane's chain, R consecutive lanes reading R consecutive items ----
    const uint32_t tl = (t < t1 && !over) ? t : 0xFFFFFFFFu;
#pragma unroll
    for (int p = 0; p < R; ++p) {
      const uint32_t q = (uint32_t)p * LG + g;
      const uint32_t tq = __shfl(tl, q, 64), t1q = __shfl(t1, q, 64);
      const uint32_t idx = tq + i0;
      if (tq != 0xFFFFFFFFu && idx < t1q) {
        const uint4 a = Y.rec(idx);
        const uint32_t xw = Y.nex[idx];
        const uint32_t nx = yd_nex_count(xw);
        const bool x2 = yd_nex_x2(xw), far = nx > 1 && !x2;  // far: the exons lie in the groups' arrays (x2: they ride in the word)
        const uint32_t o = (!yd_d || far) ? v.group(idx) : 0u;  // (yd_d: the distances go to the items' own slots — only such an item
                                                                // looks its group up, for the exon list)
        const uint32_t w = q * RS + i0;
        const uint32_t axo = far ? Y.xo_of(a, o) : 0u;
        const uint32_t x2e0 = a.y + yd_x2_a(xw) - 1u;
        S_start[w] = a.y;
        S_xo[w] = axo;
        S_e0[w] = far ? ex_e[axo] : (x2 ? x2e0 : a.z);  // (a single exon ends where the read ends; the exon arrays follow the groups:
                                                        // items next to each other in a chain read next to each other)
        S_s1[w] = far ? ex_s[axo + 1] : (x2 ? x2e0 + yd_x2_g(xw) + 1u : 0u);
        S_e1[w] = far ? ex_e[axo + 1] : (x2 ? a.z : 0u);
        S_nex[w] = nx;
        S_o[w] = o;
      }
    }
    __syncthreads();  // (one wave per block)
#pragma unroll 1
    for (int i = 0; i < R; ++i) {
      const bool live = t < t1 && !over;
      if (!__any(live)) break;
      const uint32_t w = l * RS + (uint32_t)i;
      const uint32_t rstart = S_start[w], xo = S_xo[w], nex = live ? S_nex[w] : 0u, e0 = S_e0[w], s1 = S_s1[w], e1 = S_e1[w], o = S_o[w];
      if (live) {
        int d;
        if (last_pos == rstart) {  // processRead :221-228
          d = last_dist;
        } else {
          d = 0;
          int p = 0;  // leading nodes that start before the read (the list is sorted)
#pragma unroll
          for (int j = 0; j < YL_CAP; ++j) p += (j < cnt && ns[j] < rstart) ? 1 : 0;
          if (p > 0) {
            uint32_t pS, pE;
            YL_GET(pS, ns, p - 1)
            YL_GET(pE, ne, p - 1)
            if (pE >= rstart) d = (int)(rstart - pS);
            if (d == 0)
              for (int r = 0; r < p; ++r) YL_REMOVE1(0)  // clearTo(prev)
          }
          last_pos = rstart;
          last_dist = d;
        }
        // mergeRead :167-219
        if (cnt == 0) {
          if (nex > (uint32_t)YL_CAP) {
            over = true;
          } else {
            ns[0] = rstart;
            ne[0] = e0;
            ns[1] = s1;
            ne[1] = e1;
            for (uint32_t k = 2; k < nex; ++k) {
              const uint32_t xs = ex_s[xo + k], xe = ex_e[xo + k];
              YL_PUT(ns, (int)k, xs)
              YL_PUT(ne, (int)k, xe)
            }
            cnt = (int)nex;
          }
        } else {
          int cur = 0;
          for (uint32_t k = 0; k < nex && !over; ++k) {
            const uint32_t es = k == 0 ? rstart : (k == 1 ? s1 : ex_s[xo + k]);
            const uint32_t ee = k == 0 ? e0 : (k == 1 ? e1 : ex_e[xo + k]);
            // the first node at or behind `cur` that the exon precedes or overlaps; none: this exon and the rest are dropped
            uint32_t stop = 0;
#pragma unroll
            for (int j = 0; j < YL_CAP; ++j) stop |= (j >= cur && j < cnt && (ee < ns[j] || es <= ne[j])) ? (1u << j) : 0u;
            if (!stop) break;
            const int n = __builtin_ctz(stop);
            uint32_t nS, nE;
            YL_GET(nS, ns, n)
            YL_GET(nE, ne, n)
            if (ee < nS) {  // insert before n :182-191
              if (cnt == YL_CAP) {
                over = true;
                break;
              }
#pragma unroll
              for (int j = YL_CAP - 1; j > 0; --j) {
                ns[j] = j > n ? ns[j - 1] : ns[j];
                ne[j] = j > n ? ne[j - 1] : ne[j];
              }
              YL_PUT(ns, n, es)
              YL_PUT(ne, n, ee)
              ++cnt;
              cur = n + 1;  // (the node the walk stands on moved up by one)
            } else {  // overlap :194-212
              const uint32_t newS = es < nS ? es : nS;
              uint32_t newE = ee > nE ? ee : nE;
              while (n + 1 < cnt) {  // swallow followers; stops behind the first one that extends the node
                uint32_t xS, xE;
                YL_GET(xS, ns, n + 1)
                if (xS > newE) break;
                YL_GET(xE, ne, n + 1)
                YL_REMOVE1(n + 1)
                if (xE > newE) {
                  newE = xE;
                  break;
                }
              }
              YL_PUT(ns, n, newS)
              YL_PUT(ne, n, newE)
              cur = n;
            }
          }
        }
        if (!over) {
          if (yd_d)
            yd_d[t] = d > 0 ? (uint32_t)d : 0u;
          else if (d > 0)
            atomicMax(&g_yd[o], d);
        }
        ++t;
      }
    }
    __syncthreads();  // (the staged items are consumed before the next refill overwrites them)
  }
  if (over) ids_over[atomicAdd(n_over, 1u)] = c;
#undef YL_GET
#undef YL_PUT
#undef YL_REMOVE1
}

// Wave-native GSegList: the sorted node list lives one node per lane (ns, ne in registers of lane i = node i),
// control flow is wave-uniform, list surgery is ballots + shuffles.  One 64-thread block per long chain.
// A list that would need more than 64 nodes hands the chain over to yd_run_k (ids_over).
// inclusive prefix max over the 64 lanes (DPP row shifts + row broadcasts, identity 0)
__device__ __forceinline__ uint32_t wave_prefix_max(uint32_t pm) {
  pm = max(pm, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x111, 0xf, 0xf, false));
  pm = max(pm, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x112, 0xf, 0xf, false));
  pm = max(pm, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x114, 0xf, 0xf, false));
  pm = max(pm, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x118, 0xf, 0xf, false));
  pm = max(pm, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x142, 0xa, 0xf, false));
  pm = max(pm, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x143, 0xc, 0xf, false));
  return pm;
}
__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

__global__ __launch_bounds__(64) void yd_wave_k(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ nids, uint32_t nchains,
                                                uint32_t nit, const uint32_t* __restrict__ chain_first, YdItems Y,
                                                YdWords v, const uint32_t* __restrict__ noff,
                                                const uint32_t* __restrict__ ex_s, const uint32_t* __restrict__ ex_e,
                                                int32_t* __restrict__ g_yd, uint32_t* __restrict__ yd_d, uint32_t* __restrict__ ids_over,
                                                uint32_t* __restrict__ n_over) {
  if (blockIdx.x >= *nids) return;
  const uint32_t c = ids[blockIdx.x];
  const uint32_t t0 = chain_first[c];
  const uint32_t t1 = (c + 1 < nchains) ? chain_first[c + 1] : nit;
  const int lane = (int)threadIdx.x;
  uint32_t ns = 0, ne = 0;  // node `lane`
  int cnt = 0;              // uniform
  uint32_t last_pos = 0;
  int last_dist = -1;
  bool overflow = false;
  // Scalar mirror of the ACTIVE island: node P = [SP, EP], the node behind it starts at DP (none: ~0).  P is the node the last item's
  // processRead found as `prev` — in a pile-up the island the reads are extending.  (Round 3 mirrored node 0: but once a read
  // starts beyond the end of node 0 without clearing it — it lies inside a later island, d > 0, and clearTo only runs when d == 0 —
  // node 0 stays in the list, dead, and every later item of the chain failed the fast test and went through the general path
  // one by one: 11.9 M of config 3's 13.0 M single items, two thirds of the kernel's instructions.)
  // What the mirror says is true of the list whenever mirror_ok: any surgery at or before node P + 1 clears the flag, and the
  // next item looks P up again.  Nodes before P are dead for every later item (sorted, disjoint: their ends lie below SP).
  uint32_t SP = 0, EP = 0, DP = 0xFFFFFFFFu;
  int P = -1;
  bool mirror_ok = false;
  // batch loader: lane l holds item tb+l.  The next batch is requested before the current one is processed so that
  // its global-load latency hides behind the (long, scalar-ish) item loop.
  struct Batch {
    uint32_t start, nex, xo, o, e0, s1, e1, s2, e2;
  };
  auto load_batch = [&](uint32_t tb) {
    Batch b;
    const uint32_t t = tb + (uint32_t)lane;
    const bool have = t < t1;
    const uint4 it = have ? Y.rec(t) : make_uint4(0u, 0u, 0u, 0u);
    b.start = it.y;
    const uint32_t xw = have ? Y.nex[t] : 0u;
    b.nex = yd_nex_count(xw);
    const bool x2 = yd_nex_x2(xw), far = b.nex > 1u && !x2;  // far: the exons lie in the groups' arrays (x2: they ride in the word)
    b.o = have && (!yd_d || far) ? v.group(t) : 0u;  // (yd_d: only such an item looks its group up, for the exon list)
    b.xo = far ? Y.xo_of(it, b.o) : 0u;
    const uint32_t x2e0 = it.y + yd_x2_a(xw) - 1u;
    b.e0 = far ? ex_e[b.xo] : (x2 ? x2e0 : it.z);  // first exon end (its start is the read start); a single exon ends where the
                                                   // read ends: no gather (three items in four, a 64-byte sector each)
    b.s1 = far ? ex_s[b.xo + 1] : (x2 ? x2e0 + yd_x2_g(xw) + 1u : 0u);
    b.e1 = far ? ex_e[b.xo + 1] : (x2 ? it.z : 0u);
    b.s2 = (far && b.nex > 2) ? ex_s[b.xo + 2] : 0u;
    b.e2 = (far && b.nex > 2) ? ex_e[b.xo + 2] : 0u;
    return b;
  };
  auto set_mirror = [&](int p) {  // (uniform) node p becomes the active island
    P = p;
    if (p >= 0) {
      SP = rl(ns, p);
      EP = rl(ne, p);
      DP = p + 1 < cnt ? rl(ns, p + 1) : 0xFFFFFFFFu;
    }
    mirror_ok = true;
  };
  Batch nxt = load_batch(t0);
  for (uint32_t tb = t0; tb < t1 && !overflow; tb += 64) {
    const Batch cur = nxt;
    if (tb + 64 < t1) nxt = load_batch(tb + 64);
    const bool have = tb + (uint32_t)lane < t1;
    const uint32_t it_start = cur.start, it_nex = cur.nex, it_xo = cur.xo, it_o = cur.o, it_e0 = cur.e0;
    const uint32_t it_s1 = cur.s1, it_e1 = cur.e1, it_s2 = cur.s2, it_e2 = cur.e2;
    int it_d = 0;
    const int nb = (int)((t1 - tb) < 64u ? (t1 - tb) : 64u);
    for (int j = 0; j < nb && !overflow; ++j) {
      const uint32_t nex = rl(it_nex, j);
      const uint32_t hd_start = rl(it_start, j), hd_e0 = rl(it_e0, j);
      if (!mirror_ok) {  // the island this item would extend: the last node that starts before it
        const uint64_t lt = __ballot(lane < cnt && ns < hd_start);
        set_mirror((lt == ~0ull ? 64 : __builtin_ctzll(~lt)) - 1);
      }
      // ---- fast path: a maximal run of items that leave the list structure alone.  Exon 0 of such an item starts inside the
      // active island (SP < start <= EP) and stays clear of the next node (end < DP): processRead finds prev = node P (the node
      // behind it starts beyond EP) with prev.end >= start, so d = start - SP, and mergeRead — whose walk passes the dead nodes
      // before P without stopping — only stretches node P's end: no clearTo, no insertion, no swallow.  Exons 1 and 2 (spliced
      // reads) must each start inside an existing later node and end before that node's successor starts: mergeRead then only
      // raises that node's end (max), again without surgery.  Node starts never move inside a run and every raised end stays
      // below the following start, so testing against the list as it was when the run began is exact.  The run is found for all
      // remaining lanes of the batch at once: a wave prefix-max of the exon-0 ends gives node P's end before each item; the
      // later nodes' ends are max-reduced through LDS afterwards.
      // (the head item's exon 0 is tested on the scalar side first, so an item that cannot start a run costs three
      // compares rather than the whole window analysis)
      if (nex <= 3u && P >= 0 && SP < hd_start && hd_e0 < DP && hd_start <= EP) {
        const bool inwin = lane >= j && lane < nb;
        const uint32_t pm = wave_prefix_max(inwin ? it_e0 : 0u);
        uint32_t ex = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pm, 0x138, 0xf, 0xf, false);  // wave_shr:1 -> exclusive
        uint32_t ebefore = ex > EP ? ex : EP;
        bool xok = it_nex == 1u;
        bool u1 = false, u2 = false;  // exon 1 / exon 2 raises the end of node n1 / n2
        uint32_t n1 = 0, n2 = 0;
        const uint64_t multi = __ballot(inwin && it_nex >= 2u && it_nex <= 3u);
        if (multi != 0 && cnt - P <= YD_FAST_NODES) {
          bool h1 = false, h2 = false;
          uint32_t UL;  // upper bound, over this window, of the end of the last node
          if (P + 1 == cnt) {
            UL = ebefore;  // node P is the last node; its end just before this item (own exon 0 ends before exon 1 starts)
          } else {
            uint32_t Sm = DP, Em = 0;  // DP == start of node P + 1
            for (int m = P + 1; m < cnt; ++m) {
              Em = rl(ne, m);
              const uint32_t NS = (m + 1 < cnt) ? rl(ns, m + 1) : 0xFFFFFFFFu;
              if (Sm <= it_s1 && it_s1 <= Em && it_e1 < NS) {
                h1 = true;
                n1 = (uint32_t)m;
              }
              if (Sm <= it_s2 && it_s2 <= Em && it_e2 < NS) {
                h2 = true;
                n2 = (uint32_t)m;
              }
              Sm = NS;
            }
            const uint32_t L = (uint32_t)cnt - 1u;
            uint32_t cand = 0;
            if (inwin && it_nex >= 2u && h1 && n1 == L) cand = it_e1;
            if (inwin && it_nex == 3u && h1 && h2 && n2 == L && it_e2 > cand) cand = it_e2;
            const uint32_t wm = rl(wave_prefix_max(cand), 63);
            UL = wm > Em ? wm : Em;
          }
          // an exon that starts beyond the end of the last node runs off the list: it and the rest of the read are dropped
          const bool off1 = it_s1 > UL, off2 = it_s2 > UL;
          if (it_nex == 2u) {
            xok = h1 || off1;
            u1 = h1;
          } else if (it_nex == 3u) {
            xok = off1 || (h1 && (h2 || off2));
            u1 = h1;
            u2 = h1 && h2;
          }
        }
        bool okl = inwin && xok && SP < it_start && it_e0 < DP && it_start <= ebefore;
        uint64_t mk = __ballot(okl) >> j;
        int r = mk == ~0ull ? 64 : __builtin_ctzll(~mk);
        if (r > nb - j) r = nb - j;
        if (r > 0) {
          const bool inrun = lane >= j && lane < j + r;
          if (inrun) it_d = (int)(it_start - SP);
          uint32_t newE = rl(pm, j + r - 1);
          if (newE > EP) EP = newE;
          if (lane == P) ne = EP;
          if (__ballot(inrun && u1) != 0) {  // raise the ends of the nodes the later exons landed in
            __shared__ uint32_t upd[64];
            upd[lane] = 0u;
            __syncthreads();
            if (inrun && u1) atomicMax(&upd[n1], it_e1);
            if (inrun && u2) atomicMax(&upd[n2], it_e2);
            __syncthreads();
            const uint32_t u = upd[lane];
            if (lane > P && lane < cnt && u > ne) ne = u;
            __syncthreads();
          }
          last_pos = rl(it_start, j + r - 1);
          last_dist = (int)(last_pos - SP);
          j += r - 1;
          continue;
        }
      }
      const uint32_t rstart = hd_start;
      const uint32_t xo = rl(it_xo, j);
      const uint32_t e0 = hd_e0;
      int d;
      // ---- processRead :221-250
      if (last_pos == rstart) {
        d = last_dist;
      } else {
        d = 0;
        if (cnt >= 1) {
          int np;  // leading run of nodes that start before the read (the list is sorted)
          if (P >= 0 && SP < rstart && DP >= rstart) {
            np = P + 1;  // still node P: all scalar
          } else {
            const uint64_t lt = __ballot(lane < cnt && ns < rstart);
            np = lt == ~0ull ? 64 : __builtin_ctzll(~lt);
            if (np > 0) set_mirror(np - 1);
          }
          if (np > 0) {
            if (EP >= rstart) {
              d = (int)(rstart - SP);
            } else {  // clearTo(prev): drop the first np nodes
              if (np == 1) {
                ns = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ns, 0x130, 0xf, 0xf, false);
                ne = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ne, 0x130, 0xf, 0xf, false);
              } else {
                ns = __shfl(ns, lane + np, 64);
                ne = __shfl(ne, lane + np, 64);
              }
              cnt -= np;
              set_mirror(cnt > 0 ? 0 : -1);  // (what follows tests the read against the new first node)
            }
          }
        }
        last_pos = rstart;
        last_dist = d;
      }
      // ---- mergeRead :167-219
      if (cnt == 0) {
        if (nex > 64) {
          overflow = true;
        } else {
          if (nex <= 3u) {  // the first three exons travel with the batch: no global load on the critical path
            const uint32_t s1 = rl(it_s1, j), e1 = rl(it_e1, j), s2 = rl(it_s2, j), e2 = rl(it_e2, j);
            if (lane == 0) {
              ns = rstart;
              ne = e0;
            } else if (lane == 1) {
              ns = s1;
              ne = e1;
            } else if (lane == 2) {
              ns = s2;
              ne = e2;
            }
          } else if ((uint32_t)lane < nex) {
            ns = ex_s[xo + lane];
            ne = ex_e[xo + lane];
          }
          cnt = (int)nex;
          mirror_ok = false;
        }
      } else {
        int cur = 0;
        uint32_t k = 0;
        // exon 0 against the active island, all scalar, when it starts inside it and swallows nothing (the walk passes the
        // dead nodes before P: they end below SP <= start)
        // (a first node may also be entered from the left — the usual thing right after a clearTo —: nothing lies before it)
        if (mirror_ok && P >= 0 && rstart <= EP && (P == 0 ? e0 >= SP : SP <= rstart)) {
          uint32_t newE = e0 > EP ? e0 : EP;
          if (newE < DP) {
            if (rstart < SP) SP = rstart;
            EP = newE;
            if (lane == P) {
              ns = SP;
              ne = EP;
            }
            k = 1;
            cur = P;
          }
        }
        if (k < nex) {
          const uint32_t s1 = rl(it_s1, j), e1 = rl(it_e1, j), s2 = rl(it_s2, j), e2 = rl(it_e2, j);
          for (; k < nex; ++k) {
            uint32_t es = k == 0 ? rstart : (k == 1 ? s1 : (k == 2 ? s2 : ex_s[xo + k]));
            uint32_t ee = k == 0 ? e0 : (k == 1 ? e1 : (k == 2 ? e2 : ex_e[xo + k]));
            uint64_t stop = __ballot(lane >= cur && lane < cnt && (ee < ns || es <= ne));
            if (stop == 0) break;  // ran off the list: this exon and the rest are dropped
            int n = __builtin_ctzll(stop);
            if (n <= P + 1) mirror_ok = false;
            uint32_t nS = rl(ns, n), nE = rl(ne, n);
            if (ee < nS) {  // insert before n
              if (cnt == 64) {
                overflow = true;
                break;
              }
              // one-lane shifts are DPP wave shifts (single VALU op) instead of ds_bpermute round trips
              uint32_t us = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ns, 0x138, 0xf, 0xf, false);
              uint32_t ue = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ne, 0x138, 0xf, 0xf, false);
              if (lane > n) {
                ns = us;
                ne = ue;
              } else if (lane == n) {
                ns = es;
                ne = ee;
              }
              cnt++;
              cur = n + 1;
            } else {  // overlap: union, then swallow followers (stops after the first one that extends the node)
              uint32_t newS = es < nS ? es : nS;
              uint32_t newE = ee > nE ? ee : nE;
              while (n + 1 < cnt) {
                uint32_t xS = rl(ns, n + 1);
                if (xS > newE) break;
                uint32_t xE = rl(ne, n + 1);
                uint32_t ds = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ns, 0x130, 0xf, 0xf, false);
                uint32_t de = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ne, 0x130, 0xf, 0xf, false);
                if (lane > n) {
                  ns = ds;
                  ne = de;
                }
                cnt--;
                if (xE > newE) {
                  newE = xE;
                  break;
                }
              }
              if (lane == n) {
                ns = newS;
                ne = newE;
              }
              cur = n;
            }
          }
        }
      }
      if (lane == j) it_d = d;
    }
    if (!overflow && have) {
      if (yd_d)
        yd_d[tb + (uint32_t)lane] = it_d > 0 ? (uint32_t)it_d : 0u;  // (one 256-byte store per batch)
      else if (it_d > 0)
        atomicMax(&g_yd[it_o], it_d);
    }
  }
  if (overflow && lane == 0) ids_over[atomicAdd(n_over, 1u)] = c;
}

// ---- outputs ---------------------------------------------------------------------------------------------
__global__ void col_write_yd_k(uint32_t ng, const uint32_t* __restrict__ gperm, GroupAcc G, const int32_t* __restrict__ g_yd,
                               uint32_t cap, int32_t* __restrict__ yd) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng || o >= cap) return;
  int dmax = G.ydin ? (int)G.ydin[gperm[o]] : 0;  // int dmax=spd.maxYD (tiebrush.cpp:511); null: no carried YD (plain inputs, window path)
  int d2 = g_yd[o];
  if (d2 > dmax) dmax = d2;
  yd[o] = dmax > 0 ? dmax : 0;
}

// Items placed by list: the chain kernels leave every item's distance in the item's own slot (yd_d, written where the item was read:
// whole lines) instead of one device-scope atomicMax per item on the groups' array — on a chip of eight L2s such an atomic is a round
// trip to the memory side, 22 bytes of write traffic each by the counters (3.7 GB per launch of yd_wave_k on config 3).  This pass
// folds them per tile of YS_NT groups: the tile's items of list c are the segment [base[c], base[c] + n[c]) of that list (the
// placement's own table), their group words say where they belong, and the maxima meet in LDS.  It writes the YD output itself
// (col_write_yd_k's job on the other paths: int dmax = spd.maxYD, tiebrush.cpp:511-524).
__global__ __launch_bounds__(512) void yd_lgather_k(uint32_t ng, uint32_t ntiles, const uint32_t* __restrict__ table,
                                                    const uint64_t* __restrict__ totals, const uint32_t* __restrict__ yd_d,
                                                    const uint32_t* __restrict__ item, const uint32_t* __restrict__ gperm, GroupAcc G,
                                                    uint32_t cap, int32_t* __restrict__ yd) {
  __shared__ uint32_t base[YS_NL], lsum[YS_NL];
  __shared__ int32_t mx[YS_NT];
  const uint32_t t = threadIdx.x, b = blockIdx.x;
  for (uint32_t g = t; g < YS_NT; g += 512) mx[g] = 0;
  uint32_t first = 0, next = 0;
  if (t < YS_NL) {  // (the list's total and the tile's two table entries in one round trip)
    const uint32_t tot = (uint32_t)totals[t];
    first = table[(size_t)t * ntiles + b];
    next = b + 1 < ntiles ? table[(size_t)t * ntiles + b + 1] : tot;
    lsum[t] = tot;
  }
  __syncthreads();
  uint32_t lb = 0;
  if (t < YS_NL)
    for (uint32_t c = 0; c < t; ++c) lb += lsum[c];
  __syncthreads();
  if (t < YS_NL) {
    base[t] = lb + first;
    lsum[t] = next - first;
  }
  __syncthreads();
  // a wave per segment (a tile holds ~ 56 items of a list: one round of 64 lanes, whole lines, no search for the item's list)
  // (the sixteen segments of a wave are asked for together: one round trip to memory per block instead of sixteen dependent ones —
  // the pass is bound by that latency, not by its 1.3 GB)
  const uint32_t wv = t >> 6, ln = t & 63u;
  constexpr uint32_t NSEG = YS_NL / (512 / 64);
  uint32_t dv[NSEG], gv[NSEG];
#pragma unroll
  for (uint32_t j = 0; j < NSEG; ++j) {
    const uint32_t c = wv + j * (512 / 64);
    const uint32_t p0 = base[c], n = lsum[c];
    dv[j] = ln < n ? yd_d[p0 + ln] : 0u;
    gv[j] = ln < n ? item[p0 + ln] : 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < NSEG; ++j)
    if (dv[j]) atomicMax(&mx[gv[j] - b * YS_NT], (int32_t)dv[j]);
  for (uint32_t j = 0; j < NSEG; ++j) {  // segments of more than 64 items (a tile of 1024 groups may hold 1024 of a list)
    const uint32_t c = wv + j * (512 / 64);
    const uint32_t p0 = base[c], n = lsum[c];
    for (uint32_t i = 64 + ln; i < n; i += 64) {
      const uint32_t d = yd_d[p0 + i];
      if (d) atomicMax(&mx[item[p0 + i] - b * YS_NT], (int32_t)d);
    }
  }
  __syncthreads();
  for (uint32_t g = t; g < YS_NT; g += 512) {
    const uint32_t o = b * YS_NT + g;
    if (o >= ng || o >= cap) continue;
    int dmax = G.ydin ? (int)G.ydin[gperm[o]] : 0;
    const int d2 = mx[g];
    if (d2 > dmax) dmax = d2;
    yd[o] = dmax > 0 ? dmax : 0;
  }
}

// TIED_ONLY: the compaction pass of the window path has written every group at its key-order place (WgDirectOut), which is its output
// place unless it belongs to a tie set of several groups (tie[] marks the set heads): only those are written here, from the
// accumulators, through the set's permutation — one byte per group read for the rest.
template <bool TIED_ONLY>
__global__ void col_write_k(const uint64_t* __restrict__ png, const uint32_t* __restrict__ gperm, GroupAcc G,
                            const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, uint32_t cap,
                            uint32_t* __restrict__ rep, double* __restrict__ yc, int64_t* __restrict__ yx,
                            int32_t* __restrict__ g_start, int32_t* __restrict__ g_end, const int32_t* __restrict__ effend,
                            int32_t* __restrict__ rep_effend, uint64_t* __restrict__ g_key, int strategy) {
  const uint32_t ng = (uint32_t)*png;
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng || o >= cap) return;
  if constexpr (TIED_ONLY) {
    if (G.tie[o] && (o + 1 >= ng || G.tie[o + 1])) return;  // a set of one: already in place
  }
  uint32_t sg = gperm[o];
  rep[o] = (uint32_t)(G.rep[sg] & 0xFFFFFFFFull);
  if (rep_effend) rep_effend[o] = effend ? effend[rep[o]] : (int32_t)(uint32_t)(G.rep[sg] >> 32);
  yc[o] = G.yc[sg];
  yx[o] = (G.yxin ? (int64_t)G.yxin[sg] : 0ll) + (int64_t)G.ns[sg];
  uint32_t q = G.first[sg];
  int32_t st = (int32_t)(uint32_t)((shi[q] >> 2) & 0x7FFFFFFFull);
  if (g_start) g_start[o] = st;
  if (g_end) g_end[o] = st + (int32_t)(uint32_t)(slo[q] >> 32) - 1;
  if (g_key) {  // tbk_groups_out.g_key: the place from the group key; the shape only where the key word says what the CIGAR looks like
    const uint64_t lo = slo[q];
    const uint32_t h32 = (uint32_t)lo;
    uint32_t shape = 0;
    if (strategy == TBK_STRAT_CIGAR || strategy == TBK_STRAT_CLIP) {  // (-E codes speak of exons, which may hold I and D)
      if (h32 == (0x80000000u | C_M)) shape = 0x80000000u;
      if ((h32 >> 30) == 3u) shape = h32;
    }
    g_key[2 * (size_t)o] = shi[q];
    g_key[2 * (size_t)o + 1] = (lo & 0xFFFFFFFF00000000ull) | shape;
  }
}
__global__ void col_recgroup_k(const uint64_t* __restrict__ pm, const uint32_t* __restrict__ val, const uint32_t* __restrict__ sgid,
                               const uint32_t* __restrict__ ginv, int32_t* __restrict__ rec_group) {
  const uint32_t m = (uint32_t)*pm;
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < m) rec_group[val[q]] = (int32_t)ginv[sgid[q]];
}

__global__ void col_init_groups_k(const uint64_t* __restrict__ png, GroupAcc G, int32_t* __restrict__ g_yd) {
  const uint32_t ng = (uint32_t)*png;
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  G.yc[g] = 0.0;
  G.ns[g] = 0;
  G.yxin[g] = 0;
  G.ydin[g] = 0;
  G.rep[g] = ~0ull;
  g_yd[g] = 0;
}

struct YdJob {  // everything the YD stage needs from the main stage (device pointers stay valid until it has run)
  ColIn I;
  uint32_t m, ng, cap;
  const uint32_t* val;
  const uint8_t* flags;
  const uint16_t* fidx;
  const uint32_t *sgid, *ginv, *gperm;
  GroupAcc G;
  const uint64_t *shi, *slo;
  int32_t* g_yd;
  int32_t* out_yd;
  // window path: incidences instead of sorted records (val / flags / fidx / sgid are unused then)
  bool win = false;
  uint32_t np = 0;
  const uint16_t* pfile = nullptr;
  const uint32_t *pgrp = nullptr, *gpoff = nullptr;
  const uint64_t* gfmask = nullptr;  // (window path, <= 64 files) the groups' files as bit masks instead of incidences
};

}  // namespace

// The YD stage (tiebrush.cpp:511-524 for every flushed group) — runs on `ctx`'s stream / arena / scalars, which may be
// the calling context (inline) or the context's private side context on a helper thread (deferred, overlapping the
// caller's next calls such as the tiecov chain, which does not depend on YD).
int tbk_collapse_yd_run(tbk_ctx* ctx, void* jobp) {
  const YdJob J = *(const YdJob*)jobp;
  delete (YdJob*)jobp;
  const uint32_t B = 256;
  const uint32_t m = J.m, ng = J.ng;
  const ColIn& I = J.I;
  uint64_t* sc = ctx->d_scalars;
  TBK_HIP(hipSetDevice(ctx->device));
  // (phases=1: the stage's wall time between its read-backs, to stderr)
  const bool ph = ctx->dbg.phases;
  const auto p0 = std::chrono::steady_clock::now();
  auto p1 = p0, p2 = p0, p3 = p0;
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  TBK_HIP(hipMemsetAsync(sc, 0, 32 * sizeof(uint64_t), ctx->stream));
  {
    uint32_t *icnt = nullptr, *ioff = nullptr, *ocnt = nullptr, *ooff = nullptr;
    uint64_t nit64;
    const bool by_list = J.win && (!J.pgrp || tbk_yd_by_list(ctx, I.k));  // (no per-incidence group array: the window stage has decided)
    static_assert(YS_NL == 128, "tbk_yd_by_list (wgroup.h) knows the number of lists");
    const uint32_t ys_tiles = cdiv(ng, YS_NT);
    const YsIn S{ng, J.gperm, J.G.ns, J.gpoff, J.pfile, J.shi, J.gfmask};
    uint32_t* ys_table = nullptr;
    uint4* ys_agg = nullptr;
    uint64_t* ys_totals = nullptr;
    uint64_t* ys_files = nullptr;
    uint32_t *yd_d = nullptr, *ys_item = nullptr;
    YdGroups Q{};
    if (by_list) {  // the groups' coordinates first (the first pass folds them); then items and aggregates per (list, tile of groups)
      Q.pk = ws_alloc<uint4>(ctx, ng);
      Q.nex = ws_alloc<uint32_t>(ctx, ng);
      Q.xoff = ws_alloc<uint32_t>(ctx, ng);
      ys_table = ws_alloc<uint32_t>(ctx, (size_t)YS_NL * ys_tiles);
      ys_agg = ws_alloc<uint4>(ctx, (size_t)YS_NL * ys_tiles);
      ys_totals = ws_alloc<uint64_t>(ctx, YS_NL + 1);
      ys_files = ws_alloc<uint64_t>(ctx, 2 * (size_t)ng);
      if (!Q.xoff || !ys_table || !ys_agg || !ys_totals || !ys_files) return TBK_ENOMEM;
      TBK_LAUNCH(ctx, "yd_groups", yd_groups_k, cdiv(ng, B), B, 0, I, ng, J.gperm, J.G, J.shi, J.slo, Q);
      TBK_TRY(tbk_exscan_u32(ctx, Q.nex, Q.xoff, ng, sc + 5));
      TBK_LAUNCH(ctx, "yd_lcount", yd_lcount_k, ys_tiles, YS_NT, 0, S, Q.pk, ys_tiles, ys_table, ys_agg, ys_files);
      TBK_LAUNCH(ctx, "yd_lscan", yd_lscan_k, YS_NL, 1024, 0, ys_table, ys_tiles, ys_totals, (unsigned long long*)(sc + 2));
      TBK_LAUNCH(ctx, "yd_lscan", yd_lagg_scan_k, YS_NL, 256, 0, ys_agg, ys_tiles);
    } else if (J.win) {  // items per output group straight from the per-group sample counts
      ocnt = ws_alloc<uint32_t>(ctx, ng);
      ooff = ws_alloc<uint32_t>(ctx, ng);
      if (!ooff) return TBK_ENOMEM;
      TBK_LAUNCH(ctx, "yd_gcount", yd_gcount_w_k, cdiv(ng, B), B, 0, ng, J.gperm, J.G, J.shi, ocnt);
      TBK_TRY(tbk_exscan_u32(ctx, ocnt, ooff, ng, sc + 2));
    } else {
      icnt = ws_alloc<uint32_t>(ctx, m);
      ioff = ws_alloc<uint32_t>(ctx, m);
      if (!ioff) return TBK_ENOMEM;
      TBK_LAUNCH(ctx, "yd_count", yd_count_k, cdiv(m, B), B, 0, I, m, J.val, J.flags, J.fidx, icnt);
      TBK_TRY(tbk_exscan_u32(ctx, icnt, ioff, m, sc + 2));
    }
    TBK_HIP(hipMemcpyAsync(ctx->h_scalars, sc, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    nit64 = ctx->h_scalars[2];
    p1 = p2 = p3 = std::chrono::steady_clock::now();
    if (nit64 >= (1ull << 32)) return TBK_E2BIG;
    const uint32_t nit = (uint32_t)nit64;
    if (nit) {
      // items: one word each, list id (file * 2 + strand list) : 32 | group in output order : 32
      uint64_t* iv = by_list ? nullptr : ws_alloc<uint64_t>(ctx, nit);
      uint64_t* iv2 = by_list ? nullptr : ws_alloc<uint64_t>(ctx, nit);
      uint32_t* io = by_list ? ws_alloc<uint32_t>(ctx, nit) : nullptr;  // items placed by list: the item -> group array
      yd_d = by_list ? ws_alloc<uint32_t>(ctx, nit) : nullptr;          // ... and every item's distance (yd_lgather_k folds them)
      if (by_list && (!io || !yd_d)) return TBK_ENOMEM;
      ys_item = io;
      YdItems Y;
      if (by_list) {
        Y.pk = nullptr;
        Y.se = ws_alloc<uint2>(ctx, nit);
        Y.gxoff = Q.xoff;  // (by_list: the groups' records were made above)
        if (!Y.se) return TBK_ENOMEM;
      } else {
        Y.pk = ws_alloc<uint4>(ctx, nit);
      }
      Y.nex = ws_alloc<uint32_t>(ctx, nit);
      uint32_t* noff = ws_alloc<uint32_t>(ctx, nit);
      uint32_t* chain_first = ws_alloc<uint32_t>(ctx, nit);
      if (!chain_first) return TBK_ENOMEM;
      if (by_list) {
      } else if (J.win) {
        TBK_LAUNCH(ctx, "yd_fill", yd_fill_w_k, cdiv(J.np, B), B, 0, J.np, J.pfile, J.pgrp, J.gpoff, J.ginv, ooff, J.shi, iv);
      } else {
        ocnt = ws_alloc<uint32_t>(ctx, ng);
        ooff = ws_alloc<uint32_t>(ctx, ng);
        if (!ooff) return TBK_ENOMEM;
        TBK_LAUNCH(ctx, "yd_gcount", yd_gcount_k, cdiv(ng, B), B, 0, ng, m, J.gperm, J.G, ioff, icnt, ocnt);
        TBK_TRY(tbk_exscan_u32(ctx, ocnt, ooff, ng, nullptr));
        TBK_LAUNCH(ctx, "yd_fill", yd_fill_k, cdiv(m, B), B, 0, I, m, J.val, J.flags, J.fidx, J.sgid, J.ginv, ioff, ooff, J.G, iv);
      }
      if (!by_list) {
        Q.pk = ws_alloc<uint4>(ctx, ng);
        Q.nex = ws_alloc<uint32_t>(ctx, ng);
        Q.xoff = ws_alloc<uint32_t>(ctx, ng);
        if (!Q.xoff) return TBK_ENOMEM;
        TBK_LAUNCH(ctx, "yd_groups", yd_groups_k, cdiv(ng, B), B, 0, I, ng, J.gperm, J.G, J.shi, J.slo, Q);
        TBK_TRY(tbk_exscan_u32(ctx, Q.nex, Q.xoff, ng, sc + 5));
      }
      if (by_list) {
        TBK_LAUNCH(ctx, "yd_scatter", yd_lscatter_k, ys_tiles, YS_NT, 0, S, ys_tiles, ys_table, ys_totals, ys_agg, ys_files, Q, Y, io);
      } else {  // stable split by list id (file * 2 + strand list); group order is already in place.  The id range is known:
        uint32_t bits = 1;  // no scan for the varying bits
        while ((1ull << bits) < 2ull * I.k) ++bits;
        TBK_TRY(tbk_radix_sort_w64_emit(ctx, &iv, &iv2, nit, ((1ull << bits) - 1ull) << 32, true, YdEmit{Q, Y}, "yd_scatter"));
      }
      if (by_list) {  // the heads are flagged: number them (and sum the exon counts before each) in one scan of the exon-count words
        const uint32_t ynb = cdiv(nit, YN_TILE);
        unsigned long long* ypart = ws_alloc<unsigned long long>(ctx, ynb);
        if (!ypart) return TBK_ENOMEM;
        TBK_LAUNCH(ctx, "yd_number", yn_reduce_k, ynb, YN_NT, 0, Y.nex, nit, ypart);
        TBK_LAUNCH(ctx, "yd_number", yn_spine_k, 1, 1024, 0, ypart, ynb, sc + 3);
        TBK_LAUNCH(ctx, "yd_number", yn_emit_k, ynb, YN_NT, 0, Y.nex, nit, ypart, chain_first, noff);
      } else {
        YdLoad ld{iv, Y};  // chain heads (segmented running maximum of the ends) and their numbering, one pass
        YdAux ax{Y};
        YdHead hd{};
        YdStore st{Y, chain_first, noff, sc + 3, nit};
        SegMaxY ident{INT32_MIN, 0u};
        TBK_TRY((scan_two_run<8, SegMaxY, SegMaxYOp, HeadNex, HeadNexOp, YdLoad, YdAux, YdHead, YdStore>(ctx, "yd_chains", nit, ld, ax, hd, st, SegMaxYOp{}, ident,
                                                                                              HeadNexOp{}, HeadNex{0u, 0u})));
      }
      TBK_HIP(hipMemcpyAsync(ctx->h_scalars, sc, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
      TBK_HIP(hipStreamSynchronize(ctx->stream));
      const uint32_t nchains = (uint32_t)ctx->h_scalars[3];
      const uint64_t nnodes = ctx->h_scalars[4], ngex = ctx->h_scalars[5];
      p2 = p3 = std::chrono::steady_clock::now();
      if (nnodes >= (1ull << 31)) return TBK_E2BIG;
      SegNodes N;
      N.s = ws_alloc<uint32_t>(ctx, nnodes + 1);
      N.e = ws_alloc<uint32_t>(ctx, nnodes + 1);
      N.nx = ws_alloc<int32_t>(ctx, nnodes + 1);
      uint32_t* ex_s = ws_alloc<uint32_t>(ctx, ngex + 1);
      uint32_t* ex_e = ws_alloc<uint32_t>(ctx, ngex + 1);
      uint32_t* ids_lane = ws_alloc<uint32_t>(ctx, nchains);
      uint32_t* ids_long = ws_alloc<uint32_t>(ctx, nchains);
      uint32_t* ids_over = ws_alloc<uint32_t>(ctx, nchains);
      uint32_t* bcnt = ws_alloc<uint32_t>(ctx, 4 * YD_NB + 4);  // bucket counts [2 * YD_NB], cursors [2 * YD_NB + 1], [n_over]
      if (!ids_over || !bcnt) return TBK_ENOMEM;
      uint32_t* bcur = bcnt + 2 * YD_NB;
      uint32_t* n_wave = bcur + 2 * YD_NB;
      uint32_t* n_over = n_wave + 1;
      TBK_HIP(hipMemsetAsync(bcnt, 0, (4 * YD_NB + 4) * sizeof(uint32_t), ctx->stream));
      TBK_LAUNCH(ctx, "yd_gexons", yd_gexons_k, cdiv(ng, B), B, 0, I, ng, J.gperm, J.G, J.slo, Q, ex_s, ex_e);
      // chains bucketed by the log2 of their length, longest first: the lanes of a wave of yd_lane_k run chains of like length
      // (yd_wave_min: test hook — 1: every chain to yd_wave_k; huge: every chain to yd_lane_k; yd_bgrid: several chains per thread)
      const uint32_t wave_min = ctx->dbg.yd_wave_min ? ctx->dbg.yd_wave_min : YD_WAVE_MIN_DEFAULT;
      const uint32_t bgrid = ctx->dbg.yd_bgrid ? ctx->dbg.yd_bgrid : std::min(cdiv(nchains, B), YD_BGRID);
      TBK_LAUNCH(ctx, "yd_classify", yd_bucket_count_k, bgrid, B, 0, nchains, nit, wave_min, chain_first, bcnt);
      TBK_LAUNCH(ctx, "yd_classify", yd_bucket_off_k, 1, 1, 0, bcnt, bcur);
      TBK_LAUNCH(ctx, "yd_classify", yd_bucket_fill_k, bgrid, B, 0, nchains, nit, wave_min, chain_first, bcur, ids_lane, ids_long);
      uint32_t* hc = (uint32_t*)(ctx->h_scalars + 24);
      TBK_HIP(hipMemcpyAsync(hc, n_wave, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      TBK_HIP(hipStreamSynchronize(ctx->stream));
      const uint32_t n_long = hc[0], n_lane = nchains - n_long;
      p3 = std::chrono::steady_clock::now();
      // lane chains and wave chains are independent: the few long, latency-bound waves go to the auxiliary stream beside the lane
      // kernel.  (The stream was synchronised just above, so the fork needs no event; the join does.)
      const bool literal = ctx->dbg.yd_literal;  // test hook: the literal machine runs every chain
      if (literal) {
        if (n_long) TBK_HIP(hipMemcpyAsync(ids_over, ids_long, (size_t)n_long * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (n_lane) TBK_HIP(hipMemcpyAsync(ids_over + n_long, ids_lane, (size_t)n_lane * 4, hipMemcpyDeviceToDevice, ctx->stream));
        hc[1] = nchains;
        TBK_HIP(hipMemcpyAsync(n_over, hc + 1, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        TBK_HIP(hipStreamSynchronize(ctx->stream));  // (hc is reused)
      }
      hipStream_t aux = n_lane && n_long && !literal ? tbk_aux_stream(ctx) : nullptr;
      if (n_long && !literal) {
        hipStream_t keep = ctx->stream;
        if (aux) ctx->stream = aux;
        TBK_LAUNCH(ctx, "yd_wave", yd_wave_k, n_long, 64, 0, ids_long, n_wave, nchains, nit, chain_first, Y, YdWords{iv, io}, noff, ex_s, ex_e, J.g_yd,
                   yd_d, ids_over, n_over);
        ctx->stream = keep;
        if (aux) TBK_HIP(hipEventRecord(ctx->aux_done, aux));
      }
      if (n_lane && !literal)
        TBK_LAUNCH(ctx, "yd_lane", yd_lane_k<8>, cdiv(n_lane, 64), 64, 0, ids_lane, n_lane, nchains, nit, chain_first, Y, YdWords{iv, io}, ex_s, ex_e, J.g_yd,
                   yd_d, ids_over, n_over);
      if (n_long && aux) TBK_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_done, 0));
      // chains whose list outgrew its lane's / wave's slots (count only known on the device: launch for the upper bound)
      TBK_LAUNCH(ctx, "yd_run_overflow", yd_run_k, cdiv(nchains, 64), 64, 0, ids_over, n_over, nchains, nit, chain_first, Y, YdWords{iv, io}, noff, ex_s, ex_e,
                 N, J.g_yd, yd_d);
    }
    if (by_list) {  // the distances lie with the items (no item: every segment is empty): folded per tile of groups, written as YD
      TBK_LAUNCH(ctx, "yd_gather", yd_lgather_k, ys_tiles, 512, 0, ng, ys_tiles, ys_table, ys_totals, yd_d, ys_item, J.gperm, J.G, J.cap, J.out_yd);
      TBK_HIP(hipStreamSynchronize(ctx->stream));
      if (ph)
        fprintf(stderr, "YD stage ms: counts %.1f | placement + numbering %.1f | buckets %.1f | list machines + gather %.1f (arena %.2f of %.2f GB, %zu overflow chunks)\n",
                std::chrono::duration<double, std::milli>(p1 - p0).count(), std::chrono::duration<double, std::milli>(p2 - p1).count(),
                std::chrono::duration<double, std::milli>(p3 - p2).count(), since(p3), ctx->ws_off / 1e9, ctx->ws_cap / 1e9, ctx->ws_overflow.size());
      return tbk_check_launch(ctx, "collapse_yd");
    }
  }
  TBK_LAUNCH(ctx, "col_write_yd", col_write_yd_k, cdiv(ng, B), B, 0, ng, J.gperm, J.G, J.g_yd, J.cap, J.out_yd);
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "collapse_yd");
}

// tbk_warmup: what a context's FIRST YD stage pays beyond its kernels — the auxiliary stream (a hardware queue: milliseconds to create)
// and the first dispatch of each list machine on its queue — paid ahead, with launches that find no work (*nids = 0 / nids = 0).
int tbk_collapse_warm(tbk_ctx* ctx) {
  TBK_HIP(hipSetDevice(ctx->device));
  uint64_t* sc = ctx->d_scalars;
  TBK_HIP(hipMemsetAsync(sc, 0, 32 * sizeof(uint64_t), ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  uint32_t* z = (uint32_t*)sc;  // zeros: the id list, the counts, chain_first[0]
  YdItems Y{};
  Y.pk = (uint4*)sc, Y.nex = z;
  const YdWords W{nullptr, z};
  const SegNodes N{z, z, (int32_t*)z};
  hipStream_t aux = tbk_aux_stream(ctx);
  if (aux) {
    hipStream_t keep = ctx->stream;
    ctx->stream = aux;
    TBK_LAUNCH(ctx, "yd_wave", yd_wave_k, 1, 64, 0, z, z, 0u, 0u, z, Y, W, z, z, z, (int32_t*)nullptr, z + 8, z + 16, z + 1);
    ctx->stream = keep;
    TBK_HIP(hipEventRecord(ctx->aux_done, aux));
    TBK_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_done, 0));
  } else {
    TBK_LAUNCH(ctx, "yd_wave", yd_wave_k, 1, 64, 0, z, z, 0u, 0u, z, Y, W, z, z, z, (int32_t*)nullptr, z + 8, z + 16, z + 1);
  }
  TBK_LAUNCH(ctx, "yd_lane", yd_lane_k<8>, 1, 64, 0, z, 0u, 0u, 0u, z, Y, W, z, z, (int32_t*)nullptr, z + 8, z + 16, z + 1);
  TBK_LAUNCH(ctx, "yd_run_overflow", yd_run_k, 1, 64, 0, z, z, 0u, 0u, z, Y, W, z, z, z, N, (int32_t*)nullptr, z + 8);
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "warmup");
}

// =============================================================================================================
int tbk_collapse_device(tbk_ctx* ctx, const tbk_collapse_opts* o, const tbk_soa_in* in, tbk_groups_out* out) {
  const uint32_t n = in->n_records;
  const uint32_t B = 256;
  out->n_groups = 0;
  out->n_passed = 0;
  if (out->rec_group) TBK_HIP(hipMemsetAsync(out->rec_group, 0xFF, (size_t)n * 4, ctx->stream));
  if (n == 0) return 0;
  uint64_t* sc = ctx->d_scalars;  // [0]=n_pass [1]=n_groups [2]=n_items [3]=n_chains [4]=n_nodes
  ColIn I;
  I.n = n;
  I.k = in->n_files;
  {
    uint32_t* d_fo = ws_alloc<uint32_t>(ctx, in->n_files + 1);
    uint8_t* d_tb = ws_alloc<uint8_t>(ctx, in->n_files);
    if (!d_fo || !d_tb) return TBK_ENOMEM;
    // The caller's host arrays may be transient.  Small tables are staged through the context's pinned block, so the
    // upload is a true asynchronous copy and needs no wait; large ones go straight from the caller's memory and are
    // waited for.
    const size_t fo_bytes = (size_t)(in->n_files + 1) * 4, tb_bytes = in->n_files;
    if (fo_bytes + tb_bytes <= 4096 * sizeof(uint64_t)) {
      char* stage = (char*)tbk_stage_acquire(ctx);
      memcpy(stage, in->file_off, fo_bytes);
      TBK_HIP(hipMemcpyAsync(d_fo, stage, fo_bytes, hipMemcpyHostToDevice, ctx->stream));
      if (in->tbmerged) {
        memcpy(stage + fo_bytes, in->tbmerged, tb_bytes);
        TBK_HIP(hipMemcpyAsync(d_tb, stage + fo_bytes, tb_bytes, hipMemcpyHostToDevice, ctx->stream));
      } else {
        TBK_HIP(hipMemsetAsync(d_tb, 0, tb_bytes, ctx->stream));
      }
      tbk_stage_release(ctx);
    } else {
      TBK_HIP(hipMemcpyAsync(d_fo, in->file_off, fo_bytes, hipMemcpyHostToDevice, ctx->stream));
      if (in->tbmerged) {
        TBK_HIP(hipMemcpyAsync(d_tb, in->tbmerged, tb_bytes, hipMemcpyHostToDevice, ctx->stream));
      } else {
        TBK_HIP(hipMemsetAsync(d_tb, 0, tb_bytes, ctx->stream));
      }
      TBK_HIP(hipStreamSynchronize(ctx->stream));
    }
    I.file_off = d_fo;
    I.tbm = d_tb;
  }
  I.tid = in->tid;
  I.pos = in->pos;
  I.flag = in->flag;
  I.mapq = in->mapq;
  I.strand = in->strand;
  I.nh = in->nh;
  I.cig_off = in->cig_off;
  I.cig = in->cig;
  I.yc_in = in->yc_in;
  I.yx_in = in->yx_in;
  I.yd_in = in->yd_in;
  I.md_off = in->md_off;
  I.md = in->md;
  I.md_has = in->md_has;
  I.qh = in->qname_hash;
  I.qn_off = in->qname_off;
  I.qn = in->qname;
  I.qh_mask = ~0ull;
  I.qh_mask = ctx->dbg.qhash_mask;
  I.prio_hi = in->prio_hi;
  I.prio_lo = in->prio_lo;
  ColOpt O;
  O.strategy = o->strategy;
  O.max_nh = o->max_nh;
  O.min_qual = o->min_qual;
  O.keep_supp = o->keep_supplementary;
  O.keep_sec = o->keep_secondary;
  O.collapse_same = o->collapse_same;
  O.store_frac = o->store_frac;
  O.hash_mask = ctx->dbg.hash_mask;

  // per-record arrays of the key pass / scan / compaction (the raw window path never touches them: allocated when the tile takes
  // the general front end)
  uint64_t *khi = nullptr, *klo = nullptr;
  int32_t* kend = nullptr;
  uint8_t* kflags = nullptr;
  uint16_t* fidx = nullptr;
  int32_t* effend = nullptr;
  uint32_t* ceff = nullptr;  // (window path on compacted records, allocated below)
  SortBufs sb{};
  auto front_arrays = [&]() -> bool {
    if (sb.val) return true;
    khi = ws_alloc<uint64_t>(ctx, n);
    klo = ws_alloc<uint64_t>(ctx, n);
    kend = ws_alloc<int32_t>(ctx, n);
    kflags = ws_alloc<uint8_t>(ctx, n);
    fidx = ws_alloc<uint16_t>(ctx, n);
    effend = ws_alloc<int32_t>(ctx, n);
    sb.hi = ws_alloc<uint64_t>(ctx, n);
    sb.lo = ws_alloc<uint64_t>(ctx, n);
    sb.val = ws_alloc<uint32_t>(ctx, n);
    sb.hi2 = khi;  // the unsorted key arrays are dead after compaction: reuse them as the ping-pong side
    sb.lo2 = klo;
    sb.val2 = nullptr;
    return sb.val && effend;
  };
  // per-record arrays of the sort path only (the window path never touches them: allocated when the tile takes the sort path)
  uint8_t* flags = nullptr;
  uint32_t *ghead = nullptr, *gex = nullptr, *sgid = nullptr;
  auto sort_path_arrays = [&]() -> bool {
    if (sgid) return true;
    sb.val2 = ws_alloc<uint32_t>(ctx, n);
    flags = ws_alloc<uint8_t>(ctx, n);
    ghead = ws_alloc<uint32_t>(ctx, n);
    gex = ws_alloc<uint32_t>(ctx, n);
    sgid = ws_alloc<uint32_t>(ctx, n);
    return sgid != nullptr;
  };
  uint32_t* head_off = ws_alloc<uint32_t>(ctx, (size_t)in->n_files + 1);
  uint32_t* run_off = ws_alloc<uint32_t>(ctx, (size_t)in->n_files + 1);
  if (!head_off || !run_off) return TBK_ENOMEM;
  // ceil(log2(files)) merge rounds + one local pass against ~12 radix passes.  Measured on MI355X: 2 files x 1 M
  // 0.17 ms vs 0.47 ms; 16 files x 0.5 M: whole step 3.90 vs 4.13 ms; 64 files x 0.25 M (6 rounds): 8.58 vs 8.67 ms —
  // level there, so more files than that take the radix sort.
  bool use_runs = in->n_files <= 64;
  uint32_t runs_min = 32768;          // below this the tile is launch-bound either way; keep the one code path
  if (ctx->dbg.sort) {  // test hook: sort=radix / runs force one path whatever the shape
    use_runs = ctx->dbg.sort != 1 && (use_runs || ctx->dbg.sort == 2);
    if (ctx->dbg.sort == 2) runs_min = 0;
  }

  // The window path (wgroup.hip) goes from the runs to the groups without sorting the records; it covers plain BAM inputs
  // with integral YC (the ordered / carried-tag cases keep the sort path, which has the per-record order they need).
  // Small tiles of a few files are launch-bound either way and the lean run-sort path needs no read-back: it keeps them.
  bool use_win = tbk_window_supported(in->n_files) && !O.store_frac && !O.collapse_same && (n >= (8u << 20) || (in->n_files > 64 && n >= 65536));
  if (in->tbmerged)
    for (uint32_t f = 0; f < in->n_files; ++f) use_win = use_win && in->tbmerged[f] == 0;
  if (ctx->dbg.path) {  // test hook: path=sort keeps the sort path, path=window takes the window path whatever the size
    if (ctx->dbg.path == 1) use_win = false;
    if (ctx->dbg.path == 2)
      use_win = tbk_window_supported(in->n_files) && !O.store_frac && !O.collapse_same && [&] {
        bool ok = true;
        if (in->tbmerged)
          for (uint32_t f = 0; f < in->n_files; ++f) ok = ok && in->tbmerged[f] == 0;
        return ok;
      }();
  }
  // Group partials of other ranks (multi-GPU owner side): every "file" is the run of one source rank, all flagged
  // TieBrush-merged, with explicit priorities — the window path in its PART form (wgroup.hip) is the reduce-by-key.  Small tiles
  // and anything that form cannot hold (TBK_DERR_FRACTIONAL) keep the sort path, which treats them as TieBrush-merged inputs.
  bool part = false;
  if (in->tbmerged && in->prio_hi && in->prio_lo && in->yc_in && in->yx_in && in->yd_in && in->n_files <= 64 && !O.store_frac &&
      !O.collapse_same && O.strategy != TBK_STRAT_FULL) {
    part = true;
    for (uint32_t f = 0; f < in->n_files; ++f) part = part && in->tbmerged[f] != 0;
    if (ctx->dbg.path == 1) part = false;
    if (part && (n >= 65536 || ctx->dbg.path == 2)) use_win = true;
    else part = false;
  }
  // Raw window path: plain tiles without an explicit merge priority go from the input records straight to the groups — the key
  // pass, the effective-end scan and the compaction are folded into the window kernels (TBK_RAW=0: test hook, keeps them apart)
  bool use_raw = use_win;
  if (ctx->dbg.raw == 0) use_raw = false;
  if (part) use_raw = true;  // (the PART form exists in raw mode only)
  if (n >= (1u << 30) || in->n_cigar_ops >= (1u << 30)) {  // the raw window kernels address a record's fields by 32-bit byte offsets
    use_raw = false;
    if (part) part = false, use_win = false;
  }
  WgOut win_out;
  bool win_done = false;
  uint32_t m = 0, ng = 0;
  GroupAcc G{};
  int32_t* g_yd = nullptr;
  uint32_t* gperm = nullptr;
  uint32_t* ginv = nullptr;
  bool any_tbm = false;
  if (in->tbmerged)
    for (uint32_t f = 0; f < in->n_files; ++f) any_tbm |= in->tbmerged[f] != 0;
  // Every kernel reads the record / group counts from the device scalars (sc[0] = passing records, sc[1] = groups), so
  // the host only needs them where it must size something by them.  "Lean" tiles — plain BAM inputs on the run-sort
  // path, integral YC — need that nowhere: grids and arrays take the upper bound n, nothing is read back until the
  // single synchronisation at the end, and the rare events that want a different path (a bucket too long for the
  // local sort, a key-hash collision) simply restart the tile.
  bool lean = use_runs && !O.store_frac && !any_tbm && n >= runs_min;
  const uint64_t seeds[4] = {TBK_KEY_SEED0, 0xA5A5F00DCAFE1234ull, 0x0123456789ABCDEFull, 0xDEADBEEF0BADF00Dull};
  int attempt = 0;
  for (;;) {
    if (attempt == 4) return TBK_ECOLLISION;
    O.seed = seeds[attempt];
    TBK_HIP(hipMemsetAsync(sc, 0, 16 * sizeof(uint64_t), ctx->stream));
    if (use_raw && use_win) {
      uint32_t eb = 0;
      WgOut wo;
      // (the compaction pass writes the results at their key-order places; col_write_k below only rewrites the members of tie sets)
      WgDirectOut direct;
      direct.cap = out->cap_groups;
      direct.rep = out->rep, direct.yc = out->yc, direct.yx = out->yx;
      direct.g_start = out->g_start, direct.g_end = out->g_end, direct.rep_effend = out->rep_effend, direct.g_key = out->g_key;
      direct.strategy = O.strategy;
      const bool direct_on = out->rep && out->yc && out->yx;
      if (!direct_on) direct.rep = nullptr;
      TBK_TRY(tbk_window_groups(ctx, I, O.strategy, nullptr, nullptr, nullptr, nullptr, n, I.file_off, nullptr, nullptr, out->rec_group != nullptr,
                                O.seed, &wo, &eb, &O, part, &direct));
      if (eb & (TBK_DERR_RAWORDER | TBK_DERR_BIGBUCKET | TBK_DERR_FRACTIONAL)) {  // not this path's kind of input (the general front end
        TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));     // decides what is an error), or a pile-up beyond the
        use_raw = false;                                                            // group table
        if ((eb & (TBK_DERR_BIGBUCKET | TBK_DERR_FRACTIONAL)) || part) use_win = false;
        part = false;
        continue;
      }
      if (eb & TBK_DERR_COLLISION) {  // reseed
        ++attempt;
        continue;
      }
      if (eb) return tbk_derr_to_status(ctx, eb);
      m = (uint32_t)ctx->h_scalars[0];
      out->n_passed = m;
      if (m == 0) return 0;
      ng = wo.ng;
      if (ng > out->cap_groups) {
        out->n_groups = ng;
        return TBK_E2BIG;
      }
      G.yc = wo.yc;
      G.ns = wo.ns;
      G.yxin = wo.yxin;
      G.ydin = wo.ydin;
      G.rep = wo.rep;
      G.first = wo.first;
      G.tie = wo.tie;
      g_yd = ws_alloc<int32_t>(ctx, ng);
      gperm = ws_alloc<uint32_t>(ctx, ng);
      ginv = ws_alloc<uint32_t>(ctx, ng);
      if (!ginv) return TBK_ENOMEM;
      // (items placed by list leave their distances with the items, yd_lgather_k: nothing accumulates in g_yd then)
      if (!(!wo.pgrp || tbk_yd_by_list(ctx, I.k))) TBK_HIP(hipMemsetAsync(g_yd, 0, (size_t)ng * 4, ctx->stream));
      const uint64_t* png = sc + 1;  // (tbk_window_groups left the group count there)
      TBK_LAUNCH(ctx, "col_tie_sort", col_tie_sort_k, cdiv(ng, B), B, 0, I, O.strategy, png, wo.gmem, G, gperm, ginv);
      // (effend == nullptr: the effective end of the representative — or the low word of its explicit priority — rides in the
      // high word of G.rep)
      if (direct_on)
        TBK_LAUNCH(ctx, "col_write", col_write_k<true>, cdiv(ng, B), B, 0, png, gperm, G, wo.ghi, wo.glo, out->cap_groups, out->rep, out->yc,
                   out->yx, out->g_start, out->g_end, (const int32_t*)nullptr, out->rep_effend, out->g_key, O.strategy);
      else
        TBK_LAUNCH(ctx, "col_write", col_write_k<false>, cdiv(ng, B), B, 0, png, gperm, G, wo.ghi, wo.glo, out->cap_groups, out->rep, out->yc,
                   out->yx, out->g_start, out->g_end, (const int32_t*)nullptr, out->rep_effend, out->g_key, O.strategy);
      if (out->rec_group) TBK_LAUNCH(ctx, "col_recgroup", col_recgroup_w_k, cdiv(n, B), B, 0, n, wo.rec_sg, ginv, out->rec_group);
      TBK_TRY(tbk_sync_err(ctx, &eb));
      if (eb & TBK_DERR_COLLISION) {  // (the verification pass, wg_finish_raw_k): reseed
        ++attempt;
        continue;
      }
      if (eb) return tbk_derr_to_status(ctx, eb);
      win_out = wo;
      win_done = true;
      if (attempt > 0) {
        char b[96];
        snprintf(b, sizeof(b), "info: key-hash collision, reseeded %d time(s)", attempt);
        ctx->last_error = b;
      }
      break;
    }
    if (!front_arrays()) return TBK_ENOMEM;
    if (use_win && !ceff) {
      ceff = ws_alloc<uint32_t>(ctx, n);
      if (!ceff) return TBK_ENOMEM;
    }
    SortBufs s2 = sb;
    TBK_LAUNCH(ctx, "col_keys", col_keys_k, cdiv(n, B), B, 0, I, O, khi, klo, kend, kflags, fidx, ctx->d_err);
    {
      EffLoad ld{khi, kend, kflags};
      EffStore st{khi, klo, kflags, effend, s2.hi, s2.lo, s2.val, sc + 0, n, ctx->d_err, fidx, head_off, use_win ? ceff : nullptr, I.prio_hi};
      EffKey ident{0u, 0u, INT32_MIN, 0u};
      TBK_TRY((scan_op_run<EffKey, EffOp, EffLoad, EffStore>(ctx, "col_effkey_scan", n, ld, st, EffOp{}, ident, true)));
    }
    if (use_runs || use_win) TBK_LAUNCH(ctx, "col_runs", col_runs_k, cdiv(I.k + 1, B), B, 0, I.k, I.file_off, head_off, sc + 0, run_off);
    uint32_t eb = 0;
    if (use_win) {
      // ---- window path (wgroup.hip): groups straight from the position-sorted runs, no record sort ----
      TBK_TRY(tbk_sync_err(ctx, &eb));
      if (eb) return tbk_derr_to_status(ctx, eb);
      m = (uint32_t)ctx->h_scalars[0];
      out->n_passed = m;
      if (m == 0) return 0;
      WgOut wo;
      TBK_TRY(tbk_window_groups(ctx, I, O.strategy, s2.hi, s2.lo, s2.val, ceff, m, run_off, khi, klo, out->rec_group != nullptr, O.seed, &wo, &eb));
      if (eb & TBK_DERR_BIGBUCKET) {  // a pile-up with more distinct alignments than the LDS table holds: sort path
        TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
        use_win = false;
        continue;
      }
      if (eb & TBK_DERR_COLLISION) {  // reseed
        ++attempt;
        continue;
      }
      if (eb) return tbk_derr_to_status(ctx, eb);
      ng = wo.ng;
      if (ng > out->cap_groups) {
        out->n_groups = ng;
        return TBK_E2BIG;
      }
      G.yc = wo.yc;
      G.ns = wo.ns;
      G.yxin = wo.yxin;
      G.ydin = wo.ydin;
      G.rep = wo.rep;
      G.first = wo.first;
      G.tie = wo.tie;
      g_yd = ws_alloc<int32_t>(ctx, ng);
      gperm = ws_alloc<uint32_t>(ctx, ng);
      ginv = ws_alloc<uint32_t>(ctx, ng);
      if (!ginv) return TBK_ENOMEM;
      // (items placed by list leave their distances with the items, yd_lgather_k: nothing accumulates in g_yd then)
      if (!(!wo.pgrp || tbk_yd_by_list(ctx, I.k))) TBK_HIP(hipMemsetAsync(g_yd, 0, (size_t)ng * 4, ctx->stream));
      const uint64_t* png = sc + 1;  // (tbk_window_groups left the group count there)
      TBK_LAUNCH(ctx, "col_tie_sort", col_tie_sort_k, cdiv(ng, B), B, 0, I, O.strategy, png, wo.gmem, G, gperm, ginv);
      TBK_LAUNCH(ctx, "col_write", col_write_k<false>, cdiv(ng, B), B, 0, png, gperm, G, wo.ghi, wo.glo, out->cap_groups, out->rep, out->yc,
                 out->yx, out->g_start, out->g_end, effend, out->rep_effend, out->g_key, O.strategy);
      if (out->rec_group) TBK_LAUNCH(ctx, "col_recgroup", col_recgroup_w_k, cdiv(n, B), B, 0, n, wo.rec_sg, ginv, out->rec_group);
      TBK_TRY(tbk_sync_err(ctx, &eb));
      if (eb & TBK_DERR_COLLISION) {  // (the verification pass of the window path, wg_finish_k): reseed
        ++attempt;
        continue;
      }
      if (eb) return tbk_derr_to_status(ctx, eb);
      win_out = wo;
      win_done = true;
      if (attempt > 0) {
        char b[96];
        snprintf(b, sizeof(b), "info: key-hash collision, reseeded %d time(s)", attempt);
        ctx->last_error = b;
      }
      break;
    }
    if (!sort_path_arrays()) return TBK_ENOMEM;
    s2.val2 = sb.val2;
    uint32_t m_hi = n, ng_hi = n;  // what sizes grids and arrays: the counts themselves, or their upper bound
    if (!lean) {
      TBK_TRY(tbk_sync_err(ctx, &eb));
      if (eb) return tbk_derr_to_status(ctx, eb);
      m = (uint32_t)ctx->h_scalars[0];
      out->n_passed = m;
      if (m == 0) return 0;
      m_hi = m;
    }
    // The files are position-sorted runs (verified by the scan above): merge them and order each (tid,start) bucket
    // locally (msort.hip).  Many files, a small tile, or a bucket longer than the local window take the radix sort.
    bool runs_now = use_runs && (lean || m >= runs_min);
    for (;;) {
      if (runs_now)
        TBK_TRY(tbk_sort_runs(ctx, &s2, m_hi, run_off, I.k, ctx->d_err, (uint32_t*)(sc + 12)));  // sc[12]: zeroed with the counters above
      else
        TBK_TRY(tbk_radix_sort128(ctx, &s2, m));
      TBK_LAUNCH(ctx, "col_heads", col_heads_k, cdiv(m_hi, B), B, 0, I, O.strategy, sc + 0, m_hi, s2.hi, s2.lo, s2.val, fidx, flags,
                 ghead, ctx->d_err);
      TBK_TRY(tbk_exscan_u32(ctx, ghead, gex, m_hi, sc + 1));
      if (lean) break;
      TBK_TRY(tbk_sync_err(ctx, &eb));
      if (runs_now && (eb & TBK_DERR_BIGBUCKET)) {  // redo on the merged (phase-A) order, which the *2 side still holds
        std::swap(s2.hi, s2.hi2);
        std::swap(s2.lo, s2.lo2);
        std::swap(s2.val, s2.val2);
        TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
        runs_now = false;
        continue;
      }
      break;
    }
    if (!lean) {
      if (eb & TBK_DERR_COLLISION) {  // reseed
        ++attempt;
        continue;
      }
      if (eb) return tbk_derr_to_status(ctx, eb);
      ng = (uint32_t)ctx->h_scalars[1];
      if (ng > out->cap_groups) {
        out->n_groups = ng;
        return TBK_E2BIG;
      }
      ng_hi = ng;
    }
    const uint64_t *pm = sc + 0, *png = sc + 1;
    G.yc = ws_alloc<double>(ctx, ng_hi);
    G.ns = ws_alloc<uint32_t>(ctx, ng_hi);
    G.yxin = ws_alloc<long long>(ctx, ng_hi);
    G.ydin = ws_alloc<long long>(ctx, ng_hi);
    G.rep = ws_alloc<unsigned long long>(ctx, ng_hi);
    G.first = ws_alloc<uint32_t>(ctx, ng_hi);
    G.tie = ws_alloc<uint8_t>(ctx, ng_hi);
    g_yd = ws_alloc<int32_t>(ctx, ng_hi);
    gperm = ws_alloc<uint32_t>(ctx, ng_hi);
    ginv = ws_alloc<uint32_t>(ctx, ng_hi);
    if (!ginv) return TBK_ENOMEM;
    TBK_LAUNCH(ctx, "col_init_groups", col_init_groups_k, cdiv(ng_hi, B), B, 0, png, G, g_yd);
    TBK_LAUNCH(ctx, "col_reduce", col_reduce_k, cdiv(m_hi, B), B, 0, I, O, pm, s2.val, flags, gex, fidx, effend, G, sgid, ctx->d_err);
    if (I.prio_hi && I.prio_lo) TBK_LAUNCH(ctx, "col_rep_prio", col_rep_prio_k, cdiv(ng_hi, B), B, 0, I, png, pm, s2.val, G);
    if (O.collapse_same) TBK_LAUNCH(ctx, "col_same", col_same_k, cdiv(m_hi, B), B, 0, I, O, pm, s2.val, flags, sgid, fidx, G);
    TBK_LAUNCH(ctx, "col_tie_sort", col_tie_sort_k, cdiv(ng_hi, B), B, 0, I, O.strategy, png, s2.val, G, gperm, ginv);

    // ---- ordered YC when a fractional term can occur (only --store-frac and TieBrush-merged inputs can bring one;
    // such tiles are never lean, so m and ng are known here) ----
    if (O.store_frac || any_tbm) {
      TBK_TRY(tbk_sync_err(ctx, &eb));
      const bool need_ordered = O.store_frac || (eb & TBK_DERR_FRACTIONAL);
      eb &= ~TBK_DERR_FRACTIONAL;
      if (eb) return tbk_derr_to_status(ctx, eb);
      if (need_ordered) {
        TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
        SortBufs ob;
        ob.hi = ws_alloc<uint64_t>(ctx, m);
        ob.lo = ws_alloc<uint64_t>(ctx, m);
        ob.val = ws_alloc<uint32_t>(ctx, m);
        ob.hi2 = ws_alloc<uint64_t>(ctx, m);
        ob.lo2 = ws_alloc<uint64_t>(ctx, m);
        ob.val2 = ws_alloc<uint32_t>(ctx, m);
        uint8_t* fh_by_rec = ws_alloc<uint8_t>(ctx, n);
        if (!ob.val2 || !fh_by_rec) return TBK_ENOMEM;
        TBK_LAUNCH(ctx, "ord_fill", ord_fill_k, cdiv(m, B), B, 0, m, s2.val, sgid, effend, I.prio_hi, flags, ob.hi, ob.lo, ob.val, fh_by_rec);
        TBK_TRY(tbk_radix_sort128(ctx, &ob, m));
        TBK_LAUNCH(ctx, "ord_sum", ord_sum_k, cdiv(ng, 64), 64, 0, I, O, ng, m, ob.val, fidx, fh_by_rec, G);
      }
    }
    TBK_LAUNCH(ctx, "col_write", col_write_k<false>, cdiv(ng_hi, B), B, 0, png, gperm, G, s2.hi, s2.lo, out->cap_groups, out->rep, out->yc,
               out->yx, out->g_start, out->g_end, effend, out->rep_effend, out->g_key, O.strategy);
    if (out->rec_group) TBK_LAUNCH(ctx, "col_recgroup", col_recgroup_k, cdiv(m_hi, B), B, 0, pm, s2.val, sgid, ginv, out->rec_group);
    TBK_TRY(tbk_sync_err(ctx, &eb));
    if (lean) {
      if (eb & TBK_DERR_BIGBUCKET) {  // same keys again, on the radix path (the counts are read back there)
        use_runs = false;
        lean = false;
        continue;
      }
      if (eb & TBK_DERR_COLLISION) {  // reseed
        ++attempt;
        continue;
      }
      m = (uint32_t)ctx->h_scalars[0];
      ng = (uint32_t)ctx->h_scalars[1];
      out->n_passed = m;
    }
    if (eb & ~TBK_DERR_FRACTIONAL) return tbk_derr_to_status(ctx, eb);
    sb = s2;
    if (attempt > 0) {
      char b[96];
      snprintf(b, sizeof(b), "info: key-hash collision, reseeded %d time(s)", attempt);
      ctx->last_error = b;
    }
    break;
  }
  out->n_groups = ng;
  if (m == 0) return 0;
  if (ng > out->cap_groups) return TBK_E2BIG;
  TBK_TRY(tbk_check_launch(ctx, "collapse"));
  // ---- YD stage: handed to the caller (tbk_api) as a job — run inline or deferred on the side context ----
  YdJob* job = new YdJob();
  job->I = I;
  job->m = m;
  job->ng = ng;
  job->cap = out->cap_groups;
  if (win_done) {
    job->win = true;
    job->np = win_out.np;
    job->pfile = win_out.pfile;
    job->pgrp = win_out.pgrp;
    job->gpoff = win_out.gpoff;
    job->gfmask = win_out.gfmask;
  }
  job->val = sb.val;
  job->flags = flags;
  job->fidx = fidx;
  job->sgid = sgid;
  job->ginv = ginv;
  job->gperm = gperm;
  job->G = G;
  job->shi = win_done ? win_out.ghi : sb.hi;
  job->slo = win_done ? win_out.glo : sb.lo;
  job->g_yd = g_yd;
  job->out_yd = out->yd;
  ctx->yd_job = job;
  return 0;
}

// ---- tiebrush -> tiecov device chain -----------------------------------------------------------------------
namespace {
__global__ void g2c_count_k(uint32_t ng, const uint32_t* __restrict__ rep, const uint32_t* __restrict__ cig_off, uint32_t* __restrict__ cnt,
                            uint32_t* __restrict__ cfirst) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  const uint32_t c0 = cig_off[rep[o]];
  cnt[o] = cig_off[rep[o] + 1] - c0;
  cfirst[o] = c0;  // (the gather pass reads the CIGAR range from here: one scattered access fewer per representative)
}
// With tbk_groups_out.g_key: place, strand and — for the two shapes the key can describe — the CIGAR itself come from the key; only
// the other groups' representatives are fetched (a few per cent of an RNA-seq sample; every fetch is a scattered access).
__device__ __forceinline__ uint32_t g2c_key_ops(uint64_t k1) {  // CIGAR words the shape stands for (0: fetch the representative)
  const uint32_t shape = (uint32_t)k1;
  return shape == 0x80000000u ? 1u : ((shape >> 30) == 3u ? 3u : 0u);
}
__global__ void g2c_count_key_k(uint32_t ng, const uint64_t* __restrict__ key, const uint32_t* __restrict__ rep, const uint32_t* __restrict__ cig_off,
                                uint32_t* __restrict__ cnt, uint32_t* __restrict__ cfirst) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  uint32_t n = g2c_key_ops(key[2 * (size_t)o + 1]);
  if (n == 0) {
    const uint32_t c0 = cig_off[rep[o]];
    n = cig_off[rep[o] + 1] - c0;
    cfirst[o] = c0;
  }
  cnt[o] = n;
}
struct G2cPrep {  // the first pass of tbk_coverage_tile, per view record (cov.hip: cov_prep_k) — and its three scalars
  int32_t *start, *end, *yi;
  uint32_t *jcnt, *ridx;
  unsigned long long* sums;  // [0] M bases, [1] sum |YC|, [2] error bits, [3] junction items (sum of jcnt)
};
__global__ __launch_bounds__(256) void g2c_gather_key_k(uint32_t ng, const uint64_t* __restrict__ key, const double* __restrict__ yc,
                                                        const int64_t* __restrict__ yx, const uint32_t* __restrict__ cfirst,
                                                        const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cig,
                                                        const uint32_t* __restrict__ ooff, uint32_t total, int32_t* __restrict__ o_tid,
                                                        int32_t* __restrict__ o_pos, uint8_t* __restrict__ o_strand, double* __restrict__ o_yc,
                                                        int64_t* __restrict__ o_yx, uint32_t* __restrict__ o_cig_off, uint32_t* __restrict__ o_cig,
                                                        G2cPrep P) {
  unsigned long long mb = 0, ay = 0, nji = 0;
  uint32_t eb = 0;
  for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < ng; o += gridDim.x * blockDim.x) {
    const uint64_t k0 = key[2 * (size_t)o], k1 = key[2 * (size_t)o + 1];
    const int32_t tidv = (int32_t)(uint32_t)(k0 >> 33) - 1, posv = (int32_t)(uint32_t)((k0 >> 2) & 0x7FFFFFFFull) - 1;
    o_tid[o] = tidv;
    o_pos[o] = posv;
    const uint32_t sc = (uint32_t)k0 & 3u;
    o_strand[o] = sc == 0u ? (uint8_t)'+' : (sc == 1u ? (uint8_t)'-' : (uint8_t)'.');
    const double y0 = (double)(float)yc[o];  // the YC:f tag round trip (bam_aux_update_float, tiebrush.cpp:509)
    o_yc[o] = y0;
    o_yx[o] = yx[o];
    const uint32_t d = ooff[o];
    o_cig_off[o] = d;
    if (o + 1 == ng) o_cig_off[ng] = total;
    const uint32_t shape = (uint32_t)k1, span = (uint32_t)(k1 >> 32);
    int l, nex = 0;
    if (shape == 0x80000000u) {
      o_cig[d] = (span << 4) | C_M;
      l = (int)span;
      nex = 1;
      mb += span;
    } else if ((shape >> 30) == 3u) {
      const uint32_t a = (shape >> 20) & 0x3FFu, g = shape & 0xFFFFFu;
      o_cig[d] = (a << 4) | C_M;
      o_cig[d + 1] = (g << 4) | C_N;
      o_cig[d + 2] = ((span - a - g) << 4) | C_M;
      l = (int)span;
      nex = 2;
      mb += span - g;
    } else {
      const uint32_t c0 = cfirst[o], n = cnt[o];
      l = walk_exons(posv, cig + c0, n, [](int, int) {}, &nex);
      for (uint32_t k = 0; k < n; ++k) {
        const uint32_t w = cig[c0 + k];
        o_cig[d + k] = w;
        const uint32_t op = cig_op(w);
        if (op == C_M)
          mb += cig_len(w);
        else if (op != C_I && op != C_D && op != C_N && op != C_S)
          eb |= TBK_DERR_FATALOP;
      }
      if (n >= 256) eb |= TBK_DERR_NCIGAR;
    }
    double y = y0;
    if (!(y == rint(y)) || !(fabs(y) < 1073741824.0)) {
      eb |= TBK_DERR_FRACTIONAL;
      y = 0.0;
    } else {
      ay += (unsigned long long)fabs(y);
    }
    P.ridx[o] = o;
    P.yi[o] = (int32_t)y;
    P.start[o] = posv + 1;
    P.end[o] = posv + l;
    P.jcnt[o] = (uint32_t)(nex - 1);
    nji += (unsigned long long)(nex - 1);
  }
  __shared__ unsigned long long red_mb[4], red_ay[4], red_nj[4];
  __shared__ uint32_t red_e[4];
  mb = wave_sum(mb);
  ay = wave_sum(ay);
  nji = wave_sum(nji);
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) eb |= __shfl_xor(eb, dd, 64);
  if (lane_id() == 0) {
    red_mb[threadIdx.x >> 6] = mb;
    red_ay[threadIdx.x >> 6] = ay;
    red_nj[threadIdx.x >> 6] = nji;
    red_e[threadIdx.x >> 6] = eb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 4; ++q) {
      mb += red_mb[q];
      ay += red_ay[q];
      nji += red_nj[q];
      eb |= red_e[q];
    }
    if (mb) atomicAdd(&P.sums[0], mb);
    if (ay) atomicAdd(&P.sums[1], ay);
    if (eb) atomicOr(&P.sums[2], (unsigned long long)eb);
    if (nji) atomicAdd(&P.sums[3], nji);
  }
}
__global__ void g2c_gather_k(uint32_t ng, const uint32_t* __restrict__ rep, const double* __restrict__ yc, const int64_t* __restrict__ yx,
                             const int32_t* __restrict__ tid, const int32_t* __restrict__ pos, const uint8_t* __restrict__ strand,
                             const uint32_t* __restrict__ cfirst, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cig,
                             const uint32_t* __restrict__ ooff, uint32_t total, int32_t* __restrict__ o_tid, int32_t* __restrict__ o_pos,
                             uint8_t* __restrict__ o_strand, double* __restrict__ o_yc, int64_t* __restrict__ o_yx,
                             uint32_t* __restrict__ o_cig_off, uint32_t* __restrict__ o_cig) {
  uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  uint32_t r = rep[o];
  o_tid[o] = tid[r];
  o_pos[o] = pos[r];
  o_strand[o] = strand[r];
  o_yc[o] = (double)(float)yc[o];  // the YC:f tag round trip (bam_aux_update_float, tiebrush.cpp:509)
  o_yx[o] = yx[o];
  uint32_t d = ooff[o];
  o_cig_off[o] = d;
  if (o + 1 == ng) o_cig_off[ng] = total;
  const uint32_t c0 = cfirst[o], n = cnt[o];
  for (uint32_t k = 0; k < n; ++k) o_cig[d + k] = cig[c0 + k];
}
}  // namespace

extern "C" int tbk_groups_to_cov_in(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* g, tbk_cov_in* view) {
  if (!ctx || !in || !g || !view) return TBK_EINVAL;
  if (in->mem != TBK_MEM_DEVICE || g->mem != TBK_MEM_DEVICE) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  const uint32_t ng = g->n_groups;
  memset(view, 0, sizeof(*view));
  view->mem = TBK_MEM_DEVICE;
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  if (ng == 0) return 0;
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)ng * 24 + ((size_t)1 << 20)));
  return tbk_cov_view_build(ctx, in->tid, in->pos, in->strand, in->cig_off, in->cig, g->rep, g->yc, g->yx, ng, view, g->g_key);
}

// the view of ng representatives (rep[o] indexes the given record arrays) in context-owned memory; allocates from the arena as it
// stands (the caller has reserved it) and synchronises the stream
int tbk_cov_view_build(tbk_ctx* ctx, const int32_t* r_tid, const int32_t* r_pos, const uint8_t* r_strand, const uint32_t* r_cig_off,
                       const uint32_t* r_cig, const uint32_t* g_rep, const double* g_yc, const int64_t* g_yx, uint32_t ng, tbk_cov_in* view,
                       const uint64_t* g_key) {
  memset(view, 0, sizeof(*view));
  view->mem = TBK_MEM_DEVICE;
  ctx->view_prep.valid = false;  // (the context's view is about to change)
  if (ng == 0) return 0;
  uint32_t* cnt = ws_alloc<uint32_t>(ctx, ng);
  uint32_t* ooff = ws_alloc<uint32_t>(ctx, ng);
  uint32_t* cfirst = ws_alloc<uint32_t>(ctx, ng);
  if (!cfirst) return TBK_ENOMEM;
  const uint32_t B = 256;
  if (g_key)
    TBK_LAUNCH(ctx, "g2c_count", g2c_count_key_k, cdiv(ng, B), B, 0, ng, g_key, g_rep, r_cig_off, cnt, cfirst);
  else
    TBK_LAUNCH(ctx, "g2c_count", g2c_count_k, cdiv(ng, B), B, 0, ng, g_rep, r_cig_off, cnt, cfirst);
  TBK_TRY(tbk_exscan_u32(ctx, cnt, ooff, ng, ctx->d_scalars + 20));
  TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 20, ctx->d_scalars + 20, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  const uint64_t total = ctx->h_scalars[20];
  if (total >= (1ull << 32)) return TBK_E2BIG;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t need = al((size_t)ng * 4) * 2 + al(ng) + al((size_t)ng * 8) * 2 + al((size_t)(ng + 1) * 4) + al((size_t)total * 4 + 4) +
                (g_key ? al((size_t)ng * 4) * 5 : 0);
  if (need > ctx->d_view_cap) {
    if (ctx->d_view) (void)hipFree(ctx->d_view);
    ctx->d_view = nullptr;
    ctx->d_view_cap = 0;
    size_t cap = need + need / 4;
    TBK_HIP(hipMalloc((void**)&ctx->d_view, cap));
    ctx->d_view_cap = cap;
  }
  char* p = ctx->d_view;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += al(bytes);
    return r;
  };
  int32_t* o_tid = (int32_t*)take((size_t)ng * 4);
  int32_t* o_pos = (int32_t*)take((size_t)ng * 4);
  uint8_t* o_strand = (uint8_t*)take(ng);
  double* o_yc = (double*)take((size_t)ng * 8);
  int64_t* o_yx = (int64_t*)take((size_t)ng * 8);
  uint32_t* o_cig_off = (uint32_t*)take((size_t)(ng + 1) * 4);
  uint32_t* o_cig = (uint32_t*)take((size_t)total * 4 + 4);
  G2cPrep P{};
  if (g_key) {  // ... and what the first pass of tbk_coverage_tile would compute from the view (see TbkCtx::view_prep)
    P.start = (int32_t*)take((size_t)ng * 4);
    P.end = (int32_t*)take((size_t)ng * 4);
    P.yi = (int32_t*)take((size_t)ng * 4);
    P.jcnt = (uint32_t*)take((size_t)ng * 4);
    P.ridx = (uint32_t*)take((size_t)ng * 4);
    P.sums = (unsigned long long*)(ctx->d_scalars + 24);
    TBK_HIP(hipMemsetAsync(P.sums, 0, 4 * sizeof(uint64_t), ctx->stream));
    TBK_LAUNCH(ctx, "g2c_gather", g2c_gather_key_k, (cdiv(ng, B) < 4096u ? cdiv(ng, B) : 4096u), B, 0, ng, g_key, g_yc, g_yx, cfirst, cnt, r_cig, ooff,
               (uint32_t)total, o_tid, o_pos, o_strand, o_yc, o_yx, o_cig_off, o_cig, P);
    TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 24, ctx->d_scalars + 24, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  } else
    TBK_LAUNCH(ctx, "g2c_gather", g2c_gather_k, cdiv(ng, B), B, 0, ng, g_rep, g_yc, g_yx, r_tid, r_pos, r_strand, cfirst, cnt, r_cig, ooff,
               (uint32_t)total, o_tid, o_pos, o_strand, o_yc, o_yx, o_cig_off, o_cig);
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  TBK_TRY(tbk_check_launch(ctx, "groups_to_cov_in"));
  view->n_records = ng;
  view->n_cigar_ops = (uint32_t)total;
  view->tid = o_tid;
  view->pos = o_pos;
  view->flag = nullptr;  // every representative counts: tbk_coverage_tile skips its validity pass
  view->cig_off = o_cig_off;
  view->cig = o_cig;
  view->yc = o_yc;
  view->strand = o_strand;
  view->yx = o_yx;
  if (g_key) {
    auto& V = ctx->view_prep;
    V.cig = o_cig;
    V.n = ng;
    V.start = P.start;
    V.end = P.end;
    V.yi = P.yi;
    V.jcnt = P.jcnt;
    V.ridx = P.ridx;
    V.n_bases = ctx->h_scalars[24];
    V.sum_abs = ctx->h_scalars[25];
    V.err = (uint32_t)ctx->h_scalars[26];
    V.n_junc = ctx->h_scalars[27];
    V.valid = true;
  }
  return 0;
}

// ---- multi-GPU stitch: pack the groups of one rank into the exchange layout ---------------------------------
