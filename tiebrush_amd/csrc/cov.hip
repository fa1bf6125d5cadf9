// cov.hip — tiecov hot path on gfx950: bundle segmentation, CIGAR-walk per-base coverage with
// LDS-binned atomics, run-length writeback, and splice-junction reduction.
//
// Reference semantics (paths relative to /root/reference/src):
//   bundles         tiecov.cpp:443-481   new bundle iff tid changes or start > running max(end)
//   addCov          tiecov.cpp:194-223   cov[base] += YC for every M base; D/N advance; I/S ignored;
//                                        any other op is fatal (:219)
//   flushCoverage   tiecov.cpp:226-241   RLE on exact equality, zero runs skipped, runs never
//                                        cross a bundle
//   addJunction     tiecov.cpp:100-112   per bundle sum of YC per (start,end,strand)
//   flushJuncs      tiecov.cpp:114-120   sorted (start,end,strand char), numbered globally
//
// Data layout in HBM: bundles are laid end to end in a "compacted coordinate" space (cpos);
// the space is cut into tiles of COV_W bases.  One workgroup owns one tile: it adds +yc/-yc
// difference marks for every M segment into an LDS array (ds atomics), prefix-sums it in LDS
// and emits only the change points — the per-base depth array never exists in HBM.
// Segments that reach past the tile where their record starts are pre-binned ("spills") by a
// counting sort over tiles.  Integer accumulation is exact in any order; non-integral YC
// values are routed to the ordered double path (cov_tile_k<double>) which adds per base in
// record order exactly as the reference does.
#include <thread>

#include "dev_common.hpp"
#include "scan_op.hpp"
#include "tbk_internal.h"

namespace {

constexpr int COV_W = 8192;  // bases per tile
constexpr int COV_NT = 256;
constexpr int COV_PER = COV_W / COV_NT;  // 32 bases per thread in the scan phase

struct SegMax {
  int32_t mx;
  uint32_t flag;
};
struct SegMaxOp {
  __device__ __forceinline__ SegMax operator()(const SegMax& a, const SegMax& b) const {
    SegMax r;
    r.mx = b.flag ? b.mx : (a.mx > b.mx ? a.mx : b.mx);
    r.flag = a.flag | b.flag;
    return r;
  }
};

struct CovArrays {
  // per compacted (mapped) record j
  uint32_t* ridx;    // original record index
  int32_t* start;    // 1-based
  int32_t* end;      // 1-based inclusive
  int32_t* tid;
  uint32_t* bhead;   // bundle head flag
  uint32_t* bid;     // bundle id
  uint64_t* cs;      // compacted start (0-based cpos)
  uint16_t* pk;      // tile kernel input: compacted start inside its home tile (13 bits) | bundle head << 15
  int32_t* yi;       // tile kernel input: YC as int32 (valid when every YC is integral and sum |YC| < 2^31)
  // per bundle
  int32_t* b_tid;
  int32_t* b_start;
  int32_t* b_end;
  uint32_t* b_span;
  uint64_t* b_off;
};

// ---- C0/C1 -------------------------------------------------------------------------------
__global__ void cov_valid_k(const uint16_t* __restrict__ flag, uint32_t n, uint32_t* __restrict__ valid) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) valid[i] = (flag[i] & 0x4) ? 0u : 1u;
}

// scalars: [0]=n_bases [1]=sum|yc| [2]=n_introns
__global__ void cov_prep_k(uint32_t n, const uint32_t* __restrict__ valid, const uint32_t* __restrict__ vpos,
                           const int32_t* __restrict__ tid, const int32_t* __restrict__ pos,
                           const uint32_t* __restrict__ cig_off, const uint32_t* __restrict__ cig,
                           const double* __restrict__ yc, int check_ops, CovArrays A, uint32_t* __restrict__ jcnt,
                           uint64_t* __restrict__ scalars, uint32_t* __restrict__ err) {
  uint64_t mb = 0, ay = 0, nji = 0;
  uint32_t e = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (valid && !valid[i]) continue;
    uint32_t j = valid ? vpos[i] : i;
    uint32_t c0 = cig_off[i], c1 = cig_off[i + 1];
    int nex = 0;
    int l = walk_exons(pos[i], cig + c0, c1 - c0, [](int, int) {}, &nex);
    for (uint32_t k = c0; k < c1; ++k) {
      uint32_t op = cig_op(cig[k]);
      if (op == C_M)
        mb += cig_len(cig[k]);
      else if (op != C_I && op != C_D && op != C_N && op != C_S)
        e |= check_ops ? TBK_DERR_FATALOP : 0u;
    }
    if (check_ops && (c1 - c0) >= 256) e |= TBK_DERR_NCIGAR;
    double y = yc ? yc[i] : 1.0;
    if (!(y == rint(y)) || !(fabs(y) < 1073741824.0)) {
      e |= TBK_DERR_FRACTIONAL;
      y = 0.0;
    } else {
      ay += (uint64_t)fabs(y);
    }
    A.ridx[j] = i;
    A.yi[j] = (int32_t)y;  // (only read when every value turned out integral and small)
    A.start[j] = pos[i] + 1;
    A.end[j] = pos[i] + l;
    A.tid[j] = tid[i];
    if (jcnt) {
      jcnt[j] = (uint32_t)(nex - 1);
      nji += (uint64_t)(nex - 1);
    }
  }
  // block-level reduction: the three scalars share one cache line, keep the atomics to one set per block
  __shared__ unsigned long long red_mb[4], red_ay[4], red_nj[4];
  __shared__ uint32_t red_e[4];
  mb = wave_sum(mb);
  ay = wave_sum(ay);
  nji = wave_sum(nji);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) e |= __shfl_xor(e, d, 64);
  uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 0) {
    red_mb[w] = mb;
    red_ay[w] = ay;
    red_nj[w] = nji;
    red_e[w] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t nw = blockDim.x >> 6;
    for (uint32_t k = 1; k < nw; ++k) {
      mb += red_mb[k];
      ay += red_ay[k];
      nji += red_nj[k];
      e |= red_e[k];
    }
    if (mb) atomicAdd((unsigned long long*)&scalars[0], (unsigned long long)mb);
    if (ay) atomicAdd((unsigned long long*)&scalars[1], (unsigned long long)ay);
    if (nji) atomicAdd((unsigned long long*)&scalars[14], (unsigned long long)nji);
    if (e) atomicOr(err, e);
  }
}

// ---- C2: bundle heads via segmented (per tid run) running max of end -----------------------
struct BundleLoad {
  const int32_t* tid;
  const int32_t* end;
  __device__ __forceinline__ SegMax operator()(uint32_t j) const {
    SegMax s;
    s.mx = end[j];
    s.flag = (j == 0 || tid[j] != tid[j - 1]) ? 1u : 0u;
    return s;
  }
};
struct BundleAux {  // (start, tid) of the record
  const int32_t* tid;
  const int32_t* start;
  __device__ __forceinline__ int2 operator()(uint32_t j) const { return make_int2(start[j], tid[j]); }
};
struct BundleHead {
  __device__ __forceinline__ uint32_t operator()(uint32_t, const SegMax& v, const SegMax&, const SegMax& ex, const int2& a) const {
    return (v.flag || a.x > ex.mx) ? 1u : 0u;  // tiecov.cpp:443 (flag: first record of its reference sequence)
  }
};
// One pass gives every record its bundle (scan_two_run numbers the heads) and every bundle its reference sequence, start and
// end: a record closes its bundle when the next one is a head, which its own inclusive maximum already decides.
struct BundleStore {
  CovArrays A;
  uint32_t m;
  uint64_t* nb_out;
  uint32_t* err;
  __device__ __forceinline__ void operator()(uint32_t j, const SegMax& v, const SegMax& inc, const SegMax&, uint32_t head, uint32_t before,
                                             const int2& a) const {
    const int32_t t = a.y, st = a.x;
    if (!v.flag && st < A.start[j - 1]) atomicOr(err, TBK_DERR_UNSORTED);
    const uint32_t b = before + head - 1u;
    A.bhead[j] = head;
    A.bid[j] = b;
    if (head) {
      A.b_tid[b] = t;
      A.b_start[b] = st;
    }
    const bool last = j + 1 == m;
    if (last || A.tid[j + 1] != t || A.start[j + 1] > inc.mx) A.b_end[b] = inc.mx;
    if (last) *nb_out = (uint64_t)b + 1u;
  }
};

// ---- C2 again, in lean passes ----------------------------------------------------------------------------------------------------
// The same heads and numbers as the two-stage scan above (kept for reference and for the sample track), as three streaming passes
// over 4096-record tiles, a thread owning four consecutive records of every row (16-byte loads, thread order = record order):
//   (1) every tile's aggregate of the monoid below, a one-block scan over the 6 k tiles;
//   (2) with the aggregate of the records before it, every record knows whether it opens a bundle: heads counted per tile, scanned;
//   (3) the same walk again, now numbering: bundle of every record, head flags, the bundle table.
// 32 bytes read and 8 written per record instead of the look-back kernel's waits: 0.53 -> 0.3 ms on config 3.
struct CbAgg {
  int32_t first_tid, last_tid;  // last_tid == INT32_MIN: no record
  int32_t mx;                   // maximum end over the trailing records that share last_tid
  uint32_t whole;               // every record shares one tid
};
struct CbOp {
  __device__ __forceinline__ CbAgg operator()(const CbAgg& a, const CbAgg& b) const {
    if (b.last_tid == INT32_MIN) return a;
    if (a.last_tid == INT32_MIN) return b;
    CbAgg r;
    r.first_tid = a.first_tid;
    r.last_tid = b.last_tid;
    const bool joins = b.whole && b.first_tid == a.last_tid;
    r.mx = joins ? (a.mx > b.mx ? a.mx : b.mx) : b.mx;
    r.whole = joins ? a.whole : 0u;
    return r;
  }
};
constexpr uint32_t CB_NT = 256, CB_ROWS = 4, CB_TILE = CB_NT * 4 * CB_ROWS;
__device__ __forceinline__ void cb_load4(const int32_t* __restrict__ a, uint64_t i, uint32_t m, int32_t fill, int32_t v[4]) {
  if (i + 3 < m) {
    const int4 q = *reinterpret_cast<const int4*>(a + i);  // (i is a multiple of 4, the arrays 256-byte aligned)
    v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = i + e < m ? a[i + e] : fill;
  }
}
template <class T, class Op>
__device__ __forceinline__ T wave_incl_scan_op(T v, Op op) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T o = shfl_up_t(v, d);
    if ((int)lane_id() >= d) v = op(o, v);
  }
  return v;
}
// True in every thread of the block that finishes last; `done` starts at zero.  What the blocks hand to the last one travels without
// fences: thread 0 writes its block's result with agent-scope stores (cb_put), waits for them to be acknowledged and counts the block;
// the last block reads the results with agent-scope loads (cb_get).  (A __threadfence per block writes back and invalidates L2: with
// 6 k blocks it made this pass — and the junction kernels beside it — five times slower.)
__device__ __forceinline__ void cb_put(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t cb_get(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint4 cb_get4(const uint4* p) {
  const uint64_t a = cb_get(reinterpret_cast<const uint64_t*>(p)), b = cb_get(reinterpret_cast<const uint64_t*>(p) + 1);
  return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
__device__ __forceinline__ bool cb_last_block(uint32_t* __restrict__ done) {  // (called behind thread 0's cb_put's)
  __shared__ uint32_t s_last;
  if (threadIdx.x == 0) {
    // The block is counted BEHIND its results: they are write-through stores (agent scope), so what orders them before the counter is
    // the wait for their acknowledgement — stated explicitly: a workgroup-scope release fence compiles to no wait at all here (the
    // counter could pass the results, and the last block scan entries that were not there yet), an agent-scope release to an L2
    // write-back per block (buffer_wbl2: 0.37 -> 0.69 ms for the two passes, and the junction kernels beside them as much slower).
    __builtin_amdgcn_s_waitcnt(TBK_WAIT_VMCNT0);  // vmcnt(0): on gfx9 stores count there too
    __asm__ volatile("" ::: "memory");
    s_last = atomicAdd(done, 1u) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  return s_last != 0u;
}
// The tile aggregates -> for every tile the aggregate of the tiles before it, by ONE block of NT threads: a slice of the tiles per
// thread with eight loads in flight at a time, the slices' aggregates by wave scans and one fold of the wave totals.  It runs in the
// block of cb_agg_k that finishes last (cb_last_block): as a kernel of its own between the passes — one block — it waited for a CU
// with room while the junction branch's grids filled the GPU (100-200 us of a 1.4 ms coverage call, whatever the streams' priorities).
template <uint32_t NT>
__device__ __forceinline__ void cb_spine_block(uint4* __restrict__ part, uint32_t ntiles, CbAgg* wl /* [NT / 64] */) {
  const CbOp op{};
  const CbAgg none{0, INT32_MIN, INT32_MIN, 1u};
  auto un = [](const uint4& v) { return CbAgg{(int32_t)v.x, (int32_t)v.y, (int32_t)v.z, v.w}; };
  auto pk = [](const CbAgg& a) { return make_uint4((uint32_t)a.first_tid, (uint32_t)a.last_tid, (uint32_t)a.mx, a.whole); };
  const uint4 none4 = pk(none);
  const uint32_t per = (ntiles + NT - 1u) / NT, i0 = threadIdx.x * per, i1 = i0 + per < ntiles ? i0 + per : ntiles;
  CbAgg a = none;
  for (uint32_t q0 = i0; q0 < i1; q0 += 8u) {
    uint4 v8[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) v8[u] = q0 + u < i1 ? cb_get4(part + q0 + u) : none4;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) a = op(a, un(v8[u]));
  }
  const CbAgg inc = wave_incl_scan_op(a, op);
  if (lane_id() == 63) wl[threadIdx.x >> 6] = inc;
  CbAgg run = shfl_up_t(inc, 1);
  __syncthreads();  // (every slice has been read: the writes below may begin)
  {
    CbAgg acc = none;
    const uint32_t wv = threadIdx.x >> 6;
    for (uint32_t q = 0; q < wv; ++q) acc = op(acc, wl[q]);
    run = lane_id() == 0 ? acc : op(acc, run);
  }
  for (uint32_t q0 = i0; q0 < i1; q0 += 8u) {
    uint4 v8[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) v8[u] = q0 + u < i1 ? cb_get4(part + q0 + u) : none4;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
      if (q0 + u < i1) part[q0 + u] = pk(run);
      run = op(run, un(v8[u]));
    }
  }
}
__global__ __launch_bounds__(CB_NT) void cb_agg_k(uint32_t m, const int32_t* __restrict__ tid, const int32_t* __restrict__ end, uint4* __restrict__ part,
                                                  uint32_t* __restrict__ done) {
  // (the four rows' loads are issued together and their wave scans run side by side: one barrier per block, not two per row)
  constexpr uint32_t NWV = CB_NT / 64;
  __shared__ CbAgg wl[CB_ROWS * NWV];
  const CbOp op{};
  const CbAgg none{0, INT32_MIN, INT32_MIN, 1u};
  int32_t t4[CB_ROWS][4], e4[CB_ROWS][4];
#pragma unroll
  for (uint32_t r = 0; r < CB_ROWS; ++r) {
    const uint64_t i = (uint64_t)blockIdx.x * CB_TILE + ((uint64_t)r * CB_NT + threadIdx.x) * 4u;
    cb_load4(tid, i, m, 0, t4[r]);
    cb_load4(end, i, m, 0, e4[r]);
  }
#pragma unroll
  for (uint32_t r = 0; r < CB_ROWS; ++r) {
    const uint64_t i = (uint64_t)blockIdx.x * CB_TILE + ((uint64_t)r * CB_NT + threadIdx.x) * 4u;
    CbAgg a = none;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < m) a = op(a, CbAgg{t4[r][e], t4[r][e], e4[r][e], 1u});
    a = wave_incl_scan_op(a, op);
    if (lane_id() == 63) wl[r * NWV + (threadIdx.x >> 6)] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    CbAgg run = none;
#pragma unroll
    for (uint32_t q = 0; q < CB_ROWS * NWV; ++q) run = op(run, wl[q]);
    uint64_t* o = reinterpret_cast<uint64_t*>(part + blockIdx.x);
    cb_put(o, (uint64_t)(uint32_t)run.first_tid | ((uint64_t)(uint32_t)run.last_tid << 32));
    cb_put(o + 1, (uint64_t)(uint32_t)run.mx | ((uint64_t)run.whole << 32));
  }
  if (cb_last_block(done)) cb_spine_block<CB_NT>(part, gridDim.x, wl);  // (the last block: every tile's aggregate -> the aggregate before it)
}
// EMIT = false: heads per tile -> hcnt[tile]; EMIT = true: hbase[tile] = heads before the tile; everything is written
template <bool EMIT>
__global__ __launch_bounds__(CB_NT) void cb_heads_k(uint32_t m, CovArrays A, const uint4* __restrict__ part, uint32_t* __restrict__ hcnt,
                                                    const uint32_t* __restrict__ hbase, uint64_t* __restrict__ nb_out, uint32_t* __restrict__ err) {
  __shared__ CbAgg sm[CB_NT / 64];
  __shared__ uint32_t smu[8];
  const CbOp op{};
  const CbAgg none{0, INT32_MIN, INT32_MIN, 1u};
  const uint4 pv = part[blockIdx.x];
  CbAgg run{(int32_t)pv.x, (int32_t)pv.y, (int32_t)pv.z, pv.w};  // the records before the current row
  uint32_t heads_tile = 0, hrun = EMIT ? hbase[blockIdx.x] : 0u;
  bool bad = false;
#pragma unroll
  for (uint32_t r = 0; r < CB_ROWS; ++r) {
    const uint64_t i = (uint64_t)blockIdx.x * CB_TILE + ((uint64_t)r * CB_NT + threadIdx.x) * 4u;
    int32_t t4[4], s4[4], e4[4];
    cb_load4(A.tid, i, m, 0, t4);
    cb_load4(A.start, i, m, 0, s4);
    cb_load4(A.end, i, m, 0, e4);
    CbAgg a = none;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < m) a = op(a, CbAgg{t4[e], t4[e], e4[e], 1u});
    CbAgg tot;
    const CbAgg inc = block_incl_scan_op(a, op, sm, &tot);
    // the aggregate of everything before this thread's first record: run (+) the threads before it in the row
    CbAgg ex = shfl_up_t(inc, 1);
    __shared__ CbAgg wl[CB_NT / 64];
    if (lane_id() == 63) wl[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0)
      ex = none;
    else if (lane_id() == 0)
      ex = wl[(threadIdx.x >> 6) - 1];
    ex = op(run, ex);
    uint32_t hd = 0, nh = 0;
    CbAgg w = ex;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i + e < m) {
        const bool run_head = w.last_tid == INT32_MIN || t4[e] != w.last_tid;
        const bool head = run_head || s4[e] > w.mx;  // tiecov.cpp:443
        hd |= (head ? 1u : 0u) << e;
        nh += head ? 1u : 0u;
        w = op(w, CbAgg{t4[e], t4[e], e4[e], 1u});
      }
    }
    uint32_t rowh;
    uint32_t before = block_excl_sum<uint32_t, CB_NT>(nh, smu, &rowh);
    if (EMIT) {
      before += hrun;
      // the record behind this thread's four (head or not decides whether the fourth closes its bundle)
      const uint64_t nx = i + 4;
      const int32_t nt = nx < m ? A.tid[nx] : 0, ns = nx < m ? A.start[nx] : 0;
      const int32_t pst = i > 0 && i < m ? A.start[i - 1] : 0;  // (start of the record before the four: the order check)
      CbAgg v = ex;
      uint32_t b = before;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint64_t j = i + e;
        if (j < m) {
          const bool head = (hd >> e) & 1u;
          const bool run_head = v.last_tid == INT32_MIN || t4[e] != v.last_tid;
          const int32_t prev_start = e == 0 ? pst : s4[e - 1];
          if (!run_head && s4[e] < prev_start) bad = true;
          b += head ? 1u : 0u;
          const uint32_t bundle = b - 1u;
          A.bhead[j] = head ? 1u : 0u;
          A.bid[j] = bundle;
          if (head) {
            A.b_tid[bundle] = t4[e];
            A.b_start[bundle] = s4[e];
          }
          v = op(v, CbAgg{t4[e], t4[e], e4[e], 1u});
          const bool last = j + 1 == m;
          const int32_t t_next = e < 3 ? t4[e + 1] : nt, s_next = e < 3 ? s4[e + 1] : ns;
          if (last || t_next != t4[e] || s_next > v.mx) A.b_end[bundle] = v.mx;
          if (last) *nb_out = (uint64_t)bundle + 1u;
        }
      }
    }
    heads_tile += rowh;
    hrun += rowh;
    run = op(run, tot);
    __syncthreads();
  }
  if (!EMIT && threadIdx.x == 0) hcnt[blockIdx.x] = heads_tile;
  if (EMIT && bad) atomicOr(err, TBK_DERR_UNSORTED);
}

__global__ void cov_bundle_span_k(uint32_t nb, CovArrays A) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) A.b_span[b] = (uint32_t)(A.b_end[b] - A.b_start[b] + 1);
}

// ---- C5: compacted start + spill counting / filling ----------------------------------------
// pieces of the M segments of record j that do not lie in its home tile
template <class F>
__device__ __forceinline__ void for_each_spill_piece(uint64_t cs, const uint32_t* __restrict__ cig, uint32_t nc, F f) {
  uint64_t home = cs / COV_W;
  uint64_t p = cs;
  for (uint32_t k = 0; k < nc; ++k) {
    uint32_t c = cig[k];
    uint32_t op = cig_op(c), len = cig_len(c);
    if (op == C_M) {
      uint64_t a = p, b = p + len;
      while (a < b) {
        uint64_t t = a / COV_W;
        uint64_t te = (t + 1) * COV_W;
        uint64_t pe = b < te ? b : te;
        if (t != home) f((uint32_t)t, (uint32_t)(a - t * COV_W), (uint32_t)(pe - a));
        a = pe;
      }
      p = b;
    } else if (op == C_D || op == C_N) {
      p += len;
    }
  }
}

// Also fills tile_first[t] = first record whose compacted start lies in tile t or later (records are start-sorted, so the
// home records of tile t are [tile_first[t], tile_first[t+1]) — no per-tile binary search in the tile kernel).
//
// Spill pieces are binned by a counting sort over tiles.  The records of a block are consecutive in start order, so their
// pieces land in a handful of tiles just behind the block's first home tile: counts (and, in the fill pass, ranks) are
// taken in an LDS window of COV_SW tiles and reach the global per-tile counters once per block and tile, not once per
// piece — deep pile-ups made the per-piece global atomics on one counter the cost of both passes.
constexpr uint32_t COV_SW = 64;
__device__ __forceinline__ uint64_t cov_cs_of(const CovArrays& A, uint32_t j) {
  const uint32_t b = A.bid[j];
  return A.b_off[b] + (uint64_t)(A.start[j] - A.b_start[b]);
}
__global__ __launch_bounds__(256) void cov_cs_count_k(uint32_t m, CovArrays A, const uint32_t* __restrict__ cig_off,
                                                      const uint32_t* __restrict__ cig, uint32_t* __restrict__ tile_cnt,
                                                      uint32_t* __restrict__ tile_first, uint32_t ntiles) {
  __shared__ uint32_t lcnt[COV_SW];
  __shared__ uint32_t s_tbase;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x < COV_SW) lcnt[threadIdx.x] = 0;
  uint64_t cs = 0;
  bool spills = false;
  if (j < m) {
    cs = cov_cs_of(A, j);
    A.cs[j] = cs;
    A.pk[j] = (uint16_t)((uint32_t)(cs % COV_W) | (A.bhead[j] ? 0x8000u : 0u));
    const uint32_t t = (uint32_t)(cs / COV_W);
    if (threadIdx.x == 0) s_tbase = t;
    uint32_t tp = j ? (uint32_t)(cov_cs_of(A, j - 1) / COV_W) + 1u : 0u;  // tiles after the previous record's home tile
    for (; tp <= t; ++tp) tile_first[tp] = j;
    if (j + 1 == m)
      for (uint32_t u = t + 1; u <= ntiles; ++u) tile_first[u] = m;
    // a read whose reference span ends inside its home tile has no piece elsewhere: no CIGAR walk (96 % of the reads)
    spills = (uint32_t)(cs % COV_W) + (uint32_t)(A.end[j] - A.start[j] + 1) > (uint32_t)COV_W;
  }
  __syncthreads();
  const uint32_t tbase = s_tbase;
  if (spills) {
    const uint32_t i = A.ridx[j];
    for_each_spill_piece(cs, cig + cig_off[i], cig_off[i + 1] - cig_off[i], [&](uint32_t t, uint32_t, uint32_t) {
      if (t - tbase < COV_SW)
        atomicAdd(&lcnt[t - tbase], 1u);
      else
        atomicAdd(&tile_cnt[t], 1u);
    });
  }
  __syncthreads();
  if (threadIdx.x < COV_SW && lcnt[threadIdx.x]) atomicAdd(&tile_cnt[tbase + threadIdx.x], lcnt[threadIdx.x]);
}

__global__ __launch_bounds__(256) void cov_spill_fill_k(uint32_t m, CovArrays A, const uint32_t* __restrict__ cig_off,
                                                        const uint32_t* __restrict__ cig, const uint32_t* __restrict__ tile_off,
                                                        uint32_t* __restrict__ tile_fill, uint32_t* __restrict__ sp_seg,
                                                        uint32_t* __restrict__ sp_rec) {
  __shared__ uint32_t lcnt[COV_SW], lbase[COV_SW];
  __shared__ uint32_t s_tbase;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x < COV_SW) lcnt[threadIdx.x] = 0;
  uint64_t cs = 0;
  bool spills = false;
  uint32_t i = 0, nc = 0;
  if (j < m) {
    cs = A.cs[j];
    if (threadIdx.x == 0) s_tbase = (uint32_t)(cs / COV_W);
    spills = (uint32_t)(cs % COV_W) + (uint32_t)(A.end[j] - A.start[j] + 1) > (uint32_t)COV_W;
    if (spills) {
      i = A.ridx[j];
      nc = cig_off[i + 1] - cig_off[i];
    }
  }
  __syncthreads();
  const uint32_t tbase = s_tbase;
  if (spills)  // pass 1: how many pieces the block sends to each tile of its window
    for_each_spill_piece(cs, cig + cig_off[i], nc, [&](uint32_t t, uint32_t, uint32_t) {
      if (t - tbase < COV_SW) atomicAdd(&lcnt[t - tbase], 1u);
    });
  __syncthreads();
  if (threadIdx.x < COV_SW) {  // one reservation per (block, tile)
    const uint32_t c = lcnt[threadIdx.x];
    lbase[threadIdx.x] = c ? atomicAdd(&tile_fill[tbase + threadIdx.x], c) : 0u;
    lcnt[threadIdx.x] = 0;
  }
  __syncthreads();
  if (spills)  // pass 2: ranks inside the reservation from the LDS counters
    for_each_spill_piece(cs, cig + cig_off[i], nc, [&](uint32_t t, uint32_t off, uint32_t len) {
      uint32_t slot;
      if (t - tbase < COV_SW)
        slot = tile_off[t] + lbase[t - tbase] + atomicAdd(&lcnt[t - tbase], 1u);
      else
        slot = tile_off[t] + atomicAdd(&tile_fill[t], 1u);
      sp_seg[slot] = off | ((len - 1) << 16);  // off < 8192, len-1 < 8192
      sp_rec[slot] = j;
    });
}

// ---- C8: the tile kernel --------------------------------------------------------------------
__device__ __forceinline__ uint32_t padidx(uint32_t p) { return p + (p >> 5); }

template <class AccT>
__device__ __forceinline__ AccT to_acc(double y);
template <>
__device__ __forceinline__ int32_t to_acc<int32_t>(double y) { return (int32_t)y; }
template <>
__device__ __forceinline__ long long to_acc<long long>(double y) { return (long long)y; }

__device__ __forceinline__ void lds_add(int32_t* p, int32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ void lds_add(long long* p, long long v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }

// What the lean interval chain (cl_* below) wants from the tile kernel besides the change points: how many intervals the tile
// emits for certain (change points with a depth other than 0, the tentative one at the tile start aside), whether its first change
// point is tentative (bit 31 of ecnt), the depth at its first and at its last base — with these the run-length encoding needs no
// pass over the change points to count (ecnt == nullptr: not wanted).
struct ClTileOut {
  uint32_t* ecnt;
  double *fv, *lv;
};

// change point: pos = cpos | tentative<<63 ; val = depth (as double: exact for |v| < 2^53)
// IDENT: the records are their own compaction (device chain: no validity pass), ridx[j] == j.
// 512 threads per 8192-base tile (16 bases each in the scan phase) at <= 64 VGPRs: four blocks = 32 waves per CU, the
// occupancy this latency-bound kernel needs (PMC: 70 % of its wave cycles are spent parked on memory at 16 waves per CU).
constexpr int COV_R = 2;  // home records a thread has in flight: their independent loads are issued together
constexpr int COVT_NT = 512;
constexpr int COVT_PER = COV_W / COVT_NT;
template <class AccT, bool IDENT, bool LEAN>
__global__ __launch_bounds__(COVT_NT, 8) void cov_tile_k(uint32_t m, uint64_t S, CovArrays A, const uint32_t* __restrict__ cig_off,
                                                     const uint32_t* __restrict__ cig, const double* __restrict__ yc,
                                                     const uint32_t* __restrict__ tile_first,
                                                     const uint32_t* __restrict__ sp_off, const uint32_t* __restrict__ sp_seg,
                                                     const uint32_t* __restrict__ sp_rec, uint64_t* __restrict__ cp_pos,
                                                     double* __restrict__ cp_val, uint32_t* __restrict__ tile_cp_base,
                                                     uint32_t* __restrict__ tile_cp_cnt, uint32_t* __restrict__ cp_alloc,
                                                     uint32_t cp_cap, uint32_t* __restrict__ err, ClTileOut O) {
  __shared__ AccT diff[COV_W + COV_W / 32 + 1];
  __shared__ uint32_t brk[COV_W / 32];
  __shared__ uint32_t sm_u[COVT_NT / 64];
  __shared__ AccT sm_a[COVT_NT / 64];
  __shared__ uint32_t s_base, s_ne;

  const uint32_t t = threadIdx.x;
  const uint64_t tile = blockIdx.x;
  const uint64_t t0 = tile * COV_W;
  const uint32_t wlen = (uint32_t)((S - t0) < (uint64_t)COV_W ? (S - t0) : (uint64_t)COV_W);
  const uint32_t r0 = tile_first[tile], r1 = tile_first[tile + 1];
  const uint32_t s0 = sp_off[tile], s1 = sp_off[tile + 1];

  for (uint32_t p = t; p < COV_W + COV_W / 32 + 1; p += COVT_NT) diff[p] = 0;
  if (t < COV_W / 32) brk[t] = 0;
  if (t == 0) s_ne = 0;
  __syncthreads();
  // home records: COV_R per thread and round, so that the (independent) loads of a round are in flight together and only
  // the CIGAR words wait for their offsets
  for (uint32_t jb = r0; jb < r1; jb += COVT_NT * COV_R) {
    uint32_t c0[COV_R], c1[COV_R], w0[COV_R], lp[COV_R];
    AccT y[COV_R];
    bool bh[COV_R];
#pragma unroll
    for (int u = 0; u < COV_R; ++u) {
      const uint32_t j = jb + (uint32_t)u * COVT_NT + t;
      c0[u] = c1[u] = 0;
      if (j < r1) {
        const uint32_t i = IDENT ? j : A.ridx[j];
        c0[u] = cig_off[i];
        c1[u] = cig_off[i + 1];
        if constexpr (sizeof(AccT) == 4)
          y[u] = A.yi[j];
        else
          y[u] = to_acc<AccT>(yc ? yc[i] : 1.0);
        const uint32_t pk = A.pk[j];
        lp[u] = pk & 0x1FFFu;
        bh[u] = (pk & 0x8000u) != 0;
      }
    }
#pragma unroll
    for (int u = 0; u < COV_R; ++u) w0[u] = c0[u] < c1[u] ? cig[c0[u]] : 0u;
#pragma unroll
    for (int u = 0; u < COV_R; ++u) {
      if (jb + (uint32_t)u * COVT_NT + t >= r1) continue;
      uint32_t p = lp[u];  // position relative to the tile start; pieces beyond the tile were pre-binned as spills
      if (bh[u]) atomicOr(&brk[p >> 5], 1u << (p & 31));
      for (uint32_t k = c0[u]; k < c1[u]; ++k) {
        const uint32_t c = k == c0[u] ? w0[u] : cig[k];
        const uint32_t op = cig_op(c), len = cig_len(c);
        if (op == C_M) {
          if (p < (uint32_t)COV_W) {  // the part inside the home tile
            lds_add(&diff[padidx(p)], y[u]);
            if (p + len < (uint32_t)COV_W) lds_add(&diff[padidx(p + len)], (AccT)(-y[u]));
          }
          p += len;
          if (p >= (uint32_t)COV_W) break;  // everything further lies in later tiles
        } else if (op == C_D || op == C_N) {
          p += len;
          if (p >= (uint32_t)COV_W) break;
        }
      }
    }
  }
  for (uint32_t s = s0 + t; s < s1; s += COVT_NT) {
    uint32_t seg = sp_seg[s];
    uint32_t off = seg & 0xFFFFu, len = (seg >> 16) + 1;
    uint32_t i = IDENT ? sp_rec[s] : A.ridx[sp_rec[s]];
    AccT y = to_acc<AccT>(yc ? yc[i] : 1.0);
    lds_add(&diff[padidx(off)], y);
    if (off + len < COV_W) lds_add(&diff[padidx(off + len)], (AccT)(-y));
  }
  __syncthreads();
  // per-thread serial prefix over its 32 bases (kept in registers), then a block scan of the thread totals
  AccT v[COVT_PER];
  AccT run = 0;
  const uint32_t pb = t * COVT_PER;
#pragma unroll
  for (int q = 0; q < COVT_PER; ++q) {
    run += diff[padidx(pb + q)];
    v[q] = run;
  }
  AccT tot;
  const AccT texcl = block_excl_sum<AccT, COVT_NT>(run, sm_a, &tot);
  // change points
  uint32_t cnt = 0;
  const uint32_t bw = (brk[pb >> 5] >> (pb & 31u)) & ((1u << COVT_PER) - 1u);
  AccT prev = texcl;
  uint32_t mask = 0;
#pragma unroll
  for (int q = 0; q < COVT_PER; ++q) {
    const uint32_t p = pb + q;
    v[q] += texcl;
    const bool cp = (p < wlen) && (p == 0 || v[q] != prev || ((bw >> q) & 1u));
    mask |= cp ? (1u << q) : 0u;
    prev = v[q];
  }
  cnt = (uint32_t)__builtin_popcount(mask);
  if (LEAN) {
    uint32_t ne = 0;
#pragma unroll
    for (int q = 0; q < COVT_PER; ++q) {
      const bool tent = (pb + q == 0) && !((bw >> q) & 1u);
      ne += (((mask >> q) & 1u) && !tent && v[q] != 0) ? 1u : 0u;
      if (pb + q + 1 == wlen) O.lv[tile] = (double)v[q];
    }
    ne = wave_sum(ne);
    if (lane_id() == 0 && ne) atomicAdd(&s_ne, ne);
  }
  uint32_t btot;
  uint32_t cex = block_excl_sum<uint32_t, COVT_NT>(cnt, sm_u, &btot);
  if (t == 0) {
    uint32_t base = atomicAdd(cp_alloc, btot);
    s_base = base;
    tile_cp_base[tile] = base;
    tile_cp_cnt[tile] = btot;
    if ((uint64_t)base + btot > cp_cap) atomicOr(err, TBK_DERR_INTERNAL);
    if (LEAN) {
      O.ecnt[tile] = s_ne | ((bw & 1u) ? 0u : 0x80000000u);  // (thread 0 owns base 0: tentative unless a bundle starts there)
      O.fv[tile] = (double)v[0];
    }
  }
  __syncthreads();
  uint32_t o = s_base + cex;
  if ((uint64_t)s_base + btot <= cp_cap) {
#pragma unroll
    for (int q = 0; q < COVT_PER; ++q) {
      if ((mask >> q) & 1u) {
        const uint32_t p = pb + q;
        const bool tent = (p == 0) && !((bw >> q) & 1u);
        cp_pos[o] = (t0 + p) | (tent ? (1ull << 63) : 0ull);
        cp_val[o] = (double)v[q];
        ++o;
      }
    }
  }
}

// ---- C10..C12: order the change points, turn them into intervals ------------------------------
__global__ void cov_cp_gather_k(uint32_t ntiles, const uint32_t* __restrict__ tile_cp_base, const uint32_t* __restrict__ tile_cp_cnt,
                                const uint32_t* __restrict__ tile_cp_off, const uint64_t* __restrict__ cp_pos,
                                const double* __restrict__ cp_val, uint64_t* __restrict__ sp, double* __restrict__ sv) {
  uint32_t tile = blockIdx.x;
  uint32_t base = tile_cp_base[tile], cnt = tile_cp_cnt[tile], off = tile_cp_off[tile];
  for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) {
    sp[off + k] = cp_pos[base + k];
    sv[off + k] = cp_val[base + k];
  }
}

__device__ __forceinline__ bool cp_kept(const uint64_t* sp, const double* sv, uint32_t q) {
  if (!(sp[q] >> 63)) return true;
  return q > 0 && sv[q] != sv[q - 1];
}


// The run-length encoding in two streaming passes over tiles of IV_TILE change points (row r of a tile: 256 consecutive change
// points, thread order = their order): how many intervals a tile emits, a scan of the tile counts, then every interval written at its
// place — no per-change-point flag, offset and end-position arrays in between.
constexpr uint32_t IV_NT = 256, IV_ROWS = 8, IV_TILE = IV_NT * IV_ROWS;
__device__ __forceinline__ bool iv_emits(const uint64_t* __restrict__ sp, const double* __restrict__ sv, uint32_t q) {
  return cp_kept(sp, sv, q) && sv[q] != 0.0;
}
__global__ __launch_bounds__(IV_NT) void iv_count_k(uint32_t ncp, const uint64_t* __restrict__ sp, const double* __restrict__ sv, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t sm[IV_NT / 64];
  uint32_t n = 0;
#pragma unroll
  for (uint32_t r = 0; r < IV_ROWS; ++r) {
    const uint64_t q = (uint64_t)blockIdx.x * IV_TILE + (uint64_t)r * IV_NT + threadIdx.x;
    n += q < ncp && iv_emits(sp, sv, (uint32_t)q) ? 1u : 0u;
  }
  n = wave_sum(n);
  if (lane_id() == 0) sm[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ __launch_bounds__(IV_NT) void iv_emit_k(uint32_t ncp, uint64_t S, uint32_t nb, CovArrays A, const uint64_t* __restrict__ sp,
                                                   const double* __restrict__ sv, const uint32_t* __restrict__ off, const uint32_t* __restrict__ tile_b,
                                                   uint32_t cap, int32_t* __restrict__ iv_tid, int32_t* __restrict__ iv_start, int32_t* __restrict__ iv_end,
                                                   double* __restrict__ iv_val) {
  __shared__ uint32_t sm[8];
  uint32_t run = off[blockIdx.x];
#pragma unroll 1
  for (uint32_t r = 0; r < IV_ROWS; ++r) {
    const uint64_t q64 = (uint64_t)blockIdx.x * IV_TILE + (uint64_t)r * IV_NT + threadIdx.x;
    const uint32_t q = (uint32_t)q64;
    const bool e = q64 < ncp && iv_emits(sp, sv, q);
    uint32_t tot;
    const uint32_t o = run + block_excl_sum<uint32_t, IV_NT>(e ? 1u : 0u, sm, &tot);
    run += tot;
    if (!e || o >= cap) continue;
    uint32_t nx = q + 1;  // the next change point that is kept ends the interval
    while (nx < ncp && !cp_kept(sp, sv, nx)) ++nx;
    const uint64_t endp = nx < ncp ? (sp[nx] & ~(1ull << 63)) : S;
    const uint64_t p = sp[q] & ~(1ull << 63);
    const uint32_t tl = (uint32_t)(p / COV_W);
    uint32_t lo = tile_b[tl], hi = tile_b[tl + 1] + 1u;  // last bundle with b_off <= p: between the bundles of the tile's two ends
    if (hi > nb) hi = nb;
    while (hi - lo > 1) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (A.b_off[mid] <= p)
        lo = mid;
      else
        hi = mid;
    }
    const int32_t s0 = A.b_start[lo] - 1 + (int32_t)(p - A.b_off[lo]);
    iv_tid[o] = A.b_tid[lo];
    iv_start[o] = s0;
    iv_end[o] = s0 + (int32_t)(endp - p);
    iv_val[o] = sv[q];
  }
}

// tile_b[t] = last bundle with b_off <= t * COV_W (t <= ntiles): an interval's bundle is then a few bisection steps inside its tile's
// bundles instead of eighteen over all of them
__global__ void cov_tile_bundle_k(uint32_t ntiles, uint32_t nb, const uint64_t* __restrict__ b_off, uint32_t* __restrict__ tile_b) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  const uint64_t p = (uint64_t)t * COV_W;
  uint32_t lo = 0, hi = nb;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (b_off[mid] <= p)
      lo = mid;
    else
      hi = mid;
  }
  tile_b[t] = lo;
}

// ---- the lean interval chain (integral YC) --------------------------------------------------------------------------------------
// The same intervals as the chain above with one read-back instead of four and 11 launches instead of 25.  What makes it lean:
//  * a record's compacted start needs no bundle table and no scan over the bundles.  With C_j = the sum, over the bundle heads h <= j,
//    of X_h = start_h - 1 - (running maximum of `end` over the records before h on the reference before h; 0 before the first
//    record) the compacted start is cs_j = start_j - 1 - C_j: a head lands right behind the last base of the bundle before it,
//    whether that bundle lies on the same reference or on the one before.  X rides with the head counts of the bundle passes
//    (cl_heads_k<false>: heads and sum of X per 4096-record tile; one small scan; cl_heads_k<true>: the same walk, numbering), so
//    the pass that numbers the bundles also writes every record's tile-kernel word, the bundle table with its offsets, the first
//    record and the first bundle of every coverage tile, counts the pieces that leave their home tiles and lists the few
//    records that have such pieces (cov_bundle_span + scan, cov_cs_count, cov_tile_bundle and the first half of cov_spill_fill);
//  * the spill pieces are filled from that list (2 % of the records) instead of a second pass over all of them;
//  * the tile kernel leaves behind how many intervals each tile emits (ClTileOut), so the run-length encoding is one small pass
//    over the tiles and one pass over the change points where they lie — no gather into tile order, no counting pass.
// The tile arrays are sized before the number of tiles is known; a tile beyond their capacity raises scalar 13 and the caller
// runs the chain above instead.
struct ClTiles {
  uint32_t* first;  // first record whose compacted start lies in the tile or later
  uint32_t* tb;     // bundle that holds the tile's first base
  uint32_t* cnt;    // pieces binned into the tile from other home tiles (zeroed by the caller)
  uint32_t cap;     // tiles the arrays hold (entries 0 .. cap - 1; entry ntiles is written too)
};
constexpr uint32_t CL_SC_NB = 3, CL_SC_S = 4, CL_SC_NSPILL = 5, CL_SC_NSL = 12, CL_SC_REFUSED = 13;

// the spilling records of one block of cl_heads_k<true> (4096 records): a chunk of the list, the block's first home tile with it
struct ClChunks {
  uint32_t *base, *cnt, *tbase;  // per block
  uint32_t* j;                   // the list: record
  uint64_t* cs;                  //           its compacted start
};
// one block: exclusive prefixes of the per-tile head counts and X sums (the same shape as cb_spine_block, run by the last block of
// cl_heads_k<false>)
template <uint32_t NT>
__device__ __forceinline__ void cl_scan_block(uint32_t nt, const uint32_t* __restrict__ hcnt, const long long* __restrict__ xsum,
                                              uint32_t* __restrict__ hbase, long long* __restrict__ xbase, uint32_t* sh /* [NT / 64] */,
                                              long long* sx /* [NT / 64] */) {
  const uint32_t per = (nt + NT - 1u) / NT, i0 = threadIdx.x * per, i1 = i0 + per < nt ? i0 + per : nt;
  uint32_t h = 0;
  long long x = 0;
  for (uint32_t q0 = i0; q0 < i1; q0 += 8u) {
    uint32_t h8[8];
    long long x8[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
      h8[u] = q0 + u < i1 ? __hip_atomic_load(hcnt + q0 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      x8[u] = q0 + u < i1 ? __hip_atomic_load(xsum + q0 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ll;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
      h += h8[u];
      x += x8[u];
    }
  }
  uint32_t hi_ = h;
  long long xi_ = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t oh = __shfl_up(hi_, d, 64);
    const long long ox = __shfl_up(xi_, d, 64);
    if ((int)lane_id() >= d) {
      hi_ += oh;
      xi_ += ox;
    }
  }
  if (lane_id() == 63) {
    sh[threadIdx.x >> 6] = hi_;
    sx[threadIdx.x >> 6] = xi_;
  }
  __syncthreads();
  uint32_t rh = hi_ - h;
  long long rx = xi_ - x;
  for (uint32_t q = 0; q < (threadIdx.x >> 6); ++q) {
    rh += sh[q];
    rx += sx[q];
  }
  for (uint32_t q0 = i0; q0 < i1; q0 += 8u) {
    uint32_t h8[8];
    long long x8[8];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
      h8[u] = q0 + u < i1 ? __hip_atomic_load(hcnt + q0 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      x8[u] = q0 + u < i1 ? __hip_atomic_load(xsum + q0 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ll;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
      if (q0 + u < i1) {
        hbase[q0 + u] = rh;
        xbase[q0 + u] = rx;
      }
      rh += h8[u];
      rx += x8[u];
    }
  }
}

template <bool EMIT>
__global__ __launch_bounds__(CB_NT) void cl_heads_k(uint32_t m, CovArrays A, const uint4* __restrict__ part /* per tile: the aggregate of the tiles before it */,
                                                    uint32_t* __restrict__ hcnt, long long* __restrict__ xsum, const uint32_t* __restrict__ hbase,
                                                    const long long* __restrict__ xbase, const uint32_t* __restrict__ cig_off,
                                                    const uint32_t* __restrict__ cig, ClTiles T, ClChunks L, uint64_t* __restrict__ sc,
                                                    uint32_t* __restrict__ err, uint32_t* __restrict__ done, uint32_t* __restrict__ hbase_out,
                                                    long long* __restrict__ xbase_out) {
  // The four rows of a block go through the scans together: their loads are issued at once, their wave scans run side by side and the
  // sixteen (row, wave) aggregates meet in LDS behind ONE barrier per scan (the row-by-row form had twenty barriers per block and a
  // round trip to memory per row).
  constexpr uint32_t NWV = CB_NT / 64;
  __shared__ CbAgg wl[CB_ROWS * NWV];
  __shared__ uint32_t smu[CB_ROWS * NWV];
  __shared__ long long smx[CB_ROWS * NWV];
  // EMIT: the block's spilling records and its piece counts gather in LDS and reach global memory once per block — a reservation
  // per row on one word (24 k returning atomics on config 3) was most of this kernel's time
  __shared__ uint32_t lcnt[EMIT ? COV_SW : 1];
  __shared__ uint16_t l_j[EMIT ? CB_TILE : 1];
  __shared__ uint32_t l_cs[EMIT ? CB_TILE : 1];  // (low word: the starts of a block lie within 2^32 compacted bases of its first, or the chain is refused)
  __shared__ uint32_t s_tbase, s_nsl, s_np, s_slbase;
  __shared__ uint64_t s_cs0, s_cs1;
  const CbOp op{};
  const CbAgg none{0, INT32_MIN, INT32_MIN, 1u};
  auto un = [](const uint4& v) { return CbAgg{(int32_t)v.x, (int32_t)v.y, (int32_t)v.z, v.w}; };
  const uint32_t wv = threadIdx.x >> 6;
  uint32_t heads_tile = 0;
  long long x_tile = 0;
  bool bad = false, refused = false;
  if (EMIT) {
    if (threadIdx.x < COV_SW) lcnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_nsl = s_np = 0;
  }
  int32_t t4[CB_ROWS][4], s4[CB_ROWS][4], e4[CB_ROWS][4];
#pragma unroll
  for (uint32_t r = 0; r < CB_ROWS; ++r) {
    const uint64_t i = (uint64_t)blockIdx.x * CB_TILE + ((uint64_t)r * CB_NT + threadIdx.x) * 4u;
    cb_load4(A.tid, i, m, 0, t4[r]);
    cb_load4(A.start, i, m, 0, s4[r]);
    cb_load4(A.end, i, m, 0, e4[r]);
  }
  // what came before the block (cb_spine_block, cl_scan_block)
  const CbAgg run0 = un(part[blockIdx.x]);
  const uint32_t hrun0 = EMIT ? hbase[blockIdx.x] : 0u;
  const long long xrun0 = EMIT ? xbase[blockIdx.x] : 0ll;
  CbAgg ex[CB_ROWS];  // the aggregate of everything before this thread's first record of the row
#pragma unroll
  for (uint32_t r = 0; r < CB_ROWS; ++r) {
    const uint64_t i = (uint64_t)blockIdx.x * CB_TILE + ((uint64_t)r * CB_NT + threadIdx.x) * 4u;
    CbAgg a = none;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < m) a = op(a, CbAgg{t4[r][e], t4[r][e], e4[r][e], 1u});
    const CbAgg inc = wave_incl_scan_op(a, op);
    if (lane_id() == 63) wl[r * NWV + wv] = inc;
    ex[r] = shfl_up_t(inc, 1);  // (lane 0: replaced below)
  }
  __syncthreads();
  {
    CbAgg acc = run0;  // run0 (+) the (row, wave) aggregates before this thread's wave, rows in order
#pragma unroll
    for (uint32_t q = 0; q < CB_ROWS * NWV; ++q) {
      if (q % NWV == wv) ex[q / NWV] = lane_id() == 0 ? acc : op(acc, ex[q / NWV]);
      acc = op(acc, wl[q]);
    }
  }
  uint32_t hdn[CB_ROWS];  // head flags of the four records : 4 | their number << 4
  uint32_t hinc[CB_ROWS];
  long long xs[CB_ROWS], xinc[CB_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < CB_ROWS; ++r) {
    const uint64_t i = (uint64_t)blockIdx.x * CB_TILE + ((uint64_t)r * CB_NT + threadIdx.x) * 4u;
    uint32_t hd = 0, nh = 0;
    xs[r] = 0;
    CbAgg w = ex[r];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i + e < m) {
        const bool first = w.last_tid == INT32_MIN;
        const bool head = first || t4[r][e] != w.last_tid || s4[r][e] > w.mx;  // tiecov.cpp:443
        if (head) {
          xs[r] += (long long)s4[r][e] - 1ll - (first ? 0ll : (long long)w.mx);
          hd |= 1u << e;
          ++nh;
        }
        w = op(w, CbAgg{t4[r][e], t4[r][e], e4[r][e], 1u});
      }
    }
    hdn[r] = hd | (nh << 4);
    // heads and X sums before this thread in its wave's part of the row
    uint32_t hi_ = nh;
    long long xi_ = xs[r];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t oh = __shfl_up(hi_, d, 64);
      const long long ox = __shfl_up(xi_, d, 64);
      if ((int)lane_id() >= d) {
        hi_ += oh;
        xi_ += ox;
      }
    }
    if (lane_id() == 63) {
      smu[r * NWV + wv] = hi_;
      smx[r * NWV + wv] = xi_;
    }
    hinc[r] = hi_;
    xinc[r] = xi_;
  }
  __syncthreads();
  // per row: bundles before this thread's records, the X sum before them, and the (reference, running end) the row's walk starts from
  uint32_t bf[CB_ROWS];
  long long cf[CB_ROWS];
  {
    uint32_t ah = 0;
    long long ax = 0;
#pragma unroll
    for (uint32_t q = 0; q < CB_ROWS * NWV; ++q) {
      if (q % NWV == wv) {
        bf[q / NWV] = hrun0 + ah + hinc[q / NWV] - (hdn[q / NWV] >> 4);
        cf[q / NWV] = xrun0 + ax + xinc[q / NWV] - xs[q / NWV];
      }
      ah += smu[q];
      ax += smx[q];
    }
    heads_tile = ah;
    x_tile = ax;
  }
  if (EMIT) {
    // the numbering walk, row by row, on what the scans left behind: five words per row, chosen by the row (the rows' records are
    // read again, out of L2: kept in registers across an unrolled walk they cost the kernel three quarters of its resident waves)
    const uint32_t hd_all = (hdn[0] & 15u) | ((hdn[1] & 15u) << 4) | ((hdn[2] & 15u) << 8) | ((hdn[3] & 15u) << 12);
    static_assert(CB_ROWS == 4, "the row selects below");
#define CL_ROW(a, r) ((r) == 0 ? a[0] : (r) == 1 ? a[1] : (r) == 2 ? a[2] : a[3])
    const int32_t vt[CB_ROWS] = {ex[0].last_tid, ex[1].last_tid, ex[2].last_tid, ex[3].last_tid};
    const int32_t vm[CB_ROWS] = {ex[0].mx, ex[1].mx, ex[2].mx, ex[3].mx};
#pragma unroll 1
    for (uint32_t r = 0; r < CB_ROWS; ++r) {
      const uint64_t i = (uint64_t)blockIdx.x * CB_TILE + ((uint64_t)r * CB_NT + threadIdx.x) * 4u;
      int32_t rt[4], rs[4], re[4];
      cb_load4(A.tid, i, m, 0, rt);
      cb_load4(A.start, i, m, 0, rs);
      cb_load4(A.end, i, m, 0, re);
      const int32_t pst = i > 0 && i < m ? A.start[i - 1] : 0;  // (the record before the four: order check, its tile)
      const uint32_t hd = (hd_all >> (4u * r)) & 15u;
      long long C = CL_ROW(cf, r);  // the sum of X over the heads before this thread's records
      uint32_t b = CL_ROW(bf, r);
      int32_t v_tid = CL_ROW(vt, r), v_mx = CL_ROW(vm, r);  // the walk's state: last reference (INT32_MIN: none), running maximum of `end` on it
      uint64_t prev_cs = i > 0 && i < m ? (uint64_t)((long long)pst - 1ll - C) : 0ull;
      uint64_t pk4 = 0;  // the four tile-kernel words of this thread's records: one store
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint64_t j = i + e;
        if (j < m) {
          const bool head = (hd >> e) & 1u;
          const bool run_head = v_tid == INT32_MIN || rt[e] != v_tid;
          const int32_t prev_start = e == 0 ? pst : rs[e - 1];
          if (!run_head && rs[e] < prev_start) bad = true;
          b += head ? 1u : 0u;
          if (head) C += (long long)rs[e] - 1ll - (v_tid == INT32_MIN ? 0ll : (long long)v_mx);
          const uint32_t bundle = b - 1u;
          const uint64_t cs = (uint64_t)((long long)rs[e] - 1ll - C);
          pk4 |= (uint64_t)((uint32_t)(cs % COV_W) | (head ? 0x8000u : 0u)) << (16 * e);
          if (head) {
            A.b_tid[bundle] = rt[e];
            A.b_start[bundle] = rs[e];
            A.b_off[bundle] = cs;
          }
          const uint64_t tl = cs / COV_W;
          if (r == 0 && threadIdx.x == 0 && e == 0) {  // (read behind the barrier that closes the rows)
            s_tbase = (uint32_t)tl;
            s_cs0 = cs;
          }
          if (j + 1 == m || j + 1 == ((uint64_t)blockIdx.x + 1u) * CB_TILE) s_cs1 = cs;  // the block's last record
          for (uint64_t tp = j ? prev_cs / COV_W + 1u : 0u; tp <= tl; ++tp) {  // the tiles whose first record this one is
            if (tp >= T.cap) {
              refused = true;
              break;
            }
            T.first[tp] = (uint32_t)j;
            T.tb[tp] = bundle - ((head && cs > tp * COV_W) ? 1u : 0u);
          }
          v_mx = (!run_head && v_mx > re[e]) ? v_mx : re[e];  // (CbOp on a single record)
          v_tid = rt[e];
          if (j + 1 == m) {
            const uint64_t S = (uint64_t)((long long)v_mx - C);
            const uint64_t nt = (S + COV_W - 1) / COV_W;
            sc[CL_SC_NB] = (uint64_t)bundle + 1u;
            sc[CL_SC_S] = S;
            if (nt >= T.cap) {
              refused = true;
            } else {
              for (uint64_t u = tl + 1; u <= nt; ++u) {
                T.first[u] = m;
                T.tb[u] = bundle;
              }
            }
          }
          // a read whose reference span ends inside its home tile has no piece elsewhere (96 % of the reads)
          if ((uint32_t)(cs % COV_W) + (uint32_t)(re[e] - rs[e] + 1) > (uint32_t)COV_W) {
            const uint32_t k = atomicAdd(&s_nsl, 1u);
            l_j[k] = (uint16_t)(j - (uint64_t)blockIdx.x * CB_TILE);
            l_cs[k] = (uint32_t)cs;
          }
          prev_cs = cs;
        }
      }
      if (i + 3 < m) {
        *reinterpret_cast<uint64_t*>(A.pk + i) = pk4;  // (i is a multiple of 4, the array 256-byte aligned)
      } else {
        for (int e = 0; e < 4; ++e)
          if (i + e < m) A.pk[i + e] = (uint16_t)(pk4 >> (16 * e));
      }
    }
#undef CL_ROW
    __syncthreads();
  }
  if (!EMIT) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(hcnt + blockIdx.x, heads_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(xsum + blockIdx.x, x_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (cb_last_block(done)) cl_scan_block<CB_NT>(gridDim.x, hcnt, xsum, hbase_out, xbase_out, smu, smx);  // (the last block: the prefixes the numbering pass starts from)
  }
  if (EMIT) {
    // the block's spilling records: count their pieces per tile (LDS window behind the block's first home tile), one reservation
    // in the list, one chunk descriptor
    const uint32_t nsl = s_nsl, tbase = s_tbase;
    const uint64_t cs0 = s_cs0;
    if (s_cs1 - cs0 >= (1ull << 32)) refused = true;
    auto full_cs = [&](uint32_t k) { return cs0 + (uint64_t)(uint32_t)(l_cs[k] - (uint32_t)cs0); };
    if (nsl) {  // (uniform)
      if (threadIdx.x == 0) s_slbase = (uint32_t)atomicAdd((unsigned long long*)&sc[CL_SC_NSL], (unsigned long long)nsl);
      uint32_t np = 0;
      for (uint32_t k = threadIdx.x; k < nsl; k += CB_NT) {
        const uint32_t ri = A.ridx[(uint64_t)blockIdx.x * CB_TILE + l_j[k]];
        const uint32_t c0 = cig_off[ri];
        for_each_spill_piece(full_cs(k), cig + c0, cig_off[ri + 1] - c0, [&](uint32_t t, uint32_t, uint32_t) {
          ++np;
          if (t >= T.cap)
            refused = true;
          else if (t - tbase < COV_SW)
            atomicAdd(&lcnt[t - tbase], 1u);
          else
            atomicAdd(&T.cnt[t], 1u);
        });
      }
      np = wave_sum(np);
      if (lane_id() == 0 && np) atomicAdd(&s_np, np);
      __syncthreads();
      const uint32_t slb = s_slbase;
      for (uint32_t k = threadIdx.x; k < nsl; k += CB_NT) {
        L.j[slb + k] = (uint32_t)((uint64_t)blockIdx.x * CB_TILE + l_j[k]);
        L.cs[slb + k] = full_cs(k);
      }
      if (threadIdx.x < COV_SW && lcnt[threadIdx.x] && tbase + threadIdx.x < T.cap) atomicAdd(&T.cnt[tbase + threadIdx.x], lcnt[threadIdx.x]);
      if (threadIdx.x == 0) {
        atomicAdd((unsigned long long*)&sc[CL_SC_NSPILL], (unsigned long long)s_np);
        L.base[blockIdx.x] = slb;
        L.tbase[blockIdx.x] = tbase;
      }
    }
    if (threadIdx.x == 0) L.cnt[blockIdx.x] = nsl;
    if (bad) atomicOr(err, TBK_DERR_UNSORTED);
    if (refused) sc[CL_SC_REFUSED] = 1;
  }
}

// the spill pieces of the listed records -> their tiles' bins: one block per chunk of the list (the spilling records of 4096
// consecutive records: their pieces fall into a few tiles behind the chunk's first home tile), counts and ranks in an LDS window
// of those tiles, one reservation per (chunk, tile)
__global__ __launch_bounds__(256) void cl_spill_fill_k(ClChunks L, CovArrays A, const uint32_t* __restrict__ cig_off, const uint32_t* __restrict__ cig,
                                                       const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ tile_fill,
                                                       uint32_t* __restrict__ sp_seg, uint32_t* __restrict__ sp_rec) {
  __shared__ uint32_t lcnt[COV_SW], lbase[COV_SW];
  const uint32_t n = L.cnt[blockIdx.x];
  if (!n) return;
  const uint32_t base = L.base[blockIdx.x], tbase = L.tbase[blockIdx.x];
  if (threadIdx.x < COV_SW) lcnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n; k += 256) {  // pass 1: how many pieces the chunk sends to each tile of its window
    const uint32_t i = A.ridx[L.j[base + k]];
    const uint32_t c0 = cig_off[i];
    for_each_spill_piece(L.cs[base + k], cig + c0, cig_off[i + 1] - c0, [&](uint32_t t, uint32_t, uint32_t) {
      if (t - tbase < COV_SW) atomicAdd(&lcnt[t - tbase], 1u);
    });
  }
  __syncthreads();
  if (threadIdx.x < COV_SW) {
    const uint32_t c = lcnt[threadIdx.x];
    lbase[threadIdx.x] = c ? atomicAdd(&tile_fill[tbase + threadIdx.x], c) : 0u;
    lcnt[threadIdx.x] = 0;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n; k += 256) {  // pass 2: ranks inside the reservation from the LDS counters
    const uint32_t j = L.j[base + k];
    const uint32_t i = A.ridx[j];
    const uint32_t c0 = cig_off[i];
    for_each_spill_piece(L.cs[base + k], cig + c0, cig_off[i + 1] - c0, [&](uint32_t t, uint32_t off, uint32_t len) {
      uint32_t slot;
      if (t - tbase < COV_SW)
        slot = tile_off[t] + lbase[t - tbase] + atomicAdd(&lcnt[t - tbase], 1u);
      else
        slot = tile_off[t] + atomicAdd(&tile_fill[t], 1u);
      sp_seg[slot] = off | ((len - 1) << 16);
      sp_rec[slot] = j;
    });
  }
}

// intervals per tile: the tile kernel's count, plus the tentative change point at the tile start when the depth changes there
__device__ __forceinline__ bool cl_first_kept(const ClTileOut& O, uint32_t t) {
  return !(O.ecnt[t] >> 31) || (t > 0 && O.fv[t] != O.lv[t - 1]);
}
__global__ void cl_iv_count_k(uint32_t ntiles, ClTileOut O, uint32_t* __restrict__ icnt) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const uint32_t c = O.ecnt[t];
  icnt[t] = (c & 0x7FFFFFFFu) + (((c >> 31) && cl_first_kept(O, t) && O.fv[t] != 0.0) ? 1u : 0u);
}
// one block per coverage tile: its change points, where cov_tile_k left them, become its intervals
__global__ __launch_bounds__(256) void cl_iv_emit_k(uint32_t ntiles, uint64_t S, uint32_t nb, CovArrays A, const uint64_t* __restrict__ cp_pos,
                                                    const double* __restrict__ cp_val, const uint32_t* __restrict__ tile_cp_base,
                                                    const uint32_t* __restrict__ tile_cp_cnt, ClTileOut O, const uint32_t* __restrict__ ioff,
                                                    const uint32_t* __restrict__ tile_b, uint32_t cap, int32_t* __restrict__ iv_tid,
                                                    int32_t* __restrict__ iv_start, int32_t* __restrict__ iv_end, double* __restrict__ iv_val) {
  __shared__ uint32_t sm[8];
  const uint32_t t = blockIdx.x;
  const uint32_t base = tile_cp_base[t], cnt = tile_cp_cnt[t];
  const bool kept0 = cl_first_kept(O, t);
  uint32_t run = ioff[t];
  constexpr uint64_t TENT = 1ull << 63;
  for (uint32_t k0 = 0; k0 < cnt; k0 += 256) {
    const uint32_t k = k0 + threadIdx.x;
    const bool have = k < cnt;
    const uint64_t pr = have ? cp_pos[base + k] : 0ull;
    const double val = have ? cp_val[base + k] : 0.0;
    const bool e = have && (!(pr >> 63) || kept0) && val != 0.0;
    uint32_t tot;
    const uint32_t o = run + block_excl_sum<uint32_t, 256>(e ? 1u : 0u, sm, &tot);
    run += tot;
    if (!e || o >= cap) continue;
    uint64_t endp = S;  // the next change point that is kept ends the interval
    if (k + 1 < cnt) {
      endp = cp_pos[base + k + 1] & ~TENT;  // (only a tile's first change point can be tentative)
    } else {
      for (uint32_t t2 = t + 1; t2 < ntiles; ++t2) {
        if (cl_first_kept(O, t2)) {
          endp = (uint64_t)t2 * COV_W;
          break;
        }
        if (tile_cp_cnt[t2] > 1u) {
          endp = cp_pos[tile_cp_base[t2] + 1u] & ~TENT;
          break;
        }
      }
    }
    const uint64_t p = pr & ~TENT;
    uint32_t lo = tile_b[t], hi = tile_b[t + 1] + 1u;  // last bundle with b_off <= p: between the bundles of the tile's two ends
    if (hi > nb) hi = nb;
    while (hi - lo > 1) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (A.b_off[mid] <= p)
        lo = mid;
      else
        hi = mid;
    }
    const int32_t s0 = A.b_start[lo] - 1 + (int32_t)(p - A.b_off[lo]);
    iv_tid[o] = A.b_tid[lo];
    iv_start[o] = s0;
    iv_end[o] = s0 + (int32_t)(endp - p);
    iv_val[o] = val;
  }
}

// ---- ordered double path (non-integral YC): per base, add in record order -------------------
// One thread per base of the tile; candidate segments = spills (sorted by record) then home
// records in file order — the same order in which the reference's addCov touches the base.
// SAMPLE = true computes tiecov -s instead (addMean, tiecov.cpp:155-185): per base a float32 running mean of YX
// in record order, mean += (val - mean) / cnt; cnt++ starting from (0, 1); the emitted value is ceil(mean).
template <bool SAMPLE>
__global__ __launch_bounds__(COV_NT) void cov_tile_ordered_k(uint32_t m, uint64_t S, CovArrays A, const uint32_t* __restrict__ cig_off,
                                                             const uint32_t* __restrict__ cig, const double* __restrict__ yc,
                                                             const int64_t* __restrict__ yx, const uint32_t* __restrict__ tile_first,
                                                             const uint32_t* __restrict__ sp_off, const uint32_t* __restrict__ sp_seg,
                                                             const uint32_t* __restrict__ sp_rec, uint64_t* __restrict__ cp_pos,
                                                             double* __restrict__ cp_val, uint32_t* __restrict__ tile_cp_base,
                                                             uint32_t* __restrict__ tile_cp_cnt, uint32_t* __restrict__ cp_alloc,
                                                             uint32_t cp_cap, uint32_t* __restrict__ err) {
  __shared__ double depth[COV_W + COV_W / 32 + 1];
  __shared__ uint32_t brk[COV_W / 32];
  __shared__ uint32_t sm_u[8];
  __shared__ uint32_t s_base;
  const uint32_t t = threadIdx.x;
  const uint64_t tile = blockIdx.x;
  const uint64_t t0 = tile * COV_W;
  const uint32_t wlen = (uint32_t)((S - t0) < (uint64_t)COV_W ? (S - t0) : (uint64_t)COV_W);
  float* s_mean = reinterpret_cast<float*>(depth);        // SAMPLE: slot q -> mean at [2q], cnt at [2q+1]
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(depth);
  auto add_base = [&](uint32_t q, double y, float val) {
    if constexpr (SAMPLE) {
      float mean = s_mean[2 * q];
      uint32_t c = s_cnt[2 * q + 1];
      mean += (val - mean) / (float)c;
      s_mean[2 * q] = mean;
      s_cnt[2 * q + 1] = c + 1;
    } else {
      depth[q] += y;
    }
  };
  auto base_value = [&](uint32_t q) -> double {
    if constexpr (SAMPLE)
      return (double)(unsigned long long)ceilf(s_mean[2 * q]);
    else
      return depth[q];
  };
  for (uint32_t p = t; p < COV_W + COV_W / 32 + 1; p += COV_NT) {
    if constexpr (SAMPLE) {
      s_mean[2 * p] = 0.0f;
      s_cnt[2 * p + 1] = 1u;
    } else {
      depth[p] = 0.0;
    }
  }
  if (t < COV_W / 32) brk[t] = 0;
  __syncthreads();
  const uint32_t r0 = tile_first[tile], r1 = tile_first[tile + 1];
  const uint32_t s0 = sp_off[tile], s1 = sp_off[tile + 1];
  // Spills: the bin order is arbitrary (atomic slot assignment).  Serialise by record index:
  // repeatedly take the smallest record id greater than the last one processed (selection by
  // the whole block; spill counts per tile are small).  Each step adds one record's pieces.
  // Every base is owned by exactly one thread (p % COV_NT == t), so no atomics are needed and
  // each base sees its additions in increasing record order.
  {
    long long last = -1;
    __shared__ uint32_t s_next;
    for (;;) {
      if (t == 0) s_next = 0xFFFFFFFFu;
      __syncthreads();
      uint32_t best = 0xFFFFFFFFu;
      for (uint32_t s = s0 + t; s < s1; s += COV_NT) {
        uint32_t r = sp_rec[s];
        if ((long long)r > last && r < best) best = r;
      }
      if (best != 0xFFFFFFFFu) atomicMin(&s_next, best);
      __syncthreads();
      uint32_t nx = s_next;
      __syncthreads();
      if (nx == 0xFFFFFFFFu) break;
      double y = yc ? yc[A.ridx[nx]] : 1.0;
      float val = SAMPLE ? (float)(int)(float)(yx ? yx[A.ridx[nx]] : 1) : 0.0f;
      for (uint32_t s = s0; s < s1; ++s) {
        if (sp_rec[s] != nx) continue;
        uint32_t seg = sp_seg[s];
        uint32_t off = seg & 0xFFFFu, len = (seg >> 16) + 1;
        // thread t owns bases p with p % COV_NT == t
        uint32_t first = off + ((t + COV_NT - (off % COV_NT)) % COV_NT);
        for (uint32_t p = first; p < off + len; p += COV_NT) add_base(padidx(p), y, val);
      }
      last = nx;
    }
  }
  __syncthreads();
  for (uint32_t j = r0; j < r1; ++j) {  // home records, file order; all threads walk the same record
    uint32_t i = A.ridx[j];
    double y = yc ? yc[i] : 1.0;
    float val = SAMPLE ? (float)(int)(float)(yx ? yx[i] : 1) : 0.0f;
    uint64_t p = A.cs[j];
    if (t == 0 && A.bhead[j]) brk[(uint32_t)(p - t0) >> 5] |= 1u << ((uint32_t)(p - t0) & 31);
    uint32_t c0 = cig_off[i], c1 = cig_off[i + 1];
    for (uint32_t k = c0; k < c1; ++k) {
      uint32_t c = cig[k];
      uint32_t op = cig_op(c), len = cig_len(c);
      if (op == C_M) {
        uint64_t a = p, b = p + len;
        if (a < t0 + COV_W) {
          uint32_t la = (uint32_t)(a - t0);
          uint32_t lb = (uint32_t)((b < t0 + COV_W ? b : t0 + COV_W) - t0);
          uint32_t first = la + ((t + COV_NT - (la % COV_NT)) % COV_NT);
          for (uint32_t q = first; q < lb; q += COV_NT) add_base(padidx(q), y, val);
        }
        p = b;
      } else if (op == C_D || op == C_N) {
        p += len;
      }
    }
  }
  __syncthreads();
  uint32_t cnt = 0;
  const uint32_t pb = t * COV_PER;
  uint32_t bw = brk[t];
  uint32_t mask = 0;
  for (int q = 0; q < COV_PER; ++q) {
    uint32_t p = pb + q;
    double d = base_value(padidx(p));
    double pv = p ? base_value(padidx(p - 1)) : 0.0;
    bool cp = (p < wlen) && (p == 0 || d != pv || ((bw >> q) & 1u));
    mask |= cp ? (1u << q) : 0u;
    cnt += cp ? 1u : 0u;
  }
  uint32_t btot;
  uint32_t cex = block_excl_sum<uint32_t, COV_NT>(cnt, sm_u, &btot);
  if (t == 0) {
    uint32_t base = atomicAdd(cp_alloc, btot);
    s_base = base;
    tile_cp_base[tile] = base;
    tile_cp_cnt[tile] = btot;
    if ((uint64_t)base + btot > cp_cap) atomicOr(err, TBK_DERR_INTERNAL);
  }
  __syncthreads();
  uint32_t o = s_base + cex;
  if ((uint64_t)s_base + btot <= cp_cap) {
    for (int q = 0; q < COV_PER; ++q) {
      if ((mask >> q) & 1u) {
        uint32_t p = pb + q;
        bool tent = (p == 0) && !((bw >> q) & 1u);
        cp_pos[o] = (t0 + p) | (tent ? (1ull << 63) : 0ull);
        cp_val[o] = base_value(padidx(p));
        ++o;
      }
    }
  }
}

// ---- junctions ---------------------------------------------------------------------------------
// sort key of a junction item: hi = tid : 32 | start : 32, lo = (end - start + 1) : 32 | strand char : 8.  Bundles are
// disjoint coordinate ranges in (tid, start) order, so this is the per-bundle (start, end, strand) order of the
// reference (tiecov.cpp:104, junction flush per bundle) without needing the bundle ids — the branch can start as
// soon as the valid records are known.  The length form keeps the high bytes of `lo` constant (fewer radix passes).
__global__ void junc_fill_k(uint32_t m, CovArrays A, const int32_t* __restrict__ tid, const int32_t* __restrict__ pos,
                            const uint32_t* __restrict__ cig_off, const uint32_t* __restrict__ cig,
                            const uint8_t* __restrict__ strand, const uint32_t* __restrict__ joff, uint64_t* __restrict__ hi,
                            uint64_t* __restrict__ lo, uint32_t* __restrict__ val) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  uint32_t i = A.ridx[j];
  uint32_t o = joff[j];
  uint64_t t = (uint64_t)(uint32_t)tid[i] << 32;
  uint32_t st = strand ? strand[i] : (uint32_t)'.';
  int prev_end = 0;
  int k = 0, nex = 0;
  walk_exons(pos[i], cig + cig_off[i], cig_off[i + 1] - cig_off[i],
             [&](int es, int ee) {
               if (k > 0) {  // CJunc(exons[i-1].end+1, exons[i].start-1, strand) tiecov.cpp:104
                 hi[o] = t | (uint32_t)(prev_end + 1);
                 lo[o] = ((uint64_t)(uint32_t)(es - 1 - prev_end) << 8) | st;  // end - start + 1 (0 for an empty N)
                 val[o] = j;
                 ++o;
               }
               prev_end = ee;
               ++k;
             },
             &nex);
}

// Integral YC: the junction items of a block of consecutive records (reads of one locus repeat the same few introns) are summed
// in an LDS table first, and only one item per distinct junction and block — with its partial sum — goes on to the sort: on
// config 3 some 6 M items become a few hundred thousand.  The table's claim word holds the junction exactly (start relative to
// the block's first record : 24 | length : 31 | strand code : 2 | 1); an item that does not fit that form (another reference
// sequence than the block's first record, a start 2^24 bases on) or finds the table full travels on its own.  Sums are integers
// (the caller has checked every YC is integral and the total below 2^52), so the order of the additions is free.
constexpr uint32_t JA_REC = 4096;   // records per block (16 per thread: a block's fixed cost — clearing and flushing its 16 KB table — is what the
                                    // pass is made of when one record in twelve is spliced; 1024 records per block: 0.36 ms on config 3)
constexpr uint32_t JA_SLOTS = 1024;
__global__ __launch_bounds__(256) void junc_agg_k(uint32_t m, CovArrays A, const int32_t* __restrict__ tid, const int32_t* __restrict__ pos,
                                                  const uint32_t* __restrict__ cig_off, const uint32_t* __restrict__ cig,
                                                  const uint8_t* __restrict__ strand, const double* __restrict__ yc, const uint32_t* __restrict__ jcnt,
                                                  uint64_t* __restrict__ hi, uint64_t* __restrict__ lo, double* __restrict__ pv,
                                                  unsigned long long* __restrict__ n_out) {
  __shared__ unsigned long long claim[JA_SLOTS];
  __shared__ unsigned long long sum[JA_SLOTS];
  const uint32_t j0 = blockIdx.x * JA_REC;
  for (uint32_t q = threadIdx.x; q < JA_SLOTS; q += 256) {
    claim[q] = ~0ull;
    sum[q] = 0;
  }
  const uint32_t i0 = A.ridx[j0];
  const int32_t tid0 = tid[i0], start0 = pos[i0];
  __syncthreads();
  auto emit = [&](uint64_t h, uint64_t l, double v) {  // an item on its own
    const unsigned long long at = atomicAdd(n_out, 1ull);
    hi[at] = h;
    lo[at] = l;
    pv[at] = v;
  };
  for (uint32_t r = 0; r < JA_REC / 256; ++r) {
    const uint32_t j = j0 + r * 256 + threadIdx.x;
    if (j >= m) continue;
    if (!jcnt[j]) continue;
    const uint32_t i = A.ridx[j];
    const uint32_t nc = cig_off[i + 1] - cig_off[i];
    const uint64_t t = (uint64_t)(uint32_t)tid[i] << 32;
    const uint32_t st = strand ? strand[i] : (uint32_t)'.';
    const uint32_t sc = st == '+' ? 0u : (st == '-' ? 1u : (st == '.' ? 2u : 3u));
    const long long y = yc ? (long long)yc[i] : 1ll;
    const bool same_ref = tid[i] == tid0;
    int prev_end = 0, k = 0, nex = 0;
    walk_exons(pos[i], cig + cig_off[i], nc,
               [&](int es, int ee) {
                 if (k > 0) {  // CJunc(exons[i-1].end+1, exons[i].start-1, strand) tiecov.cpp:104
                   const uint32_t js = (uint32_t)(prev_end + 1), len = (uint32_t)(es - 1 - prev_end);
                   const uint64_t rel = (uint64_t)((int64_t)js - (int64_t)start0);
                   bool done = false;
                   if (same_ref && rel < (1ull << 24) && len < (1u << 31) && sc < 3u) {
                     const unsigned long long key = (rel << 34) | ((unsigned long long)len << 3) | (sc << 1) | 1ull;
                     uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 54);  // 10 bits
                     for (uint32_t probe = 0; probe < 32 && !done; ++probe) {
                       unsigned long long cur = *(volatile unsigned long long*)&claim[h];
                       if (cur == ~0ull) {
                         cur = atomicCAS(&claim[h], ~0ull, key);  // (the word found there if another lane was first)
                         if (cur == ~0ull) cur = key;
                       }
                       if (cur == key) {
                         atomicAdd(&sum[h], (unsigned long long)y);
                         done = true;
                       }
                       h = (h + 1) & (JA_SLOTS - 1);
                     }
                   }
                   if (!done) emit(t | js, ((uint64_t)len << 8) | st, (double)y);
                 }
                 prev_end = ee;
                 ++k;
               },
               &nex);
  }
  __syncthreads();
  // flush: one reservation per block
  uint32_t mine = 0;
  for (uint32_t q = threadIdx.x; q < JA_SLOTS; q += 256) mine += claim[q] != ~0ull ? 1u : 0u;
  __shared__ uint32_t s_cnt[4];
  const uint32_t inc = wave_incl_sum(mine);
  if (lane_id() == 63) s_cnt[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t before = inc - mine;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_cnt[w];
  __shared__ unsigned long long s_at;
  if (threadIdx.x == 255) s_at = (before + mine) ? atomicAdd(n_out, (unsigned long long)(before + mine)) : 0ull;
  __syncthreads();
  unsigned long long at = s_at + before;
  for (uint32_t q = threadIdx.x; q < JA_SLOTS; q += 256) {
    const unsigned long long key = claim[q];
    if (key == ~0ull) continue;
    const uint32_t js = (uint32_t)((int64_t)start0 + (int64_t)(key >> 34));
    const uint32_t len = (uint32_t)((key >> 3) & 0x7FFFFFFFull), sc = (uint32_t)((key >> 1) & 3ull);
    hi[at] = ((uint64_t)(uint32_t)tid0 << 32) | js;
    lo[at] = ((uint64_t)len << 8) | (sc == 0 ? (uint32_t)'+' : (sc == 1 ? (uint32_t)'-' : (uint32_t)'.'));
    pv[at] = (double)(long long)sum[q];
    ++at;
  }
}
__global__ void junc_iota_k(uint32_t n, uint32_t* __restrict__ v) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) v[q] = q;
}

__global__ void junc_head_k(uint32_t nj, const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint32_t* __restrict__ head) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nj) return;
  head[q] = (q == 0 || hi[q] != hi[q - 1] || lo[q] != lo[q - 1]) ? 1u : 0u;
}

// one thread per unique junction: sums its members in sorted order = record order (the sort is
// stable and the fill order is record order), exactly the order of CJunc::add (tiecov.cpp:88-90)
__global__ void junc_write_k(uint32_t nj, CovArrays A, const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo,
                             const uint32_t* __restrict__ val, const uint32_t* __restrict__ head, const uint32_t* __restrict__ hoff,
                             const double* __restrict__ yc, uint32_t cap, int32_t* __restrict__ j_tid, int32_t* __restrict__ j_start,
                             int32_t* __restrict__ j_end, uint8_t* __restrict__ j_strand, double* __restrict__ j_val) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nj || !head[q]) return;
  uint32_t o = hoff[q];
  if (o >= cap) return;
  double s = 0.0;
  uint32_t r = q;
  do {
    s += yc ? yc[A.ridx[val[r]]] : 1.0;
    ++r;
  } while (r < nj && !head[r]);
  const int32_t start = (int32_t)(uint32_t)(hi[q] & 0xFFFFFFFFu);
  j_tid[o] = (int32_t)(uint32_t)(hi[q] >> 32);
  j_start[o] = start - 1;
  j_end[o] = start + (int32_t)(uint32_t)(lo[q] >> 8) - 1;
  j_strand[o] = (uint8_t)(lo[q] & 0xFFu);
  j_val[o] = s;
}


// Integral YC (the common case): the order of the additions cannot matter, so the members of a junction are summed in
// parallel — wave-segmented partial sums, one double atomic per (wave, junction) — instead of one thread walking all
// the members of a junction (a junction of a highly expressed gene has tens of thousands).
__global__ void junc_head_write_k(uint32_t nj, const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo,
                                  const uint32_t* __restrict__ head, const uint32_t* __restrict__ hoff, uint32_t cap,
                                  int32_t* __restrict__ j_tid, int32_t* __restrict__ j_start, int32_t* __restrict__ j_end,
                                  uint8_t* __restrict__ j_strand, double* __restrict__ j_val) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nj || !head[q]) return;
  uint32_t o = hoff[q];
  if (o >= cap) return;
  const int32_t start = (int32_t)(uint32_t)(hi[q] & 0xFFFFFFFFu);
  j_tid[o] = (int32_t)(uint32_t)(hi[q] >> 32);
  j_start[o] = start - 1;
  j_end[o] = start + (int32_t)(uint32_t)(lo[q] >> 8) - 1;
  j_strand[o] = (uint8_t)(lo[q] & 0xFFu);
  j_val[o] = 0.0;
}
template <bool PARTIAL /* val indexes partial sums (junc_agg_k) instead of records */>
__global__ void junc_sum_k(uint32_t nj, CovArrays A, const uint32_t* __restrict__ val, const uint32_t* __restrict__ head,
                           const uint32_t* __restrict__ hoff, const double* __restrict__ yc, uint32_t cap, double* __restrict__ j_val) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = q < nj;
  uint32_t o = 0xFFFFFFFFu;
  double v = 0.0;
  if (act) {
    o = hoff[q] + head[q] - 1u;
    v = PARTIAL ? yc[val[q]] : (yc ? yc[A.ridx[val[q]]] : 1.0);
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {  // inclusive segmented sum over lanes holding the same junction
    double pv = __shfl_up(v, d, 64);
    uint32_t po = __shfl_up(o, d, 64);
    if ((int)lane_id() >= d && po == o) v += pv;
  }
  uint32_t no = __shfl_down(o, 1, 64);
  const bool last = lane_id() == 63 || no != o;
  if (act && last && o < cap) atomicAdd(&j_val[o], v);
}


// ---- junctions without a sort (integral YC) -------------------------------------------------------------------------------------------
// junc_agg_k leaves one item per distinct junction and block of JA_REC records; the same junction comes from a few neighbouring
// blocks (the reads that span it start within a read length of each other).  Every item has a HOME block (of JH_REC records): the last block whose first
// record starts at or before the junction's first base, in (reference, start) order.  Equal junctions share their home, and homes are
// in key order, so: count the items per home (jh_home_k), scan, scatter (jh_scatter_k), sort and sum every home's few items in LDS
// (jh_sort_k), scan the numbers of distinct junctions, write (jh_write_k) — no global sort (8 radix passes, 24 launches, for some
// 10^5 items) and no read-back before the end.  A home with more items than a block sorts (a raw, uncollapsed input piled a thousand
// deep) raises scalar 12 and the radix path takes the call.
constexpr uint32_t JH_CAP = 1024;
constexpr uint32_t JH_REC = 1024;  // records per home block (independent of junc_agg_k's blocks: an item finds its home by its key)
__global__ void jh_blockkey_k(uint32_t nblk, CovArrays A, const int32_t* __restrict__ tid, const int32_t* __restrict__ pos, uint64_t* __restrict__ bkey) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const uint32_t i = A.ridx[(size_t)b * JH_REC];
  bkey[b] = ((uint64_t)(uint32_t)tid[i] << 32) | (uint32_t)(pos[i] + 1);  // (reference, 1-based start) of the block's first record
}
// item q (hi = reference : 32 | first base of the junction : 32, the form junc_agg_k writes) -> its home, counted
__global__ void jh_home_k(const unsigned long long* __restrict__ n_items, uint32_t cap_items, const uint64_t* __restrict__ hi, uint32_t nblk,
                          const uint64_t* __restrict__ bkey, uint32_t* __restrict__ home, uint32_t* __restrict__ cnt) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long n = *n_items;
  if (q >= n || q >= cap_items) return;
  const uint64_t k = hi[q];
  uint32_t lo = 0, up = nblk;  // last block with bkey <= k (block 0 when none: the order check of the interval chain reports such input)
  while (up - lo > 1) {
    const uint32_t mid = lo + ((up - lo) >> 1);
    if (bkey[mid] <= k)
      lo = mid;
    else
      up = mid;
  }
  home[q] = lo;
  atomicAdd(&cnt[lo], 1u);
}
__global__ void jh_scatter_k(const unsigned long long* __restrict__ n_items, uint32_t cap_items, const uint64_t* __restrict__ hi,
                             const uint64_t* __restrict__ lo, const double* __restrict__ pv, const uint32_t* __restrict__ home,
                             const uint32_t* __restrict__ off, uint32_t* __restrict__ fill, uint64_t* __restrict__ hi2, uint64_t* __restrict__ lo2,
                             double* __restrict__ pv2) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long n = *n_items;
  if (q >= n || q >= cap_items) return;
  const uint32_t h = home[q];
  const uint32_t at = off[h] + atomicAdd(&fill[h], 1u);
  hi2[at] = hi[q];
  lo2[at] = lo[q];
  pv2[at] = pv[q];
}
// Homes with at most 64 items (nearly all): one wave per home, an item per lane, a bitonic network over the lanes (xor shuffles), equal
// keys summed by a segmented scan (integers held in doubles: any order) — no LDS, no barrier.  The distinct junctions are left at the
// front of the home's segment, their number in ucnt.  Larger homes: jh_sort_big_k.
__global__ __launch_bounds__(256) void jh_sort_k(uint32_t nblk, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                                 uint64_t* __restrict__ hi2, uint64_t* __restrict__ lo2, double* __restrict__ pv2,
                                                 uint32_t* __restrict__ ucnt, uint32_t* __restrict__ big, uint64_t* __restrict__ nbig) {
  const uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (h >= nblk) return;
  const uint32_t n = cnt[h];
  if (n > 64u) {  // listed for jh_sort_big_k
    if (lane_id() == 0) big[atomicAdd((unsigned long long*)nbig, 1ull)] = h;
    return;
  }
  if (n == 0) {
    if (lane_id() == 0) ucnt[h] = 0;
    return;
  }
  const uint32_t base = off[h], l = lane_id();
  uint64_t kh = l < n ? hi2[base + l] : ~0ull, kl = l < n ? lo2[base + l] : ~0ull;
  double kv = l < n ? pv2[base + l] : 0.0;
#pragma unroll
  for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      const uint64_t oh = __shfl_xor(kh, (int)j, 64), ol = __shfl_xor(kl, (int)j, 64);
      const double ov = __shfl_xor(kv, (int)j, 64);
      const bool lower = (l & j) == 0, up = (l & k) == 0;
      const bool gt = kh > oh || (kh == oh && kl > ol);     // mine > partner's
      const bool lt = oh > kh || (oh == kh && ol > kl);     // mine < partner's
      // the lower lane of a pair keeps the smaller key in an ascending run, the larger one in a descending run
      const bool take = lower ? (up ? gt : lt) : (up ? lt : gt);
      if (take) kh = oh, kl = ol, kv = ov;
    }
  const uint64_t ph = __shfl_up(kh, 1, 64), pl = __shfl_up(kl, 1, 64);
  const bool head = l < n && (l == 0 || kh != ph || kl != pl);
  const uint64_t hm = __ballot(head);
  // inclusive segmented sum over the lanes of one junction (the segment of lane l starts at the last head at or before it)
  const uint64_t upto = hm & ((2ull << l) - 1ull);
  const uint32_t seg0 = upto ? 63u - (uint32_t)__builtin_clzll(upto) : 0u;
  double s = kv;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(s, d, 64);
    if (l >= d && l - d >= seg0) s += o;
  }
  const bool last = l < n && (l + 1 == n || ((hm >> (l + 1)) & 1ull));
  if (last) {
    const uint32_t r = (uint32_t)__builtin_popcountll(hm & ((2ull << l) - 1ull)) - 1u;  // rank of this junction among the home's
    hi2[base + r] = kh;
    lo2[base + r] = kl;
    pv2[base + r] = s;
  }
  if (l == 0) ucnt[h] = (uint32_t)__builtin_popcountll(hm);
}
// the homes with more than 64 items (listed by jh_sort_k), a block at a time: sorted by (hi, lo) in LDS (bitonic), equal keys summed
__global__ __launch_bounds__(256) void jh_sort_big_k(const uint32_t* __restrict__ big, const uint64_t* __restrict__ nbig, const uint32_t* __restrict__ cnt,
                                                     const uint32_t* __restrict__ off, uint64_t* __restrict__ hi2, uint64_t* __restrict__ lo2,
                                                     double* __restrict__ pv2, uint32_t* __restrict__ ucnt, uint64_t* __restrict__ overflow, uint32_t cap) {
  __shared__ uint64_t kh[JH_CAP], kl[JH_CAP];
  __shared__ double kv[JH_CAP];
  __shared__ uint32_t sm[8];
  const uint32_t nb_ = (uint32_t)*nbig;
  for (uint32_t bi = blockIdx.x; bi < nb_; bi += gridDim.x) {
  const uint32_t h = big[bi], n = cnt[h];
  if (n > cap) {
    if (threadIdx.x == 0) {
      *overflow = 1;
      ucnt[h] = 0;
    }
    continue;
  }
  const uint32_t base = off[h];
  uint32_t np = 64;
  while (np < n) np <<= 1;
  for (uint32_t q = threadIdx.x; q < np; q += 256) {
    kh[q] = q < n ? hi2[base + q] : ~0ull;
    kl[q] = q < n ? lo2[base + q] : ~0ull;
    kv[q] = q < n ? pv2[base + q] : 0.0;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= np; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t q = threadIdx.x; q < np; q += 256) {
        const uint32_t p = q ^ j;
        if (p > q) {
          const bool up = (q & k) == 0;
          const uint64_t ah = kh[q], al = kl[q], bh = kh[p], bl = kl[p];
          const bool gt = ah > bh || (ah == bh && al > bl);
          if (gt == up) {
            kh[q] = bh, kl[q] = bl, kh[p] = ah, kl[p] = al;
            const double t = kv[q];
            kv[q] = kv[p];
            kv[p] = t;
          }
        }
      }
      __syncthreads();
    }
  // heads, their ranks, their sums
  uint32_t run = 0;
  for (uint32_t q0 = 0; q0 < n; q0 += 256) {
    const uint32_t q = q0 + threadIdx.x;
    const bool head = q < n && (q == 0 || kh[q] != kh[q - 1] || kl[q] != kl[q - 1]);
    uint32_t tot;
    const uint32_t r = run + block_excl_sum<uint32_t, 256>(head ? 1u : 0u, sm, &tot);
    run += tot;
    if (head) {
      double s = kv[q];
      for (uint32_t e = q + 1; e < n && kh[e] == kh[q] && kl[e] == kl[q]; ++e) s += kv[e];
      hi2[base + r] = kh[q];  // (r <= q, and every item of the segment is in LDS: writing the front of the segment is safe)
      lo2[base + r] = kl[q];
      pv2[base + r] = s;
    }
  }
  if (threadIdx.x == 0) ucnt[h] = run;
  __syncthreads();  // (the LDS arrays are loaded again for the block's next home)
  }
}
__global__ void jh_write_k(uint32_t nblk, const uint32_t* __restrict__ ucnt, const uint32_t* __restrict__ uoff, const uint32_t* __restrict__ off,
                           const uint64_t* __restrict__ hi2, const uint64_t* __restrict__ lo2, const double* __restrict__ pv2, uint32_t cap,
                           int32_t* __restrict__ j_tid, int32_t* __restrict__ j_start, int32_t* __restrict__ j_end, uint8_t* __restrict__ j_strand,
                           double* __restrict__ j_val) {
  // a wave per home (most homes hold a handful of junctions)
  const uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (h >= nblk) return;
  const uint32_t n = ucnt[h], o0 = uoff[h], b0 = off[h];
  for (uint32_t u = lane_id(); u < n; u += 64) {
    const uint32_t o = o0 + u;
    if (o >= cap) break;
    const uint64_t kh = hi2[b0 + u], kl = lo2[b0 + u];
    const int32_t start = (int32_t)(uint32_t)(kh & 0xFFFFFFFFu);
    j_tid[o] = (int32_t)(uint32_t)(kh >> 32);
    j_start[o] = start - 1;
    j_end[o] = start + (int32_t)(uint32_t)(kl >> 8) - 1;
    j_strand[o] = (uint8_t)(kl & 0xFFu);
    j_val[o] = pv2[b0 + u];
  }
}

}  // namespace

// =============================================================================================
namespace {
__global__ void sample_convert_k(uint32_t n, const double* __restrict__ v, float denom, int64_t* __restrict__ cnt, float* __restrict__ heat) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long c = (unsigned long long)v[i];
  cnt[i] = (int64_t)c;
  const float mint = 0.1f, maxt = 1.5f;  // normalize(bsam,0.1,1.5,n) tiecov.cpp:316-323, float32 arithmetic
  const float mult = (maxt - mint);
  heat[i] = ((float)c / denom) * mult + mint;
}
}  // namespace

// ---- junction branch ---------------------------------------------------------------------------------------
// items (one per N op of a valid record) -> sort by (tid,start | length,strand) -> heads -> ordered sums.  Independent
// of the interval branch once the valid records are compacted, so tbk_coverage_device runs it on a side context (own stream, own
// arena, own host thread) while the main stream builds the intervals; `ctx` is whichever context it runs on.
constexpr uint32_t COV_SIDE_MIN = 1u << 16;  // below this many records the fork costs more than it hides
static int junc_branch(tbk_ctx* ctx, uint32_t m, const CovArrays& A, const tbk_cov_in* in, const uint32_t* jcnt, tbk_cov_out* out,
                       bool integral, uint64_t nj_known, uint32_t* nj_out, uint32_t* nju_out) {
  const uint32_t B = 256;
  uint64_t* sc = ctx->d_scalars;
  *nj_out = *nju_out = 0;
  const bool agg = integral && !ctx->dbg.no_junc_agg;
  uint32_t* joff = nullptr;
  uint32_t nj;
  if (agg && nj_known != ~0ull) {  // the block sums need no per-record offsets, and the first pass left the total behind
    if (nj_known >= (1ull << 32)) return TBK_E2BIG;
    nj = (uint32_t)nj_known;
  } else {
    joff = ws_alloc<uint32_t>(ctx, m);
    if (!joff) return TBK_ENOMEM;
    TBK_TRY(tbk_exscan_u32(ctx, jcnt, joff, m, sc + 7));
    TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 7, sc + 7, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    nj = (uint32_t)ctx->h_scalars[7];
  }
  *nj_out = nj;
  if (!nj) return 0;
  SortBufs sb;
  sb.hi = ws_alloc<uint64_t>(ctx, nj);
  sb.lo = ws_alloc<uint64_t>(ctx, nj);
  sb.val = ws_alloc<uint32_t>(ctx, nj);
  sb.hi2 = ws_alloc<uint64_t>(ctx, nj);
  sb.lo2 = ws_alloc<uint64_t>(ctx, nj);
  sb.val2 = ws_alloc<uint32_t>(ctx, nj);
  uint32_t* head = ws_alloc<uint32_t>(ctx, nj);
  uint32_t* hoff = ws_alloc<uint32_t>(ctx, nj);
  if (!hoff) return TBK_ENOMEM;
  uint32_t ns = nj;   // items that reach the sort
  double* pv = nullptr;
  if (agg) {  // block-level sums first: one item per distinct junction and block of records
    pv = ws_alloc<double>(ctx, nj);
    if (!pv) return TBK_ENOMEM;
    TBK_HIP(hipMemsetAsync(sc + 11, 0, 3 * sizeof(uint64_t), ctx->stream));  // [11] items, [12] a home too full, [13] homes of more than 64 items
    TBK_LAUNCH(ctx, "junc_agg", junc_agg_k, cdiv(m, JA_REC), 256, 0, m, A, in->tid, in->pos, in->cig_off, in->cig, in->strand, in->yc, jcnt, sb.hi, sb.lo,
               pv, (unsigned long long*)(sc + 11));
    if (!ctx->dbg.junc_radix) {  // (junc_radix: test hook, the sort below)
      // every item to its home block, the homes sorted one by one: no read-back until the junctions are written
      const uint32_t nblk = cdiv(m, JH_REC);
      uint64_t* bkey = ws_alloc<uint64_t>(ctx, nblk);
      uint32_t* hcnt = ws_alloc<uint32_t>(ctx, (size_t)nblk * 2);  // counts | fill cursors
      uint32_t* hoff2 = ws_alloc<uint32_t>(ctx, nblk);
      uint32_t* ucnt = ws_alloc<uint32_t>(ctx, nblk);
      uint32_t* uoff = ws_alloc<uint32_t>(ctx, nblk);
      double* pv2 = ws_alloc<double>(ctx, nj);
      if (!bkey || !hcnt || !hoff2 || !ucnt || !uoff || !pv2) return TBK_ENOMEM;
      uint32_t* hfill = hcnt + nblk;
      TBK_HIP(hipMemsetAsync(hcnt, 0, (size_t)nblk * 2 * 4, ctx->stream));
      TBK_LAUNCH(ctx, "junc_home", jh_blockkey_k, cdiv(nblk, B), B, 0, nblk, A, in->tid, in->pos, bkey);
      TBK_LAUNCH(ctx, "junc_home", jh_home_k, cdiv(nj, B), B, 0, (const unsigned long long*)(sc + 11), nj, sb.hi, nblk, bkey, head /* home */, hcnt);
      TBK_TRY(tbk_exscan_u32(ctx, hcnt, hoff2, nblk, nullptr));
      TBK_LAUNCH(ctx, "junc_home", jh_scatter_k, cdiv(nj, B), B, 0, (const unsigned long long*)(sc + 11), nj, sb.hi, sb.lo, pv, head, hoff2, hfill, sb.hi2,
                 sb.lo2, pv2);
      uint32_t jh_cap = JH_CAP;
      if (ctx->dbg.jh_cap && ctx->dbg.jh_cap < JH_CAP) jh_cap = ctx->dbg.jh_cap;  // test hook: force the fall-back
      uint32_t* big = hoff;  // (nj entries, unused on this path: fewer than nj / 64 homes can hold more than 64 items)
      TBK_LAUNCH(ctx, "junc_sort", jh_sort_k, cdiv((uint64_t)nblk * 64, B), B, 0, nblk, hcnt, hoff2, sb.hi2, sb.lo2, pv2, ucnt, big, sc + 13);
      TBK_LAUNCH(ctx, "junc_sort", jh_sort_big_k, 512, 256, 0, big, sc + 13, hcnt, hoff2, sb.hi2, sb.lo2, pv2, ucnt, sc + 12, jh_cap);
      TBK_TRY(tbk_exscan_u32(ctx, ucnt, uoff, nblk, sc + 9));
      TBK_LAUNCH(ctx, "junc_write", jh_write_k, cdiv((uint64_t)nblk * 64, B), B, 0, nblk, ucnt, uoff, hoff2, sb.hi2, sb.lo2, pv2, out->cap_junctions, out->j_tid,
                 out->j_start, out->j_end, out->j_strand, out->j_val);
      TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 9, sc + 9, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
      TBK_HIP(hipStreamSynchronize(ctx->stream));
      if (!ctx->h_scalars[12]) {
        *nju_out = (uint32_t)ctx->h_scalars[9];
        return tbk_check_launch(ctx, "junctions");
      }
      ns = (uint32_t)ctx->h_scalars[11];  // a home overflowed: the items are still where junc_agg_k left them
    } else {
      TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 11, sc + 11, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
      TBK_HIP(hipStreamSynchronize(ctx->stream));
      ns = (uint32_t)ctx->h_scalars[11];
    }
    TBK_LAUNCH(ctx, "junc_iota", junc_iota_k, cdiv(ns, B), B, 0, ns, sb.val);
  } else {
    TBK_LAUNCH(ctx, "junc_fill", junc_fill_k, cdiv(m, B), B, 0, m, A, in->tid, in->pos, in->cig_off, in->cig, in->strand, joff, sb.hi, sb.lo, sb.val);
  }
  TBK_TRY(tbk_radix_sort128(ctx, &sb, ns));
  TBK_LAUNCH(ctx, "junc_head", junc_head_k, cdiv(ns, B), B, 0, ns, sb.hi, sb.lo, head);
  TBK_TRY(tbk_exscan_u32(ctx, head, hoff, ns, sc + 9));
  if (integral) {
    TBK_LAUNCH(ctx, "junc_head_write", junc_head_write_k, cdiv(ns, B), B, 0, ns, sb.hi, sb.lo, head, hoff, out->cap_junctions, out->j_tid,
               out->j_start, out->j_end, out->j_strand, out->j_val);
    if (agg)
      TBK_LAUNCH(ctx, "junc_sum", junc_sum_k<true>, cdiv(ns, B), B, 0, ns, A, sb.val, head, hoff, pv, out->cap_junctions, out->j_val);
    else
      TBK_LAUNCH(ctx, "junc_sum", junc_sum_k<false>, cdiv(ns, B), B, 0, ns, A, sb.val, head, hoff, in->yc, out->cap_junctions, out->j_val);
  } else {
    TBK_LAUNCH(ctx, "junc_write", junc_write_k, cdiv(nj, B), B, 0, nj, A, sb.hi, sb.lo, sb.val, head, hoff, in->yc, out->cap_junctions,
               out->j_tid, out->j_start, out->j_end, out->j_strand, out->j_val);
  }
  TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 9, sc + 9, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  *nju_out = (uint32_t)ctx->h_scalars[9];
  return tbk_check_launch(ctx, "junctions");
}

struct JuncSide {  // the junction branch on the side context's worker; collected before tbk_coverage_device returns, whatever the path
  TbkWorker* w = nullptr;
  bool th = false;  // posted and not yet collected
  int rc = 0;
  int join() {
    if (th) {
      w->wait();
      th = false;
    }
    return rc;
  }
  ~JuncSide() { (void)join(); }
};

// The lean interval chain (cl_* kernels): bundles, compacted starts, tile tables and spill counts in three passes and a small scan,
// ONE read-back (number of bundles, span, spill pieces, spilling records), then spill fill from the list, the tile kernel, a pass
// over the tiles and the interval writer.  Returns 1 when the tile tables proved too small (the caller then runs the general
// chain), a negative status on errors.  The interval count is left in scalar 8 for the caller's final read-back.
static int cov_intervals_lean(tbk_ctx* ctx, const tbk_cov_in* in, tbk_cov_out* out, CovArrays A, uint32_t m, bool all_valid, uint64_t sum_abs) {
  const uint32_t B = 256;
  uint64_t* sc = ctx->d_scalars;
  const uint32_t cbt = cdiv(m, CB_TILE);
  uint4* cpart = ws_alloc<uint4>(ctx, cbt);
  uint32_t* hcnt = ws_alloc<uint32_t>(ctx, cbt);
  uint32_t* hbase = ws_alloc<uint32_t>(ctx, cbt);
  long long* xsum = ws_alloc<long long>(ctx, cbt);
  long long* xbase = ws_alloc<long long>(ctx, cbt);
  ClChunks L;
  L.base = ws_alloc<uint32_t>(ctx, cbt);
  L.cnt = ws_alloc<uint32_t>(ctx, cbt);
  L.tbase = ws_alloc<uint32_t>(ctx, cbt);
  L.j = ws_alloc<uint32_t>(ctx, m);
  L.cs = ws_alloc<uint64_t>(ctx, m);
  ClTiles T;
  // tiles: unknown until the passes have run.  m / 8 covers an average of 1024 compacted bases per record (a collapsed RNA-seq
  // sample has a handful), 2^20 tiles any small input
  uint64_t cap64 = (uint64_t)m / 8u > (1ull << 20) ? (uint64_t)m / 8u : (1ull << 20);
  if (ctx->dbg.cov_tile_cap) cap64 = ctx->dbg.cov_tile_cap;  // test hook: force the refusal
  T.cap = (uint32_t)cap64;
  T.first = ws_alloc<uint32_t>(ctx, (size_t)T.cap + 1);
  T.tb = ws_alloc<uint32_t>(ctx, (size_t)T.cap + 1);
  T.cnt = ws_alloc<uint32_t>(ctx, (size_t)T.cap + 1);
  if (!cpart || !hcnt || !hbase || !xsum || !xbase || !L.base || !L.cnt || !L.tbase || !L.j || !L.cs || !T.first || !T.tb || !T.cnt) return TBK_ENOMEM;
  TBK_HIP(hipMemsetAsync(T.cnt, 0, ((size_t)T.cap + 1) * 4, ctx->stream));
  uint32_t* done = (uint32_t*)(sc + 16);  // [0] cb_agg_k's blocks, [1] cl_heads_k<false>'s: the last one of each runs the scan over the tiles
  TBK_HIP(hipMemsetAsync(done, 0, sizeof(uint64_t), ctx->stream));
  TBK_LAUNCH(ctx, "cov_bundles", cb_agg_k, cbt, CB_NT, 0, m, A.tid, A.end, cpart, done);
  TBK_LAUNCH(ctx, "cov_bundles", cl_heads_k<false>, cbt, CB_NT, 0, m, A, cpart, hcnt, xsum, (const uint32_t*)nullptr, (const long long*)nullptr, in->cig_off,
             in->cig, T, L, sc, ctx->d_err, done + 1, hbase, xbase);
  TBK_LAUNCH(ctx, "cov_place", cl_heads_k<true>, cbt, CB_NT, 0, m, A, cpart, hcnt, xsum, hbase, xbase, in->cig_off, in->cig, T, L, sc, ctx->d_err,
             (uint32_t*)nullptr, (uint32_t*)nullptr, (long long*)nullptr);
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) return tbk_derr_to_status(ctx, eb);
  if (ctx->h_scalars[CL_SC_REFUSED]) return 1;
  const uint32_t nb = (uint32_t)ctx->h_scalars[CL_SC_NB];
  const uint64_t S = ctx->h_scalars[CL_SC_S], nspill = ctx->h_scalars[CL_SC_NSPILL];
  const uint32_t nsl = (uint32_t)ctx->h_scalars[CL_SC_NSL];
  out->span_bases = S;
  const uint32_t ntiles = (uint32_t)((S + COV_W - 1) / COV_W);  // (< T.cap)
  if (nspill >= (1ull << 32)) return TBK_E2BIG;
  const uint64_t cp_cap64 = 2ull * ((uint64_t)in->n_cigar_ops + nspill) + nb + ntiles + 16;  // <= 2 per M segment piece + 1 per bundle + 1 per tile
  if (cp_cap64 >= (1ull << 32)) return TBK_E2BIG;
  const uint32_t cp_cap = (uint32_t)cp_cap64;
  uint32_t* tile_off = ws_alloc<uint32_t>(ctx, (size_t)ntiles + 1);
  uint32_t* tile_fill = ws_alloc<uint32_t>(ctx, (size_t)ntiles + 1);
  uint32_t* sp_seg = ws_alloc<uint32_t>(ctx, nspill + 1);
  uint32_t* sp_rec = ws_alloc<uint32_t>(ctx, nspill + 1);
  uint64_t* cp_pos = ws_alloc<uint64_t>(ctx, cp_cap);
  double* cp_val = ws_alloc<double>(ctx, cp_cap);
  uint32_t* tile_cp_base = ws_alloc<uint32_t>(ctx, ntiles);
  uint32_t* tile_cp_cnt = ws_alloc<uint32_t>(ctx, ntiles);
  uint32_t* icnt = ws_alloc<uint32_t>(ctx, ntiles);
  uint32_t* ioff = ws_alloc<uint32_t>(ctx, ntiles);
  ClTileOut O;
  O.ecnt = ws_alloc<uint32_t>(ctx, ntiles);
  O.fv = ws_alloc<double>(ctx, ntiles);
  O.lv = ws_alloc<double>(ctx, ntiles);
  uint32_t* cp_alloc = (uint32_t*)(sc + 10);
  if (!tile_off || !tile_fill || !sp_seg || !sp_rec || !cp_pos || !cp_val || !tile_cp_base || !tile_cp_cnt || !icnt || !ioff || !O.ecnt || !O.fv || !O.lv)
    return TBK_ENOMEM;
  TBK_TRY(tbk_exscan_u32(ctx, T.cnt, tile_off, ntiles + 1, nullptr));
  if (nsl) {
    TBK_HIP(hipMemsetAsync(tile_fill, 0, ((size_t)ntiles + 1) * 4, ctx->stream));
    TBK_LAUNCH(ctx, "cov_spill_fill", cl_spill_fill_k, cbt, B, 0, L, A, in->cig_off, in->cig, tile_off, tile_fill, sp_seg, sp_rec);
  }
#define COV_TILE_LAUNCH(ACC, ID)                                                                                                  \
  TBK_LAUNCH(ctx, "cov_tile", (cov_tile_k<ACC, ID, true>), ntiles, COVT_NT, 0, m, S, A, in->cig_off, in->cig, in->yc, T.first, tile_off, \
             sp_seg, sp_rec, cp_pos, cp_val, tile_cp_base, tile_cp_cnt, cp_alloc, cp_cap, ctx->d_err, O)
  const bool small = sum_abs < (1ull << 31);  // int32 accumulators when the depth cannot overflow them
  if (small && all_valid)
    COV_TILE_LAUNCH(int32_t, true);
  else if (small)
    COV_TILE_LAUNCH(int32_t, false);
  else if (all_valid)
    COV_TILE_LAUNCH(long long, true);
  else
    COV_TILE_LAUNCH(long long, false);
#undef COV_TILE_LAUNCH
  TBK_LAUNCH(ctx, "cov_iv_count", cl_iv_count_k, cdiv(ntiles, B), B, 0, ntiles, O, icnt);
  TBK_TRY(tbk_exscan_u32(ctx, icnt, ioff, ntiles, sc + 8));
  TBK_LAUNCH(ctx, "cov_iv_write", cl_iv_emit_k, ntiles, 256, 0, ntiles, S, nb, A, cp_pos, cp_val, tile_cp_base, tile_cp_cnt, O, ioff, T.tb, out->cap_intervals,
             out->iv_tid, out->iv_start, out->iv_end, out->iv_val);
  return 0;
}

static int cov_run(tbk_ctx* ctx, const tbk_cov_in* in, tbk_cov_out* out, bool sample_mode) {
  const uint32_t n = in->n_records;
  const bool want_cov = out->cap_intervals > 0, want_j = out->cap_junctions > 0;
  out->n_intervals = out->n_junctions = 0;
  out->n_bases = 0;
  out->span_bases = 0;
  if (n == 0) return 0;
  const uint32_t B = 256;
  uint64_t* sc = ctx->d_scalars;  // [0]=n_bases [1]=sum|yc| [2]=m [3]=nb [4]=S [5]=nspill [6]=ncp [7]=nj [8]=niv [9]=nju
  TBK_HIP(hipMemsetAsync(sc, 0, 16 * sizeof(uint64_t), ctx->stream));

  uint32_t* valid = ws_alloc<uint32_t>(ctx, n);
  uint32_t* vpos = ws_alloc<uint32_t>(ctx, n);
  CovArrays A;
  A.ridx = ws_alloc<uint32_t>(ctx, n);
  A.start = ws_alloc<int32_t>(ctx, n);
  A.end = ws_alloc<int32_t>(ctx, n);
  A.tid = ws_alloc<int32_t>(ctx, n);
  A.bhead = ws_alloc<uint32_t>(ctx, n);
  A.bid = ws_alloc<uint32_t>(ctx, n);
  A.cs = ws_alloc<uint64_t>(ctx, n);
  A.pk = ws_alloc<uint16_t>(ctx, n);
  A.yi = ws_alloc<int32_t>(ctx, n);
  A.b_tid = ws_alloc<int32_t>(ctx, n);
  A.b_start = ws_alloc<int32_t>(ctx, n);
  A.b_end = ws_alloc<int32_t>(ctx, n);
  A.b_span = ws_alloc<uint32_t>(ctx, n);
  A.b_off = ws_alloc<uint64_t>(ctx, n + 1);
  uint32_t* jcnt = want_j ? ws_alloc<uint32_t>(ctx, n) : nullptr;
  if (!A.b_off || (want_j && !jcnt)) return TBK_ENOMEM;
  uint32_t nj = 0, nju = 0;
  JuncSide side;

  // flag == NULL: every record counts (the device chain hands over representatives, which all passed the collapse
  // filters) — no validity flags, no compaction
  const bool all_valid = in->flag == nullptr;
  if (!all_valid) {
    TBK_LAUNCH(ctx, "cov_valid", cov_valid_k, cdiv(n, B), B, 0, in->flag, n, valid);
    TBK_TRY(tbk_exscan_u32(ctx, valid, vpos, n, sc + 2));
  }
  // the context's own view, built from keys, comes with this pass's results (the view builder had the CIGAR words in hand)
  const auto& V = ctx->view_prep;
  const bool prepared = V.valid && all_valid && !sample_mode && in->mem == TBK_MEM_DEVICE && in->cig == V.cig && n == V.n &&
                        !ctx->dbg.cov_prep;  // (cov_prep: test hook, run the pass anyway)
  uint32_t eb = 0;
  if (prepared) {
    A.ridx = V.ridx;
    A.start = V.start;
    A.end = V.end;
    A.tid = const_cast<int32_t*>(in->tid);
    A.yi = V.yi;
    if (want_j) jcnt = V.jcnt;
    eb = V.err;
    if (!want_cov) eb &= ~(uint32_t)(TBK_DERR_FATALOP | TBK_DERR_NCIGAR);
    ctx->h_scalars[0] = V.n_bases;
    ctx->h_scalars[1] = V.sum_abs;
  } else {
    TBK_LAUNCH(ctx, "cov_prep", cov_prep_k, (cdiv(n, B) < 4096u ? cdiv(n, B) : 4096u), B, 0, n, all_valid ? (const uint32_t*)nullptr : valid, vpos, in->tid, in->pos, in->cig_off, in->cig,
               sample_mode ? (const double*)nullptr : in->yc, want_cov ? 1 : 0, A, jcnt, sc, ctx->d_err);
    TBK_TRY(tbk_sync_err(ctx, &eb));
  }
  const bool fractional = (eb & TBK_DERR_FRACTIONAL) != 0;
  eb &= ~TBK_DERR_FRACTIONAL;
  if (eb) return tbk_derr_to_status(ctx, eb);
  if (fractional && !prepared) TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
  const uint32_t m = all_valid ? n : (uint32_t)ctx->h_scalars[2];
  out->n_bases = ctx->h_scalars[0];
  const uint64_t sum_abs = ctx->h_scalars[1];
  const bool integral = !fractional && !sample_mode && sum_abs < (1ull << 52);  // junction sums may then be formed in any order
  const uint64_t nj_known = prepared ? V.n_junc : (want_j ? ctx->h_scalars[14] : ~0ull);  // junction items: both first passes count them
  if (m == 0) return 0;
  if (want_j && want_cov && m >= COV_SIDE_MIN) {  // the valid records are compacted (the stream was just synchronised): fork the junction branch
    tbk_ctx* jc = tbk_side_ctx(ctx);
    if (jc) {
      const size_t hint = (size_t)m * 8 + (size_t)in->n_cigar_ops * 64 + ((size_t)4 << 20);
      if (!ctx->side_worker) ctx->side_worker = new TbkWorker();
      side.w = ctx->side_worker;
      side.th = true;
      side.w->post([&side, jc, hint, m, &A, in, jcnt, out, integral, nj_known, &nj, &nju]() {
        side.rc = tbk_side_begin(jc, hint);
        if (side.rc == 0) side.rc = junc_branch(jc, m, A, in, jcnt, out, integral, nj_known, &nj, &nju);
        tbk_side_end(jc);
      });
    }
  }

  // integral YC, intervals wanted: the lean chain (TBK_COV_LEGACY: test hook, the general chain below)
  bool lean_done = false;
  if (want_cov && !fractional && !sample_mode && !ctx->dbg.cov_legacy && !ctx->dbg.cov_bundle_scan) {
    const size_t ws_mark = ctx->ws_off;
    const int lrc = cov_intervals_lean(ctx, in, out, A, m, all_valid, sum_abs);
    if (lrc < 0) return lrc;
    lean_done = lrc == 0;
    if (!lean_done) {  // tile tables too small for this input: give the arena back and take the general chain
      if (ctx->ws_overflow.empty()) ctx->ws_off = ws_mark;
      TBK_HIP(hipMemsetAsync(sc + 3, 0, 11 * sizeof(uint64_t), ctx->stream));
    }
  }
  if (!lean_done) {
  // bundles
  {
    if (ctx->dbg.cov_bundle_scan) {  // test hook: the two-stage look-back scan
      BundleLoad ld{A.tid, A.end};
      BundleAux ax{A.tid, A.start};
      BundleHead hd{};
      BundleStore st{A, m, sc + 3, ctx->d_err};
      SegMax ident{INT32_MIN, 0u};
      TBK_TRY((scan_two_run<4, SegMax, SegMaxOp, uint32_t, SoPlusU32, BundleLoad, BundleAux, BundleHead, BundleStore>(ctx, "cov_bundles", m, ld, ax, hd, st, SegMaxOp{}, ident,
                                                                                                          SoPlusU32{}, 0u)));
    } else {
      const uint32_t cbt = cdiv(m, CB_TILE);
      uint4* cpart = ws_alloc<uint4>(ctx, cbt);
      uint32_t* hcnt = ws_alloc<uint32_t>(ctx, cbt);
      uint32_t* hbase = ws_alloc<uint32_t>(ctx, cbt);
      if (!cpart || !hcnt || !hbase) return TBK_ENOMEM;
      uint32_t* done = (uint32_t*)(sc + 16);  // (cb_agg_k's last block scans the tile aggregates)
      TBK_HIP(hipMemsetAsync(done, 0, sizeof(uint64_t), ctx->stream));
      TBK_LAUNCH(ctx, "cov_bundles", cb_agg_k, cbt, CB_NT, 0, m, A.tid, A.end, cpart, done);
      TBK_LAUNCH(ctx, "cov_bundles", cb_heads_k<false>, cbt, CB_NT, 0, m, A, cpart, hcnt, (const uint32_t*)nullptr, sc + 3, ctx->d_err);
      TBK_TRY(tbk_exscan_u32(ctx, hcnt, hbase, cbt, nullptr));
      TBK_LAUNCH(ctx, "cov_bundles", cb_heads_k<true>, cbt, CB_NT, 0, m, A, cpart, hcnt, hbase, sc + 3, ctx->d_err);
    }
  }
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) return tbk_derr_to_status(ctx, eb);
  const uint32_t nb = (uint32_t)ctx->h_scalars[3];
  TBK_LAUNCH(ctx, "cov_bundle_span", cov_bundle_span_k, cdiv(nb, B), B, 0, nb, A);
  TBK_TRY(tbk_exscan_u32_u64(ctx, A.b_span, A.b_off, nb, sc + 4));
  // b_off[nb] = S (device-to-device copy of the total)
  TBK_HIP(hipMemcpyAsync(A.b_off + nb, sc + 4, sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
  TBK_HIP(hipMemcpyAsync(ctx->h_scalars, sc, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  const uint64_t S = ctx->h_scalars[4];
  out->span_bases = S;
  const uint64_t ntiles64 = (S + COV_W - 1) / COV_W;
  if (ntiles64 >= (1ull << 31)) return TBK_E2BIG;
  const uint32_t ntiles = (uint32_t)ntiles64;

  if (want_cov) {
    uint32_t* tile_cnt = ws_alloc<uint32_t>(ctx, ntiles + 1);
    uint32_t* tile_off = ws_alloc<uint32_t>(ctx, ntiles + 1);
    uint32_t* tile_fill = ws_alloc<uint32_t>(ctx, ntiles + 1);
    uint32_t* tile_first = ws_alloc<uint32_t>(ctx, ntiles + 1);
    if (!tile_fill || !tile_first) return TBK_ENOMEM;
    TBK_HIP(hipMemsetAsync(tile_cnt, 0, (size_t)(ntiles + 1) * 4, ctx->stream));
    TBK_HIP(hipMemsetAsync(tile_fill, 0, (size_t)(ntiles + 1) * 4, ctx->stream));
    TBK_LAUNCH(ctx, "cov_cs_count", cov_cs_count_k, cdiv(m, B), B, 0, m, A, in->cig_off, in->cig, tile_cnt, tile_first, ntiles);
    TBK_TRY(tbk_exscan_u32(ctx, tile_cnt, tile_off, ntiles + 1, sc + 5));
    TBK_HIP(hipMemcpyAsync(ctx->h_scalars, sc, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t nspill = ctx->h_scalars[5];
    if (nspill >= (1ull << 32)) return TBK_E2BIG;
    uint32_t* sp_seg = ws_alloc<uint32_t>(ctx, nspill + 1);
    uint32_t* sp_rec = ws_alloc<uint32_t>(ctx, nspill + 1);
    // change points: <= 2 per M segment piece + 1 per bundle + 1 per tile
    const uint64_t cp_cap64 = 2ull * ((uint64_t)in->n_cigar_ops + nspill) + nb + ntiles + 16;
    if (cp_cap64 >= (1ull << 32)) return TBK_E2BIG;
    const uint32_t cp_cap = (uint32_t)cp_cap64;
    uint64_t* cp_pos = ws_alloc<uint64_t>(ctx, cp_cap);
    double* cp_val = ws_alloc<double>(ctx, cp_cap);
    uint64_t* sp = ws_alloc<uint64_t>(ctx, cp_cap);
    double* sv = ws_alloc<double>(ctx, cp_cap);
    uint32_t* tile_cp_base = ws_alloc<uint32_t>(ctx, ntiles);
    uint32_t* tile_cp_cnt = ws_alloc<uint32_t>(ctx, ntiles);
    uint32_t* tile_cp_off = ws_alloc<uint32_t>(ctx, ntiles);
    uint32_t* cp_alloc = (uint32_t*)(sc + 10);
    if (!cp_pos || !cp_val || !sp || !sv || !tile_cp_off) return TBK_ENOMEM;
    if (nspill)
      TBK_LAUNCH(ctx, "cov_spill_fill", cov_spill_fill_k, cdiv(m, B), B, 0, m, A, in->cig_off, in->cig, tile_off, tile_fill,
                 sp_seg, sp_rec);
    if (sample_mode) {
      TBK_LAUNCH(ctx, "sample_tile", (cov_tile_ordered_k<true>), ntiles, COV_NT, 0, m, S, A, in->cig_off, in->cig, in->yc, in->yx,
                 tile_first, tile_off, sp_seg, sp_rec, cp_pos, cp_val, tile_cp_base, tile_cp_cnt, cp_alloc, cp_cap, ctx->d_err);
    } else if (fractional) {
      TBK_LAUNCH(ctx, "cov_tile_ordered", (cov_tile_ordered_k<false>), ntiles, COV_NT, 0, m, S, A, in->cig_off, in->cig, in->yc,
                 in->yx, tile_first, tile_off, sp_seg, sp_rec, cp_pos, cp_val, tile_cp_base, tile_cp_cnt, cp_alloc, cp_cap, ctx->d_err);
    } else {
#define COV_TILE_LAUNCH(ACC, ID)                                                                                                   \
  TBK_LAUNCH(ctx, "cov_tile", (cov_tile_k<ACC, ID, false>), ntiles, COVT_NT, 0, m, S, A, in->cig_off, in->cig, in->yc, tile_first, tile_off, \
             sp_seg, sp_rec, cp_pos, cp_val, tile_cp_base, tile_cp_cnt, cp_alloc, cp_cap, ctx->d_err, ClTileOut{nullptr, nullptr, nullptr})
      const bool small = sum_abs < (1ull << 31);  // int32 accumulators when the depth cannot overflow them
      if (small && all_valid)
        COV_TILE_LAUNCH(int32_t, true);
      else if (small)
        COV_TILE_LAUNCH(int32_t, false);
      else if (all_valid)
        COV_TILE_LAUNCH(long long, true);
      else
        COV_TILE_LAUNCH(long long, false);
#undef COV_TILE_LAUNCH
    }
    TBK_TRY(tbk_exscan_u32(ctx, tile_cp_cnt, tile_cp_off, ntiles, sc + 6));
    TBK_LAUNCH(ctx, "cov_cp_gather", cov_cp_gather_k, ntiles, 64, 0, ntiles, tile_cp_base, tile_cp_cnt, tile_cp_off, cp_pos, cp_val,
               sp, sv);
    TBK_TRY(tbk_sync_err(ctx, &eb));
    if (eb) return tbk_derr_to_status(ctx, eb);
    const uint32_t ncp = (uint32_t)ctx->h_scalars[6];
    if (ncp) {
      const uint32_t ivt = cdiv(ncp, IV_TILE);
      uint32_t* icnt = ws_alloc<uint32_t>(ctx, ivt);
      uint32_t* ioff = ws_alloc<uint32_t>(ctx, ivt);
      uint32_t* tile_b = ws_alloc<uint32_t>(ctx, (size_t)ntiles + 2);
      if (!icnt || !ioff || !tile_b) return TBK_ENOMEM;
      TBK_LAUNCH(ctx, "cov_iv_write", cov_tile_bundle_k, cdiv(ntiles + 1, B), B, 0, ntiles, nb, A.b_off, tile_b);
      TBK_LAUNCH(ctx, "cov_iv_count", iv_count_k, ivt, IV_NT, 0, ncp, sp, sv, icnt);
      TBK_TRY(tbk_exscan_u32(ctx, icnt, ioff, ivt, sc + 8));
      TBK_LAUNCH(ctx, "cov_iv_write", iv_emit_k, ivt, IV_NT, 0, ncp, S, nb, A, sp, sv, ioff, tile_b, out->cap_intervals, out->iv_tid, out->iv_start,
                 out->iv_end, out->iv_val);
    }
  }

  }  // (!lean_done)

  if (want_j && !side.th) TBK_TRY(junc_branch(ctx, m, A, in, jcnt, out, integral, nj_known, &nj, &nju));
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) return tbk_derr_to_status(ctx, eb);
  TBK_TRY(tbk_check_launch(ctx, "coverage"));
  if (side.th) {
    const int jrc = side.join();
    ctx->side_times_pending = true;
    if (jrc != 0) {
      ctx->last_error = ctx->side_ctx->last_error;
      return jrc;
    }
  }
  if (want_cov) {
    out->n_intervals = (uint32_t)ctx->h_scalars[8];
    if (out->n_intervals > out->cap_intervals) return TBK_E2BIG;
  }
  if (want_j && nj) {
    out->n_junctions = nju;
    if (out->n_junctions > out->cap_junctions) return TBK_E2BIG;
  }
  return 0;
}

int tbk_coverage_device(tbk_ctx* ctx, const tbk_cov_in* in, tbk_cov_out* out) { return cov_run(ctx, in, out, false); }

// tiecov -s (addMean / discretize / normalize / flushCoverage(pair), tiecov.cpp:155-185, :277-323)
int tbk_sample_device(tbk_ctx* ctx, const tbk_cov_in* in, int32_t num_samples, tbk_sample_out* out) {
  out->n_intervals = 0;
  if (out->cap_intervals == 0 || !out->iv_tid || !out->iv_start || !out->iv_end || !out->iv_count || !out->iv_heat) return TBK_EINVAL;
  tbk_cov_out co;
  memset(&co, 0, sizeof(co));
  co.mem = TBK_MEM_DEVICE;
  co.cap_intervals = out->cap_intervals;
  co.iv_tid = out->iv_tid;
  co.iv_start = out->iv_start;
  co.iv_end = out->iv_end;
  co.iv_val = ws_alloc<double>(ctx, out->cap_intervals);
  if (!co.iv_val) return TBK_ENOMEM;
  TBK_TRY(cov_run(ctx, in, &co, true));
  out->n_intervals = co.n_intervals;
  if (co.n_intervals) {
    TBK_LAUNCH(ctx, "sample_convert", sample_convert_k, cdiv(co.n_intervals, 256), 256, 0, co.n_intervals, co.iv_val, (float)num_samples,
               out->iv_count, out->iv_heat);
    TBK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return tbk_check_launch(ctx, "sample");
}
