// msort.hip — the record sort of the collapse stage for inputs that arrive as a few position-sorted runs (one per
// input file, which is what `tiebrush` requires of its BAMs: tmerge.cpp keeps one cursor per file and always takes the
// smallest head).  Instead of twelve LSD radix passes over the 128-bit key, the order is built from the structure:
//
//   phase A  merge the runs by position P = hi >> 2 (tid,start) — ceil(log2(runs)) rounds of pairwise merge-path
//            merges, stable (left run first on ties), so equal P stay in input (file, index) order;
//   phase B  inside every bucket of equal P order by (strand, span, hash) — buckets are short (a few reads share a
//            start), so a block takes a window of consecutive buckets into LDS and ranks each record by counting.
//
//            A bucket longer than a window (thousands of reads starting on one base) is sorted by one block with a
//            stable LSD radix sort over its varying key bytes (msort_big_k).
//
// The result is the same permutation the stable 128-bit radix sort produces.  All integer work, HBM/LDS-bound; no MFMA.
#include "dev_common.hpp"
#include "tbk_internal.h"

namespace {
constexpr int MG_NT = 256;
constexpr int MG_E = 8;
constexpr int MG_T = MG_NT * MG_E;  // outputs per tile

// run boundaries of round r: run q of the round is input runs [q << r, (q + 1) << r)
__device__ __forceinline__ uint32_t run_start(const uint32_t* __restrict__ run_off, uint32_t k, uint32_t r, uint32_t q) {
  uint64_t f = (uint64_t)q << r;
  return run_off[f < k ? (uint32_t)f : k];
}

// Merge path, left run first on ties: the number of elements the first d outputs of merge(A, B) take from A.
// All 64 lanes of a wave call it; every round probes 64 split points at once, so a run of a million elements takes
// four rounds of (two parallel) global loads instead of twenty dependent ones.
__device__ __forceinline__ uint32_t mpath_wave(const uint64_t* __restrict__ A, uint32_t nA, const uint64_t* __restrict__ B, uint32_t nB,
                                               uint32_t d) {
  uint32_t lo = d > nB ? d - nB : 0u;  // g(lo) holds by construction
  uint32_t hi = d < nA ? d : nA;
  const uint32_t lane = lane_id();
  while (lo < hi) {  // invariant: g(lo) true, answer in [lo, hi];  g(i) = A[i-1] <= B[d-i]  (monotone: true ... true false ... false)
    const uint32_t span = hi - lo;
    const uint32_t step = (span + 63u) / 64u;
    uint64_t p64 = (uint64_t)lo + (uint64_t)(lane + 1u) * step;
    const uint32_t p = p64 < hi ? (uint32_t)p64 : hi;
    const bool g = (A[p - 1] >> 2) <= (B[d - p] >> 2);
    const uint64_t t = __ballot(g);
    const int c = t == ~0ull ? 64 : __builtin_ctzll(~t);  // leading run of true probes
    uint32_t nlo = lo, nhi = hi;
    if (c > 0) nlo = (uint32_t)__shfl((int)p, c - 1, 64);
    if (c < 64) nhi = (uint32_t)__shfl((int)p, c, 64) - 1u;
    lo = nlo;
    hi = nhi < nlo ? nlo : nhi;
  }
  return lo;
}

// one round: every pair of adjacent runs of round r becomes one run of round r + 1
__global__ __launch_bounds__(MG_NT) void msort_merge_k(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo,
                                                       const uint32_t* __restrict__ val, uint64_t* __restrict__ ohi,
                                                       uint64_t* __restrict__ olo, uint32_t* __restrict__ oval,
                                                       const uint32_t* __restrict__ run_off, uint32_t k, uint32_t r) {
  // (the grid covers the host's upper bound of the record count; the count itself is the end of the last run)
  const uint32_t n = run_off[k];
  if ((uint64_t)blockIdx.x * MG_T >= n) return;
  __shared__ uint64_t keys[MG_T];
  __shared__ uint32_t src[MG_T];
  __shared__ uint32_t s_i[2];
  const uint32_t o0 = blockIdx.x * MG_T;
  const uint32_t o1 = (n - o0) < (uint32_t)MG_T ? n : o0 + MG_T;
  const uint32_t npairs = (uint32_t)((((uint64_t)k + (1ull << (r + 1)) - 1) >> (r + 1)));
  // first pair that reaches beyond o0
  uint32_t j;
  {
    uint32_t a = 0, b = npairs;  // smallest j with run_start(2(j+1)) > o0
    while (a < b) {
      uint32_t mid = (a + b) >> 1;
      if (run_start(run_off, k, r, 2 * (mid + 1)) > o0)
        b = mid;
      else
        a = mid + 1;
    }
    j = a;
  }
  uint32_t o = o0;
  while (o < o1 && j < npairs) {
    const uint32_t ps = run_start(run_off, k, r, 2 * j), pm = run_start(run_off, k, r, 2 * j + 1), pe = run_start(run_off, k, r, 2 * j + 2);
    if (pe <= o) {
      ++j;
      continue;
    }
    const uint32_t seg_end = pe < o1 ? pe : o1;
    const uint64_t* A = hi + ps;
    const uint64_t* Bp = hi + pm;
    const uint32_t nA = pm - ps, nB = pe - pm;
    const uint32_t d0 = o - ps, d1 = seg_end - ps;
    if (threadIdx.x < 128) {  // wave 0: lower diagonal, wave 1: upper diagonal
      const uint32_t w = threadIdx.x >> 6;
      const uint32_t i = mpath_wave(A, nA, Bp, nB, w == 0 ? d0 : d1);
      if (lane_id() == 0) s_i[w] = i;
    }
    __syncthreads();
    const uint32_t i0 = s_i[0], i1 = s_i[1];
    const uint32_t b0 = d0 - i0, b1 = d1 - i1;
    const uint32_t na = i1 - i0, nb = b1 - b0, L = na + nb;  // L == d1 - d0 <= MG_T
    for (uint32_t x = threadIdx.x; x < L; x += MG_NT) keys[x] = (x < na ? A[i0 + x] : Bp[b0 + (x - na)]) >> 2;
    __syncthreads();
    // every thread merges MG_E consecutive outputs of the segment
    {
      const uint32_t dl = threadIdx.x * MG_E < L ? threadIdx.x * MG_E : L;
      uint32_t a = dl > nb ? dl - nb : 0u, b = dl < na ? dl : na;
      while (a < b) {  // largest i with keys[i-1] <= keysB[dl-i]
        uint32_t mid = (a + b + 1) >> 1;
        if (keys[mid - 1] <= keys[na + dl - mid])
          a = mid;
        else
          b = mid - 1;
      }
      uint32_t ia = a, ib = dl - a;
#pragma unroll
      for (int e = 0; e < MG_E; ++e) {
        const uint32_t x = dl + e;
        if (x < L) {
          const bool takeA = ib >= nb || (ia < na && keys[ia] <= keys[na + ib]);
          src[x] = takeA ? ps + i0 + ia : pm + b0 + ib;
          if (takeA)
            ++ia;
          else
            ++ib;
        }
      }
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < L; x += MG_NT) {  // striped: coalesced stores, two nearly contiguous gather streams
      const uint32_t s = src[x];
      ohi[o + x] = hi[s];
      olo[o + x] = lo[s];
      oval[o + x] = val[s];
    }
    __syncthreads();
    o = seg_end;
    ++j;
  }
}

constexpr int RF_NT = 256;
constexpr int RF_W = 1024;    // positions a block owns (it sorts the buckets whose first record lies among them)
constexpr int RF_CAP = 2048;  // window: the owned positions plus the rest of the last owned bucket
constexpr int RF_E = RF_CAP / RF_NT;

// phase B.  Input sorted by P (stable); output: inside each bucket ordered by (strand, lo), equal keys in input order.
__global__ __launch_bounds__(RF_NT) void msort_refine_k(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo,
                                                        const uint32_t* __restrict__ val, uint64_t* __restrict__ ohi,
                                                        uint64_t* __restrict__ olo, uint32_t* __restrict__ oval,
                                                        const uint32_t* __restrict__ pm, uint32_t* __restrict__ big,
                                                        uint32_t* __restrict__ nbig) {
  const uint32_t m = *pm;
  if ((uint64_t)blockIdx.x * RF_W >= m) return;
  __shared__ uint64_t K[RF_CAP];
  __shared__ uint8_t head[RF_CAP];
  __shared__ uint16_t bs[RF_CAP], bend[RF_CAP];
  __shared__ int s_wmax[RF_NT / 64];
  __shared__ uint32_t s_first, s_end, s_lasthead;
  const uint32_t r0 = blockIdx.x * RF_W;
  const uint32_t r1 = (m - r0) < (uint32_t)RF_W ? m : r0 + RF_W;
  const uint32_t lim = (m - r1) < (uint32_t)(RF_CAP - RF_W) ? m : r1 + (RF_CAP - RF_W);
  if (threadIdx.x == 0) {
    s_first = 0xFFFFFFFFu;
    s_end = 0xFFFFFFFFu;
    s_lasthead = 0u;
  }
  __syncthreads();
  {
    uint32_t fmin = 0xFFFFFFFFu, emin = 0xFFFFFFFFu, lmax = 0u;
    for (uint32_t i = r0 + threadIdx.x; i < lim; i += RF_NT) {
      const bool h = i == 0 || (hi[i] >> 2) != (hi[i - 1] >> 2);
      if (h) {
        if (i < r1) {
          if (i < fmin) fmin = i;
          if (i > lmax) lmax = i;
        } else if (i < emin) {
          emin = i;
        }
      }
    }
    if (fmin != 0xFFFFFFFFu) {
      atomicMin(&s_first, fmin);
      atomicMax(&s_lasthead, lmax);
    }
    if (emin != 0xFFFFFFFFu) atomicMin(&s_end, emin);
  }
  __syncthreads();
  const uint32_t ws = s_first;
  if (ws == 0xFFFFFFFFu) return;  // the whole range lies inside a bucket that started earlier: its owner handles it (or hands it on)
  uint32_t we = s_end;
  if (we == 0xFFFFFFFFu) {
    if (lim == m) {
      we = m;
    } else {  // the last owned bucket does not end inside the window: msort_big_k takes it, the earlier ones stay here
      we = s_lasthead;
      if (threadIdx.x == 0) big[atomicAdd(nbig, 1u)] = we;
      if (we == ws) return;
    }
  }
  const uint32_t nw = we - ws;  // <= RF_CAP
  uint64_t rh[RF_E], rl[RF_E];
  uint32_t rv[RF_E];
#pragma unroll
  for (int e = 0; e < RF_E; ++e) {
    const uint32_t x = (uint32_t)e * RF_NT + threadIdx.x;
    if (x < nw) {
      rh[e] = hi[ws + x];
      rl[e] = lo[ws + x];
      rv[e] = val[ws + x];
      K[x] = ((rh[e] & 3ull) << 62) | rl[e];  // span < 2^30 (checked when the keys are built): strand fits above it
      head[x] = (x == 0 || (rh[e] >> 2) != (hi[ws + x - 1] >> 2)) ? 1 : 0;
    }
  }
  __syncthreads();
  // bucket bounds of every window position: bs[x] = first position of x's bucket (running max of the head positions,
  // blocked ownership for the scan), bend[first position] = one past the bucket's last position
  {
    const uint32_t p0 = threadIdx.x * RF_E;
    int loc[RF_E];
    int cur = -1;
#pragma unroll
    for (int e = 0; e < RF_E; ++e) {
      const uint32_t p = p0 + e;
      if (p < nw && head[p]) cur = (int)p;
      loc[e] = cur;
    }
    int inc = cur;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int o = __shfl_up(inc, d, 64);
      if ((int)lane_id() >= d && o > inc) inc = o;
    }
    if (lane_id() == 63) s_wmax[threadIdx.x >> 6] = inc;
    __syncthreads();
    int pre = __shfl_up(inc, 1, 64);
    if (lane_id() == 0) pre = -1;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w)
      if (s_wmax[w] > pre) pre = s_wmax[w];
#pragma unroll
    for (int e = 0; e < RF_E; ++e) {
      const uint32_t p = p0 + e;
      if (p < nw) bs[p] = (uint16_t)(loc[e] >= 0 ? loc[e] : pre);
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < RF_E; ++e) {
    const uint32_t x = (uint32_t)e * RF_NT + threadIdx.x;
    if (x < nw) {
      if (x > 0 && head[x]) bend[bs[x - 1]] = (uint16_t)x;
      if (x == nw - 1) bend[bs[x]] = (uint16_t)nw;
    }
  }
  __syncthreads();
  // rank by counting inside the bucket; the bounds are known, so the LDS reads of a loop are independent and pipeline
#pragma unroll
  for (int e = 0; e < RF_E; ++e) {
    const uint32_t x = (uint32_t)e * RF_NT + threadIdx.x;
    if (x < nw) {
      const uint64_t kx = K[x];
      const uint32_t b0 = bs[x], b1 = bend[b0];
      // rank = records of the bucket with a smaller key, plus equal ones that come earlier.  With kx1 = kx + 1 for the
      // earlier part both halves are one "K[j] < bound" test: K[j] <= kx  <=>  K[j] < kx + 1 (kx + 1 cannot wrap: the
      // top two bits of a key are a strand code <= 2).
      uint32_t c0 = 0, c1 = 0;
      const uint64_t kx1 = kx + 1;
      uint32_t j = b0;
      for (; j + 7 < b1; j += 8) {
        uint64_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = K[j + u];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          c0 += kk[u] < (j + u < x ? kx1 : kx) ? 1u : 0u;
          c1 += kk[u + 1] < (j + u + 1 < x ? kx1 : kx) ? 1u : 0u;
        }
      }
      for (; j + 1 < b1; j += 2) {
        const uint64_t ka = K[j], kb = K[j + 1];
        c0 += ka < (j < x ? kx1 : kx) ? 1u : 0u;
        c1 += kb < (j + 1 < x ? kx1 : kx) ? 1u : 0u;
      }
      for (; j < b1; ++j) c0 += K[j] < (j < x ? kx1 : kx) ? 1u : 0u;
      const uint32_t dst = ws + b0 + c0 + c1;
      ohi[dst] = rh[e];
      olo[dst] = rl[e];
      oval[dst] = rv[e];
    }
  }
}


// ---- buckets too long for a refine window ---------------------------------------------------------------------
__device__ __forceinline__ uint64_t match_digit8(uint32_t d, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    bool bit = (d >> b) & 1u;
    uint64_t bal = __ballot(valid && bit);
    peers &= bit ? bal : ~bal;
  }
  return peers;
}

constexpr uint32_t MS_BIG_MAX = 1u << 17;  // records one block still sorts; beyond that the tile goes to the radix sort

// One block per long bucket (thousands of reads starting on one base: rRNA, mitochondrial genes, deep amplicons): a
// stable block-level LSD radix sort of the bucket by K = (strand, span, hash), 8-bit digits, only over digits that
// vary inside the bucket, ping-ponging between the refine input and output ranges of the bucket.
__global__ __launch_bounds__(256) void msort_big_k(uint64_t* __restrict__ hi, uint64_t* __restrict__ lo, uint32_t* __restrict__ val,
                                                   uint64_t* __restrict__ ohi, uint64_t* __restrict__ olo, uint32_t* __restrict__ oval,
                                                   const uint32_t* __restrict__ pm, const uint32_t* __restrict__ big,
                                                   const uint32_t* __restrict__ nbig, uint32_t grid_cap, uint32_t* __restrict__ err) {
  const uint32_t nb_ = *nbig;
  if (nb_ > grid_cap && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, TBK_DERR_BIGBUCKET);
  if (blockIdx.x >= nb_) return;
  __shared__ uint32_t s_be;
  __shared__ unsigned long long s_and, s_or;
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wcnt[4][256];
  __shared__ uint32_t sm[8];
  const uint32_t m = *pm;
  const uint32_t b0 = big[blockIdx.x];
  const uint64_t P0 = hi[b0] >> 2;
  if (threadIdx.x == 0) {
    s_be = m;
    s_and = ~0ull;
    s_or = 0ull;
  }
  __syncthreads();
  for (uint32_t base = b0 + 1; base < m; base += 256) {  // end of the bucket
    const uint32_t i = base + threadIdx.x;
    const bool diff = i < m && (hi[i] >> 2) != P0;
    if (diff) atomicMin(&s_be, i);
    if (__syncthreads_or(diff ? 1 : 0)) break;
  }
  __syncthreads();
  const uint32_t n = s_be - b0;
  if (n > MS_BIG_MAX && n > m / 64) {  // one block would hold up the whole (small) tile: let the caller's whole-tile radix sort (all CUs) have it.
    // In a large tile a deep pile-up is one of many (config 3 at full size: ~10^5 buckets beyond a window, the deepest 5 x 10^5
    // records of 3 x 10^8) and the blocks of this kernel run side by side, so it stays here.
    if (threadIdx.x == 0) atomicOr(err, TBK_DERR_BIGBUCKET);
    return;
  }
  {
    uint64_t a = ~0ull, o = 0ull;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
      const uint64_t k = ((hi[b0 + i] & 3ull) << 62) | lo[b0 + i];
      a &= k;
      o |= k;
    }
    atomicAnd(&s_and, (unsigned long long)a);
    atomicOr(&s_or, (unsigned long long)o);
  }
  __syncthreads();
  const uint64_t vary = s_and ^ s_or;
  uint64_t *sh = hi + b0, *sl = lo + b0, *dh = ohi + b0, *dl = olo + b0;
  uint32_t *sv = val + b0, *dv = oval + b0;
  bool in_src_side = true;  // where the current order lives: the refine-input side (true) or the output side
  const uint32_t w = threadIdx.x >> 6;
  for (uint32_t shift = 0; shift < 64; shift += 8) {
    if (((vary >> shift) & 0xFFull) == 0) continue;
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
      const uint64_t k = ((sh[i] & 3ull) << 62) | sl[i];
      atomicAdd(&hist[(uint32_t)(k >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    {
      uint32_t tot;
      const uint32_t c = hist[threadIdx.x];
      const uint32_t ex = block_excl_sum<uint32_t, 256>(c, sm, &tot);
      hist[threadIdx.x] = ex;  // running output base of digit threadIdx.x
    }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n; c0 += 256) {  // chunks in order: stable
      const uint32_t i = c0 + threadIdx.x;
      const bool valid = i < n;
      uint64_t h = 0, l = 0;
      uint32_t v = 0, d = 0;
      if (valid) {
        h = sh[i];
        l = sl[i];
        v = sv[i];
        d = (uint32_t)((((h & 3ull) << 62) | l) >> shift) & 0xFFu;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) wcnt[q][threadIdx.x] = 0;
      __syncthreads();
      const uint64_t peers = match_digit8(d, valid);
      const uint32_t rank = (uint32_t)__builtin_popcountll(peers & lanemask_lt());
      if (valid && rank == 0) wcnt[w][d] = (uint32_t)__builtin_popcountll(peers);
      __syncthreads();
      if (valid) {
        uint32_t off = hist[d] + rank;
        for (uint32_t q = 0; q < w; ++q) off += wcnt[q][d];
        dh[off] = h;
        dl[off] = l;
        dv[off] = v;
      }
      __syncthreads();
      hist[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
      __syncthreads();
    }
    uint64_t* th = sh;
    sh = dh;
    dh = th;
    uint64_t* tl = sl;
    sl = dl;
    dl = tl;
    uint32_t* tv = sv;
    sv = dv;
    dv = tv;
    in_src_side = !in_src_side;
  }
  if (in_src_side) {  // an even number of passes (or none): the order lives on the input side, the result belongs on the other
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
      dh[i] = sh[i];
      dl[i] = sl[i];
      dv[i] = sv[i];
    }
  }
}
}  // namespace

// Sort b (nruns runs, run f = [run_off[f], run_off[f+1]), each non-decreasing in hi >> 2 and in input order; the record
// count run_off[nruns] is read on the device, n_hi is the host's upper bound of it and only sizes the grids) exactly
// as tbk_radix_sort128 would.  `nbig` is a device word the caller has zeroed (it counts the buckets too long for a
// phase-B window; those are sorted by msort_big_k).  More than 262144 of them, or one of more than 2^17 records that is also more than 1/64 of the tile, set
// TBK_DERR_BIGBUCKET in *err; the caller then swaps b's two sides back (still a valid stable input) and runs the radix
// sort on it.
int tbk_sort_runs(tbk_ctx* ctx, SortBufs* b, uint32_t n_hi, const uint32_t* d_run_off, uint32_t nruns, uint32_t* err,
                  uint32_t* nbig) {
  const uint32_t n = n_hi;
  if (n == 0) return 0;
  uint32_t rounds = 0;
  while ((1u << rounds) < nruns) ++rounds;
  const uint32_t tiles = cdiv(n, MG_T);
  for (uint32_t r = 0; r < rounds; ++r) {
    TBK_LAUNCH(ctx, "msort_merge", msort_merge_k, tiles, MG_NT, 0, b->hi, b->lo, b->val, b->hi2, b->lo2, b->val2, d_run_off, nruns, r);
    std::swap(b->hi, b->hi2);
    std::swap(b->lo, b->lo2);
    std::swap(b->val, b->val2);
  }
  const uint32_t big_cap = n / (RF_CAP - RF_W) + 1;  // a long bucket holds more than a window's overhang
  uint32_t* big = ws_alloc<uint32_t>(ctx, big_cap);
  if (!big) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, "msort_refine", msort_refine_k, cdiv(n, RF_W), RF_NT, 0, b->hi, b->lo, b->val, b->hi2, b->lo2, b->val2, d_run_off + nruns,
             big, nbig);
  // (their number is only known on the device: launch for a generous bound, surplus blocks leave at once; more long
  // buckets than that — a pathological tile — raise TBK_DERR_BIGBUCKET and the caller falls back to the radix sort)
  const uint32_t big_grid = big_cap < 262144u ? big_cap : 262144u;
  TBK_LAUNCH(ctx, "msort_big", msort_big_k, big_grid, 256, 0, b->hi, b->lo, b->val, b->hi2, b->lo2, b->val2, d_run_off + nruns, big, nbig,
             big_grid, err);
  std::swap(b->hi, b->hi2);  // (on TBK_DERR_BIGBUCKET the caller swaps back: the *2 side then holds the phase-A output)
  std::swap(b->lo, b->lo2);
  std::swap(b->val, b->val2);
  return tbk_check_launch(ctx, "sort_runs");
}
