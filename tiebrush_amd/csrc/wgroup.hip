// wgroup.hip — the "window path" of the collapse stage on gfx950: from the k position-sorted input files straight to the
// collapsed groups, every record read once, nothing sorted.
//
// What it replaces (reference: /root/reference/src/tiebrush.cpp:477-499 addPData — find-or-insert of a read into the
// sorted group list of its (tid,start) bucket — tiebrush.cpp:532-541 passes_options, and tmerge.cpp:331-344, the k-way merge
// that feeds them): the sort path (collapse.hip + msort.hip) computes a key per record, scans the merge priority, compacts,
// orders all m records by the 128-bit key and derives the groups from adjacent records.  Deep data makes that wasteful: config 3
// collapses 320 M records into 25 M groups, and a pile-up of 10^5 reads on one base is sorted only to find its few dozen
// distinct alignments.
//
// Here the coordinate axis is cut into windows by splitters chosen from a sample of the files' (tid, pos) keys (every s-th
// record; every g-th sorted sample is a splitter; a splitter value that repeats — a pile-up — gets a window of its own).  Each
// file contributes one contiguous piece to a window (wg_offsets_stream_k: one streaming pass, which also checks that the files
// are sorted), so a window's records are k coalesced reads.  One workgroup per window (wg_hash_window): the records stream
// through once, pieces laid end to end; per record the filter verdict and the 128-bit key come from its raw fields
// (record_key, strategy.hpp), the merge priority — the per-file running maximum of the read ends inside a run of equal
// starts — from a segmented prefix maximum along the piece (DPP inside a wave, LDS across waves, a carry across chunks).  The
// passing records go into an LDS hash table keyed by a seeded 64-bit fingerprint of the key, the key stored beside it and
// compared by every record that lands on the slot; per group: wave-aggregated count, bitset of the samples seen, atomic
// minimum of (priority, record) for the representative.  The distinct groups are ranked by key at the end and written with
// their (group, sample) incidences.  Windows with more distinct groups than the table holds go to a second tier with a larger
// table, then to the LDS sort kernel (at most WG_CAP records, by construction of the splitters: index permutation
// merge-sorted by (key, load order), groups from blocked runs).  Every record whose key word is a hash (not an exact code,
// strategy.hpp) is compared with a member of its group under the exact strategy key (a collision raises TBK_DERR_COLLISION
// and the host reseeds), so grouping is exact.  A scan over the per-window counts and a compaction pass put groups and
// incidences in key order.  What cannot be handled (more distinct groups in a pile-up than the larger table holds, k > 1024)
// raises TBK_DERR_BIGBUCKET and the tile takes the sort path; input the raw form does not take (an inversion, a mapped read
// without a reference) raises TBK_DERR_RAWORDER and the general front end decides.  The compacted form (chi / clo / cval /
// ceff prepared by col_keys_k + col_effkey_scan) runs through the same kernels (RAW = false).
//
// All integer work.  The first-tier kernel is bound by vector-instruction issue (record decode, probing, atomics: about 450
// VALU instructions per 64 records), not by its 13.6 GB of HBM traffic; see DESIGN.md §3.  No MFMA.
#include <stdlib.h>

#include <algorithm>

#include "dev_common.hpp"
#include "strategy.hpp"
#include "tbk_internal.h"
#include "wgroup.h"

namespace {

constexpr int WG_NT = 512;               // threads per window block
constexpr int WG_E = 8;                  // records per thread in the LDS path
constexpr int WG_CAP = WG_NT * WG_E;     // 4096 records sorted in LDS
constexpr uint32_t WG_T = 1024;          // records between splitters: a window holds < 2 T + k s <= WG_CAP records
constexpr uint32_t WG_KS = 2048;         // k * s budget (2 T + k s = WG_CAP)
constexpr int WG_NW = WG_NT / 64;
constexpr int WG_R = 4;                  // records per thread and chunk in the pile-up path
constexpr int WG_RS = 1;                 // raw form: records per thread decoded together
constexpr int WG_RR = 1;                 // raw form: records per thread and chunk.  One, not two: the kernel waits on its dependent loads
                                         // (piece -> fields -> CIGAR words) most of the time and what hides them is resident waves — at one
                                         // record per thread it fits 64 VGPRs: eight waves per SIMD instead of six, i.e. four blocks of 512 threads per CU
                                         // where there were four of 384 (config 3: 10.2 -> 8.3 ms).  Fewer instructions (a leaner fingerprint, no strategy hash, one
                                         // 16-byte load for six fields) had changed nothing.

// ---- partition ------------------------------------------------------------------------------------------------
__global__ void wg_sample_k(const uint64_t* __restrict__ chi, uint32_t m, uint32_t s, uint32_t ns, uint64_t* __restrict__ shi) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  uint64_t d = (uint64_t)j * s;
  shi[j] = chi[d < m ? d : m - 1] >> 2;
}

// The samples of k SORTED runs (sample j = key of record j * s; the runs lie end to end) need no sort to find the splitters: a sample's
// rank among all samples is its index inside its own run plus, for every other run, the number of that run's samples before it — a
// bisection per run over the run's samples where they lie (ties: the earlier run first, so every rank is taken once).  The samples whose
// rank is a multiple of g are the ones wg_split_k reads: Z[q] = the sample of rank q * g.  (The owner's side of the multi-rank protocol:
// one launch where the radix sort of the samples took ten)
__global__ __launch_bounds__(256) void pr_sample_rank_k(const uint64_t* __restrict__ chi, uint32_t m, uint32_t s, uint32_t ns,
                                                        const uint32_t* __restrict__ run_off, uint32_t k, uint32_t g, uint64_t* __restrict__ Z) {
  __shared__ uint32_t j0[64 + 1];  // first sample of run r (k <= PR_MAXRUNS = 64); j0[k] = ns
  if (threadIdx.x <= k) j0[threadIdx.x] = threadIdx.x == k ? ns : (uint32_t)(((uint64_t)run_off[threadIdx.x] + s - 1u) / s);
  __syncthreads();
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  const uint64_t K = chi[(uint64_t)j * s] >> 2;  // (j * s < m: ns = ceil(m / s))
  uint32_t r = 0;  // the run sample j lies in: last r with j0[r] <= j (empty runs repeat a value)
  for (uint32_t q = 1; q < k; ++q) r = j0[q] <= j ? q : r;
  uint32_t rank = j - j0[r];
  for (uint32_t q = 0; q < k; ++q) {
    if (q == r) continue;
    uint32_t lo = j0[q], hi = j0[q + 1];  // first sample of run q that is not before K (before: <, and = for the runs ahead of r)
    const uint32_t b = lo;
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      const uint64_t v = chi[(uint64_t)mid * s] >> 2;
      if (q < r ? v <= K : v < K)
        lo = mid + 1;
      else
        hi = mid;
    }
    rank += lo - b;
  }
  if (rank % g == 0) Z[rank / g] = K;
}

// Y sorted samples; splitter i = Y[(i + 1) * g].  Two bounds per splitter: first of a run of equal splitters -> (v, v + 1 if
// the run is longer than one else v); the others -> (v + 1, v + 1).  Equal consecutive bounds make empty windows.
__global__ void wg_split_k(const uint64_t* __restrict__ Y, uint32_t g, uint32_t nsp, uint64_t* __restrict__ W) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsp) return;
  const uint64_t v = Y[(uint64_t)(i + 1) * g];
  const bool has_prev = i > 0, has_next = i + 1 < nsp;
  const uint64_t pv = has_prev ? Y[(uint64_t)i * g] : 0, nv = has_next ? Y[(uint64_t)(i + 2) * g] : 0;
  const bool first = !has_prev || pv != v;
  const bool heavy = (has_prev && pv == v) || (has_next && nv == v);
  W[2 * i] = first ? v : v + 1;
  W[2 * i + 1] = (first && !heavy) ? v : v + 1;
}

// ---- raw mode: the windows are cut on the records as they come (no key pass, no compaction) -----------------------------------
// Partition key of a raw record: (tid, pos) with every tid < 0 (unplaced reads, the tail of a sorted BAM) folded onto one value
// beyond all references.  A file whose raw keys never decrease is coordinate-sorted in the sense of col_effkey_scan (mapped
// records: key = (tid + 1, pos + 1); unmapped ones carry start 0 and can neither raise a running maximum nor split a run of
// equal starts), its effective ends are prefix maxima of `end` inside runs of equal raw key, and a window bounded by raw keys
// holds every record of its (tid, start) buckets.  Anything else (an inversion, tid >= 0 with pos < 0) raises TBK_DERR_RAWORDER
// and the tile takes the general path, which decides what is an error.
__device__ __forceinline__ uint64_t raw_key(int tid, int pos) {
  return tid < 0 ? (1ull << 62) : (((uint64_t)(uint32_t)tid << 31) | ((uint32_t)pos & 0x7FFFFFFFu));
}
// The partition (sampling, then the offsets pass) streams the (tid, pos) pairs twice, and in a coordinate-sorted file the reference
// id changes a handful of times: ctid[c] is the id every record of chunk c (WG_TC consecutive records) carries when the chunk lies
// inside one file and its first and last record agree — then so does everything between them, PROVIDED the file is sorted, which the
// window kernels check record by record against the ids as they are (wg_hash_window: a record before its predecessor raises
// TBK_DERR_RAWORDER) — and WG_TC_MIXED otherwise (the partition then reads the ids of that chunk).  Two loads per 4096 records
// instead of 4096: the partition reads 1.3 GB of positions twice instead of 2.6 GB of pairs (config 3).
constexpr uint32_t WG_TC = 4096;
constexpr int32_t WG_TC_MIXED = INT32_MIN;
__global__ void wg_tidchunks_k(const int32_t* __restrict__ tid, uint32_t n, const uint32_t* __restrict__ run_off, uint32_t k, uint32_t nchunks,
                               int32_t* __restrict__ ctid) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const uint32_t a = c * WG_TC, b = n - a < WG_TC ? n - 1 : a + WG_TC - 1;  // first and last record of the chunk
  uint32_t lo = 0, hi = k;  // last f with run_off[f] <= a
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (run_off[mid] <= a)
      lo = mid;
    else
      hi = mid;
  }
  const int32_t ta = tid[a];
  ctid[c] = (b < run_off[lo + 1] && tid[b] == ta) ? ta : WG_TC_MIXED;
}
__global__ void wg_sample_raw_k(const int32_t* __restrict__ tid, const int32_t* __restrict__ pos, const int32_t* __restrict__ ctid, uint32_t n,
                                uint32_t s, uint32_t ns, uint64_t* __restrict__ shi) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  uint64_t d = (uint64_t)j * s;
  if (d >= n) d = n - 1;
  const int32_t ct = ctid[d / WG_TC];
  shi[j] = raw_key(ct == WG_TC_MIXED ? tid[d] : ct, pos[d]);
}
// The offsets matrix off[r * k + f] (r = 0: start of run f, r = nrows - 1: its end, else the first record of run f whose key is
// >= W[r - 1]) by a streaming pass (per-(row, run) searches cost 2-4x as much in scattered probes):
// a block takes WG_OC consecutive records, keeps their partition keys in LDS, and for every run segment inside it finds by two
// searches in W the bounds that fall between the key before the segment and its last key — each of those has its answer inside
// the segment (one LDS bisection).  The first segment of a run takes every bound up to its first key, the last one every bound
// beyond its keys (answer: the run's end), so each (bound, run) pair is written exactly once when the runs are sorted.  RAW: the
// same pass checks that they are (an inversion raises TBK_DERR_RAWORDER: the caller must not run the window kernels on the
// matrix).  Rows 0 / last and the columns of empty runs: wg_offsets_edges_k.
constexpr uint32_t WG_OC = 2048;
__device__ __forceinline__ uint32_t wg_upper_bound(const uint64_t* __restrict__ W, uint32_t n, uint64_t v) {  // first r with W[r] > v
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (W[mid] <= v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
// A block takes WG_OG consecutive chunks.  Inside a run the bounds of consecutive chunks follow one another: the lower end of a chunk's
// range in W is the upper end of the chunk before it, and the upper end lies a few bounds further (2.3 per chunk on config 3) — ONE
// round of 256 probes behind the lower end finds it, where the search over all of W takes three dependent rounds; those probes and the
// next chunk's positions are asked for while the current chunk's bisections run, so a chunk no longer waits for memory at all (one
// chunk per block, each behind its own four round trips, was 0.83 ms for 1.3 GB of positions).
constexpr uint32_t WG_OG = 8;
template <bool RAW>
__global__ __launch_bounds__(256) void wg_offsets_stream_k(const uint64_t* __restrict__ chi, const int32_t* __restrict__ rtid,
                                                           const int32_t* __restrict__ rpos, const int32_t* __restrict__ ctid,
                                                           const uint32_t* __restrict__ run_off, uint32_t k,
                                                           uint32_t n, const uint64_t* __restrict__ W, uint32_t nW, uint32_t nrows,
                                                           uint32_t* __restrict__ offT, uint32_t* __restrict__ err, uint32_t og /* chunks per block, <= WG_OG */) {
  __shared__ uint64_t key[WG_OC];
  __shared__ uint32_t s_cnt[2][4];  // (two parities: one barrier per round)
  __shared__ uint32_t s_loc[2][4];
  __shared__ uint32_t s_run[1024 + 1];  // the runs' offsets (k <= 1024: tbk_window_supported): read once per block, not once per chunk and step
  const uint32_t t = threadIdx.x;
  for (uint32_t f = t; f <= k; f += 256) s_run[f] = run_off[f];
  // Two upper bounds in W at once (first r with W[r] > v), one per half of the block: every round the 128 threads of a half probe
  // evenly spaced bounds of the interval left — three rounds for any nW < 2^21, where a bisection by one thread waits on ~20
  // dependent loads (that wait, not the writes, was most of this kernel).  Both halves return both answers.
  // (the parity of the count words goes on from call to call: a call that began at parity 0 again could overwrite the words a slower wave
  // is still reading in the last round of the call before — two searches in a row happen wherever a chunk holds pieces of several runs)
  uint32_t par = 0;
  auto upper_bounds2 = [&](uint64_t va, bool skip_a, uint64_t vb, bool skip_b, uint32_t* ra, uint32_t* rb) {
    const uint32_t half = t >> 7, j = t & 127u;
    uint32_t lo[2] = {0u, 0u}, hi[2] = {skip_a ? 0u : nW, skip_b ? 0u : nW};
    while (lo[0] < hi[0] || lo[1] < hi[1]) {  // (uniform)
      const uint32_t l = lo[half], h = hi[half];
      const uint32_t step = (h - l + 127u) >> 7;
      const uint32_t p = l + j * step;
      const bool le = step && p < h && W[p] <= (half ? vb : va);
      const uint64_t m = __ballot(le);
      if ((t & 63u) == 0) s_cnt[par][t >> 6] = (uint32_t)__builtin_popcountll(m);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint32_t T = s_cnt[par][2 * q] + s_cnt[par][2 * q + 1];
        const uint32_t st = (hi[q] - lo[q] + 127u) >> 7;
        if (lo[q] < hi[q]) {
          if (T == 0) {
            hi[q] = lo[q];
          } else {
            const uint32_t pT = lo[q] + T * st;  // the first probe that was beyond, or beyond the interval
            hi[q] = T < 128u && pT < hi[q] ? pT : hi[q];
            lo[q] = lo[q] + (T - 1u) * st + 1u;
            lo[q] = lo[q] > hi[q] ? hi[q] : lo[q];
          }
        }
      }
      par ^= 1u;
    }
    *ra = lo[0];
    *rb = lo[1];
  };
  static_assert(WG_TC % WG_OC == 0, "a chunk of the offsets pass lies inside one chunk of reference ids");
  constexpr uint32_t KPT = WG_OC / 256;  // keys per thread
  bool bad = false;
  // carried from chunk to chunk: the run the last segment belonged to, where it ended (record) and its upper end in W; the probes
  // W[c_rhi + t] asked for ahead (wprobe), and the next chunk's positions (npos)
  uint32_t c_f = ~0u, c_end = 0, c_rhi = 0;
  uint64_t wprobe = 0, c_klast = 0;  // (c_klast: the last key of that segment = the key before the one that continues it)
  int32_t npos[KPT];
  const uint32_t chunk0 = blockIdx.x * og;
  {
    const uint32_t i0 = chunk0 * WG_OC;
#pragma unroll
    for (uint32_t u = 0; u < KPT; ++u) npos[u] = (RAW && i0 + u * 256u + t < n) ? rpos[i0 + u * 256u + t] : 0;
  }
  for (uint32_t g = 0; g < og; ++g) {
    const uint32_t i0 = (chunk0 + g) * WG_OC;
    if (i0 >= n) break;  // (uniform)
    const uint32_t i1 = n - i0 < WG_OC ? n : i0 + WG_OC;
    if (g) __syncthreads();  // (the keys of the chunk before are no longer read)
    if constexpr (RAW) {
      const int32_t ct = ctid[i0 / WG_TC];  // (uniform) the chunk's reference id, or: read them
#pragma unroll
      for (uint32_t u = 0; u < KPT; ++u) {
        const uint32_t j = u * 256u + t;
        if (j < i1 - i0) key[j] = raw_key(ct != WG_TC_MIXED ? ct : rtid[i0 + j], npos[u]);
      }
      const uint32_t n0 = i0 + WG_OC;  // the next chunk's positions: on their way while this chunk is worked on
      if (g + 1 < og) {
#pragma unroll
        for (uint32_t u = 0; u < KPT; ++u) npos[u] = n0 + u * 256u + t < n ? rpos[n0 + u * 256u + t] : 0;
      }
    } else {
      for (uint32_t j = t; j < i1 - i0; j += 256) key[j] = chi[i0 + j] >> 2;
    }
    __syncthreads();  // (the keys; in the first chunk the runs' offsets too)
    uint32_t f_first;  // last f with run_off[f] <= i0 (every thread for itself: the table is in LDS)
    {
      uint32_t lo = 0, hi = k;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_run[mid] <= i0)
          lo = mid;
        else
          hi = mid;
      }
      f_first = lo;
    }
    uint32_t seg = 0;  // segments of this chunk so far (the parity of the probe counts' LDS words)
    for (uint32_t f = f_first; f < k; ++f) {
      const uint32_t a = s_run[f], b = s_run[f + 1];
      if (a >= i1) break;
      if (b <= i0 || a == b) continue;
      const uint32_t sa = a > i0 ? a : i0, sb = b < i1 ? b : i1;  // the run's records inside this chunk
      const bool has_lo = sa > a;                                  // (then sa == i0: the record before is outside the chunk)
      const bool cont = has_lo && c_f == f && c_end == sa;         // it continues the segment the chunk before ended with
      const uint64_t klo = !has_lo ? 0ull : cont ? c_klast : (RAW ? raw_key(rtid[sa - 1], rpos[sa - 1]) : (chi[sa - 1] >> 2));
      const uint64_t khi = key[sb - 1 - i0];
      uint32_t r_lo, r_hi;
      bool found = false, have_v = false;
      if (cont && sb == b) {  // ... and the run ends here: every bound that is left
        r_lo = c_rhi;
        r_hi = nW;
        found = true;
      } else if (cont) {
        // the segment continues the one the chunk before ended with: its range starts where that one's ended, and ends within the
        // 256 bounds behind it unless the chunk spans more of them (then: the search over all of W)
        r_lo = c_rhi;
        const bool le = r_lo + t < nW && wprobe <= khi;  // wprobe = W[r_lo + t], asked for a chunk ago
        const uint64_t mb = __ballot(le);
        uint32_t* sl = s_loc[(g + seg) & 1u];  // (a chunk has one such segment — its first —, consecutive chunks alternate)
        if ((t & 63u) == 0) sl[t >> 6] = (uint32_t)__builtin_popcountll(mb);
        __syncthreads();
        const uint32_t c = sl[0] + sl[1] + sl[2] + sl[3];
        if (c < 256u) {
          r_hi = r_lo + c;
          found = true;
          have_v = true;  // thread t < c holds bound r_lo + t in wprobe
        }
      }
      ++seg;
      if (!found) {
        upper_bounds2(klo, !has_lo, khi, sb == b, &r_lo, &r_hi);
        r_hi = sb == b ? nW : r_hi;
      }
      const uint64_t v_mine = wprobe;
      if (sb != b) {  // the run goes on in the next chunk: its probes
        c_f = f;
        c_end = sb;
        c_rhi = r_hi;
        c_klast = khi;
        wprobe = r_hi + t < nW ? W[r_hi + t] : ~0ull;
      } else {
        c_f = ~0u;
      }
      for (uint32_t r = r_lo + t; r < r_hi; r += 256) {
        const uint64_t v = have_v ? v_mine : W[r];  // (have_v: r_hi - r_lo < 256, one bound per thread at most)
        uint32_t lo = sa - i0, hi = sb - i0;  // first record of the segment with key >= v (none: the run ends here)
        while (lo < hi) {
          const uint32_t mid = lo + ((hi - lo) >> 1);
          if (key[mid] < v)
            lo = mid + 1;
          else
            hi = mid;
        }
        offT[(size_t)f * nrows + (r + 1)] = i0 + lo;  // (consecutive r: consecutive words)
      }
      if (RAW) {
        for (uint32_t j = sa + t; j < sb; j += 256) {
          const uint64_t pk = j > sa ? key[j - 1 - i0] : klo;
          bad |= pk > key[j - i0];
        }
      }
    }
  }
  if (RAW && bad) atomicOr(err, TBK_DERR_RAWORDER);
}
// (one block per run)
__global__ void wg_offsets_edges_k(const uint32_t* __restrict__ run_off, uint32_t k, uint32_t nrows, uint32_t* __restrict__ offT) {
  const uint32_t f = blockIdx.x;
  const uint32_t a = run_off[f], b = run_off[f + 1];
  uint32_t* col = offT + (size_t)f * nrows;
  if (threadIdx.x == 0) {
    col[0] = a;
    col[nrows - 1] = b;
  }
  if (a == b)
    for (uint32_t r = 1 + threadIdx.x; r + 1 < nrows; r += blockDim.x) col[r] = a;
}
// The stream kernel finds, for one run at a time, consecutive bounds: it writes the matrix run-major (offT[f * nrows + r], whole
// lines — written bound-major, every word dirtied a line of its own: 1.28 GB of writes for 92 MB of matrix on config 3), and this
// pass turns it into the bound-major form the window kernels read a row of per window, 64 x 64 words at a time through LDS.
__global__ __launch_bounds__(256) void wg_offsets_transpose_k(const uint32_t* __restrict__ offT, uint32_t k, uint32_t nrows, uint32_t* __restrict__ off) {
  __shared__ uint32_t tile[64][65];
  const uint32_t r0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
  const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (uint32_t y = ty; y < 64; y += 4) {  // tile[y][x] = offT[f0 + y][r0 + x]
    const uint32_t f = f0 + y, r = r0 + tx;
    if (f < k && r < nrows) tile[y][tx] = offT[(size_t)f * nrows + r];
  }
  __syncthreads();
  for (uint32_t y = ty; y < 64; y += 4) {  // off[r0 + y][f0 + x] = tile[x][y]
    const uint32_t r = r0 + y, f = f0 + tx;
    if (f < k && r < nrows) off[(size_t)r * k + f] = tile[tx][y];
  }
}
struct WgRaw {                 // raw mode: the window kernels read the records themselves (WgIn::chi / clo / cval / ceff are null)
  ColIn I;
  ColOpt O;
  unsigned long long* n_pass;  // passing records (one atomic per block)
  unsigned long long* n_slots; // records that left a slot in cslot (0: wg_finish_raw_k has nothing to do)
  uint32_t all_slots;          // 1: every passing record leaves its group slot in cslot (record -> group map wanted); 0: only the
                               // records whose key word is not exact (they are verified by wg_finish_raw_k)
  uint32_t sparse;             // (all_slots == 0) 1: those few records are listed per window — entry e of window w is record
                               // vsrc[wbase + e] with slot cslot[wbase + e], vcnt[w] entries — instead of marked in a cslot array that has
                               // to be cleared and searched record by record (1.3 GB each way on config 3 for 3 % of the records)
};
// The window kernels count passing records and verification entries with one atomic per block: ~ 180 K blocks on one word would
// keep one L2 channel busy for milliseconds (a word takes ~ 88 atomics per microsecond), so the counts go to WG_NSPREAD words a cache
// line apart and wg_spread_sum_k folds them into the two scalars afterwards.
constexpr uint32_t WG_NSPREAD = 64, WG_SPREAD_STRIDE = 16;  // (u64 words: 128 bytes apart)
__device__ __forceinline__ void wg_count(unsigned long long* base, unsigned long long v) {
  atomicAdd(base + (size_t)(blockIdx.x & (WG_NSPREAD - 1u)) * WG_SPREAD_STRIDE, v);
}
// Inclusive prefix maximum of x inside segments (f = 1: a segment starts at this lane), wave wide, DPP only.  Returns the scanned
// value; *fo = 1 when a segment start lies at or before this lane (the carry from earlier waves does not reach it).
__device__ __forceinline__ uint32_t wave_seg_max(uint32_t x, uint32_t f, uint32_t* fo) {
#define WG_SEG_STEP(ctrl, rm)                                                                  \
  {                                                                                            \
    const uint32_t xo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rm, 0xf, false); \
    const uint32_t fq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, ctrl, rm, 0xf, false); \
    x = f ? x : (xo > x ? xo : x);                                                             \
    f |= fq;                                                                                   \
  }
  WG_SEG_STEP(0x111, 0xf)
  WG_SEG_STEP(0x112, 0xf)
  WG_SEG_STEP(0x114, 0xf)
  WG_SEG_STEP(0x118, 0xf)
  WG_SEG_STEP(0x142, 0xa)
  WG_SEG_STEP(0x143, 0xc)
#undef WG_SEG_STEP
  *fo = f;
  return x;
}
// The raw fields of one record of a window piece, loaded in two batches so that the loads of several records are in flight
// together (A: the fixed fields, the CIGAR range and — where the lane to the left is not the record before — that record's
// position; B: the first three CIGAR words), then turned into the key, the filter verdict, the raw partition key and the value
// the record feeds into the effective-end scan.  Records that are not there load record 0 (harmless, ignored).
struct RawA {
  int32_t pos, tidv, ppos, ptid;
  uint32_t fl_mq_sc;  // flag : 16 | mapq : 8 | strand code : 2
  int32_t nh;
  uint32_t c0, nc;
};
struct RawRec {
  uint64_t hi, lo, rk, prk;
  uint32_t x;     // end of a mapped record, 0 otherwise
  uint32_t err;
  bool pass;
};
__device__ __forceinline__ RawA wg_raw_a(const ColIn& I, uint32_t i, bool need_prev) {
  RawA a;
  a.pos = I.pos[i];
  a.tidv = I.tid[i];
  a.fl_mq_sc = (uint32_t)I.flag[i] | ((uint32_t)I.mapq[i] << 16) | (strand_code(I.strand[i]) << 24);
  a.nh = I.nh[i];
  a.c0 = I.cig_off[i];
  a.nc = I.cig_off[i + 1] - a.c0;
  const uint32_t j = need_prev && i > 0 ? i - 1 : i;
  a.ppos = I.pos[j];
  a.ptid = I.tid[j];
  return a;
}
// ... with the record's CIGAR range already at hand (fetched a chunk ahead, wg_hash_window)
__device__ __forceinline__ RawA wg_raw_a(const ColIn& I, uint32_t i, bool need_prev, uint32_t c0, uint32_t c1) {
  RawA a;
  a.pos = I.pos[i];
  a.tidv = I.tid[i];
  a.fl_mq_sc = (uint32_t)I.flag[i] | ((uint32_t)I.mapq[i] << 16) | (strand_code(I.strand[i]) << 24);
  a.nh = I.nh[i];
  a.c0 = c0;
  a.nc = c1 - c0;
  const uint32_t j = need_prev && i > 0 ? i - 1 : i;
  a.ppos = I.pos[j];
  a.ptid = I.tid[j];
  return a;
}
__device__ __forceinline__ CigView wg_raw_b(const ColIn& I, const RawA& a) {
  CigView c;
  const uint32_t* safe = I.cig_off;  // (always readable)
  c.w0 = *(a.nc > 0 ? I.cig + a.c0 : safe);
  c.w1 = *(a.nc > 1 ? I.cig + a.c0 + 1 : safe);
  c.w2 = *(a.nc > 2 ? I.cig + a.c0 + 2 : safe);
  c.p = I.cig + a.c0;
  return c;
}
// The same fields as they come from memory, nothing derived from a loaded value: a record's loads can be issued a chunk ahead and
// carried across the probes and the barrier without a wait (wg_hash_window, PIPE)
struct RawL {
  int32_t pos, tidv, ppos, ptid, nh;
  uint32_t c0, nc, w0, w1, w2;
  uint16_t flag;  // (in the width they are loaded in: widening them is an instruction on the loaded value, and waits for it)
  uint8_t mapq, strand;
};
template <class T>
__device__ __forceinline__ T wg_ld(const T* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(base) + byte_off);
}
// (cols: bit 0 — a filter looks at NH (-N below its maximum), bit 1 — one looks at MAPQ (-Q above zero); a column no filter refers to
// is not read: 5 of a record's 16 bytes under the default options)
__device__ __forceinline__ RawL wg_raw_l(const ColIn& I, uint32_t i, bool need_prev, uint32_t c0, uint32_t c1, uint32_t cols = 3u) {
  RawL a;
  // byte offsets of 32 bits from the arrays' bases, which are uniform: one shift serves every 4-byte array, and the loads take the base
  // from scalar registers (the raw form runs on tiles of < 2^30 records and CIGAR words: tbk_window_groups)
  const uint32_t o4 = i << 2;
  a.pos = wg_ld(I.pos, o4);
  a.tidv = wg_ld(I.tid, o4);
  a.flag = wg_ld(I.flag, i << 1);
  a.mapq = 255;
  a.nh = 0;
  if (cols & 2u) a.mapq = wg_ld(I.mapq, i);  // (uniform)
  a.strand = wg_ld(I.strand, i);
  if (cols & 1u) a.nh = wg_ld(I.nh, o4);
  a.c0 = c0;
  a.nc = c1 - c0;
  const uint32_t p4 = need_prev && i > 0 ? o4 - 4u : o4;
  a.ppos = wg_ld(I.pos, p4);
  a.ptid = wg_ld(I.tid, p4);
  const uint32_t q4 = c0 << 2;  // (a record without the word reads word 0 of the array: always there, tbk_window_groups)
  a.w0 = wg_ld(I.cig, a.nc > 0 ? q4 : 0u);
  a.w1 = wg_ld(I.cig, a.nc > 1 ? q4 + 4u : 0u);
  a.w2 = wg_ld(I.cig, a.nc > 2 ? q4 + 8u : 0u);
  return a;
}
template <int ST>
__device__ __forceinline__ RawRec wg_raw_c(const WgRaw& R, uint32_t i, const RawA& a, const CigView& c) {
  const uint32_t fl = a.fl_mq_sc & 0xFFFFu;
  const RecKey K = record_key<ST>(R.I, R.O, i, fl, a.pos, a.tidv, (int)((a.fl_mq_sc >> 16) & 0xFFu), a.nh, a.fl_mq_sc >> 24, c, a.nc);
  RawRec r;
  r.hi = K.hi;
  r.lo = K.lo;
  r.pass = K.pass;
  r.err = K.err;
  r.rk = raw_key(a.tidv, a.pos);
  r.prk = raw_key(a.ptid, a.ppos);
  const bool mapped = !(fl & 0x4u);
  r.x = mapped && K.end > 0 ? (uint32_t)K.end : 0u;
  if ((a.tidv >= 0 && a.pos < 0) || (mapped && a.tidv < 0)) r.err |= TBK_DERR_RAWORDER;
  return r;
}

// ---- window kernel -----------------------------------------------------------------------------------------------
struct WgIn {
  const uint64_t *chi, *clo;   // compacted passing records: k runs, each non-decreasing in chi >> 2
  const uint32_t* cval;        // original record index
  const uint32_t* ceff;        // effective end of the k-way merge (cross-rank tiles: the explicit merge-order priority instead)
  const uint32_t* off;         // [(nw + 1) * k]
  const uint64_t* W;           // [nw - 1] bounds (window w = [W[w-1], W[w]))
  uint32_t k, nw;
  uint32_t rank_merge;         // TBK_WG_RANK_MERGE (test hook): rank a window's groups by the merge sort instead of the buckets
};
struct WgTemp {                // per window, at the window's record base
  uint64_t *hi, *lo;           // group key
  uint32_t *cnt, *ns, *poff;   // members, samples, first incidence (window-local)
  unsigned long long* rep;     // min (effend << 32 | record)
  uint32_t* pinc;              // incidence: sample | window-local group << 16
  uint32_t *wg_cnt, *wp_cnt;   // per window: groups, incidences
  uint32_t* wbase;             // [nw + 1] per window: record base (wg_rowsum_k)
  uint64_t* fmask;             // (<= 64 files, the YD stage places its items by list: tbk_yd_by_list) the set of a group's files as a
                               // bit mask instead of the incidence list pinc: nothing downstream walks incidences then
  uint32_t* vsrc;              // sparse verification list (WgRaw::sparse): the records ...
  uint32_t* vcnt;              // ... and their number per window
  uint32_t* cslot;             // [compacted record] window base + number of the record's group inside the window ...
  uint32_t* c2r;               // ... and from that number to the group's temp slot (window base + rank by key)
  uint32_t *yx, *yd;           // PART only: per group sum of the carried YX, maximum of the carried YD
  unsigned long long* dbg;     // optional [32]: cycles / blocks / records per block kind (TBK_WG_DEBUG)
};

__device__ __forceinline__ bool key_less(const uint64_t* hi, const uint64_t* lo, uint32_t a, uint64_t bh, uint64_t bl, uint32_t b) {
  const uint64_t ah = hi[a];
  if (ah != bh) return ah < bh;
  const uint64_t al = lo[a];
  if (al != bl) return al < bl;
  return a < b;
}

template <class T>
__device__ __forceinline__ T wg_block_excl(T v, T* sm /*WG_NW*/, T* total) {
  T inc = wave_incl_sum(v);
  const uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 63) sm[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < WG_NW; ++i) {
    T x = sm[i];
    if ((uint32_t)i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// file of item e: last f with pre[f] <= e  (pre has k + 1 entries, pre[k] = n_w; empty pieces repeat a value)
__device__ __forceinline__ uint32_t piece_of(const uint32_t* pre, uint32_t k, uint32_t e) {
  uint32_t lo = 0, hi = k;  // answer in [lo, hi)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (pre[mid] <= e)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

constexpr uint32_t WG_LDS_MAIN = WG_CAP * (8 + 8 + 4 + 2 + 2);  // sort kernel: hi, lo, val, two index permutations = 72 KiB
constexpr uint32_t WG_LDS_HASH = 40448;                        // hash kernel: the group table (four blocks of eight waves per CU)
constexpr uint32_t WG_LDS_HASH2 = 64 * 1024;                   // second tier (two blocks per CU)

// Stable merge sort of the index permutation `src` (n entries, ping-pong with `dst`; returns where the result lives) by
// (hi[idx], lo[idx], idx).  In round `len` every element finds its rank in the sibling run by a bisection without branches:
// each thread keeps up to WG_E bisections in flight, every probe is issued (clamped when it falls outside the sibling run),
// so the dependent LDS reads (index, then its key words) of the chains overlap.
template <int E>
__device__ __forceinline__ uint16_t* wg_merge_sort(uint16_t* src, uint16_t* dst, uint32_t n, const uint64_t* hi, const uint64_t* lo,
                                                   uint32_t lim /* indices are < lim */) {
  const uint32_t t = threadIdx.x;
  const int eu = (int)((n + WG_NT - 1) / WG_NT);  // item slots in use (uniform)
  // the rounds are branch-free, so a wave executes all of them whether or not it holds items: a wave whose items all lie
  // beyond n (most waves of a deep window: ~ 80 groups for 512 threads) only keeps the barriers
  const bool live = (t & ~63u) < n;
  for (uint32_t len = 1; len < n; len <<= 1) {
    if (live) {
      uint32_t idx[E], oa[E], osz[E], cntv[E], dbase[E];
      uint64_t kh[E], kl[E];
#pragma unroll
      for (int u = 0; u < E; ++u) {
        const uint32_t e = t + (uint32_t)u * WG_NT;
        cntv[u] = osz[u] = oa[u] = idx[u] = dbase[u] = 0;
        kh[u] = kl[u] = 0;
        if (u < eu && e < n) {
          const uint32_t base = e & ~(2 * len - 1);
          const uint32_t mid = base + len < n ? base + len : n;
          const uint32_t end = base + 2 * len < n ? base + 2 * len : n;
          idx[u] = src[e];
          kh[u] = hi[idx[u]];
          kl[u] = lo[idx[u]];
          if (e < mid) {  // left run: + elements of the right run below me
            oa[u] = mid;
            osz[u] = end - mid;
            dbase[u] = e;
          } else {  // right run: + elements of the left run below me
            oa[u] = base;
            osz[u] = mid - base;
            dbase[u] = base + (e - mid);
          }
        }
      }
      for (uint32_t st = len; st > 0; st >>= 1) {
        uint32_t o[E];
        uint64_t oh[E], ol[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
          if (u < eu) {
            const uint32_t c = cntv[u] + st;
            const uint32_t pos = oa[u] + (c <= osz[u] ? c : 1u) - 1u;
            o[u] = src[pos < n ? pos : 0u];
          }
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
          if (u < eu) {
            const uint32_t oo = o[u] < lim ? o[u] : 0u;
            oh[u] = hi[oo];
            ol[u] = lo[oo];
          }
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
          if (u < eu) {
            const uint32_t c = cntv[u] + st;
            const bool less = (oh[u] < kh[u]) | ((oh[u] == kh[u]) & ((ol[u] < kl[u]) | ((ol[u] == kl[u]) & (o[u] < idx[u]))));
            cntv[u] = (c <= osz[u] && less) ? c : cntv[u];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < E; ++u) {
        const uint32_t e = t + (uint32_t)u * WG_NT;
        if (u < eu && e < n) dst[dbase[u] + cntv[u]] = (uint16_t)idx[u];
      }
    }
    __syncthreads();
    uint16_t* tmp = src;
    src = dst;
    dst = tmp;
  }
  return src;
}

// Ranking without merge rounds.  The merge sort above pays log2(n) rounds of a barrier and a bisection of dependent LDS reads each
// (n ~ 140 groups per window on config 3: 8 rounds, 36 bisection steps — 6.7 us of a window's 39, measured by leaving it out), for
// keys that are nearly uniform in their leading word: (reference, start).  Here every group drops into one of 256 buckets by that
// word — offset from the window's smallest, scaled by a shift so that the largest lands in the last bucket — with one returning LDS
// atomic, one wave scans the bucket counts, and a group's rank is its bucket's offset plus the number of smaller keys inside its
// bucket (a handful of compares: ~ 0.5 neighbours per bucket; a pile-up of groups on one base degenerates into counting, which is
// still correct).  Five barriers, ~ ten LDS round trips.  Windows whose groups span more than 2^31 in (reference, start) — several
// references in one window — keep the merge sort (uniform decision).
// bk: 2 * 256 + 3 * WG_NW words of LDS that nobody else uses during the ranking (the table's claim words are dead by then).
__device__ __forceinline__ int32_t wave_incl_min_i32(int32_t v) {
#define WG_RED_STEP(ctrl, rm)                                                           \
  {                                                                                     \
    const int32_t o = __builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rm, 0xf, false); \
    v = o < v ? o : v;                                                                  \
  }
  WG_RED_STEP(0x111, 0xf)
  WG_RED_STEP(0x112, 0xf)
  WG_RED_STEP(0x114, 0xf)
  WG_RED_STEP(0x118, 0xf)
  WG_RED_STEP(0x142, 0xa)
  WG_RED_STEP(0x143, 0xc)
#undef WG_RED_STEP
  return v;  // (lane 63: the wave's minimum)
}
__device__ __forceinline__ uint32_t wave_incl_sum_dpp(uint32_t v) {
#define WG_SUM_STEP(ctrl, rm) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rm, 0xf, false);
  WG_SUM_STEP(0x111, 0xf)
  WG_SUM_STEP(0x112, 0xf)
  WG_SUM_STEP(0x114, 0xf)
  WG_SUM_STEP(0x118, 0xf)
  WG_SUM_STEP(0x142, 0xa)
  WG_SUM_STEP(0x143, 0xc)
#undef WG_SUM_STEP
  return v;
}
constexpr uint32_t WG_RANK_NB = 256;
template <int E>
__device__ __forceinline__ uint16_t* wg_bucket_rank(uint16_t* src, uint16_t* dst, uint32_t n, const uint64_t* hi, const uint64_t* lo, uint32_t lim,
                                                    uint32_t* bk) {
  const uint32_t t = threadIdx.x;
  uint32_t* cnt = bk;
  uint32_t* off = bk + WG_RANK_NB;
  int32_t* red = reinterpret_cast<int32_t*>(bk + 2 * WG_RANK_NB);  // [3][WG_NW]: minimum, - maximum, every offset fits
  const uint64_t ref = n ? hi[src[0]] >> 2 : 0ull;
  uint32_t idx[E], bkt[E], arr[E];
  int32_t dl[E];
  int32_t mn = 0x7FFFFFFF, nmx = 0x7FFFFFFF;  // (the maximum as the minimum of the negated offsets)
  bool ok = true;
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const uint32_t e = t + (uint32_t)u * WG_NT;
    idx[u] = bkt[u] = arr[u] = 0;
    dl[u] = 0;
    if (e < n) {
      idx[u] = src[e];
      const int64_t d = (int64_t)((hi[idx[u]] >> 2) - ref);
      ok = ok && d > -(1ll << 30) && d < (1ll << 30);
      dl[u] = (int32_t)d;
      mn = dl[u] < mn ? dl[u] : mn;
      nmx = -dl[u] < nmx ? -dl[u] : nmx;
    }
  }
  if (t < WG_RANK_NB) cnt[t] = 0;
  mn = wave_incl_min_i32(mn);
  nmx = wave_incl_min_i32(nmx);
  const bool wok = __all(ok);
  if (lane_id() == 63) {
    red[t >> 6] = mn;
    red[WG_NW + (t >> 6)] = nmx;
    red[2 * WG_NW + (t >> 6)] = wok ? 1 : 0;
  }
  __syncthreads();
  int32_t gmn = 0x7FFFFFFF, gnmx = 0x7FFFFFFF, gok = 1;
#pragma unroll
  for (int i = 0; i < WG_NW; ++i) {
    const int32_t a = red[i], b = red[WG_NW + i];
    gmn = a < gmn ? a : gmn;
    gnmx = b < gnmx ? b : gnmx;
    gok &= red[2 * WG_NW + i];
  }
  if (!gok) {  // (uniform) groups of several references in one window
    __syncthreads();
    return wg_merge_sort<E>(src, dst, n, hi, lo, lim);
  }
  const uint32_t range = n ? (uint32_t)(-gnmx - gmn) : 0u;  // largest offset from the smallest key word (< 2^31)
  const uint32_t sh = range < WG_RANK_NB ? 0u : (32u - (uint32_t)__builtin_clz(range)) - 8u;  // range >> sh < 256
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const uint32_t e = t + (uint32_t)u * WG_NT;
    if (e < n) {
      bkt[u] = (uint32_t)(dl[u] - gmn) >> sh;
      arr[u] = atomicAdd(&cnt[bkt[u]], 1u);
    }
  }
  __syncthreads();
  if (t < 64) {  // offsets of the buckets: four counters per lane, one wave
    const uint32_t c0 = cnt[4 * t], c1 = cnt[4 * t + 1], c2 = cnt[4 * t + 2], c3 = cnt[4 * t + 3];
    const uint32_t sum = c0 + c1 + c2 + c3;
    const uint32_t ex = wave_incl_sum_dpp(sum) - sum;
    off[4 * t] = ex;
    off[4 * t + 1] = ex + c0;
    off[4 * t + 2] = ex + c0 + c1;
    off[4 * t + 3] = ex + c0 + c1 + c2;
  }
  __syncthreads();
  uint32_t ob[E];
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const uint32_t e = t + (uint32_t)u * WG_NT;
    ob[u] = 0;
    if (e < n) {
      ob[u] = off[bkt[u]];
      dst[ob[u] + arr[u]] = (uint16_t)idx[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const uint32_t e = t + (uint32_t)u * WG_NT;
    if (e < n) {
      const uint32_t c = cnt[bkt[u]];
      const uint64_t kh = hi[idx[u]], kl = lo[idx[u]];
      uint32_t r = 0;
      for (uint32_t j = 0; j < c; ++j) {
        const uint32_t m = dst[ob[u] + j];
        const uint64_t mh = hi[m], ml = lo[m];
        r += ((mh < kh) | ((mh == kh) & ((ml < kl) | ((ml == kl) & (m < idx[u]))))) ? 1u : 0u;
      }
      src[ob[u] + r] = (uint16_t)idx[u];
    }
  }
  __syncthreads();
  return src;
}

__device__ __forceinline__ unsigned long long wg_fingerprint(uint64_t hi, uint64_t lo, uint64_t seed) {
  // (hi, lo) -> hi * K + lo is one-to-one in lo for equal hi and, for an odd seeded K, collides for two different hi only when
  // their difference times K equals the difference of the lo words; the xor-shift / odd-multiply rounds behind it are bijections
  // (two 64-bit multiplies in all.  A multiply-free fingerprint — two steps of the strategy hash, dev_common.hpp — passed every
  // test and cost 1.2 ms: 6.59 vs 5.4 ms on config 3.  The keys of a window differ in a few low bits of `start`, and shifts and
  // adds carry those into the slot index as a near-linear sequence: the probes pile up.)
  unsigned long long f = (hi ^ seed) * ((seed << 1) | 0x9E3779B97F4A7C15ull) + lo;
  f ^= f >> 29;
  f *= 0xBF58476D1CE4E5B9ull;
  f ^= f >> 32;
  return f == ~0ull ? 0ull : f;  // ~0 marks an empty slot
}

// pieces of window w: lengths, prefix (LDS pre[k + 1]), record count, the window's record base.  Returns false for a window that
// is empty by construction (equal bounds).
__device__ __forceinline__ bool wg_prologue(const WgIn& In, uint32_t w, uint32_t* pre, uint32_t* sm_u, uint32_t* n_w, uint32_t* wbase) {
  const uint32_t t = threadIdx.x, k = In.k;
  if (w > 0 && w + 1 < In.nw && In.W[w] == In.W[w - 1]) return false;  // (every splitter owns two bounds)
  const uint32_t* row0 = In.off + (size_t)w * k;
  const uint32_t* row1 = row0 + k;
  uint32_t len[3], basep = 0;  // k <= 1024 < 3 * WG_NT; thread t owns files 3t .. 3t+2 (blocked: prefix order = file order)
  uint32_t sum = 0;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const uint32_t f = t * 3 + u;
    len[u] = 0;
    if (f < k) {
      const uint32_t a = row0[f], b = row1[f];
      len[u] = b > a ? b - a : 0u;
      basep += a - In.off[f];  // row 0 = run starts
    }
    sum += len[u];
  }
  uint32_t tot;
  uint32_t ex = wg_block_excl<uint32_t>(sum, sm_u, &tot);
  *n_w = tot;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const uint32_t f = t * 3 + u;
    if (f < k) pre[f] = ex;
    ex += len[u];
  }
  if (t == 0) pre[k] = tot;
  uint32_t btot;
  (void)wg_block_excl<uint32_t>(basep, sm_u, &btot);
  *wbase = btot;
  __syncthreads();
  return true;
}

// wbase[r] = records of all runs before row r of the offsets matrix (r = w: the record base of window w); one wave per row
__global__ __launch_bounds__(256) void wg_rowsum_k(const uint32_t* __restrict__ off, uint32_t k, uint32_t nrows, uint32_t* __restrict__ wbase) {
  const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const uint32_t* row = off + (size_t)r * k;
  uint32_t sum = 0;
  for (uint32_t f = lane_id(); f < k; f += 64) sum += row[f] - off[f];  // row 0 = run starts
  sum = wave_sum(sum);
  if (lane_id() == 0) wbase[r] = sum;
}

// the windows that hold records, in window order inside a wave (the order is only a scheduling matter)
__global__ __launch_bounds__(1024) void wg_list_k(uint32_t nw, const uint32_t* __restrict__ wbase, uint32_t* __restrict__ list,
                                                   unsigned long long* __restrict__ cnt) {
  // (one atomic per 1024 windows: returning atomics on one word take ~ 12 ns each, a wave's worth apiece was 0.1 ms on config 3)
  __shared__ uint32_t sm[16];
  __shared__ unsigned long long s_base;
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = w < nw && wbase[w + 1] != wbase[w];
  uint32_t tot;
  const uint32_t ex = block_excl_sum<uint32_t, 1024>(have ? 1u : 0u, sm, &tot);
  if (threadIdx.x == 0) s_base = tot ? atomicAdd(cnt, (unsigned long long)tot) : 0ull;
  __syncthreads();
  if (have) list[s_base + ex] = w;
}

// One workgroup per window, any number of records: they stream through once, all pieces laid end to end, into an LDS table
// of groups — claim word = a 64-bit fingerprint of the key, the key itself stored beside it and compared by every record
// that lands on the slot (a fingerprint collision raises TBK_DERR_COLLISION: the host reseeds) — with per-group count,
// representative (atomic min of effective end << 32 | record), claim number and a bitset of the samples seen.  Every
// record leaves (window base + claim number of its group) in cslot; the distinct groups are then merge-sorted by key and
// written out with the claim -> rank map, from which wg_finish_k verifies the strategy key of every record against its
// group's representative and derives the record -> group map.  A window with more distinct groups than the table holds
// goes to the sort kernel's worklist (at most WG_CAP records: every window except a pile-up on a single base); a pile-up
// that overflows sends the tile to the sort path.
// (`final_tier`: an overflowing window of more than WG_CAP records cannot go on to the sort kernel)
// RAW: the pieces are pieces of the input files themselves; every record's key, filter verdict and effective end (a segmented
// prefix maximum along the piece, carried across waves, rows and chunks) are computed here, records that do not pass take no
// part in the grouping.
// PART (RAW only): the records are group partials of other ranks — TieBrush-merged records with an explicit priority (multi-GPU
// owner side): a group's count is the sum of the carried integral YC, the two words that hold the bitset of the
// samples otherwise hold the sum of the carried YX and the maximum of the carried YD, and no (group, sample) incidence is
// emitted (tiebrush.cpp:389-395, :412-419: a TieBrush-merged record enters no sample list).
template <int SORT_E /* the ranking sorts at most SORT_E * WG_NT groups */, bool RAW, int NR /* records per thread and chunk */, int ST /* RAW: strategy */,
          int GC /* > 0: at most 64 input files, a table of exactly GC slots (compile-time LDS layout); 0: sizes from the arguments */,
          bool PART = false, int KA_R = -1 /* see WG_EPILOGUE_ARGS: where the caller's WgRaw ... */, int KA_T = -1 /* ... and WgTemp lie in the kernel's argument segment */>
__device__ __forceinline__ void wg_hash_window(const WgIn& In, const WgRaw& R, const WgTemp& T, uint32_t gcap_arg, uint32_t nwords_arg, uint64_t seed,
                                               uint32_t w, unsigned char* lds, uint32_t* sm_u, uint32_t* s_misc, uint2* s_agg /* [NR * WG_NW] */,
                                               uint32_t* __restrict__ ovf, uint32_t ovf_cap, bool final_tier, uint32_t* __restrict__ err) {
  const uint32_t t = threadIdx.x;
  const uint32_t k = In.k;
  const uint32_t gcap = GC > 0 ? (uint32_t)GC : gcap_arg;
  const uint32_t nwords = GC > 0 ? 2u : nwords_arg;
  const uint32_t kcap = GC > 0 ? 64u : k;  // entries of the pieces' tables
  // (the compile-time-layout form carries no instrumentation: TBK_WG_DEBUG runs take the general form)
  unsigned long long* const dbg = GC > 0 ? nullptr : T.dbg;
  const unsigned long long t_start = dbg ? __builtin_readcyclecounter() : 0ull;
  auto dbg_done = [&](int kind, uint32_t nrec) {
    if (dbg && threadIdx.x == 0) {
      atomicAdd(&dbg[kind * 4 + 0], __builtin_readcyclecounter() - t_start);
      atomicAdd(&dbg[kind * 4 + 1], 1ull);
      atomicAdd(&dbg[kind * 4 + 2], (unsigned long long)nrec);
      atomicMax(&dbg[kind * 4 + 3], __builtin_readcyclecounter() - t_start);
    }
  };
  // piece of item e (GC > 0: six steps without a branch over the table padded to 64 entries)
  auto piece = [&](const uint32_t* pre, uint32_t e) -> uint32_t {
    if constexpr (GC > 0) {
      uint32_t lo = 0;
#pragma unroll
      for (uint32_t st = 32; st > 0; st >>= 1) lo = pre[lo + st] <= e ? lo + st : lo;
      return lo;
    } else {
      return piece_of(pre, k, e);
    }
  };
  unsigned long long t_last = t_start;
  auto phase = [&](int i) {  // thread 0's clock between phase marks (TBK_WG_DEBUG)
    if (dbg && threadIdx.x == 0) {
      const unsigned long long now = __builtin_readcyclecounter();
      atomicAdd(&dbg[16 + i], now - t_last);
      t_last = now;
    }
  };
  const uint32_t wbase = T.wbase[w];
  const uint32_t n_w = T.wbase[w + 1] - wbase;
  if (n_w == 0) {  // (also every window that is empty by construction: equal bounds)
    if (t == 0) T.wg_cnt[w] = T.wp_cnt[w] = 0;
    dbg_done(0, 0);
    return;
  }
  // LDS: gcap slots of 44 + 4 nwords bytes, then the pieces' prefix and source bases (host: wg_hash_lds)
  unsigned long long* tc = reinterpret_cast<unsigned long long*>(lds);            // [gcap] claim word (fingerprint), ~0 = empty
  uint64_t* thi = reinterpret_cast<uint64_t*>(tc + gcap);                           // [gcap] key
  uint64_t* tlo = thi + gcap;                                                       // [gcap]
  unsigned long long* trep = reinterpret_cast<unsigned long long*>(tlo + gcap);     // [gcap]
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(trep + gcap);                        // [gcap]
  uint32_t* tci = tcnt + gcap;                                                      // [gcap] claim number of the slot
  uint32_t* tbits = tci + gcap;                                                     // [gcap * nwords] samples seen
  uint32_t* pre = tbits + (size_t)gcap * nwords;                                    // [k + 1] first item of piece f
  uint32_t* rb = pre + (kcap + 1);                                                  // [k] item e of piece f is compacted record rb[f] + e
  uint16_t* pa = reinterpret_cast<uint16_t*>(rb + kcap);                            // [gcap] slots by claim number, then the ranking's ping
  uint16_t* pb = pa + gcap;                                                         // [gcap] ... and pong
  {
    const uint32_t* row0 = In.off + (size_t)w * k;
    const uint32_t* row1 = row0 + k;
    // only the claim words are cleared here: count, representative and sample bits of a slot are set by the thread that claims it,
    // before the barrier that every user of the slot waits behind (37 KB of LDS stores per window became 6 KB)
    for (uint32_t i = t; i < gcap; i += WG_NT) tc[i] = ~0ull;
    if (t == 0) {
      s_misc[0] = 0;  // distinct groups
      s_misc[1] = 0;  // overflow
      s_misc[2] = 0;  // RAW: passing records
      s_misc[3] = 0;  // RAW: records that left a slot for the verification pass
      s_misc[4] = 0;  // RAW: carry of the effective-end scan into chunk 0 ...
      s_misc[5] = 0;  // ... and into chunk 1 (alternating)
      s_misc[6] = 0;  // ONEBAR: overflow raised in an even chunk ...
      s_misc[7] = 0;  // ... in an odd chunk
    }
    if constexpr (GC > 0) {  // <= 64 files: the piece lengths are one wave's scan (a block scan is ~ 100 instructions in each of the eight waves)
      if (t < 64) {
        uint32_t a0 = 0, len0 = 0;
        if (t < k) {
          a0 = row0[t];
          const uint32_t b = row1[t];
          len0 = b > a0 ? b - a0 : 0u;
        }
        const uint32_t ex0 = wave_incl_sum(len0) - len0;
        if (t < k) {
          pre[t] = ex0;
          rb[t] = a0 - ex0;  // (mod 2^32)
        }
      }
    } else {
      // k <= 1024 < 3 * WG_NT; thread t owns the fpt files from t * fpt (blocked: prefix order = file order)
      const uint32_t fpt = (k + WG_NT - 1) / WG_NT;
      uint32_t a[3], len[3], sum = 0;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const uint32_t f = t * fpt + u;
        a[u] = len[u] = 0;
        if ((uint32_t)u < fpt && f < k) {
          a[u] = row0[f];
          const uint32_t b = row1[f];
          len[u] = b > a[u] ? b - a[u] : 0u;
        }
        sum += len[u];
      }
      uint32_t tot;
      uint32_t ex = wg_block_excl<uint32_t>(sum, sm_u, &tot);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const uint32_t f = t * fpt + u;
        if ((uint32_t)u < fpt && f < k) {
          pre[f] = ex;
          rb[f] = a[u] - ex;  // (mod 2^32)
        }
        ex += len[u];
      }
    }
    if (t == 0) pre[k] = n_w;
    if (GC > 0 && t > k && t <= 64) pre[t] = n_w;  // (padding: no item reaches these)
  }
  __syncthreads();
  phase(0);
  uint32_t npass_t = 0, nslot_t = 0, par = 0;
  // RAW, one record per thread and chunk: the piece, the record and the CIGAR range of this thread's record of the NEXT chunk are
  // fetched a chunk ahead (four registers), so that a chunk's CIGAR words are asked for together with its fields: one round trip to
  // memory per chunk instead of two dependent ones
  constexpr bool AHEAD = RAW && NR == 1 && WG_RS == 1;
  // ONEBAR: one barrier per chunk instead of two.  The second one kept a fast wave out of the next chunk while a slow one still worked
  // on this one; what they could disturb is kept apart by the chunk's parity instead — the wave aggregates of the effective-end scan
  // (s_agg), the carry (s_misc[4 + par], as before) and the overflow flag (s_misc[6 + par]: a wave that has not yet looked at this
  // chunk's flag must not see the next chunk's) — and everything else a chunk touches after its barrier (counts, representatives and
  // sample bits of slots claimed BEFORE that barrier; the slots' claim numbers) is not written by the next chunk's probes, which
  // claim and initialise NEW slots only.  Two chunks ahead is impossible: the next chunk's barrier needs every wave.
  constexpr bool ONEBAR = RAW && NR == 1;
  uint2* const s_agg0 = s_agg;
  uint32_t a_fil = 0, a_src = 0, a_c0 = 0, a_c1 = 0;
  if constexpr (AHEAD) {
    if (t < n_w) {
      a_fil = piece(pre, t);
      a_src = rb[a_fil] + t;
      a_c0 = wg_ld(R.I.cig_off, a_src << 2);
      a_c1 = wg_ld(R.I.cig_off, (a_src << 2) + 4u);
    }
  }
  // PIPE (the AHEAD form): the loads of a chunk's records are issued a chunk ahead too — right behind the previous chunk's key
  // computation, whose raw fields they replace in the registers — so that they fly while that chunk probes the table, waits at its
  // barrier and updates its groups, instead of every wave of the block sitting out a round trip to memory at the top of each chunk
  // (a build that left the loads out altogether took 1.8 - 2.4 ms off the kernel's 6.7).
  constexpr bool PIPE = AHEAD;
  RawL n_l = {};
  uint32_t n_fil = 0, n_src = 0, n_f0 = 0;
  bool n_first = true, n_fromem = false;
  auto issue = [&](uint32_t cb) {  // loads of chunk [cb, cb + WG_NT) (a_* describe it), then where the chunk behind it lies
    if (cb + (t & ~63u) < n_w) {
      const uint32_t e1 = cb + t;
      const bool act1 = e1 < n_w;
      n_fil = a_fil;
      n_src = a_src;
      n_first = !act1 || e1 == pre[n_fil];
      n_fromem = act1 && (n_first || lane_id() == 0);
      n_l = wg_raw_l(R.I, n_src, n_fromem, a_c0, a_c1, (R.O.max_nh != INT32_MAX ? 1u : 0u) | (R.O.min_qual > 0 ? 2u : 0u));
      n_f0 = n_fromem ? R.I.file_off[n_fil] : 0u;
      const uint32_t e2 = cb + WG_NT + t;
      a_fil = a_src = a_c0 = a_c1 = 0;
      if (e2 < n_w) {
        a_fil = piece(pre, e2);
        a_src = rb[a_fil] + e2;
        a_c0 = wg_ld(R.I.cig_off, a_src << 2);
        a_c1 = wg_ld(R.I.cig_off, (a_src << 2) + 4u);
      }
    }
  };
  if constexpr (PIPE) issue(0u);
  // the window streams through in chunks of WG_NT * NR records: NR records per thread so that their loads and probes
  // overlap; two barriers per chunk
  for (uint32_t c0 = 0; c0 < n_w; c0 += WG_NT * NR) {
    uint32_t slot[NR], rec[NR], fil[NR], src[NR], eff[NR];
    uint64_t kh[NR], kl[NR];
    uint32_t won = 0, actm = 0;
    uint32_t xs[NR], fs[NR];  // RAW: scanned end inside the wave row, and whether a segment start shields it from the carry
    s_agg = s_agg0 + (ONEBAR ? par * (uint32_t)(NR * WG_NW) : 0u);
    uint32_t* const ovf_flag = &s_misc[ONEBAR ? 6u + par : 1u];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const uint32_t e = c0 + (uint32_t)u * WG_NT + t;
      slot[u] = 0xFFFFFFFFu;
      rec[u] = fil[u] = src[u] = eff[u] = 0;
      kh[u] = kl[u] = 0;
      xs[u] = fs[u] = 0;
      if constexpr (!RAW) {
        if (e < n_w) {
          actm |= 1u << u;
          fil[u] = piece(pre, e);
          src[u] = rb[fil[u]] + e;
          kh[u] = In.chi[src[u]];
          kl[u] = In.clo[src[u]];
          rec[u] = In.cval[src[u]];
          eff[u] = In.ceff[src[u]];
        }
      }
    }
    if constexpr (RAW) {
      uint32_t errb = 0;
#pragma unroll
      for (int u0 = 0; u0 < NR; u0 += WG_RS) {  // WG_RS records per thread are decoded together
        // (the code below is branch-free per lane: a wave whose row lies wholly beyond the window — the tail of its last chunk —
        // would execute all of it for nothing; it leaves the aggregate an all-absent row would have left and moves on)
        if (c0 + (uint32_t)u0 * WG_NT + (t & ~63u) >= n_w) {
#pragma unroll
          for (int v = 0; v < WG_RS; ++v)
            if (lane_id() == 63) s_agg[(u0 + v) * WG_NW + (t >> 6)] = make_uint2(0u, 1u);
          // (PIPE: such a wave has asked for nothing — issue() has the same guard —, but the compiler's wait-count bookkeeping sees a path
          // from the loads before the loop to the probes that passes through here and would make EVERY wave wait for its loads before
          // the probes; an explicit wait on this path, free at run time, tells it that nothing is in flight)
          if constexpr (PIPE) __builtin_amdgcn_s_waitcnt(TBK_WAIT_VMCNT0);  // vmcnt(0), nothing else
          continue;
        }
        RawA ra[WG_RS];
        CigView cv[WG_RS];
        bool first[WG_RS], fromem[WG_RS];
        uint32_t f0[WG_RS];
#pragma unroll
        for (int v = 0; v < WG_RS; ++v) {
          const int u = u0 + v;
          const uint32_t e = c0 + (uint32_t)u * WG_NT + t;
          const bool act = e < n_w;
          if constexpr (PIPE) {  // this chunk's fields were asked for a chunk ago (0, 0 and an empty CIGAR where there is no record)
            fil[u] = n_fil;
            src[u] = n_src;
            first[v] = n_first;
            fromem[v] = n_fromem;
            f0[v] = n_f0;
            ra[v].pos = n_l.pos;
            ra[v].tidv = n_l.tidv;
            ra[v].ppos = n_l.ppos;
            ra[v].ptid = n_l.ptid;
            ra[v].fl_mq_sc = (uint32_t)n_l.flag | ((uint32_t)n_l.mapq << 16) | (strand_code(n_l.strand) << 24);
            ra[v].nh = n_l.nh;
            ra[v].c0 = n_l.c0;
            ra[v].nc = n_l.nc;
            cv[v].w0 = n_l.w0;
            cv[v].w1 = n_l.w1;
            cv[v].w2 = n_l.w2;
            cv[v].p = R.I.cig + n_l.c0;
          } else {
            fil[u] = act ? piece(pre, e) : 0u;
            src[u] = act ? rb[fil[u]] + e : 0u;
            first[v] = !act || e == pre[fil[u]];
            fromem[v] = act && (first[v] || lane_id() == 0);  // the record before it in its file is not the lane to the left
            ra[v] = wg_raw_a(R.I, src[u], fromem[v]);
            f0[v] = fromem[v] ? R.I.file_off[fil[u]] : 0u;
          }
        }
        if constexpr (!PIPE) {
#pragma unroll
          for (int v = 0; v < WG_RS; ++v) cv[v] = wg_raw_b(R.I, ra[v]);
        }
#pragma unroll
        for (int v = 0; v < WG_RS; ++v) {
          const int u = u0 + v;
          const uint32_t e = c0 + (uint32_t)u * WG_NT + t;
          const bool act = e < n_w;
          const RawRec r = wg_raw_c<ST>(R, src[u], ra[v], cv[v]);
          uint64_t rk = ~0ull;
          uint32_t x = 0;
          if (act) {
            rec[u] = src[u];
            kh[u] = r.hi;
            kl[u] = r.lo;
            rk = r.rk;
            x = r.x;
            errb |= r.err;
            if (r.pass) {
              actm |= 1u << u;
              ++npass_t;
            }
          }
          uint64_t prk = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(rk >> 32), 0x138, 0xf, 0xf, false) << 32) |
                         (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)rk, 0x138, 0xf, 0xf, false);
          if (fromem[v]) prk = src[u] > f0[v] ? r.prk : 0ull;
          if (act && prk > rk) errb |= TBK_DERR_RAWORDER;  // not coordinate-sorted as this path needs it
          const uint32_t head = (first[v] || prk != rk) ? 1u : 0u;
          xs[u] = wave_seg_max(x, head, &fs[u]);
          if (lane_id() == 63) s_agg[u * WG_NW + (t >> 6)] = make_uint2(xs[u], fs[u]);
        }
      }
      if (errb) atomicOr(err, errb);
      if constexpr (PIPE) issue(c0 + WG_NT);  // the next chunk's loads, into the registers this chunk's raw fields have just left
    }
    if (dbg) {
      if (t == 0 && (kh[0] ^ kl[0] ^ rec[0] ^ eff[0] ^ kh[NR - 1] ^ rec[NR - 1]) == 0x123456789ull) dbg[31] = 1;  // (the loads have landed)
      phase(2);
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      if ((actm >> u) & 1u) {
        const unsigned long long F = wg_fingerprint(kh[u], kl[u], seed);
        uint32_t hs = gcap < (1u << 16) ? __umul24((uint32_t)(F >> 48), gcap) >> 16 : (uint32_t)(((F >> 32) * gcap) >> 32);  // (a full-rate multiply)
        for (uint32_t probe = 0; probe < gcap; ++probe) {
          const unsigned long long cur = tc[hs];
          if (cur == F) {
            slot[u] = hs;
            break;
          }
          if (cur == ~0ull) {
            const unsigned long long old = atomicCAS(&tc[hs], ~0ull, F);
            if (old == ~0ull) {
              slot[u] = hs;
              won |= 1u << u;
              break;
            }
            if (old == F) {
              slot[u] = hs;
              break;
            }
          }
          hs = hs + 1 == gcap ? 0u : hs + 1;
        }
        if ((won >> u) & 1u) {
          thi[slot[u]] = kh[u];
          tlo[slot[u]] = kl[u];
          trep[slot[u]] = ~0ull;
          tcnt[slot[u]] = 0;
          for (uint32_t x = 0; x < nwords; ++x) tbits[slot[u] * nwords + x] = 0;
          const uint32_t ci = atomicAdd(&s_misc[0], 1u);
          tci[slot[u]] = ci;
          pa[ci] = (uint16_t)slot[u];  // (a claim takes a slot: ci < gcap)
          if (ci + 1u > (gcap >> 2) * 3u) *ovf_flag = 1;
        }
        if (slot[u] == 0xFFFFFFFFu) *ovf_flag = 1;
      }
    }
    phase(3);
    __syncthreads();  // key and claim number of every slot claimed in this chunk are visible
    phase(4);
    if (*ovf_flag) break;
    if constexpr (RAW) {  // effective ends: the carry that enters every (row, wave) of this chunk, folded in element order
      // (lane q of every wave holds aggregate q; a 16-lane segmented scan gives the carry behind each of them)
      static_assert(NR * WG_NW <= 16, "the carries are folded inside one DPP row");
      if (c0 + (t & ~63u) < n_w) {  // (a wave without a record in this chunk has no use for the carries; wave 0 always has one)
        const uint32_t cin = s_misc[4 + par];
        const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
        bool folded = false;
        if constexpr (NR == 1) {
          // The carry into a wave is the aggregate of the wave before it wherever that wave holds a segment start — the first record of a
          // piece, or a record that does not repeat its predecessor's position: nearly every wave — and the chunk's carry-out is the last
          // wave's aggregate in the same way: two broadcast reads instead of a scan over the aggregates in every wave.
          const uint2 pa = s_agg[wv > 0 ? wv - 1 : 0];
          const uint2 la = s_agg[WG_NW - 1];
          const uint32_t pf = (uint32_t)__builtin_amdgcn_readfirstlane((int)pa.y), lf = (uint32_t)__builtin_amdgcn_readfirstlane((int)la.y);
          if (wv == 0 ? lf != 0u : pf != 0u) {  // (uniform)
            const uint32_t snap0 = wv == 0 ? cin : pa.x;
            if (t == 0) s_misc[4 + (par ^ 1u)] = la.x;
            eff[0] = fs[0] ? xs[0] : (xs[0] > snap0 ? xs[0] : snap0);
            folded = true;
          }
        }
        if (!folded) {
          const uint32_t lq = lane_id();
          const uint2 a = lq < (uint32_t)(NR * WG_NW) ? s_agg[lq] : make_uint2(0u, 0u);
          uint32_t ax = a.x, af = a.y;
#define WG_FOLD_STEP(ctrl)                                                                        \
  {                                                                                               \
    const uint32_t xo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ax, ctrl, 0xf, 0xf, false); \
    const uint32_t fq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)af, ctrl, 0xf, 0xf, false); \
    ax = af ? ax : (xo > ax ? xo : ax);                                                           \
    af |= fq;                                                                                     \
  }
          WG_FOLD_STEP(0x111)  // row_shr:1, 2, 4, 8
          WG_FOLD_STEP(0x112)
          WG_FOLD_STEP(0x114)
          WG_FOLD_STEP(0x118)
#undef WG_FOLD_STEP
          const uint32_t cq = af ? ax : (ax > cin ? ax : cin);  // the carry behind aggregate lq
          uint32_t snap[NR];
#pragma unroll
          for (int u = 0; u < NR; ++u) {
            const int q = u * WG_NW + wv;  // uniform
            snap[u] = q == 0 ? cin : (uint32_t)__builtin_amdgcn_readlane((int)cq, q - 1);
          }
          if (t == 0) s_misc[4 + (par ^ 1u)] = (uint32_t)__builtin_amdgcn_readlane((int)cq, NR * WG_NW - 1);
#pragma unroll
          for (int u = 0; u < NR; ++u) eff[u] = fs[u] ? xs[u] : (xs[u] > snap[u] ? xs[u] : snap[u]);
        }
      }
      par ^= 1u;
      if (R.I.prio_hi) {  // cross-rank tiles: the merge order was fixed where the files live — the explicit priority replaces the scan
#pragma unroll
        for (int u = 0; u < NR; ++u)
          if ((actm >> u) & 1u) eff[u] = (uint32_t)R.I.prio_hi[src[u]];
        // (PIPE: a load under a lane mask whose use sits under another one is "possibly in flight" at every later write of its register
        // as far as the compiler's wait counts go — it would make every tile wait for the next chunk's loads before the next probes)
        if constexpr (PIPE) __builtin_amdgcn_s_waitcnt(TBK_WAIT_VMCNT0);
      }
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const bool act = (actm >> u) & 1u;
      if (act) {
        const uint32_t s = slot[u];
        if (thi[s] != kh[u] || tlo[s] != kl[u]) atomicOr(err, TBK_DERR_COLLISION);  // two keys, one fingerprint
        if (!RAW || R.all_slots || !((kl[u] >> 31) & 1ull)) {
          if (RAW && R.sparse) {
            const uint32_t e = atomicAdd(&s_misc[3], 1u);  // (< n_w: one entry per record at most)
            T.vsrc[wbase + e] = src[u];
            T.cslot[wbase + e] = wbase + tci[s];
          } else {
            T.cslot[src[u]] = wbase + tci[s];
            ++nslot_t;
          }
        }
        const unsigned long long rr = ((unsigned long long)eff[u] << 32) | rec[u];
        if (rr < trep[s]) atomicMin(&trep[s], rr);
        if constexpr (PART) {
          double y = R.I.yc_in[src[u]];
          y = y == 0.0 ? 1.0 : y;  // (tiebrush.cpp:389-395: an absent / zero YC counts as one)
          const long long yx = R.I.yx_in[src[u]], yd = R.I.yd_in[src[u]];
          if (!(y == rint(y)) || y < 0.0 || y >= 2147483648.0 || yx < 0 || yx >= 2147483648ll || yd >= 2147483648ll)
            atomicOr(err, TBK_DERR_FRACTIONAL);  // not this form's kind of partial: the sort path takes the tile
          atomicAdd(&tcnt[s], (uint32_t)y);
          atomicAdd(&tbits[s * nwords], (uint32_t)yx);
          if (yd > 0) atomicMax(&tbits[s * nwords + 1], (uint32_t)yd);
        } else {
          const uint32_t bi = s * nwords + (fil[u] >> 5), bm = 1u << (fil[u] & 31);
          if (!(tbits[bi] & bm)) atomicOr(&tbits[bi], bm);
        }
      }
      // counts: the leader's group by one ballot (a pile-up is mostly one group), the other lanes add for themselves
      const uint64_t am = PART ? 0ull : __ballot(act);
      if (am) {
        const int leader = __builtin_ctzll(am);
        const uint32_t s0 = __shfl(slot[u], leader, 64);
        const uint64_t same = __ballot(act && slot[u] == s0);
        if ((int)lane_id() == leader) atomicAdd(&tcnt[s0], (uint32_t)__builtin_popcountll(same));
        if (act && slot[u] != s0) atomicAdd(&tcnt[slot[u]], 1u);
      }
    }
    phase(5);
    if constexpr (!ONEBAR) __syncthreads();
    phase(6);
  }
  __syncthreads();
  const bool overflow = ONEBAR ? (s_misc[6] | s_misc[7]) != 0 : s_misc[1] != 0;
  const uint32_t d = s_misc[0];
  // What follows writes through fifteen pointers of T that the chunk loop has no use for.  As kernel arguments they are loaded at the
  // kernel's entry and stay live across the loop — more scalar registers than the hardware has, so the compiler kept ~ 60 of the loop's
  // own scalars in lanes of two vector registers and fetched them back with v_readlane at every use (78 of the ~ 650 vector
  // instructions of a chunk, in a kernel that keeps the vector ALUs busy 4/5 of the time).  Here the epilogue's pointers are read
  // from the argument segment where they are needed instead: behind an opaque move of the segment's address the loads cannot be hoisted.
  const WgTemp* TEp = &T;
  const WgRaw* REp = &R;
  if constexpr (KA_T >= 0) {
    const __attribute__((address_space(4))) unsigned char* kp =
        (const __attribute__((address_space(4))) unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    TEp = (const WgTemp*)(kp + KA_T);
    REp = (const WgRaw*)(kp + KA_R);
  }
  const WgTemp& TE = *TEp;
  const WgRaw& RE = *REp;
  if (!overflow) {
    // ---- rank the groups by key (pa holds their slots in claim order), emit groups and incidences in rank order ----
    // (the claim words are dead: their first 2.1 KB serve the ranking's buckets; gcap >= 272 slots of 8 bytes)
    uint16_t* byrank = gcap * 8u >= (2u * WG_RANK_NB + 3u * WG_NW) * 4u && !In.rank_merge
                           ? wg_bucket_rank<SORT_E>(pa, pb, d, thi, tlo, gcap, reinterpret_cast<uint32_t*>(tc))
                           : wg_merge_sort<SORT_E>(pa, pb, d, thi, tlo, gcap);  // d <= 3/4 gcap <= SORT_E * WG_NT
    // (ranking by counting smaller keys — d broadcast reads per group — was measured slower than the merge rounds: 9.1 vs 8.1 ms)
    phase(8);
    for (uint32_t g = t; g < d; g += WG_NT) TE.c2r[wbase + tci[byrank[g]]] = wbase + g;
    __syncthreads();
    const bool fm = !PART && TE.fmask != nullptr;  // (<= 64 files) the groups' files travel as bit masks: no incidence list, no offsets
    uint32_t* nsr = tci;  // (dead from here) [d] samples per group in rank order -> offsets
    if (!fm) {
      for (uint32_t g = t; g < d; g += WG_NT) {
        const uint32_t s = byrank[g];
        uint32_t c = 0;
        if constexpr (!PART)
          for (uint32_t x = 0; x < nwords; ++x) c += (uint32_t)__builtin_popcount(tbits[s * nwords + x]);
        nsr[g] = c;
      }
      __syncthreads();
      uint32_t carry = 0;
      for (uint32_t i0 = 0; i0 < d; i0 += WG_NT) {
        const uint32_t i = i0 + t;
        const uint32_t v = i < d ? nsr[i] : 0u;
        uint32_t tot;
        const uint32_t ex = wg_block_excl<uint32_t>(v, sm_u, &tot);
        if (i < d) nsr[i] = carry + ex;
        carry += tot;
      }
      if (t == 0) TE.wp_cnt[w] = carry;
    }
    if (t == 0) TE.wg_cnt[w] = d;  // (fm: wp_cnt[w] stays 0)
    if constexpr (RAW) {
      const uint32_t ws = wave_sum(npass_t);
      if (lane_id() == 0 && ws) atomicAdd(&s_misc[2], ws);
      const uint32_t wl = wave_sum(nslot_t);
      if (lane_id() == 0 && wl) atomicAdd(&s_misc[3], wl);
    }
    __syncthreads();
    if (RAW && t == 0 && s_misc[2]) wg_count(RE.n_pass, (unsigned long long)s_misc[2]);
    if (RAW && t == 0 && s_misc[3]) wg_count(RE.n_slots, (unsigned long long)s_misc[3]);
    if (RAW && t == 0 && RE.sparse) TE.vcnt[w] = s_misc[3];
    for (uint32_t g = t; g < d; g += WG_NT) {
      const uint32_t s = byrank[g];
      TE.hi[wbase + g] = thi[s];
      TE.lo[wbase + g] = tlo[s];
      TE.cnt[wbase + g] = tcnt[s];
      TE.rep[wbase + g] = trep[s];
      if constexpr (PART) {
        TE.yx[wbase + g] = tbits[s * nwords];
        TE.yd[wbase + g] = tbits[s * nwords + 1];
      }
      if (fm) {
        const uint32_t b0 = tbits[s * nwords], b1 = nwords > 1u ? tbits[s * nwords + 1] : 0u;
        TE.fmask[wbase + g] = ((uint64_t)b1 << 32) | b0;
        TE.ns[wbase + g] = (uint32_t)(__builtin_popcount(b0) + __builtin_popcount(b1));
        continue;
      }
      uint32_t pl = nsr[g];
      TE.poff[wbase + g] = pl;
      uint32_t c = 0;
      for (uint32_t x = 0; x < (PART ? 0u : nwords); ++x) {
        uint32_t bits = tbits[s * nwords + x];
        while (bits) {
          const uint32_t bpos = (uint32_t)__builtin_ctz(bits);
          bits &= bits - 1;
          TE.pinc[wbase + pl] = (x * 32 + bpos) | (g << 16);
          ++pl;
          ++c;
        }
      }
      TE.ns[wbase + g] = c;
    }
    phase(9);
    dbg_done(2, n_w);
    return;
  }
  // more distinct groups than the table holds: a window of at most WG_CAP records goes to the sort kernel's worklist, a
  // pile-up sends the tile to the sort path
  if (t == 0) {
    TE.wg_cnt[w] = TE.wp_cnt[w] = 0;
    if (final_tier && n_w > (uint32_t)WG_CAP) {
      atomicOr(err, TBK_DERR_BIGBUCKET);
    } else {
      const uint32_t i = atomicAdd(&ovf[0], 1u);
      if (i < ovf_cap)
        ovf[1 + i] = w;
      else
        atomicOr(err, TBK_DERR_BIGBUCKET);
    }
  }
}

// first tier: one block per window that holds records, a table of WG_LDS_HASH bytes (four blocks per CU)
constexpr int WG_GC64 = (WG_LDS_HASH - (8 * 64 + 8)) / (44 + 4 * 2);  // table slots of the <= 64 files form of the first tier
// WG_EPILOGUE_ARGS: the leading arguments of wg_hash_k as they lie in its argument segment (every argument at its natural alignment, in order)
struct WgHashArgs {
  WgIn In;
  WgRaw R;
  WgTemp T;
};
template <bool RAW, int ST, int GC, bool PART = false>
__global__ __launch_bounds__(WG_NT, 8) void wg_hash_k(WgIn In, WgRaw R, WgTemp T, uint32_t gcap, uint32_t nwords, uint64_t seed,
                                                      const uint32_t* __restrict__ wlist, uint32_t* __restrict__ ovf /* [0] count, [1..] windows */,
                                                      uint32_t ovf_cap, uint32_t* __restrict__ err) {
  extern __shared__ __align__(16) unsigned char lds[];
  __shared__ uint32_t sm_u[WG_NW];
  __shared__ uint32_t s_misc[8];
  __shared__ uint2 s_agg[WG_R * WG_NW];
  // (Blocks b and b + 8 are observed to share an XCD.  Handing each XCD a contiguous eighth of the window list, so that the cache lines in
  // which consecutive windows' pieces meet are found in one L2, was measured slower — 9.0 vs 7.5 ms on config 3: eight times as many
  // streams into every column — so the windows in flight stay one compact range of the tile.)
  wg_hash_window<2, RAW, RAW ? WG_RR : WG_R, ST, GC, PART, (int)offsetof(WgHashArgs, R), (int)offsetof(WgHashArgs, T)>(
      In, R, T, gcap, nwords, seed, wlist[blockIdx.x], lds, sm_u, s_misc, s_agg, ovf, ovf_cap, false, err);
}
// second tier: the windows with more distinct groups than the first table holds (shallow data) against a table of WG_LDS_HASH2 bytes
// (two blocks per CU); a fixed grid walks the first tier's worklist.  What overflows again goes to the sort kernel.
constexpr int WG_GC64_2 = (WG_LDS_HASH2 - (8 * 64 + 8)) / (44 + 4 * 2);  // ... and of the second tier
template <bool RAW, int ST, int GC, bool PART = false>
__global__ __launch_bounds__(WG_NT, 6) void wg_hash2_k(WgIn In, WgRaw R, WgTemp T, uint32_t gcap, uint32_t nwords, uint64_t seed,
                                                       const uint32_t* __restrict__ ovf_in, uint32_t* __restrict__ ovf, uint32_t ovf_cap,
                                                       uint32_t* __restrict__ err) {
  extern __shared__ __align__(16) unsigned char lds[];
  __shared__ uint32_t sm_u[WG_NW];
  __shared__ uint32_t s_misc[8];
  __shared__ uint2 s_agg[WG_R * WG_NW];
  const uint32_t cnt = ovf_in[0] < ovf_cap ? ovf_in[0] : ovf_cap;
  for (uint32_t wi = blockIdx.x; wi < cnt; wi += gridDim.x) {
    __syncthreads();  // (LDS of the previous window is free)
    wg_hash_window<3, RAW, RAW ? WG_RR : WG_R, ST, GC, PART>(In, R, T, gcap, nwords, seed, ovf_in[1 + wi], lds, sm_u, s_misc, s_agg, ovf, ovf_cap, true,
                                                             err);
  }
}

// The windows the hash kernels could not hold (more distinct groups than table slots — shallow data): at most WG_CAP records
// each, sorted in LDS.  A fixed grid walks the worklist.
// RAW: keys, filter verdicts and effective ends are computed while the window is loaded (see wg_hash_window); records that do not
// pass sort behind every key and are left out.
template <bool RAW, int ST, bool PART = false>
__global__ __launch_bounds__(WG_NT, 3) void wg_sort_k(WgIn In, WgRaw R, WgTemp T, ColIn I, int strategy, const uint32_t* __restrict__ ovf,
                                                      uint32_t ovf_cap, uint32_t* __restrict__ err) {
  __shared__ __align__(16) unsigned char lds[WG_LDS_MAIN];
  __shared__ uint32_t pre[1024 + 1];
  __shared__ uint32_t sm_u[WG_NW];
  __shared__ uint2 s_agg[WG_NW];
  __shared__ uint32_t s_carry, s_np, s_vn;
  const uint32_t t = threadIdx.x;
  const uint32_t k = In.k;
  const uint32_t cnt = ovf[0] < ovf_cap ? ovf[0] : ovf_cap;
  for (uint32_t wi = blockIdx.x; wi < cnt; wi += gridDim.x) {
    const uint32_t w = ovf[1 + wi];
    const unsigned long long t_start = T.dbg ? __builtin_readcyclecounter() : 0ull;
    auto dbg_done = [&](int kind, uint32_t nrec) {
      if (T.dbg && threadIdx.x == 0) {
        atomicAdd(&T.dbg[kind * 4 + 0], __builtin_readcyclecounter() - t_start);
        atomicAdd(&T.dbg[kind * 4 + 1], 1ull);
        atomicAdd(&T.dbg[kind * 4 + 2], (unsigned long long)nrec);
        atomicMax(&T.dbg[kind * 4 + 3], __builtin_readcyclecounter() - t_start);
      }
    };
    if (t == 0) s_vn = 0;
    __syncthreads();  // (LDS of the previous window is free)
    uint32_t n_w, wbase;
    (void)wg_prologue(In, w, pre, sm_u, &n_w, &wbase);
    const uint32_t* row0 = In.off + (size_t)w * k;
    // =========================== LDS sort path (shallow window: more groups than table slots) ===========================
    uint64_t* hi = reinterpret_cast<uint64_t*>(lds);
    uint64_t* lo = hi + WG_CAP;
    uint32_t* val = reinterpret_cast<uint32_t*>(lo + WG_CAP);  // the record (RAW: its effective end; the record is its source index)
    uint16_t* qa = reinterpret_cast<uint16_t*>(val + WG_CAP);
    uint16_t* qb = qa + WG_CAP;
    if constexpr (!RAW) {
      for (uint32_t e = t; e < n_w; e += WG_NT) {
        const uint32_t f = piece_of(pre, k, e);
        const uint32_t src = row0[f] + (e - pre[f]);
        hi[e] = In.chi[src];
        lo[e] = In.clo[src];
        val[e] = In.cval[src];
        qa[e] = (uint16_t)e;
      }
      __syncthreads();
    } else {
      if (t == 0) {
        s_carry = 0;
        s_np = 0;
      }
      __syncthreads();
      uint32_t npass_t = 0;
      for (uint32_t e0 = 0; e0 < n_w; e0 += WG_NT) {  // rows of WG_NT consecutive records; the scan carry runs along them
        const uint32_t e = e0 + t;
        const bool act = e < n_w;
        uint64_t rk = ~0ull, h = ~0ull, l = ~0ull;
        uint32_t x = 0, f = 0, srcv = 0;
        bool first = true;
        bool fromem = false;
        RawRec r{};
        if (act) {
          f = piece_of(pre, k, e);
          srcv = row0[f] + (e - pre[f]);
          first = e == pre[f];
          fromem = first || lane_id() == 0;
          const RawA ra = wg_raw_a(R.I, srcv, fromem);
          const CigView cv = wg_raw_b(R.I, ra);
          r = wg_raw_c<ST>(R, srcv, ra, cv);
          if (r.err) atomicOr(err, r.err);
          if (r.pass) {
            h = r.hi;
            l = r.lo;
            ++npass_t;
          }
          rk = r.rk;
          x = r.x;
        }
        uint64_t prk = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(rk >> 32), 0x138, 0xf, 0xf, false) << 32) |
                       (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)rk, 0x138, 0xf, 0xf, false);
        if (fromem) prk = srcv > R.I.file_off[f] ? r.prk : 0ull;
        if (act && prk > rk) atomicOr(err, TBK_DERR_RAWORDER);
        const uint32_t head = (!act || first || prk != rk) ? 1u : 0u;
        uint32_t fo;
        x = wave_seg_max(x, head, &fo);
        if (lane_id() == 63) s_agg[t >> 6] = make_uint2(x, fo);
        __syncthreads();
        uint32_t c = s_carry, snap = 0;
#pragma unroll
        for (int q = 0; q < WG_NW; ++q) {
          if ((uint32_t)q == (t >> 6)) snap = c;
          const uint2 a = s_agg[q];
          c = a.y ? a.x : (a.x > c ? a.x : c);
        }
        if (act) {
          hi[e] = h;  // (a record that does not pass: ~0, ~0 — behind every key)
          lo[e] = l;
          val[e] = R.I.prio_hi ? (uint32_t)R.I.prio_hi[srcv] : (fo ? x : (x > snap ? x : snap));
          qa[e] = (uint16_t)e;
        }
        __syncthreads();
        if (t == 0) s_carry = c;
      }
      {
        const uint32_t ws = wave_sum(npass_t);
        if (lane_id() == 0 && ws) atomicAdd(&s_np, ws);
      }
      __syncthreads();
      if (t == 0 && s_np) wg_count(R.n_pass, (unsigned long long)s_np);
    }
    uint16_t* src = wg_merge_sort<WG_E>(qa, qb, n_w, hi, lo, WG_CAP);
    if constexpr (RAW) n_w = s_np;  // the passing records are the first s_np sorted positions
    const unsigned long long t_load = t_start, t_sort = T.dbg ? __builtin_readcyclecounter() : 0ull;
    // ---- heads, group ids, incidences: thread t owns the WG_E consecutive sorted positions from t * WG_E ----
    const uint32_t q0 = t * WG_E;
    uint32_t ix[WG_E], fl[WG_E], recs[WG_E], srci[WG_E];
    uint32_t hf = 0;  // bit u: group head, bit 8 + u: first record of its sample inside the group
    {
      uint32_t pxv = 0, pf = 0;
      uint64_t ph = 0, pl = 0;
      uint32_t prec = 0;
      if (q0 > 0 && q0 <= n_w) {
        pxv = src[q0 - 1];
        ph = hi[pxv];
        pl = lo[pxv];
        pf = piece_of(pre, k, pxv);
        prec = RAW ? row0[pf] + (pxv - pre[pf]) : val[pxv];
      }
      uint32_t vb[WG_E];  // predecessor of every position that continues a group: the pair is verified below
      uint32_t need = 0;
#pragma unroll
      for (int u = 0; u < WG_E; ++u) {
        const uint32_t q = q0 + u;
        ix[u] = 0;
        fl[u] = 0;
        recs[u] = 0;
        srci[u] = 0;
        vb[u] = 0;
        if (q < n_w) {
          ix[u] = src[q];
          const uint64_t h = hi[ix[u]], l = lo[ix[u]];
          fl[u] = piece_of(pre, k, ix[u]);
          srci[u] = row0[fl[u]] + (ix[u] - pre[fl[u]]);
          recs[u] = RAW ? srci[u] : val[ix[u]];
          const bool head = q == 0 || h != ph || l != pl;
          const bool fh = head || fl[u] != pf;
          hf |= (head ? 1u : 0u) << u;
          hf |= (fh ? 1u : 0u) << (8 + u);
          vb[u] = prec;
          // (an exact key — bit 31 of the hash word, col_keys_k — needs no comparison: equal keys are equal alignments)
          if (!head && !((l >> 31) & 1ull)) need |= 1u << u;
          ph = h;
          pl = l;
          pf = fl[u];
          prec = recs[u];
        }
      }
#pragma unroll
      for (int u = 0; u < WG_E; ++u)
        if (((need >> u) & 1u) && !strategy_equal(I, strategy, recs[u], vb[u])) atomicOr(err, TBK_DERR_COLLISION);
    }
    const uint32_t nh = (uint32_t)__builtin_popcount(hf & 0xFFu), nf = PART ? 0u : (uint32_t)__builtin_popcount(hf >> 8);
    uint32_t tot;
    const uint32_t ex = wg_block_excl<uint32_t>(nh | (nf << 16), sm_u, &tot);
    const uint32_t ng_w = tot & 0xFFFFu, np_w = tot >> 16;
    uint32_t gl[WG_E];
    {
      uint32_t g = (ex & 0xFFFFu), pl = ex >> 16;  // groups / incidences before this thread's first position
#pragma unroll
      for (int u = 0; u < WG_E; ++u) {
        const uint32_t q = q0 + u;
        gl[u] = 0;
        if (q < n_w) {
          const bool head = (hf >> u) & 1u, fh = (hf >> (8 + u)) & 1u;
          g += head ? 1u : 0u;
          gl[u] = g - 1u;
          if (head) {
            T.hi[wbase + gl[u]] = hi[ix[u]];
            T.lo[wbase + gl[u]] = lo[ix[u]];
            T.poff[wbase + gl[u]] = pl;
            T.c2r[wbase + gl[u]] = wbase + gl[u];
          }
          if (!PART && fh) {
            T.pinc[wbase + pl] = fl[u] | (gl[u] << 16);
            ++pl;
          }
          if (!RAW || R.all_slots || !((lo[ix[u]] >> 31) & 1ull)) {
            if (RAW && R.sparse) {
              const uint32_t e = atomicAdd(&s_vn, 1u);
              T.vsrc[wbase + e] = srci[u];
              T.cslot[wbase + e] = wbase + gl[u];
            } else {
              T.cslot[srci[u]] = wbase + gl[u];
            }
            if (RAW) wg_count(R.n_slots, 1ull);  // (the rare tier: no aggregation)
          }
        }
      }
    }
    const unsigned long long t_heads = T.dbg ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();  // hi / lo are dead from here: their space holds the per-group accumulators
    if (RAW && t == 0 && R.sparse) T.vcnt[w] = s_vn;
    uint32_t* gcnt = reinterpret_cast<uint32_t*>(lds);
    uint32_t* gns = gcnt + WG_CAP;
    unsigned long long* grep = reinterpret_cast<unsigned long long*>(gns + WG_CAP);
    uint32_t* gyd = reinterpret_cast<uint32_t*>(grep + WG_CAP);  // (PART; the space of `val`: dead once the gathers below are done)
    for (uint32_t g = t; g < ng_w; g += WG_NT) {
      gcnt[g] = 0;
      gns[g] = 0;
      grep[g] = ~0ull;
    }
    unsigned long long rr[WG_E];
    uint32_t ycv[WG_E], yxv[WG_E], ydv[WG_E];  // PART: the carried YC / YX / YD of the partial
#pragma unroll
    for (int u = 0; u < WG_E; ++u) {  // eight independent gathers in flight
      rr[u] = q0 + u < n_w ? (((unsigned long long)(RAW ? val[ix[u]] : In.ceff[srci[u]]) << 32) | recs[u]) : ~0ull;
      ycv[u] = 1u;
      yxv[u] = ydv[u] = 0u;
      if constexpr (PART) {
        if (q0 + u < n_w) {
          double y = R.I.yc_in[srci[u]];
          y = y == 0.0 ? 1.0 : y;
          const long long yx = R.I.yx_in[srci[u]], yd = R.I.yd_in[srci[u]];
          if (!(y == rint(y)) || y < 0.0 || y >= 2147483648.0 || yx < 0 || yx >= 2147483648ll || yd >= 2147483648ll)
            atomicOr(err, TBK_DERR_FRACTIONAL);
          ycv[u] = (uint32_t)y;
          yxv[u] = (uint32_t)yx;
          ydv[u] = yd > 0 ? (uint32_t)yd : 0u;
        }
      }
    }
    __syncthreads();
    if constexpr (PART) {  // (every read of `val` is behind the barrier above)
      for (uint32_t g = t; g < ng_w; g += WG_NT) gyd[g] = 0;
      __syncthreads();
    }
    {  // runs of one group inside the thread's positions are folded in registers; one set of LDS atomics per run
      uint32_t cg = 0xFFFFFFFFu, c = 0, nsv = 0, ydm = 0;
      unsigned long long r = ~0ull;
#pragma unroll
      for (int u = 0; u < WG_E; ++u) {
        if (q0 + u < n_w) {
          if (gl[u] != cg) {
            if (cg != 0xFFFFFFFFu) {
              atomicAdd(&gcnt[cg], c);
              if (nsv) atomicAdd(&gns[cg], nsv);
              if (PART && ydm) atomicMax(&gyd[cg], ydm);
              atomicMin(&grep[cg], r);
            }
            cg = gl[u];
            c = 0;
            nsv = 0;
            ydm = 0;
            r = ~0ull;
          }
          c += ycv[u];
          nsv += PART ? yxv[u] : ((hf >> (8 + u)) & 1u);
          ydm = ydv[u] > ydm ? ydv[u] : ydm;
          r = rr[u] < r ? rr[u] : r;
        }
      }
      if (cg != 0xFFFFFFFFu) {
        atomicAdd(&gcnt[cg], c);
        if (nsv) atomicAdd(&gns[cg], nsv);
        if (PART && ydm) atomicMax(&gyd[cg], ydm);
        atomicMin(&grep[cg], r);
      }
    }
    __syncthreads();
    for (uint32_t g = t; g < ng_w; g += WG_NT) {
      T.cnt[wbase + g] = gcnt[g];
      T.ns[wbase + g] = PART ? 0u : gns[g];
      T.rep[wbase + g] = grep[g];
      if constexpr (PART) {
        T.yx[wbase + g] = gns[g];
        T.yd[wbase + g] = gyd[g];
      }
    }
    if (t == 0) {
      T.wg_cnt[w] = ng_w;
      T.wp_cnt[w] = T.fmask ? 0u : np_w;  // (fmask mode: wg_fmask_k turns this window's incidences into masks)
    }
    dbg_done(1, n_w);
    if (T.dbg && t == 0) {
      atomicAdd(&T.dbg[12], t_load - t_start);
      atomicAdd(&T.dbg[13], t_sort - t_load);
      atomicAdd(&T.dbg[14], t_heads - t_sort);
      atomicAdd(&T.dbg[15], __builtin_readcyclecounter() - t_heads);
    }
  }
}


// ---- compaction: windows -> key order --------------------------------------------------------------------------------
// (fmask mode) the windows the sort kernel took leave incidence lists (pinc, poff, ns): their groups' masks from those
__global__ __launch_bounds__(64) void wg_fmask_k(const uint32_t* __restrict__ ovf, uint32_t ovf_cap, WgTemp T) {
  const uint32_t cnt = ovf[0] < ovf_cap ? ovf[0] : ovf_cap;
  for (uint32_t wi = blockIdx.x; wi < cnt; wi += gridDim.x) {
    const uint32_t w = ovf[1 + wi];
    const uint32_t ng = T.wg_cnt[w], wb = T.wbase[w];
    for (uint32_t g = threadIdx.x; g < ng; g += 64) {
      const uint32_t p0 = T.poff[wb + g], n = T.ns[wb + g];
      uint64_t m = 0;
      for (uint32_t i = 0; i < n; ++i) m |= 1ull << (T.pinc[wb + p0 + i] & 0xFFFFu);
      T.fmask[wb + g] = m;
    }
  }
}
struct WgFinal {
  uint64_t* fmask;
  uint64_t *ghi, *glo;
  uint32_t *gmem, *gpoff, *pgrp, *first, *ns, *slot2sg;
  uint16_t* pfile;
  double* yc;
  long long *yxin, *ydin;
  unsigned long long* rep;
  WgDirectOut D;  // (rep == nullptr: the caller writes its results itself)
};
__global__ __launch_bounds__(256) void wg_compact_k(uint32_t nw, WgTemp T, const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ pbase,
                                                   WgFinal F) {
  const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;  // one wave per window (no block-wide step below)
  if (w >= nw) return;
  const uint32_t ng = T.wg_cnt[w], np = T.wp_cnt[w];
  if (!ng) return;
  const uint32_t wb = T.wbase[w], gb = gbase[w], pb = pbase[w];
  for (uint32_t g = lane; g < ng; g += 64) {
    // (every load before the first store: the arrays are not known to be distinct, a load behind a store would wait for its turn)
    const uint32_t sg = gb + g;
    const uint64_t hi = T.hi[wb + g], lo = T.lo[wb + g];
    const unsigned long long r = T.rep[wb + g];
    const uint32_t cnt = T.cnt[wb + g], nsv = T.ns[wb + g];
    const uint64_t fm = F.fmask ? T.fmask[wb + g] : 0ull;
    const uint32_t po = F.fmask ? 0u : T.poff[wb + g];
    uint32_t yx = 0, yd = 0;
    if (F.yxin) {  // (group partials only: plain inputs carry no YX / YD — the arrays are null and the writers of the results know)
      yx = T.yx[wb + g];
      yd = T.yd[wb + g];
    }
    F.ghi[sg] = hi;
    F.glo[sg] = lo;
    F.rep[sg] = r;
    F.gmem[sg] = (uint32_t)(r & 0xFFFFFFFFull);
    F.yc[sg] = (double)cnt;
    F.ns[sg] = nsv;
    if (F.yxin) {
      F.yxin[sg] = (long long)yx;
      F.ydin[sg] = (long long)yd;
    }
    F.first[sg] = sg;
    if (F.D.rep && sg < F.D.cap) {  // the caller's results in key order (col_write_k's lines; it rewrites the members of tie sets)
      const int32_t st = (int32_t)(uint32_t)((hi >> 2) & 0x7FFFFFFFull);
      F.D.rep[sg] = (uint32_t)(r & 0xFFFFFFFFull);
      if (F.D.rep_effend) F.D.rep_effend[sg] = (int32_t)(uint32_t)(r >> 32);
      F.D.yc[sg] = (double)cnt;
      F.D.yx[sg] = (F.yxin ? (int64_t)yx : 0ll) + (int64_t)nsv;
      if (F.D.g_start) F.D.g_start[sg] = st;
      if (F.D.g_end) F.D.g_end[sg] = st + (int32_t)(uint32_t)(lo >> 32) - 1;
      if (F.D.g_key) {
        const uint32_t h32 = (uint32_t)lo;
        uint32_t shape = 0;
        if (F.D.strategy == TBK_STRAT_CIGAR || F.D.strategy == TBK_STRAT_CLIP) {  // (-E codes speak of exons, which may hold I and D)
          if (h32 == (0x80000000u | C_M)) shape = 0x80000000u;
          if ((h32 >> 30) == 3u) shape = h32;
        }
        F.D.g_key[2 * (size_t)sg] = hi;
        F.D.g_key[2 * (size_t)sg + 1] = (lo & 0xFFFFFFFF00000000ull) | shape;
      }
    }
    if (F.fmask) F.fmask[sg] = fm;
    if (!F.fmask) F.gpoff[sg] = pb + po;  // (file masks: no incidence list to point into)
    if (F.slot2sg) F.slot2sg[wb + g] = sg;
  }
  for (uint32_t p = lane; p < np; p += 64) {
    const uint32_t pi = T.pinc[wb + p];
    F.pfile[pb + p] = (uint16_t)pi;
    if (F.pgrp) F.pgrp[pb + p] = gb + (pi >> 16);
  }
}
__global__ void wg_tie_k(uint32_t ng, const uint64_t* __restrict__ ghi, const uint64_t* __restrict__ glo, uint8_t* __restrict__ tie) {
  uint32_t sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= ng) return;
  tie[sg] = (sg == 0 || ghi[sg] != ghi[sg - 1] || (glo[sg] >> 32) != (glo[sg - 1] >> 32)) ? 1 : 0;
}
// RAW, sparse list (WgRaw::sparse): one 64-thread block per window walks the window's entries
__global__ __launch_bounds__(256) void wg_finish_sparse_k(uint32_t nw, const uint32_t* __restrict__ wbase, const uint32_t* __restrict__ vcnt,
                                                         const uint32_t* __restrict__ vsrc, const uint32_t* __restrict__ cslot,
                                                         const uint32_t* __restrict__ c2r, const unsigned long long* __restrict__ trep, ColIn I,
                                                         int strategy, uint32_t* __restrict__ err) {
  const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);  // one wave per window
  if (w >= nw) return;
  const uint32_t n = vcnt[w];
  if (!n) return;
  const uint32_t wb = wbase[w];
  for (uint32_t e = threadIdx.x & 63u; e < n; e += 64) {
    const uint32_t j = vsrc[wb + e];
    const uint32_t anchor = (uint32_t)(trep[c2r[cslot[wb + e]]] & 0xFFFFFFFFull);
    if (anchor != j && !strategy_equal(I, strategy, j, anchor)) atomicOr(err, TBK_DERR_COLLISION);
  }
}
// One thread per compacted record: its group's temp slot from (cslot, c2r); the strategy key of every record whose key word is
// not exact (bit 31 of the hash word, col_keys_k) is compared with its group's representative — a member of the group — so a
// hash collision cannot merge two alignments (TBK_DERR_COLLISION: the host reseeds); optionally record -> group (key order).
__global__ void wg_finish_k(uint32_t m, const uint64_t* __restrict__ clo, const uint32_t* __restrict__ cval, const uint32_t* __restrict__ cslot,
                            const uint32_t* __restrict__ c2r, const unsigned long long* __restrict__ trep, const uint32_t* __restrict__ slot2sg,
                            uint32_t* __restrict__ rec_sg, ColIn I, int strategy, uint32_t* __restrict__ err) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint32_t slot = c2r[cslot[j]];
  const bool exact = (clo[j] >> 31) & 1ull;
  if (exact && !rec_sg) return;
  const uint32_t rec = cval[j];
  if (rec_sg) rec_sg[rec] = slot2sg[slot];
  if (!exact) {
    const uint32_t anchor = (uint32_t)(trep[slot] & 0xFFFFFFFFull);
    if (anchor != rec && !strategy_equal(I, strategy, rec, anchor)) atomicOr(err, TBK_DERR_COLLISION);
  }
}

// RAW: one thread per input record; cslot holds a slot only for the records that need this pass (every passing record when the
// record -> group map is wanted, else those whose key word is not exact)
__global__ void wg_finish_raw_k(uint32_t n, const uint32_t* __restrict__ cslot, const uint32_t* __restrict__ c2r,
                                const unsigned long long* __restrict__ trep, const uint32_t* __restrict__ slot2sg, uint32_t* __restrict__ rec_sg,
                                ColIn I, int strategy, uint32_t* __restrict__ err) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t cs = cslot[j];
  if (cs == 0xFFFFFFFFu) return;
  const uint32_t slot = c2r[cs];
  if (rec_sg) rec_sg[j] = slot2sg[slot];
  const uint32_t anchor = (uint32_t)(trep[slot] & 0xFFFFFFFFull);
  if (anchor != j && !strategy_equal(I, strategy, j, anchor)) atomicOr(err, TBK_DERR_COLLISION);
}


// ==================================== group partials: the owner's merge-reduce ====================================
// Multi-GPU owner side: the rows a rank receives are n_runs runs, each already in the reference's output order
// and free of duplicates — shallow data by construction (a group has at most one partial per run), the worst case of the hash
// windows above (every window overflows both tables).  So the windows (same splitter machinery, smaller: < 3 PR_T rows) are
// MERGED instead: a row's place is its index in its own run's piece plus, for every other piece, the number of rows that precede
// it there (one bisection per piece, all in LDS; nothing to do for a single run).  The order is the output order itself — key
// high word (tid, start, strand), span (= end), then, inside a tie, the strategy compare of the reference on the CIGARs
// (tiebrush.cpp:285-345; only evaluated when two different key words meet in one tie set) — so no tie sort follows.  Equal keys
// are adjacent after the merge: one thread per group sums YC / YX, takes the maximum YD and the representative with the smallest
// (effective end, run), and compares every hashed key word's CIGAR with the group head's (TBK_DERR_COLLISION otherwise).
constexpr int PR_NT = 256;
constexpr uint32_t PR_T = 512;          // rows between splitters
constexpr uint32_t PR_KS = 512;         // runs x sample stride
constexpr uint32_t PR_CAP = 3 * PR_T;   // a window holds < 2 T + k s rows
constexpr int PR_E = (int)(PR_CAP / PR_NT);
constexpr uint32_t PR_MAXRUNS = 64;

__global__ void pr_rowkeys_k(uint32_t n2, const int32_t* __restrict__ rows, uint64_t* __restrict__ chi, int32_t* __restrict__ tid,
                             int32_t* __restrict__ pos, uint8_t* __restrict__ strand, uint32_t* __restrict__ ncig) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  const int4 q = *reinterpret_cast<const int4*>(rows + (size_t)i * TBK_PARTIAL_ROW);
  chi[i] = raw_key(q.x, q.y) << 2;
  tid[i] = q.x;
  pos[i] = q.y;
  strand[i] = (uint8_t)((uint32_t)q.z & 0xFFu);
  ncig[i] = (uint32_t)q.z >> 8;
}
__global__ void pr_tail_k(uint32_t n2, const uint64_t* __restrict__ total, uint32_t* __restrict__ cig_off) { cig_off[n2] = (uint32_t)*total; }

struct PrTemp {      // per window, at the window's row base
  uint32_t* rep;     // row of the representative
  double* yc;
  uint32_t *yx, *yd;
  uint64_t *khi, *klo;  // the group's key words as the merge holds them (place + strand code; span + key word)
  uint32_t* wg_cnt;     // [nw] groups of the window
};

__global__ __launch_bounds__(PR_NT) void pr_merge_k(const int32_t* __restrict__ rows, const uint32_t* __restrict__ off, uint32_t k, uint32_t nw,
                                                    const uint32_t* __restrict__ wbase, ColIn I, int strategy, PrTemp T,
                                                    uint32_t* __restrict__ err) {
  __shared__ uint64_t khi[PR_CAP], klo[PR_CAP];
  __shared__ uint32_t rsrc[PR_CAP];
  __shared__ uint16_t srt[PR_CAP];
  __shared__ uint8_t pf[PR_CAP];
  __shared__ uint32_t pre[PR_MAXRUNS + 1], base0[PR_MAXRUNS];
  __shared__ uint32_t sm_u[PR_NT / 64];
  const uint32_t w = blockIdx.x, t = threadIdx.x;
  const uint32_t wb = wbase[w];
  const uint32_t n_w = wbase[w + 1] - wb;
  if (n_w == 0 || n_w > PR_CAP) {
    if (t == 0) {
      T.wg_cnt[w] = 0;
      if (n_w) atomicOr(err, TBK_DERR_BIGBUCKET);  // more partials on one base than a block merges: the caller takes the general path
    }
    return;
  }
  if (t < 64) {  // pieces of the window: lengths, prefix, first row (k <= 64: one wave)
    uint32_t a = 0, len = 0;
    if (t < k) {
      a = off[(size_t)w * k + t];
      const uint32_t b = off[(size_t)(w + 1) * k + t];
      len = b > a ? b - a : 0u;
    }
    const uint32_t inc = wave_incl_sum(len);
    if (t < k) {
      pre[t] = inc - len;
      base0[t] = a;
    }
    if (t == 0) pre[k] = n_w;
  }
  __syncthreads();
  for (uint32_t e = t; e < n_w; e += PR_NT) {
    const uint32_t f = piece_of(pre, k, e);
    const uint32_t src = base0[f] + (e - pre[f]);
    const int4* rp = reinterpret_cast<const int4*>(rows + (size_t)src * TBK_PARTIAL_ROW);
    const int4 q0 = rp[0], q2 = rp[2];
    khi[e] = ((uint64_t)(uint32_t)(q0.x + 1) << 33) | ((uint64_t)(uint32_t)(q0.y + 1) << 2) | strand_code((uint8_t)((uint32_t)q0.z & 0xFFu));
    klo[e] = ((uint64_t)(uint32_t)q2.y << 32) | (uint32_t)q2.z;
    rsrc[e] = src;
    pf[e] = (uint8_t)f;
  }
  __syncthreads();
  // three-way compare of element y with (xh, xl, xsrc) in the output order; 0 = the same group
  auto cmp = [&](uint32_t y, uint64_t xh, uint64_t xl, uint32_t xsrc) -> int {
    const uint64_t yh = khi[y];
    if (yh != xh) return yh < xh ? -1 : 1;
    const uint64_t yl = klo[y];
    if (yl == xl) return 0;
    if ((yl >> 32) != (xl >> 32)) return yl < xl ? -1 : 1;
    return strategy_cmp(I, strategy, rsrc[y], xsrc);  // one tie set, two alignments: the reference's compare on the CIGARs
  };
  for (uint32_t e = t; e < n_w; e += PR_NT) {
    const uint32_t r = pf[e];
    const uint64_t xh = khi[e], xl = klo[e];
    const uint32_t xsrc = rsrc[e];
    uint32_t rank = e - pre[r];
    for (uint32_t s2 = 0; s2 < k; ++s2) {
      if (s2 == r) continue;
      uint32_t lo = pre[s2], hi = pre[s2 + 1];
      const uint32_t first = lo;
      const int lim = s2 < r ? 0 : -1;  // rows of earlier runs with an equal key come first
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (cmp(mid, xh, xl, xsrc) <= lim)
          lo = mid + 1;
        else
          hi = mid;
      }
      rank += lo - first;
    }
    srt[rank] = (uint16_t)e;
  }
  __syncthreads();
  // ---- heads and groups: thread t owns the eu consecutive sorted positions from t * eu ----
  const uint32_t eu = (n_w + PR_NT - 1) / PR_NT;
  const uint32_t q0 = t * eu;
  uint32_t hf = 0;
  {
    uint64_t ph = 0, pl = 0;
    if (q0 > 0 && q0 <= n_w) {
      const uint32_t pe = srt[q0 - 1];
      ph = khi[pe];
      pl = klo[pe];
    }
#pragma unroll
    for (int u = 0; u < PR_E; ++u) {
      const uint32_t q = q0 + (uint32_t)u;
      if ((uint32_t)u < eu && q < n_w) {
        const uint32_t e = srt[q];
        const uint64_t h = khi[e], l = klo[e];
        if (q == 0 || h != ph || l != pl) hf |= 1u << u;
        ph = h;
        pl = l;
      }
    }
  }
  uint32_t tot;
  uint32_t gid = block_excl_sum<uint32_t, PR_NT>((uint32_t)__builtin_popcount(hf), sm_u, &tot);
#pragma unroll
  for (int u = 0; u < PR_E; ++u) {
    if (!((hf >> u) & 1u)) continue;
    const uint32_t q = q0 + (uint32_t)u;
    const uint32_t e0 = srt[q];
    const uint64_t h = khi[e0], l = klo[e0];
    const bool hashed = !((l >> 31) & 1ull);
    double yc = 0.0;
    uint64_t yx = 0;  // (up to 64 runs of partials below 2^31 each: summed wide, refused below when the sum leaves 32 bits)
    uint32_t yd = 0, best = 0, beff = 0xFFFFFFFFu;
    for (uint32_t m = q; m < n_w; ++m) {  // the group's members: adjacent, one per run at most, earlier runs first
      const uint32_t e = srt[m];
      if (m > q && (khi[e] != h || klo[e] != l)) break;
      const uint32_t src = rsrc[e];
      const int4* rp = reinterpret_cast<const int4*>(rows + (size_t)src * TBK_PARTIAL_ROW);
      const int4 a = rp[0], b = rp[1], c = rp[2];
      const uint32_t eff = (uint32_t)a.w;
      if (m == q || eff < beff) {
        beff = eff;
        best = src;
      }
      yc += (double)(uint32_t)b.z;
      yx += (uint32_t)b.w;
      yd = (uint32_t)c.x > yd && c.x > 0 ? (uint32_t)c.x : yd;
      if (hashed && m > q && !strategy_equal(I, strategy, src, rsrc[e0])) atomicOr(err, TBK_DERR_COLLISION);
    }
    T.rep[wb + gid] = best;
    T.yc[wb + gid] = yc;
    if (yx > 0xFFFFFFFFull) atomicOr(err, TBK_DERR_OVERFLOW);  // TBK_E2BIG: the general path (64-bit sums) takes the tile
    T.yx[wb + gid] = (uint32_t)yx;
    T.yd[wb + gid] = yd;
    T.khi[wb + gid] = h;
    T.klo[wb + gid] = l;
    ++gid;
  }
  if (t == 0) T.wg_cnt[w] = tot;
}

__global__ __launch_bounds__(256) void pr_compact_k(uint32_t nw, const uint32_t* __restrict__ wbase, PrTemp T, const uint32_t* __restrict__ gbase,
                                                   uint32_t cap, uint32_t* __restrict__ o_rep,
                                                   double* __restrict__ o_yc, int64_t* __restrict__ o_yx, int32_t* __restrict__ o_yd,
                                                   int32_t* __restrict__ o_start, int32_t* __restrict__ o_end, uint64_t* __restrict__ o_key,
                                                   int strategy) {
  const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);  // one wave per window
  if (w >= nw) return;
  const uint32_t ng = T.wg_cnt[w];
  if (!ng) return;
  const uint32_t wb = wbase[w], gb = gbase[w];
  for (uint32_t g = threadIdx.x & 63u; g < ng; g += 64) {
    const uint32_t o = gb + g;
    if (o >= cap) continue;
    // (every load before the first store: the temp arrays are not known to be distinct from the outputs)
    const uint32_t r = T.rep[wb + g], yxv = T.yx[wb + g], ydv = T.yd[wb + g];
    const double ycv = T.yc[wb + g];
    const uint64_t h = T.khi[wb + g], l = T.klo[wb + g];
    o_rep[o] = r;
    o_yc[o] = ycv;
    o_yx[o] = (int64_t)yxv;
    o_yd[o] = (int32_t)ydv;
    if (o_start || o_end || o_key) {  // place, span and key word of the group ride with the merge (no gather of the representative's row)
      const int32_t start = (int32_t)((h >> 2) & 0x7FFFFFFFull);
      if (o_start) o_start[o] = start;
      if (o_end) o_end[o] = start + (int32_t)(uint32_t)(l >> 32) - 1;
      if (o_key) {  // tbk_groups_out.g_key of the reduced group
        const uint32_t word = (uint32_t)l;
        uint32_t shape = 0;
        if (strategy == TBK_STRAT_CIGAR || strategy == TBK_STRAT_CLIP) {
          if (word == (0x80000000u | C_M)) shape = 0x80000000u;
          if ((word >> 30) == 3u) shape = word;
        }
        o_key[2 * (size_t)o] = h;
        o_key[2 * (size_t)o + 1] = (l & 0xFFFFFFFF00000000ull) | shape;
      }
    }
  }
}

__global__ __launch_bounds__(64) void wg_spread_sum_k(const unsigned long long* __restrict__ spread, unsigned long long* __restrict__ n_pass,
                                                      unsigned long long* __restrict__ n_slots) {
  const unsigned long long a = wave_sum(spread[(size_t)threadIdx.x * WG_SPREAD_STRIDE]);
  const unsigned long long b = wave_sum(spread[(size_t)(WG_NSPREAD + threadIdx.x) * WG_SPREAD_STRIDE]);
  if (threadIdx.x == 0) {
    *n_pass += a;
    *n_slots += b;
  }
}

}  // namespace

bool tbk_window_supported(uint32_t k) { return k >= 1 && k <= 1024; }

// the offsets matrix of k runs against nW bounds, bound-major in `off` ([nrows * k], nrows = nW + 2)
static int wg_offsets_build(tbk_ctx* ctx, bool raw, const uint64_t* chi, const int32_t* rtid, const int32_t* rpos, const int32_t* ctid,
                            const uint32_t* d_run_off, uint32_t k,
                            uint32_t m, const uint64_t* W, uint32_t nW, uint32_t nrows, uint32_t* off) {
  uint32_t* offT = ws_alloc<uint32_t>(ctx, (size_t)nrows * k);
  if (!offT) return TBK_ENOMEM;
  // chunks per block: eight where that still leaves a few thousand blocks, fewer on small inputs (the owner's side of the multi-rank
  // protocol: 7 M rows as 450 blocks took twice as long as 3 600 blocks of one chunk)
  const uint32_t og = std::min<uint32_t>(WG_OG, std::max<uint32_t>(1u, m / (WG_OC * 2048u)));
  if (raw)
    TBK_LAUNCH(ctx, "wg_offsets", wg_offsets_stream_k<true>, cdiv(m, WG_OC * og), 256, 0, chi, rtid, rpos, ctid, d_run_off, k, m, W, nW, nrows, offT, ctx->d_err, og);
  else
    TBK_LAUNCH(ctx, "wg_offsets", wg_offsets_stream_k<false>, cdiv(m, WG_OC * og), 256, 0, chi, rtid, rpos, ctid, d_run_off, k, m, W, nW, nrows, offT, ctx->d_err, og);
  TBK_LAUNCH(ctx, "wg_offsets_edges", wg_offsets_edges_k, k, 256, 0, d_run_off, k, nrows, offT);
  TBK_LAUNCH(ctx, "wg_offsets_transpose", wg_offsets_transpose_k, dim3(cdiv(nrows, 64u), cdiv(k, 64u)), 256, 0, offT, k, nrows, off);
  return 0;
}

int tbk_partial_reduce_device(tbk_ctx* ctx, int strategy, const int32_t* rows, uint32_t n2, const uint32_t* run_off_host, uint32_t k,
                              const uint32_t* cig, tbk_groups_out* out, tbk_cov_in* view, const uint32_t* md_off, const uint8_t* md,
                              const uint8_t* md_has) {
  if (strategy == TBK_STRAT_FULL && !md_off) return TBK_EINVAL;
  if (k > PR_MAXRUNS) return TBK_EUNSUPPORTED;
  const uint32_t B = 256, m = n2;
  uint64_t* sc = ctx->d_scalars;
  TBK_HIP(hipMemsetAsync(sc, 0, 16 * sizeof(uint64_t), ctx->stream));
  uint32_t* d_run_off = ws_alloc<uint32_t>(ctx, (size_t)k + 1);
  if (!d_run_off) return TBK_ENOMEM;
  {
    void* stage = tbk_stage_acquire(ctx);
    memcpy(stage, run_off_host, (size_t)(k + 1) * 4);
    TBK_HIP(hipMemcpyAsync(d_run_off, stage, (size_t)(k + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    tbk_stage_release(ctx);
  }
  uint64_t* chi = ws_alloc<uint64_t>(ctx, m);
  int32_t* r_tid = ws_alloc<int32_t>(ctx, m);
  int32_t* r_pos = ws_alloc<int32_t>(ctx, m);
  uint8_t* r_strand = ws_alloc<uint8_t>(ctx, m);
  uint32_t* ncig = ws_alloc<uint32_t>(ctx, m);
  uint32_t* cig_off = ws_alloc<uint32_t>(ctx, (size_t)m + 1);
  if (!cig_off) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, "pr_rowkeys", pr_rowkeys_k, cdiv(m, B), B, 0, m, rows, chi, r_tid, r_pos, r_strand, ncig);
  TBK_TRY(tbk_exscan_u32(ctx, ncig, cig_off, m, sc + 23));
  TBK_LAUNCH(ctx, "pr_tail", pr_tail_k, 1, 1, 0, m, sc + 23, cig_off);
  // windows: sample stride s (power of two, k s <= PR_KS), g = PR_T / s samples per splitter
  uint32_t s = 1;
  while (s * 2 * k <= PR_KS && s * 2 <= PR_T) s *= 2;
  const uint32_t g = PR_T / s;
  const uint32_t ns = cdiv(m, s);
  const uint32_t nsp = ns > 1 ? (ns - 1) / g : 0;
  const uint32_t nW = 2 * nsp, nw = nW + 1, nrows = nw + 1;
  uint64_t* W = ws_alloc<uint64_t>(ctx, (size_t)nW + 1);
  uint32_t* off = ws_alloc<uint32_t>(ctx, (size_t)nrows * k);
  uint32_t* wbase = ws_alloc<uint32_t>(ctx, (size_t)nw + 1);
  uint32_t* gbase = ws_alloc<uint32_t>(ctx, nw);
  PrTemp T;
  T.rep = ws_alloc<uint32_t>(ctx, m);
  T.yc = ws_alloc<double>(ctx, m);
  T.yx = ws_alloc<uint32_t>(ctx, m);
  T.yd = ws_alloc<uint32_t>(ctx, m);
  T.khi = ws_alloc<uint64_t>(ctx, m);
  T.klo = ws_alloc<uint64_t>(ctx, m);
  T.wg_cnt = ws_alloc<uint32_t>(ctx, nw);
  if (!W || !off || !wbase || !gbase || !T.wg_cnt || !T.klo) return TBK_ENOMEM;
  if (nsp) {  // the samples' ranks by bisection in the sorted runs (pr_sample_rank_k): every g-th sample in rank order, then the bounds
    uint64_t* Z = ws_alloc<uint64_t>(ctx, (size_t)ns / g + 2);
    if (!Z) return TBK_ENOMEM;
    TBK_LAUNCH(ctx, "pr_sample_rank", pr_sample_rank_k, cdiv(ns, B), B, 0, chi, m, s, ns, d_run_off, k, g, Z);
    TBK_LAUNCH(ctx, "wg_split", wg_split_k, cdiv(nsp, B), B, 0, Z, 1u, nsp, W);
  }
  TBK_TRY(wg_offsets_build(ctx, false, chi, nullptr, nullptr, nullptr, d_run_off, k, m, W, nW, nrows, off));
  TBK_LAUNCH(ctx, "wg_rowsum", wg_rowsum_k, cdiv(nrows, 4), 256, 0, off, k, nrows, wbase);
  ColIn I{};
  I.n = m;
  I.k = k;
  I.pos = r_pos;
  I.cig_off = cig_off;
  I.cig = cig;
  I.md_off = md_off;
  I.md = md;
  I.md_has = md_has;
  TBK_LAUNCH(ctx, "pr_merge", pr_merge_k, nw, PR_NT, 0, rows, off, k, nw, wbase, I, strategy, T, ctx->d_err);
  TBK_TRY(tbk_exscan_u32(ctx, T.wg_cnt, gbase, nw, sc + 1));
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb & TBK_DERR_BIGBUCKET) {
    ctx->last_error = "more partials start on one base than a workgroup merges";
    return TBK_E2BIG;
  }
  if (eb) return tbk_derr_to_status(ctx, eb);
  const uint32_t ng = (uint32_t)ctx->h_scalars[1];
  out->n_groups = ng;
  if (ng > out->cap_groups) return TBK_E2BIG;
  if (ng == 0) return 0;
  uint64_t* okey = out->g_key;
  if (view && !okey) {  // the tiecov input of the reduced groups is built from their keys (and comes with tiecov's first pass)
    okey = ws_alloc<uint64_t>(ctx, 2 * (size_t)ng);
    if (!okey) return TBK_ENOMEM;
  }
  TBK_LAUNCH(ctx, "pr_compact", pr_compact_k, cdiv(nw, 4u), 256, 0, nw, wbase, T, gbase, out->cap_groups, out->rep, out->yc, out->yx, out->yd,
             out->g_start, out->g_end, okey, strategy);
  if (view) return tbk_cov_view_build(ctx, r_tid, r_pos, r_strand, cig_off, cig, out->rep, out->yc, out->yx, ng, view, okey);
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "partial_reduce");
}


int tbk_window_groups(tbk_ctx* ctx, const tbkd::ColIn& I, int strategy, const uint64_t* chi, const uint64_t* clo, const uint32_t* cval,
                      const uint32_t* ceff, uint32_t m, const uint32_t* d_run_off, uint64_t* scratch_hi, uint64_t* scratch_lo,
                      bool want_rec_sg, uint64_t seed, WgOut* out, uint32_t* err_bits, const tbkd::ColOpt* raw_opt, bool part,
                      const WgDirectOut* direct) {
  const uint32_t k = I.k;
  const uint32_t B = 256;
  *err_bits = 0;
  const bool raw = raw_opt != nullptr;  // the records themselves are the runs: m = I.n, d_run_off = I.file_off, chi .. ceff unused
  if (!tbk_window_supported(k) || m == 0) return TBK_EINVAL;
  if (part && (!raw || k > 64 || !I.prio_hi || !I.yc_in || !I.yx_in || !I.yd_in)) return TBK_EINVAL;
  // Everything allocated from here to the group arrays (`out`) is this stage's scratch — 80 bytes per record on the raw path — and
  // dead once its kernels have run: it comes from the end of the arena, the group arrays (a few dozen bytes per GROUP) from the
  // bottom, so that a deferred YD stage pins the bottom only and the calls behind the collapse find the rest of the arena free
  struct TopMode {
    tbk_ctx* c;
    ~TopMode() {
      c->ws_top_mode = false;
      c->ws_top = 0;  // (the kernels that read the scratch are queued: dead in stream order)
    }
  } top_mode{ctx};
  ctx->ws_top_mode = true;
  WgRaw R{};
  unsigned long long* spread = nullptr;  // (raw) the spread counters of passing records and verification entries
  if (raw) {
    R.I = I;
    if (!R.I.cig) R.I.cig = R.I.cig_off;  // (no CIGAR word in the tile: the loads of absent words read word 0 of the array, wg_raw_l)
    R.O = *raw_opt;
    R.O.seed = seed;
    spread = ws_alloc<unsigned long long>(ctx, 2 * (size_t)WG_NSPREAD * WG_SPREAD_STRIDE);
    if (!spread) return TBK_ENOMEM;
    TBK_HIP(hipMemsetAsync(spread, 0, 2 * (size_t)WG_NSPREAD * WG_SPREAD_STRIDE * sizeof(unsigned long long), ctx->stream));
    R.n_pass = spread;
    R.all_slots = want_rec_sg ? 1u : 0u;
    R.sparse = (!want_rec_sg && strategy != TBK_STRAT_FULL && !ctx->dbg.wg_dense_verify) ? 1u : 0u;  // (-L has no exact key words: every record is verified)
    R.n_slots = spread + (size_t)WG_NSPREAD * WG_SPREAD_STRIDE;
    scratch_hi = ws_alloc<uint64_t>(ctx, m);
    scratch_lo = ws_alloc<uint64_t>(ctx, m);
    if (!scratch_lo) return TBK_ENOMEM;
  }
  // sample stride s (power of two, k s <= WG_KS) and samples per splitter g = T / s
  uint32_t s = 1;
  while (s * 2 * k <= WG_KS && s * 2 <= WG_T) s *= 2;
  const uint32_t g = WG_T / s;
  const uint32_t ns = cdiv(m, s);
  const uint32_t nsp = ns > 1 ? (ns - 1) / g : 0;  // splitters Y[g], Y[2g], ... < ns
  const uint32_t nW = 2 * nsp;                       // bounds
  const uint32_t nw = nW + 1;                        // windows
  const uint32_t nrows = nw + 1;
  uint64_t* W = ws_alloc<uint64_t>(ctx, nW + 1);
  uint32_t* off = ws_alloc<uint32_t>(ctx, (size_t)nrows * k);
  if (!W || !off) return TBK_ENOMEM;
  int32_t* ctid = nullptr;  // (raw) the reference id of every chunk of WG_TC records, where it has one
  if (raw) {
    const uint32_t nchunks = cdiv(m, WG_TC);
    ctid = ws_alloc<int32_t>(ctx, nchunks);
    if (!ctid) return TBK_ENOMEM;
    TBK_LAUNCH(ctx, "wg_sample", wg_tidchunks_k, cdiv(nchunks, B), B, 0, I.tid, m, d_run_off, k, nchunks, ctid);
  }
  if (nsp) {
    uint64_t* Y = ws_alloc<uint64_t>(ctx, ns);
    uint64_t* Y2 = ws_alloc<uint64_t>(ctx, ns);
    if (!Y2) return TBK_ENOMEM;
    if (raw)
      TBK_LAUNCH(ctx, "wg_sample", wg_sample_raw_k, cdiv(ns, B), B, 0, I.tid, I.pos, ctid, m, s, ns, Y);
    else
      TBK_LAUNCH(ctx, "wg_sample", wg_sample_k, cdiv(ns, B), B, 0, chi, m, s, ns, Y);
    TBK_TRY(tbk_radix_sort_w64(ctx, &Y, &Y2, ns, ~0ull, false));
    TBK_LAUNCH(ctx, "wg_split", wg_split_k, cdiv(nsp, B), B, 0, Y, g, nsp, W);
  }
  TBK_TRY(wg_offsets_build(ctx, raw, chi, I.tid, I.pos, ctid, d_run_off, k, m, W, nW, nrows, off));
  WgTemp T;
  T.hi = scratch_hi;
  T.lo = scratch_lo;
  T.cnt = ws_alloc<uint32_t>(ctx, m);
  T.ns = ws_alloc<uint32_t>(ctx, m);
  T.poff = ws_alloc<uint32_t>(ctx, m);
  T.rep = ws_alloc<unsigned long long>(ctx, m);
  T.pinc = ws_alloc<uint32_t>(ctx, m);
  T.wg_cnt = ws_alloc<uint32_t>(ctx, nw);
  T.wp_cnt = ws_alloc<uint32_t>(ctx, nw);
  T.wbase = ws_alloc<uint32_t>(ctx, (size_t)nw + 1);
  T.cslot = ws_alloc<uint32_t>(ctx, m);
  T.fmask = nullptr;
  if (raw && !part && k <= 64 && tbk_yd_by_list(ctx, k)) {
    T.fmask = ws_alloc<uint64_t>(ctx, m);
    if (!T.fmask) return TBK_ENOMEM;
  }
  T.vsrc = nullptr;
  T.vcnt = nullptr;
  if (raw && R.sparse) {
    T.vsrc = ws_alloc<uint32_t>(ctx, m);
    T.vcnt = ws_alloc<uint32_t>(ctx, nw);
    if (!T.vsrc || !T.vcnt) return TBK_ENOMEM;
    TBK_HIP(hipMemsetAsync(T.vcnt, 0, (size_t)nw * 4, ctx->stream));
  }
  T.c2r = ws_alloc<uint32_t>(ctx, m);
  uint32_t* gbase = ws_alloc<uint32_t>(ctx, nw);
  uint32_t* pbase = ws_alloc<uint32_t>(ctx, nw);
  T.dbg = nullptr;
  T.yx = T.yd = nullptr;
  if (part) {
    T.yx = ws_alloc<uint32_t>(ctx, m);
    T.yd = ws_alloc<uint32_t>(ctx, m);
    if (!T.yd) return TBK_ENOMEM;
  }
  if (!T.pinc || !pbase || !T.c2r) return TBK_ENOMEM;
  if (raw && !R.sparse) TBK_HIP(hipMemsetAsync(T.cslot, 0xFF, (size_t)m * 4, ctx->stream));  // (only the records wg_finish_raw_k must visit get a slot)
  WgIn In{chi, clo, cval, ceff, off, W, k, nw, ctx->dbg.wg_rank_merge ? 1u : 0u};
  const uint32_t nwords = cdiv(k, 32);
  // LDS of the hash kernel: the pieces' tables (8 k + 4 bytes) and the group table share WG_LDS_HASH (four blocks per CU)
  const uint32_t gcap = (WG_LDS_HASH - (8u * k + 8u)) / (44u + 4u * nwords);
  const uint32_t lds_hash = gcap * (44u + 4u * nwords) + 8u * k + 8u;
  TBK_LAUNCH(ctx, "wg_rowsum", wg_rowsum_k, cdiv(nrows, 4), 256, 0, off, k, nrows, T.wbase);
  const uint32_t ovf_cap = nw;
  uint32_t* ovf = ws_alloc<uint32_t>(ctx, (size_t)ovf_cap + 1);
  uint32_t* ovf2 = ws_alloc<uint32_t>(ctx, (size_t)ovf_cap + 1);
  if (!ovf || !ovf2) return TBK_ENOMEM;
  TBK_HIP(hipMemsetAsync(ovf, 0, sizeof(uint32_t), ctx->stream));
  TBK_HIP(hipMemsetAsync(ovf2, 0, sizeof(uint32_t), ctx->stream));
  const uint32_t* ovf1_dbg = ovf;  // (TBK_WG_DEBUG: the first tier's overflow count)
  // the windows that hold records (every splitter owns two bounds, so about half of the windows are empty by construction):
  // the hash kernel's grid is exactly those — an empty block would hold a table's worth of LDS while it finds out
  uint32_t* wlist = ws_alloc<uint32_t>(ctx, nw);
  if (!wlist) return TBK_ENOMEM;
  uint64_t* scw = ctx->d_scalars;
  TBK_HIP(hipMemsetAsync(scw + 3, 0, sizeof(uint64_t), ctx->stream));
  TBK_HIP(hipMemsetAsync(T.wg_cnt, 0, (size_t)nw * 4, ctx->stream));
  TBK_HIP(hipMemsetAsync(T.wp_cnt, 0, (size_t)nw * 4, ctx->stream));
  TBK_LAUNCH(ctx, "wg_list", wg_list_k, cdiv(nw, 1024u), 1024, 0, nw, T.wbase, wlist, (unsigned long long*)(scw + 3));
  {
    uint32_t eb0 = 0;
    TBK_TRY(tbk_sync_err(ctx, &eb0));
    if (eb0) {  // (RAW: an input this form does not take — the offsets matrix is not to be trusted; the caller decides)
      *err_bits = eb0;
      return 0;
    }
  }
  const uint32_t nw_live = (uint32_t)ctx->h_scalars[3];
  if (nw_live) {
#define WG_BY_STRATEGY(LAUNCH)          \
  switch (strategy) {                  \
    case TBK_STRAT_CIGAR:              \
      LAUNCH(TBK_STRAT_CIGAR);         \
      break;                           \
    case TBK_STRAT_FULL:               \
      LAUNCH(TBK_STRAT_FULL);          \
      break;                           \
    case TBK_STRAT_CLIP:               \
      LAUNCH(TBK_STRAT_CLIP);          \
      break;                           \
    default:                           \
      LAUNCH(TBK_STRAT_EXON);          \
      break;                           \
  }
#define WG_L_HASH(S) TBK_LAUNCH(ctx, "wg_hash", (wg_hash_k<true, S, 0>), nw_live, WG_NT, lds_hash, In, R, T, gcap, nwords, seed, wlist, ovf, ovf_cap, ctx->d_err)
#define WG_L_HASH64(S)                                                                                                                       \
  TBK_LAUNCH(ctx, "wg_hash", (wg_hash_k<true, S, WG_GC64>), nw_live, WG_NT, WG_GC64 * 52u + 8u * 64u + 8u, In, R, T, gcap, nwords, seed, wlist, ovf, \
             ovf_cap, ctx->d_err)
    if (part) {  // group partials: one instantiation, the strategy read from the options
      R.O.strategy = strategy;
      TBK_LAUNCH(ctx, "wg_hash", (wg_hash_k<true, -1, WG_GC64, true>), nw_live, WG_NT, WG_GC64 * 52u + 8u * 64u + 8u, In, R, T, gcap, nwords, seed, wlist,
                 ovf, ovf_cap, ctx->d_err);
    } else if (raw && k <= 64 && !T.dbg) {  // compile-time table layout
      WG_BY_STRATEGY(WG_L_HASH64)
    } else if (raw) {
      WG_BY_STRATEGY(WG_L_HASH)
    } else {
      TBK_LAUNCH(ctx, "wg_hash", (wg_hash_k<false, -1, 0>), nw_live, WG_NT, lds_hash, In, R, T, gcap, nwords, seed, wlist, ovf, ovf_cap, ctx->d_err);
    }
    const uint32_t gcap2 = (WG_LDS_HASH2 - (8u * k + 8u)) / (44u + 4u * nwords);
    const uint32_t lds_hash2 = gcap2 * (44u + 4u * nwords) + 8u * k + 8u;
    if (gcap2 < 65536u) {  // (slot numbers are 16-bit in the ranking)
#define WG_L_HASH2(S)                                                                                                                      \
  TBK_LAUNCH(ctx, "wg_hash2", (wg_hash2_k<true, S, 0>), std::min<uint32_t>(nw_live, 512u), WG_NT, lds_hash2, In, R, T, gcap2, nwords, seed, ovf, ovf2, \
             ovf_cap, ctx->d_err)
#define WG_L_HASH2_64(S)                                                                                                                     \
  TBK_LAUNCH(ctx, "wg_hash2", (wg_hash2_k<true, S, WG_GC64_2>), std::min<uint32_t>(nw_live, 512u), WG_NT, WG_GC64_2 * 52u + 8u * 64u + 8u, In, R, \
             T, gcap2, nwords, seed, ovf, ovf2, ovf_cap, ctx->d_err)
      if (part) {
        TBK_LAUNCH(ctx, "wg_hash2", (wg_hash2_k<true, -1, WG_GC64_2, true>), std::min<uint32_t>(nw_live, 512u), WG_NT,
                   WG_GC64_2 * 52u + 8u * 64u + 8u, In, R, T, gcap2, nwords, seed, ovf, ovf2, ovf_cap, ctx->d_err);
      } else if (raw && k <= 64 && !T.dbg) {
        WG_BY_STRATEGY(WG_L_HASH2_64)
      } else if (raw) {
        WG_BY_STRATEGY(WG_L_HASH2)
      } else {
        TBK_LAUNCH(ctx, "wg_hash2", (wg_hash2_k<false, -1, 0>), std::min<uint32_t>(nw_live, 512u), WG_NT, lds_hash2, In, R, T, gcap2, nwords, seed, ovf,
                   ovf2, ovf_cap, ctx->d_err);
      }
      ovf = ovf2;
    }
  }
#define WG_L_SORT(S) TBK_LAUNCH(ctx, "wg_sort", (wg_sort_k<true, S>), std::min<uint32_t>(nw, 1024u), WG_NT, 0, In, R, T, I, strategy, ovf, ovf_cap, ctx->d_err)
  if (part) {
    TBK_LAUNCH(ctx, "wg_sort", (wg_sort_k<true, -1, true>), std::min<uint32_t>(nw, 1024u), WG_NT, 0, In, R, T, I, strategy, ovf, ovf_cap, ctx->d_err);
  } else if (raw) {
    WG_BY_STRATEGY(WG_L_SORT)
  } else {
    TBK_LAUNCH(ctx, "wg_sort", (wg_sort_k<false, -1>), std::min<uint32_t>(nw, 1024u), WG_NT, 0, In, R, T, I, strategy, ovf, ovf_cap, ctx->d_err);
  }
  if (T.fmask) TBK_LAUNCH(ctx, "wg_fmask", wg_fmask_k, std::min<uint32_t>(nw, 1024u), 64, 0, ovf, ovf_cap, T);
  uint64_t* sc = ctx->d_scalars;
  if (spread) TBK_LAUNCH(ctx, "wg_spread_sum", wg_spread_sum_k, 1, 64, 0, spread, (unsigned long long*)(sc + 0), (unsigned long long*)(sc + 6));
  TBK_TRY(tbk_exscan_u32(ctx, T.wg_cnt, gbase, nw, sc + 1));
  TBK_TRY(tbk_exscan_u32(ctx, T.wp_cnt, pbase, nw, sc + 2));
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  *err_bits = eb;
  if (T.dbg) {
    unsigned long long h[32];
    TBK_HIP(hipMemcpy(h, T.dbg, sizeof(h), hipMemcpyDeviceToHost));
    uint32_t o1 = 0, o2 = 0;
    TBK_HIP(hipMemcpy(&o1, ovf1_dbg, 4, hipMemcpyDeviceToHost));
    TBK_HIP(hipMemcpy(&o2, ovf, 4, hipMemcpyDeviceToHost));
    fprintf(stderr, "wg windows: %u with records, %u overflowed the first table (%u slots), %u the second\n", nw_live, o1, gcap, o2);
    static const char* ph[11] = {"prologue+init", "-", "key loads", "probe", "barrier 1", "atomics", "barrier 2", "-",
                                 "rank sort", "emit", "-"};
    unsigned long long tot = 0;
    for (int i = 0; i < 11; ++i) tot += h[16 + i];
    for (int i = 0; i < 11; ++i)
      fprintf(stderr, "wg_hash phase %-15s %8.1f Mcyc %5.1f %%\n", ph[i], h[16 + i] / 1e6, tot ? 100.0 * h[16 + i] / tot : 0.0);
    const char* nm[3] = {"empty", "sort", "hash"};
    fprintf(stderr, "wg_window lds phases (Mcyc): load %.1f  sort %.1f  heads+verify %.1f  reduce+write %.1f\n", h[12] / 1e6, h[13] / 1e6,
            h[14] / 1e6, h[15] / 1e6);
    for (int i = 0; i < 3; ++i)
      fprintf(stderr, "wg_window %-6s blocks %8llu records %10llu  cycles/block avg %9.0f max %9llu  (total %.1f Mcyc)\n", nm[i], h[i * 4 + 1],
              h[i * 4 + 2], h[i * 4 + 1] ? (double)h[i * 4] / (double)h[i * 4 + 1] : 0.0, h[i * 4 + 3], (double)h[i * 4] / 1e6);
  }
  if (eb) return 0;  // the caller decides (reseed / sort path / error)
  const uint32_t ng = (uint32_t)ctx->h_scalars[1], np = (uint32_t)ctx->h_scalars[2];
  out->ng = ng;
  out->np = np;
  ctx->ws_top_mode = false;  // the group arrays: from the bottom
  out->ghi = ws_alloc<uint64_t>(ctx, ng);
  out->glo = ws_alloc<uint64_t>(ctx, ng);
  out->gmem = ws_alloc<uint32_t>(ctx, ng);
  out->gpoff = ws_alloc<uint32_t>(ctx, ng);
  out->gfmask = T.fmask ? ws_alloc<uint64_t>(ctx, ng) : nullptr;
  if (T.fmask && !out->gfmask) return TBK_ENOMEM;
  out->pfile = ws_alloc<uint16_t>(ctx, np);
  out->pgrp = tbk_yd_by_list(ctx, k) ? nullptr : ws_alloc<uint32_t>(ctx, np);
  out->yc = ws_alloc<double>(ctx, ng);
  out->ns = ws_alloc<uint32_t>(ctx, ng);
  out->yxin = part ? ws_alloc<long long>(ctx, ng) : nullptr;
  out->ydin = part ? ws_alloc<long long>(ctx, ng) : nullptr;
  if (part && !out->ydin) return TBK_ENOMEM;
  out->rep = ws_alloc<unsigned long long>(ctx, ng);
  out->first = ws_alloc<uint32_t>(ctx, ng);
  out->tie = ws_alloc<uint8_t>(ctx, ng);
  out->rec_sg = want_rec_sg ? ws_alloc<uint32_t>(ctx, I.n) : nullptr;
  ctx->ws_top_mode = true;
  uint32_t* slot2sg = want_rec_sg ? ws_alloc<uint32_t>(ctx, m) : nullptr;  // (scratch)
  ctx->ws_top_mode = false;
  if (want_rec_sg && !slot2sg) return TBK_ENOMEM;
  if (!out->tie || (want_rec_sg && !out->rec_sg)) return TBK_ENOMEM;
  if (ng) {
    WgFinal F{out->gfmask, out->ghi, out->glo, out->gmem, out->gpoff, out->pgrp, out->first, out->ns, slot2sg, out->pfile, out->yc, out->yxin,
              out->ydin, out->rep, (raw && direct) ? *direct : WgDirectOut{}};
    TBK_LAUNCH(ctx, "wg_compact", wg_compact_k, cdiv(nw, 4u), 256, 0, nw, T, gbase, pbase, F);
    TBK_LAUNCH(ctx, "wg_tie", wg_tie_k, cdiv(ng, B), B, 0, ng, out->ghi, out->glo, out->tie);
    if (want_rec_sg) TBK_HIP(hipMemsetAsync(out->rec_sg, 0xFF, (size_t)I.n * 4, ctx->stream));  // (records that did not pass)
    if (raw) {  // (nothing to verify and no record -> group map wanted: every key word was exact)
      if (ctx->h_scalars[6] != 0 && R.sparse)
        TBK_LAUNCH(ctx, "wg_finish", wg_finish_sparse_k, cdiv(nw, 4u), 256, 0, nw, T.wbase, T.vcnt, T.vsrc, T.cslot, T.c2r, T.rep, I, strategy, ctx->d_err);
      else if (ctx->h_scalars[6] != 0)
        TBK_LAUNCH(ctx, "wg_finish", wg_finish_raw_k, cdiv(m, B), B, 0, m, T.cslot, T.c2r, T.rep, slot2sg, out->rec_sg, I, strategy, ctx->d_err);
    } else
      TBK_LAUNCH(ctx, "wg_finish", wg_finish_k, cdiv(m, B), B, 0, m, clo, cval, T.cslot, T.c2r, T.rep, slot2sg, out->rec_sg, I, strategy,
                 ctx->d_err);
  }
  return tbk_check_launch(ctx, "window_groups");
}
