// scan_op.hpp — device-wide inclusive scan with an arbitrary associative (not necessarily
// commutative) operator over a trivially-copyable T made of 32-bit words.
// Two forms: three launches (per-tile reduce, single-block spine, per-tile down-sweep) or one launch with a decoupled
// look-back (so_single_k below: scans whose elements are expensive to load gain from reading them once; cheap ones do not).
// Load/Store are functors so the per-element work of the caller fuses into the first / last pass.
#pragma once
#include "dev_common.hpp"
#include <stdlib.h>
#include <string.h>

#include "tbk_internal.h"

// Word-wise shuffle of a small POD.  The words go through a by-value array copy (bit_cast), never through a pointer
// to the live object, so the struct stays in registers (an address-taken struct ends up in scratch memory).
template <class T>
struct Words {
  uint32_t w[sizeof(T) / 4];
};
template <class T>
__device__ __forceinline__ T shfl_up_t(const T& v, int d) {
  static_assert(sizeof(T) % 4 == 0, "T must be made of 32-bit words");
  Words<T> a = __builtin_bit_cast(Words<T>, v);
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) a.w[k] = __shfl_up(a.w[k], d, 64);
  return __builtin_bit_cast(T, a);
}
template <class T>
__device__ __forceinline__ T shfl_idx_t(const T& v, int l) {
  Words<T> a = __builtin_bit_cast(Words<T>, v);
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) a.w[k] = __shfl(a.w[k], l, 64);
  return __builtin_bit_cast(T, a);
}

constexpr int SO_NT = 256;
constexpr int SO_E = 8;
constexpr int SO_TILE = SO_NT * SO_E;

// block-wide inclusive scan of one value per thread (thread order), 256 threads.
// returns inclusive result; *excl_valid=false for thread 0 (no predecessor); block total in *total.
template <class T, class Op>
__device__ __forceinline__ T block_incl_scan_op(T v, Op op, T* sm /*>=4*/, T* total) {
  T inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = shfl_up_t(inc, d);
    if ((int)lane_id() >= d) inc = op(o, inc);
  }
  uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 63) sm[w] = inc;
  __syncthreads();
  T tot = sm[0];
  if (w >= 1) {
    T pre = sm[0];
    for (uint32_t k = 1; k < w; ++k) pre = op(pre, sm[k]);
    inc = op(pre, inc);
  }
#pragma unroll
  for (int k = 1; k < SO_NT / 64; ++k) tot = op(tot, sm[k]);
  __syncthreads();
  *total = tot;
  return inc;
}

// Tile staging: global loads/stores are striped (lane-consecutive addresses, coalesced); the scan needs each thread to
// own SO_E consecutive elements, so the tile goes through LDS.  One pad element per SO_E keeps the blocked accesses
// off a single bank group (lane stride 9 elements instead of 8).
__device__ __forceinline__ uint32_t so_pad(uint32_t j) { return j + (j >> 3); }
constexpr int SO_LDS = SO_TILE + SO_TILE / 8;

template <class T, class Op, class Load>
__global__ __launch_bounds__(SO_NT) void so_reduce_k(uint32_t n, Load load, Op op, T ident, T* __restrict__ part) {
  __shared__ T tile[SO_LDS];
  __shared__ T sm[SO_NT / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SO_TILE;
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    tile[so_pad(j)] = (i < n) ? load((uint32_t)i) : ident;
  }
  __syncthreads();
  T acc = ident;
#pragma unroll
  for (int e = 0; e < SO_E; ++e) acc = op(acc, tile[so_pad(threadIdx.x * SO_E + e)]);
  T tot;
  (void)block_incl_scan_op(acc, op, sm, &tot);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// single block: part[b] := exclusive prefix (ident for b == 0)
template <class T, class Op>
__global__ __launch_bounds__(SO_NT) void so_spine_k(T* __restrict__ part, uint32_t nb, Op op, T ident) {
  __shared__ T sm[SO_NT / 64];
  __shared__ T carry_s;
  if (threadIdx.x == 0) carry_s = ident;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += SO_NT) {
    uint32_t i = base + threadIdx.x;
    T v = (i < nb) ? part[i] : ident;
    T tot;
    T inc = block_incl_scan_op(v, op, sm, &tot);
    // exclusive = carry (+) inclusive of the previous thread
    T prev = shfl_up_t(inc, 1);
    __shared__ T wl[SO_NT / 64];
    if (lane_id() == 63) wl[threadIdx.x >> 6] = inc;
    __syncthreads();
    T carry = carry_s;
    T ex;
    if (threadIdx.x == 0)
      ex = carry;
    else if (lane_id() == 0)
      ex = op(carry, wl[(threadIdx.x >> 6) - 1]);
    else
      ex = op(carry, prev);
    if (i < nb) part[i] = ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = op(carry, tot);
    __syncthreads();
  }
}

// store(i, element, inclusive, exclusive) with exclusive == op-prefix of everything before i (ident for i == 0)
// INLINE: `part` holds the raw tile totals (no spine launch) and the block folds the ones before it, in order
constexpr uint32_t SO_INLINE_NB = 2048;
template <class T, class Op, class Load, class Store, bool INLINE>
__global__ __launch_bounds__(SO_NT) void so_down_k(uint32_t n, Load load, Store store, Op op, T ident, const T* __restrict__ part) {
  __shared__ T tile[SO_LDS];
  __shared__ T sm[SO_NT / 64];
  __shared__ T wl[SO_NT / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SO_TILE;
  T mine[SO_E];  // the striped elements this thread loaded: kept for the store phase (the functor's loads are not repeated)
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    mine[e] = (i < n) ? load((uint32_t)i) : ident;
    tile[so_pad(j)] = mine[e];
  }
  __syncthreads();
  T v[SO_E];
  T acc = ident;
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {
    v[e] = tile[so_pad(threadIdx.x * SO_E + e)];
    acc = op(acc, v[e]);
  }
  T tot;
  T inc = block_incl_scan_op(acc, op, sm, &tot);  // (contains the barriers that order the reads above before the writes below)
  T prev = shfl_up_t(inc, 1);
  if (lane_id() == 63) wl[threadIdx.x >> 6] = inc;
  __syncthreads();
  T carry;
  if (INLINE) {  // thread t folds a contiguous slice of the earlier tiles' totals; the block scan keeps the slices in order
    const uint32_t nbp = blockIdx.x, per = (nbp + SO_NT - 1) / SO_NT;
    T c = ident;
    for (uint32_t b = threadIdx.x * per, e = min(nbp, b + per); b < e; ++b) c = op(c, part[b]);
    __syncthreads();  // sm / wl reuse
    (void)block_incl_scan_op(c, op, sm, &carry);
  } else {
    carry = part[blockIdx.x];
  }
  T ex;
  if (threadIdx.x == 0)
    ex = carry;
  else if (lane_id() == 0)
    ex = op(carry, wl[(threadIdx.x >> 6) - 1]);
  else
    ex = op(carry, prev);
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {  // exclusive prefix of every element back into the tile
    tile[so_pad(threadIdx.x * SO_E + e)] = ex;
    ex = op(ex, v[e]);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {  // striped again: the functor's global stores are lane-consecutive
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    if (i < n) {
      T exj = tile[so_pad(j)];
      store((uint32_t)i, mine[e], op(exj, mine[e]), exj);
    }
  }
}

// ---- single pass: decoupled look-back ---------------------------------------------------------------------------------
// One launch instead of three, every element loaded once.  Tiles take a ticket (so a tile's predecessors are resident or done),
// scan locally, publish their aggregate, look back over their predecessors' published values for their exclusive prefix and
// publish their inclusive prefix.  The hand-off follows the "data is the flag" form of cdna_hip_programming.md Guideline 16 R2:
// every 32-bit word of a published value travels in its own 8-byte granule {tag = 1, word}, written and polled with relaxed
// agent-scope atomics (write-through stores, cache-bypassing loads): no fences, no L2 write-back — the fenced look-back tried in
// round 1 was slower than three launches for exactly that cost.  The granules and the ticket are zeroed by one memset per scan.
// Every spin is bounded by wall time (SO_SPIN_TICKS of the 100 MHz constant clock: a predecessor tile that is merely descheduled
// or slow under a shared GPU — several contexts, side streams, RCCL — is waited for; only a tile that never comes raises
// TBK_DERR_INTERNAL in *err, and the kernel then finishes with garbage instead of hanging).
constexpr unsigned long long SO_SPIN_TICKS = 20ull * 100000000ull;  // 20 s
struct SoLookback {
  unsigned long long* agg;  // [nb * W] granules: tile aggregates
  unsigned long long* inc;  // [nb * W] granules: inclusive prefixes
  uint32_t* ticket;
  uint32_t* err;
};
template <class T>
__device__ __forceinline__ void so_publish(unsigned long long* g, uint32_t tile, const T& v) {
  constexpr unsigned W = sizeof(T) / 4;
  Words<T> a = __builtin_bit_cast(Words<T>, v);
#pragma unroll
  for (unsigned k = 0; k < W; ++k)
    __hip_atomic_store(g + (size_t)tile * W + k, (1ull << 32) | a.w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ bool so_peek(const unsigned long long* g, uint32_t tile, T* v) {
  constexpr unsigned W = sizeof(T) / 4;
  Words<T> a;
  bool ok = true;
#pragma unroll
  for (unsigned k = 0; k < W; ++k) {
    const unsigned long long x = __hip_atomic_load(g + (size_t)tile * W + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = ok && (x >> 32) == 1ull;
    a.w[k] = (uint32_t)x;
  }
  *v = __builtin_bit_cast(T, a);
  return ok;
}
// Wave 0 of a tile: publish the tile's aggregate, fold the predecessors' published values (older tiles first) into the
// exclusive prefix, publish the inclusive prefix.  Every lane returns the exclusive prefix.
template <class T, class Op>
__device__ __forceinline__ T so_lookback(const SoLookback& S, uint32_t tix, const T& tot, Op op, T ident) {
  const uint32_t lane = lane_id();
  if (lane == 0) {
    so_publish<T>(S.agg, tix, tot);
    if (tix == 0) so_publish<T>(S.inc, 0u, tot);
  }
  T excl = ident;
  bool failed = false;
  int64_t j = (int64_t)tix - 1;
  while (j >= 0 && !failed) {
    const int64_t jj = j - (int64_t)lane;
    const bool have = jj >= 0;
    T val = ident;
    bool is_inc = false, ok = !have;
    unsigned long long t_spin = 0;
    for (uint32_t spins = 0;; ++spins) {
      if (have && !ok) {
        is_inc = so_peek<T>(S.inc, (uint32_t)jj, &val);
        ok = is_inc || so_peek<T>(S.agg, (uint32_t)jj, &val);
      }
      if (__all(ok)) break;
      if ((spins & 1023u) == 1023u) {  // (the clock is read once per thousand polls)
        const unsigned long long now = wall_clock64();
        if (t_spin == 0) t_spin = now;
        if (__any(now - t_spin > SO_SPIN_TICKS)) {  // something is wrong — never hang (one decision for the whole wave: the vote
          failed = true;                            // above must never run with some lanes gone)
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (failed) break;
    const uint64_t im = __ballot(have && is_inc);
    const int first = im ? (int)__builtin_ctzll(im) : -1;                       // nearest tile with an inclusive prefix
    const uint64_t hm = __ballot(have);
    const int top = first >= 0 ? first : 63 - (int)__builtin_clzll(hm);          // farthest lane that takes part
    T part = ident;
    for (int l = top; l >= 0; --l) part = op(part, shfl_idx_t(val, l));           // older tiles first
    excl = op(part, excl);
    if (first >= 0) break;
    j -= 64;
  }
  if (failed && lane == 0) atomicOr(S.err, TBK_DERR_INTERNAL);
  if (lane == 0 && tix > 0) so_publish<T>(S.inc, tix, op(excl, tot));
  return excl;
}
template <class T, class Op, class Load, class Store>
__global__ __launch_bounds__(SO_NT) void so_single_k(uint32_t n, Load load, Store store, Op op, T ident, SoLookback S) {
  __shared__ T tile[SO_LDS];
  __shared__ T sm[SO_NT / 64];
  __shared__ T wl[SO_NT / 64];
  __shared__ T s_carry;
  __shared__ uint32_t s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(S.ticket, 1u);
  __syncthreads();
  const uint32_t tix = s_tile;
  const uint64_t base = (uint64_t)tix * SO_TILE;
  T mine[SO_E];
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    mine[e] = (i < n) ? load((uint32_t)i) : ident;
    tile[so_pad(j)] = mine[e];
  }
  __syncthreads();
  T v[SO_E];
  T acc = ident;
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {
    v[e] = tile[so_pad(threadIdx.x * SO_E + e)];
    acc = op(acc, v[e]);
  }
  T tot;
  T inc = block_incl_scan_op(acc, op, sm, &tot);
  T prev = shfl_up_t(inc, 1);
  if (lane_id() == 63) wl[threadIdx.x >> 6] = inc;
  if (threadIdx.x < 64) {  // wave 0: publish, look back, publish
    const T excl = so_lookback<T, Op>(S, tix, tot, op, ident);
    if (lane_id() == 0) s_carry = excl;
  }
  __syncthreads();
  const T carry = s_carry;
  T ex;
  if (threadIdx.x == 0)
    ex = carry;
  else if (lane_id() == 0)
    ex = op(carry, wl[(threadIdx.x >> 6) - 1]);
  else
    ex = op(carry, prev);
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {
    tile[so_pad(threadIdx.x * SO_E + e)] = ex;
    ex = op(ex, v[e]);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < SO_E; ++e) {
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    if (i < n) {
      T exj = tile[so_pad(j)];
      store((uint32_t)i, mine[e], op(exj, mine[e]), exj);
    }
  }
}

// ---- two scans in one pass ------------------------------------------------------------------------------------------------
// Bundles (tiecov) and YD chains share a shape: an element is a head when it compares in some way with the exclusive prefix of
// a first scan, and every element then needs a sum over the heads before it (their number; for YD also their exon counts).
// Here a tile looks back twice: once for the prefix of the caller's operator, and — after second(i, element, inclusive,
// exclusive, aux(i)) has produced each element's term of the second scan — once more for the second prefix of the tiles before
// it.  store(i, element, inclusive, exclusive, term, terms_before, aux(i)).
struct SoPlusU32 {
  __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; }
};
template <int E_, class T, class Op, class T2, class Op2, class Load, class Aux, class Second, class Store>
__global__ __launch_bounds__(SO_NT) void so_two_k(uint32_t n, Load load, Aux aux, Second second, Store store, Op op, T ident, Op2 op2, T2 ident2, SoLookback S,
                                                  SoLookback S2) {
  constexpr int TILE_ = SO_NT * E_;
  __shared__ T tile[TILE_ + TILE_ / 8];
  __shared__ T2 tile2[TILE_ + TILE_ / 8];
  __shared__ T sm[SO_NT / 64];
  __shared__ T wl[SO_NT / 64];
  __shared__ T2 sm2[SO_NT / 64];
  __shared__ T2 wl2[SO_NT / 64];
  __shared__ T s_carry;
  __shared__ T2 s_carry2;
  __shared__ uint32_t s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(S.ticket, 1u);
  __syncthreads();
  const uint32_t tix = s_tile;
  const uint64_t base = (uint64_t)tix * TILE_;
  T mine[E_];
  using AuxT = decltype(aux(0u));
  AuxT ax[E_];  // what second() and store() need of the element besides T: loaded here, with the element, so that the later phases
                // wait on no global load
#pragma unroll
  for (int e = 0; e < E_; ++e) {
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    mine[e] = (i < n) ? load((uint32_t)i) : ident;
    ax[e] = aux((uint32_t)(i < n ? i : 0u));
    tile[so_pad(j)] = mine[e];
  }
  __syncthreads();
  {
    T v[E_];
    T acc = ident;
#pragma unroll
    for (int e = 0; e < E_; ++e) {
      v[e] = tile[so_pad(threadIdx.x * E_ + e)];
      acc = op(acc, v[e]);
    }
    T tot;
    T inc = block_incl_scan_op(acc, op, sm, &tot);
    T prev = shfl_up_t(inc, 1);
    if (lane_id() == 63) wl[threadIdx.x >> 6] = inc;
    if (threadIdx.x < 64) {
      const T excl = so_lookback<T, Op>(S, tix, tot, op, ident);
      if (lane_id() == 0) s_carry = excl;
    }
    __syncthreads();
    const T carry = s_carry;
    T ex;
    if (threadIdx.x == 0)
      ex = carry;
    else if (lane_id() == 0)
      ex = op(carry, wl[(threadIdx.x >> 6) - 1]);
    else
      ex = op(carry, prev);
#pragma unroll
    for (int e = 0; e < E_; ++e) {
      tile[so_pad(threadIdx.x * E_ + e)] = ex;
      ex = op(ex, v[e]);
    }
  }
  __syncthreads();
  T2 mine2[E_];
#pragma unroll
  for (int e = 0; e < E_; ++e) {
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    mine2[e] = ident2;
    if (i < n) {
      const T exj = tile[so_pad(j)];
      mine2[e] = second((uint32_t)i, mine[e], op(exj, mine[e]), exj, ax[e]);
    }
    tile2[so_pad(j)] = mine2[e];
  }
  __syncthreads();
  {
    T2 c[E_];
    T2 a2 = ident2;
#pragma unroll
    for (int e = 0; e < E_; ++e) {
      c[e] = tile2[so_pad(threadIdx.x * E_ + e)];
      a2 = op2(a2, c[e]);
    }
    T2 tot2;
    T2 inc2 = block_incl_scan_op(a2, op2, sm2, &tot2);
    T2 prev2 = shfl_up_t(inc2, 1);
    if (lane_id() == 63) wl2[threadIdx.x >> 6] = inc2;
    if (threadIdx.x < 64) {
      const T2 excl2 = so_lookback<T2, Op2>(S2, tix, tot2, op2, ident2);
      if (lane_id() == 0) s_carry2 = excl2;
    }
    __syncthreads();
    const T2 carry2 = s_carry2;
    T2 ex2;
    if (threadIdx.x == 0)
      ex2 = carry2;
    else if (lane_id() == 0)
      ex2 = op2(carry2, wl2[(threadIdx.x >> 6) - 1]);
    else
      ex2 = op2(carry2, prev2);
#pragma unroll
    for (int e = 0; e < E_; ++e) {
      tile2[so_pad(threadIdx.x * E_ + e)] = ex2;
      ex2 = op2(ex2, c[e]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E_; ++e) {
    uint32_t j = (uint32_t)e * SO_NT + threadIdx.x;
    uint64_t i = base + j;
    if (i < n) {
      const T exj = tile[so_pad(j)];
      store((uint32_t)i, mine[e], op(exj, mine[e]), exj, mine2[e], tile2[so_pad(j)], ax[e]);
    }
  }
}

// (E: elements per thread.  Measured on config 3: the bundle scan — 25 M cheap elements — 0.84 ms at 8, 0.60 ms at 4; the YD chain
// scan — 178 M elements of 28 bytes — 2.78 ms at 8, 3.6 ms at 4, 3.5 ms at 16.)
template <int E, class T, class Op, class T2, class Op2, class Load, class Aux, class Second, class Store>
int scan_two_run(tbk_ctx* ctx, const char* name, uint32_t n, Load load, Aux aux, Second second, Store store, Op op, T ident, Op2 op2, T2 ident2) {
  if (n == 0) return 0;
  const uint32_t nb = cdiv(n, (uint32_t)SO_NT * (uint32_t)E);
  constexpr size_t W = sizeof(T) / 4, W2 = sizeof(T2) / 4;
  const size_t words = 2 * (size_t)nb * (W + W2) + 2;  // granules of both scans + the ticket word
  unsigned long long* st = ws_alloc<unsigned long long>(ctx, words);
  if (!st) return TBK_ENOMEM;
  TBK_HIP(hipMemsetAsync(st, 0, words * 8, ctx->stream));
  unsigned long long* g = st + 2;
  unsigned long long* g2 = g + 2 * (size_t)nb * W;
  SoLookback S{g, g + (size_t)nb * W, (uint32_t*)st, ctx->d_err};
  SoLookback S2{g2, g2 + (size_t)nb * W2, (uint32_t*)st, ctx->d_err};
  TBK_LAUNCH(ctx, name, (so_two_k<E, T, Op, T2, Op2, Load, Aux, Second, Store>), nb, SO_NT, 0, n, load, aux, second, store, op, ident, op2, ident2, S, S2);
  return tbk_check_launch(ctx, name);
}

template <class T, class Op, class Load, class Store>
int scan_op_run(tbk_ctx* ctx, const char* name, uint32_t n, Load load, Store store, Op op, T ident, bool single_pass = false) {
  if (n == 0) return 0;
  uint32_t nb = cdiv(n, SO_TILE);
  const int forced = ctx->dbg.scan;  // test hook: scan=lookback / 3pass forces one form for every scan
  const bool lookback = forced == 1 || (forced == 0 && single_pass);
  if (lookback && nb > 1) {
    constexpr size_t W = sizeof(T) / 4;
    const size_t words = 2 * (size_t)nb * W + 2;  // granules + the ticket word (8 bytes)
    unsigned long long* st = ws_alloc<unsigned long long>(ctx, words);
    if (!st) return TBK_ENOMEM;
    TBK_HIP(hipMemsetAsync(st, 0, words * 8, ctx->stream));
    SoLookback S{st + 2, st + 2 + (size_t)nb * W, (uint32_t*)st, ctx->d_err};
    TBK_LAUNCH(ctx, name, (so_single_k<T, Op, Load, Store>), nb, SO_NT, 0, n, load, store, op, ident, S);
    return tbk_check_launch(ctx, name);
  }
  T* part = ws_alloc<T>(ctx, nb);
  if (!part) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, name, (so_reduce_k<T, Op, Load>), nb, SO_NT, 0, n, load, op, ident, part);
  if (nb <= SO_INLINE_NB) {
    TBK_LAUNCH(ctx, name, (so_down_k<T, Op, Load, Store, true>), nb, SO_NT, 0, n, load, store, op, ident, part);
  } else {
    TBK_LAUNCH(ctx, name, (so_spine_k<T, Op>), 1, SO_NT, 0, part, nb, op, ident);
    TBK_LAUNCH(ctx, name, (so_down_k<T, Op, Load, Store, false>), nb, SO_NT, 0, n, load, store, op, ident, part);
  }
  return tbk_check_launch(ctx, name);
}
