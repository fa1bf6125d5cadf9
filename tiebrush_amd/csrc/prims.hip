// prims.hip — device-wide primitives for gfx950: exclusive scans and a 128-bit-key LSD
// radix sort ("wavefront radix over LDS-staged tiles": per-wave ballot matching for the
// stable rank, the sorted sub-tile staged in LDS so that global writes are digit-contiguous).
// All integer work, HBM-bound; no MFMA.
#include "dev_common.hpp"
#include "tbk_internal.h"
#include "rx_w64.hpp"

// =====================================================================================
// exclusive scan (u32 in, u32 or u64 out) — reduce / spine / down-sweep (one block when the input is small).
// Fusing the passes through in-kernel hand-offs (look-back chains, last-block-done tickets) was measured SLOWER on
// gfx950: every agent-scope release writes the XCD's L2 back, which costs more than a kernel boundary.
// =====================================================================================
namespace {
constexpr int SC_NT = 256;
constexpr int SC_E = 8;
constexpr int SC_TILE = SC_NT * SC_E;

__global__ __launch_bounds__(SC_NT) void scan_reduce_k(const uint32_t* __restrict__ in, uint64_t* __restrict__ part, uint32_t n) {
  __shared__ uint64_t sm[8];
  uint64_t base = (uint64_t)blockIdx.x * SC_TILE;
  uint64_t s = 0;
#pragma unroll
  for (int e = 0; e < SC_E; ++e) {
    uint64_t i = base + (uint64_t)e * SC_NT + threadIdx.x;
    if (i < n) s += in[i];
  }
  s = wave_sum(s);
  if (lane_id() == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// single block: exclusive scan of part[0..nb) in place, total -> *total (may be null).  A thread owns 16 consecutive partials
// (their loads are all in flight together): 16 K partials — 33 M elements — per round of the block.
__global__ __launch_bounds__(1024) void scan_spine_k(uint64_t* __restrict__ part, uint32_t nb, uint64_t* __restrict__ total) {
  __shared__ uint64_t sm[16];
  __shared__ uint64_t carry_s;
  constexpr uint32_t E = 16;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 1024 * E) {
    const uint32_t i0 = base + threadIdx.x * E;
    uint64_t v[E];
    uint64_t s = 0;
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      v[e] = (i0 + e < nb) ? part[i0 + e] : 0;
      s += v[e];
    }
    const uint64_t inc = wave_incl_sum(s);
    const uint32_t w = threadIdx.x >> 6;
    if (lane_id() == 63) sm[w] = inc;
    __syncthreads();
    uint64_t wb = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
      const uint64_t x = sm[k];
      if ((uint32_t)k < w) wb += x;
      tot += x;
    }
    uint64_t ex = carry_s + wb + inc - s;
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      if (i0 + e < nb) part[i0 + e] = ex;
      ex += v[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry_s;
}

// n <= SC_SMALL: the whole scan in one block (chunks of 8192 with a running carry)
constexpr uint32_t SC_SMALL = 16384;
template <class OutT>
__global__ __launch_bounds__(1024) void scan_small_k(const uint32_t* __restrict__ in, OutT* __restrict__ out, uint32_t n,
                                                     uint64_t* __restrict__ total) {
  __shared__ uint64_t sm[16];
  constexpr uint32_t E = 8;  // consecutive elements per thread and chunk
  uint64_t carry = 0;
  for (uint32_t c0 = 0; c0 < n; c0 += 1024 * E) {
    uint32_t v[E];
    uint64_t s = 0;
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      const uint32_t i = c0 + threadIdx.x * E + e;
      v[e] = i < n ? in[i] : 0u;
      s += v[e];
    }
    uint64_t tot;
    uint64_t ex = carry + block_excl_sum<uint64_t, 1024>(s, sm, &tot);
#pragma unroll
    for (uint32_t e = 0; e < E; ++e) {
      const uint32_t i = c0 + threadIdx.x * E + e;
      if (i < n) out[i] = (OutT)ex;
      ex += v[e];
    }
    carry += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

// INLINE: no spine launch — `part` holds the raw tile sums and every block adds up the ones before it (nb <= SC_INLINE_NB,
// a few KB out of L2); the last block also writes the grand total.
constexpr uint32_t SC_INLINE_NB = 4096;
template <class OutT, bool INLINE>
__global__ __launch_bounds__(SC_NT) void scan_down_k(const uint32_t* __restrict__ in, OutT* __restrict__ out,
                                                     const uint64_t* __restrict__ part, uint32_t n, uint64_t* __restrict__ total) {
  // striped global access (coalesced), blocked ownership for the scan: the tile is transposed through padded LDS
  __shared__ uint64_t sm[8];
  __shared__ uint32_t tin[SC_TILE + SC_TILE / 8];
  __shared__ uint64_t tex[SC_TILE + SC_TILE / 8];
  const uint64_t base = (uint64_t)blockIdx.x * SC_TILE;
#pragma unroll
  for (int e = 0; e < SC_E; ++e) {
    uint32_t j = (uint32_t)e * SC_NT + threadIdx.x;
    uint64_t i = base + j;
    tin[j + (j >> 3)] = (i < n) ? in[i] : 0u;
  }
  __syncthreads();
  uint32_t v[SC_E];
  uint64_t s = 0;
#pragma unroll
  for (int e = 0; e < SC_E; ++e) {
    uint32_t j = threadIdx.x * SC_E + e;
    v[e] = tin[j + (j >> 3)];
    s += v[e];
  }
  uint64_t carry;
  if (INLINE) {
    uint64_t c = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += SC_NT) c += part[b];
    uint64_t dummy;
    (void)block_excl_sum<uint64_t, SC_NT>(c, sm, &dummy);
    carry = dummy;
  } else {
    carry = part[blockIdx.x];
  }
  uint64_t tot;
  uint64_t ex = block_excl_sum<uint64_t, SC_NT>(s, sm, &tot) + carry;
  if (INLINE && total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = carry + tot;
#pragma unroll
  for (int e = 0; e < SC_E; ++e) {
    uint32_t j = threadIdx.x * SC_E + e;
    tex[j + (j >> 3)] = ex;
    ex += v[e];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < SC_E; ++e) {
    uint32_t j = (uint32_t)e * SC_NT + threadIdx.x;
    uint64_t i = base + j;
    if (i < n) out[i] = (OutT)tex[j + (j >> 3)];
  }
}

template <class OutT>
int exscan_impl(tbk_ctx* ctx, const uint32_t* in, OutT* out, uint32_t n, uint64_t* d_total) {
  if (n == 0) {
    if (d_total) {
      hipError_t e = hipMemsetAsync(d_total, 0, sizeof(uint64_t), ctx->stream);
      if (e != hipSuccess) return TBK_EHIP;
    }
    return 0;
  }
  if (n <= SC_SMALL) {
    TBK_LAUNCH(ctx, "scan_small", (scan_small_k<OutT>), 1, 1024, 0, in, out, n, d_total);
    return tbk_check_launch(ctx, "exscan");
  }
  uint32_t nb = cdiv(n, SC_TILE);
  uint64_t* part = ws_alloc<uint64_t>(ctx, nb);
  if (!part) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, "scan_reduce", scan_reduce_k, nb, SC_NT, 0, in, part, n);
  if (nb <= SC_INLINE_NB) {
    TBK_LAUNCH(ctx, "scan_down", (scan_down_k<OutT, true>), nb, SC_NT, 0, in, out, part, n, d_total);
  } else {
    TBK_LAUNCH(ctx, "scan_spine", scan_spine_k, 1, 1024, 0, part, nb, d_total);
    TBK_LAUNCH(ctx, "scan_down", (scan_down_k<OutT, false>), nb, SC_NT, 0, in, out, part, n, d_total);
  }
  return tbk_check_launch(ctx, "exscan");
}
}  // namespace

int tbk_exscan_u32(tbk_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t n, uint64_t* d_total) {
  return exscan_impl<uint32_t>(ctx, in, out, n, d_total);
}
int tbk_exscan_u32_u64(tbk_ctx* ctx, const uint32_t* in, uint64_t* out, uint32_t n, uint64_t* d_total) {
  return exscan_impl<uint64_t>(ctx, in, out, n, d_total);
}

// =====================================================================================
// 128-bit LSD radix sort, 8-bit digits, stable
// =====================================================================================
namespace {
// AND / OR of all keys: bits where and == or are constant => whole digits of them are skipped
__global__ __launch_bounds__(256) void rx_bits_k(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint32_t n,
                                                 uint64_t* __restrict__ andor /*[4]: and_hi, or_hi, and_lo, or_lo*/) {
  uint64_t ah = ~0ull, oh = 0, al = ~0ull, ol = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t h = hi[i], l = lo[i];
    ah &= h;
    oh |= h;
    al &= l;
    ol |= l;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ah &= __shfl_xor(ah, d, 64);
    oh |= __shfl_xor(oh, d, 64);
    al &= __shfl_xor(al, d, 64);
    ol |= __shfl_xor(ol, d, 64);
  }
  __shared__ uint64_t red[4][4];
  uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 0) {
    red[w][0] = ah;
    red[w][1] = oh;
    red[w][2] = al;
    red[w][3] = ol;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // one set of atomics per block: the four words are a single hot cache line
    for (int k = 1; k < 4; ++k) {
      ah &= red[k][0];
      oh |= red[k][1];
      al &= red[k][2];
      ol |= red[k][3];
    }
    atomicAnd((unsigned long long*)&andor[0], (unsigned long long)ah);
    atomicOr((unsigned long long*)&andor[1], (unsigned long long)oh);
    atomicAnd((unsigned long long*)&andor[2], (unsigned long long)al);
    atomicOr((unsigned long long*)&andor[3], (unsigned long long)ol);
  }
}

// per-tile digit counts -> table[digit * ntiles + tile]
__global__ __launch_bounds__(RX_NT) void rx_hist_k(const uint64_t* __restrict__ word, uint32_t shift, uint32_t n, uint32_t ntiles,
                                                   uint32_t iter, uint32_t* __restrict__ table) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  uint64_t base = (uint64_t)blockIdx.x * RX_SUB * iter;
  for (uint32_t it = 0; it < iter; ++it) {
    uint64_t w[RX_E];
#pragma unroll
    for (int e = 0; e < RX_E; ++e) {  // all loads of the sub-tile in flight before the matching starts
      uint64_t i = base + (uint64_t)it * RX_SUB + (uint64_t)e * RX_NT + threadIdx.x;
      w[e] = i < n ? word[i] : 0ull;
    }
#pragma unroll
    for (int e = 0; e < RX_E; ++e) {
      uint64_t i = base + (uint64_t)it * RX_SUB + (uint64_t)e * RX_NT + threadIdx.x;
      bool valid = i < n;
      uint32_t d = (uint32_t)((w[e] >> shift) & 0xFFu);
      uint64_t peers = match_digit(d, valid);
      if (valid && (peers & lanemask_lt()) == 0) atomicAdd(&h[d], (uint32_t)__popcll(peers));
    }
  }
  __syncthreads();
  table[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// block d: exclusive scan of row d of the table in place; row total -> totals[d]
__global__ __launch_bounds__(256) void rx_rowscan_k(uint32_t* __restrict__ table, uint32_t ntiles, uint32_t* __restrict__ totals) {
  __shared__ uint32_t sm[8];
  __shared__ uint32_t carry_s;
  uint32_t* row = table + (uint64_t)blockIdx.x * ntiles;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < ntiles; base += 256) {
    uint32_t i = base + threadIdx.x;
    uint32_t v = (i < ntiles) ? row[i] : 0;
    uint32_t tot;
    uint32_t ex = block_excl_sum<uint32_t, 256>(v, sm, &tot);
    uint32_t carry = carry_s;
    if (i < ntiles) row[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(RX_NT) void rx_scatter_k(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo,
                                                      const uint32_t* __restrict__ val, uint64_t* __restrict__ hi2,
                                                      uint64_t* __restrict__ lo2, uint32_t* __restrict__ val2, int use_hi,
                                                      uint32_t shift, uint32_t n, uint32_t ntiles, uint32_t iter,
                                                      const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals) {
  __shared__ uint32_t digit_base[256];  // global offset of the next element of each digit for this tile
  __shared__ uint32_t wave_cnt[4][256];
  __shared__ uint32_t lpos[256];        // start of each digit inside the staged sub-tile
  __shared__ uint32_t sm[8];
  __shared__ uint64_t s_hi[RX_SUB];
  __shared__ uint64_t s_lo[RX_SUB];
  __shared__ uint32_t s_val[RX_SUB];

  const uint32_t t = threadIdx.x;
  const uint32_t w = t >> 6;
  {
    uint32_t tot_d = totals[t], dummy;
    uint32_t dbase = block_excl_sum<uint32_t, RX_NT>(tot_d, sm, &dummy);
    digit_base[t] = dbase + table[(uint64_t)t * ntiles + blockIdx.x];
  }
  const uint64_t tile_base = (uint64_t)blockIdx.x * RX_SUB * iter;
  for (uint32_t it = 0; it < iter; ++it) {
    const uint64_t sub_base = tile_base + (uint64_t)it * RX_SUB;
    if (sub_base >= n) break;
#pragma unroll
    for (int k = 0; k < 4; ++k) wave_cnt[k][t] = 0;
    __syncthreads();
    uint64_t khi[RX_E], klo[RX_E];
    uint32_t kv[RX_E], kr[RX_E];
#pragma unroll
    for (int e = 0; e < RX_E; ++e) {
      uint64_t i = sub_base + (uint64_t)w * (64 * RX_E) + (uint64_t)e * 64 + lane_id();
      bool valid = i < n;
      khi[e] = valid ? hi[i] : ~0ull;
      klo[e] = valid ? lo[i] : ~0ull;
      kv[e] = valid ? val[i] : 0u;
    }
#pragma unroll
    for (int e = 0; e < RX_E; ++e) {
      uint64_t i = sub_base + (uint64_t)w * (64 * RX_E) + (uint64_t)e * 64 + lane_id();
      bool valid = i < n;
      uint32_t d = (uint32_t)(((use_hi ? khi[e] : klo[e]) >> shift) & 0xFFu);
      uint64_t peers = match_digit(d, valid);
      uint32_t before = (uint32_t)__popcll(peers & lanemask_lt());
      uint32_t base = valid ? wave_cnt[w][d] : 0u;
      __builtin_amdgcn_wave_barrier();
      if (valid && before == 0) wave_cnt[w][d] = base + (uint32_t)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
      kr[e] = (base + before) | (d << 16) | (valid ? 0u : 0x80000000u);  // rank < 2048 fits 16 bits
    }
    __syncthreads();
    {
      uint32_t c0 = wave_cnt[0][t], c1 = wave_cnt[1][t], c2 = wave_cnt[2][t], c3 = wave_cnt[3][t];
      uint32_t tot = c0 + c1 + c2 + c3, dummy;
      uint32_t lp = block_excl_sum<uint32_t, RX_NT>(tot, sm, &dummy);
      wave_cnt[0][t] = lp;
      wave_cnt[1][t] = lp + c0;
      wave_cnt[2][t] = lp + c0 + c1;
      wave_cnt[3][t] = lp + c0 + c1 + c2;
      lpos[t] = lp;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < RX_E; ++e) {
        if (!(kr[e] & 0x80000000u)) {
          uint32_t d = (kr[e] >> 16) & 0xFFu;
          uint32_t slot = wave_cnt[w][d] + (kr[e] & 0xFFFFu);
          s_hi[slot] = khi[e];
          s_lo[slot] = klo[e];
          s_val[slot] = kv[e];
        }
      }
      __syncthreads();
      uint32_t cnt_sub = (uint32_t)((n - sub_base) < (uint64_t)RX_SUB ? (n - sub_base) : (uint64_t)RX_SUB);
      for (uint32_t s = t; s < cnt_sub; s += RX_NT) {
        uint64_t h = s_hi[s], l = s_lo[s];
        uint32_t d = (uint32_t)(((use_hi ? h : l) >> shift) & 0xFFu);
        uint32_t g = digit_base[d] + (s - lpos[d]);
        hi2[g] = h;
        lo2[g] = l;
        val2[g] = s_val[s];
      }
      __syncthreads();
      digit_base[t] += tot;
      __syncthreads();
    }
  }
}
}  // namespace

// ---- single 64-bit words ordered by a bit range: rx_w64.hpp (the scatter pass is a template there: its last pass can hand every
// word and its final position to a functor) ----------------
// tile = RX_SUB * iter elements: small inputs get many small tiles (occupancy), big inputs bigger tiles
// (the digit x tile table stays a few MB)
uint32_t tbk_rx_iter_for(uint32_t n) {
  uint32_t it = (uint32_t)(((uint64_t)n + (uint64_t)RX_SUB * 4096 - 1) / ((uint64_t)RX_SUB * 4096));
  if (it < 1) it = 1;
  if (it > RX_MAX_ITER) it = RX_MAX_ITER;
  return it;
}
static uint32_t rx_iter_for(uint32_t n) { return tbk_rx_iter_for(n); }
int tbk_rx_hist_rowscan(tbk_ctx* ctx, const uint64_t* word, uint32_t shift, uint32_t n, uint32_t ntiles, uint32_t iter, uint32_t* table,
                        uint32_t* totals) {
  TBK_LAUNCH(ctx, "rx_hist", rx_hist_k, ntiles, RX_NT, 0, word, shift, n, ntiles, iter, table);
  TBK_LAUNCH(ctx, "rx_rowscan", rx_rowscan_k, 256, 256, 0, table, ntiles, totals);
  return 0;
}
size_t tbk_radix_ws_bytes(uint32_t n) {
  uint32_t ntiles = cdiv(n ? n : 1, RX_SUB * rx_iter_for(n));
  return (size_t)256 * ntiles * 4 + 256 * 4 + 4096;
}

int tbk_radix_sort128(tbk_ctx* ctx, SortBufs* b, uint32_t n, uint64_t only_hi, uint64_t only_lo, bool masks_are_exact) {
  if (n < 2) return 0;
  uint64_t vary_hi = only_hi, vary_lo = only_lo;
  if (!masks_are_exact) {  // find the bits that actually vary (one reduction kernel + one read-back)
    uint64_t* d_andor = ctx->d_scalars + 32;
    uint64_t init[4] = {~0ull, 0ull, ~0ull, 0ull};
    memcpy(ctx->h_scalars + 32, init, sizeof(init));
    TBK_HIP(hipMemcpyAsync(d_andor, ctx->h_scalars + 32, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    uint32_t g = cdiv(n, 256 * 16);
    if (g > 512) g = 512;
    TBK_LAUNCH(ctx, "rx_bits", rx_bits_k, g, 256, 0, b->hi, b->lo, n, d_andor);
    TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 32, d_andor, sizeof(init), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    vary_hi = (ctx->h_scalars[32] ^ ctx->h_scalars[33]) & only_hi;
    vary_lo = (ctx->h_scalars[34] ^ ctx->h_scalars[35]) & only_lo;
  }
  const uint32_t iter = rx_iter_for(n);
  uint32_t ntiles = cdiv(n, RX_SUB * iter);
  uint32_t* table = ws_alloc<uint32_t>(ctx, (size_t)256 * ntiles);
  uint32_t* totals = ws_alloc<uint32_t>(ctx, 256);
  if (!table || !totals) return TBK_ENOMEM;
  for (int word = 0; word < 2; ++word) {  // LSD: lo word first
    uint64_t vary = word == 0 ? vary_lo : vary_hi;
    for (uint32_t shift = 0; shift < 64; shift += 8) {
      if (((vary >> shift) & 0xFFull) == 0) continue;
      const uint64_t* src = word == 0 ? b->lo : b->hi;
      TBK_LAUNCH(ctx, "rx_hist", rx_hist_k, ntiles, RX_NT, 0, src, shift, n, ntiles, iter, table);
      TBK_LAUNCH(ctx, "rx_rowscan", rx_rowscan_k, 256, 256, 0, table, ntiles, totals);
      TBK_LAUNCH(ctx, "rx_scatter", rx_scatter_k, ntiles, RX_NT, 0, b->hi, b->lo, b->val, b->hi2, b->lo2, b->val2, word,
                 shift, n, ntiles, iter, table, totals);
      std::swap(b->hi, b->hi2);
      std::swap(b->lo, b->lo2);
      std::swap(b->val, b->val2);
    }
  }
  return tbk_check_launch(ctx, "radix_sort128");
}

// stable sort of 64-bit words by the bits of `mask`; mask_is_exact: the caller knows which bits can differ — otherwise one
// reduction over the words (and one read-back) finds the bits that do, and whole constant digits are skipped.  The result is in
// *w (swapped with *w2 as the passes go).
int tbk_radix_sort_w64(tbk_ctx* ctx, uint64_t** w, uint64_t** w2, uint32_t n, uint64_t mask, bool mask_is_exact) {
  return tbk_radix_sort_w64_emit(ctx, w, w2, n, mask, mask_is_exact, RxNoEmit{});
}
int tbk_rx_vary_bits(tbk_ctx* ctx, const uint64_t* w, uint32_t n, uint64_t mask, uint64_t* vary) {
  uint64_t* d_andor = ctx->d_scalars + 32;
  uint64_t init[4] = {~0ull, 0ull, ~0ull, 0ull};
  memcpy(ctx->h_scalars + 32, init, sizeof(init));
  TBK_HIP(hipMemcpyAsync(d_andor, ctx->h_scalars + 32, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  uint32_t g = cdiv(n, 256 * 16);
  if (g > 512) g = 512;
  TBK_LAUNCH(ctx, "rx_bits", rx_bits_k, g, 256, 0, w, w, n, d_andor);
  TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 32, d_andor, sizeof(init), hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  *vary = (ctx->h_scalars[32] ^ ctx->h_scalars[33]) & mask;
  return 0;
}
