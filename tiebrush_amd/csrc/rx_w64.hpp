// rx_w64.hpp — stable LSD radix sort of single 64-bit words by a bit range (e.g. id : value packed into one word), 8-bit digits,
// tiled: the histogram / row-scan kernels live in prims.hip, the scatter pass is a template here so that a caller can hang a
// functor on the LAST pass — emit(final position, word) runs where the word is written, while the tile's neighbourhood of the
// input is still cache-hot (collapse.hip: the YD items pick up their group's coordinates there instead of in a gather pass over
// the sorted items, whose reads would touch a line per item).
#pragma once
#include <type_traits>
#include <utility>

#include "dev_common.hpp"
#include "tbk_internal.h"

constexpr int RX_NT = 256;
constexpr int RX_E = 8;                    // elements per lane per sub-tile
constexpr int RX_SUB = RX_NT * RX_E;       // 2048
constexpr int RX_MAX_ITER = 16;            // sub-tiles per tile (runtime choice: tile = RX_SUB * iter)

uint32_t tbk_rx_iter_for(uint32_t n);
// per-tile digit counts of `word` at `shift` -> table, row scans -> table / totals (prims.hip)
int tbk_rx_hist_rowscan(tbk_ctx* ctx, const uint64_t* word, uint32_t shift, uint32_t n, uint32_t ntiles, uint32_t iter, uint32_t* table,
                        uint32_t* totals);
// the bits of `mask` that differ between some two words (one reduction + one read-back)
int tbk_rx_vary_bits(tbk_ctx* ctx, const uint64_t* w, uint32_t n, uint64_t mask, uint64_t* vary);

__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    bool bit = (d >> b) & 1u;
    uint64_t bal = __ballot(valid && bit);
    peers &= bit ? bal : ~bal;
  }
  return peers;
}

struct RxNoEmit {
  __device__ __forceinline__ void operator()(uint32_t, uint64_t) const {}
};

template <class Emit>
__global__ __launch_bounds__(RX_NT) void w64_scatter_k(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t shift, uint32_t n,
                                                       uint32_t ntiles, uint32_t iter, const uint32_t* __restrict__ table,
                                                       const uint32_t* __restrict__ totals, int do_emit, Emit emit) {
  __shared__ uint32_t digit_base[256];
  __shared__ uint32_t wave_cnt[4][256];
  __shared__ uint32_t lpos[256];
  __shared__ uint32_t sm[8];
  __shared__ uint64_t s_w[RX_SUB];
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
    uint64_t kk[RX_E];
    uint32_t kr[RX_E];
#pragma unroll
    for (int e = 0; e < RX_E; ++e) {
      uint64_t i = sub_base + (uint64_t)w * (64 * RX_E) + (uint64_t)e * 64 + lane_id();
      kk[e] = i < n ? in[i] : ~0ull;
    }
#pragma unroll
    for (int e = 0; e < RX_E; ++e) {
      uint64_t i = sub_base + (uint64_t)w * (64 * RX_E) + (uint64_t)e * 64 + lane_id();
      bool valid = i < n;
      uint32_t d = (uint32_t)((kk[e] >> shift) & 0xFFu);
      uint64_t peers = match_digit(d, valid);
      uint32_t before = (uint32_t)__popcll(peers & lanemask_lt());
      uint32_t base = valid ? wave_cnt[w][d] : 0u;
      __builtin_amdgcn_wave_barrier();
      if (valid && before == 0) wave_cnt[w][d] = base + (uint32_t)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
      kr[e] = (base + before) | (d << 16) | (valid ? 0u : 0x80000000u);
    }
    __syncthreads();
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
        s_w[wave_cnt[w][d] + (kr[e] & 0xFFFFu)] = kk[e];
      }
    }
    __syncthreads();
    uint32_t cnt_sub = (uint32_t)((n - sub_base) < (uint64_t)RX_SUB ? (n - sub_base) : (uint64_t)RX_SUB);
    for (uint32_t q = t; q < cnt_sub; q += RX_NT) {
      uint64_t kq = s_w[q];
      uint32_t d = (uint32_t)((kq >> shift) & 0xFFu);
      const uint32_t g = digit_base[d] + (q - lpos[d]);
      out[g] = kq;
      if (do_emit) emit(g, kq);
    }
    __syncthreads();
    digit_base[t] += tot;
    __syncthreads();
  }
}

// stable sort of 64-bit words by the bits of `mask`; mask_is_exact: the caller knows which bits can differ — otherwise one
// reduction over the words (and one read-back) finds the bits that do, and whole constant digits are skipped.  The result is in
// *w (swapped with *w2 as the passes go).  emit(final position, word) is called for every word in the last pass (when no bit of
// the mask varies there is no pass: a flat kernel calls it).
template <class Emit>
__global__ void w64_emit_flat_k(const uint64_t* __restrict__ in, uint32_t n, Emit emit) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) emit(i, in[i]);
}
template <class Emit>
int tbk_radix_sort_w64_emit(tbk_ctx* ctx, uint64_t** w, uint64_t** w2, uint32_t n, uint64_t mask, bool mask_is_exact, Emit emit,
                            const char* scatter_name = "rx_scatter") {
  constexpr bool has_emit = !std::is_same<Emit, RxNoEmit>::value;
  if (n == 0) return 0;
  uint64_t vary = mask;
  if (n < 2) vary = 0;
  if (n >= 2 && !mask_is_exact) TBK_TRY(tbk_rx_vary_bits(ctx, *w, n, mask, &vary));
  const uint32_t iter = tbk_rx_iter_for(n);
  uint32_t ntiles = cdiv(n, RX_SUB * iter);
  uint32_t* table = ws_alloc<uint32_t>(ctx, (size_t)256 * ntiles);
  uint32_t* totals = ws_alloc<uint32_t>(ctx, 256);
  if (!table || !totals) return TBK_ENOMEM;
  int last = -1;
  for (uint32_t shift = 0; shift < 64; shift += 8)
    if (((vary >> shift) & 0xFFull) != 0) last = (int)shift;
  for (uint32_t shift = 0; shift < 64; shift += 8) {
    if (((vary >> shift) & 0xFFull) == 0) continue;
    TBK_TRY(tbk_rx_hist_rowscan(ctx, *w, shift, n, ntiles, iter, table, totals));
    TBK_LAUNCH(ctx, scatter_name, (w64_scatter_k<Emit>), ntiles, RX_NT, 0, *w, *w2, shift, n, ntiles, iter, table, totals,
               (has_emit && (int)shift == last) ? 1 : 0, emit);
    std::swap(*w, *w2);
  }
  if (has_emit && last < 0) TBK_LAUNCH(ctx, scatter_name, (w64_emit_flat_k<Emit>), cdiv(n, 256), 256, 0, *w, n, emit);
  return tbk_check_launch(ctx, "radix_sort_w64");
}
