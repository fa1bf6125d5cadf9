// dev_common.hpp — device helpers: CIGAR walks, hashing, wave64 primitives.  gfx950 (wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

// The kernels are written for the gfx9 family's ISA (CDNA: gfx90a / gfx942 / gfx950; the Makefile builds gfx950): 64-wide waves, DPP row
// operations, and s_waitcnt's gfx9 field layout — TBK_WAIT_VMCNT0 is "vmcnt(0), nothing else" THERE, where stores count in vmcnt too.
// On gfx10 and later stores are counted by vscnt and the immediate's fields lie elsewhere: the hand-ordered hand-overs (cov.hip:
// cb_last_block, wgroup.hip: PIPE) would compile and silently lose their ordering.  So another target does not compile at all.
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx90a__) || defined(__gfx940__) || defined(__gfx941__) || defined(__gfx942__) || defined(__gfx950__))
#error "tiebrush_amd's kernels are written for the gfx9 / CDNA ISA (gfx950; wave64, gfx9 s_waitcnt encoding): build with ARCH=gfx950"
#endif
#define TBK_WAIT_VMCNT0 0x0F70

enum : uint32_t { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5, C_P = 6, C_EQ = 7, C_X = 8, C_B = 9 };
__device__ __forceinline__ uint32_t cig_op(uint32_t c) { return c & 0xFu; }
__device__ __forceinline__ uint32_t cig_len(uint32_t c) { return c >> 4; }

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// 64-bit mixing (splitmix64 finaliser) — used for the strategy-key hash
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
// One step of the strategy-key hash: the state as two 32-bit halves, shifts, adds and xors only.  (It was a splitmix round: two 64-bit
// multiplies, each three quarter-rate instructions.  The raw window kernel runs this code in nearly every wave — one record in thirty
// needs a hash, a wave holds 64 — and is bound by its vector instructions: wg_hash_k 6.74 -> see DESIGN.md §3.)
// For a fixed word the step is a bijection of the state, and for a fixed state it is one-to-one in the word: two keys of equal
// length that differ in a single word never collide.  Everything else collides at the rate of a 31-bit hash, and every user of the
// hash verifies the keys themselves behind it (wg_finish_raw_k, col_heads_k; a mismatch reseeds).
__device__ __forceinline__ uint64_t hash_step(uint64_t h, uint64_t v) {
  uint32_t a = (uint32_t)h, b = (uint32_t)(h >> 32);
  a += (uint32_t)v;
  a += a << 10;
  a ^= a >> 6;
  b += (uint32_t)(v >> 32);
  b ^= a;
  b += b << 3;
  b ^= b >> 11;
  b += b << 15;
  a ^= b >> 7;
  return ((uint64_t)b << 32) | a;
}

// CIGAR words of one record with the first three already in registers (loaded together with the record's other fields, ahead
// of their use); longer CIGARs read on from memory
struct CigView {
  uint32_t w0, w1, w2;
  const uint32_t* p;
  __device__ __forceinline__ uint32_t operator[](uint32_t k) const { return k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : p[k])); }
};
// f(k, word) for every CIGAR word in order.  For a CigView the first three steps are straight-line code on registers.
template <class F>
__device__ __forceinline__ void cig_for_each(const uint32_t* c, uint32_t n, F f) {
  for (uint32_t k = 0; k < n; ++k) f(k, c[k]);
}
template <class F>
__device__ __forceinline__ void cig_for_each(const CigView& c, uint32_t n, F f) {
  if (n > 0) f(0u, c.w0);
  if (n > 1) f(1u, c.w1);
  if (n > 2) f(2u, c.w2);
  for (uint32_t k = 3; k < n; ++k) f(k, c.p[k]);
}

// Reference GSamRecord::setupCoordinates (/root/reference/src/GSam.cpp:351-417), literal:
// calls on_exon(start1,end1) for every exon in order, returns l (reference length so that
// end = pos + l) and the exon count through *nex.  Unmapped records are the caller's business.
// (C: anything indexable that yields the CIGAR words — a pointer, or a view with the first words in registers)
template <class C, class F>
__device__ __forceinline__ int walk_exons(int32_t pos, C cig, uint32_t n, F on_exon, int* nex) {
  int l = 0, cnt = 0;
  int exstart = pos;
  bool intron = false, ins = false;
  // (the switch of the reference over the operation, as masks: M = X D add to the length and end an intron / insertion state,
  // N closes an exon — unless an insertion directly follows an intron — and opens the next, S H only reset the state, I sets it)
  cig_for_each(cig, n, [&](uint32_t, uint32_t c) {
    const uint32_t op = cig_op(c);
    const bool is_m = (0x185u >> op) & 1u, is_n = op == C_N, is_sh = (0x30u >> op) & 1u, is_i = op == C_I;
    if (is_n && (!ins || !intron)) {
      on_exon(exstart + 1, pos + l);
      cnt++;
    }
    l += (is_m || is_n) ? (int)cig_len(c) : 0;
    exstart = is_n ? pos + l : exstart;
    intron = is_n ? true : ((is_m || is_sh) ? false : intron);
    ins = is_i ? true : ((is_m || is_sh) ? false : ins);
  });
  on_exon(exstart + 1, pos + l);
  cnt++;
  *nex = cnt;
  return l;
}

// reference length only (end = pos + l)
template <class C>
__device__ __forceinline__ int cigar_reflen(C cig, uint32_t n) {
  int l = 0;
  // M,=,X,D,N consume the reference: ops 0,2,3,7,8
  cig_for_each(cig, n, [&](uint32_t, uint32_t c) { l += ((0x18Du >> cig_op(c)) & 1u) ? (int)cig_len(c) : 0; });
  return l;
}

// ---- wave64 scans ---------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T wave_incl_sum(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if ((int)lane_id() >= d) v += o;
  }
  return v;
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// block-wide exclusive sum for 256-thread blocks; returns exclusive prefix, *total = block total.
// `sm` must hold >= 8 T's.
template <class T, int NT>
__device__ __forceinline__ T block_excl_sum(T v, T* sm, T* total) {
  constexpr int NW = NT / 64;
  T inc = wave_incl_sum(v);
  uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 63) sm[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    T x = sm[i];
    if ((uint32_t)i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
