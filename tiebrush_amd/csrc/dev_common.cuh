// dev_common.cuh — device helpers: CIGAR walks, hashing, wave64 primitives.  gfx950 (wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

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
__device__ __forceinline__ uint64_t hash_step(uint64_t h, uint64_t v) { return mix64(h ^ (v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2))); }

// Reference GSamRecord::setupCoordinates (/root/reference/src/GSam.cpp:351-417), literal:
// calls on_exon(start1,end1) for every exon in order, returns l (reference length so that
// end = pos + l) and the exon count through *nex.  Unmapped records are the caller's business.
// (C: anything indexable that yields the CIGAR words — a pointer, or a view with the first words in registers)
template <class C, class F>
__device__ __forceinline__ int walk_exons(int32_t pos, C cig, uint32_t n, F on_exon, int* nex) {
  int l = 0, cnt = 0;
  int exstart = pos;
  bool intron = false, ins = false;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t c = cig[i];
    uint32_t op = cig_op(c);
    switch (op) {
      case C_EQ:
      case C_X:
      case C_M:
      case C_D:
        l += (int)cig_len(c);
        intron = false;
        ins = false;
        break;
      case C_N:
        if (!ins || !intron) {
          on_exon(exstart + 1, pos + l);
          cnt++;
        }
        l += (int)cig_len(c);
        exstart = pos + l;
        intron = true;
        break;
      case C_S:
      case C_H:
        intron = false;
        ins = false;
        break;
      case C_I:
        ins = true;
        break;
      default:  // P and unknown: nothing
        break;
    }
  }
  on_exon(exstart + 1, pos + l);
  cnt++;
  *nex = cnt;
  return l;
}

// reference length only (end = pos + l)
template <class C>
__device__ __forceinline__ int cigar_reflen(C cig, uint32_t n) {
  int l = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t c = cig[i];
    uint32_t op = cig_op(c);
    // M,=,X,D,N consume the reference: ops 0,2,3,7,8
    if ((0x18Du >> op) & 1u) l += (int)cig_len(c);
  }
  return l;
}

// CIGAR words of one record with the first three already in registers (loaded together with the record's other fields, ahead
// of their use); longer CIGARs read on from memory
struct CigView {
  uint32_t w0, w1, w2;
  const uint32_t* p;
  __device__ __forceinline__ uint32_t operator[](uint32_t k) const { return k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : p[k])); }
};

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
