// crc32_block.hpp — CRC-32 (RFC 1952: reflected 0xEDB88320) of a byte run that lies in LDS, by a whole workgroup.
// A CRC register is a linear function of (register, data) over GF(2): the run is cut into 128-byte chunks counted from its END (so that
// only the first chunk is short), every thread runs the table-driven update over one chunk — the first with the initial 0xFFFFFFFF, the
// others from 0 —, and the registers are folded pairwise: advancing a register through 128 << k zero bytes is a 32 x 32 bit matrix (the
// k-th is the square of the one before), so chunk j joins chunk j + 2^k by one matrix-vector product, nine levels for 512 chunks.
// Used behind the inflate (bamdev.hip: the member's CRC32 is checked as htslib checks it) and in front of the deflate (bgzdef.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int CRCB_CHUNK = 128;
constexpr int CRCB_MAXCHUNKS = 512;           // 64 KiB
constexpr uint32_t CRCB_LDS_WORDS = 256 + 9 * 32 + CRCB_MAXCHUNKS;   // table, matrices, registers

__device__ __forceinline__ uint32_t crcb_matvec(const uint32_t* M, uint32_t v) {
  uint32_t r = 0;
#pragma unroll 1
  for (int b = 0; b < 32; ++b) r ^= ((v >> b) & 1u) ? M[b] : 0u;
  return r;
}

// the table and the matrices (once per workgroup: they do not depend on the data); work = CRCB_LDS_WORDS words of LDS
__device__ __forceinline__ void crcb_setup(uint32_t* work, uint32_t tid, uint32_t nt) {
  uint32_t* crct = work;
  uint32_t* crcm = work + 256;
  for (uint32_t i = tid; i < 256; i += nt) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    crct[i] = c;
  }
  __syncthreads();
  if (tid < 32) {
    uint32_t s = 1u << tid;
    for (int b = 0; b < CRCB_CHUNK; ++b) s = crct[s & 255u] ^ (s >> 8);
    crcm[tid] = s;
  }
  __syncthreads();
#pragma unroll 1
  for (int k = 1; k < 9; ++k) {
    if (tid < 32) crcm[k * 32 + tid] = crcb_matvec(crcm + (k - 1) * 32, crcm[(k - 1) * 32 + tid]);
    __syncthreads();
  }
}

// CRC-32 of bytes[0, n) (n <= 64 KiB, in LDS); every thread of the workgroup (nt >= 64, all of them call) gets the value.
__device__ __forceinline__ uint32_t crcb_run(uint32_t* work, const uint8_t* bytes, uint32_t n, uint32_t tid, uint32_t nt) {
  const uint32_t* crct = work;
  const uint32_t* crcm = work + 256;
  uint32_t* crcv = work + 256 + 9 * 32;
  const uint32_t nchunk = (n + CRCB_CHUNK - 1) / CRCB_CHUNK;
  for (uint32_t j = tid; j < CRCB_MAXCHUNKS; j += nt) {
    uint32_t s = 0;
    if (j < nchunk) {
      const uint32_t hi = n - (uint32_t)CRCB_CHUNK * j, lo = hi >= (uint32_t)CRCB_CHUNK ? hi - CRCB_CHUNK : 0u;
      s = (j == nchunk - 1) ? 0xFFFFFFFFu : 0u;
      for (uint32_t b = lo; b < hi; ++b) s = crct[(s ^ bytes[b]) & 255u] ^ (s >> 8);
    }
    crcv[j] = s;
  }
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < 9; ++k) {
    const uint32_t step = 1u << k;
    for (uint32_t j = tid; j < CRCB_MAXCHUNKS; j += nt)
      if ((j & (2 * step - 1)) == 0 && j + step < nchunk) crcv[j] ^= crcb_matvec(crcm + k * 32, crcv[j + step]);   // (chunks past the run hold 0)
    __syncthreads();
  }
  const uint32_t r = n ? ~crcv[0] : 0u;
  __syncthreads();
  return r;
}
