// bgzdef.hip — the OUTPUT side of the BGZF path on gfx950: flushPData's tagging of the representatives
// (/root/reference/src/tiebrush.cpp:506-525; bam_aux_update_* rules GSam.h:300-305) and GSamWriter::write -> sam_write1 -> bgzf_write
// (GSam.h:648-653; zlib's deflate inside htslib) as kernels — the mirror of bamdev.hip's inflate.
//
//   tbk_bam_encode     representatives (tile indices into the tile tbk_bam_decode left on the device, and / or raw records the host
//                      hands over) + YC / YX / YD  ->  a run of whole BGZF members in host memory, ready to be appended to the output
//     enc_plan_k         a thread per record: aux walk, does the record carry YC / YX / YD already, its tagged length
//     enc_emit_k         16 lanes per record: the record copied into the payload stream with block_size and the three tags appended
//                        (a record that carries one of the tags already: one lane edits it field by field, htslib's rules)
//     enc_cuts_k         members cut at record boundaries (every member begins with a record, like htslib's bgzf_flush_try)
//   tbk_bgzf_deflate   a byte run -> BGZF members (the kernel alone; what the tests drive with arbitrary payloads)
//     bgz_deflate_k      ONE workgroup per member, persistent over the member list: the payload staged in LDS; eight waves find
//                        matches (two hash tables in LDS — 4-byte and 8-byte keys — each remembering the latest position: a
//                        returning ds_max both inserts a position and hands back the latest earlier one, also among the 64 lanes of
//                        one instruction), a ninth wave walks the lazy parse (zlib's rule) one round behind them with the match lengths
//                        of 64 positions in registers, and writes tokens; then histograms, CRC-32 (a chunk per thread, folded with
//                        zero-advance matrices), minimum-redundancy code lengths (deflate_codes.h, one lane), the dynamic block header,
//                        and the token bits OR-ed into LDS at their prefix-summed bit offsets; one coalesced copy to the member's slot
//     bgz_gather_k       the members' slots -> one contiguous run (sizes prefix-summed on the device)
//
// Byte work: no MFMA.  The deflate kernel is bound by LDS round trips of dependent byte compares, not by HBM (it reads a payload byte
// once and writes a quarter of one).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "crc32_block.hpp"
#include "deflate_codes.h"
#include "dev_common.hpp"
#include "tbk_internal.h"

namespace {

constexpr int DF_MW = 8;                      // matching waves
constexpr int DF_PW = DF_MW;                  // the parsing wave
constexpr int DF_IW = DF_MW + 1;              // the inserting wave
constexpr int DF_NT = (DF_MW + 2) * 64;
constexpr int DF_ROUND = DF_MW * 64;          // positions per round
constexpr uint32_t DF_MAXPAY = 0xff00;        // payload bytes per member (htslib's BGZF_BLOCK_SIZE)
constexpr uint32_t DF_SLOT = 65536;           // bytes of a member's slot (a BGZF member is at most 64 KiB)
constexpr int DF_H4 = 10, DF_H8 = 10;         // bits of the two hash tables
constexpr uint32_t DF_GOOD = 16;              // zlib level 6's max_lazy: a match this long is taken without a look at the next position;
                                              // also how far the matching waves compare (the parse finds a longer match's end itself)
// LDS map (bytes)
constexpr uint32_t L_IN = 0;                              // payload (zero padded), later the member as it will be written
constexpr uint32_t L_IN_BYTES = 65328;
constexpr uint32_t L_T4 = L_IN + L_IN_BYTES;              // u32[1 << DF_H4]   position + 1 of the latest occurrence, 0 = none
constexpr uint32_t L_T8 = L_T4 + (4u << DF_H4);           // u32[1 << DF_H8]
constexpr uint32_t L_CB = L_T8 + (4u << DF_H8);           // two buffers of {u16 candidate of the 4-byte table [512], u16 of the 8-byte table [512]}
constexpr uint32_t L_CB_ONE = DF_ROUND * 4;
constexpr uint32_t L_MB = L_CB + 2 * L_CB_ONE;            // two buffers of {u16 dist - 1 [512], u8 compared length (0, 4 .. 16) [512]}
constexpr uint32_t L_MB_ONE = DF_ROUND * 3;
constexpr uint32_t L_CTL = L_MB + 2 * L_MB_ONE;           // u32[16] control words
constexpr uint32_t L_END = L_CTL + 64;
// after the parse the tables and match buffers are dead: the coder's arrays overlay them
constexpr uint32_t L_HL = L_T4;                           // u32[288] literal / length frequencies
constexpr uint32_t L_HD = L_HL + 288 * 4;                 // u32[32]  distance frequencies
constexpr uint32_t L_HC = L_HD + 32 * 4;                  // u32[32]  code-length-code frequencies
constexpr uint32_t L_KEY = L_HC + 32 * 4;                 // u32[288] sort keys (frequency << 9 | symbol), sorted
constexpr uint32_t L_A = L_KEY + 288 * 4;                 // u32[288] dfl_code_lengths' array
constexpr uint32_t L_LENL = L_A + 288 * 4;                // u8[288] + u8[32] + u8[32]: code lengths of the three codes
constexpr uint32_t L_LEND = L_LENL + 288;
constexpr uint32_t L_LENC = L_LEND + 32;
constexpr uint32_t L_CODL = L_LENC + 32;                  // u16[288], u16[32], u16[32]
constexpr uint32_t L_CODD = L_CODL + 288 * 2;
constexpr uint32_t L_CODC = L_CODD + 32 * 2;
constexpr uint32_t L_RLE = L_CODC + 32 * 2;               // u16[320]
constexpr uint32_t L_BLC = L_RLE + 320 * 2;               // u32[20]
constexpr uint32_t L_CRCT = L_BLC + 20 * 4;               // u32[CRCB_LDS_WORDS] CRC-32 work area
constexpr uint32_t L_WSUM = L_CRCT + 4 * CRCB_LDS_WORDS;  // (crc32_block.hpp: table, zero-advance matrices, per-chunk registers); u32[2][16] per-wave bit totals of the token scan
constexpr uint32_t L_OVER_END = L_WSUM + 2 * 16 * 4;
static_assert(L_OVER_END <= L_CTL, "the coder's arrays must fit where the tables were");
static_assert(L_END <= 81920, "two workgroups per CU");

enum { C_NTOK = 0, C_NEXT = 1, C_M = 2, C_XBITS = 3, C_MODE = 4, C_BITS = 5, C_HLIT = 6, C_HDIST = 7, C_HCLEN = 8, C_NRLE = 9, C_CRC = 10 };

__device__ __forceinline__ uint32_t ld32u(const uint32_t* w, uint32_t p) {  // the four bytes at byte offset p of an LDS array
  const uint32_t i = p >> 2;
  return __builtin_amdgcn_alignbyte(w[i + 1], w[i], p & 3u);
}

__device__ __forceinline__ uint32_t match_len(const uint32_t* w, uint32_t p, uint32_t q, uint32_t lim) {
  uint32_t l = 0;
  while (l < lim) {
    const uint32_t x = ld32u(w, p + l) ^ ld32u(w, q + l);
    if (x) {
      l += (uint32_t)__builtin_ctz(x) >> 3;
      break;
    }
    l += 4;
  }
  return l < lim ? l : lim;
}

struct DfMember {
  uint64_t src;   // offset of the member's payload in the source run
  uint32_t n;     // payload bytes (<= DF_MAXPAY; 0: nothing is written)
  uint32_t pad;
};

// bits into the member image in LDS (zeroed before); several threads may touch one word
__device__ __forceinline__ void put_bits(uint32_t* out, uint32_t bitpos, uint64_t v, uint32_t nbits) {
  if (nbits == 0) return;
  const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
  const uint64_t lo = v << sh;  // (v < 2^48: bits 0 .. 79 over three words)
  atomicOr(&out[w], (uint32_t)lo);
  if (sh + nbits > 32) atomicOr(&out[w + 1], (uint32_t)(lo >> 32));
  if (sh + nbits > 64) atomicOr(&out[w + 2], (uint32_t)(v >> (64 - sh)));
}

extern __shared__ __align__(16) uint8_t df_lds[];
#define DF_U32(off) ((uint32_t*)(df_lds + (off)))
#define DF_U16(off) ((uint16_t*)(df_lds + (off)))
#define DF_U8(off) (df_lds + (off))

// Code lengths and canonical codes of one of the three codes, by the whole workgroup: the used symbols are rank-sorted by (frequency,
// symbol) a thread each, one lane runs deflate_codes.h on the sorted frequencies.  zlib's rule: at least two symbols get a length, so
// that every code is complete.  (Out of line: the kernel's phases do not share registers across each other that way.)
__device__ __noinline__ void df_build_code(uint32_t freq_off, int nsym, int maxbits, uint32_t len_off, uint32_t code_off) {
  uint32_t* const freq = DF_U32(freq_off);
  uint8_t* const len = DF_U8(len_off);
  uint16_t* const code = DF_U16(code_off);
  uint32_t* const ctl = DF_U32(L_CTL);
  uint32_t* const key = DF_U32(L_KEY);
  uint32_t* const arr = DF_U32(L_A);
  uint32_t* const blc = DF_U32(L_BLC);
  const uint32_t tid = threadIdx.x;
  if (tid == 0) {
    int nz = 0;
    for (int s = 0; s < nsym; ++s) nz += freq[s] != 0;
    for (int s = 0; nz < 2 && s < nsym; ++s)
      if (freq[s] == 0) freq[s] = 1, ++nz;
    ctl[C_MODE] = 0;
  }
  __syncthreads();
  if ((int)tid < nsym) {
    len[tid] = 0;
    const uint32_t f = freq[tid];
    if (f) {
      const uint32_t k = f << 9 | tid;
      uint32_t rank = 0;
      for (int s = 0; s < nsym; ++s) {
        const uint32_t fs = freq[s];
        rank += (fs != 0 && (fs << 9 | (uint32_t)s) < k) ? 1u : 0u;
      }
      key[rank] = k;
      atomicAdd(&ctl[C_MODE], 1u);
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int mused = (int)ctl[C_MODE];
    for (int i = 0; i < mused; ++i) arr[i] = key[i] >> 9;
    dfl_code_lengths(arr, mused, maxbits, blc);
    for (int i = 0; i < mused; ++i) len[key[i] & 511u] = (uint8_t)arr[i];
    dfl_canonical_codes(len, nsym, maxbits, code, blc);
  }
  __syncthreads();
}

// CRC-32 of the n payload bytes in LDS -> ctl[C_CRC] (crc32_block.hpp: a chunk per thread, folded with zero-advance matrices)
__device__ __noinline__ void df_crc32(uint32_t n) {
  uint32_t* const work = DF_U32(L_CRCT);
  crcb_setup(work, threadIdx.x, DF_NT);
  const uint32_t c = crcb_run(work, DF_U8(L_IN), n, threadIdx.x, DF_NT);
  if (threadIdx.x == 0) DF_U32(L_CTL)[C_CRC] = c;
  __syncthreads();
}

// the dynamic block header (one lane) and the tokens' bits (everyone) into the member image in LDS; returns the first free bit
__device__ __noinline__ uint32_t df_write_block(const uint32_t* __restrict__ tok, uint32_t ntok) {
  uint32_t* const inw = DF_U32(L_IN);
  uint32_t* const ctl = DF_U32(L_CTL);
  const uint8_t* const lenL = DF_U8(L_LENL);
  const uint8_t* const lenD = DF_U8(L_LEND);
  const uint8_t* const lenC = DF_U8(L_LENC);
  const uint16_t* const codL = DF_U16(L_CODL);
  const uint16_t* const codD = DF_U16(L_CODD);
  const uint16_t* const codC = DF_U16(L_CODC);
  const uint16_t* const rle = DF_U16(L_RLE);
  uint32_t* const wsum = DF_U32(L_WSUM);
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  for (uint32_t i = tid; i < (L_IN_BYTES >> 2); i += DF_NT) inw[i] = 0;
  __syncthreads();
  const uint32_t hbits = ctl[C_BITS];
  if (tid == 0) {
    uint32_t bp = 18 * 8;
    auto put = [&](uint32_t v, uint32_t nb) {
      put_bits(inw, bp, v, nb);
      bp += nb;
    };
    const uint32_t hlit = ctl[C_HLIT], hdist = ctl[C_HDIST], hclen = ctl[C_HCLEN], nr = ctl[C_NRLE];
    put(1, 1);
    put(2, 2);
    put(hlit - 257, 5);
    put(hdist - 1, 5);
    put(hclen - 4, 4);
    for (uint32_t i = 0; i < hclen; ++i) put(lenC[dfl_cl_order((int)i)], 3);
    for (uint32_t i = 0; i < nr; ++i) {
      const uint32_t sy = rle[i] & 0xFFu;
      put(codC[sy], lenC[sy]);
      if (sy >= 16) put((uint32_t)rle[i] >> 8, sy == 16 ? 2u : (sy == 17 ? 3u : 7u));
    }
  }
  uint32_t base = 18 * 8 + hbits;
#pragma unroll 1
  for (uint32_t i0 = 0; i0 < ntok; i0 += DF_NT) {
    const uint32_t i = i0 + tid;
    uint64_t v = 0;
    uint32_t nb = 0;
    if (i < ntok) {
      const uint32_t t = tok[i];
      if (t & 0x80000000u) {
        uint32_t c, e, x;
        dfl_len_code((t >> 16) & 0x1FFu, &c, &e, &x);
        v = codL[257 + c];
        nb = lenL[257 + c];
        v |= (uint64_t)x << nb;
        nb += e;
        dfl_dist_code((t & 0xFFFFu) + 1u, &c, &e, &x);
        v |= (uint64_t)codD[c] << nb;
        nb += lenD[c];
        v |= (uint64_t)x << nb;
        nb += e;
      } else {
        v = codL[t];
        nb = lenL[t];
      }
    }
    uint32_t inc = nb;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d);
      if ((int)lane >= d) inc += o;
    }
    uint32_t* const ws2 = wsum + ((i0 / DF_NT) & 1u) * 16;
    if (lane == 63) ws2[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < DF_NT / 64; ++w) {
      const uint32_t x = ws2[w];
      before += (w < (int)wave) ? x : 0u;
      all += x;
    }
    put_bits(inw, base + before + inc - nb, v, nb);
    base += all;
  }
  if (tid == 0) put_bits(inw, base, codL[256], lenL[256]);
  return base + lenL[256];
}

#ifdef DF_PROF
__device__ unsigned long long df_prof[16];
#define DF_STAMP(k)                                                      \
  do {                                                                   \
    if (threadIdx.x == 0) {                                              \
      const unsigned long long now_ = clock64();                         \
      atomicAdd(&df_prof[k], now_ - t_prev_);                            \
      t_prev_ = now_;                                                    \
    }                                                                    \
  } while (0)
#else
#define DF_STAMP(k)
#endif

__global__ __launch_bounds__(DF_NT) void bgz_deflate_k(uint32_t nmem, const DfMember* __restrict__ mem, const uint8_t* __restrict__ src,
                                                        uint8_t* __restrict__ slots, uint32_t* __restrict__ msize, uint32_t* __restrict__ tokens,
                                                        uint32_t* __restrict__ counter, uint32_t* __restrict__ err) {
  uint32_t* const inw = DF_U32(L_IN);
  uint8_t* const inb = DF_U8(L_IN);
  uint32_t* const t4 = DF_U32(L_T4);
  uint32_t* const t8 = DF_U32(L_T8);
  uint32_t* const ctl = DF_U32(L_CTL);
  uint32_t* const hl = DF_U32(L_HL);
  uint32_t* const hd = DF_U32(L_HD);
  uint32_t* const hc = DF_U32(L_HC);
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  uint32_t* const tok = tokens + (size_t)blockIdx.x * DF_MAXPAY;

  for (;;) {
    __syncthreads();  // (the previous member's image has left LDS; ctl is free)
    if (tid == 0) ctl[C_M] = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t m = ctl[C_M];
    if (m >= nmem) return;  // (every wave of the block reads the same word: the block leaves together)
#ifdef DF_PROF
    unsigned long long t_prev_ = clock64();
#endif
    const DfMember M = mem[m];
    const uint32_t n = M.n;
    if (n == 0 || n > DF_MAXPAY) {
      if (tid == 0) {
        msize[m] = 0;
        if (n > DF_MAXPAY) atomicOr(err, 1u);
      }
      continue;
    }
    // ---- stage the payload, clear the tables ----
    {
      const uint8_t* s = src + M.src;
      const uint32_t nw = n >> 2;
      for (uint32_t i = tid; i < nw; i += DF_NT) {
        uint32_t v;
        __builtin_memcpy(&v, s + 4 * (size_t)i, 4);  // (the run starts at any byte: unaligned dword loads)
        inw[i] = v;
      }
      if (tid == 0) {
        uint32_t v = 0;
        for (uint32_t b = nw * 4; b < n; ++b) v |= (uint32_t)s[b] << (8 * (b & 3u));
        inw[nw] = v;
      }
      for (uint32_t i = nw + 1 + tid; i < (L_IN_BYTES >> 2); i += DF_NT) inw[i] = 0;
      for (uint32_t i = tid; i < (1u << DF_H4) + (1u << DF_H8); i += DF_NT) t4[i] = 0;  // (t8 follows t4)
    }
    __syncthreads();
    DF_STAMP(0);
    // ---- the parse, three stages a round apart: the inserting wave hands every position of round r + 1 its candidates — the latest
    // earlier position with the same 4 bytes, with the same 8 bytes: one wave, its LDS atomics in program order, so a position meets
    // exactly the positions before it, the same ones in every run —; the eight matching waves compare round r's candidates over
    // the first 16 bytes; the parsing wave walks round r - 1's tokens (zlib's lazy rule) and measures a match that is taken in full ----
    const uint32_t rounds = (n + DF_ROUND - 1) / DF_ROUND;
    uint32_t ntok = 0, carry = 0;  // (live in the parsing wave only, uniform)
    if (wave >= DF_MW) __builtin_amdgcn_s_setprio(3);  // the two serial waves pace the round: they issue ahead of the matching waves
#ifdef DF_PROF
    unsigned long long busy_ = 0;
#endif
#pragma unroll 1
    for (int r = -1; r <= (int)rounds; ++r) {
#ifdef DF_PROF
      const unsigned long long r0_ = clock64();
#endif
      if (wave == DF_IW) {
        if (r + 1 < (int)rounds) {
          uint16_t* const cb = (uint16_t*)(df_lds + L_CB + ((uint32_t)(r + 1) & 1u) * L_CB_ONE);
#pragma unroll 4
          for (uint32_t sb = 0; sb < DF_MW; ++sb) {
            const uint32_t j = sb * 64 + lane, p = (uint32_t)(r + 1) * DF_ROUND + j;
            uint32_t k4 = 0xFFFFu, k8 = 0xFFFFu;
            if (p + 4 <= n) {
              const uint32_t i = p >> 2, sh = p & 3u;
              const uint32_t a = inw[i], b = inw[i + 1], c = inw[i + 2];
              const uint32_t w0 = __builtin_amdgcn_alignbyte(b, a, sh), w1 = __builtin_amdgcn_alignbyte(c, b, sh);
              const uint32_t c4 = atomicMax(&t4[(w0 * 0x9E3779B1u) >> (32 - DF_H4)], p + 1);
              if (c4 != 0 && c4 <= p && p + 1 - c4 <= 32768u) k4 = c4 - 1;
              if (p + 8 <= n) {
                const uint32_t c8 = atomicMax(&t8[(w0 * 0x9E3779B1u + w1 * 0x85EBCA77u) >> (32 - DF_H8)], p + 1);
                if (c8 != 0 && c8 <= p && c8 != c4 && p + 1 - c8 <= 32768u) k8 = c8 - 1;
              }
            }
            cb[j] = (uint16_t)k4;
            cb[DF_ROUND + j] = (uint16_t)k8;
          }
        }
      } else if (wave < DF_MW) {
        if (r >= 0 && r < (int)rounds) {
          const uint16_t* const cb = (const uint16_t*)(df_lds + L_CB + ((uint32_t)r & 1u) * L_CB_ONE);
          uint8_t* const mb = df_lds + L_MB + ((uint32_t)r & 1u) * L_MB_ONE;
          uint16_t* const mdist = (uint16_t*)mb;
          uint8_t* const mlen = mb + DF_ROUND * 2;
          const uint32_t p = (uint32_t)r * DF_ROUND + tid;
          uint32_t best = 0, bestd = 0;
          const uint32_t k4 = cb[tid], k8 = cb[DF_ROUND + tid];
          if ((k4 & k8) != 0xFFFFu) {
            const uint32_t lim = min(DF_GOOD, n - p);
            const uint32_t i = p >> 2, sh = p & 3u;
            const uint32_t a0 = inw[i], a1 = inw[i + 1], a2 = inw[i + 2], a3 = inw[i + 3], a4 = inw[i + 4];
            const uint32_t w0 = __builtin_amdgcn_alignbyte(a1, a0, sh), w1 = __builtin_amdgcn_alignbyte(a2, a1, sh);
            const uint32_t w2 = __builtin_amdgcn_alignbyte(a3, a2, sh), w3 = __builtin_amdgcn_alignbyte(a4, a3, sh);
            auto compared = [&](uint32_t q) -> uint32_t {  // bytes that agree among the first 16 (0 when the first four do not)
              const uint32_t x0 = ld32u(inw, q) ^ w0;
              if (x0) return 0u;
              const uint32_t x1 = ld32u(inw, q + 4) ^ w1, x2 = ld32u(inw, q + 8) ^ w2, x3 = ld32u(inw, q + 12) ^ w3;
              uint32_t l = x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : (x2 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : (x3 ? 12u + ((uint32_t)__builtin_ctz(x3) >> 3) : 16u));
              return l < lim ? l : lim;
            };
            if (k4 != 0xFFFFu) {
              best = compared(k4);
              bestd = p - k4;
            }
            if (k8 != 0xFFFFu) {
              const uint32_t l = compared(k8);
              if (l > best) best = l, bestd = p - k8;
            }
            if (best < 4) best = 0, bestd = 0;
          }
          mlen[tid] = (uint8_t)best;
          mdist[tid] = (uint16_t)(bestd ? bestd - 1 : 0);
        }
      } else if (r >= 1) {
        const uint32_t rr = (uint32_t)r - 1;
        const uint8_t* const mb = df_lds + L_MB + (rr & 1u) * L_MB_ONE;
        const uint16_t* const mdist = (const uint16_t*)mb;
        const uint8_t* const mlen = mb + DF_ROUND * 2;
        const uint32_t rbase = rr * DF_ROUND;
        const uint32_t rcnt = min((uint32_t)DF_ROUND, n - rbase);
#pragma unroll 1
        for (uint32_t sb = 0; sb * 64 < rcnt; ++sb) {
          const uint32_t cnt = min(64u, rcnt - sb * 64);
          if (carry >= cnt) {
            carry -= cnt;
            continue;
          }
          const uint32_t j = sb * 64 + lane;
          const uint32_t Lv = j < rcnt ? (uint32_t)mlen[j] : 0u;
          const uint32_t Nv = j + 1 < rcnt ? (uint32_t)mlen[j + 1] : 0u;  // (no look beyond the round)
          const uint32_t Dv = (uint32_t)mdist[j < rcnt ? j : 0];
          const uint32_t Bv = (uint32_t)inb[rbase + j];  // (the byte, should the position go out as a literal)
          // what the walk does at a position that has a match, decided by every lane for itself: 1 = zlib's lazy rule (a longer match
          // begins at the next byte: this byte goes out as a literal), otherwise the match is taken with this length (DF_GOOD: at
          // least that — the walk measures it)
          const uint32_t Sv = (Lv < DF_GOOD && Nv > Lv) ? 1u : Lv;
          uint32_t Fv = Lv;  // the length the token is written with
          const uint64_t mm = __ballot(Lv >= 4);
          uint64_t matm = 0, covm = 0;  // match starts; positions inside a taken match (its start included)
          uint32_t i = __builtin_amdgcn_readfirstlane(carry);
          const uint32_t ucnt = __builtin_amdgcn_readfirstlane(cnt);
          if (i) covm = (1ull << i) - 1ull;  // (the tail of the previous batch's last match; i < ucnt <= 64)
          // the serial part: from match to match
          while (i < ucnt) {
            const uint64_t rest = mm >> i;
            if (rest == 0) break;
            i += (uint32_t)__builtin_ctzll(rest);
            uint32_t full = __builtin_amdgcn_readlane(Sv, i);
            if (full == 1u) {
              i += 1;
              continue;
            }
            if (full >= DF_GOOD) {
              // 16 bytes agree and maybe more: the 64 lanes compare the next 256 bytes at once
              const uint32_t p = rbase + sb * 64 + i;
              const uint32_t q = p - (__builtin_amdgcn_readlane(Dv, i) + 1u);
              const uint32_t x = ld32u(inw, p + DF_GOOD + 4 * lane) ^ ld32u(inw, q + DF_GOOD + 4 * lane);
              const uint64_t ne = __ballot(x != 0);
              if (ne) {
                const uint32_t f = (uint32_t)__builtin_ctzll(ne);
                const uint32_t xb = __builtin_amdgcn_readlane(x, f);
                full = DF_GOOD + 4 * f + ((uint32_t)__builtin_ctz(xb) >> 3);
              } else {
                full = DF_GOOD + 256;
              }
              full = min(full, min(258u, n - p));
            }
            matm |= 1ull << i;
            covm |= (full >= 64u - i ? ~0ull : ((1ull << full) - 1ull)) << i;
            if (lane == i) Fv = full;
            i += full;
          }
          carry = i > ucnt ? i - ucnt : 0u;
          const uint64_t valid = ucnt == 64 ? ~0ull : ((1ull << ucnt) - 1ull);
          const uint64_t sel = valid & (matm | ~covm);  // token starts: taken matches and every position outside them
          if ((sel >> lane) & 1ull) {
            const uint32_t rank = __popcll(sel & ((1ull << lane) - 1ull));
            const uint32_t t = ((matm >> lane) & 1ull) ? (0x80000000u | (Fv << 16) | Dv) : Bv;
            tok[ntok + rank] = t;
          }
          ntok += (uint32_t)__popcll(sel);
        }
      }
#ifdef DF_PROF
      busy_ += clock64() - r0_;
#endif
      __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef DF_PROF
    if (lane == 0 && (wave == 0 || wave == DF_PW || wave == DF_IW)) atomicAdd(&df_prof[wave == 0 ? 8 : (wave == DF_PW ? 9 : 10)], busy_);
#endif
    DF_STAMP(1);
    if (wave == DF_PW && lane == 0) ctl[C_NTOK] = ntok;
    // ---- frequencies (the tables are dead: their memory holds the coder's arrays now) ----
    for (uint32_t i = tid; i < 288 + 32 + 32; i += DF_NT) hl[i] = 0;  // (hd, hc follow hl)
    if (tid == 0) ctl[C_XBITS] = 0;
    __syncthreads();
    ntok = ctl[C_NTOK];
    {
      uint32_t xb = 0;
      for (uint32_t i = tid; i < ntok; i += DF_NT) {
        const uint32_t t = tok[i];
        if (t & 0x80000000u) {
          uint32_t c, e, v;
          dfl_len_code((t >> 16) & 0x1FFu, &c, &e, &v);
          atomicAdd(&hl[257 + c], 1u);
          xb += e;
          dfl_dist_code((t & 0xFFFFu) + 1u, &c, &e, &v);
          atomicAdd(&hd[c], 1u);
          xb += e;
        } else {
          atomicAdd(&hl[t], 1u);
        }
      }
      if (xb) atomicAdd(&ctl[C_XBITS], xb);
      if (tid == 0) atomicAdd(&hl[256], 1u);
    }
    __syncthreads();
    DF_STAMP(2);
    df_crc32(n);
    DF_STAMP(3);
    // ---- code lengths: literal / length, distance, then the code-length code over their run-length form ----
    df_build_code(L_HL, DFL_NLIT, 15, L_LENL, L_CODL);
    df_build_code(L_HD, DFL_NDIST, 15, L_LEND, L_CODD);
    if (tid == 0) {
      const uint8_t* const lenL = DF_U8(L_LENL);
      const uint8_t* const lenD = DF_U8(L_LEND);
      int hlit = DFL_NLIT, hdist = DFL_NDIST;
      while (hlit > 257 && lenL[hlit - 1] == 0) --hlit;
      while (hdist > 1 && lenD[hdist - 1] == 0) --hdist;
      uint8_t* all = DF_U8(L_A);  // (free between the builds)
      for (int i = 0; i < hlit; ++i) all[i] = lenL[i];
      for (int i = 0; i < hdist; ++i) all[hlit + i] = lenD[i];
      ctl[C_NRLE] = (uint32_t)dfl_rle_lengths(all, hlit + hdist, DF_U16(L_RLE), hc);
      ctl[C_HLIT] = (uint32_t)hlit, ctl[C_HDIST] = (uint32_t)hdist;
    }
    __syncthreads();
    df_build_code(L_HC, DFL_NCL, 7, L_LENC, L_CODC);
    DF_STAMP(4);
    // ---- the block's size in bits; dynamic codes or a stored block ----
    {
      const uint8_t* const lenL = DF_U8(L_LENL);
      const uint8_t* const lenD = DF_U8(L_LEND);
      uint32_t bits = 0;
      for (uint32_t s = tid; s < DFL_NLIT; s += DF_NT) bits += hl[s] * lenL[s];
      if (tid < DFL_NDIST) bits += hd[tid] * lenD[tid];
      if (tid == 0) ctl[C_BITS] = 0;
      __syncthreads();
      if (bits) atomicAdd(&ctl[C_BITS], bits);
      __syncthreads();
    }
    if (tid == 0) {
      const uint8_t* const lenC = DF_U8(L_LENC);
      const uint16_t* const rle = DF_U16(L_RLE);
      int hclen = DFL_NCL;
      while (hclen > 4 && lenC[dfl_cl_order(hclen - 1)] == 0) --hclen;
      uint32_t hb = 3 + 5 + 5 + 4 + 3 * (uint32_t)hclen;
      const uint32_t nr = ctl[C_NRLE];
      for (uint32_t i = 0; i < nr; ++i) {
        const uint32_t sy = rle[i] & 0xFFu;
        hb += lenC[sy] + (sy == 16 ? 2u : (sy == 17 ? 3u : (sy == 18 ? 7u : 0u)));
      }
      const uint32_t total = hb + ctl[C_BITS] + ctl[C_XBITS];
      ctl[C_HCLEN] = (uint32_t)hclen;
      ctl[C_MODE] = total >= n * 8u + 32u ? 0u : 2u;  // (a stored block: 3 bits, padding, LEN, NLEN, the bytes)
      ctl[C_BITS] = hb;
    }
    __syncthreads();
    const uint32_t mode = ctl[C_MODE];
    uint8_t* const slot = slots + (size_t)m * DF_SLOT;
    uint32_t total_bytes;  // of the member
    if (mode == 0) {
      // ---- stored: header and trailer by one thread, the payload straight from the source run ----
      total_bytes = 18 + 5 + n + 8;
      const uint8_t* s = src + M.src;
      for (uint32_t i = tid; i < n; i += DF_NT) slot[18 + 5 + i] = s[i];
      if (tid == 0) {
        slot[18] = 1;  // BFINAL, BTYPE 00, padding
        slot[19] = (uint8_t)n, slot[20] = (uint8_t)(n >> 8), slot[21] = (uint8_t)~n, slot[22] = (uint8_t)(~n >> 8);
      }
    } else {
      // ---- the member image in LDS: 18 header bytes, the deflate bits, 8 trailer bytes ----
      DF_STAMP(5);
      const uint32_t endbit = df_write_block(tok, ntok);
      __syncthreads();
      DF_STAMP(6);
      const uint32_t dbytes = (endbit - 18 * 8 + 7) >> 3;
      total_bytes = 18 + dbytes + 8;
      __syncthreads();
      if (tid == 0) {
        const uint32_t crc = ctl[C_CRC];
        uint8_t* t = inb + 18 + dbytes;
        t[0] = (uint8_t)crc, t[1] = (uint8_t)(crc >> 8), t[2] = (uint8_t)(crc >> 16), t[3] = (uint8_t)(crc >> 24);
        t[4] = (uint8_t)n, t[5] = (uint8_t)(n >> 8), t[6] = 0, t[7] = 0;
      }
    }
    if (tid == 0) {
      // gzip member header with the BGZF extra field: BSIZE = member size - 1
      uint8_t* h = mode == 0 ? slot : inb;
      const uint32_t bs = total_bytes - 1;
      const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)bs, (uint8_t)(bs >> 8)};
      for (int i = 0; i < 18; ++i) h[i] = hdr[i];
      if (mode == 0) {
        const uint32_t crc = ctl[C_CRC];
        uint8_t* t = slot + 18 + 5 + n;
        t[0] = (uint8_t)crc, t[1] = (uint8_t)(crc >> 8), t[2] = (uint8_t)(crc >> 16), t[3] = (uint8_t)(crc >> 24);
        t[4] = (uint8_t)n, t[5] = (uint8_t)(n >> 8), t[6] = 0, t[7] = 0;
      }
      msize[m] = total_bytes;
    }
    if (mode != 0) {
      __syncthreads();
      uint32_t* const so = (uint32_t*)slot;  // (slots are 64 KiB apart: aligned)
      for (uint32_t i = tid; i < ((total_bytes + 3u) >> 2); i += DF_NT) so[i] = inw[i];
    }
    DF_STAMP(7);
  }
}

// the members' slots -> one run; a block per member
__global__ __launch_bounds__(256) void bgz_gather_k(uint32_t nmem, const uint8_t* __restrict__ slots, const uint32_t* __restrict__ msize,
                                                    const uint64_t* __restrict__ moff, uint8_t* __restrict__ out) {
  const uint32_t m = blockIdx.x;
  if (m >= nmem) return;
  const uint32_t sz = msize[m];
  const uint8_t* s = slots + (size_t)m * DF_SLOT;
  uint8_t* d = out + moff[m];
  // the destination starts at any byte: bytes up to its first 4-byte boundary, then dwords read unaligned from the slot
  const uint32_t head = min(sz, (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u));
  if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
  const uint32_t nw = (sz - head) >> 2;
  for (uint32_t i = threadIdx.x; i < nw; i += 256) {
    uint32_t v;
    __builtin_memcpy(&v, s + head + 4 * (size_t)i, 4);
    *(uint32_t*)(d + head + 4 * (size_t)i) = v;
  }
  const uint32_t tail0 = head + nw * 4;
  if (threadIdx.x < sz - tail0) d[tail0 + threadIdx.x] = s[tail0 + threadIdx.x];
}

// ---- records: plan, emit, cut --------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// size of the aux field at a (tag, type, value) inside [a, end), 0 when it is malformed (host: bam.cpp aux_field_size)
__device__ uint32_t aux_size_dev(const uint8_t* a, const uint8_t* end) {
  if (a + 3 > end) return 0;
  const uint8_t t = a[2];
  uint32_t sz;
  switch (t) {
    case 'A': case 'c': case 'C': sz = 1; break;
    case 's': case 'S': sz = 2; break;
    case 'i': case 'I': case 'f': sz = 4; break;
    case 'd': sz = 8; break;
    case 'Z': case 'H': {
      const uint8_t* q = a + 3;
      while (q < end && *q) ++q;
      if (q >= end) return 0;
      return (uint32_t)(q - a) + 1;
    }
    case 'B': {
      if (a + 8 > end) return 0;
      uint32_t es;
      switch (a[3]) {
        case 'c': case 'C': es = 1; break;
        case 's': case 'S': es = 2; break;
        case 'i': case 'I': case 'f': es = 4; break;
        default: return 0;
      }
      const uint64_t cnt = rd32(a + 4);
      const uint64_t tot = 8 + cnt * es;
      if (tot > (uint64_t)(end - a)) return 0;
      return (uint32_t)tot;
    }
    default: return 0;
  }
  return a + 3 + sz <= end ? 3 + sz : 0;
}

struct EncSrc {
  const uint8_t* dev_inf;    // inflated streams of the tile decoded on this context (may be null)
  const uint64_t* dev_rec;   // [n_dev] offsets of its records (block_size field)
  uint32_t n_dev;
  const uint8_t* blob;       // raw records the host handed over (block_size first), device copy
  const uint64_t* blob_off;  // [n_blob + 1]
  const uint32_t* blob_slot; // [n] for a group whose representative is not a device record: its slot in the blob
  uint32_t n_blob;           // records in the blob (tbk_enc_in::n_host): a slot beyond it is the caller's mistake, not a fault
};

__device__ __forceinline__ const uint8_t* enc_record(const EncSrc& S, const uint32_t* __restrict__ rep, uint32_t g, uint32_t* len) {
  const uint32_t r = rep[g];
  const uint8_t* p;
  if (r < S.n_dev) {
    p = S.dev_inf + S.dev_rec[r];
  } else {  // (enc_plan_k has checked the slot: the later passes only run when every group's record is there)
    if (!S.blob || S.blob_slot[g] >= S.n_blob) {
      *len = 0;
      return nullptr;
    }
    p = S.blob + S.blob_off[S.blob_slot[g]];
  }
  *len = rd32(p);  // block_size
  return p + 4;
}

// bam_aux_update_int's type and width for a new tag (htslib 1.18; host: BamRec::update_int)
__device__ __forceinline__ void int_tag_form(int64_t val, uint8_t* type, uint32_t* sz) {
  if (val < -32768) *type = 'i', *sz = 4;
  else if (val < -128) *type = 's', *sz = 2;
  else if (val < 0) *type = 'c', *sz = 1;
  else if (val < 255) *type = 'C', *sz = 1;
  else if (val < 65535) *type = 'S', *sz = 2;
  else *type = 'I', *sz = 4;
}

constexpr uint32_t ENC_NOTFRESH = 0x80000000u;

// a thread per output record: does it carry YC / YX / YD already, and how long is it once tagged
__global__ __launch_bounds__(256) void enc_plan_k(uint32_t n, EncSrc S, const uint32_t* __restrict__ rep, const double* __restrict__ yc, const int64_t* __restrict__ yx,
                                                  const int32_t* __restrict__ yd, uint32_t* __restrict__ olen, uint32_t* __restrict__ err) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  uint32_t len;
  const uint8_t* p = enc_record(S, rep, g, &len);
  if (!p) {  // rep[g] >= n_dev without a host record behind it (no blob, or host_slot[g] >= n_host): TBK_EINVAL, not a GPU fault
    atomicOr(err, 4u);
    olen[g] = 0;
    return;
  }
  if (len < 32) {
    atomicOr(err, 2u);
    olen[g] = 0;
    return;
  }
  const uint32_t l_qname = p[8], n_cig = p[12] | (uint32_t)p[13] << 8, l_seq = rd32(p + 16);
  const uint64_t aux0 = 32ull + l_qname + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq;
  if (aux0 > len) {
    atomicOr(err, 2u);
    olen[g] = 0;
    return;
  }
  const uint8_t* a = p + aux0;
  const uint8_t* end = p + len;
  // first occurrences (bam_aux_get) of the three tags
  const uint8_t *fc = nullptr, *fx = nullptr, *fd = nullptr;
  while (a + 3 <= end) {
    const uint32_t sz = aux_size_dev(a, end);
    if (!sz) break;
    if (a[0] == 'Y') {
      if (a[1] == 'C' && !fc) fc = a;
      if (a[1] == 'X' && !fx) fx = a;
      if (a[1] == 'D' && !fd) fd = a;
    }
    a += sz;
  }
  const int64_t vx = yx[g];
  const int32_t vd = yd[g];
  const bool x_ok = vx >= INT32_MIN && vx <= (int64_t)UINT32_MAX;  // (bam_aux_update_int refuses anything else: no edit)
  int64_t out = (int64_t)len + 4;
  // YC: 'f' stays, 'd' shrinks by four, anything else is left alone; missing: appended
  if (!fc) out += 7;
  else if (fc[2] == 'd') out -= 4;
  auto int_edit = [&](const uint8_t* f, int64_t val) {
    uint8_t ty;
    uint32_t sz;
    int_tag_form(val, &ty, &sz);
    if (!f) {
      out += 3 + sz;
      return;
    }
    uint32_t old;
    switch (f[2]) {
      case 'c': case 'C': old = 1; break;
      case 's': case 'S': old = 2; break;
      case 'i': case 'I': old = 4; break;
      default: return;  // not an integer tag: the stale value survives
    }
    if (old < sz) out += sz - old;
  };
  if (x_ok) int_edit(fx, vx);
  if (vd > 0) int_edit(fd, (int64_t)vd);
  else if (fd) out -= aux_size_dev(fd, end);
  (void)yc;
  olen[g] = (uint32_t)out | ((fc || fx || fd || !x_ok) ? ENC_NOTFRESH : 0u);
}

// the tagged record of group g written at o (block_size first).  Fresh records (no YC / YX / YD yet): 16 lanes copy, lane 0 appends.
__global__ __launch_bounds__(256) void enc_emit_k(uint32_t n, EncSrc S, const uint32_t* __restrict__ rep, const double* __restrict__ yc, const int64_t* __restrict__ yx,
                                                  const int32_t* __restrict__ yd, const uint32_t* __restrict__ olen, const uint64_t* __restrict__ ooff,
                                                  uint8_t* __restrict__ outp) {
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15u;
  if (g >= n) return;
  const uint32_t ol = olen[g];
  const uint32_t out_len = ol & ~ENC_NOTFRESH;
  if (out_len < 4) return;
  uint32_t len;
  const uint8_t* p = enc_record(S, rep, g, &len);
  uint8_t* o = outp + ooff[g];
  const uint32_t bs = out_len - 4;
  if (!(ol & ENC_NOTFRESH)) {
    // block_size | the record as it is | YC:f | YX | YD when positive
    if (sub == 0) {
      o[0] = (uint8_t)bs, o[1] = (uint8_t)(bs >> 8), o[2] = (uint8_t)(bs >> 16), o[3] = (uint8_t)(bs >> 24);
      uint8_t* t = o + 4 + len;
      const float f = (float)yc[g];
      uint32_t fb;
      __builtin_memcpy(&fb, &f, 4);
      t[0] = 'Y', t[1] = 'C', t[2] = 'f', t[3] = (uint8_t)fb, t[4] = (uint8_t)(fb >> 8), t[5] = (uint8_t)(fb >> 16), t[6] = (uint8_t)(fb >> 24);
      t += 7;
      auto put_int = [&](uint8_t t1, int64_t val) {
        uint8_t ty;
        uint32_t sz;
        int_tag_form(val, &ty, &sz);
        t[0] = 'Y', t[1] = t1, t[2] = ty;
        const uint32_t uv = (uint32_t)val;
        for (uint32_t q = 0; q < sz; ++q) t[3 + q] = (uint8_t)(uv >> (8 * q));
        t += 3 + sz;
      };
      put_int('X', yx[g]);
      if (yd[g] > 0) put_int('D', (int64_t)yd[g]);
    }
    // (source and destination start at any byte: byte copies up to the destination's 4-byte boundary, then dwords)
    uint8_t* d = o + 4;
    const uint32_t head = min(len, (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u));
    if (sub < head) d[sub] = p[sub];
    const uint32_t nw = (len - head) >> 2;
    for (uint32_t i = sub; i < nw; i += 16) {
      uint32_t v;
      __builtin_memcpy(&v, p + head + 4 * (size_t)i, 4);
      *(uint32_t*)(d + head + 4 * (size_t)i) = v;
    }
    const uint32_t t0 = head + nw * 4;
    if (sub < len - t0) d[t0 + sub] = p[t0 + sub];
    return;
  }
  if (sub != 0) return;
  // ---- a record that carries one of the tags already: the edits of bam_aux_update_float / _int / bam_aux_del, in the order
  // flushPData makes them (YC, YX, YD), each on the first occurrence of its tag; a missing tag is appended ----
  o[0] = (uint8_t)bs, o[1] = (uint8_t)(bs >> 8), o[2] = (uint8_t)(bs >> 16), o[3] = (uint8_t)(bs >> 24);
  uint8_t* w = o + 4;
  const uint32_t l_qname = p[8], n_cig = p[12] | (uint32_t)p[13] << 8, l_seq = rd32(p + 16);
  const uint32_t aux0 = 32u + l_qname + 4u * n_cig + (l_seq + 1) / 2 + l_seq;
  for (uint32_t i = 0; i < aux0; ++i) w[i] = p[i];
  w += aux0;
  const uint8_t* a = p + aux0;
  const uint8_t* end = p + len;
  const int64_t vx = yx[g];
  const int32_t vd = yd[g];
  const bool x_ok = vx >= INT32_MIN && vx <= (int64_t)UINT32_MAX;
  const float f = (float)yc[g];
  uint32_t fb;
  __builtin_memcpy(&fb, &f, 4);
  bool sc = false, sx = false, sd = false;
  auto put_val = [&](uint32_t uv, uint32_t sz) {
    for (uint32_t q = 0; q < sz; ++q) *w++ = (uint8_t)(uv >> (8 * q));
  };
  auto int_field = [&](const uint8_t* fld, int64_t val) {  // an existing integer tag takes the value (width never shrinks)
    uint8_t ty;
    uint32_t sz;
    int_tag_form(val, &ty, &sz);
    uint32_t old;
    switch (fld[2]) {
      case 'c': case 'C': old = 1; break;
      case 's': case 'S': old = 2; break;
      case 'i': case 'I': old = 4; break;
      default: old = 0; break;
    }
    if (old == 0) return false;
    if (old >= sz) {
      sz = old;
      ty = (uint8_t)((val < 0 ? "\0cs\0i" : "\0CS\0I")[old]);
    }
    *w++ = fld[0], *w++ = fld[1], *w++ = ty;
    put_val((uint32_t)val, sz);
    return true;
  };
  while (a + 3 <= end) {
    const uint32_t sz = aux_size_dev(a, end);
    if (!sz) break;
    bool done = false;
    if (a[0] == 'Y') {
      if (a[1] == 'C' && !sc) {
        sc = true;
        if (a[2] == 'f' || a[2] == 'd') {
          *w++ = 'Y', *w++ = 'C', *w++ = 'f';
          put_val(fb, 4);
          done = true;
        }
      } else if (a[1] == 'X' && !sx) {
        sx = true;
        if (x_ok) done = int_field(a, vx);
      } else if (a[1] == 'D' && !sd) {
        sd = true;
        if (vd > 0) done = int_field(a, (int64_t)vd);
        else done = true;  // bam_aux_del
      }
    }
    if (!done)
      for (uint32_t i = 0; i < sz; ++i) *w++ = a[i];
    a += sz;
  }
  for (; a < end; ++a) *w++ = *a;  // (whatever follows a malformed field stays as it is)
  if (!sc) {
    *w++ = 'Y', *w++ = 'C', *w++ = 'f';
    put_val(fb, 4);
  }
  auto append_int = [&](uint8_t t1, int64_t val) {
    uint8_t ty;
    uint32_t sz;
    int_tag_form(val, &ty, &sz);
    *w++ = 'Y', *w++ = t1, *w++ = ty;
    put_val((uint32_t)val, sz);
  };
  if (!sx && x_ok) append_int('X', vx);
  if (!sd && vd > 0) append_int('D', (int64_t)vd);
}

// member m covers the records that begin in [m * B, (m + 1) * B): cut[m] = the first record offset >= m * B
__global__ __launch_bounds__(256) void enc_cuts_k(uint32_t nmem, uint32_t B, const uint64_t* __restrict__ ooff, uint32_t n, uint64_t* __restrict__ cut) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m > nmem) return;
  const uint64_t want = (uint64_t)m * B;
  uint32_t lo = 0, hi = n;  // ooff[n] = total
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (ooff[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  cut[m] = m == nmem ? ooff[n] : ooff[lo];
}
__global__ __launch_bounds__(256) void enc_members_k(uint32_t nmem, const uint64_t* __restrict__ cut, DfMember* __restrict__ mem) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nmem) return;
  DfMember M;
  M.src = cut[m];
  M.n = (uint32_t)(cut[m + 1] - cut[m]);
  M.pad = 0;
  mem[m] = M;
}
__global__ __launch_bounds__(256) void enc_maxlen_k(uint32_t n, const uint32_t* __restrict__ olen, uint32_t* __restrict__ mx) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = g < n ? (olen[g] & ~ENC_NOTFRESH) : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d));
  if ((threadIdx.x & 63u) == 0 && v) atomicMax(mx, v);
}
__global__ __launch_bounds__(256) void enc_strip_k(uint32_t n, uint32_t* __restrict__ olen_plain, const uint32_t* __restrict__ olen) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= n) olen_plain[g] = g < n ? (olen[g] & ~ENC_NOTFRESH) : 0u;
}

// the encoder's device buffers: owned by the context, grown as needed, freed with it
struct EncState {
  void* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
enum { EB_PAY = 0, EB_SLOTS = 1, EB_PACKED = 2, EB_TOK = 3, EB_SMALL = 4, EB_BLOB = 5, EB_GRP = 6, EB_MISC = 7 };

int enc_buf(tbk_ctx* ctx, int which, size_t bytes, void** out) {
  EncState* E = (EncState*)ctx->enc;
  if (!E) ctx->enc = E = new EncState();
  if (bytes > E->cap[which]) {
    if (E->p[which]) (void)hipFree(E->p[which]);
    E->p[which] = nullptr;
    E->cap[which] = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc(&E->p[which], want) != hipSuccess) {
      (void)hipGetLastError();
      return TBK_ENOMEM;
    }
    E->cap[which] = want;
  }
  *out = E->p[which];
  return 0;
}

int deflate_grid(tbk_ctx* ctx) {
  static int blocks_per_cu = 0;
  if (!blocks_per_cu) {
    (void)hipFuncSetAttribute((const void*)bgz_deflate_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_END);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bgz_deflate_k, DF_NT, L_END) != hipSuccess || nb < 1) nb = 1;
    blocks_per_cu = nb > 2 ? 2 : nb;
  }
  return ctx->num_cu * blocks_per_cu;
}

// payload run (device) + member table (device) -> packed members in *d_out (device, EB_PACKED) and their total size
int deflate_members(tbk_ctx* ctx, const uint8_t* d_src, const DfMember* d_mem, uint32_t nmem, uint8_t** d_out, uint64_t* total) {
  *total = 0;
  *d_out = nullptr;
  if (nmem == 0) return 0;
  const int grid = std::min<int>(deflate_grid(ctx), (int)nmem);
  uint8_t* slots;
  uint32_t* tok;
  uint8_t* small;
  TBK_TRY(enc_buf(ctx, EB_SLOTS, (size_t)nmem * DF_SLOT, (void**)&slots));
  TBK_TRY(enc_buf(ctx, EB_TOK, (size_t)deflate_grid(ctx) * DF_MAXPAY * 4, (void**)&tok));
  TBK_TRY(enc_buf(ctx, EB_SMALL, ((size_t)nmem + 1) * 12 + 256, (void**)&small));
  uint32_t* msize = (uint32_t*)small;                                               // [nmem + 1]
  uint64_t* moff = (uint64_t*)(small + ((((size_t)nmem + 1) * 4 + 255) & ~(size_t)255));  // [nmem + 1]
  uint32_t* counter = (uint32_t*)(ctx->d_scalars + 2);
  TBK_HIP(hipMemsetAsync(ctx->d_scalars, 0, 16 * sizeof(uint64_t), ctx->stream));
  TBK_HIP(hipMemsetAsync(msize + nmem, 0, 4, ctx->stream));
  TBK_LAUNCH(ctx, "bgz_deflate", bgz_deflate_k, grid, DF_NT, L_END, nmem, d_mem, d_src, slots, msize, tok, counter, ctx->d_err);
  TBK_TRY(tbk_exscan_u32_u64(ctx, msize, moff, nmem + 1, ctx->d_scalars + 1));
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) {
    ctx->last_error = "bgz_deflate: a member beyond 0xff00 payload bytes";
    return TBK_EINVAL;
  }
  *total = ctx->h_scalars[1];
  uint8_t* packed;
  TBK_TRY(enc_buf(ctx, EB_PACKED, (size_t)*total + 16, (void**)&packed));
  TBK_LAUNCH(ctx, "bgz_gather", bgz_gather_k, nmem, 256, 0, nmem, slots, msize, moff, packed);
  *d_out = packed;
  return 0;
}

}  // namespace

#ifdef DF_PROF
extern "C" int tbk_debug_deflate_phases(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(df_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(df_prof), z, sizeof(z));
  }
  return 0;
}
#endif

void tbk_enc_free(tbk_ctx* ctx) {
  EncState* E = (EncState*)ctx->enc;
  if (!E) return;
  for (int i = 0; i < 8; ++i)
    if (E->p[i]) (void)hipFree(E->p[i]);
  delete E;
  ctx->enc = nullptr;
}

extern "C" int tbk_bgzf_deflate(tbk_ctx* ctx, const uint8_t* src, uint64_t n, int src_mem, const uint64_t* cuts, uint32_t n_members, uint8_t* out,
                                uint64_t out_cap, uint64_t* out_bytes) {
  if (!ctx || !out_bytes || (n && !src)) return TBK_EINVAL;
  *out_bytes = 0;
  if (n == 0) return 0;
  TBK_HIP(hipSetDevice(ctx->device));
  // the member table: the caller's cut points, or 0xff00-byte pieces
  std::vector<DfMember> mt;
  if (cuts) {
    if (n_members == 0 || cuts[0] != 0 || cuts[n_members] != n) return TBK_EINVAL;
    for (uint32_t m = 0; m < n_members; ++m) {
      if (cuts[m + 1] < cuts[m] || cuts[m + 1] - cuts[m] > DF_MAXPAY) return TBK_EINVAL;
      mt.push_back(DfMember{cuts[m], (uint32_t)(cuts[m + 1] - cuts[m]), 0});
    }
  } else {
    for (uint64_t o = 0; o < n; o += DF_MAXPAY) mt.push_back(DfMember{o, (uint32_t)std::min<uint64_t>(DF_MAXPAY, n - o), 0});
  }
  if (mt.size() >= (1ull << 31)) return TBK_E2BIG;
  const uint32_t nmem = (uint32_t)mt.size();
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)nmem * 16 + ((size_t)1 << 20)));  // (the scans' partial sums)
  const uint8_t* d_src = src;
  if (src_mem != TBK_MEM_DEVICE) {
    uint8_t* pay;
    TBK_TRY(enc_buf(ctx, EB_PAY, n + 16, (void**)&pay));
    TBK_HIP(hipMemcpyAsync(pay, src, n, hipMemcpyHostToDevice, ctx->stream));
    d_src = pay;
  }
  DfMember* d_mem;
  TBK_TRY(enc_buf(ctx, EB_MISC, (size_t)nmem * sizeof(DfMember), (void**)&d_mem));
  TBK_HIP(hipMemcpyAsync(d_mem, mt.data(), (size_t)nmem * sizeof(DfMember), hipMemcpyHostToDevice, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));  // (mt is a local)
  uint8_t* packed;
  uint64_t total;
  TBK_TRY(deflate_members(ctx, d_src, d_mem, nmem, &packed, &total));
  *out_bytes = total;
  if (total > out_cap || (total && !out)) return TBK_E2BIG;
  TBK_HIP(hipMemcpyAsync(out, packed, total, hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "bgzf_deflate");
}

extern "C" int tbk_bam_encode(tbk_ctx* ctx, const tbk_enc_in* in, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes, uint64_t* payload_bytes) {
  if (!ctx || !in || !out_bytes) return TBK_EINVAL;
  *out_bytes = 0;
  if (payload_bytes) *payload_bytes = 0;
  const uint32_t n = in->n;
  if (n == 0) return 0;
  const bool kept = in->mem == TBK_MEM_KEPT;  // the columns of a context's last collapse (tbk_collapse_opts.keep_results), read where they lie
  tbk_ctx* const src = in->from ? in->from : ctx;  // whose kept results / decoded tile (read only; the same device)
  if (src->device != ctx->device) return TBK_EINVAL;
  if (kept ? (!src->kept || (uint64_t)in->first + n > src->kept_n) : (!in->rep || !in->yc || !in->yx || !in->yd)) return TBK_EINVAL;
  if (in->mem != TBK_MEM_HOST && in->mem != TBK_MEM_DEVICE && !kept) return TBK_EINVAL;
  if (in->n_host && (!in->host_blob || !in->host_off || !in->host_slot)) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  EncSrc S;
  memset(&S, 0, sizeof(S));
  S.n_dev = in->n_dev;
  if (in->n_dev) {
    if (!tbk_bam_dev_records(src, &S.dev_inf, &S.dev_rec, &S.n_dev) || S.n_dev < in->n_dev) return TBK_EINVAL;
    S.n_dev = in->n_dev;
  }
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  hipStream_t st = ctx->stream;
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n * 16 + ((size_t)1 << 20)));  // (the scans' partial sums)
  // ---- the groups' columns and the host's records onto the device ----
  const size_t a256 = 255;
  auto al = [&](size_t x) { return (x + a256) & ~a256; };
  const size_t grp_bytes = al((size_t)n * 4) * 3 + al((size_t)n * 8) * 2 + al(((size_t)n + 1) * 4) * 2 + al(((size_t)n + 1) * 8) + 4096;
  uint8_t* gb;
  TBK_TRY(enc_buf(ctx, EB_GRP, grp_bytes, (void**)&gb));
  auto take = [&](size_t bytes) {
    uint8_t* r = gb;
    gb += al(bytes);
    return r;
  };
  const uint32_t* d_rep = (uint32_t*)take((size_t)n * 4);
  const int32_t* d_yd = (int32_t*)take((size_t)n * 4);
  uint32_t* d_slot = (uint32_t*)take((size_t)n * 4);
  const double* d_yc = (double*)take((size_t)n * 8);
  const int64_t* d_yx = (int64_t*)take((size_t)n * 8);
  uint32_t* d_olen = (uint32_t*)take(((size_t)n + 1) * 4);
  uint32_t* d_olen2 = (uint32_t*)take(((size_t)n + 1) * 4);
  uint64_t* d_ooff = (uint64_t*)take(((size_t)n + 1) * 8);
  if (kept) {
    d_rep = src->kept_rep + in->first, d_yd = src->kept_yd + in->first, d_yc = src->kept_yc + in->first, d_yx = src->kept_yx + in->first;
  } else {
    const hipMemcpyKind kind = in->mem == TBK_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    TBK_HIP(hipMemcpyAsync((void*)d_rep, in->rep, (size_t)n * 4, kind, st));
    TBK_HIP(hipMemcpyAsync((void*)d_yd, in->yd, (size_t)n * 4, kind, st));
    TBK_HIP(hipMemcpyAsync((void*)d_yc, in->yc, (size_t)n * 8, kind, st));
    TBK_HIP(hipMemcpyAsync((void*)d_yx, in->yx, (size_t)n * 8, kind, st));
  }
  if (in->n_host) {
    const uint64_t hb = in->host_off[in->n_host];
    uint8_t* blob;
    TBK_TRY(enc_buf(ctx, EB_BLOB, al(hb + 16) + al(((size_t)in->n_host + 1) * 8), (void**)&blob));
    uint64_t* d_boff = (uint64_t*)(blob + al(hb + 16));
    TBK_HIP(hipMemcpyAsync(blob, in->host_blob, hb, hipMemcpyHostToDevice, st));
    TBK_HIP(hipMemcpyAsync(d_boff, in->host_off, ((size_t)in->n_host + 1) * 8, hipMemcpyHostToDevice, st));
    TBK_HIP(hipMemcpyAsync(d_slot, in->host_slot, (size_t)n * 4, hipMemcpyHostToDevice, st));
    S.blob = blob;
    S.blob_off = d_boff;
    S.blob_slot = d_slot;
    S.n_blob = in->n_host;
  }
  // ---- plan, offsets, payload ----
  TBK_HIP(hipMemsetAsync(ctx->d_scalars, 0, 16 * sizeof(uint64_t), st));
  uint32_t* d_max = (uint32_t*)(ctx->d_scalars + 3);
  TBK_LAUNCH(ctx, "enc_plan", enc_plan_k, cdiv(n, 256), 256, 0, n, S, d_rep, d_yc, d_yx, d_yd, d_olen, ctx->d_err);
  TBK_LAUNCH(ctx, "enc_maxlen", enc_maxlen_k, cdiv(n, 256), 256, 0, n, d_olen, d_max);
  TBK_LAUNCH(ctx, "enc_strip", enc_strip_k, cdiv((uint64_t)n + 1, 256), 256, 0, n, d_olen2, d_olen);
  TBK_TRY(tbk_exscan_u32_u64(ctx, d_olen2, d_ooff, n + 1, ctx->d_scalars + 1));
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) {
    ctx->last_error = (eb & 4u) ? "bam_encode: a representative beyond n_dev without a host record (host_slot >= n_host, or no host_blob)"
                                : "bam_encode: a malformed record among the representatives";
    return TBK_EINVAL;
  }
  const uint64_t total = ctx->h_scalars[1];
  const uint32_t maxrec = (uint32_t)ctx->h_scalars[3];
  if (payload_bytes) *payload_bytes = total;
  if (maxrec + 256u > DF_MAXPAY) return TBK_EUNSUPPORTED;  // a record as long as a member: the host writer cuts such a stream
  const uint32_t B = DF_MAXPAY - maxrec;                    // a member's records begin inside B bytes and end inside 0xff00
  const uint64_t nmem64 = (total + B - 1) / B;
  if (nmem64 >= (1ull << 31)) return TBK_E2BIG;
  const uint32_t nmem = (uint32_t)nmem64;
  uint8_t* pay;
  TBK_TRY(enc_buf(ctx, EB_PAY, total + 16, (void**)&pay));
  TBK_LAUNCH(ctx, "enc_emit", enc_emit_k, cdiv((uint64_t)n * 16, 256), 256, 0, n, S, d_rep, d_yc, d_yx, d_yd, d_olen, d_ooff, pay);
  uint8_t* misc;
  TBK_TRY(enc_buf(ctx, EB_MISC, al(((size_t)nmem + 1) * 8) + (size_t)nmem * sizeof(DfMember), (void**)&misc));
  uint64_t* d_cut = (uint64_t*)misc;
  DfMember* d_mem = (DfMember*)(misc + al(((size_t)nmem + 1) * 8));
  TBK_LAUNCH(ctx, "enc_cuts", enc_cuts_k, cdiv((uint64_t)nmem + 1, 256), 256, 0, nmem, B, d_ooff, n, d_cut);
  TBK_LAUNCH(ctx, "enc_members", enc_members_k, cdiv(nmem, 256), 256, 0, nmem, d_cut, d_mem);
  uint8_t* packed;
  uint64_t ztotal;
  TBK_TRY(deflate_members(ctx, pay, d_mem, nmem, &packed, &ztotal));
  *out_bytes = ztotal;
  if (ztotal > out_cap || (ztotal && !out)) return TBK_E2BIG;
  TBK_HIP(hipMemcpyAsync(out, packed, ztotal, hipMemcpyDeviceToHost, st));
  TBK_HIP(hipStreamSynchronize(st));
  return tbk_check_launch(ctx, "bam_encode");
}
