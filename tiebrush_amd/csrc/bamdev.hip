// bamdev.hip — BGZF inflate and BAM record decode on gfx950: what GSamReader::next() -> sam_read1() does per record on the
// host (/root/reference/src/GSam.h:506-516, htslib bgzf_read_block + bam_read1) done for whole files at once on the device.
//
//   tbk_bam_decode    compressed BGZF members of k files (host memory) -> device-resident SoA tile (tbk_soa_in, TBK_MEM_DEVICE)
//     bgz_inflate_wave_k  one wave per BGZF member (members are independent raw-deflate streams of <= 64 KiB): RFC 1951 decoder
//                       with multi-bit tables in LDS, the last 8 KiB of output in an LDS ring; bgz_crc_k: CRC32 and ISIZE of every
//                       member are verified like htslib does
//     bam_index_k       one workgroup per file walks the record chain (block_size -> next record) through LDS-staged chunks
//     bam_fields_k      one thread per record: core fields, the aux scan of the host loader (NH, XS / ts -> spliceStrand,
//                       carried YC / YX / YD of TieBrush-merged inputs, MD and QNAME sizes), field-length validation
//     bam_fill_k        CIGAR words, MD bytes, names + name hash at the offsets the scans produced
//   tbk_bam_gather    the raw records behind tile indices (the representatives), packed, to host memory for tagging
//
// Integer / byte work; the inflate is bound by its dependent bit-serial decode (every member in flight at once hides it:
// ~30 k members on 256 CUs), not by HBM.  No MFMA.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "crc32_block.hpp"
#include "dev_common.hpp"
#include "tbk_internal.h"

namespace {

// ---- RFC 1951 inflate ------------------------------------------------------------------------

struct BitIn {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t buf;
  int cnt;
  bool bad;
  __device__ __forceinline__ void fill() {
    while (cnt <= 56) {
      uint64_t b = 0;
      if (p < end)
        b = *p;
      ++p;  // past the end: zeros flow in; `bad` is raised when more than 8 bytes beyond the end are consumed
      buf |= b << cnt;
      cnt += 8;
    }
  }
  __device__ __forceinline__ uint32_t bits(int n) {  // n <= 32
    if (cnt < n) fill();
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1ull));
    buf >>= n;
    cnt -= n;
    return v;
  }
  __device__ __forceinline__ bool overrun() const { return p > end + 8; }
};

__constant__ uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// canonical Huffman table from code lengths (puff.c's construct): cnt[len] = codes of that length, sym = symbols in code
// order.  Returns < 0 for an over-subscribed set, > 0 for an incomplete one, 0 for a complete one.
__device__ int huff_build(uint16_t* cnt, uint16_t* sym, const uint8_t* lens, int n) {
  for (int l = 0; l <= 15; ++l) cnt[l] = 0;
  for (int s = 0; s < n; ++s) cnt[lens[s]]++;
  if (cnt[0] == n) return 0;  // no codes: complete, but decoding will fail
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= cnt[l];
    if (left < 0) return left;
  }
  uint16_t offs[16];
  offs[1] = 0;
  for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + cnt[l];
  for (int s = 0; s < n; ++s)
    if (lens[s] != 0) sym[offs[lens[s]]++] = (uint16_t)s;
  return left;
}

__device__ __forceinline__ int huff_decode(BitIn& in, const uint16_t* cnt, const uint16_t* sym) {
  int code = 0, first = 0, index = 0;
  if (in.cnt < 15) in.fill();
  uint32_t bitbuf = (uint32_t)in.buf;
  for (int len = 1; len <= 15; ++len) {
    code |= (int)(bitbuf & 1u);
    bitbuf >>= 1;
    const int count = cnt[len];
    if (code - count < first) {
      in.buf >>= len;
      in.cnt -= len;
      return sym[index + (code - first)];
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

struct BgzMember {
  uint64_t src;   // offset of the deflate stream in the compressed buffer
  uint64_t dst;   // offset of the member's payload in the inflated buffer
  uint32_t clen;  // deflate bytes
  uint32_t isize; // payload bytes
  uint32_t crc;   // CRC32 of the payload
  uint32_t file;
};

// ---- RFC 1951 inflate, one WAVE per member (round 4) ----------------------------------------------------------------------------
// The lane-per-member kernel above runs 64 different decoders in lock step: every branch of every decoder is executed by the whole
// wave, every output byte is a store of its own, every match copy goes through memory.  Here a wave works on ONE member: the decode
// is executed uniformly (all lanes hold the same bit buffer and take the same branches: no divergence), symbols come from multi-bit
// tables in LDS (10 bits for literals / lengths, 9 for distances; longer codes fall back to the canonical walk), the last 32 KiB
// of output — all a deflate distance can reach — live in an LDS ring, a match is copied by the 64 lanes together, and the output
// leaves the ring in coalesced runs of 16 KiB.  The compressed stream is staged through LDS 2 KiB at a time.  CRC32 is checked by
// bgz_crc_k afterwards (a workgroup per member).
#ifndef IW_WIN_BYTES
#define IW_WIN_BYTES 8192
#endif
// IW_WIN: the LDS ring.  Not the 32 KiB a deflate distance can reach but the 8 KiB nearly every distance in a BAM stream stays within
// (the previous records: names, tags, quality runs): four times as many waves per CU hide each other's LDS round trips, and the
// rare match from farther back is read from the output in memory, which is flushed every IW_FLUSH bytes (before the ring is).
constexpr int IW_WIN = IW_WIN_BYTES, IW_FLUSH = IW_WIN / 4, IW_CIN = 2048, IW_LBITS = 10, IW_DBITS = 9;

struct IwBits {  // uniform across the wave
  uint64_t buf;
  int cnt;
  uint32_t cpos;   // next byte of the stream to enter the buffer (multiple of 4)
  uint32_t cbase;  // stream offset of cin[0] (multiple of 4)
};

__global__ __launch_bounds__(64) void bgz_inflate_wave_k(uint32_t nmem, const BgzMember* __restrict__ mem, const uint8_t* __restrict__ src,
                                                         uint8_t* __restrict__ dst, uint32_t* __restrict__ err) {
  __shared__ __align__(16) uint8_t win[IW_WIN];
  __shared__ __align__(16) uint8_t cin[IW_CIN];
  __shared__ uint16_t ltab[1 << IW_LBITS], dtab[1 << IW_DBITS];  // entry: symbol << 4 | code length; 0: not in the table
  __shared__ uint16_t lcnt[16], lsym[288], dcnt[16], dsym[32];    // canonical form (puff.c): codes per length, symbols in code order
  __shared__ uint8_t lens[32 + 320];  // [0, 19): the code-length code; [32, 32 + 316): literal / length and distance code lengths
  __shared__ uint16_t codes[288];
  const uint32_t lane = threadIdx.x;
  const uint32_t m = blockIdx.x;
  if (m >= nmem) return;
  const BgzMember M = mem[m];
  const uint8_t* cs = src + M.src;
  uint8_t* out = dst + M.dst;
  const uint32_t clen = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.clen), cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.isize);
  IwBits B{0ull, 0, 0u, 0u};
  auto stage = [&](uint32_t base) {  // cin <- stream bytes [base, base + IW_CIN), zeros beyond the member
    __syncthreads();
    uint8_t t[IW_CIN / 64];  // (all the loads first: a loop of load-and-store waits for memory once per byte)
#pragma unroll
    for (int q = 0; q < IW_CIN / 64; ++q) {
      const uint32_t i = (uint32_t)q * 64u + lane;
      t[q] = base + i < clen ? cs[base + i] : (uint8_t)0;
    }
#pragma unroll
    for (int q = 0; q < IW_CIN / 64; ++q) cin[(uint32_t)q * 64u + lane] = t[q];
    __syncthreads();
    B.cbase = base;
  };
  auto refill = [&]() {  // at least 32 bits in the buffer afterwards
    while (B.cnt <= 32) {
      if (B.cpos + 4u > B.cbase + IW_CIN) stage(B.cpos);
      // (what comes out of LDS is the same in every lane; saying so — readfirstlane — keeps the decoder's state in scalar registers:
      // scalar shifts and branches instead of 64-lane ones under lane masks)
      const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<const uint32_t*>(cin + (B.cpos - B.cbase)));
      B.buf |= (uint64_t)w << B.cnt;
      B.cnt += 32;
      B.cpos += 4;
    }
  };
  auto bits = [&](int n) -> uint32_t {  // n <= 32
    if (B.cnt < n) refill();
    const uint32_t v = (uint32_t)(B.buf & ((1ull << n) - 1ull));
    B.buf >>= n;
    B.cnt -= n;
    return v;
  };
  auto overrun = [&]() { return B.cpos - (uint32_t)(B.cnt >> 3) > clen + 8u; };  // more than 8 bytes beyond the stream consumed
  // canonical walk over (cnt, sym): a code of any length (the slow path of the table decode and the code-length code)
  auto walk = [&](const uint16_t* cnt, const uint16_t* sym) -> int {
    if (B.cnt < 15) refill();
    int code = 0, first = 0, index = 0;
    uint32_t bb = (uint32_t)B.buf;
    for (int len = 1; len <= 15; ++len) {
      code |= (int)(bb & 1u);
      bb >>= 1;
      const int count = __builtin_amdgcn_readfirstlane((int)cnt[len]);
      if (code - count < first) {
        B.buf >>= len;
        B.cnt -= len;
        return __builtin_amdgcn_readfirstlane((int)sym[index + (code - first)]);
      }
      index += count;
      first += count;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  };
  // cnt / sym and the multi-bit table of a code given by lens[off .. off + n); puff.c's verdict: < 0 over-subscribed, > 0 incomplete
  auto build = [&](uint16_t* cnt, uint16_t* sym, uint16_t* tab, int tbits, int off, int n) -> int {
    __syncthreads();
    if (lane < 16) cnt[lane] = 0;
    for (uint32_t i = lane; i < (1u << tbits); i += 64) tab[i] = 0;
    __syncthreads();
    if (lane == 0) {
      for (int sy = 0; sy < n; ++sy) cnt[lens[off + sy]]++;
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)cnt[0]) == n) return 0;  // no codes: complete, but decoding will fail
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= __builtin_amdgcn_readfirstlane((int)cnt[l]);
      if (left < 0) return left;
    }
    if (lane == 0) {  // symbols in code order, and every symbol's canonical code
      uint16_t offs[16], next[16];
      offs[1] = 0;
      for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + cnt[l];
      uint32_t c = 0;
      for (int l = 1; l <= 15; ++l) {  // first code of every length (RFC 1951 3.2.2)
        next[l] = (uint16_t)c;
        c = (c + cnt[l]) << 1;
      }
      for (int sy = 0; sy < n; ++sy) {
        const int l = lens[off + sy];
        if (l != 0) {
          sym[offs[l]++] = (uint16_t)sy;
          codes[sy] = next[l]++;
        }
      }
    }
    __syncthreads();
    for (int sy = (int)lane; sy < n; sy += 64) {  // the table: every index whose low l bits are the (bit-reversed) code
      const int l = lens[off + sy];
      if (l == 0 || l > tbits) continue;
      const uint32_t rev = __brev((uint32_t)codes[sy]) >> (32 - l);
      const uint16_t e = (uint16_t)((sy << 4) | l);
      for (uint32_t k = rev; k < (1u << tbits); k += 1u << l) tab[k] = e;
    }
    __syncthreads();
    return left;
  };
  auto decode = [&](const uint16_t* tab, int tbits, const uint16_t* cnt, const uint16_t* sym) -> int {
    if (B.cnt < 15) refill();
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[(uint32_t)B.buf & ((1u << tbits) - 1u)]);
    if (e) {
      B.buf >>= (e & 15);
      B.cnt -= (e & 15);
      return e >> 4;
    }
    return walk(cnt, sym);
  };
  uint32_t o = 0, flushed = 0;
  bool bad = false;
  stage(0);
  auto flush = [&](uint32_t upto) {  // ring -> memory, bytes [flushed, upto)
    __syncthreads();
    for (uint32_t i = flushed + lane; i < upto; i += 64) out[i] = win[i & (IW_WIN - 1)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a match from beyond the ring reads these bytes back from memory)
    __syncthreads();
    flushed = upto;
  };
  for (;;) {
    const uint32_t last = bits(1);
    const uint32_t type = bits(2);
    if (type == 0) {  // stored
      B.buf >>= (B.cnt & 7);
      B.cnt -= (B.cnt & 7);
      const uint32_t len = bits(16), nlen = bits(16);
      if ((len ^ 0xFFFFu) != nlen || o + len > cap) {
        bad = true;
        break;
      }
      for (uint32_t i = 0; i < len; ++i) {
        const uint32_t v = bits(8);
        if (lane == 0) win[o & (IW_WIN - 1)] = (uint8_t)v;
        ++o;
        if (o - flushed >= (uint32_t)IW_FLUSH) flush(o);
      }
      if (overrun()) {  // a stored block cut off by the end of the member (zeros flowed in)
        bad = true;
        break;
      }
    } else if (type == 3) {
      bad = true;
      break;
    } else {
      if (type == 1) {  // fixed codes
        __syncthreads();
        for (uint32_t sy = lane; sy < 288; sy += 64) lens[sy] = sy < 144 ? 8 : (sy < 256 ? 9 : (sy < 280 ? 7 : 8));
        if (lane < 30) lens[288 + lane] = 5;
        __syncthreads();
        (void)build(lcnt, lsym, ltab, IW_LBITS, 0, 288);
        (void)build(dcnt, dsym, dtab, IW_DBITS, 288, 30);
      } else {  // dynamic codes
        const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
        if (nlen > 286 || ndist > 30) {
          bad = true;
          break;
        }
        __syncthreads();
        if (lane < 19) lens[lane] = 0;
        __syncthreads();
        for (int i = 0; i < ncode; ++i) {
          const uint32_t v = bits(3);
          if (lane == 0) lens[kClOrder[i]] = (uint8_t)v;
        }
        if (build(lcnt, lsym, ltab, 7, 0, 19) != 0) {  // the code-length code must be complete (its table: the low 7 bits of ltab)
          bad = true;
          break;
        }
        // the code lengths themselves: decoded with the code-length code, written behind it (lens[32 ..]) so that the table of
        // that code stays valid while they are read
        int idx = 0;
        constexpr int LO = 32;
        while (idx < nlen + ndist) {
          const int sy = decode(ltab, 7, lcnt, lsym);
          if (sy < 0) {
            bad = true;
            break;
          }
          if (sy < 16) {
            if (lane == 0) lens[LO + idx] = (uint8_t)sy;
            ++idx;
          } else {
            int rep, val = 0;
            if (sy == 16) {
              if (idx == 0) {
                bad = true;
                break;
              }
              __syncthreads();
              val = __builtin_amdgcn_readfirstlane((int)lens[LO + idx - 1]);
              rep = 3 + (int)bits(2);
            } else if (sy == 17) {
              rep = 3 + (int)bits(3);
            } else {
              rep = 11 + (int)bits(7);
            }
            if (idx + rep > nlen + ndist) {
              bad = true;
              break;
            }
            if (lane == 0)
              for (int q = 0; q < rep; ++q) lens[LO + idx + q] = (uint8_t)val;
            idx += rep;
            __syncthreads();
          }
        }
        if (bad) break;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)lens[LO + 256]) == 0) {  // no end-of-block code
          bad = true;
          break;
        }
        int e = build(lcnt, lsym, ltab, IW_LBITS, LO, nlen);
        if (e < 0 || (e > 0 && nlen - __builtin_amdgcn_readfirstlane((int)lcnt[0]) != 1)) {  // over-subscribed, or incomplete with more than one code
          bad = true;
          break;
        }
        e = build(dcnt, dsym, dtab, IW_DBITS, LO + nlen, ndist);
        if (e < 0 || (e > 0 && ndist - __builtin_amdgcn_readfirstlane((int)dcnt[0]) != 1)) {
          bad = true;
          break;
        }
      }
      // literals and length / distance pairs
      for (;;) {
        int sy = decode(ltab, IW_LBITS, lcnt, lsym);
        if (sy < 0 || overrun()) {
          bad = true;
          break;
        }
        if (sy < 256) {
          if (o >= cap) {
            bad = true;
            break;
          }
          win[o & (IW_WIN - 1)] = (uint8_t)sy;  // (every lane the same byte to the same place: one LDS write)
          ++o;
        } else if (sy == 256) {
          break;
        } else {
          sy -= 257;
          if (sy >= 29) {
            bad = true;
            break;
          }
          // base and extra bits of the length / distance codes by arithmetic (RFC 1951 3.2.5): a table in constant memory was a
          // scalar load, two of them dependent, per match
          const uint32_t lx = sy < 8 || sy == 28 ? 0u : ((uint32_t)sy >> 2) - 1u;
          const uint32_t lb = sy < 8 ? (uint32_t)sy + 3u : (sy == 28 ? 258u : ((4u + ((uint32_t)sy & 3u)) << lx) + 3u);
          const uint32_t len = lb + (lx ? bits((int)lx) : 0u);
          const int ds = decode(dtab, IW_DBITS, dcnt, dsym);
          if (ds < 0 || ds >= 30) {
            bad = true;
            break;
          }
          const uint32_t dx = ds < 2 ? 0u : ((uint32_t)ds >> 1) - 1u;
          const uint32_t db = ds < 2 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << dx) + 1u;
          const uint32_t dist = db + (dx ? bits((int)dx) : 0u);
          if (dist > o || o + len > cap) {
            bad = true;
            break;
          }
          // The copy.  The block is one wave and a wave's LDS instructions execute in the order they were issued: a read sees the
          // ring as the writes before it left it, no barrier needed.  A distance of 64 and more: no lane reads what a lane of the
          // same instruction writes, chunks of 64 bytes in order are the byte-by-byte copy.  Shorter distances: byte j of the
          // match is byte (j mod dist) of the `dist` bytes before it, all of which exist before the copy starts (a distance of 1
          // — a run — is one byte for every lane).
          if (dist > (uint32_t)(IW_WIN - 64)) {  // from beyond the ring: those bytes are in memory (flushed: o - flushed < IW_FLUSH + 258)
            for (uint32_t j0 = 0; j0 < len; j0 += 64) {
              const uint32_t j = j0 + lane;
              // (read past this CU's vector cache, which may hold the line as it was before the last flush)
              if (j < len) win[(o + j) & (IW_WIN - 1)] = __hip_atomic_load(out + (o - dist + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          } else if (dist >= 64u || dist >= len) {
            for (uint32_t j0 = 0; j0 < len; j0 += 64) {
              const uint32_t j = j0 + lane;
              if (j < len) win[(o + j) & (IW_WIN - 1)] = win[(o - dist + j) & (IW_WIN - 1)];
            }
          } else if (dist == 1u) {
            const uint8_t v = win[(o - 1u) & (IW_WIN - 1)];
            for (uint32_t j0 = 0; j0 < len; j0 += 64) {
              const uint32_t j = j0 + lane;
              if (j < len) win[(o + j) & (IW_WIN - 1)] = v;
            }
          } else {
            uint8_t v[5];  // len <= 258: at most five chunks, all read before the first is written
#pragma unroll
            for (int q = 0; q < 5; ++q) {
              const uint32_t j = (uint32_t)q * 64u + lane;
              v[q] = j < len ? win[(o - dist + j % dist) & (IW_WIN - 1)] : (uint8_t)0;
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
              const uint32_t j = (uint32_t)q * 64u + lane;
              if (j < len) win[(o + j) & (IW_WIN - 1)] = v[q];
            }
          }
          o += len;
        }
        if (o - flushed >= (uint32_t)IW_FLUSH) flush(o);
      }
      if (bad) break;
    }
    if (last) break;
    if (overrun()) {
      bad = true;
      break;
    }
  }
  if (!bad) flush(o);
  if (!bad && o != cap) bad = true;
  if (bad && lane == 0) atomicOr(err, 1u);
}

// CRC32 and ISIZE of every member, as htslib checks them (RFC 1952).  A workgroup per member: the payload is staged in LDS with
// coalesced loads and its CRC computed a chunk per thread (crc32_block.hpp).  (Round 4: a lane per member walking its 64 KiB through
// slicing tables — 64 different cache lines per load instruction: 121 ms for the 21 GB of 45 inputs, an eighth of the device decode.)
constexpr int CRC_NT = 256;
constexpr uint32_t CRC_LDS = 65536 + 16 + 4 * CRCB_LDS_WORDS;
__global__ __launch_bounds__(CRC_NT) void bgz_crc_k(uint32_t nmem, const BgzMember* __restrict__ mem, const uint8_t* __restrict__ dst,
                                                   uint32_t* __restrict__ err) {
  extern __shared__ __align__(16) uint8_t crc_lds[];
  uint32_t* const work = (uint32_t*)(crc_lds + 65536 + 16);
  crcb_setup(work, threadIdx.x, CRC_NT);
  for (uint32_t m = blockIdx.x; m < nmem; m += gridDim.x) {
    const BgzMember M = mem[m];
    const uint8_t* p = dst + M.dst;
    const uint32_t n = M.isize;
    // (a member's payload starts at any byte: bytes up to the first 16-byte boundary, then whole vectors)
    const uint32_t head = min(n, (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u));
    if (threadIdx.x < head) crc_lds[threadIdx.x] = p[threadIdx.x];
    const uint32_t nv = (n - head) >> 4;
    const uint4* pv = (const uint4*)(p + head);
    for (uint32_t i = threadIdx.x; i < nv; i += CRC_NT) {
      const uint4 v = pv[i];
      uint8_t* q = crc_lds + head + 16 * i;  // (head bytes in: not 16-byte aligned in LDS — four dword stores)
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
      __builtin_memcpy(q, w, 16);
    }
    const uint32_t t0 = head + nv * 16;
    if (threadIdx.x < n - t0) crc_lds[t0 + threadIdx.x] = p[t0 + threadIdx.x];
    __syncthreads();
    const uint32_t c = crcb_run(work, crc_lds, n, threadIdx.x, CRC_NT);
    if (threadIdx.x == 0 && c != M.crc) atomicOr(err, 1u);
  }
}

// the inflate: a wave per member, then the CRC pass (a lane per member).  (Rounds 3 - 4 also kept a lane-per-member decoder — 64
// different decoders in lock step —: 664 ms for 7.6 GB of inputs (round 4) against 274 ms; it lives in the history.)
static int bgz_inflate_launch(tbk_ctx* ctx, uint32_t nmem, const BgzMember* d_mt, const uint8_t* d_comp, uint8_t* d_out) {
  TBK_LAUNCH(ctx, "bgz_inflate", bgz_inflate_wave_k, nmem, 64, 0, nmem, d_mt, d_comp, d_out, ctx->d_err);
  {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)bgz_crc_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CRC_LDS);
      attr = true;
    }
    TBK_LAUNCH(ctx, "bgz_crc", bgz_crc_k, std::min<uint32_t>(nmem, (uint32_t)ctx->num_cu * 2u), CRC_NT, CRC_LDS, nmem, d_mt, d_out, ctx->d_err);
  }
  return 0;
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------
extern "C" int tbk_bgzf_inflate(tbk_ctx* ctx, const uint8_t* comp, uint64_t comp_bytes, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes,
                                int mem) {
  if (!ctx || !comp || !out || !out_bytes) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  // member table from the gzip headers (host side: a few bytes per 64 KiB member)
  std::vector<BgzMember> mt;
  uint64_t off = 0, total = 0;
  while (off < comp_bytes) {
    if (off + 18 > comp_bytes || comp[off] != 0x1f || comp[off + 1] != 0x8b || comp[off + 2] != 8 || !(comp[off + 3] & 4)) return TBK_EINVAL;
    const uint32_t xlen = comp[off + 10] | (comp[off + 11] << 8);
    uint64_t p = off + 12, end = p + xlen;
    if (end > comp_bytes) return TBK_EINVAL;
    int bsize = -1;
    while (p + 4 <= end) {
      const uint32_t slen = comp[p + 2] | (comp[p + 3] << 8);
      if (p + 4 + slen > end) break;  // (a subfield that runs past XLEN is never read)
      if (comp[p] == 'B' && comp[p + 1] == 'C' && slen == 2) bsize = comp[p + 4] | (comp[p + 5] << 8);
      p += 4 + slen;
    }
    if (bsize < 0 || (uint64_t)bsize + 1 < 12ull + xlen + 8 || off + (uint64_t)bsize + 1 > comp_bytes) return TBK_EINVAL;
    const uint8_t* t = comp + off + bsize + 1 - 8;
    BgzMember m;
    m.src = off + 12 + xlen;
    m.clen = (uint32_t)((uint64_t)bsize + 1 - 8 - (12 + xlen));
    m.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    m.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    if (m.isize > 65536) return TBK_EINVAL;
    m.dst = total;
    m.file = 0;
    total += m.isize;
    if (m.isize) mt.push_back(m);
    off += (uint64_t)bsize + 1;
  }
  *out_bytes = total;
  if (total > out_cap) return TBK_E2BIG;
  if (mt.empty()) return 0;
  tbk_prof_begin_call(ctx);
  TBK_TRY(tbk_ws_reserve(ctx, comp_bytes + total + mt.size() * sizeof(BgzMember) + ((size_t)1 << 20)));
  uint8_t* d_comp = ws_alloc<uint8_t>(ctx, comp_bytes);
  BgzMember* d_mt = ws_alloc<BgzMember>(ctx, mt.size());
  uint8_t* d_out = mem == TBK_MEM_DEVICE ? out : ws_alloc<uint8_t>(ctx, total);
  if (!d_comp || !d_mt || !d_out) return TBK_ENOMEM;
  TBK_HIP(hipMemcpyAsync(d_comp, comp, comp_bytes, hipMemcpyHostToDevice, ctx->stream));
  TBK_HIP(hipMemcpyAsync(d_mt, mt.data(), mt.size() * sizeof(BgzMember), hipMemcpyHostToDevice, ctx->stream));
  TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
  TBK_TRY(bgz_inflate_launch(ctx, (uint32_t)mt.size(), d_mt, d_comp, d_out));
  if (mem != TBK_MEM_DEVICE) TBK_HIP(hipMemcpyAsync(out, d_out, total, hipMemcpyDeviceToHost, ctx->stream));
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));  // (also waits for the member table upload: `mt` dies with this frame)
  tbk_prof_end_call(ctx);
  if (eb) {
    ctx->last_error = "corrupt BGZF member (deflate stream, ISIZE or CRC32)";
    return TBK_EINVAL;
  }
  return tbk_check_launch(ctx, "bgzf_inflate");
}

// =====================================================================================================================
// BAM decode on the device: inflated streams -> record index -> SoA tile
// =====================================================================================================================
namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// one thread per file: "BAM\1", l_text, text, n_ref, references -> offset of the first alignment record
__global__ void bam_header_k(uint32_t k, const uint8_t* __restrict__ inf, const uint64_t* __restrict__ fbase, const uint64_t* __restrict__ fbytes,
                             uint64_t* __restrict__ first, int32_t* __restrict__ nref, uint32_t* __restrict__ err) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= k) return;
  const uint8_t* p = inf + fbase[f];
  const uint64_t n = fbytes[f];
  bool bad = n < 12 || p[0] != 'B' || p[1] != 'A' || p[2] != 'M' || p[3] != 1;
  uint64_t o = 0;
  int32_t nr = 0;
  if (!bad) {
    const uint64_t l_text = ld32(p + 4);
    o = 8 + l_text;
    if (o + 4 > n) {
      bad = true;
    } else {
      nr = (int32_t)ld32(p + o);
      o += 4;
      for (int32_t i = 0; i < nr && !bad; ++i) {
        if (o + 4 > n) {
          bad = true;
          break;
        }
        const uint64_t l_name = ld32(p + o);
        if (o + 8 + l_name > n) bad = true;
        o += 8 + l_name;
      }
    }
  }
  first[f] = o;
  nref[f] = nr;
  if (bad) atomicOr(err, 2u);
}

// One workgroup per file walks the record chain (block_size -> next record): the stream is staged through LDS in chunks, one
// lane follows the chain inside the chunk (an LDS read per record instead of a memory round trip), the offsets found are
// written out by the whole group.  rec[cap_off[f] + i] = offset of record i of file f in the inflated buffer.
constexpr int IDX_NT = 256;
constexpr uint32_t IDX_CH = 48 * 1024;
__global__ __launch_bounds__(IDX_NT) void bam_index_k(const uint8_t* __restrict__ inf, const uint64_t* __restrict__ fbase,
                                                      const uint64_t* __restrict__ fbytes, const uint64_t* __restrict__ first,
                                                      const uint64_t* __restrict__ cap_off, uint64_t* __restrict__ rec, uint32_t* __restrict__ cnt,
                                                      uint32_t* __restrict__ err) {
  __shared__ __align__(16) uint8_t buf[IDX_CH + 16];
  __shared__ uint32_t list[IDX_CH / 36 + 2];
  __shared__ uint32_t s_n;
  __shared__ unsigned long long s_next;
  __shared__ uint32_t s_bad;
  const uint32_t f = blockIdx.x;
  const uint8_t* base = inf + fbase[f];
  const uint64_t n = fbytes[f];
  uint64_t p = first[f];
  uint64_t total = 0;
  uint64_t* out = rec + cap_off[f];
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  while (p < n) {
    const uint64_t c0 = p & ~(uint64_t)15;  // 16-byte aligned chunk start (the buffers are 256-byte aligned and files start 16-aligned)
    const uint32_t cl = (uint32_t)((n - c0) < (uint64_t)IDX_CH ? (n - c0) : (uint64_t)IDX_CH);
    for (uint32_t i = threadIdx.x * 16; i < cl; i += IDX_NT * 16) {
      if (i + 16 <= cl) {
        *reinterpret_cast<uint4*>(buf + i) = *reinterpret_cast<const uint4*>(base + c0 + i);
      } else {
        for (uint32_t b = i; b < cl; ++b) buf[b] = base[c0 + b];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t q = (uint32_t)(p - c0), m = 0;
      bool bad = false;
      while (q + 4 <= cl) {
        const uint32_t bs = (uint32_t)buf[q] | ((uint32_t)buf[q + 1] << 8) | ((uint32_t)buf[q + 2] << 16) | ((uint32_t)buf[q + 3] << 24);
        if (bs < 32 || c0 + q + 4 + (uint64_t)bs > n) {
          bad = true;
          break;
        }
        list[m++] = q;
        q += 4 + bs;
      }
      if (!bad && q < cl && c0 + cl == n) bad = true;  // fewer than 4 bytes left at the end of the file: a truncated record
      s_n = m;
      s_next = c0 + q;
      if (bad) s_bad = 1;
    }
    __syncthreads();
    const uint32_t m = s_n;
    for (uint32_t i = threadIdx.x; i < m; i += IDX_NT) out[total + i] = fbase[f] + c0 + list[i];
    total += m;
    const uint64_t np = s_next;
    const bool bad = s_bad != 0;
    __syncthreads();
    if (bad || np <= p) {  // (no progress cannot happen: a chunk is longer than a block_size field plus the alignment slack)
      if (threadIdx.x == 0) atomicOr(err, 4u);
      break;
    }
    p = np;
  }
  if (threadIdx.x == 0) cnt[f] = (uint32_t)(total < 0xFFFFFFFFull ? total : 0xFFFFFFFFull);
}

// The record index without the chain.  htslib starts a new BGZF block when the next record does not fit into the current one
// (bgzf_flush_try in bam_write1), so in a BAM file written through it every block begins with a record: a lane per member walks
// its own 64 KiB — ~ 270 dependent loads instead of a file's million — and the walks are the file's chain IF each ends exactly where
// its member ends.  A member whose walk does not (a writer that cuts blocks at a fixed size, a record longer than a block,
// a corrupt length) flags its file, and flagged files take the chain kernel above, which also decides what is an error.
//   bam_imem_count_k: records per member;  bam_imem_scan_k: per file, the members' counts -> offsets, the file's total;
//   bam_imem_emit_k: the walks again, writing rec[cap_off[f] + offset + i]
__device__ __forceinline__ uint32_t bam_imem_walk(const uint8_t* __restrict__ inf, const BgzMember& M, uint64_t s0, uint64_t fend, uint64_t* out, bool* aligned) {
  const uint64_t end = M.dst + M.isize;
  uint64_t p = M.dst > s0 ? M.dst : s0;
  uint32_t c = 0;
  bool ok = true;
  if (end > s0) {
    while (p < end) {
      if (p + 4 > fend) {
        ok = false;
        break;
      }
      const uint32_t bs = ld32(inf + p);
      if (bs < 32 || p + 4 + (uint64_t)bs > fend) {
        ok = false;
        break;
      }
      if (out) out[c] = p;
      ++c;
      p += 4 + (uint64_t)bs;
    }
    ok = ok && p == end;
  }
  *aligned = ok;
  return c;
}
__global__ void bam_imem_count_k(uint32_t nmem, const BgzMember* __restrict__ mem, const uint8_t* __restrict__ inf, const uint64_t* __restrict__ fbase,
                                 const uint64_t* __restrict__ fbytes, const uint64_t* __restrict__ first, uint32_t* __restrict__ mcnt,
                                 uint32_t* __restrict__ fflag) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nmem) return;
  const BgzMember M = mem[m];
  bool aligned;
  mcnt[m] = bam_imem_walk(inf, M, fbase[M.file] + first[M.file], fbase[M.file] + fbytes[M.file], nullptr, &aligned);
  if (!aligned) fflag[M.file] = 1u;
}
__global__ __launch_bounds__(256) void bam_imem_scan_k(const uint32_t* __restrict__ mfirst /* [k + 1] */, uint32_t* __restrict__ mcnt /* in: counts, out: offsets */,
                                                       uint32_t* __restrict__ cnt) {
  __shared__ uint32_t sm[8];
  const uint32_t f = blockIdx.x, m0 = mfirst[f], m1 = mfirst[f + 1];
  uint32_t carry = 0;
  for (uint32_t b = m0; b < m1; b += 256) {
    const uint32_t i = b + threadIdx.x;
    const uint32_t v = i < m1 ? mcnt[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_sum<uint32_t, 256>(v, sm, &tot);
    if (i < m1) mcnt[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) cnt[f] = carry;
}
__global__ void bam_imem_emit_k(uint32_t nmem, const BgzMember* __restrict__ mem, const uint8_t* __restrict__ inf, const uint64_t* __restrict__ fbase,
                                const uint64_t* __restrict__ fbytes, const uint64_t* __restrict__ first, const uint64_t* __restrict__ cap_off,
                                const uint32_t* __restrict__ moff, uint64_t* __restrict__ rec) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nmem) return;
  const BgzMember M = mem[m];
  bool aligned;
  (void)bam_imem_walk(inf, M, fbase[M.file] + first[M.file], fbase[M.file] + fbytes[M.file], rec + cap_off[M.file] + moff[m], &aligned);
}

struct BamSoA {
  int32_t *tid, *pos, *nh;
  uint16_t* flag;
  uint8_t *mapq, *strand;
  uint32_t* ncig;      // per record CIGAR operation count -> cig_off by a scan
  double* yc;
  int64_t *yx, *yd;
  uint32_t *nmd, *nqn; // MD / name byte counts (optional)
  uint8_t* md_has;
};

// one thread per record: validation as the host loader's (bam.cpp index_records), core fields, one aux scan
__global__ void bam_fields_k(uint32_t n, const uint8_t* __restrict__ inf, const uint64_t* __restrict__ rec, const uint32_t* __restrict__ file_off,
                             uint32_t k, const uint8_t* __restrict__ tbm, const int32_t* __restrict__ nref, BamSoA S, int want_md, int want_qn,
                             uint32_t* __restrict__ err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = k;  // file of record i
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (file_off[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  const uint32_t f = lo;
  const uint8_t* r = inf + rec[i];
  const uint32_t bs = ld32(r);
  r += 4;
  const int32_t tid = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4);
  const uint32_t l_read_name = r[8], mapq = r[9], n_cigar = ld16(r + 12), flag = ld16(r + 14);
  const int32_t l_seq = (int32_t)ld32(r + 16), mtid = (int32_t)ld32(r + 20);
  const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + (l_seq < 0 ? 0ull : ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq);
  const bool name_ok = l_read_name >= 1 && 32 + l_read_name <= bs && r[32 + l_read_name - 1] == 0;
  const int32_t nt = nref[f];
  if (l_seq < 0 || need > bs || !name_ok || tid < -1 || tid >= nt || mtid < -1 || mtid >= nt) {
    atomicOr(err, 8u);
    S.ncig[i] = 0;
    if (want_md) S.nmd[i] = 0;
    if (want_qn) S.nqn[i] = 0;
    return;
  }
  S.tid[i] = tid;
  S.pos[i] = pos;
  S.flag[i] = (uint16_t)flag;
  S.mapq[i] = (uint8_t)mapq;
  S.ncig[i] = n_cigar;
  if (want_qn) S.nqn[i] = l_read_name - 1;
  // aux scan; bam_aux_get semantics = first occurrence of each tag
  const uint8_t* a = r + need;
  const uint8_t* e = r + bs;
  const bool tb = tbm[f] != 0;
  char xs = 0, ts = 0;
  int32_t nh = TBK_NH_ABSENT;
  double yc = 0.0;
  int64_t yx = 1, yd = 0;
  uint32_t nmd = 0, seen = 0;
  uint8_t md_has = 0;
  auto aux_i = [](const uint8_t* s) -> int64_t {  // bam_aux2i
    switch (*s) {
      case 'c': return (int8_t)s[1];
      case 'C': return s[1];
      case 's': return (int16_t)ld16(s + 1);
      case 'S': return ld16(s + 1);
      case 'i': return (int32_t)ld32(s + 1);
      case 'I': return ld32(s + 1);
    }
    return 0;
  };
  while (a + 3 <= e) {
    const uint8_t* s = a + 2;
    const uint8_t ty = *s;
    uint64_t sz = 0;  // tag + type + value
    switch (ty) {
      case 'A': case 'c': case 'C': sz = 4; break;
      case 's': case 'S': sz = 5; break;
      case 'i': case 'I': case 'f': sz = 7; break;
      case 'd': sz = 11; break;
      case 'Z': case 'H': {
        const uint8_t* z = s + 1;
        while (z < e && *z) ++z;
        sz = z < e ? (uint64_t)(z - a) + 1 : 0;
        break;
      }
      case 'B': {
        if (a + 8 <= e) {
          const uint8_t st = s[1];
          const uint32_t cntb = ld32(s + 2);
          const uint32_t es = (st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : (st == 'i' || st == 'I' || st == 'f') ? 4u : 0u;
          sz = es ? 8ull + (uint64_t)cntb * es : 0;
        }
        break;
      }
    }
    if (!sz || a + sz > e) break;
    if (a[0] == 'N' && a[1] == 'H' && !(seen & 1)) {
      seen |= 1;
      nh = (int32_t)aux_i(s);
    } else if (a[0] == 'X' && a[1] == 'S' && !(seen & 2)) {
      seen |= 2;
      xs = (ty == 'A' || ty == 'Z') ? (char)s[1] : 0;
    } else if (a[0] == 't' && a[1] == 's' && !(seen & 4)) {
      seen |= 4;
      ts = (ty == 'A' || ty == 'Z') ? (char)s[1] : 0;
    } else if (tb && a[0] == 'Y' && a[1] == 'C' && !(seen & 8)) {
      seen |= 8;
      if (ty == 'f') {
        yc = (double)__uint_as_float(ld32(s + 1));
      } else if (ty == 'd') {
        yc = __longlong_as_double((long long)((uint64_t)ld32(s + 1) | ((uint64_t)ld32(s + 5) << 32)));
      } else {
        yc = (double)aux_i(s);
      }
    } else if (tb && a[0] == 'Y' && a[1] == 'X' && !(seen & 16)) {
      seen |= 16;
      yx = aux_i(s);
    } else if (tb && a[0] == 'Y' && a[1] == 'D' && !(seen & 32)) {
      seen |= 32;
      yd = aux_i(s);
    } else if (want_md && a[0] == 'M' && a[1] == 'D' && !(seen & 64)) {
      seen |= 64;
      if (ty == 'Z') {
        nmd = (uint32_t)(sz - 4);
        md_has = 1;
      }
    }
    a += sz;
  }
  S.nh[i] = nh;
  char c = xs;  // GSamRecord::spliceStrand (GSam.cpp:464-475)
  if (c == 0 && (ts == '+' || ts == '-')) c = (flag & 0x10) ? (ts == '+' ? '-' : '+') : ts;
  S.strand[i] = (uint8_t)((c == '+' || c == '-') ? c : '.');
  if (S.yc) {
    S.yc[i] = yc;
    S.yx[i] = yx;
    S.yd[i] = yd;
  }
  if (want_md) {
    S.nmd[i] = nmd;
    S.md_has[i] = md_has;
  }
}

__global__ void bam_fill_k(uint32_t n, const uint8_t* __restrict__ inf, const uint64_t* __restrict__ rec, const uint32_t* __restrict__ cig_off,
                           uint32_t* __restrict__ cig, const uint16_t* __restrict__ flag, const uint32_t* __restrict__ md_off, uint8_t* __restrict__ md,
                           const uint32_t* __restrict__ qn_off, uint8_t* __restrict__ qn, uint64_t* __restrict__ qh) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = inf + rec[i] + 4;
  const uint32_t l_read_name = r[8];
  const uint32_t c0 = cig_off[i], nc = cig_off[i + 1] - c0;
  const uint8_t* cp = r + 32 + l_read_name;
  for (uint32_t q = 0; q < nc; ++q) cig[c0 + q] = ld32(cp + 4 * q);
  if (qn_off) {
    const uint32_t o = qn_off[i], l = qn_off[i + 1] - o;
    uint64_t h = 0xCBF29CE484222325ull;  // FNV-1a over the name bytes and pairOrder + 1 (tmerge.cpp tbh_qname_hash)
    for (uint32_t q = 0; q < l; ++q) {
      const uint8_t b = r[32 + q];
      qn[o + q] = b;
      h = (h ^ b) * 0x100000001B3ull;
    }
    const uint32_t fl = flag[i];
    const uint32_t po = (fl & 0x40) ? 1u : ((fl & 0x80) ? 2u : 0u);
    h = (h ^ (po + 1)) * 0x100000001B3ull;
    qh[i] = h;
  }
  if (md_off && md_off[i + 1] > md_off[i]) {  // copy the MD:Z payload (found again: first MD tag)
    const uint32_t bs = ld32(r - 4);
    const int32_t l_seq = (int32_t)ld32(r + 16);
    const uint8_t* a = cp + 4ull * nc + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    const uint8_t* e = r + bs;
    while (a + 3 <= e) {
      const uint8_t ty = a[2];
      uint64_t sz = 0;
      switch (ty) {
        case 'A': case 'c': case 'C': sz = 4; break;
        case 's': case 'S': sz = 5; break;
        case 'i': case 'I': case 'f': sz = 7; break;
        case 'd': sz = 11; break;
        case 'Z': case 'H': {
          const uint8_t* z = a + 3;
          while (z < e && *z) ++z;
          sz = z < e ? (uint64_t)(z - a) + 1 : 0;
          break;
        }
        case 'B': {
          if (a + 8 <= e) {
            const uint8_t st = a[3];
            const uint32_t cntb = ld32(a + 4);
            const uint32_t es = (st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : (st == 'i' || st == 'I' || st == 'f') ? 4u : 0u;
            sz = es ? 8ull + (uint64_t)cntb * es : 0;
          }
          break;
        }
      }
      if (!sz || a + sz > e) break;
      if (a[0] == 'M' && a[1] == 'D') {
        if (ty == 'Z') {
          const uint32_t o = md_off[i];
          for (uint32_t q = 0; q + 4 < sz; ++q) md[o + q] = a[3 + q];
        }
        break;
      }
      a += sz;
    }
  }
}

__global__ void bam_compact_k(uint32_t k, const uint64_t* __restrict__ cap_off, const uint32_t* __restrict__ file_off, const uint64_t* __restrict__ rec_in,
                              uint64_t* __restrict__ rec_out) {
  const uint32_t f = blockIdx.y;
  const uint32_t n = file_off[f + 1] - file_off[f];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) rec_out[file_off[f] + i] = rec_in[cap_off[f] + i];
}

// gather: sizes, then bytes
__global__ void bam_recsize_k(uint32_t n, const uint32_t* __restrict__ idx, const uint8_t* __restrict__ inf, const uint64_t* __restrict__ rec,
                              uint32_t nrec, uint32_t* __restrict__ sz, uint32_t* __restrict__ err) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint32_t i = idx[g];
  if (i >= nrec) {
    atomicOr(err, 16u);
    sz[g] = 0;
    return;
  }
  sz[g] = 4 + ld32(inf + rec[i]);
}
__global__ void bam_reccopy_k(uint32_t n, const uint32_t* __restrict__ idx, const uint8_t* __restrict__ inf, const uint64_t* __restrict__ rec,
                              const uint64_t* __restrict__ off, uint8_t* __restrict__ out) {
  // one wave per record: 64 consecutive bytes per step
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (g >= n) return;
  const uint8_t* s = inf + rec[idx[g]];
  const uint64_t o = off[g];
  const uint32_t len = (uint32_t)(off[g + 1] - o);
  for (uint32_t b = lane_id(); b < len; b += 64) out[o + b] = s[b];
}

struct BamDev {
  uint8_t* inf = nullptr;      // inflated streams of all files, each starting 256-byte aligned
  uint64_t* rec = nullptr;     // [n] offset of every record (its block_size field) in `inf`
  uint32_t n = 0;
  std::vector<void*> owned;    // device allocations of the decoded tile
};

}  // namespace

static void bamdev_free(tbk_ctx* ctx) {
  BamDev* B = (BamDev*)ctx->bam_dev;
  if (!B) return;
  for (void* p : B->owned) (void)hipFree(p);
  delete B;
  ctx->bam_dev = nullptr;
}

// ---- host -> device through a pinned ring -----------------------------------------------------------------------------------
// The compressed files come in the caller's pageable memory (or a mapping of the page cache).  A plain hipMemcpy from there is staged by
// the runtime on one thread (~ 10 GB/s); here a few threads copy 16 MB chunks into page-locked slots of the context and queue each
// slot's DMA on an upload stream of its own, so the link runs near its rate and — the caller uploads group after group — the inflate
// of one group of files runs while the next group is on its way.
namespace {
struct Stager {
  static constexpr int T = 3, PER = 2;
  static constexpr size_t S = (size_t)16 << 20;
  uint8_t* pin[T * PER] = {};
  hipEvent_t ev[T * PER] = {};
  bool used[T * PER] = {};
  hipStream_t up = nullptr;
  hipEvent_t done = nullptr;
};
struct StageChunk {
  uint8_t* dst;
  const uint8_t* src;
  size_t len;
};
}  // namespace

void tbk_stager_free(tbk_ctx* ctx) {
  Stager* S = (Stager*)ctx->stager;
  if (!S) return;
  for (int i = 0; i < Stager::T * Stager::PER; ++i) {
    if (S->pin[i]) tbk_host_free(S->pin[i]);
    if (S->ev[i]) (void)hipEventDestroy(S->ev[i]);
  }
  if (S->done) (void)hipEventDestroy(S->done);
  if (S->up) (void)hipStreamDestroy(S->up);
  delete S;
  ctx->stager = nullptr;
}

static Stager* stager_get(tbk_ctx* ctx) {
  if (ctx->stager) return (Stager*)ctx->stager;
  Stager* S = new Stager();
  bool ok = hipStreamCreateWithFlags(&S->up, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&S->done, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < Stager::T * Stager::PER; ++i)
    ok = tbk_host_alloc(Stager::S, (void**)&S->pin[i]) == 0 && hipEventCreateWithFlags(&S->ev[i], hipEventDisableTiming) == hipSuccess;  // (huge pages + registration: tbk_api.hip)
  ctx->stager = S;
  if (!ok) {
    (void)hipGetLastError();
    tbk_stager_free(ctx);
    return nullptr;
  }
  return S;
}

// queues the copies of `chunks` on the upload stream and makes ctx->stream wait for them; returns when the last chunk has left the
// caller's memory for a pinned slot (the DMA may still be running).  Without a stager (no pinned memory to be had): plain copies.
static int staged_upload(tbk_ctx* ctx, const std::vector<StageChunk>& chunks) {
  if (chunks.empty()) return 0;
  Stager* S = stager_get(ctx);
  if (!S) {
    for (const StageChunk& c : chunks) TBK_HIP(hipMemcpyAsync(c.dst, c.src, c.len, hipMemcpyHostToDevice, ctx->stream));
    return 0;
  }
  std::atomic<size_t> next{0};
  std::atomic<int> bad{0};
  const int device = ctx->device;
  auto work = [&](int t) {
    (void)hipSetDevice(device);
    int turn = 0;
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= chunks.size() || bad.load()) break;
      const int s = t * Stager::PER + (turn++ % Stager::PER);
      if (S->used[s] && hipEventSynchronize(S->ev[s]) != hipSuccess) bad.store(1);
      memcpy(S->pin[s], chunks[i].src, chunks[i].len);
      if (hipMemcpyAsync(chunks[i].dst, S->pin[s], chunks[i].len, hipMemcpyHostToDevice, S->up) != hipSuccess || hipEventRecord(S->ev[s], S->up) != hipSuccess) bad.store(1);
      S->used[s] = true;
    }
  };
  const int nt = (int)std::min<size_t>(Stager::T, chunks.size());
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  if (bad.load()) {
    ctx->last_error = "staged upload failed";
    (void)hipGetLastError();
    return TBK_EHIP;
  }
  TBK_HIP(hipEventRecord(S->done, S->up));
  TBK_HIP(hipStreamWaitEvent(ctx->stream, S->done, 0));
  return 0;
}

bool tbk_bam_dev_records(tbk_ctx* ctx, const uint8_t** inf, const uint64_t** rec, uint32_t* n) {
  BamDev* B = (BamDev*)ctx->bam_dev;
  if (!B || !B->inf || !B->rec) return false;
  *inf = B->inf, *rec = B->rec, *n = B->n;
  return true;
}

extern "C" void tbk_bam_release(tbk_ctx* ctx) {
  if (!ctx) return;
  (void)tbk_collapse_finish_yd(ctx);  // (a deferred YD stage may still read the decoded tile)
  bamdev_free(ctx);
}

template <class T>
static T* bd_alloc(BamDev* B, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
  B->owned.push_back(p);
  return (T*)p;
}

extern "C" int tbk_bam_decode(tbk_ctx* ctx, uint32_t n_files, const uint8_t* const* comp, const uint64_t* comp_bytes, const uint8_t* tbmerged,
                              int want_md, int want_names, tbk_soa_in* tile, uint32_t* file_off_out) {
  if (!ctx || !comp || !comp_bytes || !tile || !file_off_out || n_files == 0 || n_files > 65535) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  TBK_TRY(tbk_collapse_finish_yd(ctx));  // (a deferred YD stage may still read the tile of the previous decode)
  bamdev_free(ctx);
  tbk_prof_begin_call(ctx);
  const uint32_t k = n_files;
  // ---- member table of every file (the files are walked side by side: a mapping of the page cache faults its pages in as the walk
  // touches them); inflated streams laid out file after file, 256-byte aligned ----
  std::vector<BgzMember> mt;
  std::vector<uint64_t> cbase(k + 1, 0), fbase(k, 0), fbytes(k, 0);
  uint64_t ctot = 0, itot = 0;
  {
    std::vector<std::vector<BgzMember>> fm(k);
    std::vector<uint64_t> fi(k, 0);
    std::atomic<uint32_t> nf{0};
    std::atomic<int> bad{0};
    auto walk = [&]() {
      for (;;) {
        const uint32_t f = nf.fetch_add(1);
        if (f >= k || bad.load()) break;
        const uint8_t* c = comp[f];
        const uint64_t cb = comp_bytes[f];
        if (!c) {
          bad.store(1);
          break;
        }
        uint64_t off = 0, isz = 0;
        while (off < cb) {
          if (off + 18 > cb || c[off] != 0x1f || c[off + 1] != 0x8b || c[off + 2] != 8 || !(c[off + 3] & 4)) break;
          const uint32_t xlen = c[off + 10] | (c[off + 11] << 8);
          uint64_t p = off + 12, end = p + xlen;
          if (end > cb) break;
          int bsize = -1;
          while (p + 4 <= end) {
            const uint32_t slen = c[p + 2] | (c[p + 3] << 8);
            if (p + 4 + slen > end) break;  // (a subfield that runs past XLEN is never read)
            if (c[p] == 'B' && c[p + 1] == 'C' && slen == 2) bsize = c[p + 4] | (c[p + 5] << 8);
            p += 4 + slen;
          }
          if (bsize < 0 || (uint64_t)bsize + 1 < 12ull + xlen + 8 || off + (uint64_t)bsize + 1 > cb) break;
          const uint8_t* t = c + off + bsize + 1 - 8;
          BgzMember m;
          m.src = off + 12 + xlen;  // (inside the file: the bases are added below)
          m.clen = (uint32_t)((uint64_t)bsize + 1 - 8 - (12 + xlen));
          m.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
          m.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
          if (m.isize > 65536) break;
          m.dst = isz;
          m.file = f;
          isz += m.isize;
          if (m.isize) fm[f].push_back(m);
          off += (uint64_t)bsize + 1;
        }
        if (off != cb) bad.store(1);  // (anything htslib would reject)
        fi[f] = isz;
      }
    };
    const int nw = (int)std::min<uint32_t>(k, 8);
    std::vector<std::thread> th;
    for (int t = 1; t < nw; ++t) th.emplace_back(walk);
    walk();
    for (auto& x : th) x.join();
    if (bad.load()) return TBK_EINVAL;
    size_t nm = 0;
    for (uint32_t f = 0; f < k; ++f) nm += fm[f].size();
    mt.reserve(nm);
    for (uint32_t f = 0; f < k; ++f) {
      cbase[f] = ctot;
      itot = (itot + 255) & ~(uint64_t)255;
      fbase[f] = itot;
      for (BgzMember m : fm[f]) {
        m.src += ctot;
        m.dst += itot;
        mt.push_back(m);
      }
      itot += fi[f];
      fbytes[f] = fi[f];
      ctot += comp_bytes[f];
    }
  }
  cbase[k] = ctot;
  if (mt.empty()) {
    memset(tile, 0, sizeof(*tile));
    for (uint32_t f = 0; f <= k; ++f) file_off_out[f] = 0;
    return TBK_EINVAL;  // no BAM header anywhere
  }
  BamDev* B = new BamDev();
  ctx->bam_dev = B;
  B->inf = bd_alloc<uint8_t>(B, itot + 64);
  // work memory (arena): compressed bytes, member table, per-file tables, the over-sized record list
  std::vector<uint64_t> cap_off(k + 1, 0);
  for (uint32_t f = 0; f < k; ++f) cap_off[f + 1] = cap_off[f] + fbytes[f] / 36 + 1;
  TBK_TRY(tbk_ws_reserve(ctx, ctot + mt.size() * sizeof(BgzMember) + cap_off[k] * 8 + (size_t)k * 64 + ((size_t)4 << 20)));
  uint8_t* d_comp = ws_alloc<uint8_t>(ctx, ctot);
  BgzMember* d_mt = ws_alloc<BgzMember>(ctx, mt.size());
  uint64_t* d_tab = ws_alloc<uint64_t>(ctx, (size_t)4 * k + 4);  // fbase, fbytes, first, cap_off
  int32_t* d_nref = ws_alloc<int32_t>(ctx, k);
  uint32_t* d_cnt = ws_alloc<uint32_t>(ctx, k);
  uint64_t* d_rec0 = ws_alloc<uint64_t>(ctx, cap_off[k]);
  uint8_t* d_tbm = ws_alloc<uint8_t>(ctx, k);
  if (!B->inf || !d_comp || !d_mt || !d_tab || !d_rec0 || !d_tbm) return TBK_ENOMEM;
  TBK_HIP(hipMemcpyAsync(d_mt, mt.data(), mt.size() * sizeof(BgzMember), hipMemcpyHostToDevice, ctx->stream));
  TBK_HIP(hipMemcpyAsync(d_tab, fbase.data(), k * 8, hipMemcpyHostToDevice, ctx->stream));
  TBK_HIP(hipMemcpyAsync(d_tab + k, fbytes.data(), k * 8, hipMemcpyHostToDevice, ctx->stream));
  TBK_HIP(hipMemcpyAsync(d_tab + 3 * k, cap_off.data(), k * 8, hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint8_t> tb(k, 0);
  if (tbmerged) memcpy(tb.data(), tbmerged, k);
  TBK_HIP(hipMemcpyAsync(d_tbm, tb.data(), k, hipMemcpyHostToDevice, ctx->stream));
  TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
  // the files go up in groups of ~ 192 MB through the pinned ring, and a group is inflated while the next one is on its way
  {
    size_t m0 = 0;
    for (uint32_t f0 = 0; f0 < k;) {
      std::vector<StageChunk> chunks;
      uint64_t gbytes = 0;
      uint32_t f1 = f0;
      while (f1 < k && (f1 == f0 || gbytes + comp_bytes[f1] <= ((uint64_t)192 << 20))) {
        for (uint64_t o = 0; o < comp_bytes[f1]; o += Stager::S)
          chunks.push_back(StageChunk{d_comp + cbase[f1] + o, comp[f1] + o, (size_t)std::min<uint64_t>(Stager::S, comp_bytes[f1] - o)});
        gbytes += comp_bytes[f1];
        ++f1;
      }
      TBK_TRY(staged_upload(ctx, chunks));
      size_t m1 = m0;
      while (m1 < mt.size() && mt[m1].file < f1) ++m1;
      if (m1 > m0) TBK_TRY(bgz_inflate_launch(ctx, (uint32_t)(m1 - m0), d_mt + m0, d_comp, B->inf));
      m0 = m1;
      f0 = f1;
    }
  }
  TBK_LAUNCH(ctx, "bam_header", bam_header_k, cdiv(k, 64), 64, 0, k, B->inf, d_tab, d_tab + k, d_tab + 2 * k, d_nref, ctx->d_err);
  // the record index: a lane per member where every member begins with a record (htslib's writers), the chain per file otherwise
  std::vector<uint32_t> cnt(k, 0);
  uint32_t eb = 0;
  bool chain = ctx->dbg.index_chain;  // (test hook: the chain kernel whatever the files look like)
  if (!chain) {
    std::vector<uint32_t> mfirst(k + 1, 0);
    for (const BgzMember& m : mt) mfirst[m.file + 1]++;
    for (uint32_t f = 0; f < k; ++f) mfirst[f + 1] += mfirst[f];
    uint32_t* d_mfirst = ws_alloc<uint32_t>(ctx, k + 1);
    uint32_t* d_mcnt = ws_alloc<uint32_t>(ctx, mt.size());
    uint32_t* d_fflag = ws_alloc<uint32_t>(ctx, k);
    if (!d_mfirst || !d_mcnt || !d_fflag) return TBK_ENOMEM;
    TBK_HIP(hipMemcpyAsync(d_mfirst, mfirst.data(), (k + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    TBK_HIP(hipMemsetAsync(d_fflag, 0, k * 4, ctx->stream));
    TBK_LAUNCH(ctx, "bam_index", bam_imem_count_k, cdiv((uint32_t)mt.size(), 64u), 64, 0, (uint32_t)mt.size(), d_mt, B->inf, d_tab, d_tab + k, d_tab + 2 * k, d_mcnt, d_fflag);
    TBK_LAUNCH(ctx, "bam_index", bam_imem_scan_k, k, 256, 0, d_mfirst, d_mcnt, d_cnt);
    TBK_LAUNCH(ctx, "bam_index", bam_imem_emit_k, cdiv((uint32_t)mt.size(), 64u), 64, 0, (uint32_t)mt.size(), d_mt, B->inf, d_tab, d_tab + k, d_tab + 2 * k, d_tab + 3 * k, d_mcnt,
               d_rec0);
    std::vector<uint32_t> fflag(k, 0);
    TBK_HIP(hipMemcpyAsync(fflag.data(), d_fflag, k * 4, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipMemcpyAsync(cnt.data(), d_cnt, k * 4, hipMemcpyDeviceToHost, ctx->stream));
    TBK_TRY(tbk_sync_err(ctx, &eb));
    for (uint32_t f = 0; f < k; ++f) chain = chain || fflag[f] != 0;  // (one file that needs the chain: every file takes it — the kernel is a block per file)
  }
  if (chain && !eb) {
    TBK_LAUNCH(ctx, "bam_index_chain", bam_index_k, k, IDX_NT, 0, B->inf, d_tab, d_tab + k, d_tab + 2 * k, d_tab + 3 * k, d_rec0, d_cnt, ctx->d_err);
    TBK_HIP(hipMemcpyAsync(cnt.data(), d_cnt, k * 4, hipMemcpyDeviceToHost, ctx->stream));
    TBK_TRY(tbk_sync_err(ctx, &eb));
  }
  if (eb) {
    ctx->last_error = (eb & 1u) ? "corrupt BGZF member (deflate stream, ISIZE or CRC32)" : (eb & 2u) ? "not a BAM stream / truncated header" : "corrupt BAM record chain";
    bamdev_free(ctx);
    return TBK_EINVAL;
  }
  uint64_t ntot = 0;
  file_off_out[0] = 0;
  for (uint32_t f = 0; f < k; ++f) {
    ntot += cnt[f];
    if (ntot >= (1ull << 32)) {
      bamdev_free(ctx);
      return TBK_E2BIG;
    }
    file_off_out[f + 1] = (uint32_t)ntot;
  }
  const uint32_t n = (uint32_t)ntot;
  B->n = n;
  B->rec = bd_alloc<uint64_t>(B, n);
  uint32_t* d_fo = ws_alloc<uint32_t>(ctx, k + 1);
  if (!B->rec || !d_fo) return TBK_ENOMEM;
  TBK_HIP(hipMemcpyAsync(d_fo, file_off_out, (k + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n) TBK_LAUNCH(ctx, "bam_compact", bam_compact_k, dim3(64, k), 256, 0, k, d_tab + 3 * k, d_fo, d_rec0, B->rec);
  // ---- SoA ----
  bool any_tb = false;
  for (uint32_t f = 0; f < k; ++f) any_tb |= tb[f] != 0;
  BamSoA S{};
  S.tid = bd_alloc<int32_t>(B, n);
  S.pos = bd_alloc<int32_t>(B, n);
  S.nh = bd_alloc<int32_t>(B, n);
  S.flag = bd_alloc<uint16_t>(B, n);
  S.mapq = bd_alloc<uint8_t>(B, n);
  S.strand = bd_alloc<uint8_t>(B, n);
  S.ncig = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
  uint32_t* cig_off = bd_alloc<uint32_t>(B, (size_t)n + 1);
  if (any_tb) {
    S.yc = bd_alloc<double>(B, n);
    S.yx = bd_alloc<int64_t>(B, n);
    S.yd = bd_alloc<int64_t>(B, n);
  }
  uint32_t *md_off = nullptr, *qn_off = nullptr;
  if (want_md) {
    S.nmd = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
    S.md_has = bd_alloc<uint8_t>(B, n);
    md_off = bd_alloc<uint32_t>(B, (size_t)n + 1);
  }
  if (want_names) {
    S.nqn = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
    qn_off = bd_alloc<uint32_t>(B, (size_t)n + 1);
  }
  if (!S.strand || !S.ncig || !cig_off || (any_tb && !S.yd) || (want_md && !md_off) || (want_names && !qn_off)) return TBK_ENOMEM;
  uint64_t* sc = ctx->d_scalars;
  uint64_t ncig = 0, nmd = 0, nqn = 0;
  uint32_t* cig = nullptr;
  uint8_t *md = nullptr, *qn = nullptr;
  uint64_t* qh = nullptr;
  if (n) {
    TBK_LAUNCH(ctx, "bam_fields", bam_fields_k, cdiv(n, 256), 256, 0, n, B->inf, B->rec, d_fo, k, d_tbm, d_nref, S, want_md, want_names, ctx->d_err);
    TBK_HIP(hipMemsetAsync(S.ncig + n, 0, 4, ctx->stream));
    TBK_TRY(tbk_exscan_u32(ctx, S.ncig, cig_off, n + 1, sc + 1));
    if (want_md) {
      TBK_HIP(hipMemsetAsync(S.nmd + n, 0, 4, ctx->stream));
      TBK_TRY(tbk_exscan_u32(ctx, S.nmd, md_off, n + 1, sc + 2));
    }
    if (want_names) {
      TBK_HIP(hipMemsetAsync(S.nqn + n, 0, 4, ctx->stream));
      TBK_TRY(tbk_exscan_u32(ctx, S.nqn, qn_off, n + 1, sc + 3));
    }
    TBK_TRY(tbk_sync_err(ctx, &eb));
    if (eb) {
      ctx->last_error = "malformed BAM record (field lengths / reference id outside the record / header)";
      bamdev_free(ctx);
      return TBK_EINVAL;
    }
    ncig = ctx->h_scalars[1];
    nmd = want_md ? ctx->h_scalars[2] : 0;
    nqn = want_names ? ctx->h_scalars[3] : 0;
    if (ncig >= (1ull << 32) || nmd >= (1ull << 32) || nqn >= (1ull << 32)) {
      bamdev_free(ctx);
      return TBK_E2BIG;
    }
    cig = bd_alloc<uint32_t>(B, ncig);
    if (want_md) md = bd_alloc<uint8_t>(B, nmd);
    if (want_names) {
      qn = bd_alloc<uint8_t>(B, nqn);
      qh = bd_alloc<uint64_t>(B, n);
    }
    if (!cig || (want_md && !md) || (want_names && !qh)) return TBK_ENOMEM;
    TBK_LAUNCH(ctx, "bam_fill", bam_fill_k, cdiv(n, 256), 256, 0, n, B->inf, B->rec, cig_off, cig, S.flag, md_off, md, qn_off, qn, qh);
    TBK_HIP(hipStreamSynchronize(ctx->stream));
  }
  memset(tile, 0, sizeof(*tile));
  tile->mem = TBK_MEM_DEVICE;
  tile->n_files = k;
  tile->n_records = n;
  tile->n_cigar_ops = (uint32_t)ncig;
  tile->file_off = file_off_out;
  tile->tbmerged = tbmerged;
  tile->tid = S.tid;
  tile->pos = S.pos;
  tile->flag = S.flag;
  tile->mapq = S.mapq;
  tile->strand = S.strand;
  tile->nh = S.nh;
  tile->cig_off = cig_off;
  tile->cig = cig;
  tile->yc_in = S.yc;
  tile->yx_in = S.yx;
  tile->yd_in = S.yd;
  tile->md_off = md_off;
  tile->md = md;
  tile->md_has = S.md_has;
  tile->qname_hash = qh;
  tile->qname_off = qn_off;
  tile->qname = qn;
  tbk_prof_end_call(ctx);
  // (the arena for the collapse of this tile is the caller's to reserve — tbk_reserve_tile —: it knows whether the tile will be joined
  // with a host part first, and an allocation of gigabytes is now and then 0.1 s of the driver's time: one, not two)
  return tbk_check_launch(ctx, "bam_decode");
}

namespace {
__global__ void join_cig_off_k(uint32_t nb, const uint32_t* __restrict__ src /* [nb + 1], starts at 0 */, uint32_t base, uint32_t* __restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= nb) dst[i] = src[i] + base;
}
}  // namespace

extern "C" int tbk_reserve_tile(tbk_ctx* ctx, uint64_t n_records, uint64_t n_cigar_ops) {
  if (!ctx) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  // (a third more than the window path and its YD stage need: the deferred YD stage borrows its range of this arena only while a
  // quarter of the arena stays free behind it, tbk_api.hip — otherwise it allocates an arena of its own inside the call)
  return tbk_ws_presize(ctx, ((size_t)n_records * (84 + 40) + (size_t)n_cigar_ops * 8) / 3 * 4 + ((size_t)16 << 20));
}

// one device tile: the files decoded on this context, then the files a host decoder took (include/tbk.h)
extern "C" int tbk_tile_join(tbk_ctx* ctx, const tbk_soa_in* a, const tbk_soa_in* b, tbk_soa_in* out, uint32_t* file_off_out, uint8_t* tbmerged_out) {
  if (!ctx || !ctx->bam_dev || !a || !b || !out || !file_off_out || !tbmerged_out) return TBK_EINVAL;
  if (a->mem != TBK_MEM_DEVICE || b->mem != TBK_MEM_HOST) return TBK_EINVAL;
  if (a->yc_in || a->yx_in || a->yd_in || a->md_off || a->qname_off || a->prio_hi || b->yc_in || b->yx_in || b->yd_in || b->md_off || b->qname_off ||
      b->prio_hi)
    return TBK_EUNSUPPORTED;
  BamDev* B = (BamDev*)ctx->bam_dev;
  TBK_HIP(hipSetDevice(ctx->device));
  TBK_TRY(tbk_collapse_finish_yd(ctx));
  const uint64_t na = a->n_records, nb = b->n_records, ca = a->n_cigar_ops, cb = b->n_cigar_ops;
  if (na + nb >= (1ull << 32) || ca + cb >= (1ull << 32) || (uint64_t)a->n_files + b->n_files > 65535) return TBK_E2BIG;
  const size_t n = (size_t)(na + nb);
  int32_t* tid = bd_alloc<int32_t>(B, n);
  int32_t* pos = bd_alloc<int32_t>(B, n);
  int32_t* nh = bd_alloc<int32_t>(B, n);
  uint16_t* flag = bd_alloc<uint16_t>(B, n);
  uint8_t* mapq = bd_alloc<uint8_t>(B, n);
  uint8_t* strand = bd_alloc<uint8_t>(B, n);
  uint32_t* cig_off = bd_alloc<uint32_t>(B, n + 1);
  uint32_t* cig = bd_alloc<uint32_t>(B, (size_t)(ca + cb) + 1);
  uint32_t* tmp = bd_alloc<uint32_t>(B, (size_t)nb + 1);  // the host part's offsets as they come
  if (!tid || !pos || !nh || !flag || !mapq || !strand || !cig_off || !cig || !tmp) return TBK_ENOMEM;
  hipStream_t st = ctx->stream;
#define TBK_JOIN(dst, fa, fb, T)                                                                             \
  if (na) TBK_HIP(hipMemcpyAsync(dst, a->fa, (size_t)na * sizeof(T), hipMemcpyDeviceToDevice, st));          \
  if (nb) TBK_HIP(hipMemcpyAsync(dst + na, b->fb, (size_t)nb * sizeof(T), hipMemcpyHostToDevice, st));
  TBK_JOIN(tid, tid, tid, int32_t)
  TBK_JOIN(pos, pos, pos, int32_t)
  TBK_JOIN(nh, nh, nh, int32_t)
  TBK_JOIN(flag, flag, flag, uint16_t)
  TBK_JOIN(mapq, mapq, mapq, uint8_t)
  TBK_JOIN(strand, strand, strand, uint8_t)
#undef TBK_JOIN
  if (ca) TBK_HIP(hipMemcpyAsync(cig, a->cig, (size_t)ca * 4, hipMemcpyDeviceToDevice, st));
  if (cb) TBK_HIP(hipMemcpyAsync(cig + ca, b->cig, (size_t)cb * 4, hipMemcpyHostToDevice, st));
  if (na) TBK_HIP(hipMemcpyAsync(cig_off, a->cig_off, (size_t)na * 4, hipMemcpyDeviceToDevice, st));
  TBK_HIP(hipMemcpyAsync(tmp, b->cig_off, ((size_t)nb + 1) * 4, hipMemcpyHostToDevice, st));
  join_cig_off_k<<<cdiv((uint32_t)nb + 1u, 256u), 256, 0, st>>>((uint32_t)nb, tmp, (uint32_t)ca, cig_off + na);
  TBK_HIP(hipStreamSynchronize(st));  // (the host arrays may go once this returns)
  const uint32_t ka = a->n_files, kb = b->n_files;
  for (uint32_t f = 0; f <= ka; ++f) file_off_out[f] = a->file_off[f];
  for (uint32_t f = 1; f <= kb; ++f) file_off_out[ka + f] = (uint32_t)na + b->file_off[f];
  for (uint32_t f = 0; f < ka; ++f) tbmerged_out[f] = a->tbmerged ? a->tbmerged[f] : 0;
  for (uint32_t f = 0; f < kb; ++f) tbmerged_out[ka + f] = b->tbmerged ? b->tbmerged[f] : 0;
  memset(out, 0, sizeof(*out));
  out->mem = TBK_MEM_DEVICE;
  out->n_files = ka + kb;
  out->n_records = (uint32_t)n;
  out->n_cigar_ops = (uint32_t)(ca + cb);
  out->file_off = file_off_out;
  out->tbmerged = tbmerged_out;
  out->tid = tid;
  out->pos = pos;
  out->flag = flag;
  out->mapq = mapq;
  out->strand = strand;
  out->nh = nh;
  out->cig_off = cig_off;
  out->cig = cig;
  return tbk_check_launch(ctx, "tile_join");
}

// the raw records (block_size field included, i.e. framed as in the BAM stream) behind tile indices, packed in the order given
extern "C" int tbk_bam_records(tbk_ctx* ctx, const uint32_t* idx, uint32_t n, int idx_mem, uint8_t* out, uint64_t out_cap, uint64_t* out_off) {
  if (!ctx || !ctx->bam_dev || !out_off || (n && (!idx || !out))) return TBK_EINVAL;
  BamDev* B = (BamDev*)ctx->bam_dev;
  TBK_HIP(hipSetDevice(ctx->device));
  out_off[0] = 0;
  if (n == 0) return 0;
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n * 24 + ((size_t)4 << 20)));
  const uint32_t* d_idx = idx;
  if (idx_mem != TBK_MEM_DEVICE) {
    uint32_t* t = ws_alloc<uint32_t>(ctx, n);
    if (!t) return TBK_ENOMEM;
    TBK_HIP(hipMemcpyAsync(t, idx, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    d_idx = t;
  }
  uint32_t* sz = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
  uint64_t* off = ws_alloc<uint64_t>(ctx, (size_t)n + 1);
  if (!sz || !off) return TBK_ENOMEM;
  TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
  TBK_LAUNCH(ctx, "bam_recsize", bam_recsize_k, cdiv(n, 256), 256, 0, n, d_idx, B->inf, B->rec, B->n, sz, ctx->d_err);
  TBK_HIP(hipMemsetAsync(sz + n, 0, 4, ctx->stream));
  TBK_TRY(tbk_exscan_u32_u64(ctx, sz, off, n + 1, ctx->d_scalars + 1));
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) return TBK_EINVAL;
  const uint64_t total = ctx->h_scalars[1];
  if (total > out_cap) {
    out_off[n] = total;
    return TBK_E2BIG;
  }
  uint8_t* d_out = ws_alloc<uint8_t>(ctx, total);
  if (!d_out) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, "bam_reccopy", bam_reccopy_k, cdiv((uint64_t)n * 64, 256), 256, 0, n, d_idx, B->inf, B->rec, off, d_out);
  TBK_HIP(hipMemcpyAsync(out, d_out, total, hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipMemcpyAsync(out_off, off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "bam_records");
}
