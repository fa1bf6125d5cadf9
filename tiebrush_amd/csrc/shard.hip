// shard.hip — device side of the multi-GPU path (tiebrush_amd/dist.py): shuffle, then collapse.
//
// Every rank holds some of the input files.  Instead of collapsing locally and stitching partial groups afterwards,
// the ranks agree on coordinate cuts that no read of any file spans (global bundle boundaries), send every passing
// record to the rank that owns its coordinate range, and each rank then runs the ordinary tbk_collapse_tile /
// tbk_coverage_tile on complete data: one collapse per record, YD computed once on whole groups, no second exchange.
//
//   tbk_shard_prepare   per record: merge key, filter verdict (passes_options, tiebrush.cpp:532-541), the effective
//                       end of the k-way merge (per-file running max, tmerge.h:28-50 — computed BEFORE filtering, so it
//                       travels with the record as its explicit priority), and the per-file running max of the read
//                       ends that the cut search needs
//   tbk_shard_probe_*   for candidate cuts: the farthest read end before each cut / the next record start after a key
//   tbk_shard_pack      passing records -> 24-byte rows + CIGAR words grouped by (destination rank, file), in file order
//   tbk_shard_unpack    received rows -> the SoA arrays of a tile whose "files" are all the input files
//
//
// Group partials (the default multi-rank protocol; the record shuffle above stays as the fallback for carried
// fractional YC): every rank collapses its own files with the ordinary single-GPU path and ships one 40-byte row per LOCAL
// group — key fields, local YC / YX / YD, and the merge priority (effective end, file, index) of its local representative — plus
// that representative's CIGAR to the rank that owns the group's coordinate range; the owner collapses the partials as
// TieBrush-merged records with an explicit priority (sum YC, sum YX, max YD, argmin priority: dupAdd is associative,
// tiebrush.cpp:408-436; processRead reads only the representative's start and exons, which every member of a group shares,
// :225-249, so the YD of a sample is final on the rank that holds it).
//   tbk_partial_keys    per local group: cut-search key (tid + 1, start) and running keyed maximum of end + 1
//   tbk_partial_pack    groups -> rows + CIGAR words in group order (destinations are contiguous group ranges: no reordering)
//   tbk_partial_unpack  received rows -> the SoA arrays of a tile of TieBrush-merged records with explicit priorities
//
// All integer / byte work, HBM-bound, no MFMA.  No data-path collective lives here: the exchange itself is
// torch.distributed (RCCL over xGMI) in dist.py.
#include "dev_common.hpp"
#include "scan_op.hpp"
#include "strategy.hpp"
#include "tbk_internal.h"
#include "wgroup.h"

namespace {
constexpr int SH_B = 256;

struct ShOpt {
  int max_nh, min_qual;
  uint8_t keep_supp, keep_sec;
};

// key = (tid + 1) : 32 | pos + 1 : 31 (for a mapped read the GSamRecord 1-based start) — per file nondecreasing in a
// coordinate-sorted BAM: unmapped reads placed at their mate's position keep that position, unplaced reads (tid < 0, the tail
// of the file) take a key beyond every reference
__global__ void shard_keys_k(uint32_t n, uint32_t k, const uint32_t* __restrict__ file_off, const int32_t* __restrict__ tid,
                             const int32_t* __restrict__ pos, const uint16_t* __restrict__ flag, const uint8_t* __restrict__ mapq,
                             const int32_t* __restrict__ nh, const uint32_t* __restrict__ cig_off, const uint32_t* __restrict__ cig,
                             ShOpt O, int64_t* __restrict__ key, int32_t* __restrict__ kend, uint8_t* __restrict__ kfl,
                             uint16_t* __restrict__ fidx, uint32_t* __restrict__ err) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    bool pass;
    uint32_t lo = 0, hi = k;  // last f with file_off[f] <= i
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (file_off[mid] <= i)
        lo = mid;
      else
        hi = mid;
    }
    const uint16_t fl = flag[i];
    int start = 0, end = 0;
    if (!(fl & 0x4)) {
      int l = cigar_reflen(cig + cig_off[i], cig_off[i + 1] - cig_off[i]);
      start = pos[i] + 1;
      end = pos[i] + l;
    }
    pass = true;  // passes_options, tiebrush.cpp:532-541
    if (!O.keep_supp && (fl & 0x800)) pass = false;
    if (!O.keep_sec && (fl & 0x100)) pass = false;
    if (fl & 0x4) pass = false;
    if ((int)mapq[i] < O.min_qual) pass = false;
    int h = nh[i] == TBK_NH_ABSENT ? 0 : nh[i];
    if (h > O.max_nh) pass = false;
    if (pass && (start < 0 || tid[i] < -1 || (int64_t)end - start + 1 >= (1ll << 30))) atomicOr(err, TBK_DERR_SPAN);
    key[i] = tid[i] < 0 ? (int64_t)(1ll << 62) : (((int64_t)(uint32_t)(tid[i] + 1) << 31) | (uint32_t)(pos[i] + 1));
    kend[i] = end;
    kfl[i] = (pass ? 1u : 0u) | (i == file_off[lo] ? 2u : 0u);
    fidx[i] = (uint16_t)lo;
  }
}

struct ShKey {  // scan element (32-bit words): running (key, end) maximum and running keyed-end maximum, both per file
  uint32_t kh, kl;
  int32_t kend;
  uint32_t mh, ml;
  uint32_t head;
};
struct ShOp {
  __device__ __forceinline__ ShKey operator()(const ShKey& a, const ShKey& b) const {
    const uint64_t ak = ((uint64_t)a.kh << 32) | a.kl, bk = ((uint64_t)b.kh << 32) | b.kl;
    const uint64_t am = ((uint64_t)a.mh << 32) | a.ml, bm = ((uint64_t)b.mh << 32) | b.ml;
    const bool take_b = b.head || bk > ak || (bk == ak && b.kend > a.kend);
    const bool m_b = b.head || bm > am;
    ShKey r;
    r.kh = take_b ? b.kh : a.kh;
    r.kl = take_b ? b.kl : a.kl;
    r.kend = take_b ? b.kend : a.kend;
    r.mh = m_b ? b.mh : a.mh;
    r.ml = m_b ? b.ml : a.ml;
    r.head = a.head | b.head;
    return r;
  }
};
struct ShLoad {
  const int64_t* key;
  const int32_t* kend;
  const uint8_t* kfl;
  __device__ __forceinline__ ShKey operator()(uint32_t i) const {
    const uint64_t k = (uint64_t)key[i];
    // (tid + 1) : 32 | end + 1 : 31 — a cut must lie beyond end + 1: the YD lists may hold a node (end + 1, end) of a CIGAR
    // that ends in an intron
    const uint64_t m = (k & ~0x7FFFFFFFull) | (uint32_t)(kend[i] + 1);
    ShKey e;
    e.kh = (uint32_t)(k >> 32);
    e.kl = (uint32_t)k;
    e.kend = kend[i];
    e.mh = (uint32_t)(m >> 32);
    e.ml = (uint32_t)m;
    e.head = (kfl[i] >> 1) & 1u;
    return e;
  }
};
struct ShStore {
  const int64_t* key;
  int32_t* effend;
  int64_t* emax;
  uint32_t* err;
  __device__ __forceinline__ void operator()(uint32_t i, const ShKey&, const ShKey& inc, const ShKey&) const {
    const uint64_t ik = ((uint64_t)inc.kh << 32) | inc.kl;
    if (ik != (uint64_t)key[i]) atomicOr(err, TBK_DERR_UNSORTED);  // an earlier record of the file has a larger (tid,start)
    effend[i] = inc.kend;
    emax[i] = (int64_t)(((uint64_t)inc.mh << 32) | inc.ml);
  }
};

// one thread per (cut, file): the farthest keyed read end among the file's records that start before the cut
__global__ void shard_probe_max_k(uint32_t nc, uint32_t k, const uint32_t* __restrict__ file_off, const int64_t* __restrict__ key,
                                  const int64_t* __restrict__ emax, const int64_t* __restrict__ cuts, long long* __restrict__ m_out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nc * k) return;
  const uint32_t c = t / k, f = t % k;
  const int64_t p = cuts[c];
  uint32_t lo = file_off[f], hi = file_off[f + 1];
  const uint32_t f0 = lo;
  while (lo < hi) {  // first record with key >= p
    uint32_t mid = (lo + hi) >> 1;
    if (key[mid] < p)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo > f0) atomicMax(&m_out[c], (long long)emax[lo - 1]);
}
// one thread per (cut, file): the first record start of the file that lies beyond the keyed end m[c]
__global__ void shard_probe_next_k(uint32_t nc, uint32_t k, const uint32_t* __restrict__ file_off, const int64_t* __restrict__ key,
                                   const int64_t* __restrict__ m, long long* __restrict__ nxt_out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nc * k) return;
  const uint32_t c = t / k, f = t % k;
  const int64_t v = m[c];
  uint32_t lo = file_off[f], hi = file_off[f + 1];
  const uint32_t f1 = hi;
  while (lo < hi) {  // first record with key > v
    uint32_t mid = (lo + hi) >> 1;
    if (key[mid] <= v)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < f1) atomicMin(&nxt_out[c], (long long)key[lo]);
}

__global__ void shard_passcig_k(uint32_t n, const uint8_t* __restrict__ pass, const uint32_t* __restrict__ cig_off, uint32_t* __restrict__ p32,
                                uint32_t* __restrict__ c32) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = pass[i] & 1u;
  p32[i] = p;
  c32[i] = p ? cig_off[i + 1] - cig_off[i] : 0u;
}

// (destination, file) blocks: a file is sorted, so the records of file f that go to rank d are the contiguous range
// between the lower bounds of cut d-1 and cut d.  One thread per block computes the range and its passing-row /
// CIGAR-word counts; thread 0 then lays the blocks out destination-major, file-minor.
// tab layout (int64): [world][k] x {first record, rows, words, row base, word base}
__global__ void shard_table_k(uint32_t n, uint32_t world, uint32_t k, const uint32_t* __restrict__ file_off, const int64_t* __restrict__ key,
                              const int64_t* __restrict__ cuts, const uint32_t* __restrict__ pr, const uint32_t* __restrict__ cg,
                              const uint64_t* __restrict__ tot_rows, const uint64_t* __restrict__ tot_words, long long* __restrict__ tab) {
  const uint32_t nblk = world * k;
  for (uint32_t t = threadIdx.x; t < nblk; t += blockDim.x) {
    const uint32_t d = t / k, f = t % k;
    auto lower = [&](uint32_t which) -> uint32_t {  // first record of file f with key >= cut[which - 1]; which == 0: file start, == world: file end
      if (which == 0) return file_off[f];
      if (which == world) return file_off[f + 1];
      const int64_t p = cuts[which - 1];
      uint32_t lo = file_off[f], hi = file_off[f + 1];
      while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (key[mid] < p)
          lo = mid + 1;
        else
          hi = mid;
      }
      return lo;
    };
    const uint32_t a = lower(d), b = lower(d + 1);
    const uint64_t pa = a < n ? pr[a] : *tot_rows, pb = b < n ? pr[b] : *tot_rows;
    const uint64_t ca = a < n ? cg[a] : *tot_words, cb = b < n ? cg[b] : *tot_words;
    tab[(size_t)t * 5 + 0] = a;
    tab[(size_t)t * 5 + 1] = (long long)(pb - pa);
    tab[(size_t)t * 5 + 2] = (long long)(cb - ca);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long rb = 0, wb = 0;
    for (uint32_t t = 0; t < nblk; ++t) {
      tab[(size_t)t * 5 + 3] = rb;
      tab[(size_t)t * 5 + 4] = wb;
      rb += tab[(size_t)t * 5 + 1];
      wb += tab[(size_t)t * 5 + 2];
    }
  }
}

// rows: 6 x int32 per passing record = {tid, pos, strand, n_cigar, effective end, index inside its file}
__global__ void shard_scatter_k(uint32_t n, uint32_t world, uint32_t k, const uint32_t* __restrict__ file_off, const int64_t* __restrict__ key,
                                const int64_t* __restrict__ cuts, const uint8_t* __restrict__ pass, const uint16_t* __restrict__ fidx,
                                const uint32_t* __restrict__ pr, const uint32_t* __restrict__ cg, const long long* __restrict__ tab,
                                const int32_t* __restrict__ tid, const int32_t* __restrict__ pos, const uint8_t* __restrict__ strand,
                                const uint32_t* __restrict__ cig_off, const uint32_t* __restrict__ cig, const int32_t* __restrict__ effend,
                                int32_t* __restrict__ rows, uint32_t* __restrict__ cig_out, int64_t* __restrict__ src_idx) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !(pass[i] & 1u)) return;
  const int64_t kx = key[i];
  uint32_t d = 0;  // number of cuts <= key
  {
    uint32_t lo = 0, hi = world - 1;
    while (lo < hi) {
      uint32_t mid = (lo + hi) >> 1;
      if (cuts[mid] <= kx)
        lo = mid + 1;
      else
        hi = mid;
    }
    d = lo;
  }
  const uint32_t f = fidx[i];
  const long long* T = tab + ((size_t)d * k + f) * 5;
  const uint32_t a = (uint32_t)T[0];
  const uint64_t o = (uint64_t)T[3] + (pr[i] - pr[a]);
  const uint64_t w = (uint64_t)T[4] + (cg[i] - cg[a]);
  const uint32_t c0 = cig_off[i], nc = cig_off[i + 1] - c0;
  int32_t* R = rows + o * 6;
  R[0] = tid[i];
  R[1] = pos[i];
  R[2] = strand[i];
  R[3] = (int32_t)nc;
  R[4] = effend[i];
  R[5] = (int32_t)(i - file_off[f]);
  src_idx[o] = i;
  for (uint32_t q = 0; q < nc; ++q) cig_out[w + q] = cig[c0 + q];
}

__global__ void shard_fidx_k(uint32_t n, uint32_t k, const uint32_t* __restrict__ fo, uint16_t* __restrict__ fidx) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = k;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (fo[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  fidx[i] = (uint16_t)lo;
}
__global__ void shard_ncig_k(uint32_t n2, const int32_t* __restrict__ rows, uint32_t* __restrict__ nc) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n2) nc[j] = (uint32_t)rows[(size_t)j * 6 + 3];
}
// received rows -> SoA; the record's file is the run it lies in (file_off2 over ALL input files), its explicit priority
// (effective end, then file and index: the merge order of the reference) rides in prio_hi / prio_lo
__global__ void shard_unpack_k(uint32_t n2, uint32_t K, const int32_t* __restrict__ rows, const uint32_t* __restrict__ file_off2,
                               const uint64_t* __restrict__ total_words, int32_t* __restrict__ tid, int32_t* __restrict__ pos,
                               uint16_t* __restrict__ flag, uint8_t* __restrict__ mapq, uint8_t* __restrict__ strand, int32_t* __restrict__ nh,
                               uint32_t* __restrict__ cig_off, int64_t* __restrict__ prio_hi, int64_t* __restrict__ prio_lo) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n2) return;
  const int32_t* R = rows + (size_t)j * 6;
  uint32_t lo = 0, hi = K;  // last f with file_off2[f] <= j
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (file_off2[mid] <= j)
      lo = mid;
    else
      hi = mid;
  }
  tid[j] = R[0];
  pos[j] = R[1];
  flag[j] = 0;
  mapq[j] = 255;
  strand[j] = (uint8_t)R[2];
  nh[j] = TBK_NH_ABSENT;
  prio_hi[j] = (int64_t)R[4];
  prio_lo[j] = ((int64_t)lo << 32) | (uint32_t)R[5];
  if (j + 1 == n2) cig_off[n2] = (uint32_t)*total_words;
}
}  // namespace

static int shard_upload_file_off(tbk_ctx* ctx, const uint32_t* host_fo, uint32_t k, uint32_t** d_fo) {
  *d_fo = ws_alloc<uint32_t>(ctx, (size_t)k + 1);
  if (!*d_fo) return TBK_ENOMEM;
  const size_t bytes = (size_t)(k + 1) * 4;
  if (bytes <= 4096 * sizeof(uint64_t)) {  // staged through the pinned block: asynchronous, the caller's array may be transient
    void* stage = tbk_stage_acquire(ctx);
    memcpy(stage, host_fo, bytes);
    TBK_HIP(hipMemcpyAsync(*d_fo, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
    tbk_stage_release(ctx);
  } else {
    TBK_HIP(hipMemcpyAsync(*d_fo, host_fo, bytes, hipMemcpyHostToDevice, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return 0;
}

extern "C" int tbk_shard_prepare(tbk_ctx* ctx, const tbk_collapse_opts* o, const tbk_soa_in* in, int64_t* key, int64_t* emax,
                                 int32_t* effend, uint8_t* pass) {
  if (!ctx || !o || !in || !key || !emax || !effend || !pass) return TBK_EINVAL;
  if (in->mem != TBK_MEM_DEVICE || in->n_files == 0 || in->n_files > 65535 || !in->file_off) return TBK_EINVAL;
  if (o->flags_mask != 0 || o->keep_unmapped) return TBK_EUNSUPPORTED;
  TBK_HIP(hipSetDevice(ctx->device));
  const uint32_t n = in->n_records;
  if (n == 0) return 0;
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n * 16 + ((size_t)1 << 20)));
  uint64_t* sc = ctx->d_scalars;
  TBK_HIP(hipMemsetAsync(sc, 0, 16 * sizeof(uint64_t), ctx->stream));
  uint32_t* d_fo = nullptr;
  TBK_TRY(shard_upload_file_off(ctx, in->file_off, in->n_files, &d_fo));
  int32_t* kend = ws_alloc<int32_t>(ctx, n);
  uint16_t* fidx = ws_alloc<uint16_t>(ctx, n);
  if (!kend || !fidx) return TBK_ENOMEM;
  ShOpt O{o->max_nh, o->min_qual, o->keep_supplementary, o->keep_secondary};
  TBK_LAUNCH(ctx, "shard_keys", shard_keys_k, cdiv(n, SH_B), SH_B, 0, n, in->n_files, d_fo, in->tid, in->pos, in->flag, in->mapq, in->nh,
             in->cig_off, in->cig, O, key, kend, pass, fidx, ctx->d_err);
  {
    ShLoad ld{key, kend, pass};
    ShStore st{key, effend, emax, ctx->d_err};
    ShKey ident{0u, 0u, INT32_MIN, 0u, 0u, 0u};
    TBK_TRY((scan_op_run<ShKey, ShOp, ShLoad, ShStore>(ctx, "shard_eff_scan", n, ld, st, ShOp{}, ident, true)));
  }
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) return tbk_derr_to_status(ctx, eb);
  return tbk_check_launch(ctx, "shard_prepare");
}

// m_out[c] (caller-initialised to -1) := max(m_out[c], farthest keyed end before cuts[c]); everything device-resident
extern "C" int tbk_shard_probe_max(tbk_ctx* ctx, const uint32_t* file_off, uint32_t n_files, const int64_t* key, const int64_t* emax,
                                   const int64_t* cuts, uint32_t n_cuts, int64_t* m_out) {
  if (!ctx || !file_off || n_files == 0) return TBK_EINVAL;
  if (n_cuts == 0 || file_off[n_files] == 0) return 0;  // (a rank whose files hold no records contributes nothing)
  if (!key || !emax || !cuts || !m_out) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)1 << 20));
  uint32_t* d_fo = nullptr;
  TBK_TRY(shard_upload_file_off(ctx, file_off, n_files, &d_fo));
  TBK_LAUNCH(ctx, "shard_probe_max", shard_probe_max_k, cdiv((size_t)n_cuts * n_files, SH_B), SH_B, 0, n_cuts, n_files, d_fo, key, emax, cuts,
             (long long*)m_out);
  TBK_HIP(hipStreamSynchronize(ctx->stream));  // the caller reads the result on its own stream / hands it to a collective
  return tbk_check_launch(ctx, "shard_probe_max");
}
// nxt_out[c] (caller-initialised to +inf) := min(nxt_out[c], first record start beyond m[c])
extern "C" int tbk_shard_probe_next(tbk_ctx* ctx, const uint32_t* file_off, uint32_t n_files, const int64_t* key, const int64_t* m,
                                    uint32_t n_cuts, int64_t* nxt_out) {
  if (!ctx || !file_off || n_files == 0) return TBK_EINVAL;
  if (n_cuts == 0 || file_off[n_files] == 0) return 0;
  if (!key || !m || !nxt_out) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)1 << 20));
  uint32_t* d_fo = nullptr;
  TBK_TRY(shard_upload_file_off(ctx, file_off, n_files, &d_fo));
  TBK_LAUNCH(ctx, "shard_probe_next", shard_probe_next_k, cdiv((size_t)n_cuts * n_files, SH_B), SH_B, 0, n_cuts, n_files, d_fo, key, m,
             (long long*)nxt_out);
  TBK_HIP(hipStreamSynchronize(ctx->stream));  // the caller reads the result on its own stream / hands it to a collective
  return tbk_check_launch(ctx, "shard_probe_next");
}

// rows [<= n_records][6] int32, cig_out [<= n_cigar_ops] uint32, src_idx [<= n_records] int64 (row -> local record), tab
// [world][n_files][5] int64 on the DEVICE (first record, rows, words, row base, word base per (destination, file))
extern "C" int tbk_shard_pack(tbk_ctx* ctx, const tbk_soa_in* in, const int64_t* key, const uint8_t* pass, const int32_t* effend,
                              const int64_t* cuts, uint32_t world, int32_t* rows, uint32_t* cig_out, int64_t* src_idx, int64_t* tab) {
  if (!ctx || !in || !tab || world == 0) return TBK_EINVAL;
  // (a rank whose files hold no records still takes part: its table of counts is all zero)
  if (in->n_records && (!key || !pass || !effend || !rows || !cig_out || !src_idx)) return TBK_EINVAL;
  if (world > 1 && !cuts) return TBK_EINVAL;
  if (in->mem != TBK_MEM_DEVICE || in->n_files == 0 || in->n_files > 65535 || !in->file_off) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  const uint32_t n = in->n_records, k = in->n_files;
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n * 24 + ((size_t)1 << 20)));
  uint64_t* sc = ctx->d_scalars;
  uint32_t* d_fo = nullptr;
  TBK_TRY(shard_upload_file_off(ctx, in->file_off, k, &d_fo));
  uint32_t* p32 = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
  uint32_t* c32 = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
  uint32_t* pr = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
  uint32_t* cg = ws_alloc<uint32_t>(ctx, (size_t)n + 1);
  uint16_t* fidx = ws_alloc<uint16_t>(ctx, (size_t)n + 1);
  if (!fidx) return TBK_ENOMEM;
  if (n) {
    TBK_LAUNCH(ctx, "shard_passcig", shard_passcig_k, cdiv(n, SH_B), SH_B, 0, n, pass, in->cig_off, p32, c32);
  }
  TBK_TRY(tbk_exscan_u32(ctx, p32, pr, n, sc + 21));
  TBK_TRY(tbk_exscan_u32(ctx, c32, cg, n, sc + 22));
  TBK_LAUNCH(ctx, "shard_table", shard_table_k, 1, 256, 0, n, world, k, d_fo, key, cuts, pr, cg, sc + 21, sc + 22, (long long*)tab);
  if (n) {
    // (the file index of a record: recomputed here, prepare's scratch is gone)
    TBK_LAUNCH(ctx, "shard_fidx", shard_fidx_k, cdiv(n, SH_B), SH_B, 0, n, k, d_fo, fidx);
    TBK_LAUNCH(ctx, "shard_scatter", shard_scatter_k, cdiv(n, SH_B), SH_B, 0, n, world, k, d_fo, key, cuts, pass, fidx, pr, cg,
               (const long long*)tab, in->tid, in->pos, in->strand, in->cig_off, in->cig, effend, rows, cig_out, src_idx);
  }
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "shard_pack");
}

// rows of all source ranks, concatenated in (source rank, file) order; file_off2[K + 1] (host) = run boundaries over all K
// input files.  Fills the SoA arrays of the tile (cig_off gets n2 + 1 entries); the CIGAR words are used as received.
extern "C" int tbk_shard_unpack(tbk_ctx* ctx, const int32_t* rows, uint32_t n2, const uint32_t* file_off2, uint32_t K, int32_t* tid,
                                int32_t* pos, uint16_t* flag, uint8_t* mapq, uint8_t* strand, int32_t* nh, uint32_t* cig_off, int64_t* prio_hi,
                                int64_t* prio_lo) {
  if (!ctx || !file_off2 || K == 0 || !cig_off) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n2 * 8 + ((size_t)1 << 20)));
  if (n2 == 0) {
    TBK_HIP(hipMemsetAsync(cig_off, 0, 4, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  if (!rows || !tid || !pos || !flag || !mapq || !strand || !nh || !prio_hi || !prio_lo) return TBK_EINVAL;
  uint32_t* d_fo = nullptr;
  TBK_TRY(shard_upload_file_off(ctx, file_off2, K, &d_fo));
  uint32_t* nc = ws_alloc<uint32_t>(ctx, n2);
  if (!nc) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, "shard_ncig", shard_ncig_k, cdiv(n2, SH_B), SH_B, 0, n2, rows, nc);
  TBK_TRY(tbk_exscan_u32(ctx, nc, cig_off, n2, ctx->d_scalars + 23));
  TBK_LAUNCH(ctx, "shard_unpack", shard_unpack_k, cdiv(n2, SH_B), SH_B, 0, n2, K, rows, d_fo, ctx->d_scalars + 23, tid, pos, flag, mapq, strand, nh,
             cig_off, prio_hi, prio_lo);
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "shard_unpack");
}

// ==================================== group partials ====================================
namespace {
struct PmKey {  // scan element: running maximum of a 64-bit word (two 32-bit halves)
  uint32_t h, l;
};
struct PmOp {
  __device__ __forceinline__ PmKey operator()(const PmKey& a, const PmKey& b) const {
    const uint64_t x = ((uint64_t)a.h << 32) | a.l, y = ((uint64_t)b.h << 32) | b.l;
    return y > x ? b : a;
  }
};
struct PmLoad {
  const uint32_t* rep;
  const int32_t* tid;
  const int32_t* g_end;
  const uint64_t* gkey;  // tbk_groups_out.g_key when the caller has it: the reference id without the gather through rep
  __device__ __forceinline__ PmKey operator()(uint32_t o) const {
    // (tid + 1) : 32 | end + 1 : 31 — the same keyed end as ShLoad: a cut lies beyond end + 1
    const uint32_t t1 = gkey ? (uint32_t)(gkey[2 * (size_t)o] >> 33) : (uint32_t)(tid[rep[o]] + 1);
    const uint64_t m = ((uint64_t)t1 << 31) | (uint32_t)(g_end[o] + 1);
    return PmKey{(uint32_t)(m >> 32), (uint32_t)m};
  }
};
struct PmStore {
  int64_t* emax;
  __device__ __forceinline__ void operator()(uint32_t o, const PmKey&, const PmKey& inc, const PmKey&) const {
    emax[o] = (int64_t)(((uint64_t)inc.h << 32) | inc.l);
  }
};
// key of a local group for the cut search + what the 32-bit row fields cannot hold (flag word: bit 0)
__global__ void partial_keys_k(uint32_t ng, const uint32_t* __restrict__ rep, const int32_t* __restrict__ tid, const int32_t* __restrict__ g_start,
                               const double* __restrict__ yc, const int64_t* __restrict__ yx, const uint64_t* __restrict__ gkey,
                               int64_t* __restrict__ key, uint32_t* __restrict__ bad) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  const uint32_t t1 = gkey ? (uint32_t)(gkey[2 * (size_t)o] >> 33) : (uint32_t)(tid[rep[o]] + 1);
  key[o] = (int64_t)(((uint64_t)t1 << 31) | (uint32_t)g_start[o]);
  const double y = yc[o];
  if (!(y == rint(y)) || y < 1.0 || y >= 2147483648.0 || yx[o] < 0 || yx[o] >= 2147483648ll) atomicOr(bad, 1u);
}
// With tbk_groups_out.g_key a group whose alignment is a single M or M N M (the key says so) is packed from its key: place, strand,
// key word and the CIGAR words themselves — the representative's record is fetched only for the other groups (soft clips do not
// travel then: neither the owner's comparisons under -P nor tiecov look at them).
__device__ __forceinline__ uint32_t partial_key_ops(const uint64_t* __restrict__ gkey, uint32_t o) {
  if (!gkey) return 0u;
  const uint32_t shape = (uint32_t)gkey[2 * (size_t)o + 1];
  return shape == 0x80000000u ? 1u : ((shape >> 30) == 3u ? 3u : 0u);
}
__global__ void partial_ncig_k(uint32_t ng, const uint32_t* __restrict__ rep, const uint32_t* __restrict__ cig_off, const uint64_t* __restrict__ gkey,
                               uint32_t* __restrict__ cnt, uint32_t* __restrict__ cfirst) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  const uint32_t nk = partial_key_ops(gkey, o);
  if (nk) {
    cnt[o] = nk;
    cfirst[o] = 0;
    return;
  }
  const uint32_t r = rep[o], c0 = cig_off[r];
  cnt[o] = cig_off[r + 1] - c0;
  cfirst[o] = c0;
}
// tab[d] = {first group, rows, words} of destination d: groups with cuts[d - 1] <= key < cuts[d]
__global__ void partial_table_k(uint32_t ng, uint32_t world, const int64_t* __restrict__ key, const int64_t* __restrict__ cuts,
                                const uint32_t* __restrict__ woff, const uint64_t* __restrict__ tot_words, long long* __restrict__ tab) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= world) return;
  auto lower = [&](uint32_t which) -> uint32_t {
    if (which == 0) return 0u;
    if (which == world) return ng;
    const int64_t p = cuts[which - 1];
    uint32_t lo = 0, hi = ng;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (key[mid] < p)
        lo = mid + 1;
      else
        hi = mid;
    }
    return lo;
  };
  const uint32_t a = lower(d), b = lower(d + 1);
  const uint64_t wa = a < ng ? woff[a] : *tot_words, wb = b < ng ? woff[b] : *tot_words;
  tab[d * 3 + 0] = a;
  tab[d * 3 + 1] = (long long)(b - a);
  tab[d * 3 + 2] = (long long)(wb - wa);
}
// rows: TBK_PARTIAL_ROW x int32 per local group, in group (= output) order.  Words 9 / 10 are the low half of the group key
// (strategy.hpp, record_key: reference span and the 32-bit key word — an exact code or the strategy hash under TBK_KEY_SEED0) of
// the representative, which every member of the group shares: the owner groups partials by (tid, pos, strand, span, word)
// without walking a CIGAR, and verifies the hashed ones.
__global__ void partial_rows_k(uint32_t ng, uint32_t k, uint32_t first_fidx, ColIn I, ColOpt O, const uint32_t* __restrict__ rep,
                               const double* __restrict__ yc, const int64_t* __restrict__ yx, const int32_t* __restrict__ yd,
                               const int32_t* __restrict__ effend, const uint32_t* __restrict__ cfirst, const uint32_t* __restrict__ cnt,
                               const uint32_t* __restrict__ woff, const uint64_t* __restrict__ gkey, int32_t* __restrict__ rows,
                               uint32_t* __restrict__ cig_out, uint32_t* __restrict__ err) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  const uint32_t r = rep[o];
  const uint32_t* file_off = I.file_off;
  const int32_t *tid = I.tid, *pos = I.pos;
  const uint8_t* strand = I.strand;
  const uint32_t* cig = I.cig;
  uint32_t lo = 0, hi = k;  // last f with file_off[f] <= r
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (file_off[mid] <= r)
      lo = mid;
    else
      hi = mid;
  }
  const uint32_t nc = cnt[o], c0 = cfirst[o], w = woff[o];
  int32_t R[TBK_PARTIAL_ROW];
  const uint32_t nk = partial_key_ops(gkey, o);
  uint64_t klo;
  if (nk) {  // everything but the merge priority and the sums comes from the key
    const uint64_t k0 = gkey[2 * (size_t)o], k1 = gkey[2 * (size_t)o + 1];
    const uint32_t sc = (uint32_t)k0 & 3u, span = (uint32_t)(k1 >> 32), shape = (uint32_t)k1;
    R[0] = (int32_t)(uint32_t)(k0 >> 33) - 1;
    R[1] = (int32_t)(uint32_t)((k0 >> 2) & 0x7FFFFFFFull) - 1;
    R[2] = (int32_t)((sc == 0u ? (uint32_t)'+' : (sc == 1u ? (uint32_t)'-' : (uint32_t)'.')) | (nk << 8));
    klo = k1;  // (span : 32 | exact code : 32 — the code of record_key for these two shapes is the shape word)
    if (nk == 1u) {
      cig_out[w] = (span << 4) | C_M;
    } else {
      const uint32_t a = (shape >> 20) & 0x3FFu, g = shape & 0xFFFFFu;
      cig_out[w] = (a << 4) | C_M;
      cig_out[w + 1] = (g << 4) | C_N;
      cig_out[w + 2] = ((span - a - g) << 4) | C_M;
    }
  } else {
    const RecKey K = record_key(I, O, r, I.flag[r], pos[r], tid[r], (int)I.mapq[r], I.nh[r], strand_code(strand[r]), cig + c0, nc);
    if (K.err || !K.pass) atomicOr(err, K.err | TBK_DERR_INTERNAL);  // (a representative passes the filters by construction)
    R[0] = tid[r];
    R[1] = pos[r];
    R[2] = (int32_t)((uint32_t)strand[r] | (nc << 8));
    klo = K.lo;
    for (uint32_t q = 0; q < nc; ++q) cig_out[w + q] = cig[c0 + q];
  }
  R[3] = effend[o];
  R[4] = (int32_t)(first_fidx + lo);
  R[5] = (int32_t)(r - file_off[lo]);
  R[6] = (int32_t)(uint32_t)yc[o];
  R[7] = (int32_t)yx[o];
  R[8] = yd[o];
  R[9] = (int32_t)(uint32_t)(klo >> 32);
  R[10] = (int32_t)(uint32_t)klo;
  R[11] = 0;
  static_assert(TBK_PARTIAL_ROW == 12, "row layout");
  int4* dst = reinterpret_cast<int4*>(rows + (size_t)o * TBK_PARTIAL_ROW);  // (48-byte rows: 16-byte aligned)
#pragma unroll
  for (int q = 0; q < TBK_PARTIAL_ROW / 4; ++q) dst[q] = make_int4(R[4 * q], R[4 * q + 1], R[4 * q + 2], R[4 * q + 3]);
}
__global__ void partial_ncig2_k(uint32_t n2, const int32_t* __restrict__ rows, uint32_t* __restrict__ nc) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n2) nc[j] = (uint32_t)rows[(size_t)j * TBK_PARTIAL_ROW + 2] >> 8;
}
__global__ void partial_unpack_k(uint32_t n2, const int32_t* __restrict__ rows, const uint64_t* __restrict__ total_words, int32_t* __restrict__ tid,
                                 int32_t* __restrict__ pos, uint16_t* __restrict__ flag, uint8_t* __restrict__ mapq, uint8_t* __restrict__ strand,
                                 int32_t* __restrict__ nh, uint32_t* __restrict__ cig_off, double* __restrict__ yc_in, int64_t* __restrict__ yx_in,
                                 int64_t* __restrict__ yd_in, int64_t* __restrict__ prio_hi, int64_t* __restrict__ prio_lo) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n2) return;
  const int4* src = reinterpret_cast<const int4*>(rows + (size_t)j * TBK_PARTIAL_ROW);  // (48-byte rows: 16-byte aligned)
  const int4 q0 = src[0], q1 = src[1], q2 = src[2];
  const int2 a = make_int2(q0.x, q0.y), b = make_int2(q0.z, q0.w), c = make_int2(q1.x, q1.y), d = make_int2(q1.z, q1.w), e = make_int2(q2.x, q2.y);
  tid[j] = a.x;
  pos[j] = a.y;
  flag[j] = 0;
  mapq[j] = 255;
  strand[j] = (uint8_t)((uint32_t)b.x & 0xFFu);
  nh[j] = TBK_NH_ABSENT;
  prio_hi[j] = (int64_t)b.y;
  prio_lo[j] = ((int64_t)c.x << 32) | (uint32_t)c.y;
  yc_in[j] = (double)(uint32_t)d.x;
  yx_in[j] = (int64_t)d.y;
  yd_in[j] = (int64_t)e.x;
  if (j + 1 == n2) cig_off[n2] = (uint32_t)*total_words;
}
}  // namespace

extern "C" int tbk_partial_keys(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* g, int64_t* key, int64_t* emax,
                                uint32_t* not_packable) {
  if (!ctx || !in || !g || !not_packable) return TBK_EINVAL;
  *not_packable = 0;
  if (in->mem != TBK_MEM_DEVICE || g->mem != TBK_MEM_DEVICE) return TBK_EINVAL;
  const uint32_t ng = g->n_groups;
  if (ng == 0) return 0;
  if (!key || !emax || !g->rep || !g->yc || !g->yx || !g->g_start || !g->g_end || !in->tid) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)ng * 8 + ((size_t)1 << 20)));
  uint64_t* sc = ctx->d_scalars;
  TBK_HIP(hipMemsetAsync(sc, 0, 16 * sizeof(uint64_t), ctx->stream));
  TBK_LAUNCH(ctx, "partial_keys", partial_keys_k, cdiv(ng, SH_B), SH_B, 0, ng, g->rep, in->tid, g->g_start, g->yc, g->yx, g->g_key, key, (uint32_t*)(sc + 8));
  {
    PmLoad ld{g->rep, in->tid, g->g_end, g->g_key};
    PmStore st{emax};
    TBK_TRY((scan_op_run<PmKey, PmOp, PmLoad, PmStore>(ctx, "partial_emax_scan", ng, ld, st, PmOp{}, PmKey{0u, 0u})));
  }
  uint32_t eb = 0;
  TBK_TRY(tbk_sync_err(ctx, &eb));
  if (eb) return tbk_derr_to_status(ctx, eb);
  *not_packable = (uint32_t)ctx->h_scalars[8];
  return tbk_check_launch(ctx, "partial_keys");
}

extern "C" int tbk_partial_pack(tbk_ctx* ctx, const tbk_collapse_opts* o, const tbk_soa_in* in, const tbk_groups_out* g, const int64_t* key,
                                const int64_t* cuts, uint32_t world, uint32_t first_fidx, int32_t* rows, uint32_t* cig_out, int64_t* tab) {
  if (!ctx || !o || !in || !g || !tab || world == 0) return TBK_EINVAL;
  if (o->strategy < 0 || o->strategy > 3) return TBK_EUNSUPPORTED;
  if (o->strategy == TBK_STRAT_FULL && in->n_records && (!in->md_off || !in->md_has)) return TBK_EINVAL;  // (-L: the MD strings ride along, tbk_partial_pack_md)
  if (in->mem != TBK_MEM_DEVICE || g->mem != TBK_MEM_DEVICE || in->n_files == 0 || in->n_files > 65535 || !in->file_off) return TBK_EINVAL;
  const uint32_t ng = g->n_groups;
  if (ng && (!key || !rows || !cig_out || !g->rep || !g->yc || !g->yx || !g->yd || !g->rep_effend)) return TBK_EINVAL;
  if (world > 1 && !cuts) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)ng * 16 + ((size_t)1 << 20)));
  uint64_t* sc = ctx->d_scalars;
  uint32_t* cnt = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  uint32_t* cfirst = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  uint32_t* woff = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  if (!woff) return TBK_ENOMEM;
  uint32_t* d_fo = nullptr;
  TBK_TRY(shard_upload_file_off(ctx, in->file_off, in->n_files, &d_fo));
  const uint64_t* pk_key = o->strategy == TBK_STRAT_FULL ? nullptr : g->g_key;  // (-L: the key describes no alignment by itself, MD is part of it)
  if (ng) TBK_LAUNCH(ctx, "partial_ncig", partial_ncig_k, cdiv(ng, SH_B), SH_B, 0, ng, g->rep, in->cig_off, pk_key, cnt, cfirst);
  TBK_TRY(tbk_exscan_u32(ctx, cnt, woff, ng, sc + 21));
  TBK_LAUNCH(ctx, "partial_table", partial_table_k, cdiv(world, 64), 64, 0, ng, world, key, cuts, woff, sc + 21, (long long*)tab);
  if (ng) {
    ColIn I{};
    I.n = in->n_records;
    I.k = in->n_files;
    I.file_off = d_fo;
    I.tid = in->tid;
    I.pos = in->pos;
    I.flag = in->flag;
    I.mapq = in->mapq;
    I.strand = in->strand;
    I.nh = in->nh;
    I.cig_off = in->cig_off;
    I.cig = in->cig;
    I.md_off = in->md_off;
    I.md = in->md;
    I.md_has = in->md_has;
    ColOpt O{};
    O.strategy = o->strategy;
    O.max_nh = o->max_nh;
    O.min_qual = o->min_qual;
    O.keep_supp = o->keep_supplementary;
    O.keep_sec = o->keep_secondary;
    O.seed = TBK_KEY_SEED0;
    O.hash_mask = ctx->dbg.hash_mask;
    TBK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
    TBK_LAUNCH(ctx, "partial_rows", partial_rows_k, cdiv(ng, SH_B), SH_B, 0, ng, in->n_files, first_fidx, I, O, g->rep, g->yc, g->yx, g->yd,
               g->rep_effend, cfirst, cnt, woff, pk_key, rows, cig_out, ctx->d_err);
    uint32_t eb = 0;
    TBK_TRY(tbk_sync_err(ctx, &eb));
    if (eb) return tbk_derr_to_status(ctx, eb);
  }
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "partial_pack");
}

// ---- the sender's side of the group-partials protocol in three stages that never wait for the device (ABI 7) ---------------------
// The first version of the protocol walked every cut forward in rounds — two all-reduces and a device -> host decision per round, a
// handful of rounds per step — and read counts back between its stages.  Here the host only queues work: between the stages lie the
// collectives (all-gathers of small device arrays), and the one thing the host ever reads is the table of the exchange, gathered from
// every rank at once, with every verdict the stages reached riding along in its flag words.
namespace {
constexpr uint32_t PM_SAMPLES = 64, PM_META = TBK_PARTIAL_META, PM_B = TBK_PARTIAL_BUNDLES, PM_CAND = TBK_PARTIAL_CAND;
constexpr int64_t PM_INF = (int64_t)1 << 62;
static_assert(PM_CAND == 2 + 2 * PM_B && PM_META == PM_SAMPLES + 4, "include/tbk.h");

__global__ void partial_meta_k(uint32_t ng, const int64_t* __restrict__ key, uint32_t n_files, uint32_t first_fidx, int64_t carry,
                               const uint32_t* __restrict__ bad, int64_t* __restrict__ meta) {
  const uint32_t i = threadIdx.x;
  if (i < PM_SAMPLES) meta[i] = ng ? key[(size_t)((uint64_t)i * ng / PM_SAMPLES)] : PM_INF;
  if (i == PM_SAMPLES) meta[i] = n_files;
  if (i == PM_SAMPLES + 1) meta[i] = first_fidx;
  if (i == PM_SAMPLES + 2) meta[i] = (int64_t)(*bad & 1u);
  if (i == PM_SAMPLES + 3) meta[i] = carry;
}

// the world - 1 splitter targets: the j / world quantiles of the valid samples of every rank (one block; bitonic sort in LDS)
__global__ __launch_bounds__(1024) void partial_targets_k(uint32_t world, const int64_t* __restrict__ allmeta, int64_t* __restrict__ tgt) {
  __shared__ int64_t v[4096];
  const uint32_t n = world * PM_SAMPLES;  // <= 4096
  uint32_t m = 1;
  while (m < n) m <<= 1;
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) v[i] = i < n ? allmeta[(size_t)(i / PM_SAMPLES) * PM_META + (i % PM_SAMPLES)] : PM_INF;
  __syncthreads();
  for (uint32_t k = 2; k <= m; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const int64_t a = v[i], b = v[l];
          if (((i & k) == 0) == (a > b)) v[i] = b, v[l] = a;
        }
      }
      __syncthreads();
    }
  __shared__ uint32_t nvalid;
  if (threadIdx.x == 0) {
    uint32_t lo = 0, hi = n;  // first index holding PM_INF
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (v[mid] < PM_INF) lo = mid + 1;
      else hi = mid;
    }
    nvalid = lo;
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x + 1; j < world; j += blockDim.x) tgt[j - 1] = nvalid ? v[(size_t)((uint64_t)j * nvalid / world)] : PM_INF;
}

// One block per cut: the local bundles around the target.  A group opens a local bundle when it starts beyond the running maximum
// of the ends before it (key[i] > emax[i - 1]).  Out: [0] the farthest end of what lies before the first bundle head at or behind the
// target, [1] the horizon — the start of the first local bundle this list does NOT describe (PM_INF: there is none) —, then PM_B
// pairs (start of a bundle, farthest end of everything up to its last group).
__global__ __launch_bounds__(256) void partial_cands_k(uint32_t ng, const int64_t* __restrict__ key, const int64_t* __restrict__ emax,
                                                       const int64_t* __restrict__ tgt, int64_t* __restrict__ cand) {
  const uint32_t c = blockIdx.x;
  int64_t* out = cand + (size_t)c * PM_CAND;
  __shared__ uint32_t heads[PM_B + 1];
  __shared__ uint32_t nheads, idx0s;
  if (threadIdx.x == 0) {
    const int64_t t = tgt[c];
    uint32_t lo = 0, hi = ng;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (key[mid] < t) lo = mid + 1;
      else hi = mid;
    }
    idx0s = lo;
    nheads = 0;
  }
  __syncthreads();
  const uint32_t idx0 = idx0s;
  constexpr uint32_t LIMIT = 1u << 22;  // groups looked at before the list gives up: what lies beyond is behind the horizon
  uint32_t base = idx0;
  bool hit_limit = false;
  while (base < ng) {
    if (base - idx0 >= LIMIT) {
      hit_limit = true;
      break;
    }
    const uint32_t i = base + threadIdx.x;
    const bool head = i < ng && (i == 0 || key[i] > emax[i - 1]);
    // the block's heads in index order: ballots per wave, waves in turn
    const uint64_t bal = __ballot(head);
    __shared__ uint32_t wcnt[4];
    if ((threadIdx.x & 63u) == 0) wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = nheads;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wcnt[w];
    if (head) {
      const uint32_t r = before + (uint32_t)__popcll(bal & ((1ull << (threadIdx.x & 63u)) - 1ull));
      if (r <= PM_B) heads[r] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) nheads += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
    if (nheads > PM_B) break;
    base += 256;
  }
  if (threadIdx.x == 0) {
    const uint32_t nh = nheads < PM_B + 1 ? nheads : PM_B + 1;
    const uint32_t h1 = nh ? heads[0] : ng;  // everything before the first head is one stretch with the part before the target
    out[0] = h1 ? emax[h1 - 1] : (int64_t)-1;
    // the horizon: the (PM_B + 1)-th head if the walk saw one, the key where the walk gave up, or nothing left at all
    int64_t hz = PM_INF;
    if (nh == PM_B + 1) hz = key[heads[PM_B]];
    else if (hit_limit) hz = key[base];
    out[1] = hz;
    const uint32_t last_known = nh == PM_B + 1 ? heads[PM_B] : (hit_limit ? base : ng);  // one past the last group the list covers
    for (uint32_t b = 0; b < PM_B; ++b) {
      if (b < nh && b < PM_B) {
        const uint32_t nexth = (b + 1 < nh) ? heads[b + 1] : last_known;
        out[2 + 2 * b] = key[heads[b]];
        out[3 + 2 * b] = emax[nexth - 1];
      } else {
        out[2 + 2 * b] = PM_INF;
        out[3 + 2 * b] = -1;
      }
    }
  }
}

// One block per cut: the smallest key that is a clean cut for EVERY rank — no group of any rank that starts before it ends at or behind
// it — among the target and the bundle starts the ranks listed, judged from the lists alone.  A candidate beyond some rank's horizon
// cannot be judged and is skipped; when nothing qualifies the cut stays unsettled (flag bit 1) and the caller walks it the slow way.
__global__ __launch_bounds__(256) void partial_choose_k(uint32_t world, const int64_t* __restrict__ allcand /* [world][world - 1][PM_CAND] */,
                                                        const int64_t* __restrict__ tgt, int64_t* __restrict__ cuts, uint32_t* __restrict__ flags) {
  const uint32_t c = blockIdx.x, nc = world - 1;
  const int64_t t = tgt[c];
  __shared__ unsigned long long best;
  if (threadIdx.x == 0) best = ~0ull;
  __syncthreads();
  const uint32_t ncand = 1 + world * PM_B;
  for (uint32_t q = threadIdx.x; q < ncand; q += blockDim.x) {
    int64_t x;
    if (q == 0) x = t;
    else x = allcand[((size_t)((q - 1) / PM_B) * nc + c) * PM_CAND + 2 + 2 * ((q - 1) % PM_B)];
    if (x >= PM_INF || x < t) continue;
    bool clean = true;
    for (uint32_t r = 0; r < world && clean; ++r) {
      const int64_t* L = allcand + ((size_t)r * nc + c) * PM_CAND;
      if (x > L[1]) {  // beyond what rank r described
        clean = false;
        break;
      }
      int64_t e = L[0];  // farthest end of rank r's groups that start before x (an upper bound: running maxima)
      for (uint32_t b = 0; b < PM_B; ++b)
        if (L[2 + 2 * b] < x) e = L[3 + 2 * b];
      clean = e < x;
    }
    if (clean) atomicMin(&best, (unsigned long long)x);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (t >= PM_INF) cuts[c] = PM_INF;  // (no samples anywhere: nothing to cut)
    else if (best == ~0ull) {
      cuts[c] = PM_INF;
      atomicOr(flags, 2u);
    } else cuts[c] = (int64_t)best;
  }
}

// every settled cut is clean, and so is any later cut in the place of an earlier one: the cuts in ascending order (the lists of two
// cuts may judge one candidate differently, both conservatively)
__global__ void partial_cuts_sorted_k(uint32_t nc, int64_t* __restrict__ cuts) {
  if (blockIdx.x || threadIdx.x) return;
  for (uint32_t c = nc - 1; c-- > 0;)
    if (cuts[c] > cuts[c + 1]) cuts[c] = cuts[c + 1];
}

// tabx = the table of the exchange ([world][3]) + {flags, n_files, first_fidx, carry}: flags = bit 0 the partials cannot be packed,
// bit 1 a cut is unsettled, bits 8.. the error bits kernels raised
__global__ void partial_tabx_k(uint32_t world, const long long* __restrict__ tab, const int64_t* __restrict__ mymeta, const uint32_t* __restrict__ flags,
                               const uint32_t* __restrict__ derr, long long* __restrict__ tabx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < world * 3) tabx[i] = tab[i];
  if (i == 0) {
    tabx[world * 3 + 0] = (long long)((*flags & 3u) | (uint32_t)(mymeta[PM_SAMPLES + 2] & 1) | (*derr << 8));
    tabx[world * 3 + 1] = mymeta[PM_SAMPLES];
    tabx[world * 3 + 2] = mymeta[PM_SAMPLES + 1];
    tabx[world * 3 + 3] = mymeta[PM_SAMPLES + 3];
  }
}
}  // namespace

extern "C" int tbk_partial_stage_keys(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* g, int64_t* key, int64_t* emax, uint32_t first_fidx,
                                      int64_t carry, int64_t* meta) {
  if (!ctx || !in || !g || !meta) return TBK_EINVAL;
  if (in->mem != TBK_MEM_DEVICE || g->mem != TBK_MEM_DEVICE) return TBK_EINVAL;
  const uint32_t ng = g->n_groups;
  if (ng && (!key || !emax || !g->rep || !g->yc || !g->yx || !g->g_start || !g->g_end || !in->tid)) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)ng * 8 + ((size_t)1 << 20)));
  uint64_t* sc = ctx->d_scalars;
  TBK_HIP(hipMemsetAsync(sc, 0, 16 * sizeof(uint64_t), ctx->stream));
  if (ng) {
    TBK_LAUNCH(ctx, "partial_keys", partial_keys_k, cdiv(ng, SH_B), SH_B, 0, ng, g->rep, in->tid, g->g_start, g->yc, g->yx, g->g_key, key, (uint32_t*)(sc + 8));
    PmLoad ld{g->rep, in->tid, g->g_end, g->g_key};
    PmStore st{emax};
    TBK_TRY((scan_op_run<PmKey, PmOp, PmLoad, PmStore>(ctx, "partial_emax_scan", ng, ld, st, PmOp{}, PmKey{0u, 0u})));
  }
  TBK_LAUNCH(ctx, "partial_meta", partial_meta_k, 1, 128, 0, ng, key, in->n_files, first_fidx, carry, (const uint32_t*)(sc + 8), meta);
  return tbk_check_launch(ctx, "partial_stage_keys");
}

extern "C" int tbk_partial_stage_cands(tbk_ctx* ctx, const int64_t* key, const int64_t* emax, uint32_t ng, const int64_t* allmeta, uint32_t world,
                                       int64_t* targets, int64_t* cands) {
  if (!ctx || !allmeta || world == 0 || world > 64) return TBK_EINVAL;
  if (world == 1) return 0;
  if (!targets || !cands || (ng && (!key || !emax))) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_LAUNCH(ctx, "partial_targets", partial_targets_k, 1, 1024, 0, world, allmeta, targets);
  TBK_LAUNCH(ctx, "partial_cands", partial_cands_k, world - 1, 256, 0, ng, key, emax, targets, cands);
  return tbk_check_launch(ctx, "partial_stage_cands");
}

extern "C" int tbk_partial_stage_pack(tbk_ctx* ctx, const tbk_collapse_opts* o, const tbk_soa_in* in, const tbk_groups_out* g, const int64_t* key,
                                      const int64_t* mymeta, const int64_t* allcands, const int64_t* targets, uint32_t world, uint32_t first_fidx,
                                      int64_t* cuts, int32_t* rows, uint32_t* cig_out, int64_t* tabx) {
  if (!ctx || !o || !in || !g || !tabx || !mymeta || world == 0 || world > 64) return TBK_EINVAL;
  if (o->strategy < 0 || o->strategy > 3) return TBK_EUNSUPPORTED;
  if (o->strategy == TBK_STRAT_FULL && in->n_records && (!in->md_off || !in->md_has)) return TBK_EINVAL;  // (-L: the MD strings ride along, tbk_partial_pack_md)
  if (in->mem != TBK_MEM_DEVICE || g->mem != TBK_MEM_DEVICE || in->n_files == 0 || in->n_files > 65535 || !in->file_off) return TBK_EINVAL;
  const uint32_t ng = g->n_groups;
  if (ng && (!key || !rows || !cig_out || !g->rep || !g->yc || !g->yx || !g->yd || !g->rep_effend)) return TBK_EINVAL;
  if (world > 1 && (!allcands || !targets || !cuts)) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)ng * 16 + ((size_t)1 << 20)));
  uint64_t* sc = ctx->d_scalars;
  uint32_t* d_flags = (uint32_t*)(sc + 9);  // (word 8 holds stage 1's verdict, the error word is 15: neither is cleared here)
  TBK_HIP(hipMemsetAsync(d_flags, 0, sizeof(uint64_t), ctx->stream));
  if (world > 1) {
    TBK_LAUNCH(ctx, "partial_choose", partial_choose_k, world - 1, 256, 0, world, allcands, targets, cuts, d_flags);
    if (world > 2) TBK_LAUNCH(ctx, "partial_choose", partial_cuts_sorted_k, 1, 1, 0, world - 1, cuts);
  }
  uint32_t* cnt = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  uint32_t* cfirst = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  uint32_t* woff = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  long long* tab = (long long*)ws_alloc<uint64_t>(ctx, (size_t)world * 3);
  if (!woff || !tab) return TBK_ENOMEM;
  uint32_t* d_fo = nullptr;
  TBK_TRY(shard_upload_file_off(ctx, in->file_off, in->n_files, &d_fo));
  const uint64_t* pk_key = o->strategy == TBK_STRAT_FULL ? nullptr : g->g_key;  // (-L: the key describes no alignment by itself, MD is part of it)
  if (ng) TBK_LAUNCH(ctx, "partial_ncig", partial_ncig_k, cdiv(ng, SH_B), SH_B, 0, ng, g->rep, in->cig_off, pk_key, cnt, cfirst);
  TBK_TRY(tbk_exscan_u32(ctx, cnt, woff, ng, sc + 21));
  TBK_LAUNCH(ctx, "partial_table", partial_table_k, cdiv(world, 64), 64, 0, ng, world, key, cuts, woff, sc + 21, tab);
  if (ng) {
    ColIn I{};
    I.n = in->n_records;
    I.k = in->n_files;
    I.file_off = d_fo;
    I.tid = in->tid;
    I.pos = in->pos;
    I.flag = in->flag;
    I.mapq = in->mapq;
    I.strand = in->strand;
    I.nh = in->nh;
    I.cig_off = in->cig_off;
    I.cig = in->cig;
    I.md_off = in->md_off;
    I.md = in->md;
    I.md_has = in->md_has;
    ColOpt O{};
    O.strategy = o->strategy;
    O.max_nh = o->max_nh;
    O.min_qual = o->min_qual;
    O.keep_supp = o->keep_supplementary;
    O.keep_sec = o->keep_secondary;
    O.seed = TBK_KEY_SEED0;
    O.hash_mask = ctx->dbg.hash_mask;
    TBK_LAUNCH(ctx, "partial_rows", partial_rows_k, cdiv(ng, SH_B), SH_B, 0, ng, in->n_files, first_fidx, I, O, g->rep, g->yc, g->yx, g->yd,
               g->rep_effend, cfirst, cnt, woff, pk_key, rows, cig_out, ctx->d_err);
  }
  TBK_LAUNCH(ctx, "partial_tabx", partial_tabx_k, cdiv(world * 3 + 1, 64), 64, 0, world, tab, mymeta, d_flags, ctx->d_err, (long long*)tabx);
  return tbk_check_launch(ctx, "partial_stage_pack");
}

// ---- -L across ranks: the MD strings of the representatives travel beside the rows (ABI 7) -------------------------------------
namespace {
__global__ void partial_mdlen_k(uint32_t ng, const uint32_t* __restrict__ rep, const uint32_t* __restrict__ md_off, const uint8_t* __restrict__ md_has,
                                uint32_t* __restrict__ len) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o > ng) return;
  uint32_t l = 0;
  if (o < ng) {
    const uint32_t r = rep[o];
    l = md_has[r] ? md_off[r + 1] - md_off[r] : 0u;
  }
  len[o] = l;
}
// the bytes in group order; row word 11 = length | "has an MD tag" << 31 (an empty MD string is not an absent one: cmpFull, tiebrush.cpp:285-302)
__global__ void partial_mdcopy_k(uint32_t ng, const uint32_t* __restrict__ rep, const uint32_t* __restrict__ md_off, const uint8_t* __restrict__ md,
                                 const uint8_t* __restrict__ md_has, const uint32_t* __restrict__ moff, int32_t* __restrict__ rows, uint8_t* __restrict__ out) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ng) return;
  const uint32_t r = rep[o];
  const uint32_t has = md_has[r];
  const uint32_t m0 = md_off[r], l = has ? md_off[r + 1] - m0 : 0u;
  rows[(size_t)o * TBK_PARTIAL_ROW + 11] = (int32_t)(l | (has ? 0x80000000u : 0u));
  const uint32_t w = moff[o];
  for (uint32_t q = 0; q < l; ++q) out[w + q] = md[m0 + q];
}
__global__ void partial_mdtab_k(uint32_t world, uint32_t ng, const long long* __restrict__ tab, const uint32_t* __restrict__ moff, const uint64_t* __restrict__ total,
                                long long* __restrict__ md_tab) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= world) return;
  const uint32_t a = (uint32_t)tab[d * 3], b = a + (uint32_t)tab[d * 3 + 1];
  const uint64_t wa = a < ng ? moff[a] : *total, wb = b < ng ? moff[b] : *total;
  md_tab[d] = (long long)(wb - wa);
}
__global__ void partial_mdlen2_k(uint32_t n2, const int32_t* __restrict__ rows, uint32_t* __restrict__ len, uint8_t* __restrict__ has) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n2) return;
  const uint32_t w = j < n2 ? (uint32_t)rows[(size_t)j * TBK_PARTIAL_ROW + 11] : 0u;
  len[j] = w & 0x7FFFFFFFu;
  if (j < n2) has[j] = (uint8_t)(w >> 31);
}
}  // namespace

extern "C" int tbk_partial_pack_md(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* g, const int64_t* tab, uint32_t world, int32_t* rows,
                                   uint8_t* md_out, int64_t* md_tab) {
  if (!ctx || !in || !g || !tab || !md_tab || world == 0) return TBK_EINVAL;
  if (in->mem != TBK_MEM_DEVICE || g->mem != TBK_MEM_DEVICE) return TBK_EINVAL;
  const uint32_t ng = g->n_groups;
  if (ng && (!in->md_off || !in->md_has || !g->rep || !rows || !md_out)) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)ng * 12 + ((size_t)1 << 20)));
  uint32_t* len = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  uint32_t* moff = ws_alloc<uint32_t>(ctx, (size_t)ng + 1);
  if (!len || !moff) return TBK_ENOMEM;
  uint64_t* sc = ctx->d_scalars;
  TBK_LAUNCH(ctx, "partial_md", partial_mdlen_k, cdiv((uint64_t)ng + 1, SH_B), SH_B, 0, ng, g->rep, in->md_off, in->md_has, len);
  TBK_TRY(tbk_exscan_u32(ctx, len, moff, ng + 1, sc + 22));
  if (ng) TBK_LAUNCH(ctx, "partial_md", partial_mdcopy_k, cdiv(ng, SH_B), SH_B, 0, ng, g->rep, in->md_off, in->md, in->md_has, moff, rows, md_out);
  TBK_LAUNCH(ctx, "partial_md", partial_mdtab_k, cdiv(world, 64), 64, 0, world, ng, (const long long*)tab, moff, sc + 22, (long long*)md_tab);
  return tbk_check_launch(ctx, "partial_pack_md");
}

extern "C" int tbk_partial_unpack_md(tbk_ctx* ctx, const int32_t* rows, uint32_t n2, uint32_t* md_off, uint8_t* md_has) {
  if (!ctx || !md_off || (n2 && (!rows || !md_has))) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n2 * 8 + ((size_t)1 << 20)));
  uint32_t* len = ws_alloc<uint32_t>(ctx, (size_t)n2 + 1);
  if (!len) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, "partial_md", partial_mdlen2_k, cdiv((uint64_t)n2 + 1, SH_B), SH_B, 0, n2, rows, len, md_has);
  TBK_TRY(tbk_exscan_u32(ctx, len, md_off, n2 + 1, ctx->d_scalars + 22));
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "partial_unpack_md");
}

extern "C" int tbk_partial_unpack(tbk_ctx* ctx, const int32_t* rows, uint32_t n2, int32_t* tid, int32_t* pos, uint16_t* flag, uint8_t* mapq,
                                  uint8_t* strand, int32_t* nh, uint32_t* cig_off, double* yc_in, int64_t* yx_in, int64_t* yd_in,
                                  int64_t* prio_hi, int64_t* prio_lo) {
  if (!ctx || !cig_off) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n2 * 8 + ((size_t)1 << 20)));
  if (n2 == 0) {
    TBK_HIP(hipMemsetAsync(cig_off, 0, 4, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  if (!rows || !tid || !pos || !flag || !mapq || !strand || !nh || !yc_in || !yx_in || !yd_in || !prio_hi || !prio_lo) return TBK_EINVAL;
  uint32_t* nc = ws_alloc<uint32_t>(ctx, n2);
  if (!nc) return TBK_ENOMEM;
  TBK_LAUNCH(ctx, "partial_ncig2", partial_ncig2_k, cdiv(n2, SH_B), SH_B, 0, n2, rows, nc);
  TBK_TRY(tbk_exscan_u32(ctx, nc, cig_off, n2, ctx->d_scalars + 23));
  TBK_LAUNCH(ctx, "partial_unpack", partial_unpack_k, cdiv(n2, SH_B), SH_B, 0, n2, rows, ctx->d_scalars + 23, tid, pos, flag, mapq, strand, nh,
             cig_off, yc_in, yx_in, yd_in, prio_hi, prio_lo);
  TBK_HIP(hipStreamSynchronize(ctx->stream));
  return tbk_check_launch(ctx, "partial_unpack");
}

// The owner's reduce-by-key of the partials it received (wgroup.hip, tbk_partial_reduce_device): arguments checked here.
extern "C" int tbk_partial_reduce(tbk_ctx* ctx, const tbk_collapse_opts* o, const int32_t* rows, uint32_t n2, const uint32_t* run_off,
                                  uint32_t n_runs, const uint32_t* cig, tbk_groups_out* out, tbk_cov_in* view) {
  if (o && o->strategy == TBK_STRAT_FULL) return TBK_EUNSUPPORTED;  // (-L: tbk_partial_reduce_md, the MD strings must come along)
  return tbk_partial_reduce_md(ctx, o, rows, n2, run_off, n_runs, cig, nullptr, out, view);
}

extern "C" int tbk_partial_reduce_md(tbk_ctx* ctx, const tbk_collapse_opts* o, const int32_t* rows, uint32_t n2, const uint32_t* run_off,
                                     uint32_t n_runs, const uint32_t* cig, const uint8_t* md, tbk_groups_out* out, tbk_cov_in* view) {
  if (!ctx || !o || !out || !run_off || n_runs == 0) return TBK_EINVAL;
  if (o->strategy < 0 || o->strategy > 3) return TBK_EUNSUPPORTED;
  if (run_off[0] != 0 || run_off[n_runs] != n2) return TBK_EINVAL;
  if (out->mem != TBK_MEM_DEVICE || (n2 && (!rows || !out->rep || !out->yc || !out->yx || !out->yd))) return TBK_EINVAL;
  if (!tbk_window_supported(n_runs)) return TBK_EUNSUPPORTED;
  TBK_HIP(hipSetDevice(ctx->device));
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  out->n_groups = 0;
  out->n_passed = n2;
  if (view) {
    memset(view, 0, sizeof(*view));
    view->mem = TBK_MEM_DEVICE;
  }
  if (n2 == 0) return 0;
  TBK_TRY(tbk_ws_reserve(ctx, (size_t)n2 * 128 + ((size_t)8 << 20)));
  uint32_t* md_off = nullptr;
  uint8_t* md_has = nullptr;
  if (o->strategy == TBK_STRAT_FULL) {  // the rows' MD strings: lengths in word 11, bytes in row order
    uint32_t* len = ws_alloc<uint32_t>(ctx, (size_t)n2 + 1);
    md_off = ws_alloc<uint32_t>(ctx, (size_t)n2 + 1);
    md_has = ws_alloc<uint8_t>(ctx, (size_t)n2 + 1);
    if (!len || !md_off || !md_has) return TBK_ENOMEM;
    TBK_LAUNCH(ctx, "partial_md", partial_mdlen2_k, cdiv((uint64_t)n2 + 1, SH_B), SH_B, 0, n2, rows, len, md_has);
    TBK_TRY(tbk_exscan_u32(ctx, len, md_off, n2 + 1, ctx->d_scalars + 22));
  }
  return tbk_partial_reduce_device(ctx, o->strategy, rows, n2, run_off, n_runs, cig, out, view, md_off, md, md_has);
}
