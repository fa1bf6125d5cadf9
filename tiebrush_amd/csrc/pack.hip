// pack.hip — the packed wire form of a tile (include/tbk.h: tbk_packed_in, ABI version 5) and its expansion on the device.
//
// A tile that starts in host memory crosses PCIe before anything else happens to it, and the structure of arrays of tbk_soa_in is
// 20 bytes per record plus the CIGAR words (tid, pos, flag, mapq, strand, NH, CIGAR offset): the link, not the GPU, sets the pace of
// the host -> host path (config 3: 8.7 GB in, 1.1 GB out, 21 ms of kernels).  The packed form carries the same records in 9 bytes
// plus the CIGAR words: the reference id as runs (a coordinate-sorted file changes it a handful of times), flag / strand / MAPQ / NH
// in one word, the CIGAR length as a byte instead of an offset.  tbk_unpack_tile copies the packed arrays and rebuilds the
// structure of arrays in context-owned device memory (pos and the CIGAR words land where they stay; one kernel writes the other
// columns, one scan the CIGAR offsets): the tile it returns goes to tbk_collapse_tile as any device-resident tile does.
// The reference has no counterpart (its records never leave the host: GSam.h:506-516, tmerge.cpp:331-344).
#include "dev_common.hpp"
#include "tbk_internal.h"

namespace {

__global__ void unpack_k(uint32_t n, const uint32_t* __restrict__ meta, const uint8_t* __restrict__ ncig, const uint32_t* __restrict__ run_end,
                         const int32_t* __restrict__ run_tid, uint32_t n_runs, int32_t* __restrict__ tid, uint16_t* __restrict__ flag,
                         uint8_t* __restrict__ mapq, uint8_t* __restrict__ strand, int32_t* __restrict__ nh, uint32_t* __restrict__ cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t m = meta[i];
  flag[i] = (uint16_t)(m & 0xFFFu);
  const uint32_t sc = (m >> 12) & 3u;
  strand[i] = sc == 0u ? (uint8_t)'+' : (sc == 1u ? (uint8_t)'-' : (uint8_t)'.');
  mapq[i] = (uint8_t)((m >> 14) & 0xFFu);
  const uint32_t nc = m >> 22;
  nh[i] = nc == 1022u ? TBK_NH_ABSENT : (int32_t)nc;  // (1023: the escape list overwrites it)
  cnt[i] = ncig[i];                                     // (255: the escape list overwrites it)
  uint32_t lo = 0, hi = n_runs;                         // first run that ends beyond i
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (run_end[mid] > i)
      hi = mid;
    else
      lo = mid + 1;
  }
  tid[i] = lo < n_runs ? run_tid[lo] : -1;
}
__global__ void unpack_esc_k(uint32_t n_nh, const uint32_t* __restrict__ nh_idx, const int32_t* __restrict__ nh_val, uint32_t n_nc,
                             const uint32_t* __restrict__ nc_idx, const uint32_t* __restrict__ nc_val, uint32_t n, int32_t* __restrict__ nh,
                             uint32_t* __restrict__ cnt) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n_nh && nh_idx[q] < n) nh[nh_idx[q]] = nh_val[q];
  if (q < n_nc && nc_idx[q] < n) cnt[nc_idx[q]] = nc_val[q];
}

}  // namespace

extern "C" int tbk_unpack_tile(tbk_ctx* ctx, const tbk_packed_in* in, tbk_soa_in* tile) {
  if (!ctx || !in || !tile) return TBK_EINVAL;
  if (in->n_files == 0 || in->n_files > 65535 || !in->file_off || in->file_off[0] != 0 || in->file_off[in->n_files] != in->n_records) return TBK_EINVAL;
  const size_t n = in->n_records, nc = in->n_cigar_ops;
  if (n && (!in->pos || !in->meta || !in->ncig || (nc && !in->cig) || !in->tid_run_end || !in->tid_run_tid || in->n_tid_runs == 0)) return TBK_EINVAL;
  if ((in->n_nh_esc && (!in->nh_esc_idx || !in->nh_esc_val)) || (in->n_ncig_esc && (!in->ncig_esc_idx || !in->ncig_esc_val))) return TBK_EINVAL;
  if (n && in->tid_run_end[in->n_tid_runs - 1] != in->n_records) return TBK_EINVAL;
  TBK_HIP(hipSetDevice(ctx->device));
  // a deferred YD stage of the previous collapse still reads the tile this call is about to overwrite (or free): wait for it first
  TBK_TRY(tbk_collapse_finish_yd(ctx));
  memset(tile, 0, sizeof(*tile));
  tile->mem = TBK_MEM_DEVICE;
  tile->n_files = in->n_files;
  tile->n_records = in->n_records;
  tile->n_cigar_ops = in->n_cigar_ops;
  tile->file_off = in->file_off;
  ctx->unpack_tbm.assign(in->n_files, 0);
  tile->tbmerged = ctx->unpack_tbm.data();
  if (n == 0) return 0;
  // the unpacked tile: context-owned, grown when a larger tile comes, valid until the next tbk_unpack_tile / tbk_destroy
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t need = al(n * 4) * 3 + al(n * 2) + al(n) * 2 + al((n + 1) * 4) + al(nc * 4 + 4);
  if (need > ctx->d_unpack_cap) {
    if (ctx->d_unpack) (void)hipFree(ctx->d_unpack);
    ctx->d_unpack = nullptr;
    ctx->d_unpack_cap = 0;
    const size_t cap = need + need / 16;
    TBK_HIP(hipMalloc((void**)&ctx->d_unpack, cap));
    ctx->d_unpack_cap = cap;
  }
  char* p = ctx->d_unpack;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += al(bytes);
    return r;
  };
  int32_t* d_tid = (int32_t*)take(n * 4);
  int32_t* d_pos = (int32_t*)take(n * 4);
  int32_t* d_nh = (int32_t*)take(n * 4);
  uint16_t* d_flag = (uint16_t*)take(n * 2);
  uint8_t* d_mapq = (uint8_t*)take(n);
  uint8_t* d_strand = (uint8_t*)take(n);
  uint32_t* d_cig_off = (uint32_t*)take((n + 1) * 4);
  uint32_t* d_cig = (uint32_t*)take(nc * 4 + 4);
  tbk_prof_begin_call(ctx);
  struct ProfEnd {
    tbk_ctx* c;
    ~ProfEnd() { tbk_prof_end_call(c); }
  } prof_end{ctx};
  const size_t esc = (size_t)in->n_nh_esc * 8 + (size_t)in->n_ncig_esc * 8;
  TBK_TRY(tbk_ws_reserve(ctx, n * 9 + (size_t)in->n_tid_runs * 8 + esc + ((size_t)4 << 20)));
  uint32_t* d_meta = ws_alloc<uint32_t>(ctx, n);
  uint8_t* d_ncig = ws_alloc<uint8_t>(ctx, n);
  uint32_t* d_cnt = ws_alloc<uint32_t>(ctx, n);
  uint32_t* d_rend = ws_alloc<uint32_t>(ctx, in->n_tid_runs);
  int32_t* d_rtid = ws_alloc<int32_t>(ctx, in->n_tid_runs);
  if (!d_meta || !d_ncig || !d_cnt || !d_rend || !d_rtid) return TBK_ENOMEM;
  hipStream_t s = ctx->stream;
  TBK_HIP(hipMemcpyAsync(d_pos, in->pos, n * 4, hipMemcpyHostToDevice, s));
  TBK_HIP(hipMemcpyAsync(d_meta, in->meta, n * 4, hipMemcpyHostToDevice, s));
  TBK_HIP(hipMemcpyAsync(d_ncig, in->ncig, n, hipMemcpyHostToDevice, s));
  if (nc) TBK_HIP(hipMemcpyAsync(d_cig, in->cig, nc * 4, hipMemcpyHostToDevice, s));
  TBK_HIP(hipMemcpyAsync(d_rend, in->tid_run_end, (size_t)in->n_tid_runs * 4, hipMemcpyHostToDevice, s));
  TBK_HIP(hipMemcpyAsync(d_rtid, in->tid_run_tid, (size_t)in->n_tid_runs * 4, hipMemcpyHostToDevice, s));
  const uint32_t B = 256;
  TBK_LAUNCH(ctx, "unpack", unpack_k, cdiv(n, B), B, 0, (uint32_t)n, d_meta, d_ncig, d_rend, d_rtid, in->n_tid_runs, d_tid, d_flag, d_mapq, d_strand, d_nh,
             d_cnt);
  const uint32_t ne = in->n_nh_esc > in->n_ncig_esc ? in->n_nh_esc : in->n_ncig_esc;
  if (ne) {
    uint32_t* e_nhi = ws_alloc<uint32_t>(ctx, in->n_nh_esc);
    int32_t* e_nhv = ws_alloc<int32_t>(ctx, in->n_nh_esc);
    uint32_t* e_nci = ws_alloc<uint32_t>(ctx, in->n_ncig_esc);
    uint32_t* e_ncv = ws_alloc<uint32_t>(ctx, in->n_ncig_esc);
    if (!e_nhi || !e_nhv || !e_nci || !e_ncv) return TBK_ENOMEM;
    if (in->n_nh_esc) {
      TBK_HIP(hipMemcpyAsync(e_nhi, in->nh_esc_idx, (size_t)in->n_nh_esc * 4, hipMemcpyHostToDevice, s));
      TBK_HIP(hipMemcpyAsync(e_nhv, in->nh_esc_val, (size_t)in->n_nh_esc * 4, hipMemcpyHostToDevice, s));
    }
    if (in->n_ncig_esc) {
      TBK_HIP(hipMemcpyAsync(e_nci, in->ncig_esc_idx, (size_t)in->n_ncig_esc * 4, hipMemcpyHostToDevice, s));
      TBK_HIP(hipMemcpyAsync(e_ncv, in->ncig_esc_val, (size_t)in->n_ncig_esc * 4, hipMemcpyHostToDevice, s));
    }
    TBK_LAUNCH(ctx, "unpack", unpack_esc_k, cdiv(ne, B), B, 0, in->n_nh_esc, e_nhi, e_nhv, in->n_ncig_esc, e_nci, e_ncv, (uint32_t)n, d_nh, d_cnt);
  }
  TBK_TRY(tbk_exscan_u32(ctx, d_cnt, d_cig_off, (uint32_t)n, ctx->d_scalars + 28));
  TBK_HIP(hipMemcpyAsync(d_cig_off + n, ctx->d_scalars + 28, 4, hipMemcpyDeviceToDevice, s));  // (little endian: the low word of the total)
  TBK_HIP(hipMemcpyAsync(ctx->h_scalars + 28, ctx->d_scalars + 28, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  TBK_HIP(hipStreamSynchronize(s));
  if (ctx->h_scalars[28] != (uint64_t)nc) {
    ctx->last_error = "tbk_unpack_tile: the CIGAR counts do not add up to n_cigar_ops";
    return TBK_EINVAL;
  }
  tile->tid = d_tid;
  tile->pos = d_pos;
  tile->flag = d_flag;
  tile->mapq = d_mapq;
  tile->strand = d_strand;
  tile->nh = d_nh;
  tile->cig_off = d_cig_off;
  tile->cig = d_cig;
  return tbk_check_launch(ctx, "unpack");
}
