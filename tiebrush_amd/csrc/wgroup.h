// wgroup.h — host interface of the window path (wgroup.hip) used by the collapse pipeline (collapse.hip).
#pragma once
#include <stdlib.h>

#include "strategy.hpp"
#include "tbk_internal.h"

struct WgOut {
  uint32_t ng = 0, np = 0;          // groups; (group, sample) incidences
  uint64_t *ghi = nullptr, *glo = nullptr;  // [ng] group keys, in key order (the order of the sort path's groups)
  uint32_t* gmem = nullptr;         // [ng] the representative record of the group (any member serves the comparators)
  uint32_t* gpoff = nullptr;        // [ng] first incidence of the group
  uint16_t* pfile = nullptr;        // [np] sample (input file) of the incidence; a group's incidences are in file order
  uint32_t* pgrp = nullptr;         // [np] group of the incidence; null when the YD stage places its items by list (tbk_yd_by_list)
  uint64_t* gfmask = nullptr;       // [ng] (<= 64 files and tbk_yd_by_list) the group's files as a bit mask: np = 0, no incidence arrays
  uint32_t* rec_sg = nullptr;       // optional [n]: group (key order) of every passing record, 0xFFFFFFFF otherwise
  // per-group accumulators, as the sort path's reduction leaves them
  double* yc = nullptr;
  uint32_t* ns = nullptr;
  long long *yxin = nullptr, *ydin = nullptr;
  unsigned long long* rep = nullptr;
  uint32_t* first = nullptr;        // == identity: the "sorted arrays" of the later stages are the group arrays
  uint8_t* tie = nullptr;
};

// The caller's result arrays (tbk_groups_out), for the compaction pass to fill in key order — which IS the output order except inside
// a tie set (groups that share bucket, strand and end: the reference comparator orders those, col_tie_sort_k), so the collapse only
// rewrites the members of tie sets afterwards (col_write_k<tied only>) instead of reading every group's accumulators back.
struct WgDirectOut {
  uint32_t cap = 0;  // entries the arrays hold (a tile with more groups is refused by the caller: nothing beyond cap is written)
  uint32_t* rep = nullptr;
  double* yc = nullptr;
  int64_t* yx = nullptr;
  int32_t *g_start = nullptr, *g_end = nullptr, *rep_effend = nullptr;
  uint64_t* g_key = nullptr;
  int strategy = 0;
};

// chi / clo / cval: the compacted passing records (k runs, run f = [run_off[f], run_off[f+1]), device offsets), m of them (host
// value); ceff: the effective end of the k-way merge of every compacted record (or, for cross-rank tiles, the low word of the
// explicit merge priority); scratch_hi / scratch_lo: two dead 8-byte-per-record arrays (>= m) to work in.
// Returns 0 and fills *out (arrays from ctx's workspace), or a TBK status; TBK_DERR_BIGBUCKET / _COLLISION come back in *err_bits
// (the counts in *out are then meaningless).  The last kernel queued (wg_finish) may still raise TBK_DERR_COLLISION in ctx->d_err:
// the caller's next read-back must treat it as a reseed request.
int tbk_window_groups(tbk_ctx* ctx, const tbkd::ColIn& I, int strategy, const uint64_t* chi, const uint64_t* clo, const uint32_t* cval,
                      const uint32_t* ceff, uint32_t m, const uint32_t* d_run_off, uint64_t* scratch_hi, uint64_t* scratch_lo,
                      bool want_rec_sg, uint64_t seed, WgOut* out, uint32_t* err_bits, const tbkd::ColOpt* raw_opt = nullptr, bool part = false,
                      const WgDirectOut* direct = nullptr /* RAW only */);
// raw_opt != nullptr — RAW mode: the windows are cut on the input records themselves (chi .. ceff and the scratch arrays are
// unused and may be null, m = I.n, d_run_off = I.file_off on the device); keys, the filter and the effective ends are computed
// inside the window kernels, the number of passing records is added to ctx->d_scalars[0] (zeroed by the caller).
// TBK_DERR_RAWORDER in *err_bits: the input is not of the shape this mode takes — run the general path.
// part (RAW only, k <= 64, every file TieBrush-merged, explicit priorities): the records are group partials of other ranks
//  — WgOut::yc is the sum of the carried integral YC, yxin / ydin the sum / maximum of the carried YX / YD, no
// incidences (np = 0); TBK_DERR_FRACTIONAL in *err_bits: a carried value this form cannot hold — run the sort path.
bool tbk_window_supported(uint32_t k);
// The YD stage of a window-path tile places its items by list without a sort when the tile has at most 64 input files (collapse.hip:
// yd_lcount_k / yd_lscatter_k) and needs no per-incidence group array then (yd_radix: test hook, the radix split for any tile).
inline bool tbk_yd_by_list(const tbk_ctx* ctx, uint32_t k) { return 2u * k <= 128u && !ctx->dbg.yd_radix; }

// Owner side of the group-partials protocol: merge n_runs runs of partial rows (TBK_PARTIAL_ROW words each, every
// run in its rank's output order) in output order and reduce equal keys; see tbk_partial_reduce in include/tbk.h.  The arena
// must be reserved by the caller.
int tbk_partial_reduce_device(tbk_ctx* ctx, int strategy, const int32_t* rows, uint32_t n2, const uint32_t* run_off_host, uint32_t n_runs,
                              const uint32_t* cig, tbk_groups_out* out, tbk_cov_in* view, const uint32_t* md_off = nullptr,
                              const uint8_t* md = nullptr, const uint8_t* md_has = nullptr);  // (-L: the rows' MD strings as CSR)
