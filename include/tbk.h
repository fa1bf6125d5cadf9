/* tbk.h — C ABI of the MI355X (gfx950) tiebrush/tiecov hot path.
 *
 * This is the drop-in boundary: a host program that has decoded
 * BAM records into a structure-of-arrays tile calls these entry points in place
 * of the reference's per-record loops.  Plain C structs of pointers + counts, no
 * C++/torch types.  Every buffer is owned by the caller; pointers are either
 * host pointers (TBK_MEM_HOST: staged through the context's pinned buffers) or
 * device pointers (TBK_MEM_DEVICE: used in place, resident in HBM).
 *
 * Reference interfaces each entry point replaces (all under /root/reference/src):
 *   tbk_collapse_tile  <-  the main loop of tiebrush:  TInputFiles::next()
 *                          (tmerge.cpp:331-344, order tmerge.h:28-50),
 *                          passes_options (tiebrush.cpp:532-541), addPData
 *                          (:477-499), SPData::settle/dupAdd (:378-436),
 *                          flushPData incl. the GSegList YD machine (:501-530,
 *                          :111-250), GSamRecord::setupCoordinates
 *                          (GSam.cpp:351-417).
 *   tbk_coverage_tile  <-  the main loop of tiecov: bundle logic
 *                          (tiecov.cpp:443-481), addCov (:194-223),
 *                          flushCoverage (:226-241), addJunction/flushJuncs
 *                          (:100-120).
 *   tbk_sample_tile    <-  tiecov -s: addMean (:155-185), discretize/normalize/
 *                          flushCoverage(pair) (:277-323).
 *
 * Return value: 0 on success, negative tbk_status otherwise.  Never throws,
 * never exits.  A context is not thread-safe; use one per host thread.
 */
#ifndef TBK_H_
#define TBK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBK_ABI_VERSION 8

typedef struct tbk_ctx tbk_ctx;

typedef enum tbk_status {
  TBK_OK = 0,
  TBK_EINVAL = -1,       /* bad argument / inconsistent sizes                        */
  TBK_ENOMEM = -2,       /* device or pinned allocation failed                       */
  TBK_EHIP = -3,         /* a HIP runtime call or kernel failed (see tbk_last_error) */
  TBK_E2BIG = -4,        /* output capacity too small; required sizes are reported   */
  TBK_EUNSUPPORTED = -5, /* option outside the pinned semantics (-F, -M)             */
  TBK_EUNSORTED = -6,    /* an input file is not (tid,pos)-sorted                    */
  TBK_EFATALOP = -7,     /* tiecov: CIGAR op other than M/I/D/N/S (tiecov.cpp:219)   */
  TBK_ECOLLISION = -8,   /* internal: key-hash collision survived all reseeds        */
  TBK_ENODEVICE = -9     /* no gfx950 device / device ordinal out of range           */
} tbk_status;

/* tiebrush.cpp:82-87 (values differ from the reference enum on purpose: ABI-stable) */
typedef enum tbk_strategy {
  TBK_STRAT_CIGAR = 0, /* default: same CIGAR                       (cmpCigar     :304-310) */
  TBK_STRAT_FULL = 1,  /* -L: same CIGAR and MD                     (cmpFull      :285-302) */
  TBK_STRAT_CLIP = 2,  /* -P: same CIGAR after soft-clip stripping  (cmpCigarClip :312-332) */
  TBK_STRAT_EXON = 3   /* -E: same exon coordinates                 (cmpExons     :334-345) */
} tbk_strategy;

typedef enum tbk_mem {
  TBK_MEM_HOST = 0,
  TBK_MEM_DEVICE = 1,
  TBK_MEM_KEPT = 2 /* (ABI version 8) tbk_enc_in only: the results the context kept from its last collapse (tbk_collapse_opts.keep_results) */
} tbk_mem;

#define TBK_NH_ABSENT INT32_MIN /* record has no NH tag */

/* Options of the collapse path = struct Options (tiebrush.cpp:89-98) + mrgStrategy. */
typedef struct tbk_collapse_opts {
  int32_t strategy;           /* tbk_strategy                                         */
  int32_t max_nh;             /* -N, default INT32_MAX                                */
  int32_t min_qual;           /* -Q, default -1                                       */
  uint32_t flags_mask;        /* -F, must be 0 (semantics unpinned)   */
  uint8_t keep_supplementary; /* -S                                                   */
  uint8_t keep_secondary;     /* --keep-secondary                                     */
  uint8_t keep_unmapped;      /* -M, must be 0                     */
  uint8_t collapse_same;      /* -A: needs qname_hash                                 */
  uint8_t store_frac;         /* --store-frac: YC += 1/NH                             */
  uint8_t defer_yd;           /* TBK_MEM_DEVICE only: return as soon as rep/yc/yx/coordinates are final and compute
                                 the YD column on a side stream; tbk_collapse_finish_yd() completes out->yd.  Lets
                                 the caller overlap the (YD-independent) tiecov chain with the YD list machine.   */
  uint8_t keep_results;       /* (ABI version 8; was reserved, 0) the context keeps a device copy of the call's rep / yc / yx / yd until its
                                 next tbk_collapse_tile: tbk_bam_encode reads them there (tbk_enc_in.mem = TBK_MEM_KEPT) and
                                 tbk_kept_results hands any range of them out.  With it a TBK_MEM_HOST caller may leave out->rep / yc /
                                 yx / yd NULL: what the output side needs on the host is `rep` (to gather the records it decoded
                                 itself) — fetched with tbk_kept_results once n_groups says how long it is —, the tags' values never
                                 cross the link.  Not with defer_yd.                                                            */
  uint8_t reserved[1];
} tbk_collapse_opts;

/* One tile of decoded records, file-major: records of input file f occupy
 * [file_off[f], file_off[f+1]) in file order.  file_off and tbmerged are always
 * HOST arrays (tiny); every other pointer lives in `mem`. */
typedef struct tbk_soa_in {
  int32_t mem;              /* tbk_mem                                               */
  uint32_t n_files;
  uint32_t n_records;
  uint32_t n_cigar_ops;     /* == cig_off[n_records]                                 */
  const uint32_t* file_off; /* HOST [n_files+1]                                      */
  const uint8_t* tbmerged;  /* HOST [n_files] 1 = written by TieBrush (tmerge.cpp:70-77) */
  const int32_t* tid;       /* refID                                                 */
  const int32_t* pos;       /* 0-based leftmost position (bam core.pos)              */
  const uint16_t* flag;
  const uint8_t* mapq;
  const uint8_t* strand;    /* '+','-','.' = GSamRecord::spliceStrand (GSam.cpp:464-475) */
  const int32_t* nh;        /* NH value or TBK_NH_ABSENT                             */
  const uint32_t* cig_off;  /* [n_records+1] CSR offsets into cig                    */
  const uint32_t* cig;      /* [n_cigar_ops] BAM encoding len<<4|op                  */
  /* only read for records of tbmerged files; may be NULL when no file is tbmerged */
  const double* yc_in;      /* tag_float("YC"), 0.0 if absent                        */
  const int64_t* yx_in;     /* tag_int("YX",1)                                       */
  const int64_t* yd_in;     /* tag_int("YD",0)                                       */
  /* only for TBK_STRAT_FULL: MD:Z payload without NUL; md_off[i]==md_off[i+1] and md_has[i]==0 when absent */
  const uint32_t* md_off;   /* [n_records+1]                                         */
  const uint8_t* md;
  const uint8_t* md_has;    /* [n_records]                                           */
  /* only for collapse_same (-A): 64-bit hash of (qname bytes, pairOrder) — a filter only: equal hashes are confirmed by a
   * byte compare of the names below and of pairOrder (flag 0x40 / 0x80), as the reference's strcmp does (tiebrush.cpp:422-424) */
  const uint64_t* qname_hash;
  /* only when stitching partial groups of several ranks: explicit merge-order priority of
   * each record; the representative of a group is then argmin (prio_hi, prio_lo) over its members.
   * prio_hi = effective end of the partial's representative, prio_lo = (global file index << 32) | index in file */
  const uint64_t* prio_hi;
  const uint64_t* prio_lo;
  /* only for collapse_same (-A), required with it: read names without NUL as CSR (ABI version 2: fields appended) */
  const uint32_t* qname_off; /* [n_records+1]                                        */
  const uint8_t* qname;
} tbk_soa_in;

/* Collapsed groups in the order the reference writes them (flushPData order). */
typedef struct tbk_groups_out {
  int32_t mem;           /* tbk_mem of the arrays below                               */
  uint32_t cap_groups;   /* capacity of each array                                    */
  uint32_t* rep;         /* index (into the tile) of the representative record        */
  double* yc;            /* accYC (double); the YC tag is (float)yc                   */
  int64_t* yx;           /* accYX + popcount(samples)                                 */
  int32_t* yd;           /* final YD (0 = tag removed)                                */
  int32_t* g_start;      /* optional (may be NULL): 1-based start of the group        */
  int32_t* g_end;        /* optional: 1-based end                                     */
  int32_t* rec_group;    /* optional [n_records]: output index of each record's group, -1 = filtered */
  int32_t* rep_effend;   /* optional: merge-order key of the representative = running max of `end` in its file
                            (tmerge.h:28-50); what a cross-rank stitch needs to pick the global representative */
  uint64_t* g_key;       /* optional [2 * cap_groups]: where the group lies and, when it is that simple, what its alignment looks like —
                            word 0 = tid + 1 : 31 | 1-based start : 31 | strand code : 2 ('+' 0, '-' 1, '.' 2), word 1 = span : 32 |
                            shape : 32, shape = 0x80000000 (one M of `span` bases, soft clips aside), 0xC0000000 | a << 20 | g (a M, g N,
                            span - a - g M), or 0 (anything else: look at the representative).  With it tbk_groups_to_cov_in builds
                            the tiecov input of the representatives without fetching them (GSam.cpp:351-417 gives the same exons) */
  uint32_t n_groups;     /* written by the callee = outCounter (tiebrush.cpp:528)     */
  uint32_t n_passed;     /* written by the callee = inCounter  (tiebrush.cpp:573)     */
} tbk_groups_out;

/* Input of the coverage path: records of ONE collapsed (or plain) BAM in file order. */
typedef struct tbk_cov_in {
  int32_t mem;
  uint32_t n_records;
  uint32_t n_cigar_ops;
  const int32_t* tid;
  const int32_t* pos;
  const uint16_t* flag;    /* only bit 0x4 (unmapped: record skipped) is read; NULL = every record counts */
  const uint32_t* cig_off;
  const uint32_t* cig;
  const double* yc;        /* YC tag as double, 1.0 when absent (tiecov.cpp:482-485)  */
  const uint8_t* strand;   /* spliceStrand; only read when junctions are requested    */
  const int64_t* yx;       /* tag_int("YX",1); only read by tbk_sample_tile           */
} tbk_cov_in;

typedef struct tbk_cov_out {
  int32_t mem;
  uint32_t cap_intervals;  /* 0 = coverage not requested                              */
  int32_t* iv_tid;         /* bedgraph rows: tid, 0-based start, end (exclusive), value */
  int32_t* iv_start;
  int32_t* iv_end;
  double* iv_val;
  uint32_t cap_junctions;  /* 0 = junctions not requested                             */
  int32_t* j_tid;          /* junction rows in flushJuncs order; row i is JUNC%08d i+1 */
  int32_t* j_start;        /* already start-1 (BED)                                   */
  int32_t* j_end;
  uint8_t* j_strand;
  double* j_val;
  uint32_t n_intervals;    /* written by the callee                                   */
  uint32_t n_junctions;    /* written by the callee                                   */
  uint64_t n_bases;        /* written by the callee: sum of M-op lengths of mapped records */
  uint64_t span_bases;     /* written by the callee: sum of bundle spans              */
} tbk_cov_out;

typedef struct tbk_sample_out {
  int32_t mem;
  uint32_t cap_intervals;
  int32_t* iv_tid;
  int32_t* iv_start;
  int32_t* iv_end;
  int64_t* iv_count;       /* ceil(running mean of YX)                                */
  float* iv_heat;          /* normalised to [0.1,1.5] by num_samples (tiecov.cpp:316-323) */
  uint32_t n_intervals;
} tbk_sample_out;

/* Per-kernel timing of the most recent call (HIP events on the context's stream;
 * only recorded when profiling was enabled with tbk_set_profiling). */
typedef struct tbk_kernel_time {
  const char* name; /* static string */
  float ms;         /* summed over launches of that kernel in the call */
  uint32_t launches;
} tbk_kernel_time;

/* ---- lifetime ---------------------------------------------------------------- */
int tbk_abi_version(void);
int tbk_create(int device_ordinal, tbk_ctx** out);
void tbk_destroy(tbk_ctx* ctx);
const char* tbk_strerror(int status);
const char* tbk_last_error(const tbk_ctx* ctx); /* detail of the last TBK_EHIP etc. */

/* Use an external HIP stream (e.g. torch's current stream) instead of the
 * context's own; NULL restores the internal one; TBK_STREAM_DEFAULT selects the
 * device's default stream (whose handle is the null pointer). */
#define TBK_STREAM_DEFAULT ((void*)1)
int tbk_set_stream(tbk_ctx* ctx, void* hip_stream);
void* tbk_get_stream(tbk_ctx* ctx);
int tbk_set_profiling(tbk_ctx* ctx, int enabled);
/* Test hooks and forced path choices of a context: "key=value,key=value" (NULL or "": the defaults).  A context starts with what
 * the environment variable TBK_DEBUG held when it was created — the only time the library looks at the environment; the keys are
 * listed with struct TbkDebug in tiebrush_amd/csrc/tbk_internal.h (path, raw, sort, scan, hash_mask, yd_wave_min, cov_legacy, ...). */
int tbk_set_debug(tbk_ctx* ctx, const char* spec);
int tbk_kernel_times(tbk_ctx* ctx, tbk_kernel_time* out, int cap); /* returns count */

/* Pinned host memory helpers for TBK_MEM_HOST callers. */
int tbk_host_alloc(size_t bytes, void** out);
void tbk_host_free(void* p);

/* ---- hot path ---------------------------------------------------------------- */
void tbk_collapse_opts_default(tbk_collapse_opts* o);
int tbk_collapse_tile(tbk_ctx* ctx, const tbk_collapse_opts* opts, const tbk_soa_in* in, tbk_groups_out* out);
/* Waits for a deferred YD stage (no-op when none is pending); returns its status.  The arrays of the deferred call's
 * tbk_soa_in / tbk_groups_out must stay alive until then.  Implied by the next tbk_collapse_tile and by tbk_destroy. */
int tbk_collapse_finish_yd(tbk_ctx* ctx);
/* (ABI version 8) Optional.  Pays what a context's first tbk_collapse_tile otherwise pays inside the call, beyond its kernels: the
 * auxiliary stream of the YD stage (a hardware queue) and the first dispatch of its list machines.  A command line calls it on the
 * thread that brings the device up, beside its decode; a long-lived caller never needs it. */
int tbk_warmup(tbk_ctx* ctx);
int tbk_coverage_tile(tbk_ctx* ctx, const tbk_cov_in* in, tbk_cov_out* out);
int tbk_sample_tile(tbk_ctx* ctx, const tbk_cov_in* in, int32_t num_samples, tbk_sample_out* out);

/* Device-side chaining tiebrush -> tiecov without a host round trip: builds the
 * tbk_cov_in view of the collapsed records (representatives in output order, yc =
 * (double)(float)accYC exactly as the YC:f tag round-trips) in context-owned
 * device memory.  The returned view is valid until the next call on `ctx`. */
int tbk_groups_to_cov_in(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* groups, tbk_cov_in* view);

/* ---- BGZF / BAM decode on the device ---------------------------------------------------------------
 * What GSamReader::next() -> sam_read1() (GSam.h:506-516; htslib bgzf_read_block + inflate) does on the host. */

/* Inflate a run of whole BGZF members (`comp`, host memory: gzip members with the BC extra field, raw deflate, <= 64 KiB
 * payload each) into `out` (host, or device when mem == TBK_MEM_DEVICE).  One GPU lane per member; ISIZE and CRC32 of
 * every member are verified.  *out_bytes = payload size; TBK_E2BIG (with *out_bytes set) when out_cap is too small,
 * TBK_EINVAL for anything htslib would reject. */
int tbk_bgzf_inflate(tbk_ctx* ctx, const uint8_t* comp, uint64_t comp_bytes, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes, int mem);

/* Decode whole BAM files on the device: comp[f] / comp_bytes[f] = the BGZF members of input file f (host memory, the 28-byte
 * EOF member may be included).  Inflate, BAM header skip, record index, field validation (what htslib's bam_read1 rejects is
 * TBK_EINVAL) and the aux scan of the host loader (NH; XS / ts -> spliceStrand, GSam.cpp:464-475; carried YC / YX / YD of
 * the files flagged in tbmerged; MD with want_md; QNAME + its hash with want_names) run as kernels; *tile describes a
 * DEVICE-resident tile in context-owned memory (valid until tbk_bam_release or the next tbk_bam_decode), ready for
 * tbk_collapse_tile; file_off_out[n_files + 1] (host, caller's) receives the record ranges and is what tile->file_off
 * points to.  The inflated records stay on the device for tbk_bam_records. */
int tbk_bam_decode(tbk_ctx* ctx, uint32_t n_files, const uint8_t* const* comp, const uint64_t* comp_bytes, const uint8_t* tbmerged,
                   int want_md, int want_names, tbk_soa_in* tile, uint32_t* file_off_out);
/* The raw records (block_size field first, as in the BAM stream) behind n tile indices (idx in idx_mem), packed in that
 * order into `out` (host); out_off[n + 1] (host) = their byte offsets.  TBK_E2BIG with out_off[n] = needed bytes. */
int tbk_bam_records(tbk_ctx* ctx, const uint32_t* idx, uint32_t n, int idx_mem, uint8_t* out, uint64_t out_cap, uint64_t* out_off);
void tbk_bam_release(tbk_ctx* ctx);
/* (ABI version 6) One tile out of two: the files a host decoder took (host_part: a HOST tile, plain inputs only) behind the files
 * tbk_bam_decode took on this context (dev_part: the tile it returned) — a host and the GPU inflating their shares of the inputs
 * side by side (the reference's a2, GSamReader::next, GSam.h:506-516, is the end-to-end limiter: SURVEY.md §8 f1).  *out is a
 * DEVICE tile in context-owned memory (released with the decoded files), its files dev_part's then host_part's, record indices
 * below dev_part->n_records the ones tbk_bam_records knows; file_off_out[n_files + 1] and tbmerged_out[n_files] are the caller's.
 * TBK_EUNSUPPORTED when either part carries YC / YX / YD, MD or names. */
int tbk_tile_join(tbk_ctx* ctx, const tbk_soa_in* dev_part, const tbk_soa_in* host_part, tbk_soa_in* out, uint32_t* file_off_out,
                  uint8_t* tbmerged_out);
/* (ABI version 6) Size the context's work arena now for a tbk_collapse_tile of a tile of about this shape (the window path and its
 * deferred YD stage): growing the arena is a free and an allocation of gigabytes, ~ 0.1 s that a host can put beside its own
 * decoding instead of inside the collapse call.  Purely an optimisation: a collapse sizes the arena itself when it has to. */
int tbk_reserve_tile(tbk_ctx* ctx, uint64_t n_records, uint64_t n_cigar_ops);

/* ---- BGZF / BAM encode on the device (ABI version 7; SURVEY.md §8 f1, the output side) ------------------------------------------
 * What flushPData's tagging (tiebrush.cpp:506-525: YC:f always, YX:i always, YD:i when positive and removed otherwise;
 * bam_aux_update_float / _int / bam_aux_del of htslib 1.18 through GSam.h:300-305) and GSamWriter::write -> sam_write1 -> bgzf_write
 * (GSam.h:648-653; zlib level 6 inside htslib) do per output record on the host. */

/* Deflate a byte run into whole BGZF members (gzip members with the BC extra field, one dynamic-Huffman or stored block each, CRC32
 * and ISIZE set) written back to back into `out` (HOST).  src lives in src_mem.  Members hold cuts[m + 1] - cuts[m] payload bytes
 * (cuts[n_members + 1] HOST, cuts[0] = 0, cuts[n_members] = n, each piece <= 0xff00) or, with cuts == NULL, 0xff00 bytes each.
 * *out_bytes = size of the run; TBK_E2BIG (with *out_bytes set) when out_cap is too small.  The EOF member is the caller's. */
int tbk_bgzf_deflate(tbk_ctx* ctx, const uint8_t* src, uint64_t n, int src_mem, const uint64_t* cuts, uint32_t n_members, uint8_t* out,
                     uint64_t out_cap, uint64_t* out_bytes);

/* The output records of a collapse: group g (output order) is the raw record of its representative with the three tags set. */
typedef struct tbk_enc_in {
  int32_t mem;               /* tbk_mem of rep / yc / yx / yd                                                         */
  uint32_t n;                /* output records                                                                       */
  const uint32_t* rep;       /* [n] tile index of the representative (tbk_groups_out.rep)                             */
  const double* yc;          /* [n] tbk_groups_out.yc / yx / yd                                                       */
  const int64_t* yx;
  const int32_t* yd;
  uint32_t n_dev;            /* tile indices below n_dev are records of the tile tbk_bam_decode left on this context: their
                                bytes are read where they lie.  0: no such tile                                       */
  uint32_t n_host;           /* the other representatives, handed over by the caller:                                 */
  const uint8_t* host_blob;  /* HOST: their raw records (block_size field first, as in the BAM stream), packed       */
  const uint64_t* host_off;  /* HOST [n_host + 1]: byte offsets into host_blob                                       */
  const uint32_t* host_slot; /* HOST [n]: for a group whose rep >= n_dev, its record's index in host_off (others ignored) */
  uint32_t first;            /* (ABI version 8) mem == TBK_MEM_KEPT: the records are groups [first, first + n) of the kept results; rep /
                                yc / yx / yd are not read (a caller that gathers host records still needs its own copy of rep)  */
  tbk_ctx* from;             /* (ABI version 8) NULL, or ANOTHER context of the same device whose kept results (TBK_MEM_KEPT) and decoded
                                tile (n_dev) this call reads — nothing of `from` is written, so two contexts may encode different
                                chunks of one collapse side by side (one's copies under the other's deflate) while `from` itself
                                encodes too; the caller keeps `from` from collapsing / decoding / releasing meanwhile             */
} tbk_enc_in;
/* Tag, frame (block_size) and BGZF-deflate the n records on the device: `out` (HOST) receives a run of whole members — every member
 * begins with a record, as htslib cuts them — that a BAM writer appends behind its header; the EOF member is the caller's.
 * *out_bytes = size of the run, *payload_bytes (optional) = the tagged records' bytes before compression.  TBK_E2BIG (with
 * *out_bytes set) when out_cap is too small; TBK_EUNSUPPORTED when a record is nearly as long as a member (the caller's host
 * writer takes such an output); TBK_EINVAL for a malformed record. */
int tbk_bam_encode(tbk_ctx* ctx, const tbk_enc_in* in, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes, uint64_t* payload_bytes);
/* (ABI version 8) Groups [first, first + n) of the results kept by the last tbk_collapse_tile with keep_results, into HOST arrays (any
 * of them may be NULL) — what a caller whose device writer refused a chunk needs for its host writer (flushPData's tag values,
 * tiebrush.cpp:506-525).  TBK_EINVAL when nothing is kept or the range ends behind the kept groups. */
int tbk_kept_results(tbk_ctx* ctx, uint32_t first, uint32_t n, uint32_t* rep, double* yc, int64_t* yx, int32_t* yd);

/* ---- Packed wire form of a tile (ABI version 5) ---------------------------------------------------------------------------
 * What a host decoder can hand over instead of tbk_soa_in when the tile has to cross PCIe: the same records in 9 bytes plus the
 * CIGAR words instead of 20 plus the CIGAR words (the link, not the GPU, paces the host -> host path).  Every pointer is HOST
 * memory (pinned with tbk_host_alloc for full link speed).  Plain inputs only (no TieBrush-merged file, no MD, no names). */
typedef struct tbk_packed_in {
  uint32_t n_files;
  uint32_t n_records;
  uint32_t n_cigar_ops;
  uint32_t n_tid_runs;
  const uint32_t* file_off;     /* [n_files + 1], as in tbk_soa_in                                                   */
  const uint32_t* tid_run_end;  /* [n_tid_runs] ascending, last == n_records: records [tid_run_end[r - 1], tid_run_end[r])
                                   share tid_run_tid[r] (a coordinate-sorted file changes its refID a few times)       */
  const int32_t* tid_run_tid;   /* [n_tid_runs]                                                                      */
  const int32_t* pos;           /* [n_records]                                                                       */
  const uint32_t* meta;         /* [n_records] flag : 12 | strand ('+' 0, '-' 1, '.' 2) : 2 | mapq : 8 | NH code : 10
                                   (NH 0 .. 1021 as it is, 1022 = no NH tag, 1023 = see nh_esc_*)                      */
  const uint8_t* ncig;          /* [n_records] number of CIGAR operations, 255 = see ncig_esc_*                       */
  const uint32_t* cig;          /* [n_cigar_ops] BAM encoding, record after record                                    */
  uint32_t n_nh_esc, n_ncig_esc;
  const uint32_t* nh_esc_idx;   /* records whose NH is outside 0 .. 1021                                              */
  const int32_t* nh_esc_val;
  const uint32_t* ncig_esc_idx; /* records with 255 CIGAR operations or more                                          */
  const uint32_t* ncig_esc_val;
} tbk_packed_in;
/* Copies the packed arrays to the device and rebuilds the structure of arrays there: *tile describes a DEVICE-resident tile in
 * context-owned memory (valid until the next tbk_unpack_tile on `ctx` or tbk_destroy; tile->file_off is in->file_off), ready for
 * tbk_collapse_tile.  TBK_EINVAL when the counts are inconsistent. */
int tbk_unpack_tile(tbk_ctx* ctx, const tbk_packed_in* in, tbk_soa_in* tile);

/* ---- Multi-GPU: shuffle, then collapse (the reference has no counterpart — its only parallelism is
 * tiewrap.py:96-126, batches of files re-collapsed hierarchically).  Every rank holds some input files; the ranks agree
 * on coordinate cuts no read spans, every passing record moves to the rank owning its range, and that rank runs the
 * ordinary tbk_collapse_tile / tbk_coverage_tile on complete data.  These entry points are the device side; the
 * exchange itself is the caller's (torch.distributed over RCCL in tiebrush_amd/dist.py).  All arrays are device memory
 * unless stated; file_off arrays are host memory as in tbk_soa_in. */

/* Per record of `in` (n_records): key = (tid+1)<<31 | pos+1, 1<<62 for tid < 0 (nondecreasing inside a file, else TBK_EUNSORTED),
 * emax = per-file running max of (tid+1)<<31 | (end + 1) over ALL records (a cut is valid where the next start exceeds it), effend = the effective end of the reference's
 * k-way merge (tmerge.h:28-50; it depends on filtered records too, so it is computed here and travels as the record's
 * explicit priority), pass bit 0 = passes_options (tiebrush.cpp:532-541) under `opts`. */
int tbk_shard_prepare(tbk_ctx* ctx, const tbk_collapse_opts* opts, const tbk_soa_in* in, int64_t* key, int64_t* emax, int32_t* effend,
                      uint8_t* pass);
/* m_out[c] = max(m_out[c], farthest keyed read end among the local records that start before cuts[c]) */
int tbk_shard_probe_max(tbk_ctx* ctx, const uint32_t* file_off, uint32_t n_files, const int64_t* key, const int64_t* emax,
                        const int64_t* cuts, uint32_t n_cuts, int64_t* m_out);
/* nxt_out[c] = min(nxt_out[c], first local record start beyond the keyed end m[c]) */
int tbk_shard_probe_next(tbk_ctx* ctx, const uint32_t* file_off, uint32_t n_files, const int64_t* key, const int64_t* m, uint32_t n_cuts,
                         int64_t* nxt_out);
/* Passing records -> rows[<= n_records][TBK_SHARD_ROW] = {tid, pos, strand, n_cigar, effend, index inside its file} and their
 * CIGAR words, grouped by (destination rank = number of cuts <= key, file), file order inside; src_idx[row] = local
 * record.  tab[world][n_files][5] (int64, device) = {first record, rows, words, row base, word base} per block. */
#define TBK_SHARD_ROW 6
int tbk_shard_pack(tbk_ctx* ctx, const tbk_soa_in* in, const int64_t* key, const uint8_t* pass, const int32_t* effend, const int64_t* cuts,
                   uint32_t world, int32_t* rows, uint32_t* cig_out, int64_t* src_idx, int64_t* tab);
/* Received rows (all sources, in (source rank, file) order; file_off2[K+1] host = run boundaries over all K input files)
 * -> the SoA arrays of a tile for tbk_collapse_tile: flag 0, mapq 255, NH absent, cig_off (n2+1 entries) from the
 * n_cigar column, prio_hi = effend, prio_lo = file << 32 | index in file. */
int tbk_shard_unpack(tbk_ctx* ctx, const int32_t* rows, uint32_t n2, const uint32_t* file_off2, uint32_t K, int32_t* tid, int32_t* pos,
                     uint16_t* flag, uint8_t* mapq, uint8_t* strand, int32_t* nh, uint32_t* cig_off, int64_t* prio_hi, int64_t* prio_lo);

/* ---- Multi-GPU: collapse locally, exchange group partials (ABI version 3).  Every rank runs
 * tbk_collapse_tile on its own files and ships ONE row per local group to the rank that owns the group's coordinate range:
 * {key fields, local YC / YX / YD, merge priority (effective end, file, index) of the local representative} plus that
 * representative's CIGAR.  The owner collapses the partials as TieBrush-merged records with explicit priorities — sum YC, sum
 * YX, max YD, argmin priority: SPData::dupAdd is associative (tiebrush.cpp:408-436) and GSegList::processRead reads only the
 * representative's start and exons, equal for every member of a group (:225-249, call site :511-524), so the YD of a
 * sample's list is final on the rank that holds the sample.  The reference's own ancestor of this is tiewrap.py:96-126
 * (batches re-collapsed hierarchically); unlike it, the explicit priority keeps the flat run's representative.  Exact for
 * integral YC; carried fractional YC (inputs written with --store-frac) needs the record shuffle above. */

/* Per local group o of `g` (output order; g_start / g_end required): key[o] = (tid + 1) << 31 | start and emax[o] = running
 * maximum over the groups <= o of (tid + 1) << 31 | (end + 1) — what tbk_shard_probe_max / _next take with one "file" of
 * n_groups entries.  *not_packable (host) = 1 when some YC is not an integer in [1, 2^31) or some YX does not fit 31 bits: the
 * caller must then take the record shuffle. */
int tbk_partial_keys(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* g, int64_t* key, int64_t* emax, uint32_t* not_packable);
/* rows[n_groups][TBK_PARTIAL_ROW] (int32, 48 bytes) = {tid, pos, strand | n_cigar << 8, effective end of the representative, its
 * global file index (first_fidx + local file), its index inside that file, YC, YX, YD, reference span, key word, 0} and the
 * representatives' CIGAR words, both in group order: the groups of destination d (cuts[d - 1] <= key < cuts[d]) are contiguous.
 * Key word: what identifies the alignment among reads with equal (tid, start, strand, span) under opts->strategy — an exact code
 * or a 31-bit hash with a fixed seed, equal on every rank (the owner verifies hashed words against the CIGARs).
 * tab[world][3] (int64, device) = {first group, rows, words} per destination.  g must carry rep_effend and a final yd
 * (tbk_collapse_finish_yd); opts = the options the groups were collapsed with. */
#define TBK_PARTIAL_ROW 12
int tbk_partial_pack(tbk_ctx* ctx, const tbk_collapse_opts* opts, const tbk_soa_in* in, const tbk_groups_out* g, const int64_t* key,
                     const int64_t* cuts, uint32_t world, uint32_t first_fidx, int32_t* rows, uint32_t* cig_out, int64_t* tab);
/* (ABI version 7) The sender's side in three stages that only QUEUE device work — nothing is read back, nothing waits —, with the
 * caller's collectives (all-gathers of the small device arrays below) in between; what the host finally reads is one gathered table.
 * All arrays are device memory.  The reference has no counterpart (tiewrap.py:96-126 starts processes and re-collapses files).
 *   1. tbk_partial_stage_keys: key / emax as tbk_partial_keys, and meta[TBK_PARTIAL_META] = 64 sampled keys (1 << 62 when there are
 *      no groups), n_files, first_fidx, "not packable" (0 / 1), `carry` (a word of the caller's that travels with the gather);
 *   2. all-gather meta -> allmeta[world][TBK_PARTIAL_META]; tbk_partial_stage_cands: targets[world - 1] = the splitter quantiles of all
 *      samples, cands[world - 1][TBK_PARTIAL_CAND] = per cut the LOCAL bundle structure behind the target: {farthest end before the
 *      first local bundle start at or behind the target, horizon = start of the first bundle the list does not describe (1 << 62:
 *      none), then TBK_PARTIAL_BUNDLES pairs (bundle start, farthest end up to its last group)};
 *   3. all-gather cands -> allcands[world][world - 1][TBK_PARTIAL_CAND]; tbk_partial_stage_pack: cuts[world - 1] = per cut the smallest
 *      key at or behind its target that no group of ANY rank reaches across, judged from the lists (a cut no list can settle is left
 *      at 1 << 62 with flag bit 1); rows / cig_out as tbk_partial_pack; tabx[world * 3 + 4] = tab of tbk_partial_pack, then {flags: bit
 *      0 not packable, bit 1 a cut unsettled, bits 8.. device error bits; n_files; first_fidx; carry}.
 * The caller all-gathers tabx, reads it (its one synchronisation), and — all flags clear — knows every rank's send and receive counts. */
#define TBK_PARTIAL_BUNDLES 15
#define TBK_PARTIAL_CAND (2 + 2 * TBK_PARTIAL_BUNDLES)
#define TBK_PARTIAL_META (64 + 4)
int tbk_partial_stage_keys(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* g, int64_t* key, int64_t* emax, uint32_t first_fidx, int64_t carry,
                           int64_t* meta);
int tbk_partial_stage_cands(tbk_ctx* ctx, const int64_t* key, const int64_t* emax, uint32_t n_groups, const int64_t* allmeta, uint32_t world,
                            int64_t* targets, int64_t* cands);
int tbk_partial_stage_pack(tbk_ctx* ctx, const tbk_collapse_opts* opts, const tbk_soa_in* in, const tbk_groups_out* g, const int64_t* key,
                           const int64_t* mymeta, const int64_t* allcands, const int64_t* targets, uint32_t world, uint32_t first_fidx, int64_t* cuts,
                           int32_t* rows, uint32_t* cig_out, int64_t* tabx);
/* (ABI version 7) -L across ranks: cmpFull (tiebrush.cpp:285-302) compares the MD strings behind the CIGARs, so the representatives'
 * MD strings travel beside the rows.  tbk_partial_pack / tbk_partial_stage_pack take TBK_STRAT_FULL when `in` carries md_off / md /
 * md_has (the key word then hashes CIGAR and MD; nothing is packed from g_key).  tbk_partial_pack_md, after either: md_out = the MD bytes
 * of the local groups' representatives in group order, row word 11 = length | (has an MD tag) << 31, md_tab[world] = bytes per
 * destination (tab = the [world][3] table the pack wrote: tabx starts with it).  All arrays device memory; nothing is read back. */
int tbk_partial_pack_md(tbk_ctx* ctx, const tbk_soa_in* in, const tbk_groups_out* g, const int64_t* tab, uint32_t world, int32_t* rows, uint8_t* md_out,
                        int64_t* md_tab);
/* md_off[n2 + 1] / md_has[n2] of received rows (their word 11): the MD columns of the tile tbk_partial_unpack describes */
int tbk_partial_unpack_md(tbk_ctx* ctx, const int32_t* rows, uint32_t n2, uint32_t* md_off, uint8_t* md_has);
/* Received rows (one run per source rank, each in that rank's output order) -> the SoA arrays of a tile whose "files" are the
 * source ranks, all flagged tbmerged: flag 0, mapq 255, NH absent, cig_off (n2 + 1 entries), yc_in / yx_in / yd_in = the
 * partial's YC / YX / YD, prio_hi = effective end, prio_lo = global file << 32 | index in file.  tbk_collapse_tile on that
 * tile (every filter open) is the reduce-by-key of §8e. */
int tbk_partial_unpack(tbk_ctx* ctx, const int32_t* rows, uint32_t n2, int32_t* tid, int32_t* pos, uint16_t* flag, uint8_t* mapq,
                       uint8_t* strand, int32_t* nh, uint32_t* cig_off, double* yc_in, int64_t* yx_in, int64_t* yd_in, int64_t* prio_hi,
                       int64_t* prio_lo);
/* The owner's step in one call, on the rows as received: rows[n2][TBK_PARTIAL_ROW] of n_runs source ranks (run_off[n_runs + 1],
 * host; every run in its rank's output order), cig = their CIGAR words in row order.  The runs are merged in the reference's
 * output order — (tid, start), strand, end, then the strategy compare of tiebrush.cpp:285-345 inside a tie — and partials with
 * equal keys are reduced: out->yc = sum YC, yx = sum YX, yd = max YD, rep = ROW index of the partial with the smallest
 * (effective end, run, row) = the flat run's representative (tmerge.h:28-50); g_start / g_end optional; n_passed = n2.  `view`
 * (optional) receives tiecov's input of the reduced groups as tbk_groups_to_cov_in builds it (context-owned, valid until the
 * next call).  TBK_ECOLLISION: two alignments share a hashed key word; TBK_E2BIG: more partials start on one base than a
 * workgroup merges — in both cases tbk_partial_unpack + tbk_collapse_tile take the tile. */
int tbk_partial_reduce(tbk_ctx* ctx, const tbk_collapse_opts* opts, const int32_t* rows, uint32_t n2, const uint32_t* run_off,
                       uint32_t n_runs, const uint32_t* cig, tbk_groups_out* out, tbk_cov_in* view);
/* (ABI version 7) the same with the rows' MD strings (`md`: the bytes in row order, lengths in row word 11; NULL unless opts->strategy
 * is TBK_STRAT_FULL, which tbk_partial_reduce itself refuses) */
int tbk_partial_reduce_md(tbk_ctx* ctx, const tbk_collapse_opts* opts, const int32_t* rows, uint32_t n2, const uint32_t* run_off, uint32_t n_runs,
                          const uint32_t* cig, const uint8_t* md, tbk_groups_out* out, tbk_cov_in* view);

#ifdef __cplusplus
}
#endif
#endif /* TBK_H_ */
