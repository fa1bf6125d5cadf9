/* tb_oracle.h — CPU oracle for the tiebrush/tiecov hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A literal, single-threaded C restatement of the reference algorithm
 * (/root/reference/src, cited function by function in tb_oracle.c).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library — as the checker, never as the thing shipped or measured as "ours".
 *
 * Parity status: PINNED for default-mode collapse (+ -A), tbMerged re-collapse,
 * coverage, junctions and the sample track's first four columns against the
 * reference's golden fixtures (tests/golden/, see tests/test_oracle_golden.py and
 * the normaliser of SURVEY.md §4.4).  UNPINNED by any reference fixture (literal
 * restatement only): -L/-P/-E strategies, -N/-Q/-S/--keep-secondary filters,
 * --store-frac, soft clips / indels / = X ops, the CIGAR memcmp tie-break.
 * The real reference cannot be built in this image (needs htslib 1.18, gclib,
 * libBigWig — all un-vendored and absent; no network), so there is no
 * oracle/_ref.
 */
#ifndef TB_ORACLE_H_
#define TB_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBO_NH_ABSENT INT32_MIN

enum { TBO_STRAT_CIGAR = 0, TBO_STRAT_FULL = 1, TBO_STRAT_CLIP = 2, TBO_STRAT_EXON = 3 };

enum {
  TBO_OK = 0,
  TBO_EINVAL = -1,
  TBO_ENOMEM = -2,
  TBO_E2BIG = -4,
  TBO_EUNSUPPORTED = -5,
  TBO_EFATALOP = -7
};

typedef struct tbo_opts {
  int32_t strategy;
  int32_t max_nh;   /* INT32_MAX */
  int32_t min_qual; /* -1 */
  uint32_t flags_mask;
  uint8_t keep_supplementary, keep_secondary, keep_unmapped, collapse_same, store_frac;
} tbo_opts;

typedef struct tbo_in {
  uint32_t n_files, n_records;
  const uint32_t* file_off; /* [n_files+1] */
  const uint8_t* tbmerged;  /* [n_files]   */
  const int32_t* tid;
  const int32_t* pos;
  const uint16_t* flag;
  const uint8_t* mapq;
  const uint8_t* strand;
  const int32_t* nh;
  const uint32_t* cig_off;
  const uint32_t* cig;
  const double* yc_in;
  const int64_t* yx_in;
  const int64_t* yd_in;
  const uint32_t* md_off;
  const uint8_t* md;
  const uint8_t* md_has;
  /* -A needs the real names: CSR of NUL-less qnames */
  const uint32_t* qn_off;
  const uint8_t* qn;
} tbo_in;

typedef struct tbo_groups {
  uint32_t cap;
  uint32_t* rep;
  double* yc;
  int64_t* yx;
  int32_t* yd;
  int32_t* g_start;   /* optional */
  int32_t* g_end;     /* optional */
  int32_t* rec_group; /* optional [n_records] */
  uint32_t* merge_order; /* optional [n_records]: record index popped at each step */
  uint32_t n_groups, n_passed;
} tbo_groups;

typedef struct tbo_cov_in {
  uint32_t n_records;
  const int32_t* tid;
  const int32_t* pos;
  const uint16_t* flag;
  const uint32_t* cig_off;
  const uint32_t* cig;
  const double* yc;
  const uint8_t* strand;
  const int64_t* yx;
} tbo_cov_in;

typedef struct tbo_cov_out {
  uint32_t cap_intervals;
  int32_t *iv_tid, *iv_start, *iv_end;
  double* iv_val;
  uint32_t cap_junctions;
  int32_t *j_tid, *j_start, *j_end;
  uint8_t* j_strand;
  double* j_val;
  uint32_t cap_sample; /* 0 = sample track off */
  int32_t num_samples;
  int32_t *s_tid, *s_start, *s_end;
  int64_t* s_count;
  float* s_heat;
  uint32_t n_intervals, n_junctions, n_sample;
  uint64_t n_bases, span_bases;
} tbo_cov_out;

void tbo_opts_default(tbo_opts* o);
/* GSamRecord::setupCoordinates (GSam.cpp:351-417).  exons (pairs start,end) may be NULL;
 * returns the number of exons (0 for unmapped). */
int tbo_setup_coordinates(uint16_t flag, int32_t pos, const uint32_t* cig, uint32_t n_cig, int32_t* start,
                          int32_t* end, int32_t* exons, uint32_t exon_cap);
/* GSamRecord::spliceStrand (GSam.cpp:464-475): xs/ts = first char of an A/Z typed tag or 0 */
char tbo_splice_strand(char xs, char ts, uint16_t flag);
int tbo_collapse(const tbo_opts* o, const tbo_in* in, tbo_groups* out);
int tbo_coverage(const tbo_cov_in* in, tbo_cov_out* out);

#ifdef __cplusplus
}
#endif
#endif
