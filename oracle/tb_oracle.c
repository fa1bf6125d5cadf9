/* tb_oracle.c — literal CPU restatement of the tiebrush/tiecov hot path.
 * TEST INFRASTRUCTURE ONLY (see tb_oracle.h for who may load it and for the
 * parity-pinning status).  Every function cites the reference lines it follows;
 * paths are relative to /root/reference/src.  Single-threaded on purpose: the
 * reference is single-threaded and this is what bench.py times as cpu_baseline
 * (kind "port").
 */
#include "tb_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CIG_OP(c) ((c) & 0xFu)
#define CIG_LEN(c) ((c) >> 4)
enum { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5, C_P = 6, C_EQ = 7, C_X = 8, C_B = 9 };

void tbo_opts_default(tbo_opts* o) {
  /* struct Options, tiebrush.cpp:89-98; keep_unmapped is overwritten to false unless -M (:644) */
  memset(o, 0, sizeof(*o));
  o->strategy = TBO_STRAT_CIGAR;
  o->max_nh = INT_MAX;
  o->min_qual = -1;
}

/* ---------------------------------------------------------------------------------
 * GSamRecord::setupCoordinates — GSam.cpp:351-417 (literal, incl. the ins-in-intron case)
 * ------------------------------------------------------------------------------- */
int tbo_setup_coordinates(uint16_t flag, int32_t pos, const uint32_t* cig, uint32_t n_cig, int32_t* start,
                          int32_t* end, int32_t* exons, uint32_t exon_cap) {
  *start = 0; /* GSeg default for unmapped */
  *end = 0;
  if (flag & 0x4) return 0; /* :354 */
  int l = 0;
  int nex = 0;
  *start = pos + 1; /* :360 */
  int exstart = pos;
  int intron = 0, ins = 0;
  for (uint32_t i = 0; i < n_cig; ++i) {
    unsigned op = CIG_OP(cig[i]);
    switch (op) {
      case C_EQ:
      case C_X:
      case C_M:
      case C_D: /* :367-373 */
        l += (int)CIG_LEN(cig[i]);
        intron = 0;
        ins = 0;
        break;
      case C_N: /* :374-385 */
        if (!ins || !intron) {
          if (exons && (uint32_t)nex < exon_cap) {
            exons[2 * nex] = exstart + 1;
            exons[2 * nex + 1] = pos + l;
          }
          nex++;
        }
        l += (int)CIG_LEN(cig[i]);
        exstart = pos + l;
        intron = 1;
        break;
      case C_S: /* :386-391 */
        intron = 0;
        ins = 0;
        break;
      case C_H: /* :392-395 */
        intron = 0;
        ins = 0;
        break;
      case C_I: /* :396-400 */
        ins = 1;
        break;
      case C_P: /* :401-403 */
        break;
      default: /* :404-406 prints a warning and continues */
        break;
    }
  }
  if (exons && (uint32_t)nex < exon_cap) { /* :409-412 */
    exons[2 * nex] = exstart + 1;
    exons[2 * nex + 1] = pos + l;
  }
  nex++;
  *end = pos + l; /* :413 */
  return nex;
}

/* GSamRecord::spliceStrand — GSam.cpp:464-475 */
char tbo_splice_strand(char xs, char ts, uint16_t flag) {
  char c = xs;
  if (c == 0) {
    char m = ts;
    if (m == '+' || m == '-') {
      if (flag & 0x10)
        c = (m == '+') ? '-' : '+';
      else
        c = m;
    }
  }
  return (c == '+' || c == '-') ? c : '.';
}

/* ---------------------------------------------------------------------------------
 * GSegList — tiebrush.cpp:111-250 (literal linked list, incl. the mergeRead tail drop)
 * ------------------------------------------------------------------------------- */
typedef struct SegNode {
  uint32_t start, end;
  struct SegNode* next;
} SegNode;

typedef struct SegList {
  SegNode* startNode;
  uint32_t last_pos;
  int last_dist;
} SegList;

static SegNode* node_new(uint32_t s, uint32_t e, SegNode* nx) {
  SegNode* n = (SegNode*)malloc(sizeof(SegNode));
  n->start = s;
  n->end = e;
  n->next = nx;
  return n;
}

static void seglist_clear(SegList* L) { /* :140-149 */
  SegNode* p = L->startNode;
  while (p) {
    SegNode* nx = p->next;
    free(p);
    p = nx;
  }
  L->startNode = NULL;
}

static void seglist_reset(SegList* L) { /* :132-138 */
  seglist_clear(L);
  L->last_pos = 0;
  L->last_dist = -1;
}

static void seglist_clearTo(SegList* L, SegNode* toNode) { /* :151-165 */
  SegNode* p = L->startNode;
  while (p && p != toNode) {
    SegNode* nx = p->next;
    free(p);
    p = nx;
  }
  SegNode* nx = toNode->next;
  free(toNode);
  L->startNode = nx;
}

static void seglist_mergeRead(SegList* L, const int32_t* exons, int nex) { /* :167-219 */
  if (L->startNode == NULL) {
    L->startNode = node_new((uint32_t)exons[0], (uint32_t)exons[1], NULL);
    SegNode* cn = L->startNode;
    for (int i = 1; i < nex; i++) {
      SegNode* n = node_new((uint32_t)exons[2 * i], (uint32_t)exons[2 * i + 1], NULL);
      cn->next = n;
      cn = n;
    }
    return;
  }
  SegNode* n = L->startNode;
  SegNode* prev = NULL;
  for (int i = 0; i < nex; i++) {
    uint32_t es = (uint32_t)exons[2 * i], ee = (uint32_t)exons[2 * i + 1];
    while (n) {
      if (ee < n->start) { /* insert before n :182-191 */
        SegNode* nw = node_new(es, ee, n);
        if (n == L->startNode)
          L->startNode = nw;
        else
          prev->next = nw;
        prev = nw;
        break;
      }
      if (es <= n->end) { /* overlap :194-212 */
        if (es < n->start) n->start = es;
        if (ee > n->end) n->end = ee;
        SegNode* next = n->next;
        while (next && next->start <= n->end) {
          uint32_t nend = next->end;
          n->next = next->next;
          free(next);
          next = n->next;
          if (nend > n->end) {
            n->end = nend;
            break;
          }
        }
        break;
      }
      prev = n; /* :214-216 */
      n = n->next;
    }
    /* n == NULL here: the exon (and every later one) is silently dropped */
  }
}

static int seglist_processRead(SegList* L, uint32_t rstart, const int32_t* exons, int nex) { /* :221-250 */
  if (L->last_pos == rstart) {
    seglist_mergeRead(L, exons, nex);
    return L->last_dist;
  }
  int d = 0;
  SegNode* node = L->startNode;
  SegNode* prev = NULL;
  while (node && node->start < rstart) {
    prev = node;
    node = node->next;
  }
  if (prev) {
    if (prev->end >= rstart) d = (int)(rstart - prev->start);
    if (d == 0) seglist_clearTo(L, prev);
  }
  if (L->last_pos != rstart) {
    L->last_pos = rstart;
    L->last_dist = d;
  }
  seglist_mergeRead(L, exons, nex);
  return d;
}

/* ---------------------------------------------------------------------------------
 * collapse state
 * ------------------------------------------------------------------------------- */
typedef struct Group { /* SPData, tiebrush.cpp:350-473 */
  uint32_t rep;
  char tstrand;
  double accYC;
  int64_t accYX;
  int64_t maxYD;
  uint64_t* samples; /* GBitVec(n_files) */
  int dupCount;
} Group;

typedef struct Ctx {
  const tbo_opts* o;
  const tbo_in* in;
  int32_t* st; /* 1-based start (uint in the reference) */
  int32_t* en;
  uint16_t* fidx_of; /* not used for ordering; convenience */
  int32_t* exbuf;    /* scratch for exon lists */
  int32_t* exbuf2;
  uint32_t excap;
  uint32_t nwords;
} Ctx;

static inline uint32_t ncig(const tbo_in* in, uint32_t i) { return in->cig_off[i + 1] - in->cig_off[i]; }
static inline const uint32_t* cigp(const tbo_in* in, uint32_t i) { return in->cig + in->cig_off[i]; }

/* cmpCigar — tiebrush.cpp:304-310 (cmpFlags contributes 0 because flags_mask==0) */
static int cmpCigar(const Ctx* c, uint32_t a, uint32_t b) {
  uint32_t na = ncig(c->in, a), nb = ncig(c->in, b);
  if (na != nb) return (int)na - (int)nb;
  if (na == 0) return 0;
  return memcmp(cigp(c->in, a), cigp(c->in, b), na * sizeof(uint32_t));
}

/* cmpFull — tiebrush.cpp:285-302 */
static int cmpFull(const Ctx* c, uint32_t a, uint32_t b) {
  uint32_t na = ncig(c->in, a), nb = ncig(c->in, b);
  if (na != nb) return (int)na - (int)nb;
  int cc = 0;
  if (na > 0) cc = memcmp(cigp(c->in, a), cigp(c->in, b), na * sizeof(uint32_t));
  if (cc != 0) return cc;
  int ha = c->in->md_has ? c->in->md_has[a] : 0;
  int hb = c->in->md_has ? c->in->md_has[b] : 0;
  if (!ha || !hb) {
    if (ha == hb) return 0;
    if (ha) return 1;
    return -1;
  }
  /* strcmp on NUL-terminated strings == memcmp on the common prefix, then length */
  uint32_t la = c->in->md_off[a + 1] - c->in->md_off[a], lb = c->in->md_off[b + 1] - c->in->md_off[b];
  uint32_t m = la < lb ? la : lb;
  int r = m ? memcmp(c->in->md + c->in->md_off[a], c->in->md + c->in->md_off[b], m) : 0;
  if (r) return r;
  if (la == lb) return 0;
  return la < lb ? -1 : 1;
}

/* cmpCigarClip — tiebrush.cpp:312-332 */
static int cmpCigarClip(const Ctx* c, uint32_t a, uint32_t b) {
  uint32_t al = ncig(c->in, a), bl = ncig(c->in, b);
  const uint32_t* as = cigp(c->in, a);
  const uint32_t* bs = cigp(c->in, b);
  while (al > 0 && CIG_OP(*as) == C_S) {
    as++;
    al--;
  }
  while (al > 0 && CIG_OP(as[al - 1]) == C_S) al--;
  while (bl > 0 && CIG_OP(*bs) == C_S) {
    bs++;
    bl--;
  }
  while (bl > 0 && CIG_OP(bs[bl - 1]) == C_S) bl--;
  if (al != bl) return (int)al - (int)bl;
  if (al == 0) return 0;
  return memcmp(as, bs, al * sizeof(uint32_t));
}

static int exons_of(const Ctx* c, uint32_t i, int32_t* buf) {
  int32_t s, e;
  return tbo_setup_coordinates(c->in->flag[i], c->in->pos[i], cigp(c->in, i), ncig(c->in, i), &s, &e, buf, c->excap);
}

/* cmpExons — tiebrush.cpp:334-345 */
static int cmpExons(const Ctx* c, uint32_t a, uint32_t b) {
  int na = exons_of(c, a, c->exbuf), nb = exons_of(c, b, c->exbuf2);
  if (na != nb) return na - nb;
  for (int i = 0; i < na; i++) {
    if (c->exbuf[2 * i] != c->exbuf2[2 * i]) return c->exbuf[2 * i] - c->exbuf2[2 * i];
    if (c->exbuf[2 * i + 1] != c->exbuf2[2 * i + 1]) return c->exbuf[2 * i + 1] - c->exbuf2[2 * i + 1];
  }
  return 0;
}

/* SPData::operator< — tiebrush.cpp:438-457 */
static int group_less(const Ctx* c, uint32_t ra, char sa, uint32_t rb, char sb) {
  const tbo_in* in = c->in;
  if (in->tid[ra] != in->tid[rb]) return in->tid[ra] < in->tid[rb];
  if (c->st[ra] != c->st[rb]) return (uint32_t)c->st[ra] < (uint32_t)c->st[rb];
  if (sa != sb) return sa < sb;
  if (c->en[ra] != c->en[rb]) return (uint32_t)c->en[ra] < (uint32_t)c->en[rb];
  switch (c->o->strategy) {
    case TBO_STRAT_FULL:
      return cmpFull(c, ra, rb) < 0;
    case TBO_STRAT_CIGAR:
      return cmpCigar(c, ra, rb) < 0;
    case TBO_STRAT_CLIP:
      return cmpCigarClip(c, ra, rb) < 0;
    case TBO_STRAT_EXON:
      return cmpExons(c, ra, rb) < 0;
  }
  return 0;
}

/* passes_options — tiebrush.cpp:532-541 */
static int passes_options(const Ctx* c, uint32_t i) {
  const tbo_opts* o = c->o;
  uint16_t f = c->in->flag[i];
  if (!o->keep_supplementary && (f & 0x800)) return 0;
  if (!o->keep_secondary && (f & 0x100)) return 0;
  if (!o->keep_unmapped && (f & 0x4)) return 0;
  if ((int)c->in->mapq[i] < o->min_qual) return 0;
  int nh = (c->in->nh[i] == TBO_NH_ABSENT) ? 0 : c->in->nh[i]; /* tag_int("NH") default 0 */
  if (nh > o->max_nh) return 0;
  return 1;
}

/* GSamRecord::pairOrder — GSam.h:314-320 */
static int pair_order(uint16_t f) {
  if (f & 0x40) return 1;
  if (f & 0x80) return 2;
  return 0;
}

static int names_equal(const tbo_in* in, uint32_t a, uint32_t b) {
  uint32_t la = in->qn_off[a + 1] - in->qn_off[a], lb = in->qn_off[b + 1] - in->qn_off[b];
  return la == lb && memcmp(in->qn + in->qn_off[a], in->qn + in->qn_off[b], la) == 0;
}

/* TInputRecord::operator< — tmerge.h:28-50 ("decreasing location sort") */
typedef struct Head {
  uint32_t gi;
  uint32_t fidx;
} Head;

static int head_less(const Ctx* c, const Head* a, const Head* b) {
  const tbo_in* in = c->in;
  int t1 = in->tid[a->gi], t2 = in->tid[b->gi];
  if (t1 == t2) {
    uint32_t s1 = (uint32_t)c->st[a->gi], s2 = (uint32_t)c->st[b->gi];
    if (s1 != s2) return s1 > s2;
    uint32_t e1 = (uint32_t)c->en[a->gi], e2 = (uint32_t)c->en[b->gi];
    if (e1 != e2) return e1 > e2;
    /* fidx equal cannot happen: one head per file (the strcmp branch :43 is unreachable) */
    return a->fidx > b->fidx;
  }
  return t1 > t2;
}

/* GList<TInputRecord>::Add on a sorted list: binary search + memmove (tmerge.cpp:326,339) */
static void heads_insert(const Ctx* c, Head* hs, int* cnt, Head h) {
  int lo = 0, hi = *cnt;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (head_less(c, &hs[mid], &h))
      lo = mid + 1;
    else
      hi = mid;
  }
  memmove(hs + lo + 1, hs + lo, (size_t)(*cnt - lo) * sizeof(Head));
  hs[lo] = h;
  (*cnt)++;
}

static inline void bit_set(uint64_t* w, uint32_t i) { w[i >> 6] |= 1ull << (i & 63); }
static inline int bit_test(const uint64_t* w, uint32_t i) { return (int)((w[i >> 6] >> (i & 63)) & 1); }

typedef struct Bucket {
  Group** g;
  int n, cap;
} Bucket;

/* flushPData — tiebrush.cpp:501-530 */
static int flush_bucket(Ctx* c, Bucket* B, SegList* fsegs, SegList* rsegs, tbo_groups* out) {
  for (int i = 0; i < B->n; i++) {
    Group* g = B->g[i];
    int64_t accYX = g->accYX;
    int dsamples = 0;
    for (uint32_t w = 0; w < c->nwords; w++) dsamples += __builtin_popcountll(g->samples[w]);
    accYX += dsamples;
    int dmax = (int)g->maxYD;
    int nex = exons_of(c, g->rep, c->exbuf);
    uint32_t rstart = (uint32_t)c->st[g->rep];
    for (uint32_t s = 0; s < c->in->n_files; s++) {
      if (!bit_test(g->samples, s)) continue;
      if (g->tstrand == '+' || g->tstrand == '.') {
        int r = seglist_processRead(&fsegs[s], rstart, c->exbuf, nex);
        if (r > dmax) dmax = r;
      }
      if (g->tstrand == '-' || g->tstrand == '.') {
        int r = seglist_processRead(&rsegs[s], rstart, c->exbuf, nex);
        if (r > dmax) dmax = r;
      }
    }
    if (out->n_groups >= out->cap) return TBO_E2BIG;
    uint32_t k = out->n_groups++;
    out->rep[k] = g->rep;
    out->yc[k] = g->accYC;
    out->yx[k] = accYX;
    out->yd[k] = dmax > 0 ? dmax : 0;
    if (out->g_start) out->g_start[k] = c->st[g->rep];
    if (out->g_end) out->g_end[k] = c->en[g->rep];
    free(g->samples);
    free(g);
  }
  B->n = 0;
  return 0;
}

/* addPData + SPData::settle/dupAdd — tiebrush.cpp:477-499, :378-436 */
static int add_pdata(Ctx* c, Bucket* B, uint32_t gi, uint32_t fidx, tbo_groups* out, Group*** slot_out) {
  const tbo_in* in = c->in;
  char ts = (char)in->strand[gi];
  int idx = 0, found = 0;
  if (B->n > 0) { /* GList::AddIfNew: binary search derived from operator< only */
    int l = 0, h = B->n - 1;
    while (l <= h) {
      int i = (l + h) >> 1;
      Group* gg = B->g[i];
      int cmp;
      if (group_less(c, gg->rep, gg->tstrand, gi, ts))
        cmp = -1;
      else if (group_less(c, gi, ts, gg->rep, gg->tstrand))
        cmp = 1;
      else
        cmp = 0;
      if (cmp < 0)
        l = i + 1;
      else {
        h = i - 1;
        if (cmp == 0) {
          found = 1;
          l = i;
        }
      }
    }
    idx = l;
  }
  int tb = in->tbmerged ? in->tbmerged[fidx] : 0;
  if (found) { /* dupAdd :408-436 */
    Group* g = B->g[idx];
    if (tb) {
      double yc = in->yc_in ? in->yc_in[gi] : 0.0;
      if (yc == 0.0) yc = 1.0;
      g->accYC += yc;
      g->accYX += in->yx_in ? in->yx_in[gi] : 1;
      int64_t vyd = in->yd_in ? in->yd_in[gi] : 0;
      if (vyd > g->maxYD) g->maxYD = vyd;
    } else {
      if (!c->o->collapse_same || !bit_test(g->samples, fidx) ||
          pair_order(in->flag[gi]) != pair_order(in->flag[g->rep]) || !names_equal(in, g->rep, gi)) {
        if (c->o->store_frac) {
          int nh = (in->nh[gi] == TBO_NH_ABSENT) ? 1 : in->nh[gi];
          g->accYC += 1.0 / nh;
        } else {
          g->accYC += 1.0;
        }
        g->dupCount++;
        bit_set(g->samples, fidx);
      }
    }
    *slot_out = &B->g[idx];
    return 0;
  }
  /* new group: settle :378-406 */
  Group* g = (Group*)calloc(1, sizeof(Group));
  if (!g) return TBO_ENOMEM;
  g->rep = gi;
  g->tstrand = ts;
  g->samples = (uint64_t*)calloc(c->nwords, sizeof(uint64_t));
  if (tb) {
    g->accYC = in->yc_in ? in->yc_in[gi] : 0.0;
    if (g->accYC == 0.0) g->accYC = 1.0;
    g->accYX = in->yx_in ? in->yx_in[gi] : 1;
    g->maxYD = in->yd_in ? in->yd_in[gi] : 0;
  } else {
    if (c->o->store_frac) {
      int nh = (in->nh[gi] == TBO_NH_ABSENT) ? 1 : in->nh[gi];
      g->accYC = 1.0 / nh;
    } else {
      g->accYC = 1.0;
    }
    g->dupCount++;
    bit_set(g->samples, fidx);
  }
  if (B->n == B->cap) {
    B->cap = B->cap ? B->cap * 2 : 16;
    B->g = (Group**)realloc(B->g, (size_t)B->cap * sizeof(Group*));
  }
  memmove(B->g + idx + 1, B->g + idx, (size_t)(B->n - idx) * sizeof(Group*));
  B->g[idx] = g;
  B->n++;
  *slot_out = &B->g[idx];
  return 0;
}

int tbo_collapse(const tbo_opts* o, const tbo_in* in, tbo_groups* out) {
  if (!o || !in || !out) return TBO_EINVAL;
  if (o->flags_mask != 0) return TBO_EUNSUPPORTED; /* -F: comparator is inconsistent, SURVEY.md §3.2 */
  if (o->keep_unmapped) return TBO_EUNSUPPORTED;   /* -M: SURVEY.md A.4 #2 */
  if (o->collapse_same && !in->qn_off) return TBO_EINVAL;
  if (o->strategy == TBO_STRAT_FULL && !in->md_off) return TBO_EINVAL;
  uint32_t N = in->n_records, K = in->n_files;
  Ctx c;
  memset(&c, 0, sizeof(c));
  c.o = o;
  c.in = in;
  c.st = (int32_t*)malloc(sizeof(int32_t) * (N + 1));
  c.en = (int32_t*)malloc(sizeof(int32_t) * (N + 1));
  uint32_t maxc = 1;
  for (uint32_t i = 0; i < N; i++) {
    uint32_t n = ncig(in, i);
    if (n > maxc) maxc = n;
    tbo_setup_coordinates(in->flag[i], in->pos[i], cigp(in, i), n, &c.st[i], &c.en[i], NULL, 0);
  }
  c.excap = maxc + 1;
  c.exbuf = (int32_t*)malloc(sizeof(int32_t) * 2 * c.excap);
  c.exbuf2 = (int32_t*)malloc(sizeof(int32_t) * 2 * c.excap);
  c.nwords = (K + 63) / 64;
  if (c.nwords == 0) c.nwords = 1;
  SegList* fsegs = (SegList*)calloc(K ? K : 1, sizeof(SegList)); /* RDistanceData :256-270 */
  SegList* rsegs = (SegList*)calloc(K ? K : 1, sizeof(SegList));
  for (uint32_t s = 0; s < K; s++) {
    seglist_reset(&fsegs[s]);
    seglist_reset(&rsegs[s]);
  }
  Head* heads = (Head*)malloc(sizeof(Head) * (K + 1));
  uint32_t* nextrec = (uint32_t*)malloc(sizeof(uint32_t) * (K + 1));
  int nheads = 0;
  /* TInputFiles::start — tmerge.cpp:319-327 */
  for (uint32_t f = 0; f < K; f++) {
    nextrec[f] = in->file_off[f];
    if (nextrec[f] < in->file_off[f + 1]) {
      Head h = {nextrec[f]++, f};
      heads_insert(&c, heads, &nheads, h);
    }
  }
  Bucket B = {NULL, 0, 0};
  out->n_groups = 0;
  out->n_passed = 0;
  int rc = 0;
  /* rec_group bookkeeping: remember the group object each record joined */
  Group** joined = NULL;
  uint32_t* pending = NULL; /* records of the open bucket */
  uint32_t npending = 0, pcap = 0;
  if (out->rec_group) {
    joined = (Group**)malloc(sizeof(Group*) * (N + 1));
    for (uint32_t i = 0; i < N; i++) out->rec_group[i] = -1;
  }
  int newChr = 0;
  int prev_pos = -1, prev_tid = -1;
  uint32_t step = 0;
  /* main loop — tiebrush.cpp:570-591; TInputFiles::next — tmerge.cpp:331-344 */
  while (nheads > 0) {
    Head cur = heads[--nheads]; /* Pop(): last = lowest coordinate */
    if (nextrec[cur.fidx] < in->file_off[cur.fidx + 1]) {
      Head h = {nextrec[cur.fidx]++, cur.fidx};
      heads_insert(&c, heads, &nheads, h);
    }
    if (out->merge_order) out->merge_order[step] = cur.gi;
    step++;
    if (!passes_options(&c, cur.gi)) continue;
    out->n_passed++;
    int tid = in->tid[cur.gi];
    int pos = c.st[cur.gi];
    if (tid != prev_tid) {
      if (prev_tid != -1) newChr = 1;
      prev_tid = tid;
      prev_pos = -1;
    }
    if (pos != prev_pos) {
      if (out->rec_group) { /* resolve group output indices of the bucket being flushed */
        for (uint32_t p = 0; p < npending; p++) {
          Group* gj = joined[pending[p]];
          for (int q = 0; q < B.n; q++)
            if (B.g[q] == gj) {
              out->rec_group[pending[p]] = (int32_t)(out->n_groups + (uint32_t)q);
              break;
            }
        }
        npending = 0;
      }
      rc = flush_bucket(&c, &B, fsegs, rsegs, out);
      if (rc) goto done;
      prev_pos = pos;
    }
    if (newChr) {
      for (uint32_t s = 0; s < K; s++) { /* rspacing.reset() :586-589 */
        seglist_reset(&fsegs[s]);
        seglist_reset(&rsegs[s]);
      }
      newChr = 0;
    }
    Group** slot = NULL;
    rc = add_pdata(&c, &B, cur.gi, cur.fidx, out, &slot);
    if (rc) goto done;
    if (out->rec_group) {
      joined[cur.gi] = *slot;
      if (npending == pcap) {
        pcap = pcap ? pcap * 2 : 1024;
        pending = (uint32_t*)realloc(pending, pcap * sizeof(uint32_t));
      }
      pending[npending++] = cur.gi;
    }
  }
  if (out->rec_group) {
    for (uint32_t p = 0; p < npending; p++) {
      Group* gj = joined[pending[p]];
      for (int q = 0; q < B.n; q++)
        if (B.g[q] == gj) {
          out->rec_group[pending[p]] = (int32_t)(out->n_groups + (uint32_t)q);
          break;
        }
    }
  }
  rc = flush_bucket(&c, &B, fsegs, rsegs, out);
done:
  for (int i = 0; i < B.n; i++) {
    free(B.g[i]->samples);
    free(B.g[i]);
  }
  free(B.g);
  for (uint32_t s = 0; s < K; s++) {
    seglist_clear(&fsegs[s]);
    seglist_clear(&rsegs[s]);
  }
  free(fsegs);
  free(rsegs);
  free(heads);
  free(nextrec);
  free(joined);
  free(pending);
  free(c.st);
  free(c.en);
  free(c.exbuf);
  free(c.exbuf2);
  return rc;
}

/* ---------------------------------------------------------------------------------
 * tiecov — tiecov.cpp
 * ------------------------------------------------------------------------------- */
typedef struct Junc { /* CJunc :62-96 */
  int start, end;
  char strand;
  double dupcount;
} Junc;

static int junc_less(const Junc* a, const Junc* b) { /* :74-86 */
  if (a->start == b->start) {
    if (a->end == b->end) return a->strand < b->strand;
    return a->end < b->end;
  }
  return a->start < b->start;
}

typedef struct CovState {
  double* bcov; /* GVec<double> */
  size_t bcov_n, bcov_cap;
  float* smean; /* pair<float,uint64_t>.first */
  uint64_t* scnt;
  size_t bsam_n, bsam_cap;
  Junc* juncs;
  int njuncs, juncs_cap;
  int juncCount;
} CovState;

static void bcov_set_count(CovState* s, size_t n) { /* GVec::setCount(n) / setCount(n,0.0): new slots zero */
  if (n > s->bcov_cap) {
    size_t nc = s->bcov_cap ? s->bcov_cap : 1024;
    while (nc < n) nc *= 2;
    s->bcov = (double*)realloc(s->bcov, nc * sizeof(double));
    s->bcov_cap = nc;
  }
  if (n > s->bcov_n) memset(s->bcov + s->bcov_n, 0, (n - s->bcov_n) * sizeof(double));
  s->bcov_n = n;
}

static void bsam_resize(CovState* s, size_t n) { /* vector::resize(n,{0,1}) */
  if (n > s->bsam_cap) {
    size_t nc = s->bsam_cap ? s->bsam_cap : 1024;
    while (nc < n) nc *= 2;
    s->smean = (float*)realloc(s->smean, nc * sizeof(float));
    s->scnt = (uint64_t*)realloc(s->scnt, nc * sizeof(uint64_t));
    s->bsam_cap = nc;
  }
  for (size_t i = s->bsam_n; i < n; i++) {
    s->smean[i] = 0.0f;
    s->scnt[i] = 1;
  }
  s->bsam_n = n;
}

/* flushCoverage(FILE*) — tiecov.cpp:226-241 */
static int flush_cov(CovState* s, int tid, int b_start, tbo_cov_out* out) {
  if (tid < 0 || b_start <= 0) return 0;
  size_t i = 0;
  b_start--;
  while (i < s->bcov_n) {
    double ival = s->bcov[i];
    size_t j = i + 1;
    while (j < s->bcov_n && ival == s->bcov[j]) j++;
    if (ival != 0.0) {
      if (out->n_intervals >= out->cap_intervals) return TBO_E2BIG;
      uint32_t k = out->n_intervals++;
      out->iv_tid[k] = tid;
      out->iv_start[k] = b_start + (int)i;
      out->iv_end[k] = b_start + (int)j;
      out->iv_val[k] = ival;
    }
    i = j;
  }
  return 0;
}

/* discretize + normalize + flushCoverage(pair) — tiecov.cpp:277-299, :316-323 */
static int flush_sample(CovState* s, int tid, int b_start, tbo_cov_out* out) {
  for (size_t i = 0; i < s->bsam_n; i++) { /* discretize :294-299 */
    s->scnt[i] = (uint64_t)ceilf(s->smean[i]);
    s->smean[i] = 0;
  }
  float denom = (float)out->num_samples; /* normalize(bsam,0.1,1.5,n) :316-323 */
  float mint = 0.1f, maxt = 1.5f;
  float mult = (maxt - mint);
  for (size_t i = 0; i < s->bsam_n; i++) s->smean[i] = ((float)s->scnt[i] / denom) * mult + mint;
  if (tid < 0 || b_start <= 0) return 0;
  size_t i = 0;
  b_start--;
  while (i < s->bsam_n) {
    uint64_t ival = s->scnt[i];
    float hval = s->smean[i];
    size_t j = i + 1;
    while (j < s->bsam_n && ival == s->scnt[j]) j++;
    if (ival != 0) {
      if (out->n_sample >= out->cap_sample) return TBO_E2BIG;
      uint32_t k = out->n_sample++;
      out->s_tid[k] = tid;
      out->s_start[k] = b_start + (int)i;
      out->s_end[k] = b_start + (int)j;
      out->s_count[k] = (int64_t)ival;
      out->s_heat[k] = hval;
    }
    i = j;
  }
  return 0;
}

/* flushJuncs — tiecov.cpp:114-120 + CJunc::write :90-95 */
static int flush_juncs(CovState* s, int tid, tbo_cov_out* out) {
  for (int i = 0; i < s->njuncs; i++) {
    s->juncCount++;
    if (out->n_junctions >= out->cap_junctions) return TBO_E2BIG;
    uint32_t k = out->n_junctions++;
    out->j_tid[k] = tid;
    out->j_start[k] = s->juncs[i].start - 1;
    out->j_end[k] = s->juncs[i].end;
    out->j_strand[k] = (uint8_t)s->juncs[i].strand;
    out->j_val[k] = s->juncs[i].dupcount;
  }
  s->njuncs = 0;
  return 0;
}

/* addJunction — tiecov.cpp:100-112 (GArray<CJunc> sorted unique, AddIfNew) */
static void add_junction(CovState* s, const int32_t* exons, int nex, char strand, double dupcount) {
  for (int i = 1; i < nex; i++) {
    Junc j = {exons[2 * (i - 1) + 1] + 1, exons[2 * i] - 1, strand, dupcount};
    int lo = 0, hi = s->njuncs;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (junc_less(&s->juncs[mid], &j))
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < s->njuncs && !junc_less(&j, &s->juncs[lo])) {
      s->juncs[lo].dupcount += j.dupcount; /* CJunc::add :88-90 */
      continue;
    }
    if (s->njuncs == s->juncs_cap) {
      s->juncs_cap = s->juncs_cap ? s->juncs_cap * 2 : 64;
      s->juncs = (Junc*)realloc(s->juncs, (size_t)s->juncs_cap * sizeof(Junc));
    }
    memmove(s->juncs + lo + 1, s->juncs + lo, (size_t)(s->njuncs - lo) * sizeof(Junc));
    s->juncs[lo] = j;
    s->njuncs++;
  }
}

int tbo_coverage(const tbo_cov_in* in, tbo_cov_out* out) {
  if (!in || !out) return TBO_EINVAL;
  int want_cov = out->cap_intervals > 0, want_j = out->cap_junctions > 0, want_s = out->cap_sample > 0;
  if (want_s && out->num_samples <= 0) return TBO_EINVAL; /* load_sample_info GError, commons.h:47-71 */
  CovState s;
  memset(&s, 0, sizeof(s));
  out->n_intervals = out->n_junctions = out->n_sample = 0;
  out->n_bases = 0;
  out->span_bases = 0;
  uint32_t maxc = 1;
  for (uint32_t i = 0; i < in->n_records; i++) {
    uint32_t n = in->cig_off[i + 1] - in->cig_off[i];
    if (n > maxc) maxc = n;
  }
  int32_t* exons = (int32_t*)malloc(sizeof(int32_t) * 2 * (maxc + 1));
  int prev_tid = -1;
  int b_end = 0, b_start = 0;
  int rc = 0;
  /* main loop — tiecov.cpp:435-499 */
  for (uint32_t i = 0; i < in->n_records; i++) {
    if (in->flag[i] & 0x4) continue; /* :436 */
    const uint32_t* cig = in->cig + in->cig_off[i];
    uint32_t nc = in->cig_off[i + 1] - in->cig_off[i];
    int32_t rstart, rend;
    int nex = tbo_setup_coordinates(in->flag[i], in->pos[i], cig, nc, &rstart, &rend, exons, maxc + 1);
    int endpos = rend;
    if (in->tid[i] != prev_tid || (int)rstart > b_end) { /* :443 */
      if (prev_tid >= 0) {
        if (want_cov && (rc = flush_cov(&s, prev_tid, b_start, out))) goto done;
        if (want_s && (rc = flush_sample(&s, prev_tid, b_start, out))) goto done;
        if (want_j && (rc = flush_juncs(&s, prev_tid, out))) goto done;
      }
      b_start = rstart;
      b_end = endpos;
      if (want_cov) {
        s.bcov_n = 0;
        bcov_set_count(&s, (size_t)(b_end - b_start + 1));
      }
      if (want_s) {
        s.bsam_n = 0;
        bsam_resize(&s, (size_t)(b_end - b_start + 1));
      }
      out->span_bases += (uint64_t)(b_end - b_start + 1);
      prev_tid = in->tid[i];
    } else if (b_end < endpos) { /* :472-481 */
      out->span_bases += (uint64_t)(endpos - b_end);
      b_end = endpos;
      if (want_cov) bcov_set_count(&s, (size_t)(b_end - b_start + 1));
      if (want_s) bsam_resize(&s, (size_t)(b_end - b_start + 1));
    }
    double accYC = in->yc ? in->yc[i] : 1.0; /* :482-485, defaults applied by the decoder */
    if (nc >= 256) { /* uint8_t loop counter never terminates, tiecov.cpp:198 */
      rc = TBO_EUNSUPPORTED;
      goto done;
    }
    /* addCov :194-223 / addMean :155-185 share the walk */
    {
      int pos = in->pos[i];
      int b0 = b_start - 1;
      int val = want_s ? (int)(float)(in->yx ? in->yx[i] : 1) : 0; /* float accYX=(float)tag_int; addMean(int val) */
      for (uint32_t k = 0; k < nc; k++) {
        int op = (int)CIG_OP(cig[k]);
        int oplen = (int)CIG_LEN(cig[k]);
        switch (op) {
          case C_I:
            break;
          case C_D:
            pos += oplen;
            break;
          case C_N:
            pos += oplen;
            break;
          case C_S:
            break;
          case C_M:
            out->n_bases += (uint64_t)oplen;
            for (int q = 0; q < oplen; q++) {
              if (want_cov) s.bcov[pos - b0] += accYC;
              if (want_s) {
                size_t x = (size_t)(pos - b0);
                s.smean[x] += ((float)val - s.smean[x]) / (float)s.scnt[x];
                s.scnt[x]++;
              }
              pos++;
            }
            break;
          default: /* GError :219-220 (only reached when addCov/addMean run, i.e. -c or -s) */
            if (want_cov || want_s) {
              rc = TBO_EFATALOP;
              goto done;
            }
            break;
        }
      }
    }
    if (want_j && nex > 1) add_junction(&s, exons, nex, in->strand ? (char)in->strand[i] : '.', accYC);
  }
  /* final flushes :500-513 */
  if (want_cov && (rc = flush_cov(&s, prev_tid, b_start, out))) goto done;
  if (want_s && (rc = flush_sample(&s, prev_tid, b_start, out))) goto done;
  if (want_j && (rc = flush_juncs(&s, prev_tid, out))) goto done;
done:
  free(exons);
  free(s.bcov);
  free(s.smean);
  free(s.scnt);
  free(s.juncs);
  return rc;
}
