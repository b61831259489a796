/*
 * oracle/dada_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C CPU restatement of the reference's dada() hot path,
 *   dada_uniques -> run_dada            /root/reference/src/Rmain.cpp:30-336
 * i.e. k-mer screen -> banded ends-free NW -> substitutions -> lambda -> Poisson
 * abundance p-value -> shuffle / bud, plus the output tables of src/error.cpp.
 * Every function cites the reference file:line it follows.  Written from the
 * behavioural spec in SURVEY.md Appendix A as structure-of-arrays C; it is pinned by
 * tests/test_oracle_vs_ref.py against oracle/_ref (the reference's own C++ compiled in
 * place) on the reference's fixtures, on seeded synthetic samples and on random pairs,
 * and by the committed goldens in tests/golden/ that _ref generated.
 *
 * Third-party arithmetic: ppois() lives in oracle/rmath_ppois.c ("parity unpinned"
 * against genuine libRmath — see that file's header).
 *
 * nwalign_endsfree_homo (homopolymer gap penalty != gap penalty, reached with VECTORIZED_ALIGNMENT off,
 * R/dada.R:229-231) and the global nwalign of C_nwalign(endsfree=FALSE) are restated in nw_general().
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

double dada2_oracle_ppois(double x, double lambda, int lower_tail);

#define KMER_SIZE 5            /* dada.h:27 */
#define NKMER 1024             /* 4^5 */
#define MAX_SHUFFLE 10         /* dada.h:30 */
#define GAP_GLYPH 9999         /* dada.h:31 */
#define TAIL_APPROX_CUTOFF 1e-7 /* dada.h:25 */
#define SEQLEN 9999            /* dada.h:24 */
#define NA_INTEGER (-2147483647 - 1)

typedef struct {               /* == include/dada2hip.h dada2hip_opts, Rmain.cpp:33-47 */
  double kdist_cutoff, omegaA, omegaP, omegaC, min_fold;
  int32_t match, mismatch, gap, homo_gap, band_size, max_clust, min_hamming, min_abund;
  int32_t use_kmers, detect_singletons, use_quals, final_consensus, vectorized_alignment;
  int32_t multithread, verbose, SSE, gapless, greedy;
} oracle_opts;

static double na_real(void) { union { double d; uint64_t u; } v; v.u = 0x7FF00000000007A2ULL; return v.d; }

/* ------------------------------------------------------------------ k-mers --- */
/* kmers.cpp:207-243 assign_kmer (u16 counts), :158-204 assign_kmer8 (saturate at 255),
   :246-279 assign_kmer_order.  seq is 1..4 coded (misc.cpp:38), index = base-4 number,
   first base most significant (kmers.cpp:176-184). */
static void kmer_tables(const uint8_t *seq, int len, uint16_t *k16, uint8_t *k8, uint16_t *kord)
{
  int i, j;
  memset(k16, 0, NKMER * sizeof(uint16_t));
  for (i = 0; i + KMER_SIZE <= len; i++) {
    unsigned km = 0;
    for (j = 0; j < KMER_SIZE; j++) km = 4 * km + (unsigned)(seq[i + j] - 1);
    k16[km]++;
    if (kord) kord[i] = (uint16_t)km;
  }
  for (i = 0; i < NKMER; i++) k8[i] = k16[i] < 255 ? (uint8_t)k16[i] : 255;
}

/* kmers.cpp:58-93 kmer_dist_SSEi_8: 16 saturating u8 lane sums of min(); any lane == 255
   => overflow => -1.  :29-50 kmer_dist_SSEi / :13-27 kmer_dist: u16 wrap-around sum. */
static double kdist_u8(const uint8_t *a, int la, const uint8_t *b, int lb)
{
  unsigned lane[16] = {0}, i, dotsum = 0;
  int overflow = 0;
  for (i = 0; i < NKMER; i++) {
    unsigned m = a[i] < b[i] ? a[i] : b[i], s = lane[i & 15] + m;
    lane[i & 15] = s > 255 ? 255 : s;
  }
  for (i = 0; i < 16; i++) { if (lane[i] == 255) overflow = 1; dotsum += lane[i]; }
  if (overflow) return -1.;
  return 1. - ((double)(uint16_t)dotsum) / ((la < lb ? la : lb) - KMER_SIZE + 1.);
}
static double kdist_u16(const uint16_t *a, int la, const uint16_t *b, int lb)
{
  uint16_t dotsum = 0;
  int i;
  for (i = 0; i < NKMER; i++) dotsum = (uint16_t)(dotsum + (a[i] < b[i] ? a[i] : b[i]));
  return 1. - ((double)dotsum) / ((la < lb ? la : lb) - KMER_SIZE + 1.);
}
/* kmers.cpp:121-150 kord_dist_SSEi (first min(len)-4 positions) / :102-117 kord_dist
   (-1 when lengths differ; used when SSE==0, nwalign_endsfree.cpp:36-40). */
static double kodist(const uint16_t *a, int la, const uint16_t *b, int lb, int sse)
{
  int klen = (la < lb ? la : lb) - KMER_SIZE + 1, i;
  uint16_t dotsum = 0;
  if (sse < 1 && la != lb) return -1.0;
  for (i = 0; i < klen; i++) dotsum = (uint16_t)(dotsum + (a[i] == b[i]));
  return 1. - ((double)dotsum) / ((la < lb ? la : lb) - KMER_SIZE + 1.);
}

/* --------------------------------------------------------------- alignment --- */
/* Homopolymer flags of nwalign_endsfree_homo (nwalign_endsfree.cpp:227-256): position k is 1 when it lies in a run
   of at least three equal bases. */
static void homo_flags(const uint8_t *s, int len, uint8_t *h)
{
  int i = 0, j, k;
  for (j = 0; j < len; j++)
    if (j == len - 1 || s[j] != s[j + 1]) {
      for (k = i; k <= j; k++) h[k] = (j - i >= 2) ? 1 : 0;
      i = j + 1;
    }
}

/* Banded Needleman-Wunsch of the three scalar aligners of nwalign_endsfree.cpp:
     endsfree, homo_gap_p == gap_p   nwalign_endsfree       :76-216  (== nwalign_vectorized2 with end_gap_p = 0,
                                                                       nwalign_vectorized.cpp:71-318; identical output)
     endsfree, homo_gap_p != gap_p   nwalign_endsfree_homo  :220-396 (a gap opposite a base of a homopolymer run of >= 3 costs
                                                                       homo_gap_p: left move -> the raw's base, up move -> the centre's)
     !endsfree                       nwalign                :403-537 (global: the first row / column cost gap_p per step, no free
                                                                       moves along the last row / column; homo_gap_p is ignored,
                                                                       evaluate.cpp:44-48)
   s1 = centre (rows), s2 = raw (cols).  Tie-break up > left > diag (:146-156).  Out-of-band neighbours read a large negative
   sentinel (-9999 at :113-119; INT16_MIN-min(..) at vectorized :106-112).
   Returns alignment length; al0/al1 (capacity len1+len2+1) get the gapped strings, gap='-'. */
static int nw_general(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int match, int mismatch,
                      int gap_p, int homo_gap_p, int endsfree, int band, int sentinel, uint8_t *al0, uint8_t *al1)
{
  int i, j, ncol = len2 + 1, lband, rband, n = 0;
  const int homo = endsfree && homo_gap_p != gap_p;
  int *d = (int *)malloc(sizeof(int) * (size_t)(len1 + 1) * ncol);
  uint8_t *p = (uint8_t *)malloc((size_t)(len1 + 1) * ncol);
  uint8_t *h1 = (uint8_t *)calloc((size_t)len1 + 1, 1), *h2 = (uint8_t *)calloc((size_t)len2 + 1, 1);
  if (homo) { homo_flags(s1, len1, h1); homo_flags(s2, len2, h2); }
  d[0] = 0; p[0] = endsfree ? 3 : 0;
  for (i = 1; i <= len1; i++) { d[i * ncol] = endsfree ? 0 : d[(i - 1) * ncol] + gap_p; p[i * ncol] = 3; }
  for (j = 1; j <= len2; j++) { d[j] = endsfree ? 0 : d[j - 1] + gap_p; p[j] = 2; }
  if (endsfree) p[0] = 2;                              /* (:88-98: the top-row loop runs last; never queried) */
  lband = band + (len1 > len2 ? len1 - len2 : 0);     /* :100-111 */
  rband = band + (len2 > len1 ? len2 - len1 : 0);
  if (band >= 0 && (band < len1 || band < len2)) {    /* :113-119 */
    for (i = 0; i <= len1; i++) {
      if (i - lband - 1 >= 0) d[i * ncol + i - lband - 1] = sentinel;
      if (i + rband + 1 <= len2) d[i * ncol + i + rband + 1] = sentinel;
    }
  }
  for (i = 1; i <= len1; i++) {
    int l = 1, r = len2;
    if (band >= 0) { l = i - lband; if (l < 1) l = 1; r = i + rband; if (r > len2) r = len2; }
    for (j = l; j <= r; j++) {
      int left = d[i * ncol + j - 1] + ((endsfree && i == len1) ? 0 : ((homo && h2[j - 1]) ? homo_gap_p : gap_p));
      int up = d[(i - 1) * ncol + j] + ((endsfree && j == len2) ? 0 : ((homo && h1[i - 1]) ? homo_gap_p : gap_p));
      int diag = d[(i - 1) * ncol + j - 1] + (s1[i - 1] == s2[j - 1] ? match : mismatch);
      if (up >= diag && up >= left) { d[i * ncol + j] = up; p[i * ncol + j] = 3; }
      else if (left >= diag) { d[i * ncol + j] = left; p[i * ncol + j] = 2; }
      else { d[i * ncol + j] = diag; p[i * ncol + j] = 1; }
    }
  }
  i = len1; j = len2;
  while (i > 0 || j > 0) {                             /* :164-188 */
    switch (p[i * ncol + j]) {
      case 1: al0[n] = s1[--i]; al1[n] = s2[--j]; break;
      case 2: al0[n] = '-'; al1[n] = s2[--j]; break;
      default: al0[n] = s1[--i]; al1[n] = '-'; break;
    }
    n++;
  }
  for (i = 0; i < n / 2; i++) {
    uint8_t t = al0[i]; al0[i] = al0[n - 1 - i]; al0[n - 1 - i] = t;
    t = al1[i]; al1[i] = al1[n - 1 - i]; al1[n - 1 - i] = t;
  }
  free(d); free(p); free(h1); free(h2);
  return n;
}
static int nw_endsfree(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int match, int mismatch,
                       int gap_p, int band, int sentinel, uint8_t *al0, uint8_t *al1)
{
  return nw_general(s1, len1, s2, len2, match, mismatch, gap_p, gap_p, 1, band, sentinel, al0, al1);
}

/* nwalign_endsfree.cpp:539-555 nwalign_gapless: shorter padded with '-' at the end. */
static int nw_gapless(const uint8_t *s1, int len1, const uint8_t *s2, int len2, uint8_t *al0, uint8_t *al1)
{
  int n = len1 > len2 ? len1 : len2, i;
  for (i = 0; i < n; i++) { al0[i] = i < len1 ? s1[i] : '-'; al1[i] = i < len2 ? s2[i] : '-'; }
  return n;
}

/* Sub (dada.h:53-62) as produced by al2subs (nwalign_endsfree.cpp:570-639) + the q0/q1
   fill of sub_new (:653-662). */
typedef struct {
  int valid, nsubs, len0;
  uint16_t *map, *pos;
  uint8_t *nt0, *nt1, *q0, *q1;
} Sub;

static void sub_release(Sub *s) { free(s->map); free(s->pos); free(s->nt0); free(s->nt1); free(s->q0); free(s->q1); memset(s, 0, sizeof(*s)); }

static void al2subs(const uint8_t *al0, const uint8_t *al1, int n, Sub *s)
{
  int i, i0 = -1, i1 = -1, len0 = 0, nsubs = 0;
  for (i = 0; i < n; i++) {
    int is0 = al0[i] >= 1 && al0[i] <= 5, is1 = al1[i] >= 1 && al1[i] <= 5;
    if (is0) len0++;
    if (is0 && is1 && al0[i] != al1[i] && al0[i] != 5 && al1[i] != 5) nsubs++;
  }
  s->valid = 1; s->len0 = len0; s->nsubs = 0;
  s->map = (uint16_t *)malloc(sizeof(uint16_t) * (len0 + 1));
  s->pos = (uint16_t *)malloc(sizeof(uint16_t) * (nsubs + 1));
  s->nt0 = (uint8_t *)malloc(nsubs + 1); s->nt1 = (uint8_t *)malloc(nsubs + 1);
  s->q0 = (uint8_t *)malloc(nsubs + 1); s->q1 = (uint8_t *)malloc(nsubs + 1);
  for (i = 0; i < n; i++) {
    int is0 = al0[i] >= 1 && al0[i] <= 5, is1 = al1[i] >= 1 && al1[i] <= 5;
    if (is0) i0++;
    if (is1) i1++;
    if (is0) s->map[i0] = is1 ? (uint16_t)i1 : GAP_GLYPH;
    if (is0 && is1 && al0[i] != al1[i] && al0[i] != 5 && al1[i] != 5) {
      s->pos[s->nsubs] = (uint16_t)i0; s->nt0[s->nsubs] = al0[i]; s->nt1[s->nsubs] = al1[i]; s->nsubs++;
    }
  }
}

/* ------------------------------------------------------------ sample state --- */
typedef struct { unsigned i, index; double lambda; unsigned hamming; } Comp; /* dada.h:42-47 */

typedef struct {
  int nraw, maxlen, ncol, has_quals;
  /* Raw (dada.h:65-80) as SoA */
  uint8_t **seq; int *len; uint8_t **qual; unsigned *reads; uint8_t *prior;
  uint16_t *k16; uint8_t *k8; uint16_t *kord;
  double *p, *E_minmax; Comp *comp; uint8_t *lock, *correct;
  const double *err;   /* row-major [16][ncol] (cluster.cpp:166-170) */
  oracle_opts o;
  /* B / Bi (dada.h:85-123) */
  int nclust, maxclust; unsigned reads_total, nalign, nshroud;
  struct Bi { unsigned *raw; unsigned nraw, maxraw, reads, center; int update_e, check_locks; double self;
              char birth_type; unsigned birth_from; double birth_pval, birth_fold, birth_e; Comp birth_comp;
              Comp *comp; unsigned ncomp, maxcomp; } *bi;
  char errmsg[256];
} S;

/* Value read for out-of-band neighbours: -9999 in nwalign_endsfree (nwalign_endsfree.cpp:113-119),
   INT16_MIN - min(mismatch, gap, match, 0) in nwalign_vectorized2 (nwalign_vectorized.cpp:106). */
static int nw_sentinel(const oracle_opts *o)
{
  int m = 0;
  if (!o->vectorized_alignment) return -9999;
  if (o->mismatch < m) m = o->mismatch;
  if (o->gap < m) m = o->gap;
  if (o->match < m) m = o->match;
  return -32768 - m;
}

/* raw_align + al2subs + sub_new  (nwalign_endsfree.cpp:10-73, :642-672) */
static void make_sub(S *s, unsigned c, unsigned r, int use_kmers, double cutoff, Sub *sub, double *kd_out, double *ko_out)
{
  const oracle_opts *o = &s->o;
  double kd = 0.0, ko = -1.0;
  int n, i;
  uint8_t *al0, *al1;
  memset(sub, 0, sizeof(*sub));
  if (use_kmers) {
    if (o->SSE == 2) {
      kd = kdist_u8(s->k8 + (size_t)c * NKMER, s->len[c], s->k8 + (size_t)r * NKMER, s->len[r]);
      if (kd < 0) kd = kdist_u16(s->k16 + (size_t)c * NKMER, s->len[c], s->k16 + (size_t)r * NKMER, s->len[r]);
    } else kd = kdist_u16(s->k16 + (size_t)c * NKMER, s->len[c], s->k16 + (size_t)r * NKMER, s->len[r]);
    if (o->gapless) ko = kodist(s->kord + (size_t)c * s->maxlen, s->len[c], s->kord + (size_t)r * s->maxlen, s->len[r], o->SSE);
  }
  if (kd_out) *kd_out = kd;
  if (ko_out) *ko_out = ko;
  if (use_kmers && kd > cutoff) return;                         /* shrouded: NULL sub (:51-53) */
  al0 = (uint8_t *)malloc(s->len[c] + s->len[r] + 2);
  al1 = (uint8_t *)malloc(s->len[c] + s->len[r] + 2);
  if (o->band_size == 0 || (o->gapless && ko == kd))             /* :54-55 */
    n = nw_gapless(s->seq[c], s->len[c], s->seq[r], s->len[r], al0, al1);
  else                                                           /* :57-64: vectorized | endsfree_homo | endsfree */
    n = nw_general(s->seq[c], s->len[c], s->seq[r], s->len[r], o->match, o->mismatch, o->gap,
                   (!o->vectorized_alignment && o->homo_gap != o->gap && o->homo_gap <= 0) ? o->homo_gap : o->gap, 1, o->band_size,
                   nw_sentinel(o), al0, al1);
  al2subs(al0, al1, n, sub);
  if (s->has_quals)
    for (i = 0; i < sub->nsubs; i++) { sub->q0[i] = s->qual[c][sub->pos[i]]; sub->q1[i] = s->qual[r][sub->map[sub->pos[i]]]; }
  free(al0); free(al1);
}

/* compute_lambda / compute_lambda_ts  (pval.cpp:92-141 / :144-197): sequential fp64 product
   over raw positions; returns <0 on the reference's error conditions. */
static double lambda_of(S *s, unsigned r, const Sub *sub)
{
  int len1 = s->len[r], pos1, i;
  double lambda = 1.0;
  static __thread unsigned tvec[SEQLEN];
  if (!sub->valid) return 0.0;
  for (pos1 = 0; pos1 < len1; pos1++) tvec[pos1] = 5u * (unsigned)(s->seq[r][pos1] - 1);
  for (i = 0; i < sub->nsubs; i++) tvec[sub->map[sub->pos[i]]] = 4u * (sub->nt0[i] - 1) + (sub->nt1[i] - 1);
  for (pos1 = 0; pos1 < len1; pos1++) {
    unsigned q = s->o.use_quals && s->has_quals ? s->qual[r][pos1] : 0;
    if (q > (unsigned)(s->ncol - 1)) { snprintf(s->errmsg, sizeof s->errmsg, "Rounded quality exceeded range of err lookup table."); return -1.; }
    lambda = lambda * s->err[tvec[pos1] * s->ncol + q];
  }
  if (lambda < 0 || lambda > 1) { snprintf(s->errmsg, sizeof s->errmsg, "Bad lambda."); return -1.; }
  return lambda;
}

/* containers.cpp:150-197 */
static void bi_add_raw(S *s, int i, unsigned r)
{
  struct Bi *b = &s->bi[i];
  if (b->nraw >= b->maxraw) { b->maxraw = b->maxraw ? b->maxraw * 2 : 64; b->raw = (unsigned *)realloc(b->raw, sizeof(unsigned) * b->maxraw); }
  b->raw[b->nraw++] = r; b->reads += s->reads[r]; b->update_e = 1;
}
static unsigned bi_pop_raw(S *s, int i, unsigned slot)
{
  struct Bi *b = &s->bi[i];
  unsigned r = b->raw[slot];
  b->raw[slot] = b->raw[b->nraw - 1];      /* swap-with-last (containers.cpp:187) */
  b->nraw--; b->reads -= s->reads[r]; b->update_e = 1;
  return r;
}
static int b_add_bi(S *s)
{
  if (s->nclust >= s->maxclust) { s->maxclust = s->maxclust ? s->maxclust * 2 : 64; s->bi = (struct Bi *)realloc(s->bi, sizeof(struct Bi) * s->maxclust); }
  memset(&s->bi[s->nclust], 0, sizeof(struct Bi));
  s->bi[s->nclust].update_e = 1; s->bi[s->nclust].check_locks = 1;   /* containers.cpp:53-68 */
  return s->nclust++;
}
/* cluster.cpp:371-386 bi_assign_center: first max-reads member; unlock all */
static void bi_assign_center(S *s, int i)
{
  struct Bi *b = &s->bi[i];
  unsigned r, max_reads = 0;
  b->center = 0xFFFFFFFFu;
  for (r = 0; r < b->nraw; r++) {
    s->lock[b->raw[r]] = 0;
    if (s->reads[b->raw[r]] > max_reads) { b->center = b->raw[r]; max_reads = s->reads[b->center]; }
  }
  b->check_locks = 1;
}

/* b_compare (cluster.cpp:13-88) == b_compare_parallel (:152-204) for everything but the
   nalign counter (serial: only real alignments; parallel: every raw, :180). */
static int b_compare(S *s, int i, double cutoff)
{
  struct Bi *b = &s->bi[i];
  unsigned index, c = b->center, creads = s->reads[c];
  for (index = 0; index < (unsigned)s->nraw; index++) {
    Sub sub; double lambda; Comp cp;
    memset(&sub, 0, sizeof sub);
    if (s->o.greedy && (s->reads[index] > creads || s->lock[index])) { /* skipped: NULL sub (:56-59) */ }
    else {
      make_sub(s, c, index, s->o.use_kmers, cutoff, &sub, NULL, NULL);
      if (!s->o.multithread) { s->nalign++; if (!sub.valid) s->nshroud++; }
    }
    if (s->o.multithread) s->nalign++;
    lambda = lambda_of(s, index, &sub);
    if (lambda < 0) { sub_release(&sub); return 1; }
    if (index == c) b->self = lambda;
    if (lambda * s->reads_total > s->E_minmax[index]) {                   /* :73-84 */
      if (lambda * creads > s->E_minmax[index]) s->E_minmax[index] = lambda * creads;
      cp.i = (unsigned)i; cp.index = index; cp.lambda = lambda; cp.hamming = sub.valid ? (unsigned)sub.nsubs : 0xFFFFFFFFu;
      if (b->ncomp >= b->maxcomp) { b->maxcomp = b->maxcomp ? b->maxcomp * 2 : 256; b->comp = (Comp *)realloc(b->comp, sizeof(Comp) * b->maxcomp); }
      b->comp[b->ncomp++] = cp;
      if (i == 0 || index == c) s->comp[index] = cp;
    }
    sub_release(&sub);
  }
  return 0;
}

/* b_shuffle2 (cluster.cpp:210-266) */
static int b_shuffle2(S *s)
{
  int i, shuffled = 0;
  unsigned index, cind;
  double *emax = (double *)malloc(sizeof(double) * s->nraw);
  Comp **cmax = (Comp **)malloc(sizeof(Comp *) * s->nraw);
  for (index = 0; index < (unsigned)s->nraw; index++) { cmax[index] = &s->bi[0].comp[index]; emax[index] = cmax[index]->lambda * s->bi[0].reads; }
  for (i = 1; i < s->nclust; i++)
    for (cind = 0; cind < s->bi[i].ncomp; cind++) {
      Comp *cp = &s->bi[i].comp[cind];
      double e = cp->lambda * s->bi[i].reads;
      if (e > emax[cp->index]) { cmax[cp->index] = cp; emax[cp->index] = e; }
    }
  for (i = 0; i < s->nclust; i++) {
    int r;
    for (r = (int)s->bi[i].nraw - 1; r >= 0; r--) {
      unsigned raw = s->bi[i].raw[r];
      if (cmax[raw]->i != (unsigned)i) {
        if (raw == s->bi[i].center) continue;
        bi_pop_raw(s, i, (unsigned)r);
        bi_add_raw(s, (int)cmax[raw]->i, raw);
        s->comp[raw] = *cmax[raw];
        shuffled = 1;
      }
    }
  }
  free(cmax); free(emax);
  return shuffled;
}

/* calc_pA (pval.cpp:44-64) */
double oracle_calc_pA(int reads, double E_reads, int prior)
{
  double norm, pval = dada2_oracle_ppois((double)(reads - 1), E_reads, 0);
  if (!prior) {
    norm = 1.0 - exp(-E_reads);
    if (norm < TAIL_APPROX_CUTOFF) norm = E_reads - 0.5 * E_reads * E_reads;
    pval = pval / norm;
  }
  return pval;
}
/* get_pA (pval.cpp:67-89) */
static double get_pA(S *s, unsigned raw, int i)
{
  double lambda = s->comp[raw].lambda;
  unsigned hamming = s->comp[raw].hamming;
  if (s->reads[raw] == 1 && !s->prior[raw] && !s->o.detect_singletons) return 1.;
  if (hamming == 0) return 1.;
  if (lambda == 0) return 0.;
  return oracle_calc_pA((int)s->reads[raw], lambda * s->bi[i].reads, s->prior[raw] || s->o.detect_singletons);
}
/* b_p_update (pval.cpp:14-40) */
static void b_p_update(S *s)
{
  int i; unsigned r;
  for (i = 0; i < s->nclust; i++) {
    struct Bi *b = &s->bi[i];
    if (b->update_e) { for (r = 0; r < b->nraw; r++) s->p[b->raw[r]] = get_pA(s, b->raw[r], i); b->update_e = 0; }
    if (s->o.greedy && b->check_locks) {
      for (r = 0; r < b->nraw; r++) {
        unsigned raw = b->raw[r];
        double E_center = s->reads[b->center] * s->comp[raw].lambda;
        if (E_center > s->reads[raw]) s->lock[raw] = 1;
        if (raw == b->center) s->lock[raw] = 1;
      }
      b->check_locks = 0;
    }
  }
}

/* b_bud (cluster.cpp:274-350) */
static int b_bud(S *s)
{
  int i, mini = -1, minr = -1, mini_p = -1, minr_p = -1;
  unsigned r, minraw = s->bi[0].center, minraw_p = s->bi[0].center, raw;
  double pA, pP, expected;
  for (i = 0; i < s->nclust; i++)
    for (r = 1; r < s->bi[i].nraw; r++) {
      raw = s->bi[i].raw[r];
      if (s->reads[raw] < (unsigned)s->o.min_abund) continue;
      if ((int)s->comp[raw].hamming >= s->o.min_hamming) {
        if (s->o.min_fold <= 1 || ((double)s->reads[raw]) >= s->o.min_fold * s->comp[raw].lambda * s->bi[i].reads) {
          if (s->p[raw] < s->p[minraw] || (s->p[raw] == s->p[minraw] && s->reads[raw] > s->reads[minraw])) { mini = i; minr = (int)r; minraw = raw; }
          if (s->prior[raw] && (s->p[raw] < s->p[minraw_p] || (s->p[raw] == s->p[minraw_p] && s->reads[raw] > s->reads[minraw_p]))) { mini_p = i; minr_p = (int)r; minraw_p = raw; }
        }
      }
    }
  pA = s->p[minraw] * s->nraw;
  pP = s->p[minraw_p];
  if (pA < s->o.omegaA && mini >= 0) {
    expected = s->comp[minraw].lambda * s->bi[mini].reads;
    raw = bi_pop_raw(s, mini, (unsigned)minr);
    i = b_add_bi(s);
    s->bi[i].birth_type = 'A'; s->bi[i].birth_from = (unsigned)mini; s->bi[i].birth_pval = pA;
    s->bi[i].birth_fold = s->reads[raw] / expected; s->bi[i].birth_e = expected; s->bi[i].birth_comp = s->comp[minraw];
    bi_add_raw(s, i, raw); bi_assign_center(s, i);
    return i;
  } else if (pP < s->o.omegaP && mini_p >= 0) {
    expected = s->comp[minraw_p].lambda * s->bi[mini_p].reads;
    raw = bi_pop_raw(s, mini_p, (unsigned)minr_p);
    i = b_add_bi(s);
    s->bi[i].birth_type = 'P'; s->bi[i].birth_from = 0; /* never set in the reference (cluster.cpp:331-345): indeterminate */
    s->bi[i].birth_pval = pP; s->bi[i].birth_fold = s->reads[raw] / expected; s->bi[i].birth_e = expected;
    s->bi[i].birth_comp = s->comp[minraw_p];
    bi_add_raw(s, i, raw); bi_assign_center(s, i);
    return i;
  }
  return 0;
}

/* ----------------------------------------------------------------- results --- */
typedef struct {
  int nclust, nraw, maxlen, ncol, nbirth_subs;
  unsigned nalign, nshroud;
  char **cl_sequence;
  int *cl_abundance, *cl_n0, *cl_n1, *cl_nunq, *cl_birth_from, *cl_birth_ham, *cl_center;
  double *cl_pval, *cl_birth_pval, *cl_birth_fold, *cl_birth_qave;
  int *bs_pos, *bs_clust; char *bs_ref, *bs_sub; double *bs_qual;
  int *subqual;            /* 16 x ncol, column-major like R */
  double *clusterquals;    /* maxlen x nclust, column-major */
  int *map; double *pval;
} oracle_result;

void oracle_result_free(oracle_result *r)
{
  int i;
  if (!r) return;
  if (r->cl_sequence) for (i = 0; i < r->nclust; i++) free(r->cl_sequence[i]);
  free(r->cl_sequence); free(r->cl_abundance); free(r->cl_n0); free(r->cl_n1); free(r->cl_nunq);
  free(r->cl_birth_from); free(r->cl_birth_ham); free(r->cl_center); free(r->cl_pval); free(r->cl_birth_pval);
  free(r->cl_birth_fold); free(r->cl_birth_qave); free(r->bs_pos); free(r->bs_clust); free(r->bs_ref);
  free(r->bs_sub); free(r->bs_qual); free(r->subqual); free(r->clusterquals); free(r->map); free(r->pval);
  free(r);
}

static const char NT[6] = { '?', 'A', 'C', 'G', 'T', 'N' };

static void s_free(S *s)
{
  int i;
  for (i = 0; i < s->nraw; i++) { if (s->seq) free(s->seq[i]); if (s->qual) free(s->qual[i]); }
  for (i = 0; i < s->nclust; i++) { free(s->bi[i].raw); free(s->bi[i].comp); }
  free(s->seq); free(s->qual); free(s->len); free(s->reads); free(s->prior); free(s->k16); free(s->k8); free(s->kord);
  free(s->p); free(s->E_minmax); free(s->comp); free(s->lock); free(s->correct); free(s->bi); free((void *)s->err);
}

/* dada_uniques (Rmain.cpp:30-295) + run_dada (:297-336).  err: column-major 16 x ncol;
   quals: column-major maxlen x nraw (positions are rows) or NULL. */
oracle_result *oracle_run(int nraw, const char *const *seqs, const int *abund, const unsigned char *priors,
                          const double *err, int err_ncol, const double *quals, int quals_nrow,
                          const oracle_opts *o, char *errbuf, int errlen)
{
  S st, *s = &st;
  oracle_result *R = NULL;
  Sub *subs = NULL, *bsubs = NULL;
  int i, index, maxlen = 0, minlen = SEQLEN, max_clust, newi;
  unsigned r;
  double *e;
  memset(s, 0, sizeof *s);
#define FAIL(msg) do { if (errbuf) snprintf(errbuf, errlen, "%s", msg); goto fail; } while (0)
  /* validation, Rmain.cpp:52-78 */
  if (nraw == 0) FAIL("Zero input sequences.");
  for (index = 0; index < nraw; index++) { int l = (int)strlen(seqs[index]); if (l > maxlen) maxlen = l; if (l < minlen) minlen = l; }
  if (maxlen >= SEQLEN) FAIL("Input sequences exceed the maximum allowed string length.");
  if (minlen <= KMER_SIZE) FAIL("Input sequences must all be longer than the kmer-size (5).");
  if (quals && quals_nrow != maxlen) FAIL("Sequence must have associated qualities for each nucleotide position.");
  s->o = *o; s->nraw = nraw; s->maxlen = maxlen; s->ncol = err_ncol; s->has_quals = quals != NULL;
  e = (double *)malloc(sizeof(double) * 16 * err_ncol);          /* row-major copy, cluster.cpp:166-170 */
  for (i = 0; i < 16; i++) for (index = 0; index < err_ncol; index++) e[i * err_ncol + index] = err[index * 16 + i];
  s->err = e;
  /* raws, Rmain.cpp:102-120, containers.cpp:19-43 */
  s->seq = (uint8_t **)calloc(nraw, sizeof(uint8_t *)); s->qual = (uint8_t **)calloc(nraw, sizeof(uint8_t *));
  s->len = (int *)malloc(sizeof(int) * nraw); s->reads = (unsigned *)malloc(sizeof(unsigned) * nraw);
  s->prior = (uint8_t *)malloc(nraw); s->p = (double *)calloc(nraw, sizeof(double));
  s->E_minmax = (double *)malloc(sizeof(double) * nraw); s->comp = (Comp *)calloc(nraw, sizeof(Comp));
  s->lock = (uint8_t *)calloc(nraw, 1); s->correct = (uint8_t *)malloc(nraw);
  for (index = 0; index < nraw; index++) {
    int l = (int)strlen(seqs[index]), pos;
    s->len[index] = l; s->seq[index] = (uint8_t *)malloc(l + 1);
    for (pos = 0; pos < l; pos++) {
      char ch = seqs[index][pos];                                      /* nt2int, misc.cpp:38-68 */
      s->seq[index][pos] = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'T' ? 4 : ch == 'N' ? 5 : 0;
      if (s->seq[index][pos] < 1 || s->seq[index][pos] > 4) FAIL("Non-ACGT sequences in compute_lambda.");
    }
    s->seq[index][l] = 0;
    if (quals) {
      s->qual[index] = (uint8_t *)malloc(l);
      for (pos = 0; pos < l; pos++) s->qual[index][pos] = (uint8_t)round(quals[(size_t)index * maxlen + pos]); /* containers.cpp:34 */
    }
    s->reads[index] = (unsigned)abund[index]; s->prior[index] = priors ? priors[index] != 0 : 0;
    s->E_minmax[index] = -999.0; s->correct[index] = 1; s->reads_total += s->reads[index];
  }
  if (o->use_kmers) {                                               /* Rmain.cpp:122-155 */
    s->k8 = (uint8_t *)malloc((size_t)nraw * NKMER); s->k16 = (uint16_t *)malloc((size_t)nraw * NKMER * 2);
    s->kord = (uint16_t *)calloc((size_t)nraw * maxlen, 2);
    for (index = 0; index < nraw; index++)
      kmer_tables(s->seq[index], s->len[index], s->k16 + (size_t)index * NKMER, s->k8 + (size_t)index * NKMER, s->kord + (size_t)index * maxlen);
  }
  /* run_dada, Rmain.cpp:297-336; b_init containers.cpp:111-137 */
  b_add_bi(s);
  s->bi[0].birth_type = 'I'; s->bi[0].birth_fold = 1.0; s->bi[0].birth_e = s->reads_total;
  for (index = 0; index < nraw; index++) bi_add_raw(s, 0, (unsigned)index);
  bi_assign_center(s, 0);
  if (b_compare(s, 0, 1.0)) FAIL(s->errmsg);
  b_p_update(s);
  max_clust = o->max_clust < 1 ? nraw : o->max_clust;
  while (s->nclust < max_clust && (newi = b_bud(s))) {
    int nshuffle = 0, shuffled;
    if (b_compare(s, newi, o->kdist_cutoff)) FAIL(s->errmsg);
    do { shuffled = b_shuffle2(s); } while (shuffled && ++nshuffle < MAX_SHUFFLE);
    b_p_update(s);
  }
  /* final subs, Rmain.cpp:172-236 */
  subs = (Sub *)calloc(nraw, sizeof(Sub)); bsubs = (Sub *)calloc(s->nclust, sizeof(Sub));
  for (i = 0; i < s->nclust; i++) {
    for (r = 0; r < s->bi[i].nraw; r++) make_sub(s, s->bi[i].center, s->bi[i].raw[r], 0, 1.0, &subs[s->bi[i].raw[r]], NULL, NULL);
    if (i > 0) make_sub(s, s->bi[s->bi[i].birth_comp.i].center, s->bi[i].center, o->use_kmers, 1.0, &bsubs[i], NULL, NULL);
  }
  R = (oracle_result *)calloc(1, sizeof *R);
  R->nclust = s->nclust; R->nraw = nraw; R->maxlen = maxlen; R->ncol = s->has_quals ? err_ncol : 1;
  R->nalign = s->nalign; R->nshroud = s->nshroud;
  R->pval = (double *)malloc(sizeof(double) * nraw); R->map = (int *)malloc(sizeof(int) * nraw);
  /* final per-raw p, Rmain.cpp:238-252 */
  for (i = 0; i < s->nclust; i++)
    for (r = 0; r < s->bi[i].nraw; r++) {
      unsigned raw = s->bi[i].raw[r];
      if (s->bi[i].center == raw) s->p[raw] = 1.0;
      else { s->p[raw] = oracle_calc_pA((int)s->reads[raw], s->comp[raw].lambda * s->bi[i].reads, 1); if (s->p[raw] < o->omegaC) s->correct[raw] = 0; }
      R->pval[raw] = s->p[raw];
    }
  /* b_make_clustering_df, error.cpp:9-127 */
  {
    int C = s->nclust;
    double *tot_e = (double *)calloc(C, sizeof(double));
    int *center_of = (int *)malloc(sizeof(int) * nraw);
    R->cl_sequence = (char **)calloc(C, sizeof(char *));
    R->cl_abundance = (int *)calloc(C, sizeof(int)); R->cl_n0 = (int *)calloc(C, sizeof(int)); R->cl_n1 = (int *)calloc(C, sizeof(int));
    R->cl_nunq = (int *)calloc(C, sizeof(int)); R->cl_birth_from = (int *)calloc(C, sizeof(int)); R->cl_birth_ham = (int *)calloc(C, sizeof(int));
    R->cl_center = (int *)calloc(C, sizeof(int));
    R->cl_pval = (double *)calloc(C, sizeof(double)); R->cl_birth_pval = (double *)calloc(C, sizeof(double));
    R->cl_birth_fold = (double *)calloc(C, sizeof(double)); R->cl_birth_qave = (double *)calloc(C, sizeof(double));
    for (index = 0; index < nraw; index++) center_of[index] = -1;
    for (i = 0; i < C; i++) {
      struct Bi *b = &s->bi[i];
      unsigned max_reads = 0; int max_raw = -1, pos;
      for (r = 0; r < b->nraw; r++) if (s->reads[b->raw[r]] > max_reads) { max_raw = (int)b->raw[r]; max_reads = s->reads[max_raw]; }
      R->cl_sequence[i] = (char *)calloc(maxlen + 1, 1);
      if (max_raw >= 0) for (pos = 0; pos < s->len[max_raw]; pos++) R->cl_sequence[i][pos] = NT[s->seq[max_raw][pos]];
      R->cl_center[i] = (int)b->center;
      for (r = 0; r < b->nraw; r++) {
        unsigned raw = b->raw[r];
        if (!s->correct[raw]) continue;
        R->cl_abundance[i] += (int)s->reads[raw]; R->cl_nunq[i]++;
        if (subs[raw].valid) { if (subs[raw].nsubs == 0) R->cl_n0[i] += (int)s->reads[raw]; if (subs[raw].nsubs == 1) R->cl_n1[i] += (int)s->reads[raw]; }
      }
      if (i == 0) {
        R->cl_birth_pval[i] = na_real(); R->cl_birth_from[i] = NA_INTEGER; R->cl_birth_fold[i] = na_real();
        R->cl_birth_ham[i] = NA_INTEGER; R->cl_birth_qave[i] = na_real();
      } else {
        R->cl_birth_from[i] = (int)b->birth_from + 1; R->cl_birth_pval[i] = b->birth_pval; R->cl_birth_fold[i] = b->birth_fold;
        R->cl_birth_ham[i] = (int)b->birth_comp.hamming;
        if (s->has_quals) {
          double q_ave = 0.0; int k;
          if (bsubs[i].valid) { for (k = 0; k < bsubs[i].nsubs; k++) q_ave += bsubs[i].q1[k]; q_ave = q_ave / ((double)bsubs[i].nsubs); }
          R->cl_birth_qave[i] = q_ave;
        } else R->cl_birth_qave[i] = na_real();
      }
      center_of[b->center] = i;
    }
    for (i = 0; i < C; i++)                                            /* post-hoc pval, error.cpp:101-119 */
      for (r = 0; r < s->bi[i].ncomp; r++) {
        int j = center_of[s->bi[i].comp[r].index];
        if (j >= 0 && j != i) tot_e[j] += s->bi[i].comp[r].lambda * s->bi[i].reads;
      }
    for (i = 0; i < C; i++) R->cl_pval[i] = oracle_calc_pA((int)s->reads[s->bi[i].center], tot_e[i], 1);
    free(tot_e); free(center_of);
  }
  /* b_make_transition_by_quality_matrix, error.cpp:131-172 ; cluster quality matrix :225-258 */
  R->subqual = (int *)calloc((size_t)16 * R->ncol, sizeof(int));
  R->clusterquals = (double *)calloc((size_t)maxlen * s->nclust, sizeof(double));
  for (i = 0; i < s->nclust; i++) {
    unsigned c = s->bi[i].center; int pos0, clen = s->len[c];
    unsigned *nreads = (unsigned *)calloc(maxlen, sizeof(unsigned));
    for (r = 0; r < s->bi[i].nraw; r++) {
      unsigned raw = s->bi[i].raw[r];
      Sub *sub = &subs[raw];
      if (!s->correct[raw] || !sub->valid) continue;
      for (pos0 = 0; pos0 < clen; pos0++) {
        unsigned pos1 = sub->map[pos0], t, q;
        if (pos1 == GAP_GLYPH) continue;
        t = 4u * (s->seq[c][pos0] - 1) + (s->seq[raw][pos1] - 1);
        q = s->has_quals ? s->qual[raw][pos1] : 0;
        R->subqual[(size_t)q * 16 + t] = (int)((unsigned)R->subqual[(size_t)q * 16 + t] + s->reads[raw]);
        if (s->has_quals) { nreads[pos0] += s->reads[raw]; R->clusterquals[(size_t)i * maxlen + pos0] += (double)(unsigned)(s->qual[raw][pos1] * s->reads[raw]); }
      }
    }
    if (s->has_quals) {
      for (pos0 = 0; pos0 < clen; pos0++) R->clusterquals[(size_t)i * maxlen + pos0] = R->clusterquals[(size_t)i * maxlen + pos0] / nreads[pos0];
      for (pos0 = clen; pos0 < maxlen; pos0++) R->clusterquals[(size_t)i * maxlen + pos0] = na_real();
    }
    free(nreads);
  }
  /* b_make_birth_subs_df, error.cpp:261-300 */
  for (i = 0; i < s->nclust; i++) if (bsubs[i].valid) R->nbirth_subs += bsubs[i].nsubs;
  R->bs_pos = (int *)malloc(sizeof(int) * (R->nbirth_subs + 1)); R->bs_clust = (int *)malloc(sizeof(int) * (R->nbirth_subs + 1));
  R->bs_ref = (char *)malloc(R->nbirth_subs + 1); R->bs_sub = (char *)malloc(R->nbirth_subs + 1);
  R->bs_qual = (double *)malloc(sizeof(double) * (R->nbirth_subs + 1));
  { int j = 0, k;
    for (i = 0; i < s->nclust; i++) if (bsubs[i].valid) for (k = 0; k < bsubs[i].nsubs; k++, j++) {
      R->bs_pos[j] = bsubs[i].pos[k] + 1; R->bs_ref[j] = NT[bsubs[i].nt0[k]]; R->bs_sub[j] = NT[bsubs[i].nt1[k]];
      R->bs_qual[j] = s->has_quals ? (double)bsubs[i].q1[k] : na_real(); R->bs_clust[j] = i + 1;
    } }
  /* Rmap, Rmain.cpp:268-279 */
  for (i = 0; i < s->nclust; i++) for (r = 0; r < s->bi[i].nraw; r++) { unsigned raw = s->bi[i].raw[r]; R->map[raw] = s->correct[raw] ? i + 1 : NA_INTEGER; }
  for (index = 0; index < nraw; index++) sub_release(&subs[index]);
  for (i = 0; i < s->nclust; i++) sub_release(&bsubs[i]);
  free(subs); free(bsubs); s_free(s);
  return R;
fail:
  if (subs) { for (index = 0; index < nraw; index++) sub_release(&subs[index]); free(subs); }
  if (bsubs) { for (i = 0; i < s->nclust; i++) sub_release(&bsubs[i]); free(bsubs); }
  s_free(s);
  return NULL;
#undef FAIL
}

/* ------------------------------------------------------- unit entry points --- */
static void encode(const char *a, uint8_t *o, int n) { int i; for (i = 0; i < n; i++) o[i] = a[i] == 'A' ? 1 : a[i] == 'C' ? 2 : a[i] == 'G' ? 3 : a[i] == 'T' ? 4 : 5; o[n] = 0; }
static void decode(const uint8_t *a, char *o, int n) { int i; for (i = 0; i < n; i++) o[i] = a[i] == '-' ? '-' : NT[a[i]]; o[n] = 0; }

/* gapless=1: nwalign_gapless; else the banded ends-free NW.  out0/out1 capacity len1+len2+1. */
int oracle_nwalign(const char *s1, const char *s2, int match, int mismatch, int gap, int band, int gapless,
                   char *out0, char *out1)
{
  int l1 = (int)strlen(s1), l2 = (int)strlen(s2), n;
  uint8_t *a = (uint8_t *)malloc(l1 + 1), *b = (uint8_t *)malloc(l2 + 1);
  uint8_t *al0 = (uint8_t *)malloc(l1 + l2 + 2), *al1 = (uint8_t *)malloc(l1 + l2 + 2);
  encode(s1, a, l1); encode(s2, b, l2);
  n = gapless ? nw_gapless(a, l1, b, l2, al0, al1) : nw_endsfree(a, l1, b, l2, match, mismatch, gap, band, -9999, al0, al1);
  decode(al0, out0, n); decode(al1, out1, n);
  free(a); free(b); free(al0); free(al1);
  return n;
}

/* C_nwalign (evaluate.cpp:18-62): any of the three scalar aligners.  out0/out1 capacity len1+len2+1. */
int oracle_nwalign2(const char *s1, const char *s2, int match, int mismatch, int gap, int homo_gap, int band, int endsfree,
                    char *out0, char *out1)
{
  int l1 = (int)strlen(s1), l2 = (int)strlen(s2), n;
  uint8_t *a = (uint8_t *)malloc(l1 + 1), *b = (uint8_t *)malloc(l2 + 1);
  uint8_t *al0 = (uint8_t *)malloc(l1 + l2 + 2), *al1 = (uint8_t *)malloc(l1 + l2 + 2);
  encode(s1, a, l1); encode(s2, b, l2);
  n = nw_general(a, l1, b, l2, match, mismatch, gap, endsfree ? homo_gap : gap, endsfree, band, -9999, al0, al1);
  decode(al0, out0, n); decode(al1, out1, n);
  free(a); free(b); free(al0); free(al1);
  return n;
}

/* One comparison (cluster.cpp:121-143 without the greedy skip): out = {lambda, hamming|-1, kdist, kodist}.
   err row-major [16][ncol]; cq/rq are the (unrounded) mean qualities or NULL. */
int oracle_compare(const char *cseq, const double *cq, const char *rseq, const double *rq, const double *err_rowmajor,
                   int ncol, const oracle_opts *o, double cutoff, double *out)
{
  S st, *s = &st; Sub sub; int k, pos; const char *sq[2]; const double *qq[2];
  memset(s, 0, sizeof *s);
  sq[0] = cseq; sq[1] = rseq; qq[0] = cq; qq[1] = rq;
  s->o = *o; s->nraw = 2; s->ncol = ncol; s->has_quals = cq && rq; s->err = err_rowmajor;
  s->seq = (uint8_t **)calloc(2, sizeof(uint8_t *)); s->qual = (uint8_t **)calloc(2, sizeof(uint8_t *)); s->len = (int *)malloc(2 * sizeof(int));
  for (k = 0; k < 2; k++) { s->len[k] = (int)strlen(sq[k]); if (s->len[k] > s->maxlen) s->maxlen = s->len[k]; }
  s->k8 = (uint8_t *)malloc(2 * NKMER); s->k16 = (uint16_t *)malloc(4 * NKMER); s->kord = (uint16_t *)calloc(2 * (size_t)s->maxlen, 2);
  for (k = 0; k < 2; k++) {
    s->seq[k] = (uint8_t *)malloc(s->len[k] + 1); encode(sq[k], s->seq[k], s->len[k]);
    if (s->has_quals) { s->qual[k] = (uint8_t *)malloc(s->len[k]); for (pos = 0; pos < s->len[k]; pos++) s->qual[k][pos] = (uint8_t)round(qq[k][pos]); }
    kmer_tables(s->seq[k], s->len[k], s->k16 + (size_t)k * NKMER, s->k8 + (size_t)k * NKMER, s->kord + (size_t)k * s->maxlen);
  }
  make_sub(s, 0, 1, o->use_kmers, cutoff, &sub, &out[2], &out[3]);
  out[0] = lambda_of(s, 1, &sub);
  out[1] = sub.valid ? (double)sub.nsubs : -1.0;
  sub_release(&sub);
  for (k = 0; k < 2; k++) { free(s->seq[k]); free(s->qual[k]); }
  free(s->seq); free(s->qual); free(s->len); free(s->k8); free(s->k16); free(s->kord);
  return out[0] < 0 ? 1 : 0;
}

/* ------------------------------------------------------------ bimera identification --- */
/* Restatement of /root/reference/src/chimera.cpp (the step after dada(): isBimeraDenovo[Table] in R/chimeras.R):
   C_is_bimera :18-59, BimeraTableParallel / C_table_bimera2 :61-208, get_ham_endsfree :211-239, get_lr :243-293.
   The alignment is the same banded ends-free NW as the denoising path, band = max_shift (chimera.cpp:26,122). */

/* get_lr (chimera.cpp:243-293).  al0 = gapped query, al1 = gapped parent, n columns. */
static void bim_get_lr(const uint8_t *al0, const uint8_t *al1, int n, int *left, int *right, int *left_oo, int *right_oo,
                       int allow_one_off, int max_shift)
{
  int pos = 0, l = 0, r = 0;
  while (pos < n && al0[pos] == '-') pos++;                           /* scan in until the query starts          */
  while (pos < n && al1[pos] == '-' && pos < max_shift) { pos++; l++; }   /* ends-free coverage until the parent starts */
  while (pos < n && al0[pos] == al1[pos]) { pos++; l++; }              /* covered until a mismatch                 */
  *left = l;
  if (allow_one_off) {
    int lo = l;
    pos++;
    if (pos < n && al0[pos] != '-') lo++;
    while (pos < n && al0[pos] == al1[pos]) { pos++; lo++; }
    *left_oo = lo;
  }
  pos = n - 1;
  while (pos >= 0 && al0[pos] == '-') pos--;
  /* (the reference compares `pos > len - max_shift` in size_t: never true when the alignment is shorter than max_shift) */
  while (pos >= 0 && al1[pos] == '-' && n >= max_shift && pos > n - max_shift) { pos--; r++; }
  while (pos >= 0 && al0[pos] == al1[pos]) { pos--; r++; }
  *right = r;
  if (allow_one_off) {
    int ro = r;
    pos--;
    if (pos >= 0 && al0[pos] != '-') ro++;
    while (pos >= 0 && al0[pos] == al1[pos]) { pos--; ro++; }
    *right_oo = ro;
  }
}

/* get_ham_endsfree (chimera.cpp:211-239): mismatching columns, end gaps not counted */
static int bim_ham_endsfree(const uint8_t *a, const uint8_t *b, int n)
{
  int i = 0, j = n - 1, pos, ham = 0, g1, g2;
  g1 = a[i] == '-'; g2 = b[i] == '-';
  while (g1 || g2) { i++; g1 = g1 && a[i] == '-'; g2 = g2 && b[i] == '-'; }
  g1 = a[j] == '-'; g2 = b[j] == '-';
  while (g1 || g2) { j--; g1 = g1 && a[j] == '-'; g2 = g2 && b[j] == '-'; }
  for (pos = i; pos <= j; pos++) if (a[pos] != b[pos]) ham++;
  return ham;
}

typedef struct { int left, right, left_oo, right_oo, ham; } BimPair;

static void bim_pair(const char *q, const char *p, int match, int mismatch, int gap_p, int max_shift, int allow_one_off, BimPair *o)
{
  int l1 = (int)strlen(q), l2 = (int)strlen(p), n;
  uint8_t *a = (uint8_t *)malloc(l1 + 1), *b = (uint8_t *)malloc(l2 + 1);
  uint8_t *al0 = (uint8_t *)malloc(l1 + l2 + 2), *al1 = (uint8_t *)malloc(l1 + l2 + 2);
  encode(q, a, l1); encode(p, b, l2);
  n = nw_endsfree(a, l1, b, l2, match, mismatch, gap_p, max_shift, -9999, al0, al1);
  al0[n] = 0; al1[n] = 0;
  o->left_oo = o->right_oo = 0;
  bim_get_lr(al0, al1, n, &o->left, &o->right, &o->left_oo, &o->right_oo, allow_one_off, max_shift);
  o->ham = bim_ham_endsfree(al0, al1, n);
  free(a); free(b); free(al0); free(al1);
}

/* the per-pair quantities of the two entry points below, for n pairs: out[5 i ..] = left, right, left_oo, right_oo, ham */
void oracle_bimera_pairs(int n, const char *const *queries, const char *const *parents, int allow_one_off, int match, int mismatch,
                         int gap_p, int max_shift, int *out)
{
  int i;
  for (i = 0; i < n; i++) {
    BimPair b;
    bim_pair(queries[i], parents[i], match, mismatch, gap_p, max_shift, allow_one_off, &b);
    out[5 * i] = b.left; out[5 * i + 1] = b.right; out[5 * i + 2] = b.left_oo; out[5 * i + 3] = b.right_oo; out[5 * i + 4] = b.ham;
  }
}

/* C_is_bimera (chimera.cpp:18-59) */
int oracle_is_bimera(const char *sq, int npars, const char *const *pars, int allow_one_off, int min_one_off_par_dist, int match,
                     int mismatch, int gap_p, int max_shift)
{
  int i, sqlen = (int)strlen(sq), rval = 0;
  int max_left = 0, max_right = 0, oml = 0, omr = 0, omlo = 0, omro = 0;
  for (i = 0; i < npars && !rval; i++) {
    BimPair b;
    bim_pair(sq, pars[i], match, mismatch, gap_p, max_shift, allow_one_off, &b);
    if (b.left + b.right >= sqlen) continue;                         /* id / pure-shift / internal-indel "parents" */
    if (b.left > max_left) max_left = b.left;
    if (b.right > max_right) max_right = b.right;
    if (allow_one_off && b.ham >= min_one_off_par_dist) {
      if (b.left > oml) oml = b.left;
      if (b.right > omr) omr = b.right;
      if (b.left_oo > omlo) omlo = b.left_oo;
      if (b.right_oo > omro) omro = b.right_oo;
    }
    if (max_right + max_left >= sqlen) rval = 1;
    if (allow_one_off && (oml + omro >= sqlen || omlo + omr >= sqlen)) rval = 1;
  }
  return rval;
}

/* C_table_bimera2 (chimera.cpp:61-208): mat = nrow (samples) x ncol (sequences), column-major */
int oracle_table_bimera2(int nrow, int ncol, const int *mat, const char *const *seqs, double min_fold, int min_abund,
                         int allow_one_off, int min_one_off_par_dist, int match, int mismatch, int gap_p, int max_shift,
                         int *nflag_out, int *nsam_out)
{
  int i, j, k;
  int *lefts = (int *)malloc(sizeof(int) * ncol), *rights = (int *)malloc(sizeof(int) * ncol);
  int *lefts_oo = (int *)malloc(sizeof(int) * ncol), *rights_oo = (int *)malloc(sizeof(int) * ncol);
  char *allowed = (char *)malloc(ncol);
  for (j = 0; j < ncol; j++) {
    int nsam = 0, nflag = 0, sqlen = (int)strlen(seqs[j]);
    for (k = 0; k < ncol; k++) { lefts[k] = rights[k] = lefts_oo[k] = rights_oo[k] = -1; allowed[k] = 0; }
    for (i = 0; i < nrow; i++) {
      int max_left = 0, max_right = 0, oml = 0, omr = 0, omlo = 0, omro = 0;
      if (mat[i + (size_t)j * nrow] <= 0) continue;
      nsam++;
      for (k = 0; k < ncol; k++) {
        if (mat[i + (size_t)k * nrow] > (min_fold * mat[i + (size_t)j * nrow]) && mat[i + (size_t)k * nrow] >= min_abund) {
          if (lefts[k] < 0) {
            BimPair b;
            bim_pair(seqs[j], seqs[k], match, mismatch, gap_p, max_shift, allow_one_off, &b);
            if (allow_one_off && b.ham >= min_one_off_par_dist) allowed[k] = 1;
            if (b.left + b.right < sqlen) { lefts[k] = b.left; rights[k] = b.right; lefts_oo[k] = b.left_oo; rights_oo[k] = b.right_oo; }
            else { lefts[k] = rights[k] = lefts_oo[k] = rights_oo[k] = 0; }
          }
          if (lefts[k] > max_left) max_left = lefts[k];
          if (rights[k] > max_right) max_right = rights[k];
          if (allow_one_off && allowed[k]) {
            if (lefts[k] > oml) oml = lefts[k];
            if (rights[k] > omr) omr = rights[k];
            if (lefts_oo[k] > omlo) omlo = lefts_oo[k];
            if (rights_oo[k] > omro) omro = rights_oo[k];
          }
        }
      }
      if (max_right + max_left >= sqlen) nflag++;
      else if (allow_one_off && (oml + omro >= sqlen || omlo + omr >= sqlen)) nflag++;
    }
    nflag_out[j] = nflag; nsam_out[j] = nsam;
  }
  free(lefts); free(rights); free(lefts_oo); free(rights_oo); free(allowed);
  return 0;
}

/* ---- mergePairs helpers (the step after dada() on paired reads; R/paired.R:159-169) ------------------------------- */
/* C_eval_pair (evaluate.cpp:73-114): matches / mismatches / indels of the internal part of an alignment (end gaps of
 * either string are not counted).  out = {match, mismatch, indel}; returns 1 for strings of different length. */
int oracle_eval_pair(const char *s1, const char *s2, int *out)
{
  int n = (int)strlen(s1), start, end, i, s1gap, s2gap, match = 0, mismatch = 0, indel = 0;
  if ((int)strlen(s2) != n) return 1;
  s1gap = s2gap = 1;
  start = -1;
  do {                                                               /* evaluate.cpp:82-89 (s[n] is the terminator) */
    start++;
    s1gap = s1gap && (s1[start] == '-');
    s2gap = s2gap && (s2[start] == '-');
  } while ((s1gap || s2gap) && start < n);
  s1gap = s2gap = 1;
  end = n;
  do {                                                               /* evaluate.cpp:91-98 */
    end--;
    if (end < 0) break;                                              /* (the reference would index s[-1] here) */
    s1gap = s1gap && (s1[end] == '-');
    s2gap = s2gap && (s2[end] == '-');
  } while ((s1gap || s2gap) && end >= start);
  for (i = start; i <= end; i++) {                                   /* evaluate.cpp:101-110 */
    if (s1[i] == '-' || s2[i] == '-') indel++;
    else if (s1[i] == s2[i]) match++;
    else mismatch++;
  }
  out[0] = match; out[1] = mismatch; out[2] = indel;
  return 0;
}

/* C_pair_consensus (evaluate.cpp:124-174): the merged sequence; `prefer` (1 = s1, 2 = s2) wins mismatches, overhangs
 * (s2 running past the start of s1, s1 past the end of s2) are cut when asked.  out needs strlen(s1)+1 bytes. */
int oracle_pair_consensus(const char *s1, const char *s2, int prefer, int trim_overhang, char *out)
{
  int n = (int)strlen(s1), i, j = 0;
  if ((int)strlen(s2) != n) return 1;
  for (i = 0; i < n; i++) {
    if (s1[i] == s2[i]) out[i] = s1[i];
    else if (s2[i] == '-') out[i] = s1[i];
    else if (s1[i] == '-') out[i] = s2[i];
    else out[i] = prefer == 1 ? s1[i] : (prefer == 2 ? s2[i] : 'N');
  }
  if (trim_overhang) {
    for (i = 0; i < n; i++) { if (s1[i] != '-') break; out[i] = '-'; }
    for (i = n - 1; i >= 0; i--) { if (s2[i] != '-') break; out[i] = '-'; }
  }
  for (i = 0; i < n; i++) if (out[i] != '-') out[j++] = out[i];
  out[j] = 0;
  return 0;
}
