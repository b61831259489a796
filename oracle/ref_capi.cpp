// oracle/ref_capi.cpp — TEST INFRASTRUCTURE ONLY.
//
// Flat C entry points over the REFERENCE's own code, compiled in place from
// /root/reference/src by oracle/Makefile into oracle/_ref/libdada2ref.so.
// This file contains no algorithm: it marshals plain arrays into the calls
//   dada_uniques()            /root/reference/src/Rmain.cpp:30
//   nwalign_endsfree()        /root/reference/src/nwalign_endsfree.cpp:76
//   nwalign_vectorized2()     /root/reference/src/nwalign_vectorized.cpp:71
//   sub_new()/compute_lambda_ts()  nwalign_endsfree.cpp:642 / pval.cpp:144
//   kmer_dist*/kord_dist*     /root/reference/src/kmers.cpp
// and hands results back by name.  Used (a) to validate the CPU restatement in
// oracle/dada_oracle.c, (b) to generate tests/golden/*, (c) as bench.py's
// cpu_baseline (kind "reference").
#include "dada.h"
#include <string>
#include <vector>

Rcpp::List dada_uniques(std::vector<std::string> seqs, std::vector<int> abundances, std::vector<bool> priors,
                        Rcpp::NumericMatrix err, Rcpp::NumericMatrix quals, int match, int mismatch, int gap,
                        bool use_kmers, double kdist_cutoff, int band_size, double omegaA, double omegaP,
                        double omegaC, bool detect_singletons, int max_clust, double min_fold, int min_hamming,
                        int min_abund, bool use_quals, bool final_consensus, bool vectorized_alignment,
                        int homo_gap, bool multithread, bool verbose, int SSE, bool gapless, bool greedy);

// chimera.cpp:18,192 (the step after dada(): bimera identification on the sequence table)
// chimera.cpp:7-8 (external linkage): what C_is_bimera reduces an alignment to
int get_ham_endsfree(const char *seq1, const char *seq2);
void get_lr(char **al, int &left, int &right, int &left_oo, int &right_oo, bool allow_one_off, int max_shift);
bool C_is_bimera(std::string sq, std::vector<std::string> pars, bool allow_one_off, int min_one_off_par_dist, int match,
                 int mismatch, int gap_p, int max_shift);
Rcpp::DataFrame C_table_bimera2(Rcpp::IntegerMatrix mat, std::vector<std::string> seqs, double min_fold, int min_abund,
                                bool allow_one_off, int min_one_off_par_dist, int match, int mismatch, int gap_p, int max_shift);

// evaluate.cpp:18,73,124 (the R-visible aligner and the two helpers mergePairs() calls, R/paired.R:159-169)
Rcpp::CharacterVector C_nwalign(std::string s1, std::string s2, int match, int mismatch, int gap_p, int homo_gap_p, int band, bool endsfree);
Rcpp::IntegerVector C_eval_pair(std::string s1, std::string s2);
Rcpp::CharacterVector C_pair_consensus(std::string s1, std::string s2, int prefer, bool trim_overhang);

extern "C" {

int dada2_shim_verbose = 0;
int dada2_shim_nthreads = 1;
void dada2_shim_set_verbose(int v) { dada2_shim_verbose = v; }

}  // extern "C"

// ---- persistent worker pool behind the shim's parallelFor (what TBB is behind the real RcppParallel) ----------------
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unistd.h>
namespace {
struct ShimPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  unsigned long gen = 0;
  int pending = 0;
  bool stop = false;
  pid_t pid = 0;
  // the job
  std::atomic<std::size_t> next{0};
  std::size_t end = 0, chunk = 1;
  void (*fn)(void *, std::size_t, std::size_t) = nullptr;
  void *ctx = nullptr;
  void work() {
    for (;;) {
      const std::size_t b = next.fetch_add(chunk);
      if (b >= end) break;
      fn(ctx, b, std::min(end, b + chunk));
    }
  }
  void loop(unsigned long seen) {
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return gen != seen; });
        seen = gen;
        if (stop) return;
      }
      work();
      {
        std::lock_guard<std::mutex> g(mu);
        if (--pending == 0) done_cv.notify_all();
      }
    }
  }
  void start(int nthreads) {
    pid = getpid();
    for (int t = 1; t < nthreads; t++) th.emplace_back([this, g0 = gen] { loop(g0); });
  }
  void shutdown() {
    { std::lock_guard<std::mutex> g(mu); stop = true; gen++; }
    cv.notify_all();
    for (auto &t : th) t.join();
  }
};
ShimPool *g_pool = nullptr;
std::mutex g_pool_mu;
}  // namespace

extern "C" {

void dada2_shim_set_threads(int n) {
  n = n < 1 ? 1 : n;
  std::lock_guard<std::mutex> g(g_pool_mu);
  if (g_pool && (g_pool->pid != getpid())) g_pool = nullptr;   // forked child: the parent's threads are not ours (leaked)
  if (g_pool && (int)g_pool->th.size() + 1 != n) { g_pool->shutdown(); delete g_pool; g_pool = nullptr; }
  if (!g_pool && n > 1) { g_pool = new ShimPool(); g_pool->start(n); }
  dada2_shim_nthreads = n;
}

void dada2_shim_parallel_for(std::size_t begin, std::size_t end, std::size_t chunk, void (*fn)(void *, std::size_t, std::size_t),
                             void *ctx) {
  ShimPool *p = nullptr;
  { std::lock_guard<std::mutex> g(g_pool_mu); p = (g_pool && g_pool->pid == getpid()) ? g_pool : nullptr; }
  if (!p) { if (end > begin) fn(ctx, begin, end); return; }
  {
    std::lock_guard<std::mutex> g(p->mu);
    p->next.store(begin); p->end = end; p->chunk = chunk ? chunk : 1; p->fn = fn; p->ctx = ctx;
    p->pending = (int)p->th.size();
    p->gen++;
  }
  p->cv.notify_all();
  p->work();
  std::unique_lock<std::mutex> l(p->mu);
  p->done_cv.wait(l, [&] { return p->pending == 0; });
}

// Same layout as include/dada2hip.h : dada2hip_opts (the scalars of Rmain.cpp:33-47;
// doubles first so there is no padding: 5*8 + 18*4 = 112 bytes).
struct ref_opts {
  double kdist_cutoff, omegaA, omegaP, omegaC, min_fold;
  int32_t match, mismatch, gap, homo_gap, band_size, max_clust, min_hamming, min_abund;
  int32_t use_kmers, detect_singletons, use_quals, final_consensus, vectorized_alignment;
  int32_t multithread, verbose, SSE, gapless, greedy;
};

struct ref_result {
  Rcpp::List res;
};

void *ref_dada_uniques(int nraw, const char *const *seqs, const int *abund, const unsigned char *priors,
                       const double *err, int err_ncol, const double *quals, int quals_nrow,
                       const ref_opts *o, char *errbuf, int errlen) {
  try {
    std::vector<std::string> s(nraw);
    std::vector<int> a(nraw);
    std::vector<bool> p(nraw);
    for (int i = 0; i < nraw; i++) {
      s[i] = seqs[i];
      a[i] = abund[i];
      p[i] = priors ? priors[i] != 0 : false;
    }
    Rcpp::NumericMatrix E(16, err_ncol);
    memcpy(E.v().data(), err, sizeof(double) * 16 * (size_t)err_ncol);
    Rcpp::NumericMatrix Q(quals ? quals_nrow : 0, quals ? nraw : 0);
    if (quals) memcpy(Q.v().data(), quals, sizeof(double) * (size_t)quals_nrow * (size_t)nraw);
    ref_result *r = new ref_result;
    r->res = dada_uniques(s, a, p, E, Q, o->match, o->mismatch, o->gap, o->use_kmers, o->kdist_cutoff,
                          o->band_size, o->omegaA, o->omegaP, o->omegaC, o->detect_singletons, o->max_clust,
                          o->min_fold, o->min_hamming, o->min_abund, o->use_quals, o->final_consensus,
                          o->vectorized_alignment, o->homo_gap, o->multithread, o->verbose, o->SSE, o->gapless,
                          o->greedy);
    return r;
  } catch (std::exception &e) {
    if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what());
    return NULL;
  }
}

void ref_result_free(void *h) { delete (ref_result *)h; }

static const Rcpp::RObj *ref_find(void *h, const char *a, const char *b) {
  ref_result *r = (ref_result *)h;
  const Rcpp::RObj *o = r->res.obj->get(a);
  if (o && b && b[0]) o = o->get(b);
  return o;
}

// kind: 0 int, 1 double, 2 string, 3 int matrix, 4 double matrix; returns element count (or -1)
long ref_result_get(void *h, const char *a, const char *b, int *kind, int *nr, int *nc, const void **data) {
  const Rcpp::RObj *o = ref_find(h, a, b);
  if (!o) return -1;
  *kind = (int)o->kind;
  *nr = o->nr;
  *nc = o->nc;
  switch (o->kind) {
    case Rcpp::RObj::INT: case Rcpp::RObj::IMAT: *data = o->iv.data(); return (long)o->iv.size();
    case Rcpp::RObj::DBL: case Rcpp::RObj::DMAT: *data = o->dv.data(); return (long)o->dv.size();
    case Rcpp::RObj::STR: *data = NULL; return (long)o->sv.size();
    default: return -1;
  }
}
const char *ref_result_str(void *h, const char *a, const char *b, long i) {
  const Rcpp::RObj *o = ref_find(h, a, b);
  if (!o || o->kind != Rcpp::RObj::STR || i < 0 || (size_t)i >= o->sv.size()) return NULL;
  return o->sv[i].c_str();
}

// which: 0 = nwalign_endsfree, 1 = nwalign_vectorized2 (end_gap 0), 2 = nwalign_gapless,
//        3 = nwalign (global, ends penalised).  ACGT in, gapped ACGT out.
int ref_nwalign(const char *s1, const char *s2, int match, int mismatch, int gap, int band, int which,
                char *out0, char *out1, char *errbuf, int errlen) {
  try {
    size_t l1 = strlen(s1), l2 = strlen(s2);
    std::vector<char> a(l1 + 1), b(l2 + 1);
    nt2int(a.data(), s1);
    nt2int(b.data(), s2);
    int score[4][4];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) score[i][j] = i == j ? match : mismatch;
    char **al;
    if (which == 0) al = nwalign_endsfree(a.data(), l1, b.data(), l2, score, gap, band);
    else if (which == 1) al = nwalign_vectorized2(a.data(), l1, b.data(), l2, (int16_t)match, (int16_t)mismatch, (int16_t)gap, 0, band);
    else if (which == 2) al = nwalign_gapless(a.data(), l1, b.data(), l2);
    else al = nwalign(a.data(), l1, b.data(), l2, score, gap, band);
    int2nt(out0, al[0]);
    int2nt(out1, al[1]);
    free(al[0]); free(al[1]); free(al);
    return 0;
  } catch (std::exception &e) {
    if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what());
    return 1;
  }
}

// C_nwvec's call (nwalign_vectorized.cpp:321-343) on ONE pair: the strings' raw bytes go to nwalign_vectorized2 as they are (no
// nt2int), end gaps free when `endsfree`, the band cast to size_t as there.  out0/out1: len1 + len2 + 1 bytes each.
int ref_nwvec_raw(const char *s1, const char *s2, int match, int mismatch, int gap_p, int band, int endsfree, char *out0, char *out1,
                  char *errbuf, int errlen) {
  try {
    char **al = nwalign_vectorized2(s1, strlen(s1), s2, strlen(s2), (int16_t)match, (int16_t)mismatch, (int16_t)gap_p,
                                    endsfree ? (int16_t)0 : (int16_t)gap_p, (size_t)band);
    strcpy(out0, al[0]);
    strcpy(out1, al[1]);
    free(al[0]); free(al[1]); free(al);
    return 0;
  } catch (std::exception &e) {
    if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what());
    return 1;
  }
}

static Raw *ref_make_raw(const char *s, const double *q, unsigned reads, size_t maxlen,
                         std::vector<uint8_t> &k8, std::vector<uint16_t> &k16, std::vector<uint16_t> &ko) {
  size_t l = strlen(s);
  std::vector<char> a(l + 1);
  nt2int(a.data(), s);
  Raw *r = raw_new(a.data(), (double *)q, reads, false);
  k8.assign(1024, 0); k16.assign(1024, 0); ko.assign(maxlen, 0);
  r->kmer8 = k8.data(); r->kmer = k16.data(); r->kord = ko.data();
  assign_kmer8(r->kmer8, r->seq, KMER_SIZE);
  assign_kmer(r->kmer, r->seq, KMER_SIZE);
  assign_kmer_order(r->kord, r->seq, KMER_SIZE);
  return r;
}

// One comparison exactly as CompareParallel does it (cluster.cpp:121-143, without the greedy skip):
// out[0]=lambda out[1]=hamming(-1 if shrouded) out[2]=kdist(SSE=2 path) out[3]=kodist
int ref_compare(const char *cseq, const double *cq, const char *rseq, const double *rq, const double *err_rowmajor,
                int ncol, const ref_opts *o, double kdist_cutoff, double *out, char *errbuf, int errlen) {
  try {
    size_t maxlen = std::max(strlen(cseq), strlen(rseq));
    std::vector<uint8_t> a8, b8; std::vector<uint16_t> a16, b16, ao, bo;
    Raw *c = ref_make_raw(cseq, cq, 1, maxlen, a8, a16, ao);
    Raw *r = ref_make_raw(rseq, rq, 1, maxlen, b8, b16, bo);
    Sub *sub = sub_new(c, r, o->match, o->mismatch, o->gap, o->homo_gap, o->use_kmers, kdist_cutoff, o->band_size,
                       o->vectorized_alignment, o->SSE, o->gapless);
    out[0] = compute_lambda_ts(r, sub, ncol, (double *)err_rowmajor, o->use_quals);
    out[1] = sub ? (double)sub->nsubs : -1.0;
    double kd = kmer_dist_SSEi_8(c->kmer8, c->length, r->kmer8, r->length, KMER_SIZE);
    if (kd < 0) kd = kmer_dist_SSEi(c->kmer, c->length, r->kmer, r->length, KMER_SIZE);
    out[2] = kd;
    out[3] = kord_dist_SSEi(c->kord, c->length, r->kord, r->length, KMER_SIZE);
    sub_free(sub);
    raw_free(c); raw_free(r);
    return 0;
  } catch (std::exception &e) {
    if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what());
    return 1;
  }
}

double ref_calc_pA(int reads, double E_reads, int prior) { return calc_pA(reads, E_reads, prior != 0); }

// C_table_bimera2 (chimera.cpp:192): mat is nrow (samples) x ncol (sequences), column-major; fills nflag[ncol], nsam[ncol]
int ref_table_bimera2(int nrow, int ncol, const int *mat, const char *const *seqs, double min_fold, int min_abund,
                      int allow_one_off, int min_one_off_par_dist, int match, int mismatch, int gap_p, int max_shift,
                      int *nflag, int *nsam, char *errbuf, int errlen) {
  try {
    Rcpp::IntegerMatrix M(nrow, ncol);
    memcpy(M.v().data(), mat, sizeof(int) * (size_t)nrow * (size_t)ncol);
    std::vector<std::string> s(ncol);
    for (int i = 0; i < ncol; i++) s[i] = seqs[i];
    Rcpp::DataFrame df = C_table_bimera2(M, s, min_fold, min_abund, allow_one_off != 0, min_one_off_par_dist, match, mismatch,
                                         gap_p, max_shift);
    const Rcpp::RObj *f = df.obj->get("nflag"), *n = df.obj->get("nsam");
    for (int i = 0; i < ncol; i++) { nflag[i] = f->iv[i]; nsam[i] = n->iv[i]; }
    return 0;
  } catch (std::exception &e) {
    if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what());
    return 1;
  }
}

// the reference's own get_lr / get_ham_endsfree (chimera.cpp:243-293, :196-239; external linkage) on its own alignment, as
// C_is_bimera calls them (chimera.cpp:26-36): out[5 i ..] = left, right, left_oo, right_oo, ham
int ref_bimera_pairs(int n, const char *const *queries, const char *const *parents, int allow_one_off, int match, int mismatch,
                     int gap_p, int max_shift, int *out) {
  try {
    for (int i = 0; i < n; i++) {
      char **al = nwalign_vectorized2(queries[i], strlen(queries[i]), parents[i], strlen(parents[i]), (int16_t)match,
                                      (int16_t)mismatch, (int16_t)gap_p, 0, max_shift);
      int left = 0, right = 0, left_oo = 0, right_oo = 0;
      get_lr(al, left, right, left_oo, right_oo, allow_one_off != 0, max_shift);
      out[5 * i] = left; out[5 * i + 1] = right; out[5 * i + 2] = left_oo; out[5 * i + 3] = right_oo;
      out[5 * i + 4] = get_ham_endsfree(al[0], al[1]);
      free(al[0]); free(al[1]); free(al);
    }
    return 0;
  } catch (std::exception &) {
    return -1;
  }
}

// C_is_bimera (chimera.cpp:18); returns 0 / 1, or -1 on error
int ref_is_bimera(const char *sq, int npars, const char *const *pars, int allow_one_off, int min_one_off_par_dist, int match,
                  int mismatch, int gap_p, int max_shift) {
  try {
    std::vector<std::string> p(npars);
    for (int i = 0; i < npars; i++) p[i] = pars[i];
    return C_is_bimera(sq, p, allow_one_off != 0, min_one_off_par_dist, match, mismatch, gap_p, max_shift) ? 1 : 0;
  } catch (std::exception &) {
    return -1;
  }
}

// C_nwalign (evaluate.cpp:18) itself: out0/out1 need len1+len2+1 bytes each
int ref_C_nwalign(const char *s1, const char *s2, int match, int mismatch, int gap_p, int homo_gap_p, int band, int endsfree,
                  char *out0, char *out1, char *errbuf, int errlen) {
  try {
    Rcpp::CharacterVector r = C_nwalign(s1, s2, match, mismatch, gap_p, homo_gap_p, band, endsfree != 0);
    strcpy(out0, r[0].c_str());
    strcpy(out1, r[1].c_str());
    return 0;
  } catch (std::exception &e) {
    if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what());
    return 1;
  }
}

// C_eval_pair (evaluate.cpp:73): out = {match, mismatch, indel}; returns 1 when the reference returned NULL
int ref_eval_pair(const char *s1, const char *s2, int *out) {
  Rcpp::IntegerVector r = C_eval_pair(s1, s2);
  if (r.is_null || r.size() != 3) return 1;
  out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
  return 0;
}

// C_pair_consensus (evaluate.cpp:124): out needs strlen(s1)+1 bytes; returns 1 when the reference returned NULL
int ref_pair_consensus(const char *s1, const char *s2, int prefer, int trim_overhang, char *out) {
  Rcpp::CharacterVector r = C_pair_consensus(s1, s2, prefer, trim_overhang != 0);
  if (r.is_null || r.size() != 1) return 1;
  strcpy(out, r[0].c_str());
  return 0;
}

}  // extern "C"
