// tests/glue/glue_capi.cpp — TEST INFRASTRUCTURE ONLY.
//
// Flat C entry points over the Rcpp glue TU of INTEGRATION.md (tests/glue/dada2hip_glue.cpp), compiled against the
// from-scratch Rcpp model of oracle/shim/Rcpp.h and linked with dada2_amd/libdadahip.so: the reference's own export
// signatures (/root/reference/src/RcppExports.cpp:18-53, :94, :227; Rmain.cpp:30-47) called the way R's .Call glue calls
// them, the Rcpp::List that comes back flattened exactly as oracle/ref_capi.cpp flattens the reference's - so that
// oracle/ref.py reads both through the same code (flavour "glue") and tests/test_glue.py can compare them field by field.
#include <Rcpp.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

Rcpp::List dada_uniques(std::vector<std::string> seqs, std::vector<int> abundances, std::vector<bool> priors,
                        Rcpp::NumericMatrix err, Rcpp::NumericMatrix quals, int match, int mismatch, int gap,
                        bool use_kmers, double kdist_cutoff, int band_size, double omegaA, double omegaP,
                        double omegaC, bool detect_singletons, int max_clust, double min_fold, int min_hamming,
                        int min_abund, bool use_quals, bool final_consensus, bool vectorized_alignment,
                        int homo_gap, bool multithread, bool verbose, int SSE, bool gapless, bool greedy);
Rcpp::CharacterVector C_nwalign(std::string s1, std::string s2, int match, int mismatch, int gap_p, int homo_gap_p, int band, bool endsfree);
Rcpp::CharacterVector C_nwvec(std::vector<std::string> s1, std::vector<std::string> s2, int16_t match, int16_t mismatch,
                              int16_t gap_p, int band, bool endsfree);
Rcpp::DataFrame C_table_bimera2(Rcpp::IntegerMatrix mat, std::vector<std::string> seqs, double min_fold, int min_abund,
                                bool allow_one_off, int min_one_off_par_dist, int match, int mismatch, int gap_p, int max_shift);

// the two CPU functions a maintainer keeps from the reference for strings with letters outside ACGT: not part of this test
Rcpp::CharacterVector C_nwalign_cpu(std::string, std::string, int, int, int, int, int, bool) { Rcpp::stop("C_nwalign_cpu: the reference's body is not linked into the glue test"); return Rcpp::CharacterVector(); }
Rcpp::CharacterVector C_nwvec_cpu(std::vector<std::string>, std::vector<std::string>, int16_t, int16_t, int16_t, int, bool) { Rcpp::stop("C_nwvec_cpu: the reference's body is not linked into the glue test"); return Rcpp::CharacterVector(); }

extern "C" {

int dada2_shim_verbose = 0;
double dada2_oracle_ppois(double, double, int) { return 0.0; }   // (Rcpp::ppois of the shim: unused by the glue)

// same layout as include/dada2hip.h : dada2hip_opts
struct glue_opts {
  double kdist_cutoff, omegaA, omegaP, omegaC, min_fold;
  int32_t match, mismatch, gap, homo_gap, band_size, max_clust, min_hamming, min_abund;
  int32_t use_kmers, detect_singletons, use_quals, final_consensus, vectorized_alignment;
  int32_t multithread, verbose, SSE, gapless, greedy;
};
struct glue_result { Rcpp::List res; };

void *ref_dada_uniques(int nraw, const char *const *seqs, const int *abund, const unsigned char *priors, const double *err,
                       int err_ncol, const double *quals, int quals_nrow, const glue_opts *o, char *errbuf, int errlen) {
  try {
    std::vector<std::string> s(nraw);
    std::vector<int> a(nraw);
    std::vector<bool> p(nraw);
    for (int i = 0; i < nraw; i++) { s[i] = seqs[i]; a[i] = abund[i]; p[i] = priors ? priors[i] != 0 : false; }
    Rcpp::NumericMatrix E(16, err_ncol);
    memcpy(E.v().data(), err, sizeof(double) * 16 * (size_t)err_ncol);
    Rcpp::NumericMatrix Q(quals ? quals_nrow : 0, quals ? nraw : 0);
    if (quals) memcpy(Q.v().data(), quals, sizeof(double) * (size_t)quals_nrow * (size_t)nraw);
    glue_result *r = new glue_result;
    r->res = dada_uniques(s, a, p, E, Q, o->match, o->mismatch, o->gap, o->use_kmers, o->kdist_cutoff, o->band_size, o->omegaA,
                          o->omegaP, o->omegaC, o->detect_singletons, o->max_clust, o->min_fold, o->min_hamming, o->min_abund,
                          o->use_quals, o->final_consensus, o->vectorized_alignment, o->homo_gap, o->multithread, o->verbose, o->SSE,
                          o->gapless, o->greedy);
    return r;
  } catch (std::exception &e) {
    if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what());
    return NULL;
  }
}
void ref_result_free(void *h) { delete (glue_result *)h; }
static const Rcpp::RObj *glue_find(void *h, const char *a, const char *b) {
  const Rcpp::RObj *o = ((glue_result *)h)->res.obj->get(a);
  if (o && b && b[0]) o = o->get(b);
  return o;
}
// kind: 0 int, 1 double, 2 string, 3 int matrix, 4 double matrix; returns element count (or -1)
long ref_result_get(void *h, const char *a, const char *b, int *kind, int *nr, int *nc, const void **data) {
  const Rcpp::RObj *o = glue_find(h, a, b);
  if (!o) return -1;
  *kind = (int)o->kind; *nr = o->nr; *nc = o->nc;
  switch (o->kind) {
    case Rcpp::RObj::INT: case Rcpp::RObj::IMAT: *data = o->iv.data(); return (long)o->iv.size();
    case Rcpp::RObj::DBL: case Rcpp::RObj::DMAT: *data = o->dv.data(); return (long)o->dv.size();
    case Rcpp::RObj::STR: *data = NULL; return (long)o->sv.size();
    default: return -1;
  }
}
const char *ref_result_str(void *h, const char *a, const char *b, long i) {
  const Rcpp::RObj *o = glue_find(h, a, b);
  if (!o || o->kind != Rcpp::RObj::STR || i < 0 || (size_t)i >= o->sv.size()) return NULL;
  return o->sv[i].c_str();
}

// C_nwalign / C_nwvec through the glue: out0 / out1 sized by the caller (len1 + len2 + 1 each)
int glue_nwalign(const char *s1, const char *s2, int match, int mismatch, int gap, int homo_gap, int band, int endsfree, char *out0,
                 char *out1, char *errbuf, int errlen) {
  try {
    Rcpp::CharacterVector r = C_nwalign(s1, s2, match, mismatch, gap, homo_gap, band, endsfree != 0);
    strcpy(out0, r[0].c_str()); strcpy(out1, r[1].c_str());
    return 0;
  } catch (std::exception &e) { if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what()); return 1; }
}
int glue_nwvec(int n, const char *const *s1, const char *const *s2, int match, int mismatch, int gap, int band, int endsfree, char **out,
               char *errbuf, int errlen) {
  try {
    std::vector<std::string> a(s1, s1 + n), b(s2, s2 + n);
    Rcpp::CharacterVector r = C_nwvec(a, b, (int16_t)match, (int16_t)mismatch, (int16_t)gap, band, endsfree != 0);
    for (int i = 0; i < 2 * n; i++) strcpy(out[i], r[i].c_str());
    return 0;
  } catch (std::exception &e) { if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what()); return 1; }
}
int glue_table_bimera2(int nrow, int ncol, const int *mat, const char *const *seqs, double min_fold, int min_abund, int allow_one_off,
                       int min_one_off_par_dist, int match, int mismatch, int gap, int max_shift, int *nflag, int *nsam, char *errbuf,
                       int errlen) {
  try {
    Rcpp::IntegerMatrix M(nrow, ncol);
    memcpy(M.v().data(), mat, sizeof(int) * (size_t)nrow * (size_t)ncol);
    std::vector<std::string> s(seqs, seqs + ncol);
    Rcpp::DataFrame d = C_table_bimera2(M, s, min_fold, min_abund, allow_one_off != 0, min_one_off_par_dist, match, mismatch, gap, max_shift);
    const Rcpp::RObj *f = d.obj->get("nflag"), *m = d.obj->get("nsam");
    for (int j = 0; j < ncol; j++) { nflag[j] = f->iv[j]; nsam[j] = m->iv[j]; }
    return 0;
  } catch (std::exception &e) { if (errbuf && errlen > 0) snprintf(errbuf, errlen, "%s", e.what()); return 1; }
}

}  // extern "C"
