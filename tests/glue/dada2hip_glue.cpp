// src/dada2hip_glue.cpp  — add to the package, remove Rmain.cpp's dada_uniques body
#include <Rcpp.h>
#include "dada2hip.h"
using namespace Rcpp;

static void hip_log(const char *m, void *) { Rprintf("%s", m); }
static int  hip_abort(void *) { try { Rcpp::checkUserInterrupt(); } catch (...) { return 1; } return 0; }

// [[Rcpp::export]]
Rcpp::List dada_uniques(std::vector<std::string> seqs, std::vector<int> abundances, std::vector<bool> priors,
                        Rcpp::NumericMatrix err, Rcpp::NumericMatrix quals, int match, int mismatch, int gap,
                        bool use_kmers, double kdist_cutoff, int band_size, double omegaA, double omegaP,
                        double omegaC, bool detect_singletons, int max_clust, double min_fold, int min_hamming,
                        int min_abund, bool use_quals, bool final_consensus, bool vectorized_alignment,
                        int homo_gap, bool multithread, bool verbose, int SSE, bool gapless, bool greedy) {
  const int n = seqs.size();
  if (quals.nrow() == 0)   // R/dada.R:337 always passes qualities; the reference would dereference raw->qual at error.cpp:158
    Rcpp::stop("dada2hip: a quality matrix is required (R/dada.R:337 always passes one).");
  std::vector<const char *> sp(n);
  std::vector<uint8_t> pr(n);
  for (int i = 0; i < n; i++) { sp[i] = seqs[i].c_str(); pr[i] = priors[i]; }
  dada2hip_opts o = { kdist_cutoff, omegaA, omegaP, omegaC, min_fold,
                      match, mismatch, gap, homo_gap, band_size, max_clust, min_hamming, min_abund,
                      use_kmers, detect_singletons, use_quals, final_consensus, vectorized_alignment,
                      multithread, verbose, SSE, gapless, greedy };
  dada2hip_hooks hooks = { hip_log, hip_abort, NULL };
  dada2hip_result *r = NULL;
  char msg[1024];
  // NumericMatrix storage is already column-major: err is 16 x Q, quals is maxlen x nraw (Rmain.cpp:69,113)
  int rc = dada2hip_dada_uniques(n, sp.data(), abundances.data(), pr.data(), &err[0], err.ncol(),
                                 &quals[0], quals.nrow(), &o, /*device*/0, &hooks,
                                 &r, msg, sizeof msg);
  if (rc) Rcpp::stop(msg);                                   // same messages as Rmain.cpp:55-78 etc.
  const int C = dada2hip_result_nclust(r), nb = dada2hip_result_nbirth_subs(r), L = dada2hip_result_maxlen(r);
  CharacterVector sq(C), ref(nb), sub(nb);
  for (int i = 0; i < C; i++) sq[i] = dada2hip_result_sequence(r, i);
  for (int j = 0; j < nb; j++) { ref[j] = std::string(1, dada2hip_result_bs_ref(r)[j]);
                                 sub[j] = std::string(1, dada2hip_result_bs_sub(r)[j]); }
  auto iv = [&](const int32_t *p, int k) { return IntegerVector(p, p + k); };
  auto dv = [&](const double *p, int k) { return NumericVector(p, p + k); };
  DataFrame clustering = DataFrame::create(
      _["sequence"] = sq, _["abundance"] = iv(dada2hip_result_abundance(r), C), _["n0"] = iv(dada2hip_result_n0(r), C),
      _["n1"] = iv(dada2hip_result_n1(r), C), _["nunq"] = iv(dada2hip_result_nunq(r), C),
      _["pval"] = dv(dada2hip_result_clust_pval(r), C), _["birth_from"] = iv(dada2hip_result_birth_from(r), C),
      _["birth_pval"] = dv(dada2hip_result_birth_pval(r), C), _["birth_fold"] = dv(dada2hip_result_birth_fold(r), C),
      _["birth_ham"] = iv(dada2hip_result_birth_ham(r), C), _["birth_qave"] = dv(dada2hip_result_birth_qave(r), C));
  DataFrame birth_subs = DataFrame::create(_["pos"] = iv(dada2hip_result_bs_pos(r), nb), _["ref"] = ref, _["sub"] = sub,
      _["qual"] = dv(dada2hip_result_bs_qual(r), nb), _["clust"] = iv(dada2hip_result_bs_clust(r), nb));
  IntegerMatrix subqual(16, dada2hip_result_ncol(r));
  std::copy(dada2hip_result_subqual(r), dada2hip_result_subqual(r) + 16 * subqual.ncol(), subqual.begin());
  NumericMatrix cq(L, C);
  std::copy(dada2hip_result_clusterquals(r), dada2hip_result_clusterquals(r) + (size_t)L * C, cq.begin());
  List out = List::create(_["clustering"] = clustering, _["birth_subs"] = birth_subs, _["subqual"] = subqual,
                          _["clusterquals"] = cq, _["map"] = iv(dada2hip_result_map(r), n),
                          _["pval"] = dv(dada2hip_result_pval(r), n));
  dada2hip_result_free(r);
  return out;   // NA_INTEGER == DADA2HIP_NA_INTEGER and NA_REAL bit patterns are already R's
}

// The reference's own body of C_nwalign stays in the package under a _cpu name for the one input class the device kernels do
// not represent: strings with letters outside ACGT (the reference maps N and '-' through nt2int).  The homopolymer-gap and
// global aligners (src/nwalign_endsfree.cpp:220-396, :403-537) run in the library since round 3.
Rcpp::CharacterVector C_nwalign_cpu(std::string s1, std::string s2, int match, int mismatch, int gap_p, int homo_gap_p,
                                    int band, bool endsfree);   // = the reference's body of evaluate.cpp:18-62, renamed

// [[Rcpp::export]]
Rcpp::CharacterVector C_nwalign(std::string s1, std::string s2, int match, int mismatch, int gap_p, int homo_gap_p,
                                int band, bool endsfree) {
  // all three aligners of evaluate.cpp:18-62 run on the device (ends-free, homopolymer gaps, global); only strings with
  // letters outside ACGT (N / IUPAC codes) stay on the reference's CPU function
  if (s1.find_first_not_of("ACGT") != std::string::npos || s2.find_first_not_of("ACGT") != std::string::npos)
    return C_nwalign_cpu(s1, s2, match, mismatch, gap_p, homo_gap_p, band, endsfree);
  std::vector<char> a(s1.size() + s2.size() + 2), b(a.size());
  char msg[512];
  if (dada2hip_nwalign(s1.c_str(), s2.c_str(), match, mismatch, gap_p, homo_gap_p, band, endsfree, 0,
                       a.data(), b.data(), msg, sizeof msg)) Rcpp::stop(msg);
  return CharacterVector::create(std::string(a.data()), std::string(b.data()));
}

Rcpp::CharacterVector C_nwvec_cpu(std::vector<std::string> s1, std::vector<std::string> s2, int16_t match, int16_t mismatch,
                                  int16_t gap_p, int band, bool endsfree);   // = the reference's nwalign_vectorized.cpp:321-343

// [[Rcpp::export]]
Rcpp::CharacterVector C_nwvec(std::vector<std::string> s1, std::vector<std::string> s2, int16_t match, int16_t mismatch,
                              int16_t gap_p, int band, bool endsfree) {
  if (s1.size() != s2.size()) Rcpp::stop("Character vectors to be aligned must be of equal length.");   // :324
  const int n = s1.size();
  std::vector<const char *> p1(n), p2(n);
  std::vector<std::vector<char>> buf(2 * n);
  std::vector<char *> out(2 * n);
  for (int i = 0; i < n; i++) {
    p1[i] = s1[i].c_str(); p2[i] = s2[i].c_str();
    buf[2 * i].resize(s1[i].size() + s2[i].size() + 1); buf[2 * i + 1].resize(buf[2 * i].size());
    out[2 * i] = buf[2 * i].data(); out[2 * i + 1] = buf[2 * i + 1].data();
  }
  char msg[512];
  const int rc = dada2hip_nwvec(n, p1.data(), p2.data(), match, mismatch, gap_p, band, endsfree, 0, out.data(), msg, sizeof msg);   // endsfree = 0: end_gap = gap_p (:333)
  if (rc == DADA2HIP_ERR_UNSUPPORTED) return C_nwvec_cpu(s1, s2, match, mismatch, gap_p, band, endsfree);   // (a pair with more than 16 distinct letters: not DNA; N / IUPAC run on the device)
  if (rc) Rcpp::stop(msg);
  Rcpp::CharacterVector rval(2 * n);                       // rval[2i], rval[2i+1] as nwalign_vectorized.cpp:336-339
  for (int i = 0; i < 2 * n; i++) rval[i] = std::string(out[i]);
  return rval;
}

// chimera.cpp:192 — isBimeraDenovoTable's worker (R/chimeras.R:236): the table stays an R integer matrix (column-major)
// [[Rcpp::export]]
Rcpp::DataFrame C_table_bimera2(Rcpp::IntegerMatrix mat, std::vector<std::string> seqs, double min_fold, int min_abund,
                                bool allow_one_off, int min_one_off_par_dist, int match, int mismatch, int gap_p, int max_shift) {
  const int ncol = mat.ncol();
  std::vector<const char *> sp(ncol);
  for (int j = 0; j < ncol; j++) sp[j] = seqs[j].c_str();
  Rcpp::IntegerVector nflag(ncol), nsam(ncol);
  char msg[512];
  if (dada2hip_table_bimera2(mat.nrow(), ncol, &mat[0], sp.data(), min_fold, min_abund, allow_one_off, min_one_off_par_dist,
                             match, mismatch, gap_p, max_shift, 0, &nflag[0], &nsam[0], msg, sizeof msg)) Rcpp::stop(msg);
  return Rcpp::DataFrame::create(_["nflag"] = nflag, _["nsam"] = nsam);
}
// C_is_bimera (chimera.cpp:18): same pattern over dada2hip_is_bimera(sq, npars, pars[], ..., &flag).
