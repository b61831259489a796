// oracle/shim/RcppParallel.h — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for <RcppParallel.h> (TBB is not installed here).  The reference
// uses exactly two things (src/cluster.cpp:90,176; src/Rmain.cpp:179,222):
// RcppParallel::Worker and RcppParallel::parallelFor(begin, end, worker, grain).
// Per-index results are independent in both call sites, so chunking across
// pooled std::thread workers gives output identical to TBB's; the thread count
// is a process global set through dada2_shim_set_threads() (1 = run inline).
#ifndef DADA2_ORACLE_SHIM_RCPPPARALLEL_H
#define DADA2_ORACLE_SHIM_RCPPPARALLEL_H

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <thread>
#include <vector>

extern "C" int dada2_shim_nthreads;

#include <Rcpp.h>

namespace RcppParallel {

// read / write views of R objects handed to workers (chimera.cpp:66-71): thin wrappers over the shim's containers
template <typename T> class RMatrix {
  std::shared_ptr<std::vector<T>> keep;   // the view keeps the object alive, as a protected SEXP would be
  const T *p;
  std::size_t nr, nc;
public:
  RMatrix(const Rcpp::Mat<T> &m) : keep(m.p), p(m.p->data()), nr((std::size_t)m.nr), nc((std::size_t)m.nc) {}
  const T *begin() const { return p; }
  std::size_t nrow() const { return nr; }
  std::size_t ncol() const { return nc; }
};
template <typename T> class RVector {
  std::shared_ptr<std::vector<T>> keep;
  T *p;
  std::size_t n;
public:
  RVector(Rcpp::Vec<T> x) : keep(x.p), p(x.p->data()), n(x.p->size()) {}   // (shares the caller's storage, as an R vector does)
  T &operator[](std::size_t i) { return p[i]; }
  const T &operator[](std::size_t i) const { return p[i]; }
  std::size_t size() const { return n; }
};

struct Worker {
  virtual ~Worker() {}
  virtual void operator()(std::size_t begin, std::size_t end) = 0;
};

// The workers are a PERSISTENT pool (oracle/ref_capi.cpp: created by dada2_shim_set_threads, parked on a condition
// variable between calls), as TBB's are behind the real RcppParallel: a run_dada of ~700 rounds issues ~700 parallelFor
// calls, and spawning + joining 255 threads for each of them was most of the "all cores" wall time of the first shim.
extern "C" void dada2_shim_parallel_for(std::size_t begin, std::size_t end, std::size_t chunk,
                                        void (*fn)(void *, std::size_t, std::size_t), void *ctx);

inline void parallelFor(std::size_t begin, std::size_t end, Worker &worker, std::size_t grainSize = 1) {
  std::size_t n = end > begin ? end - begin : 0;
  int nt = dada2_shim_nthreads;
  if (nt <= 1 || n <= grainSize) {
    if (n) worker(begin, end);
    return;
  }
  // dynamic chunks of a few grains each, handed out through an atomic cursor
  std::size_t chunk = std::max<std::size_t>(grainSize, std::min<std::size_t>(256, n / (std::size_t)(8 * nt) + 1));
  dada2_shim_parallel_for(begin, end, chunk, [](void *w, std::size_t b, std::size_t e) { (*static_cast<Worker *>(w))(b, e); }, &worker);
}

}  // namespace RcppParallel

#endif
