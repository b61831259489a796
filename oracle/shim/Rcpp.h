// oracle/shim/Rcpp.h — TEST INFRASTRUCTURE ONLY.
//
// Minimal stand-in for <Rcpp.h> so that the reference's own hot-path sources
// (/root/reference/src/{Rmain,cluster,containers,kmers,misc,pval,error,
// nwalign_endsfree,nwalign_vectorized,chimera,evaluate}.cpp) compile UNMODIFIED, in place, into
// oracle/_ref/libdada2ref.so (recipe: oracle/Makefile).  Nothing here is a copy
// of Rcpp: it is a from-scratch value-semantics model of exactly the Rcpp
// surface those nine files touch (SURVEY.md §8c lists it).  R and Rcpp are not
// installed in this image, which is why this exists.
//
// The only arithmetic supplied here rather than by the reference is
// Rcpp::ppois -> dada2_oracle_ppois (oracle/rmath_ppois.c).
#ifndef DADA2_ORACLE_SHIM_RCPP_H
#define DADA2_ORACLE_SHIM_RCPP_H

#include <cstdio>
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <climits>
#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" double dada2_oracle_ppois(double x, double lambda, int lower_tail);

#define NA_INTEGER INT_MIN
static inline double shim_na_real() {
  // R's NA_real_: quiet NaN with low word 1954.
  union { double d; uint64_t u; } v;
  v.u = 0x7FF00000000007A2ULL;
  return v.d;
}
#define NA_REAL (shim_na_real())

extern "C" int dada2_shim_verbose;  // 0 = swallow Rprintf (default)
static inline void Rprintf(const char *fmt, ...) {
  if (!dada2_shim_verbose) return;
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
}

namespace Rcpp {

inline void stop(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw std::runtime_error(buf);
}
inline void stop(const std::string &s) { throw std::runtime_error(s); }
inline void checkUserInterrupt() {}

// R vectors are reference objects: copying an Rcpp vector copies the handle, not the data (chimera.cpp:80 hands
// `flags` / `sams` to its worker BY VALUE and expects the worker's writes to land in the caller's vectors)
template <typename T> class Vec {
public:
  std::shared_ptr<std::vector<T>> p;
  Vec() : p(std::make_shared<std::vector<T>>()) {}
  explicit Vec(size_t n) : p(std::make_shared<std::vector<T>>(n, T())) {}
  template <typename It> Vec(It first, It last) : p(std::make_shared<std::vector<T>>(first, last)) {}   // (tests/glue: Rcpp's range constructor)
  typename std::vector<T>::iterator begin() { return p->begin(); }
  typename std::vector<T>::iterator end() { return p->end(); }
  std::vector<T> &v() { return *p; }
  const std::vector<T> &v() const { return *p; }
  size_t size() const { return p->size(); }
  T &operator[](size_t i) { return (*p)[i]; }
  const T &operator[](size_t i) const { return (*p)[i]; }
  T &operator()(size_t i) { return (*p)[i]; }
  const T &operator()(size_t i) const { return (*p)[i]; }
  void push_back(const T &x) { p->push_back(x); }
};

struct NilType {};                       // R_NilValue: evaluate.cpp:77,128 return it in place of a vector
struct NamedInt { std::string name; int val; };
class IntegerVector : public Vec<int> {
public:
  bool is_null = false;
  IntegerVector() {}
  explicit IntegerVector(size_t n) : Vec<int>(n) {}
  IntegerVector(size_t n, int fill) : Vec<int>(n) { for (auto &x : v()) x = fill; }   // chimera.cpp:195-196
  IntegerVector(NilType) : is_null(true) {}
  IntegerVector(const int *first, const int *last) : Vec<int>(first, last) {}
  template <typename... A> static IntegerVector create(const A &...a) {                // evaluate.cpp:112 (named integer(3))
    IntegerVector r;
    NamedInt arr[] = {NamedInt{a.name, a.val->iv.at(0)}...};
    for (auto &x : arr) r.push_back(x.val);
    return r;
  }
};
typedef IntegerVector LogicalVector;     // evaluate.cpp:184
class NumericVector : public Vec<double> {
public:
  NumericVector() {}
  explicit NumericVector(size_t n) : Vec<double>(n) {}
  NumericVector(const double *first, const double *last) : Vec<double>(first, last) {}
  static double get_na() { return NA_REAL; }
};
class CharacterVector : public Vec<std::string> {
public:
  bool is_null = false;
  CharacterVector() {}
  explicit CharacterVector(size_t n) : Vec<std::string>(n) {}
  CharacterVector(const std::string &x) { push_back(x); }   // evaluate.cpp:172 returns a std::string as character(1)
  CharacterVector(NilType) : is_null(true) {}
  static CharacterVector create(const std::string &a, const std::string &b) { CharacterVector r; r.push_back(a); r.push_back(b); return r; }
};

// (matrices are reference objects too: chimera.cpp:76 keeps a view of a by-value IntegerMatrix parameter after it is gone)
template <typename T> class Mat {
public:
  std::shared_ptr<std::vector<T>> p;
  int nr, nc;
  Mat() : p(std::make_shared<std::vector<T>>()), nr(0), nc(0) {}
  Mat(int nrow, int ncol) : p(std::make_shared<std::vector<T>>((size_t)nrow * (size_t)ncol, T())), nr(nrow), nc(ncol) {}
  std::vector<T> &v() { return *p; }
  const std::vector<T> &v() const { return *p; }
  int nrow() const { return nr; }
  int ncol() const { return nc; }
  T &operator[](size_t i) { return (*p)[i]; }                              // (column-major storage, as R's)
  const T &operator[](size_t i) const { return (*p)[i]; }
  typename std::vector<T>::iterator begin() { return p->begin(); }
  T &operator()(size_t r, size_t c) { return (*p)[c * (size_t)nr + r]; }  // column-major, as R
  const T &operator()(size_t r, size_t c) const { return (*p)[c * (size_t)nr + r]; }
};
typedef Mat<double> NumericMatrix;
typedef Mat<int> IntegerMatrix;

template <typename T> inline T as(const NumericVector &x) { return (T)x[0]; }

inline NumericVector ppois(const IntegerVector &x, double lambda, bool lower) {
  NumericVector r(x.size());
  for (size_t i = 0; i < x.size(); i++) r[i] = dada2_oracle_ppois((double)x[i], lambda, lower ? 1 : 0);
  return r;
}

// ---- named-list model for List::create / DataFrame::create ------------------
struct RObj;
typedef std::shared_ptr<RObj> RObjP;
struct RObj {
  enum Kind { INT, DBL, STR, IMAT, DMAT, LIST } kind;
  std::vector<int> iv;
  std::vector<double> dv;
  std::vector<std::string> sv;
  int nr = 0, nc = 0;
  std::vector<std::pair<std::string, RObjP>> items;
  const RObj *get(const std::string &name) const {
    for (auto &it : items)
      if (it.first == name) return it.second.get();
    return nullptr;
  }
};

class List {
public:
  RObjP obj;
  List() : obj(std::make_shared<RObj>()) { obj->kind = RObj::LIST; }
  template <typename... A> static List create(const A &...a);
};
typedef List DataFrame;

inline RObjP wrap(const IntegerVector &x) { auto o = std::make_shared<RObj>(); o->kind = RObj::INT; o->iv = x.v(); return o; }
inline RObjP wrap(const NumericVector &x) { auto o = std::make_shared<RObj>(); o->kind = RObj::DBL; o->dv = x.v(); return o; }
inline RObjP wrap(const CharacterVector &x) { auto o = std::make_shared<RObj>(); o->kind = RObj::STR; o->sv = x.v(); return o; }
inline RObjP wrap(const std::vector<std::string> &x) { auto o = std::make_shared<RObj>(); o->kind = RObj::STR; o->sv = x; return o; }
inline RObjP wrap(const IntegerMatrix &x) { auto o = std::make_shared<RObj>(); o->kind = RObj::IMAT; o->iv = x.v(); o->nr = x.nr; o->nc = x.nc; return o; }
inline RObjP wrap(const NumericMatrix &x) { auto o = std::make_shared<RObj>(); o->kind = RObj::DMAT; o->dv = x.v(); o->nr = x.nr; o->nc = x.nc; return o; }
inline RObjP wrap(const List &x) { return x.obj; }
inline RObjP wrap(int x) { auto o = std::make_shared<RObj>(); o->kind = RObj::INT; o->iv = {x}; return o; }

struct Named {
  std::string name;
  RObjP val;
};
struct NameProxy {
  std::string name;
  template <typename T> Named operator=(const T &x) const { return Named{name, wrap(x)}; }
};
struct Placeholder {
  NameProxy operator[](const char *n) const { return NameProxy{n}; }
};
static const Placeholder _ = Placeholder();

template <typename... A> List List::create(const A &...a) {
  List l;
  Named arr[] = {a...};
  for (auto &n : arr) l.obj->items.push_back({n.name, n.val});
  return l;
}

}  // namespace Rcpp

static const Rcpp::NilType R_NilValue = Rcpp::NilType();

#endif
