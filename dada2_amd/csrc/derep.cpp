// derep.cpp — dereplication front-end of the hot path: FASTQ(.gz) -> uniques, abundances,
// per-unique mean qualities, read -> unique map, in the layout dada2hip_dada_uniques takes.  Host-side C++ (the
// reference does this in R on top of ShortRead: derepFastq / qtables2, /root/reference/R/sequenceIO.R:45-124,
// :150-183); no GPU work here, the result can be handed to dada2hip_sample_from_derep without another copy.
//
// Semantics kept from the reference:
//   * reads are taken in chunks of `n` records (derepFastq's n = 1e6, :57); inside a chunk the uniques are in C-locale
//     lexical order (srsort, :161) and zero-length reads are ignored with their map entry NA (:154-158,:173-177);
//     uniques first seen in a later chunk are appended after the earlier ones (:85-88);
//   * quality sums are accumulated per unique and position and divided by the abundance at the end (:95); positions past
//     a unique's length are NA;
//   * the final order is a STABLE sort by decreasing abundance (R's order(), :98), ties keep the order above.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include <zlib.h>

#include "../../include/dada2hip.h"
#include "hostpar.h"

struct dada2hip_derep {
  std::vector<std::string> seqs;
  std::vector<const char *> seq_ptrs;
  std::vector<int32_t> abund, map;
  std::unique_ptr<double[]> quals;   // [nuniques][maxlen], NA past each unique's length (filled by the host pool: first touch in parallel)
  int32_t maxlen = 0;
  int64_t nreads = 0;
};

namespace {

double na_real() {
  union { double d; uint64_t u; } v;
  v.u = 0x7FF00000000007A2ULL;  // R's NA_real_
  return v.d;
}

// Line reader: a background thread inflates the file (gzread; plain files pass through zlib untouched) into 8 MiB
// pieces while the caller parses the previous ones; lines are found with memchr.
struct LineReader {
  static constexpr size_t PIECE = 8u << 20;
  gzFile f;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::vector<char>> ready;
  bool done = false, stop = false;
  std::string zerr;               // non-empty: the stream ended on a zlib error (corrupt / truncated .gz), not at EOF
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  explicit LineReader(gzFile f_) : f(f_) {
    th = std::thread([this] {
      for (;;) {
        std::vector<char> piece(PIECE);
        const int got = gzread(f, piece.data(), (unsigned)PIECE);
        std::unique_lock<std::mutex> lk(mu);
        if (got <= 0) {
          // gzread returns 0 at a clean end of file AND at a stream cut short; -1 on a data / CRC error: ask gzerror
          int zrc = Z_OK;
          const char *zm = gzerror(f, &zrc);
          if (got < 0 || (zrc != Z_OK && zrc != Z_STREAM_END)) zerr = zm && *zm ? zm : "read error";
          done = true; cv.notify_all(); return;
        }
        piece.resize((size_t)got);
        ready.push_back(std::move(piece));
        cv.notify_all();
        cv.wait(lk, [this] { return ready.size() < 4 || stop; });
        if (stop) return;
      }
    });
  }
  ~LineReader() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    th.join();
  }
  bool refill() {   // append the next piece behind the unread tail
    std::vector<char> piece;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return !ready.empty() || done; });
      if (ready.empty()) return false;
      piece = std::move(ready.front());
      ready.pop_front();
    }
    cv.notify_all();
    if (pos == end) { buf.swap(piece); pos = 0; end = buf.size(); return true; }
    std::vector<char> nb(end - pos + piece.size());
    memcpy(nb.data(), &buf[pos], end - pos);
    memcpy(nb.data() + (end - pos), piece.data(), piece.size());
    buf.swap(nb); pos = 0; end = buf.size();
    return true;
  }
  std::string error() { std::lock_guard<std::mutex> lk(mu); return zerr; }
  // next line without its terminator as [*p, *p + *n); valid until the next call; false at EOF
  bool next(const char **p, size_t *n) {
    for (;;) {
      char *nl = end > pos ? (char *)memchr(&buf[pos], '\n', end - pos) : nullptr;
      if (nl) {
        size_t len = (size_t)(nl - &buf[pos]);
        *p = &buf[pos];
        pos += len + 1;
        if (len && (*p)[len - 1] == '\r') len--;
        *n = len;
        return true;
      }
      if (eof) {
        if (end == pos) return false;
        size_t len = end - pos;
        *p = &buf[pos];
        pos = end;
        while (len && ((*p)[len - 1] == '\r')) len--;
        *n = len;
        return true;
      }
      if (!refill()) eof = true;
    }
  }
};

// quality storage of the uniques seen ONCE (nine in ten of a deep amplicon sample): the read's quality characters as they are,
// a byte per position - the 8-byte sums below are only set up when a unique is met a second time (a 250-nt singleton then costs
// 250 bytes of memory traffic instead of 2 000)
struct ByteArena {
  static constexpr size_t BLOCK = 8u << 20;
  std::vector<std::unique_ptr<unsigned char[]>> blocks;
  size_t used = BLOCK;
  unsigned char *take(size_t n) {
    if (n > BLOCK) { blocks.emplace_back(new unsigned char[n]); used = BLOCK; return blocks.back().get(); }
    if (used + n > BLOCK) { blocks.emplace_back(new unsigned char[BLOCK]); used = 0; }
    unsigned char *r = blocks.back().get() + used;
    used += n;
    return r;
  }
};
// quality-sum storage: fixed blocks, so growing never copies what is already there
struct SumArena {
  static constexpr size_t BLOCK = 4u << 20;
  std::vector<std::unique_ptr<int64_t[]>> blocks;
  size_t used = BLOCK;
  int64_t *take(size_t n) {
    if (n > BLOCK) { blocks.emplace_back(new int64_t[n]()); used = BLOCK; return blocks.back().get(); }   // (never for reads)
    if (used + n > BLOCK) {
      blocks.emplace_back(new int64_t[BLOCK]);
      int64_t *b = blocks.back().get();
      d2::parallel_for(BLOCK, BLOCK / 16, [b](size_t i0, size_t i1) { memset(b + i0, 0, (i1 - i0) * sizeof(int64_t)); });   // first touch in parallel
      used = 0;
    }
    int64_t *r = blocks.back().get() + used;
    used += n;
    return r;
  }
};

// std::sort over the host pool: sorted runs in parallel, then pairwise merges level by level (each level's merges in parallel).
// The same order as one std::sort for a strict weak order without equal elements (the uniques of a chunk are distinct strings).
template <typename It, typename Cmp>
void pool_sort(It first, It last, Cmp cmp) {
  const size_t n = (size_t)(last - first);
  size_t runs = 1;
  while (runs < (size_t)d2::HostPool::get().nthreads() && n / (runs * 2) >= (size_t)1 << 15) runs *= 2;
  if (runs == 1) { std::sort(first, last, cmp); return; }
  auto bound = [&](size_t r, size_t of) { return first + (ptrdiff_t)(n * r / of); };
  d2::parallel_for(runs, 1, [&](size_t r0, size_t r1) { for (size_t r = r0; r < r1; r++) std::sort(bound(r, runs), bound(r + 1, runs), cmp); });
  for (size_t width = 1; width < runs; width *= 2)
    d2::parallel_for(runs / (2 * width), 1, [&](size_t m0, size_t m1) {
      for (size_t m = m0; m < m1; m++)
        std::inplace_merge(bound(2 * m * width, runs), bound((2 * m + 1) * width, runs), bound((2 * m + 2) * width, runs), cmp);
    });
}

void set_err(char *errbuf, size_t errlen, const char *m) {
  if (errbuf && errlen) snprintf(errbuf, errlen, "%s", m);
}

}  // namespace

extern "C" {

static int derep_fastq_body(const char *path, int64_t chunk_reads, int32_t qual_offset, dada2hip_derep **out, char *errbuf,
                            size_t errlen) {
  if (out) *out = nullptr;
  if (!path || !out) { set_err(errbuf, errlen, "File paths must be provided in character format."); return DADA2HIP_ERR_INPUT; }
  gzFile f = gzopen(path, "rb");
  if (!f) { set_err(errbuf, errlen, "Not all provided files exist."); return DADA2HIP_ERR_INPUT; }
  gzbuffer(f, 1 << 20);
  struct GzGuard { gzFile f; ~GzGuard() { gzclose(f); } } guard{f};   // closed after the reader thread has joined
  if (chunk_reads <= 0) chunk_reads = 1000000;   // derepFastq(n = 1e6)
  // Uniques get a provisional id at first sight; at every chunk end the ids born in that chunk are put in lexical order
  // and appended to `seen` (the order derepFastq's merge produces).  Quality characters are summed raw, the encoding
  // offset is taken off once at the end (exact in integers), so nothing of a chunk has to be buffered.
  std::deque<std::string> store;                        // provisional id -> sequence (stable addresses)
  std::unordered_map<std::string_view, int32_t> index;  // sequence -> provisional id
  std::vector<int64_t> count;                           // per provisional id
  std::vector<int64_t *> qacc;                          // per provisional id: raw quality character sums [len]; null while the unique has one read
  std::vector<const unsigned char *> qone;              // per provisional id: the quality characters of its first read
  SumArena arena;
  ByteArena barena;
  std::vector<int32_t> seen;                            // provisional ids in derepFastq's pre-sort order
  std::vector<int32_t> map;                             // per read: provisional id, -1 for zero-length reads
  d2::knobs_reload();
  const bool times = d2::knobs().derep_times;
  using dclk = std::chrono::steady_clock;
  auto t_start = dclk::now();
  double ms_sort = 0;
  auto ms_since = [](dclk::time_point t) { return std::chrono::duration<double, std::milli>(dclk::now() - t).count(); };
  LineReader in(f);
  const char *hp, *sp, *pp, *qp;
  size_t hn, sn, pn, qn;
  index.reserve(1 << 16);
  int offset = qual_offset, minq = 255;
  int rc = DADA2HIP_OK;
  int64_t nreads = 0, in_chunk = 0;
  size_t chunk_first = 0;   // first provisional id born in the current chunk
  auto end_chunk = [&]() {
    if (offset <= 0 && minq < 255) offset = minq < 59 ? 33 : 64;   // qualityType "Auto": below ';' only Phred+33 encodings
    const size_t n0 = seen.size();
    const auto t_sort = dclk::now();
    for (size_t id = chunk_first; id < store.size(); id++) seen.push_back((int32_t)id);
    pool_sort(seen.begin() + n0, seen.end(), [&](int32_t a, int32_t b) { return store[a] < store[b]; });   // srsort: C locale (10^6 string compares x 20: the largest single piece of a chunk's work, so it goes over the host pool)
    chunk_first = store.size();
    in_chunk = 0;
    ms_sort += ms_since(t_sort);
  };
  std::string s;
  while (in.next(&hp, &hn)) {
    if (hn == 0) continue;
    bool ok = hp[0] == '@' && in.next(&sp, &sn);
    if (ok) s.assign(sp, sn);   // the window may move under the next two lines
    ok = ok && in.next(&pp, &pn) && pn > 0 && pp[0] == '+' && in.next(&qp, &qn) && qn == s.size();
    if (!ok) {
      const std::string ze = in.error();   // a record cut short by a damaged stream is a read error, not a format error
      if (!ze.empty()) set_err(errbuf, errlen, ("dada2hip: error reading " + std::string(path) + ": " + ze).c_str());
      else set_err(errbuf, errlen, "dada2hip: malformed FASTQ record");
      rc = DADA2HIP_ERR_INPUT;
      break;
    }
    nreads++;
    in_chunk++;
    if (s.empty()) {
      map.push_back(-1);
    } else {
      auto it = index.find(std::string_view(s));
      int32_t id;
      if (it == index.end()) {
        id = (int32_t)store.size();
        store.push_back(s);
        index.emplace(std::string_view(store.back()), id);
        count.push_back(0);
        qacc.push_back(nullptr);
        qone.push_back(nullptr);
      } else {
        id = it->second;
      }
      const unsigned char *qq = (const unsigned char *)qp;
      const size_t n = s.size();
      if (offset <= 0 && chunk_first == 0) for (size_t p = 0; p < n; p++) minq = std::min(minq, (int)qq[p]);
      if (++count[id] == 1) {                           // first sight: keep the characters
        unsigned char *b = barena.take(n);
        memcpy(b, qq, n);
        qone[id] = b;
      } else {
        int64_t *acc = qacc[id];
        if (!acc) {                                     // second sight: the sums start from the first read's characters
          acc = qacc[id] = arena.take(n);
          const unsigned char *b = qone[id];
          for (size_t p = 0; p < n; p++) acc[p] = b[p];
        }
        for (size_t p = 0; p < n; p++) acc[p] += qq[p];
      }
      map.push_back(id);
    }
    if (in_chunk >= chunk_reads) end_chunk();
  }
  if (rc != DADA2HIP_OK) return rc;
  {
    const std::string ze = in.error();
    if (!ze.empty()) {   // never hand back a silently truncated object
      set_err(errbuf, errlen, ("dada2hip: error reading " + std::string(path) + ": " + ze).c_str());
      return DADA2HIP_ERR_INPUT;
    }
  }
  end_chunk();
  const double ms_parse = ms_since(t_start) - ms_sort;
  const auto t_out = dclk::now();
  if (store.empty()) { set_err(errbuf, errlen, "Only zero-length sequences detected during dereplication."); return DADA2HIP_ERR_INPUT; }
  if (offset <= 0) offset = 33;
  // stable sort by decreasing abundance (sequenceIO.R:98)
  const size_t U = seen.size();
  std::vector<int32_t> ord(seen);
  std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return count[a] > count[b]; });
  std::vector<int32_t> rank(U);
  for (size_t k = 0; k < U; k++) rank[ord[k]] = (int32_t)k;
  dada2hip_derep *d = new dada2hip_derep();
  d->nreads = nreads;
  for (auto &u : store) d->maxlen = std::max<int32_t>(d->maxlen, (int32_t)u.size());
  d->seqs.resize(U); d->abund.resize(U);
  d->quals.reset(new double[U * (size_t)d->maxlen]);
  const double na = na_real();
  const size_t ml = (size_t)d->maxlen;
  d2::parallel_for(U, 256, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; k++) {
      const int32_t id = ord[k];
      d->abund[k] = (int32_t)count[id];
      double *row = &d->quals[k * ml];
      const int64_t *acc = qacc[id];
      const size_t len = store[id].size();
      if (acc) for (size_t p = 0; p < len; p++) row[p] = (double)(acc[p] - (int64_t)offset * count[id]) / (double)count[id];   // derepQuals / derepCounts (:95)
      else { const unsigned char *b = qone[id]; for (size_t p = 0; p < len; p++) row[p] = (double)((int64_t)b[p] - (int64_t)offset) / 1.0; }
      for (size_t p = len; p < ml; p++) row[p] = na;
    }
  });
  index.clear();
  for (size_t k = 0; k < U; k++) d->seqs[k] = std::move(store[ord[k]]);
  d->seq_ptrs.resize(U);
  for (size_t k = 0; k < U; k++) d->seq_ptrs[k] = d->seqs[k].c_str();
  d->map.resize(map.size());
  for (size_t i = 0; i < map.size(); i++) d->map[i] = map[i] < 0 ? DADA2HIP_NA_INTEGER : rank[map[i]];
  *out = d;
  if (times) fprintf(stderr, "[derep] %lld reads, %zu uniques: read + parse + hash %.0f ms, chunk sorts %.0f ms, output (abundance order, mean qualities, map) %.0f ms\n",
                     (long long)nreads, U, ms_parse, ms_sort, ms_since(t_out));
  return DADA2HIP_OK;
}

// no exception crosses the C ABI (include/dada2hip.h): allocation failures and thread errors become error codes
int dada2hip_derep_fastq(const char *path, int64_t chunk_reads, int32_t qual_offset, dada2hip_derep **out, char *errbuf,
                         size_t errlen) {
  try {
    return derep_fastq_body(path, chunk_reads, qual_offset, out, errbuf, errlen);
  } catch (const std::bad_alloc &) {
    set_err(errbuf, errlen, "dada2hip: out of host memory during dereplication");
  } catch (const std::exception &e) {
    set_err(errbuf, errlen, (std::string("dada2hip: ") + e.what()).c_str());
  } catch (...) {
    set_err(errbuf, errlen, "dada2hip: unknown error during dereplication");
  }
  if (out) *out = nullptr;
  return DADA2HIP_ERR_RUNTIME;
}

int32_t dada2hip_derep_nuniques(const dada2hip_derep *d) { return d ? (int32_t)d->seqs.size() : 0; }
int64_t dada2hip_derep_nreads(const dada2hip_derep *d) { return d ? d->nreads : 0; }
int32_t dada2hip_derep_maxlen(const dada2hip_derep *d) { return d ? d->maxlen : 0; }
const char *const *dada2hip_derep_seqs(const dada2hip_derep *d) { return d ? d->seq_ptrs.data() : nullptr; }
const int32_t *dada2hip_derep_abundances(const dada2hip_derep *d) { return d ? d->abund.data() : nullptr; }
const double *dada2hip_derep_quals(const dada2hip_derep *d) { return d ? d->quals.get() : nullptr; }
const int32_t *dada2hip_derep_map(const dada2hip_derep *d) { return d ? d->map.data() : nullptr; }
void dada2hip_derep_free(dada2hip_derep *d) { delete d; }

int dada2hip_sample_from_derep(const dada2hip_derep *d, const uint8_t *priors, int32_t device, dada2hip_sample **out, char *errbuf,
                               size_t errlen) {
  if (!d) { set_err(errbuf, errlen, "dada2hip: no derep object"); return DADA2HIP_ERR_INPUT; }
  return dada2hip_sample_create((int32_t)d->seqs.size(), d->seq_ptrs.data(), d->abund.data(), priors, d->quals.get(), d->maxlen, device,
                                out, errbuf, errlen);
}

}  // extern "C"
