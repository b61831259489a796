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
#include <vector>

#include <dlfcn.h>
#include <zlib.h>

#include "../../include/dada2hip.h"
#include "hostpar.h"

struct dada2hip_derep {
  std::unique_ptr<char[]> seq_blob;  // the uniques' sequences, NUL-terminated, in output order
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

// byte buffers that are not zero-filled when they are sized (they are about to be overwritten by a read or an inflate: the
// fill was a second pass over every byte of the file)
template <typename T>
struct NoInit {
  using value_type = T;
  NoInit() = default;
  template <typename U> NoInit(const NoInit<U> &) {}
  T *allocate(size_t n) { return static_cast<T *>(::operator new(n * sizeof(T))); }
  void deallocate(T *p, size_t) { ::operator delete(p); }
  template <typename U, typename... A> void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
  }
  template <typename U> bool operator==(const NoInit<U> &) const { return true; }
  template <typename U> bool operator!=(const NoInit<U> &) const { return false; }
};
using Bytes = std::vector<char, NoInit<char>>;

// Line reader: a background thread inflates the file (gzread; a plain file is read with fread - through zlib it is one more
// copy of every byte) into 8 MiB pieces while the caller parses the previous ones; lines are found with memchr.  A buffer the
// window has moved out of is RETIRED, not freed: the lines handed out stay valid until the caller says release() (the
// records of a batch are hashed in parallel before they are looked at in order).
struct LineReader {
  static constexpr size_t PIECE = 8u << 20;
  gzFile f;
  FILE *plain = nullptr;          // non-null: the file is not compressed (gzdirect) and is read directly
  std::vector<Bytes> retired;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Bytes> ready;
  bool done = false, stop = false;
  std::string zerr;               // non-empty: the stream ended on a zlib error (corrupt / truncated .gz), not at EOF
  Bytes buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  // (the whole text is in memory already: nothing to read)
  explicit LineReader(Bytes &&whole) : f(nullptr), buf(std::move(whole)) { end = buf.size(); done = true; }
  LineReader(gzFile f_, FILE *plain_) : f(f_), plain(plain_) {
    th = std::thread([this] {
      for (;;) {
        Bytes piece(PIECE);
        const int got = plain ? (int)fread(piece.data(), 1, PIECE, plain) : gzread(f, piece.data(), (unsigned)PIECE);
        std::unique_lock<std::mutex> lk(mu);
        if (got <= 0) {
          if (plain) { if (ferror(plain)) zerr = "read error"; done = true; cv.notify_all(); return; }
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
    if (th.joinable()) th.join();
  }
  bool refill() {   // append the next piece behind the unread tail
    Bytes piece;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return !ready.empty() || done; });
      if (ready.empty()) return false;
      piece = std::move(ready.front());
      ready.pop_front();
    }
    cv.notify_all();
    if (pos == end) { buf.swap(piece); pos = 0; end = buf.size(); if (!piece.empty()) retired.push_back(std::move(piece)); return true; }
    Bytes nb(end - pos + piece.size());
    memcpy(nb.data(), &buf[pos], end - pos);
    memcpy(nb.data() + (end - pos), piece.data(), piece.size());
    buf.swap(nb); pos = 0; end = buf.size();
    retired.push_back(std::move(nb));
    return true;
  }
  void release() { retired.clear(); }   // the lines handed out before the last refill are no longer needed
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

// A .gz file in one go through libdeflate, where the system has it (dlopen: the library is not a build dependency; zlib's
// streaming inflate - one thread, 150-300 MB/s of text - is 95 % of a dereplication's time on a compressed file, libdeflate's
// one-shot decoder is 2.4x faster on FASTQ).  Returns 0: not available / not applicable (the caller streams through zlib),
// 1: `text` holds the file's decompressed members, -1: the stream is damaged (`why`).  Concatenated members are decoded one
// after the other, as gzread does.
int inflate_whole(const char *path, Bytes &text, std::string &why) {
  struct Api {
    void *(*alloc)() = nullptr;
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    bool ok = false;
    Api() {
      void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
      if (!h) return;
      alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
      gunzip = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
      release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
      ok = alloc && gunzip && release;
    }
  };
  static const Api api;
  if (!api.ok) return 0;
  FILE *fp = fopen(path, "rb");
  if (!fp) return 0;
  struct Closer { FILE *p; ~Closer() { fclose(p); } } closer{fp};
  if (fseek(fp, 0, SEEK_END) != 0) return 0;
  const long fsz = ftell(fp);
  if (fsz < 18 || fsz > (1l << 30)) return 0;           // (larger files: streamed, nothing of them held twice)
  rewind(fp);
  std::vector<unsigned char, NoInit<unsigned char>> gz((size_t)fsz);
  if (fread(gz.data(), 1, (size_t)fsz, fp) != (size_t)fsz) return 0;
  if (gz[0] != 0x1F || gz[1] != 0x8B) return 0;
  void *dec = api.alloc();
  if (!dec) return 0;
  struct Rel { const Api &a; void *d; ~Rel() { a.release(d); } } rel{api, dec};
  uint32_t isize;                                        // the last member's length mod 2^32: the first guess of the text's size
  memcpy(&isize, &gz[(size_t)fsz - 4], 4);
  text.resize(std::max<size_t>((size_t)isize + 64, (size_t)fsz * 3));
  size_t in_pos = 0, out_pos = 0;
  while (in_pos + 18 <= (size_t)fsz && gz[in_pos] == 0x1F && gz[in_pos + 1] == 0x8B) {
    size_t used = 0, made = 0;
    const int r = api.gunzip(dec, &gz[in_pos], (size_t)fsz - in_pos, text.data() + out_pos, text.size() - out_pos, &used, &made);
    if (r == 3) {                                              // LIBDEFLATE_INSUFFICIENT_SPACE: the member again, into more room -
      // up to a bound: a file that inflates beyond 64 x its size or 8 GiB is left to the streaming reader, which runs
      // in constant memory (a hostile or very compressible file must not end in bad_alloc or the OOM killer here)
      const size_t cap = std::min<size_t>((size_t)8 << 30, std::max<size_t>((size_t)fsz * 64, (size_t)1 << 26));
      if (text.size() >= cap) { text.clear(); text.shrink_to_fit(); return 0; }
      text.resize(std::min(cap, text.size() * 2));
      continue;
    }
    if (r != 0) { why = "damaged gzip stream (invalid or truncated deflate data, or a CRC / length mismatch)"; return -1; }
    in_pos += used; out_pos += made;
  }
  if (in_pos == 0) { why = "damaged gzip stream"; return -1; }
  // a trailing piece that starts like a member but is too short to be one: gzread reports a truncated file, so does this
  if (in_pos + 2 <= (size_t)fsz && gz[in_pos] == 0x1F && gz[in_pos + 1] == 0x8B) { why = "damaged gzip stream (truncated trailing member)"; return -1; }
  text.resize(out_pos);
  return 1;
}

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

// Sort of DISTINCT elements over the host pool (sample sort): splitters from an evenly spaced sample, every element to its
// bucket (parallel), the buckets sorted side by side.  Any correct sort of distinct elements gives the one order, so this is
// std::sort's result.  (The first form - sorted runs, then pairwise merges level by level - spent most of its time in the last
// levels, where one or two threads merge everything: 0.64 s of a 2.0 s dereplication of 1.2 M uniques.)
template <typename Cmp>
void pool_sort(int32_t *first, size_t n, Cmp cmp) {
  const size_t nt = (size_t)d2::HostPool::get().nthreads();
  if (nt <= 1 || n < ((size_t)1 << 16)) { std::sort(first, first + n, cmp); return; }
  size_t B = 16;
  while (B < 4 * nt && B < 1024 && n / (2 * B) >= 2048) B *= 2;
  const size_t OVER = 32, ns = B * OVER;
  std::vector<int32_t> sample(ns);
  for (size_t i = 0; i < ns; i++) sample[i] = first[(size_t)((double)i * (double)n / (double)ns)];
  std::sort(sample.begin(), sample.end(), cmp);
  std::vector<int32_t> split(B - 1);
  for (size_t k = 1; k < B; k++) split[k - 1] = sample[k * OVER];
  const size_t nchunk = std::min<size_t>(4 * nt, (n + 4095) / 4096);
  auto cbound = [&](size_t c) { return n * c / nchunk; };
  std::vector<uint16_t> bucket(n);
  std::vector<size_t> cnt(nchunk * B, 0);
  d2::parallel_for(nchunk, 1, [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; c++) {
      size_t *my = &cnt[c * B];
      for (size_t i = cbound(c); i < cbound(c + 1); i++) {
        const size_t k = (size_t)(std::upper_bound(split.begin(), split.end(), first[i], cmp) - split.begin());
        bucket[i] = (uint16_t)k;
        my[k]++;
      }
    }
  });
  std::vector<size_t> start(B + 1, 0);
  for (size_t k = 0; k < B; k++) {                       // bucket-major, chunk-minor offsets
    size_t run = start[k];
    for (size_t c = 0; c < nchunk; c++) { const size_t v = cnt[c * B + k]; cnt[c * B + k] = run; run += v; }
    start[k + 1] = run;
  }
  std::vector<int32_t> tmp(n);
  d2::parallel_for(nchunk, 1, [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; c++) {
      size_t *my = &cnt[c * B];
      for (size_t i = cbound(c); i < cbound(c + 1); i++) tmp[my[bucket[i]]++] = first[i];
    }
  });
  d2::parallel_for(B, 1, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; k++) {
      std::sort(tmp.begin() + (ptrdiff_t)start[k], tmp.begin() + (ptrdiff_t)start[k + 1], cmp);
      std::copy(tmp.begin() + (ptrdiff_t)start[k], tmp.begin() + (ptrdiff_t)start[k + 1], first + start[k]);
    }
  });
}

// 64-bit hash of a sequence (multiply-fold over 8-byte words; the table below compares the bytes on a tag match, so quality
// only matters for speed)
inline uint64_t fold64(uint64_t a, uint64_t b) { const __uint128_t r = (__uint128_t)a * b; return (uint64_t)r ^ (uint64_t)(r >> 64); }
inline uint64_t seq_hash(const char *p, size_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n * 0xD6E8FEB86659FD93ull;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = fold64(h ^ w, 0xA0761D6478BD642Full); }
  if (i < n) { uint64_t w = 0; memcpy(&w, p + i, n - i); h = fold64(h ^ w, 0xE7037ED1A0B428DBull); }
  return fold64(h, 0x8EBC6AF09C88C6E3ull);
}

// sequence -> provisional id: open addressing over (32-bit tag, id) pairs, the hash computed by the caller (in parallel, for a
// whole batch of records) and kept per id for the re-insertion when the table grows
struct SeqTable {
  std::vector<uint32_t> tag;
  std::vector<int32_t> id;
  size_t mask = 0, used = 0;
  SeqTable() { resize(1 << 17); }
  void resize(size_t cap) { tag.assign(cap, 0); id.assign(cap, -1); mask = cap - 1; used = 0; }
  static uint32_t tag_of(uint64_t h) { return (uint32_t)(h >> 32); }
  void put(uint64_t h, int32_t v) {                      // (the key is known to be absent)
    size_t s = (size_t)h & mask;
    while (id[s] >= 0) s = (s + 1) & mask;
    id[s] = v; tag[s] = tag_of(h); used++;
  }
  template <typename Same>
  int32_t find(uint64_t h, Same same) const {
    const uint32_t t = tag_of(h);
    for (size_t s = (size_t)h & mask;; s = (s + 1) & mask) {
      const int32_t v = id[s];
      if (v < 0) return -1;
      if (tag[s] == t && same(v)) return v;
    }
  }
  void insert(uint64_t h, int32_t v, const std::vector<uint64_t> &hashes) {
    if ((used + 1) * 2 > mask + 1) {
      resize((mask + 1) * 4);
      for (size_t k = 0; k < hashes.size(); k++) if ((int32_t)k != v) put(hashes[k], (int32_t)k);
    }
    put(h, v);
  }
};

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
  FILE *plain = gzdirect(f) ? fopen(path, "rb") : nullptr;           // not compressed: read it without the copy through zlib
  struct GzGuard { gzFile f; FILE *p; ~GzGuard() { gzclose(f); if (p) fclose(p); } } guard{f, plain};   // closed after the reader thread has joined
  if (chunk_reads <= 0) chunk_reads = 1000000;   // derepFastq(n = 1e6)
  // Uniques get a provisional id at first sight; at every chunk end the ids born in that chunk are put in lexical order
  // and appended to `seen` (the order derepFastq's merge produces).  Quality characters are summed raw, the encoding
  // offset is taken off once at the end (exact in integers), so nothing of a chunk has to be buffered.
  //
  // The records are taken in BATCHES: the reader's lines stay valid until the batch is done (LineReader::release), the
  // sequences of a batch are hashed over the host pool, and only the table look-ups - which assign the provisional ids in
  // read order - run one after the other.
  struct Rec { const char *seq, *qual; uint32_t len; int32_t minq; uint64_t h; };
  ByteArena seq_arena;                                  // provisional id -> sequence bytes (stable addresses)
  std::vector<const char *> sptr;                       // per provisional id
  std::vector<uint32_t> slen;
  std::vector<uint64_t> shash;
  SeqTable index;                                       // sequence -> provisional id
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
  Bytes whole;
  int have_whole = 0;
  if (!plain && !d2::knobs().derep_zlib) {
    std::string why;
    have_whole = inflate_whole(path, whole, why);
    if (have_whole < 0) { set_err(errbuf, errlen, ("dada2hip: error reading " + std::string(path) + ": " + why).c_str()); return DADA2HIP_ERR_INPUT; }
  }
  std::unique_ptr<LineReader> in_holder(have_whole > 0 ? new LineReader(std::move(whole)) : new LineReader(f, plain));
  LineReader &in = *in_holder;
  const char *hp, *sp, *pp, *qp;
  size_t hn, sn, pn, qn;
  int offset = qual_offset, minq = 255;
  int rc = DADA2HIP_OK;
  int64_t nreads = 0, in_chunk = 0;
  size_t chunk_first = 0;   // first provisional id born in the current chunk
  auto seq_less = [&](int32_t a, int32_t b) {            // srsort: C locale (bytes as unsigned, a prefix first)
    const size_t la = slen[a], lb = slen[b];
    const int c = memcmp(sptr[a], sptr[b], std::min(la, lb));
    return c < 0 || (c == 0 && la < lb);
  };
  auto end_chunk = [&]() {
    if (offset <= 0 && minq < 255) offset = minq < 59 ? 33 : 64;   // qualityType "Auto": below ';' only Phred+33 encodings
    const size_t n0 = seen.size();
    const auto t_sort = dclk::now();
    for (size_t id = chunk_first; id < sptr.size(); id++) seen.push_back((int32_t)id);
    pool_sort(seen.data() + n0, seen.size() - n0, seq_less);   // (10^6 string compares x 20: the largest single piece of a chunk's work)
    chunk_first = sptr.size();
    in_chunk = 0;
    ms_sort += ms_since(t_sort);
  };
  std::vector<Rec> batch;
  const size_t BATCH = 32768;
  batch.reserve(BATCH);
  auto flush = [&]() {
    const bool want_min = offset <= 0;
    d2::parallel_for(batch.size(), 512, [&](size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; i++) {
        Rec &r = batch[i];
        r.h = seq_hash(r.seq, r.len);
        int m = 255;
        if (want_min) for (uint32_t p = 0; p < r.len; p++) m = std::min(m, (int)(unsigned char)r.qual[p]);
        r.minq = m;
      }
    });
    for (const Rec &r : batch) {
      nreads++;
      in_chunk++;
      if (r.len == 0) {
        map.push_back(-1);
      } else {
        const size_t n = r.len;
        int32_t id = index.find(r.h, [&](int32_t v) { return slen[v] == r.len && memcmp(sptr[v], r.seq, n) == 0; });
        if (id < 0) {
          id = (int32_t)sptr.size();
          char *b = (char *)seq_arena.take(n);
          memcpy(b, r.seq, n);
          sptr.push_back(b); slen.push_back(r.len); shash.push_back(r.h);
          index.insert(r.h, id, shash);
          count.push_back(0);
          qacc.push_back(nullptr);
          qone.push_back(nullptr);
        }
        const unsigned char *qq = (const unsigned char *)r.qual;
        if (offset <= 0 && chunk_first == 0) minq = std::min(minq, (int)r.minq);
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
    batch.clear();
    in.release();
  };
  while (in.next(&hp, &hn)) {
    if (hn == 0) continue;
    const bool ok = hp[0] == '@' && in.next(&sp, &sn) && in.next(&pp, &pn) && pn > 0 && pp[0] == '+' && in.next(&qp, &qn) && qn == sn;
    if (!ok) {
      const std::string ze = in.error();   // a record cut short by a damaged stream is a read error, not a format error
      if (!ze.empty()) set_err(errbuf, errlen, ("dada2hip: error reading " + std::string(path) + ": " + ze).c_str());
      else set_err(errbuf, errlen, "dada2hip: malformed FASTQ record");
      rc = DADA2HIP_ERR_INPUT;
      break;
    }
    batch.push_back(Rec{sp, qp, (uint32_t)sn, 255, 0});
    if (batch.size() >= BATCH) flush();
  }
  if (rc != DADA2HIP_OK) return rc;
  {
    const std::string ze = in.error();
    if (!ze.empty()) {   // never hand back a silently truncated object
      set_err(errbuf, errlen, ("dada2hip: error reading " + std::string(path) + ": " + ze).c_str());
      return DADA2HIP_ERR_INPUT;
    }
  }
  flush();
  end_chunk();
  const double ms_parse = ms_since(t_start) - ms_sort;
  const auto t_out = dclk::now();
  if (sptr.empty()) { set_err(errbuf, errlen, "Only zero-length sequences detected during dereplication."); return DADA2HIP_ERR_INPUT; }
  if (offset <= 0) offset = 33;
  // stable sort by decreasing abundance (sequenceIO.R:98): (abundance descending, position in `seen` ascending) is a strict
  // order without equal elements, so the pool's sort gives the stable result
  const size_t U = seen.size();
  std::vector<int32_t> pos_of(U);
  d2::parallel_for(U, 4096, [&](size_t k0, size_t k1) { for (size_t k = k0; k < k1; k++) pos_of[seen[k]] = (int32_t)k; });
  std::vector<int32_t> ord(seen);
  pool_sort(ord.data(), U, [&](int32_t a, int32_t b) { return count[a] > count[b] || (count[a] == count[b] && pos_of[a] < pos_of[b]); });
  std::vector<int32_t> rank(U);
  d2::parallel_for(U, 4096, [&](size_t k0, size_t k1) { for (size_t k = k0; k < k1; k++) rank[ord[k]] = (int32_t)k; });
  dada2hip_derep *d = new dada2hip_derep();
  std::unique_ptr<dada2hip_derep> dguard(d);
  d->nreads = nreads;
  for (size_t k = 0; k < U; k++) d->maxlen = std::max<int32_t>(d->maxlen, (int32_t)slen[k]);
  d->abund.resize(U);
  d->quals.reset(new double[U * (size_t)d->maxlen]);
  std::vector<size_t> soff(U + 1, 0);
  for (size_t k = 0; k < U; k++) soff[k + 1] = soff[k] + slen[ord[k]] + 1;
  d->seq_blob.reset(new char[soff[U]]);
  d->seq_ptrs.resize(U);
  const double na = na_real();
  const size_t ml = (size_t)d->maxlen;
  d2::parallel_for(U, 256, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; k++) {
      const int32_t id = ord[k];
      d->abund[k] = (int32_t)count[id];
      double *row = &d->quals[k * ml];
      const int64_t *acc = qacc[id];
      const size_t len = slen[id];
      if (acc) for (size_t p = 0; p < len; p++) row[p] = (double)(acc[p] - (int64_t)offset * count[id]) / (double)count[id];   // derepQuals / derepCounts (:95)
      else { const unsigned char *b = qone[id]; for (size_t p = 0; p < len; p++) row[p] = (double)((int64_t)b[p] - (int64_t)offset) / 1.0; }
      for (size_t p = len; p < ml; p++) row[p] = na;
      char *dst = d->seq_blob.get() + soff[k];
      memcpy(dst, sptr[id], len);
      dst[len] = 0;
      d->seq_ptrs[k] = dst;
    }
  });
  d->map.resize(map.size());
  d2::parallel_for(map.size(), 65536, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) d->map[i] = map[i] < 0 ? DADA2HIP_NA_INTEGER : rank[map[i]];
  });
  *out = dguard.release();
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

int32_t dada2hip_derep_nuniques(const dada2hip_derep *d) { return d ? (int32_t)d->seq_ptrs.size() : 0; }
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
  return dada2hip_sample_create((int32_t)d->seq_ptrs.size(), d->seq_ptrs.data(), d->abund.data(), priors, d->quals.get(), d->maxlen, device,
                                out, errbuf, errlen);
}

}  // extern "C"
