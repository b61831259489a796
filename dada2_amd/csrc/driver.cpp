// driver.cpp — host side of libdada2hip.so: sample residency, the divisive loop of run_dada
// (/root/reference/src/Rmain.cpp:297-336) driven over the HIP kernels, the output tables of
// src/error.cpp, and the extern "C" boundary declared in include/dada2hip.h.
//
// Division of labour in this revision: every O(nraw x length) step runs on the
// device — k-mer screen, gapless pairing, banded NW + traceback + lambda, final alignments,
// transition/quality tables.  The partition bookkeeping that consumes one (lambda, hamming) pair
// per unique per round (cluster.cpp:179-201 store filter, b_shuffle2, b_bud, b_p_update) is
// ordered, pointer-chasing integer/fp64-compare work and runs on the host from the dense device
// output.  There is no CPU implementation of any kernel: without a GPU every entry point fails.
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <memory>
#include <thread>
#include <unordered_map>

#include "engine.h"
#include "hostpar.h"
#include "knobs.h"
#include "ppois.h"

namespace d2 {

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

static double na_real() {
  union { double d; uint64_t u; } v;
  v.u = 0x7FF00000000007A2ULL;  // R's NA_real_
  return v.d;
}

struct InputError { std::string msg; };
struct PeerFailed {};   // sharded run: another rank reported a failure at an exchange point
struct RuntimeErr { int code; std::string msg; };
struct TailEntryFailed {};   // the persistent round tail could not become co-resident (nothing changed): the launch chains take over

// device / pinned buffers; the memory comes from (and returns to) the per-process allocation cache (hostpar.h)
template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) AllocCache::get().dev_release(p); }
  void alloc(size_t count) {
    if (count <= n && p) return;
    if (p) { AllocCache::get().dev_release(p); p = nullptr; }
    n = count;
    // 64 bytes of slack behind the last element: the streaming kernels read 16 bytes per lane and their last lane may start
    // inside the array and end behind it (what it reads there is never used)
    D2_HIP(AllocCache::get().dev_alloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T) + 64));
  }
  void zero(hipStream_t st) { D2_HIP(hipMemsetAsync(p, 0, n * sizeof(T), st)); }
  void free() { if (p) AllocCache::get().dev_release(p); p = nullptr; n = 0; }
};

template <typename T> struct PinBuf {
  T *p = nullptr;
  size_t n = 0;
  PinBuf() {}
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() { if (p) AllocCache::get().pin_release(p); }
  void alloc(size_t count) {
    if (count <= n && p) return;
    if (p) { AllocCache::get().pin_release(p); p = nullptr; }
    n = count;
    D2_HIP(AllocCache::get().pin_alloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)));
  }
};

}  // namespace d2

using namespace d2;

// =================================================================================================
struct dada2hip_sample {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;    // late fetches (long mover lists) that must not queue behind the next round's kernels
  hipStream_t cmp = nullptr;     // the NEXT batch's compare while the persistent tail runs the rounds of this one (created on first use)
  SampleDev D;
  DevBuf<uint32_t> seq2, heavy, reads;
  DevBuf<uint8_t> qual, nheavy, prior;
  DevBuf<uint16_t> kord, kmult;
  DevBuf<uint32_t> kbits;
  DevBuf<int32_t> len, nwflag;
  PinBuf<uint32_t> h_seq2;          // host mirror of the packed sequences (pinned: it is the upload source)
  std::vector<int32_t> h_len;
  std::vector<uint32_t> h_reads;
  std::vector<uint8_t> h_prior;
  uint64_t total_reads = 0;
  int qmax = 0;
  double ms_upload = 0;
  // per-run device work buffers (kept across runs: selfConsist passes reuse them)
  DevBuf<uint8_t> d_skip, d_cls, d_correct, d_moves;
  DevBuf<double> d_lambda, d_err;
  DevBuf<uint32_t> d_ham, scr_ptr, scr_t, d_qn, d_ctab, scr_adw, scr_ad;
  DevBuf<uint16_t> scr_foff;          // k_nw_ad -> k_ad_product: factor offsets of the launch's alignments (engine.h, SampleDev::ad_foff)
  DevBuf<AdDesc> scr_fdesc;
  int scr_adw_waves = 0, scr_adw_band = -999;
  size_t scr_adw_wpw = 0;
  DevBuf<int32_t> d_nw_list, d_gl_list, d_counters, d_thresh, scr_rows, d_work, d_chunk_centre, d_cluster_of,
      d_centre_of_cluster, d_trans, d_nsubs, d_nmoves;
  DevBuf<uint16_t> d_view, d_view_b;
  DevBuf<unsigned long long> d_qsum;
  PinBuf<int32_t> h_counters;
  NwScratch scr;
  int scr_class = -1, scr_band = 0;
  size_t scr_items = 0;          // alignments per launch the lane-kernel scratch was sized for
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::shared_ptr<void> run_cache;   // Run object: partition-state buffers reused by every dada2hip_sample_run
};

struct dada2hip_result {
  int nclust = 0, nraw = 0, maxlen = 0, ncol = 0;
  std::vector<std::string> sequence;
  std::vector<int32_t> abundance, n0, n1, nunq, birth_from, birth_ham, center, bs_pos, bs_clust, subqual, map;
  std::vector<double> clust_pval, birth_pval, birth_fold, birth_qave, bs_qual, clusterquals, pval;
  std::vector<char> bs_ref, bs_sub;
  dada2hip_stats stats;
};

namespace {

// (library load: the one environment default the library sets, knobs.h)
__attribute__((constructor)) static void d2_process_defaults() { d2::knobs_process_defaults(); }

// The persistent round tail (k3_tail) needs ALL its blocks co-resident, one per CU (a block takes its CU whole: 1024 threads at 128
// registers, 156 KB of LDS).  Launches that together want more CUs than the device has would each get part of them and wait for
// the rest until their entry barriers time out, so a device has persistent SLOTS per process: a run takes one for its rounds
// (run_v3), waiting its turn if there is none, and several runs hold slots at the same time only while their blocks together stay
// within three quarters of the CUs - the rest is for everybody's compare kernels.  Round 6: up to three (DADA2HIP_V3_SLOTS; until
// then ONE, the samples' rounds taking turns): configs[3] on one GPU, 8 x 250 k uniques = 61 blocks each, 246 -> 164 ms with
// three samples in flight (profiles/r10j_*).  Round 5 had measured side-by-side tails 2.4x SLOWER; what changed since: the tail's
// blocks own their CUs (then: 512 threads beside the compares on every CU), one launch in flight per run instead of two, and
// eight hardware queues (GPU_MAX_HW_QUEUES, below) instead of four for the six to nine streams of three samples.
struct SlotSem {
  std::mutex mu;
  std::condition_variable cv;
  int used = 0, blocks = 0;
  void acquire(int cap, int my_blocks, int ncu) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&]() { return used == 0 || (used < cap && blocks + my_blocks <= ncu - ncu / 4); });
    used++; blocks += my_blocks;
  }
  void release(int my_blocks) {
    { std::lock_guard<std::mutex> lk(mu); used--; blocks -= my_blocks; }
    cv.notify_all();
  }
};
SlotSem &persistent_slot(int device) {
  static SlotSem slots[64];
  return slots[device & 63];
}
// ... and across processes that share a GPU (several ranks on one device): an advisory lock on a file named after the
// device's PCI address, taken without waiting - a process that does not get it runs its rounds on the launch chains.
struct PersistFile {
  int fd = -1;
  bool held = false;
  int holders = 0;               // runs of THIS process that hold it (DADA2HIP_V3_SLOTS > 1: more than one)
  std::mutex mu;
  bool try_acquire(int device) {
    std::lock_guard<std::mutex> g(mu);
    if (held) { holders++; return true; }
    if (fd < 0) {
      char bus[64] = "unknown";
      if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "dev%d", device); }
      for (char *c = bus; *c; c++) if (*c == ':' || *c == '/') *c = '_';
      char path[160];
      snprintf(path, sizeof path, "/tmp/dada2hip_persistent_%s.lock", bus);
      // O_NOFOLLOW: never through a symlink somebody planted in /tmp.  A file another user created under umask 022 cannot be
      // opened for writing: flock works on a read-only descriptor too.  No descriptor at all = no way to know whether another
      // process holds the device's slot: the run then takes the launch chains, it does not assume it is alone.
      fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
      if (fd < 0) fd = open(path, O_RDONLY | O_CLOEXEC | O_NOFOLLOW);
      if (fd < 0) return false;
    }
    held = flock(fd, LOCK_EX | LOCK_NB) == 0;
    if (held) holders = 1;
    return held;
  }
  void release() {
    std::lock_guard<std::mutex> g(mu);
    if (!held || --holders > 0) return;
    if (fd >= 0) (void)flock(fd, LOCK_UN);
    held = false;
  }
};
// Boundary calls of THIS process that are running on a device right now (dada2hip_run_multi / several threads of the caller with
// a sample each): the persistent tail plans prefetch compares only for a run that has the device to itself - with a second
// sample in flight its kernels already fill the tail's gaps, and a round spinning for a prefetch that queues behind them holds
// the device's persistent slot for nothing (configs[3] on one GPU, two in flight: 236 ms without, 456 ms with; profiles/r07i)
std::atomic<int> &active_runs(int device) {
  static std::atomic<int> n[64];
  return n[device & 63];
}
PersistFile &persistent_file(int device) {
  static PersistFile files[64];
  return files[device & 63];
}

// Wait for the stream by polling: the per-round decision points (shuffle movers, bud result) sit on
// the critical path, and a polled wait returns microseconds sooner than a blocking one.
// DADA2HIP_WAIT=block (or more resident samples / ranks than host cores): sleep between polls instead of spinning.
static bool wait_blocks() { return knobs().wait_block; }
static double wait_timeout_s() { return knobs().wait_timeout_s; }
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}
void sync_spin(hipStream_t st) {
  if (wait_blocks()) { D2_HIP(hipStreamSynchronize(st)); return; }
  hipError_t e;
  const auto t0 = clk::now();
  for (unsigned spins = 0; (e = hipStreamQuery(st)) == hipErrorNotReady; spins++) {
    cpu_relax();
    if ((spins & 0x3FFF) == 0x3FFF && ms_since(t0) > wait_timeout_s() * 1e3)
      throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: timed out waiting for the device (stream wait)"};
  }
  if (e != hipSuccess)
    throw d2::DeviceError{DADA2HIP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(e) + " (stream wait)"};
}

void set_err(char *errbuf, size_t errlen, const std::string &m) {
  if (errbuf && errlen) snprintf(errbuf, errlen, "%s", m.c_str());
}

template <typename F> int guarded(char *errbuf, size_t errlen, F &&f) {
  try {
    knobs_reload();   // the environment is read HERE, once per boundary call (knobs.h)
    f();
    return DADA2HIP_OK;
  } catch (const InputError &e) {
    set_err(errbuf, errlen, e.msg);
    return DADA2HIP_ERR_INPUT;
  } catch (const DeviceError &e) {
    set_err(errbuf, errlen, e.msg);
    return e.code;
  } catch (const RuntimeErr &e) {
    set_err(errbuf, errlen, e.msg);
    return e.code;
  } catch (const std::exception &e) {
    set_err(errbuf, errlen, e.what());
    return DADA2HIP_ERR_RUNTIME;
  }
}

void select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: no HIP device available (this library has no CPU fallback)"};
  if (device < 0 || device >= n) throw DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: invalid device ordinal"};
  D2_HIP(hipSetDevice(device));
}

// value read by out-of-band neighbours (nwalign_endsfree.cpp:113-119 / nwalign_vectorized.cpp:106)
int nw_sentinel(const dada2hip_opts &o) {
  if (!o.vectorized_alignment) return -9999;
  int m = 0;
  m = std::min(m, (int)o.mismatch); m = std::min(m, (int)o.gap); m = std::min(m, (int)o.match);
  return -32768 - m;
}

// ---- sample creation: validate (Rmain.cpp:52-78), pack, upload, build k-mer records -------------
// The reference copies its inputs out of the R objects serially (Rmain.cpp:102-120).  At 10^6 uniques x 250 nt the
// boundary hands over 250 MB of characters and 2 GB of doubles, so the marshalling is spread over the host pool:
//   pass 1  strlen / abundance / prior per unique                                  (threads)
//   pass 2  2-bit packing + ACGT validation straight into a pinned buffer          (threads)  -> one async H2D
//   pass 3  (uint8) round(mean quality) of raw_new (containers.cpp:34), chunk-wise into pinned bytes (threads) -> H2D per
//           chunk on the side stream, overlapping the next chunk's conversion and the k-mer build on the main stream
// lite = nwalign / nwvec helper samples: no qualities, no k-mer records, any length >= 1.
void sample_create(dada2hip_sample *s, int32_t nraw, const char *const *seqs, const int32_t *abund,
                   const uint8_t *priors, const double *quals, int32_t quals_nrow, int device, bool lite = false) {
  auto t0 = clk::now();
  if (nraw <= 0) throw InputError{"Zero input sequences."};
  if (!seqs || !abund) throw InputError{"Sequence and abundance vectors had different lengths."};
  s->h_len.resize(nraw);
  s->h_reads.resize(nraw);
  s->h_prior.assign(nraw, 0);
  std::atomic<int> amax{0}, amin{SEQLEN};
  std::atomic<uint64_t> atot{0};
  parallel_for((size_t)nraw, 8192, [&](size_t lo, size_t hi) {
    int mx = 0, mn = SEQLEN;
    uint64_t tot = 0;
    for (size_t i = lo; i < hi; i++) {
      const int l = (int)strnlen(seqs[i], (size_t)SEQLEN + 1);
      s->h_len[i] = l;
      mx = std::max(mx, l); mn = std::min(mn, l);
      s->h_reads[i] = (uint32_t)abund[i];
      tot += (uint32_t)abund[i];
      if (priors) s->h_prior[i] = priors[i] ? 1 : 0;
    }
    int cur = amax.load();
    while (mx > cur && !amax.compare_exchange_weak(cur, mx)) {}
    cur = amin.load();
    while (mn < cur && !amin.compare_exchange_weak(cur, mn)) {}
    atot += tot;
  });
  const int maxlen = amax.load(), minlen = amin.load();
  s->total_reads = atot.load();
  if (maxlen >= SEQLEN) throw InputError{"Input sequences exceed the maximum allowed string length."};
  if (!lite) {
    if (minlen <= KMER_SIZE) throw InputError{"Input sequences must all be longer than the kmer-size (5)."};
    if (!quals)
      throw InputError{"dada2hip: a quality matrix is required (the reference reads it unconditionally, src/error.cpp:158)."};
    if (quals_nrow != maxlen) throw InputError{"Sequence must have associated qualities for each nucleotide position."};
  } else if (minlen < 1) throw InputError{"dada2hip: empty sequence."};

  select_device(device);
  s->device = device;
  D2_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  D2_HIP(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
  D2_HIP(hipEventCreate(&s->ev0));
  D2_HIP(hipEventCreate(&s->ev1));
  SampleDev &D = s->D;
  D.N = nraw; D.maxlen = maxlen; D.minlen = minlen;
  D.r_lo = 0; D.r_hi = nraw;
  D.W2 = (((maxlen + 15) / 16) + 3) & ~3;
  D.LQ = (maxlen + 15) & ~15;
  D.LK = (std::max(maxlen - KMER_SIZE + 1, 1) + 7) & ~7;
  {   // DADA2HIP_KORD_ALIGN=1: rows of k-mer records padded to 128-B cache lines (the screen's two 256-B reads per row then
      // touch 2 lines each instead of 3).  Measured 2 % on the screen at 1e6 uniques for 3 % more memory: off by default.
    const bool aligned = knobs().kord_align;
    if (aligned) D.LK = (D.LK + 63) & ~63;
  }
  D.HMAX = std::max(0, (maxlen - KMER_SIZE + 1) / (RANK_SAT + 1));

  // 2-bit packing on the host pool, straight into pinned memory (validates ACGT: R checks C_isACGT before the call,
  // R/dada.R:269); the pinned rows stay as the host mirror of the sequences (result strings are decoded from them)
  s->h_seq2.alloc((size_t)nraw * D.W2);
  std::atomic<int> bad{0};
  const int W2 = D.W2;
  uint32_t *packed = s->h_seq2.p;
  parallel_for((size_t)nraw, 2048, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      const char *q = seqs[i];
      uint32_t *row = packed + i * W2;
      const int l = s->h_len[i];
      int p = 0;
      for (int w = 0; w < W2; w++) {
        uint32_t word = 0;
        const int e = std::min(l - p, 16);
        for (int k = 0; k < e; k++, p++) {
          uint32_t c;
          switch (q[p]) {
            case 'A': c = 0; break;
            case 'C': c = 1; break;
            case 'G': c = 2; break;
            case 'T': c = 3; break;
            default: c = 0; bad.store(1, std::memory_order_relaxed);
          }
          word |= c << (k << 1);
        }
        row[w] = word;
      }
    }
  });
  if (bad.load()) throw InputError{"Invalid derep$uniques vector. Sequences must be made up only of A/C/G/T."};
  s->seq2.alloc((size_t)nraw * D.W2);
  s->nwflag.alloc(1);
  D.nw_flag = s->nwflag.p;
  {   // pointer ring of k_nw_ad (kernels.hip): 8 192 wave slots of 8 KB at 250 nt; fewer, never below 1 024, for long reads.
      // Only its geometry here: the 64-256 MB are taken for the duration of a run (ensure_ad_ring) - the throw-away
      // samples of nwvec / merge / bimera never need them, and hundreds of resident samples would each have held one
    D.ad_wpw = 64 * ((2 * maxlen + 1 + 15) / 16);
    int waves = 8192;
    while (waves > 1024 && (size_t)waves * D.ad_wpw * 4 > ((size_t)256 << 20)) waves /= 2;
    D.ad_ptr = nullptr;
    D.ad_waves = waves;
  }
  D2_HIP(hipMemsetAsync(D.nw_flag, 0, 4, s->stream));
  s->len.alloc(nraw); s->reads.alloc(nraw); s->prior.alloc(nraw); s->nheavy.alloc(nraw);
  s->qual.alloc((size_t)nraw * D.LQ);
  D.seq2 = s->seq2.p; D.len = s->len.p; D.reads = s->reads.p; D.prior = s->prior.p; D.nheavy = s->nheavy.p;
  D.qual = s->qual.p;
  D2_HIP(hipMemcpyAsync(D.seq2, packed, (size_t)nraw * D.W2 * 4, hipMemcpyHostToDevice, s->stream));
  D2_HIP(hipMemcpyAsync(D.len, s->h_len.data(), (size_t)nraw * 4, hipMemcpyHostToDevice, s->stream));
  D2_HIP(hipMemcpyAsync(D.reads, s->h_reads.data(), (size_t)nraw * 4, hipMemcpyHostToDevice, s->stream));
  D2_HIP(hipMemcpyAsync(D.prior, s->h_prior.data(), (size_t)nraw, hipMemcpyHostToDevice, s->stream));
  if (lite) {
    D2_HIP(hipMemsetAsync(D.qual, 0, (size_t)nraw * D.LQ, s->stream));
    D2_HIP(hipStreamSynchronize(s->stream));
    s->qmax = 0;
    s->ms_upload = ms_since(t0);
    return;
  }
  s->kord.alloc((size_t)nraw * D.LK);
  s->heavy.alloc((size_t)nraw * std::max(D.HMAX, 1));
  D.kord = s->kord.p; D.heavy = s->heavy.p;
  // 5-mer presence bitmaps for the batch screen's prefilter: worth their 128 B per unique while a read's k-mers leave most of
  // the 1024 bits clear (250 nt: ~215 set).  Long reads fill them (1 500 nt: ~800 set, the bound never decides) - none there
  if (maxlen - KMER_SIZE + 1 <= 512 && knobs().screen_bits != 0) {
    s->kbits.alloc((size_t)nraw * 32); s->kmult.alloc(nraw);
    D.kbits = s->kbits.p; D.kmult = s->kmult.p;
  } else { D.kbits = nullptr; D.kmult = nullptr; }
  launch_build_kmers(D, s->stream);          // overlaps the quality upload below (side stream)

  // qualities: raw_new's (uint8) round(mean quality) (containers.cpp:34) is part of the marshalling here as it is in the
  // reference: the host pool turns the R double matrix (8 B per base, 2 GB at 10^6 uniques) into the byte matrix the
  // device keeps, straight into pinned memory, and each finished chunk goes out while the next is being converted.
  // NA/NaN past a read's end are never looked at (Rmain.cpp:113 copies only the first `length` entries).
  {
    const int LQ = D.LQ;
    PinBuf<uint8_t> hq;
    hq.alloc((size_t)nraw * LQ);
    std::atomic<int> badq{0}, qmax{0};
    // Several samples in flight on this device (dada2hip_run_multi's threads share it: active_runs() is raised for the call): half
    // the pool per upload.  The uploads of the samples go through the pool one after the other and overlap the OTHER samples'
    // rounds, so their wall time is not what counts - 64 threads read the matrix at 80 GB/s for 1.6 CPU-seconds per 2 GB, 32 at
    // 60 GB/s for 1.1, and they leave the memory system and the container's CPU quota to the threads that feed the
    // device: configs[3] on one GPU, three in flight, 64 / 32 / 24 / 16 threads: 187 / 174 / 183 / 193 ms (profiles/r10u_*).
    const int marshal_threads = active_runs(device).load() > 1 ? std::max(8, HostPool::get().nthreads() / 2) : 0;
    const size_t rows_per = std::max<size_t>(1, ((size_t)16 << 20) / (size_t)LQ);
    for (size_t r0 = 0; r0 < (size_t)nraw; r0 += rows_per) {
      const size_t nr = std::min<size_t>(rows_per, nraw - r0);
      parallel_for(nr, 512, [&](size_t lo, size_t hi) {
        int mx = 0, bad = 0;
        for (size_t r = r0 + lo; r < r0 + hi; r++) {
          const double *src = quals + r * (size_t)maxlen;
          uint8_t *dst = hq.p + r * (size_t)LQ;
          const int L = s->h_len[r];
          // the row in one branch-free sweep (it vectorises: the scalar form below was 17 cycles per quality, and the 2 GB of a
          // 10^6-unique matrix made it the largest piece of the upload) - valid for values in [0, 255.5); a row with anything
          // else (negative, too large, NaN inside the read) is redone by the exact scalar rule
          int row_mx = 0;
          if (!d2::round_quality_row(src, dst, L, &row_mx)) {
            row_mx = 0;
            for (int p = 0; p < L; p++) {
              const double x = src[p];
              double xr;
              if (x >= 0.0 && x < 256.0) { const int t = (int)x; xr = (double)(t + ((x - (double)t) >= 0.5 ? 1 : 0)); }   // round(): half away from zero
              else xr = std::round(x);
              if (!(xr >= 0.0 && xr <= 255.0)) { bad = 1; xr = 0.0; }
              const int v = (int)xr;
              row_mx = std::max(row_mx, v);
              dst[p] = (uint8_t)v;
            }
          }
          mx = std::max(mx, row_mx);
          memset(dst + L, 0, (size_t)(LQ - L));
        }
        if (bad) badq.store(1, std::memory_order_relaxed);
        int cur = qmax.load();
        while (mx > cur && !qmax.compare_exchange_weak(cur, mx)) {}
      }, marshal_threads);
      D2_HIP(hipMemcpyAsync(D.qual + r0 * LQ, hq.p + r0 * LQ, nr * (size_t)LQ, hipMemcpyHostToDevice, s->side));
    }
    D2_HIP(hipStreamSynchronize(s->side));                    // (the pinned buffer goes back to the cache)
    if (badq.load()) throw InputError{"Invalid derep$quals matrix. Quality values must be positive integers."};
    s->qmax = qmax.load();
  }
  D2_HIP(hipStreamSynchronize(s->stream));
  D2_HIP(hipGetLastError());
  s->ms_upload = ms_since(t0);
}

// a traceback that left its bounds (cannot happen with valid pointers; nwalign_endsfree.cpp:185 raises the same way)
// the traceback-pointer ring of k_nw_ad for the time of one run (it comes from and goes back to the allocation cache)
void ensure_ad_ring(dada2hip_sample *s) {
  if (s->D.ad_ptr) return;
  s->scr_ad.alloc((size_t)s->D.ad_waves * s->D.ad_wpw);
  s->D.ad_ptr = s->scr_ad.p;
  // the factor-offset rows of k_ad_product: one per work slot of a launch - at most all pairs of a batch compare (8 N), at
  // most 2^20, at most 1 GB; what does not fit is multiplied up inside k_nw_ad as before
  const int stride = (s->D.maxlen + 7) & ~7;
  long long cap = std::min<long long>(1ll << 20, ((long long)1 << 30) / (2ll * stride));
  cap = std::min<long long>(cap, 8ll * s->D.N + 512);
  if (knobs().ad_fcap > 0) cap = std::max<long long>(1, std::min<long long>(cap, knobs().ad_fcap));   // test knob: the in-kernel product for what does not fit
  s->scr_foff.alloc((size_t)cap * stride);
  s->scr_fdesc.alloc((size_t)cap);
  D2_HIP(hipMemsetAsync(s->scr_fdesc.p, 0xFF, (size_t)cap * sizeof(AdDesc), s->stream));   // dest = -1: nothing to do
  s->D.ad_foff = s->scr_foff.p; s->D.ad_desc = s->scr_fdesc.p; s->D.ad_fcap = (int32_t)cap; s->D.ad_fstride = stride;
}
struct AdRingGuard {
  dada2hip_sample *s;
  ~AdRingGuard() {
    if (!s->D.ad_ptr) return;
    if (std::uncaught_exceptions() == 0) (void)hipStreamSynchronize(s->stream);   // (no launch that uses it is left in the stream)
    s->scr_ad.free(); s->scr_foff.free(); s->scr_fdesc.free();
    s->D.ad_ptr = nullptr; s->D.ad_foff = nullptr; s->D.ad_desc = nullptr; s->D.ad_fcap = 0;
  }
};

void check_nw_flag(dada2hip_sample *s) {
  int32_t f = 0;
  D2_HIP(hipMemcpy(&f, s->D.nw_flag, 4, hipMemcpyDeviceToHost));
  if (f) {
    (void)hipMemset(s->D.nw_flag, 0, 4);
    throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "N-W Align out of range."};
  }
}

// sequence of unique i as a string, decoded from the packed host mirror
std::string seq_string(const dada2hip_sample *s, int i) {
  const int l = s->h_len[i];
  std::string out((size_t)l, 'A');
  const uint32_t *row = s->h_seq2.p + (size_t)i * s->D.W2;
  for (int p = 0; p < l; p++) out[p] = "ACGT"[(row[p >> 4] >> ((p & 15) << 1)) & 3u];
  return out;
}

// items: alignments a launch will hold when that is more than one per unique (the bimera table's pairs: a few thousand
// sequences, millions of pairs - sized by N the launch would run on a dozen blocks)
void ensure_scratch(dada2hip_sample *s, int band, size_t items = 0, bool force_generic = false) {
  SampleDev &D = s->D;
  int wc = force_generic ? 0 : nw_class(band, D.maxlen, D.minlen);
  const size_t want = std::max<size_t>((size_t)D.N, items);
  if (s->scr_class == wc && s->scr_band == band && s->scr.ptr && s->scr_items >= want) return;
  s->scr_items = want;
  size_t ppw = nw_ptr_words_per_wave(wc, band, D.maxlen, D.minlen);
  int nwaves = (int)std::min<size_t>(4096, ((want + 63) / 64 + 3) & ~(size_t)3);
  const size_t budget_words = (size_t)6 << 28;  // 6 GiB of pointer scratch at most
  while (nwaves > 64 && (size_t)nwaves * ppw > budget_words) nwaves /= 2;
  nwaves = std::max(4, nwaves & ~3);
  s->scr.nwaves = nwaves;
  s->scr.ptr_words_per_wave = ppw;
  s->scr.t_words_per_wave = (size_t)((D.maxlen + 7) / 8) * 64;
  s->scr_ptr.alloc((size_t)nwaves * ppw);
  s->scr_t.alloc((size_t)nwaves * s->scr.t_words_per_wave);
  s->scr.ptr = s->scr_ptr.p;
  s->scr.tcode = s->scr_t.p;
  if (wc == 0) {
    int Wgen = band < 0 ? 2 * D.maxlen + 1 : 2 * band + (D.maxlen - D.minlen) + 1;
    s->scr.row_words_per_wave = (size_t)Wgen * 64;
    s->scr_rows.alloc((size_t)nwaves * s->scr.row_words_per_wave);
    s->scr.rows = s->scr_rows.p;
  }
  s->scr_class = wc;
  s->scr_band = band;
}

// pointer ring of the wide anti-diagonal kernel: one slot per resident wave, at most 3 GiB
void ensure_adw_scratch(dada2hip_sample *s, const AlignParams &ap) {
  if (s->scr_adw.p && s->scr_adw_band == ap.band) return;
  const size_t wpw = nw_adw_ptr_words_per_wave(s->D, ap);
  int waves = std::min(8192, nw_adw_waves(s->D, ap, s->D.N));
  while (waves > 4 && (size_t)waves * wpw > ((size_t)3 << 28)) waves = (waves / 2 + 3) & ~3;
  s->scr_adw.alloc((size_t)waves * wpw);
  s->scr_adw_waves = waves;
  s->scr_adw_wpw = wpw;
  s->scr_adw_band = ap.band;
}

// kdist > cutoff  <=>  dot < thresh[d]   with kdist = 1 - dot/d evaluated exactly as kmers.cpp:47,91 does
std::vector<int32_t> make_thresh(int maxlen, double cutoff) {
  std::vector<int32_t> t(maxlen + 2, 0);
  for (int d = 1; d <= maxlen; d++) {
    // smallest dot in [0, d] with !(1 - dot/d > cutoff); monotone in dot
    int lo = 0, hi = d + 1;
    while (lo < hi) {
      int mid = (lo + hi) / 2;
      double kd = 1. - ((double)mid) / ((double)d - 0.0);
      if (kd > cutoff) lo = mid + 1; else hi = mid;
    }
    t[d] = lo;
  }
  return t;
}

struct Comp { uint32_t i, index; double lambda; uint32_t hamming; };   // dada.h:42-47

// Host mirror of one partition (dada.h:85-105): only what is needed to replay the reference's slot
// order and to take the per-round, C-sized decisions.  All per-unique state lives on the device.
struct Bi {
  std::vector<uint32_t> raw;       // members in the reference's slot order
  uint32_t reads = 0, center = 0xFFFFFFFFu;
  char birth_type = 'I';
  uint32_t birth_from = 0;
  double birth_pval = 0, birth_fold = 1, birth_e = 0;
  Comp birth_comp{0, 0, 0, 0};
};

struct BudKeyH { double p; uint32_t reads; uint32_t pad; };

// below this many alignments in one launch the cooperative (anti-diagonal) kernel beats one-alignment-per-lane
// (measured up to 1e6 alignments: 1M-unique pass 358 -> 349 ms; beyond 4e6 untested, the lane kernel takes over)
#define COOP_MAX_BATCH (knobs().coop_max)

struct Run {
  dada2hip_sample *s;
  dada2hip_opts o;
  const dada2hip_hooks *hooks;
  int N, ncol;
  std::vector<double> err_rowmajor;
  std::vector<Bi> bi;
  std::vector<int32_t> clust_of, slot_of;       // host copies, maintained by replaying moves
  dada2hip_stats st;
  std::vector<int32_t> thresh_round, thresh_one;
  AlignParams ap;
  ScreenParams sp;
  int wclass;
  PartState P;
  int ccap = 0;                                   // capacity of the per-cluster device arrays
  // device buffers owned by the run
  DevBuf<double> d_Emin, d_clam, d_p, d_nlam, d_ph_lam;
  DevBuf<uint8_t> d_lock, d_slot0, d_upd, d_chk;
  DevBuf<int32_t> d_clof, d_ci, d_head, d_ni, d_nnext, d_ncount, d_centre, d_errflag, d_movers, d_nmovers, d_ties0, d_ties1,
      d_ph_ji, d_ph_n, d_cl_of_centre;
  DevBuf<uint32_t> d_cham, d_nham, d_creads, d_creads_snap;
  DevBuf<unsigned long long> d_totals;
  DevBuf<BudKeyH> d_partial;
  DevBuf<RoundOut> d_rout;                      // two blocks, alternating per round (k_auto_birth zeroes the next one)
  int ri = 0;
  RoundOut *ro() { return d_rout.p + ri; }
  RoundOut *ro_next() { return d_rout.p + (ri ^ 1); }
  DevBuf<uint8_t> d_lock_tmp;                   // b_p_update's lock decisions, committed by k_bud_ties
  DevBuf<int32_t> d_next;                       // [0] = centre of the speculatively launched next round, -1 = none
  PinBuf<RoundOut> h_rout;                      // the round tail's result block (one D2H per round)
  PinBuf<int32_t> h_big;                        // long mover lists
  DevBuf<int32_t> d_pool, d_thresh_one, d_thresh_round;   // zeroed counter pool; k-mer threshold tables
  size_t pool_next = 0;
  static constexpr size_t POOL_INTS = 1 << 18;
  // ---- kernel timing with HIP events on the run's stream.  Default: sampled (round 0, every 8th round, the final pass)
  // and extrapolated, because an event pair per kernel per round costs four API calls on an enqueue-bound critical
  // path.  DADA2HIP_PROFILE=1: every launch of every kernel class is timed (stats.kernel_times_sampled = 0).
  enum { EV_SCREEN = 0, EV_NW, EV_SHUFFLE, EV_PVAL, EV_BIRTH, EV_FINAL, EV_TAIL, EV_PF_SCREEN, EV_PF_NW, EV_NCLS };
  struct EvRec { hipEvent_t a, b; int cls; uint8_t ok, big; hipStream_t st; };
  std::vector<EvRec> evs;
  size_t ev_used = 0;
  int ev_round = 0;                             // rounds since the last event-timed one
  int n_round_launches = 0;
  bool profile_all = false;
  long spec_ev_nw = -1, spec_ev_screen = -1;
  int ev_begin(int cls, bool on, bool spec = false, bool big = false, hipStream_t on_stream = nullptr) {
    if (!on) return -1;
    if (ev_used == evs.size()) {
      EvRec r{};
      D2_HIP(hipEventCreate(&r.a));
      D2_HIP(hipEventCreate(&r.b));
      evs.push_back(r);
    }
    EvRec &r = evs[ev_used];
    r.cls = cls; r.ok = spec ? 0 : 1; r.big = big ? 1 : 0;
    r.st = on_stream ? on_stream : s->stream;
    D2_HIP(hipEventRecord(r.a, r.st));
    return (int)ev_used++;
  }
  void ev_end(int idx) { if (idx >= 0) D2_HIP(hipEventRecord(evs[idx].b, evs[idx].st)); }

  ~Run() {
    for (auto &e : evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto g : v2_graph) if (g) (void)hipGraphExecDestroy(g);
    for (auto e : v3_pf_ev) if (e) (void)hipEventDestroy(e);
  }

  void logf(const char *fmt, ...) {
    if (!o.verbose || !hooks || !hooks->log) return;
    char buf[512];
    va_list a;
    va_start(a, fmt);
    vsnprintf(buf, sizeof buf, fmt, a);
    va_end(a);
    hooks->log(buf, hooks->user);
  }

  // ---- one sample over several ranks (dada2hip_sample_run_sharded) ------------------------------------
  const dada2hip_shard *shard = nullptr;
  int lo = 0, hi = 0;                             // this rank's block of uniques
  std::vector<uint8_t> h_upd;                     // host copy of Bi::update_e (the device only sees its own movers)
  // (for the failure notification of dada2hip_sample_run_sharded: the exchange itself failed - it is not called again; the run's
  //  last exchange point lies behind - nobody would answer)
  bool sh_exchange_failed = false, sh_points_left = true;
  void sh_call(int kind, const void *send, int64_t nbytes, void *recv) {
    if (shard->exchange(shard->user, kind, send, nbytes, recv) != 0) {
      sh_exchange_failed = true;
      throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: the shard exchange callback failed"};
    }
  }
  // all-gather of payloads of different sizes: sizes first, then the payloads padded to the largest
  std::vector<std::vector<uint8_t>> sh_gatherv(const void *p, size_t n) {
    const int W = shard->world;
    int64_t mine = (int64_t)n;
    std::vector<int64_t> sizes(W);
    sh_call(0, &mine, 8, sizes.data());
    int64_t mx = 0;
    for (int64_t x : sizes) {
      // a rank that failed says so in place of its size (dada2hip_sample_run_sharded): every rank leaves here, together,
      // instead of waiting for it in the next collective for ever
      if (x < 0) throw PeerFailed{};
      mx = std::max(mx, x);
    }
    std::vector<std::vector<uint8_t>> out(W);
    if (mx == 0) return out;
    std::vector<uint8_t> sendb((size_t)mx, 0), recvb((size_t)mx * W);
    if (n) memcpy(sendb.data(), p, n);
    sh_call(0, sendb.data(), mx, recvb.data());
    for (int w = 0; w < W; w++) out[w].assign(recvb.begin() + (size_t)w * mx, recvb.begin() + (size_t)w * mx + (size_t)sizes[w]);
    return out;
  }
  void sh_allreduce(std::vector<int64_t> &v) {
    if (v.empty()) return;
    (void)sh_gatherv(nullptr, 0);                  // (every exchange point opens with the 8-byte size / status gather)
    sh_call(1, v.data(), (int64_t)v.size() * 8, v.data());
  }
  // the moves of one b_shuffle2 call on all ranks, in rank order (replay_moves sorts them into the reference's order)
  std::vector<int32_t> sh_all_moves(const int32_t *mine, int nm) {
    std::vector<int32_t> all;
    for (auto &part : sh_gatherv(mine, (size_t)nm * 12)) {
      const size_t k = part.size() / 4;
      const size_t o = all.size();
      all.resize(o + k);
      if (k) memcpy(all.data() + o, part.data(), k * 4);
    }
    return all;
  }
  // after the global replay: partition reads from the mirror (the device applied only its own movers' deltas), and the
  // update flags of every partition any rank's mover touched
  void sh_push_partitions(const std::vector<int32_t> &moves) {
    const int C = (int)bi.size();
    if ((int)h_upd.size() < C) h_upd.resize(C, 0);
    for (size_t k = 0; k + 2 < moves.size(); k += 3) { h_upd[moves[k + 1]] = 1; h_upd[moves[k + 2]] = 1; }
    std::vector<uint32_t> rd(C);
    for (int i = 0; i < C; i++) rd[i] = bi[i].reads;
    D2_HIP(hipMemcpyAsync(P.creads, rd.data(), (size_t)C * 4, hipMemcpyHostToDevice, s->stream));
    D2_HIP(hipMemcpyAsync(P.update_e, h_upd.data(), (size_t)C, hipMemcpyHostToDevice, s->stream));
    D2_HIP(hipStreamSynchronize(s->stream));   // (sources are locals)
  }
  // this rank's movers of the call just fetched, complete
  std::vector<int32_t> local_moves(int slot) {
    const int nm = h_rout.p->cnt[slot];
    std::vector<int32_t> mv((size_t)3 * std::max(nm, 0));
    if (nm > MOVERS_INLINE) {
      D2_HIP(hipMemcpyAsync(mv.data(), d_movers.p + (size_t)slot * 3 * N, (size_t)3 * nm * 4, hipMemcpyDeviceToHost, s->side));
      D2_HIP(hipStreamSynchronize(s->side));
    } else if (nm > 0) memcpy(mv.data(), h_rout.p->mov[slot], (size_t)3 * nm * 4);
    return mv;
  }
  // one b_shuffle2 call in sharded mode: device arg-max on the own block, then every rank replays every rank's moves
  bool sharded_shuffle() {
    D2_HIP(hipMemsetAsync(ro()->cnt, 0, 8, s->stream));
    enqueue_shuffle(0);
    fetch_round_out();
    const std::vector<int32_t> mine = local_moves(0);
    const std::vector<int32_t> all = sh_all_moves(mine.data(), (int)(mine.size() / 3));
    if (!all.empty()) replay_moves(all.data(), (int)(all.size() / 3));
    sh_push_partitions(all);
    snap_fresh = false;
    return !all.empty();
  }

  // ---- device state -----------------------------------------------------------------------------
  void alloc_state() {
    const size_t n = (size_t)N;
    if (knobs().v2_summary) {   // (what the stream still holds from before this run: not this function's time)
      const auto t_pre = clk::now();
      D2_HIP(hipStreamSynchronize(s->stream));
      fprintf(stderr, "[run] alloc_state: stream busy at entry for %.2f ms\n", ms_since(t_pre));
    }
    const auto t_as = clk::now();
    d_Emin.alloc(n); d_clam.alloc(n); d_p.alloc(n); d_lock.alloc(n); d_slot0.alloc(n); d_clof.alloc(n); d_ci.alloc(n);
    d_cham.alloc(n); d_head.alloc(n); d_ncount.alloc(1); d_errflag.alloc(1); d_movers.alloc(6 * n); d_nmovers.alloc(1);
    d_ties0.alloc(n); d_ties1.alloc(n); d_totals.alloc(4); d_partial.alloc(2 * (size_t)8192); d_rout.alloc(2); d_next.alloc(4); d_lock_tmp.alloc(n);   // (d_partial: block partials of k2_pupdate, grid capped at 8192)
    h_rout.alloc(1); d_pool.alloc(POOL_INTS);
    d_thresh_one.alloc(thresh_one.size()); d_thresh_round.alloc(thresh_round.size());
    P.E_minmax = d_Emin.p; P.comp_lam = d_clam.p; P.p = d_p.p; P.lock = d_lock.p; P.slot0 = d_slot0.p; P.clust_of = d_clof.p;
    P.comp_i = d_ci.p; P.comp_ham = d_cham.p; P.head = d_head.p; P.node_count = d_ncount.p; P.err_flag = d_errflag.p;
    P.totals = d_totals.p;
    {   // comparison store: grows by doubling (decide_bud); DADA2HIP_NODE_CAP shrinks the first allocation (test knob)
      size_t cap0 = std::max<size_t>(4 * n, 1u << 20);
      if (knobs().node_cap > 0) cap0 = std::max<size_t>((size_t)knobs().node_cap, n + 16);
      grow_nodes(cap0);
    }
    grow_clusters(256);
    hipStream_t stq = s->stream;
    const double ms_as_alloc = ms_since(t_as);
    // (buffers persist across runs of the same sample: clear what a previous run left behind)
    D2_HIP(hipMemsetAsync(P.creads, 0, (size_t)ccap * 4, stq));
    D2_HIP(hipMemsetAsync(d_creads_snap.p, 0, (size_t)ccap * 4, stq));
    D2_HIP(hipMemsetAsync(P.centre_of, 0, (size_t)ccap * 4, stq));
    D2_HIP(hipMemsetAsync(P.update_e, 0, (size_t)ccap, stq));
    D2_HIP(hipMemsetAsync(P.check_locks, 0, (size_t)ccap, stq));
    bi.clear();
    ev_used = 0;
    ev_round = 0; n_round_launches = 0;
    profile_all = knobs().profile;
    D2_HIP(hipMemsetAsync(d_rout.p, 0, 2 * sizeof(RoundOut), stq));
    D2_HIP(hipMemsetAsync(d_next.p, 0xFF, 16, stq));
    ri = 0;
    spec_launched = false;
    publish_pending = false;
    h_rout.p->seq = 0;
    launch_fill_f64(d_Emin.p, n, -999.0, stq);                   // containers.cpp:39
    D2_HIP(hipMemsetAsync(d_clam.p, 0, n * 8, stq));
    D2_HIP(hipMemsetAsync(d_p.p, 0, n * 8, stq));
    D2_HIP(hipMemsetAsync(d_lock.p, 0, n, stq));
    D2_HIP(hipMemsetAsync(d_slot0.p, 0, n, stq));
    D2_HIP(hipMemsetAsync(d_clof.p, 0, n * 4, stq));
    D2_HIP(hipMemsetAsync(d_ci.p, 0, n * 4, stq));
    D2_HIP(hipMemsetAsync(d_cham.p, 0, n * 4, stq));
    D2_HIP(hipMemsetAsync(d_head.p, 0xFF, n * 4, stq));          // -1
    D2_HIP(hipMemsetAsync(d_ncount.p, 0, 4, stq));
    D2_HIP(hipMemsetAsync(d_errflag.p, 0, 4, stq));
    D2_HIP(hipMemsetAsync(d_totals.p, 0, 32, stq));
    have_pending_store = false;
    snap_fresh = true;   // sample_run copies partition 0's reads into the snapshot right after
    D2_HIP(hipMemsetAsync(d_pool.p, 0, POOL_INTS * 4, stq));
    pool_next = 0;
    D2_HIP(hipMemcpyAsync(d_thresh_one.p, thresh_one.data(), thresh_one.size() * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(d_thresh_round.p, thresh_round.data(), thresh_round.size() * 4, hipMemcpyHostToDevice, stq));
    const double ms_as_enq = ms_since(t_as);
    D2_HIP(hipStreamSynchronize(stq));                           // the threshold vectors may be rebuilt by the next run
    if (knobs().v2_summary) fprintf(stderr, "[run] alloc_state: buffers %.2f ms, + clears enqueued %.2f, + stream drained %.2f\n", ms_as_alloc, ms_as_enq, ms_since(t_as));
  }

  void grow_nodes(size_t cap) {
    if (cap <= (size_t)P.node_cap) return;
    if (cap > 0x7FFFFFF0u) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: comparison store exceeds 2^31 entries"};
    DevBuf<int32_t> ni, nn;
    DevBuf<double> nl;
    DevBuf<uint32_t> nh;
    ni.alloc(cap); nn.alloc(cap); nl.alloc(cap); nh.alloc(cap);
    if (P.node_cap > 0) {
      D2_HIP(hipMemcpyAsync(ni.p, d_ni.p, (size_t)P.node_cap * 4, hipMemcpyDeviceToDevice, s->stream));
      D2_HIP(hipMemcpyAsync(nn.p, d_nnext.p, (size_t)P.node_cap * 4, hipMemcpyDeviceToDevice, s->stream));
      D2_HIP(hipMemcpyAsync(nl.p, d_nlam.p, (size_t)P.node_cap * 8, hipMemcpyDeviceToDevice, s->stream));
      D2_HIP(hipMemcpyAsync(nh.p, d_nham.p, (size_t)P.node_cap * 4, hipMemcpyDeviceToDevice, s->stream));
      D2_HIP(hipStreamSynchronize(s->stream));
    }
    std::swap(d_ni.p, ni.p); std::swap(d_ni.n, ni.n);
    std::swap(d_nnext.p, nn.p); std::swap(d_nnext.n, nn.n);
    std::swap(d_nlam.p, nl.p); std::swap(d_nlam.n, nl.n);
    std::swap(d_nham.p, nh.p); std::swap(d_nham.n, nh.n);
    P.node_i = d_ni.p; P.node_next = d_nnext.p; P.node_lam = d_nlam.p; P.node_ham = d_nham.p;
    P.node_cap = (int32_t)cap;
  }

  void grow_clusters(int cap) {
    if (cap <= ccap) return;
    DevBuf<uint32_t> cr, cs;
    DevBuf<int32_t> ce;
    DevBuf<uint8_t> up, ch;
    cr.alloc(cap); cs.alloc(cap); ce.alloc(cap); up.alloc(cap); ch.alloc(cap);
    hipStream_t stq = s->stream;
    D2_HIP(hipMemsetAsync(cr.p, 0, (size_t)cap * 4, stq));
    D2_HIP(hipMemsetAsync(cs.p, 0, (size_t)cap * 4, stq));
    D2_HIP(hipMemsetAsync(ce.p, 0, (size_t)cap * 4, stq));
    D2_HIP(hipMemsetAsync(up.p, 0, (size_t)cap, stq));
    D2_HIP(hipMemsetAsync(ch.p, 0, (size_t)cap, stq));
    if (ccap > 0) {
      D2_HIP(hipMemcpyAsync(cr.p, d_creads.p, (size_t)ccap * 4, hipMemcpyDeviceToDevice, stq));
      D2_HIP(hipMemcpyAsync(cs.p, d_creads_snap.p, (size_t)ccap * 4, hipMemcpyDeviceToDevice, stq));
      D2_HIP(hipMemcpyAsync(ce.p, d_centre.p, (size_t)ccap * 4, hipMemcpyDeviceToDevice, stq));
      D2_HIP(hipMemcpyAsync(up.p, d_upd.p, (size_t)ccap, hipMemcpyDeviceToDevice, stq));
      D2_HIP(hipMemcpyAsync(ch.p, d_chk.p, (size_t)ccap, hipMemcpyDeviceToDevice, stq));
    }
    D2_HIP(hipStreamSynchronize(stq));
    std::swap(d_creads.p, cr.p); std::swap(d_creads.n, cr.n);
    std::swap(d_creads_snap.p, cs.p); std::swap(d_creads_snap.n, cs.n);
    std::swap(d_centre.p, ce.p); std::swap(d_centre.n, ce.n);
    std::swap(d_upd.p, up.p); std::swap(d_upd.n, up.n);
    std::swap(d_chk.p, ch.p); std::swap(d_chk.n, ch.n);
    P.creads = d_creads.p; P.centre_of = d_centre.p; P.update_e = d_upd.p; P.check_locks = d_chk.p;
    ccap = cap;
  }

  // 8 zeroed ints from the counter pool (no per-round memset launches)
  int32_t *pool8() {
    if (pool_next + 8 > POOL_INTS) {
      D2_HIP(hipMemsetAsync(d_pool.p, 0, POOL_INTS * 4, s->stream));
      pool_next = 0;
    }
    int32_t *p = d_pool.p + pool_next;
    pool_next += 8;
    return p;
  }

  // publish a (new) partition's centre / reads / flags to the device
  void push_cluster(int i, bool update_e, bool check_locks) {
    hipStream_t stq = s->stream;
    if (i >= ccap) grow_clusters(std::max(ccap * 2, i + 1));
    const uint32_t rd = bi[i].reads;
    const int32_t ce = (int32_t)bi[i].center;
    const uint8_t u = update_e, c = check_locks;
    D2_HIP(hipMemcpyAsync(P.creads + i, &rd, 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(P.centre_of + i, &ce, 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(P.update_e + i, &u, 1, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(P.check_locks + i, &c, 1, hipMemcpyHostToDevice, stq));
    D2_HIP(hipStreamSynchronize(stq));   // sources are stack temporaries
  }

  // ---- one b_compare round, entirely on the device (cluster.cpp:90-204) ---------------------------
  // `centre` is passed explicitly: the host mirror of the new partition may be filled in after the launches.
  // spec: launched behind k_auto_birth before the host has seen the bud decision; the kernels take the centre from
  // d_next (and do nothing when it is -1); confirm_spec() books the round once the host knows the birth happened.
  bool spec_launched = false;
  int32_t *spec_ctr = nullptr;
  bool rounds_use_coop() const {
    const int f = knobs().nw_kernel;
    if (f == NWK_LANE || f == NWK_WIDE) return false;
    return nw_ad_lds_bytes(s->D, ap) > 0 && nw_ad_lds_bytes(s->D, ap) <= 150 * 1024;
  }
  void confirm_spec(int ci, int centre) {
    pending_store = StoreRound{ci, centre, (double)(uint32_t)s->total_reads, s->d_lambda.p, s->d_ham.p, spec_ctr, s->d_cls.p};
    have_pending_store = true;
    spec_launched = false;
    if (spec_ev_screen >= 0) evs[spec_ev_screen].ok = 1;
    if (spec_ev_nw >= 0) evs[spec_ev_nw].ok = 1;
    n_round_launches++;
    st.ncompare += (uint64_t)N;
  }
  void compare_round(int ci, int centre, double cutoff, bool spec = false) {
    SampleDev &D = s->D;
    hipStream_t stq = s->stream;
    auto t0 = clk::now();
    const int32_t *th = (cutoff == 1.0) ? d_thresh_one.p : d_thresh_round.p;
    int32_t *ctr = pool8();
    const bool timed = profile_all || ci == 0 || (ev_round++ % 8) == 0;
    const int evs_i = ev_begin(EV_SCREEN, timed, spec);
    spec_ev_screen = spec ? evs_i : -1;
    launch_screen(D, centre, sp, nullptr, P.lock, o.greedy, th, s->d_cls.p, s->d_lambda.p, s->d_ham.p,
                  s->d_nw_list.p, s->d_gl_list.p, ctr, s->d_ctab.p, /*build_table=*/ci == 0, spec ? d_next.p : nullptr, stq);
    ev_end(evs_i);
    // NW batch size is only known on the device: both kernels loop over the device-side count with a
    // fixed persistent grid.  Round 0 aligns every unique (lane-per-alignment kernel), later rounds a few
    // thousand (cooperative kernel).
    const int evn_i = ev_begin(EV_NW, timed, spec, /*big=*/ci == 0);
    spec_ev_nw = spec ? evn_i : -1;
    const int f = knobs().nw_kernel;
    const bool coop_ok = nw_ad_lds_bytes(D, ap) > 0 && nw_ad_lds_bytes(D, ap) <= 150 * 1024;
    bool coop = coop_ok && (ci != 0 || N < COOP_MAX_BATCH);
    if (f == NWK_LANE) coop = false;
    if (f == NWK_COOP && coop_ok) coop = true;
    // band windows too wide for the LDS-pointer kernel (ragged long reads): eight cells per lane, pointers in HBM
    bool wide = !coop && nw_adw_ok(D, ap) && nw_adw_lds_bytes(D, ap) <= 150 * 1024 && wclass != 33 && wclass != 65 &&
                N < (1 << 20);
    if (f == NWK_LANE) wide = false;
    if (f == NWK_WIDE) wide = nw_adw_ok(D, ap) && nw_adw_lds_bytes(D, ap) <= 150 * 1024;
    if (coop && !wide)   // the gapless pairings of the round share the kernel's factor/product tail
      launch_nw_ad(D, centre, nullptr, s->d_nw_list.p, ctr, 0, s->d_gl_list.p, ctr + 1, ap, s->d_err.p, s->d_lambda.p,
                   s->d_ham.p, nullptr, 0, 0, spec ? d_next.p : nullptr, stq);
    else {
      launch_gapless(D, centre, nullptr, s->d_gl_list.p, ctr + 1, 0, ap, s->d_err.p, s->d_lambda.p, s->d_ham.p, nullptr, 0, 0, stq);
      if (wide) {
        ensure_adw_scratch(s, ap);
        launch_nw_adw(D, centre, nullptr, s->d_nw_list.p, ctr, 0, ap, s->d_err.p, s->scr_adw.p, s->scr_adw_wpw,
                      s->scr_adw_waves, s->d_lambda.p, s->d_ham.p, nullptr, 0, 0, stq);
      } else {
        ensure_scratch(s, ap.band);
        launch_nw(D, wclass, centre, nullptr, s->d_nw_list.p, ctr, 0, ap, s->d_err.p, s->scr, s->d_lambda.p,
                  s->d_ham.p, nullptr, 0, 0, nullptr, 0, nullptr, stq);
      }
    }
    ev_end(evn_i);
    if (spec) { spec_ctr = ctr; spec_launched = true; st.ms_screen += ms_since(t0); return; }
    if (ci == 0) {   // round 0 is followed by b_p_update directly (Rmain.cpp:309-311)
      const int ev = ev_begin(EV_SHUFFLE, profile_all);
      if (use_v2) launch2_store0(E2, s->d_lambda.p, s->d_ham.p, s->d_cls.p, ctr, stq);
      else launch_store(P, D, ci, centre, (double)(uint32_t)s->total_reads, s->d_lambda.p, s->d_ham.p, ctr, s->d_cls.p,
                        ro()->cnt, stq);
      ev_end(ev);
    }
    else {         // later rounds: the store filter rides in front of the round's first shuffle
      pending_store = StoreRound{ci, centre, (double)(uint32_t)s->total_reads, s->d_lambda.p, s->d_ham.p, ctr, s->d_cls.p};
      have_pending_store = true;
    }
    n_round_launches++;
    st.ncompare += (uint64_t)N;
    st.ms_screen += ms_since(t0);
  }

  // ---- b_shuffle2: device arg-max + move, host replay of the moves in the reference's order -------
  // replay a batch of device moves in the reference's order: partitions ascending, slots descending
  // (cluster.cpp:242-259), keeping bi_pop_raw's swap-with-last / bi_add_raw's append (containers.cpp:150-197)
  // (persistent tail) the replay of a long mover list takes milliseconds in which the device can run out of queued launches:
  // it looks at the device's launch counter every few thousand moves
  bool v3_running = false;
  std::vector<std::pair<uint64_t, int32_t>> replay_keys, replay_tmp;
  std::vector<int32_t> replay_order;
  void replay_moves(const int32_t *mv, int nm) {
    const bool feed = v3_running && !lane_is_me();                  // (launches are the boundary thread's to send)
    if (feed && nm > 2048) v3_topup();
    // The reference pops the movers of one b_shuffle2 call partition by partition, each partition's from its LAST slot down
    // (cluster.cpp:241-262 walks r = nraw - 1 .. 0): by source partition ascending, then by the slot the unique holds now, descending.
    // One 64-bit key per move (the keys are distinct: a slot holds one unique), sorted as integers - the comparator used to look
    // slot_of[] up twice per comparison.
    replay_keys.resize((size_t)nm);
    uint32_t max_slot = 0, max_from = 0;
    for (int k = 0; k < nm; k++) {
      const uint32_t sl = (uint32_t)slot_of[mv[3 * k]], fr = (uint32_t)mv[3 * k + 1];
      max_slot = std::max(max_slot, sl); max_from = std::max(max_from, fr);
      replay_keys[k] = {((uint64_t)fr << 32) | (uint64_t)sl, (int32_t)k};
    }
    // (descending slots as ascending keys, against the largest slot of THIS list: few significant bits for the radix passes below)
    for (int k = 0; k < nm; k++) replay_keys[k].first = (replay_keys[k].first & 0xFFFFFFFF00000000ull) | (uint64_t)(max_slot - (uint32_t)replay_keys[k].first);
    if (nm < std::max(2, knobs().replay_radix_min))
      std::sort(replay_keys.begin(), replay_keys.end(), [](const std::pair<uint64_t, int32_t> &x, const std::pair<uint64_t, int32_t> &y) { return x.first < y.first; });
    else {
      // the early rounds move 10^5 uniques per call: a comparison sort of such a list was a third of the pass's replay.  LSD radix
      // sort, 11 bits a pass, over the bits the keys actually use (slots below the list's largest, then the source partitions)
      auto bits_of = [](uint32_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; };
      const int bs = bits_of(max_slot), bf = bits_of(max_from);
      replay_tmp.resize((size_t)nm);
      std::pair<uint64_t, int32_t> *src = replay_keys.data(), *dst = replay_tmp.data();
      auto pass = [&](int shift, int nbits) {
        const uint32_t mask = (1u << nbits) - 1u;
        uint32_t hist[2048 + 1] = {0};
        for (int k = 0; k < nm; k++) hist[((src[k].first >> shift) & mask) + 1]++;
        for (uint32_t d = 0; d < mask + 1u; d++) hist[d + 1] += hist[d];
        for (int k = 0; k < nm; k++) dst[hist[(src[k].first >> shift) & mask]++] = src[k];
        std::swap(src, dst);
      };
      for (int done = 0; done < bs; done += 11) pass(done, std::min(11, bs - done));
      for (int done = 0; done < bf; done += 11) pass(32 + done, std::min(11, bf - done));
      if (src != replay_keys.data()) std::copy(src, src + nm, replay_keys.data());
    }
    std::vector<int32_t> &order = replay_order;
    order.resize((size_t)nm);
    for (int k = 0; k < nm; k++) order[k] = replay_keys[k].second;
    bool slot0_changed = false;
    int tick = 0;
    const uint32_t *h_reads = s->h_reads.data();
    for (int q = 0; q < nm; q++) {
      const int k = order[q];
      if (feed && (++tick & 4095) == 0) v3_topup();
      // (a move is half a dozen dependent cache misses on a 10^6-unique mirror: the lines of the moves to come are requested in two
      //  stages - the unique's own words 24 moves ahead; 12 ahead, with its slot at hand, the member-list cells it will touch)
      if (q + 24 < nm) {
        const uint32_t rn = (uint32_t)mv[3 * order[q + 24]];
        __builtin_prefetch(&slot_of[rn], 1); __builtin_prefetch(&clust_of[rn], 1); __builtin_prefetch(&h_reads[rn], 0);
      }
      if (q + 12 < nm) {
        const int kn = order[q + 12];
        const Bi &bfn = bi[mv[3 * kn + 1]], &btn = bi[mv[3 * kn + 2]];
        const size_t sl = (size_t)slot_of[(uint32_t)mv[3 * kn]];
        if (sl < bfn.raw.size()) __builtin_prefetch(bfn.raw.data() + sl, 1);
        if (!bfn.raw.empty()) { __builtin_prefetch(bfn.raw.data() + bfn.raw.size() - 1, 0); __builtin_prefetch(&slot_of[bfn.raw.back()], 1); }
        if (btn.raw.capacity()) __builtin_prefetch(btn.raw.data() + btn.raw.size(), 1);
      }
      const uint32_t r = (uint32_t)mv[3 * k];
      const int from = mv[3 * k + 1], to = mv[3 * k + 2];
      Bi &bf = bi[from];
      const int slot = slot_of[r];
      const uint32_t last = bf.raw.back();
      bf.raw[slot] = last;
      slot_of[last] = slot;
      bf.raw.pop_back();
      bf.reads -= s->h_reads[r];
      if (slot == 0) slot0_changed = true;
      Bi &bt = bi[to];
      slot_of[r] = (int32_t)bt.raw.size();
      bt.raw.push_back(r);
      bt.reads += s->h_reads[r];
      clust_of[r] = to;
    }
    if (slot0_changed) push_slot0();
  }

  // enqueue one b_shuffle2 (device arg-max + move); count and first movers go to slot `slot` of the round's result block.
  // The arg-max uses the partition reads as of the start of the call: a snapshot, refreshed by k_apply_bud /
  // k_auto_birth ahead of every round and by an explicit copy before any further shuffle of the same round.
  // (Whether a further call would move anything is asked inside k_pupdate_budmin, which needs no snapshot.)
  bool snap_fresh = true;
  void enqueue_shuffle(int slot) {
    hipStream_t stq = s->stream;
    RoundOut *ro = this->ro();
    if (!snap_fresh)
      D2_HIP(hipMemcpyAsync(d_creads_snap.p, P.creads, (size_t)nclust_dev * 4, hipMemcpyDeviceToDevice, stq));
    const int ev = ev_begin(EV_SHUFFLE, profile_all);
    launch_shuffle(P, s->D, d_creads_snap.p, d_movers.p + (size_t)slot * 3 * N, ro->cnt + slot, ro->mov[slot],
                   have_pending_store ? &pending_store : nullptr, 0, nclust_dev, stq);
    ev_end(ev);
    have_pending_store = false;
    snap_fresh = false;
    st.nshuffle++;
  }
  int nclust_dev = 1;   // partitions the device knows about (the host mirror may lag by one birth)
  StoreRound pending_store{};
  bool have_pending_store = false;

  void fetch_round_out() {
    if (publish_pending) {
      // k_auto_birth stores the block into h_rout itself and writes the sequence number last: poll it (the speculative
      // kernels of the next round keep the stream busy meanwhile); look at the stream now and then so that a device
      // fault cannot leave us spinning
      volatile int32_t *seqp = &h_rout.p->seq;
      const auto tw = clk::now();
      const bool polite = wait_blocks();
      for (unsigned spins = 0; *seqp != publish_seq; spins++) {
        cpu_relax();
        if (polite && spins > 64) { struct timespec ts{0, 20000}; nanosleep(&ts, nullptr); }
        if ((spins & 0xFFFF) == 0xFFFF || (polite && (spins & 0xFF) == 0xFF)) {
          hipError_t e = hipStreamQuery(s->stream);
          if (e != hipSuccess && e != hipErrorNotReady)
            throw d2::DeviceError{DADA2HIP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(e) + " (round result)"};
          if (e == hipSuccess && *seqp != publish_seq)     // the stream drained and the block never came
            throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: the device did not publish the round result"};
          if (ms_since(tw) > wait_timeout_s() * 1e3)
            throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: timed out waiting for the round result"};
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      publish_pending = false;
    } else {
      D2_HIP(hipMemcpyAsync(h_rout.p, ro(), sizeof(RoundOut), hipMemcpyDeviceToHost, s->stream));
      sync_spin(s->stream);
    }
    D2_HIP(hipGetLastError());
  }
  bool publish_pending = false;
  int32_t publish_seq = 0;

  // replay the movers a real shuffle reported: `inl` = the first MOVERS_INLINE of them (host memory), the full list
  // stays in d_movers slot `slot` when there are more
  void apply_moves(int nm, const int32_t *inl, int slot) {
    if (nm <= 0) return;
    if (nm > MOVERS_INLINE) {
      // the shuffle that wrote this list has completed (its result block has been fetched); copy it on the side stream
      // so the copy does not wait for the kernels already queued on the main stream
      h_big.alloc((size_t)3 * nm);
      D2_HIP(hipMemcpyAsync(h_big.p, d_movers.p + (size_t)slot * 3 * N, (size_t)3 * nm * 4, hipMemcpyDeviceToHost, s->side));
      D2_HIP(hipStreamSynchronize(s->side));
      replay_moves(h_big.p, nm);
    } else replay_moves(inl, nm);
  }
  void apply_shuffle_result(int slot) { apply_moves(h_rout.p->cnt[slot], h_rout.p->mov[slot], slot); }

  // b_p_update + b_bud evaluation; when the rounds run on the cooperative kernel also: the unambiguous birth applied on
  // the device (k_auto_birth) and the next round's screen + alignments queued behind it, so the GPU does not wait for
  // the host's decision (the host confirms it afterwards from the same result block).
  int max_clust_run = 0;
  void enqueue_pupdate_bud(int32_t *check_cnt) {
    BudParams bp{o.min_fold, o.omegaA, o.omegaP, o.min_hamming, o.min_abund};
    const bool autob = !no_auto && rounds_use_coop() && nclust_dev < max_clust_run;
    if (autob && nclust_dev + 1 >= ccap) grow_clusters(std::max(ccap * 2, nclust_dev + 2));   // before anything is enqueued
    int ev = ev_begin(EV_PVAL, profile_all);
    launch_pupdate_bud(P, s->D, o.greedy, o.detect_singletons, bp, 1.0, s->h_reads[bi[0].center], d_partial.p, &ro()->bud,
                       d_ties0.p, d_ties1.p, nclust_dev, d_lock_tmp.p, check_cnt, s->stream);
    ev_end(ev);
    if (!autob) return;
    publish_seq = (publish_seq % 0x3FFFFFFF) + 1;
    ev = ev_begin(EV_BIRTH, profile_all);
    launch_auto_birth(P, s->D, d_creads_snap.p, ro(), o.omegaA, nclust_dev, s->d_ctab.p, ro_next()->cnt, d_next.p, h_rout.p,
                      publish_seq, s->stream);
    ev_end(ev);
    publish_pending = true;
    compare_round(nclust_dev, -1, o.kdist_cutoff, /*spec=*/true);
  }

  // The tail of one divisive round (Rmain.cpp:320-329 + the b_bud of the next iteration, :316): shuffle until
  // stable (at most MAX_SHUFFLE), b_p_update, b_bud.  The common case is "first shuffle moves, second does
  // not", so two shuffles, the p-value update and the bud evaluation are enqueued back to back and fetched
  // with ONE synchronisation and ONE copy; the p-update/bud kernels cancel themselves on the device if the second
  // shuffle still moved something, and the host then continues shuffling exactly as the reference would.
  // On return h_rout holds a valid bud evaluation; moves not yet replayed on the host are described by
  // pending_slot (the caller replays them after it has launched the next round).
  // plain: one shuffle per host round trip, every evaluation enqueued after the host mirror (and slot0[]) is current.
  // Forced by DADA2HIP_NO_SPECULATION and whenever partition 0's first slot is not its centre (input that is not
  // abundance-sorted): b_bud's "slot 0 is the centre" quirk (cluster.cpp:285) then depends on the slot order, which only
  // the host maintains, so nothing may be evaluated ahead of the host's replay.
  bool plain = false, no_auto = false;
  int pending_slot = -1;   // list slot of the last real shuffle whose moves the host has not replayed yet
  std::vector<int32_t> prev_inline;
  void round_tail(bool do_shuffle) {
    auto t0 = clk::now();
    int nsh = 0;
    int32_t *guard = nullptr;
    if (do_shuffle && plain) {                          // the reference's plain loop (test knob, unsorted input, sharded runs)
      bool shuffled = true;
      while (shuffled && nsh < MAX_SHUFFLE) {
        if (shard) shuffled = sharded_shuffle();
        else {
          D2_HIP(hipMemsetAsync(ro()->cnt, 0, 8, s->stream));
          enqueue_shuffle(0);
          fetch_round_out();
          shuffled = h_rout.p->cnt[0] > 0;
          apply_shuffle_result(0);
        }
        nsh++;
      }
      enqueue_pupdate_bud(nullptr);
      fetch_round_out();
      if (shard) std::fill(h_upd.begin(), h_upd.end(), (uint8_t)0);   // b_p_update has consumed the flags (pval.cpp:24,37)
      pending_slot = -1;
      st.ms_bookkeep += ms_since(t0);
      return;
    }
    if (shard && !do_shuffle) {                         // (after round 0: b_p_update + the first b_bud)
      enqueue_pupdate_bud(nullptr);
      fetch_round_out();
      std::fill(h_upd.begin(), h_upd.end(), (uint8_t)0);
      pending_slot = -1;
      st.ms_bookkeep += ms_since(t0);
      return;
    }
    int slot = 0;
    if (do_shuffle) {
      enqueue_shuffle(slot);
      guard = ro()->cnt + (slot ^ 1);                   // "would a second call move anything?" rides in the p-update kernel
      st.nshuffle++;
      nsh = 2;
    }
    enqueue_pupdate_bud(guard);
    fetch_round_out();
    pending_slot = do_shuffle ? slot : -1;
    if (do_shuffle) {
      if (h_rout.p->cnt[slot] == 0) st.nshuffle--;     // the reference stops after the first unmoving shuffle
      // Speculation cancelled: the check says the next call moves uniques too.  Keep shuffling like the reference, each
      // further real call again enqueued together with a check of the call after it and the p-update / bud
      // evaluation - and BEFORE the host replays the previous call's moves, so the GPU is not left waiting for that.
      while (h_rout.p->cnt[slot ^ 1] > 0) {
        const int nm_prev = h_rout.p->cnt[slot], slot_prev = slot;
        prev_inline.assign(h_rout.p->mov[slot], h_rout.p->mov[slot] + 3 * std::min(nm_prev, MOVERS_INLINE));
        st.nshuffle--; nsh--;                           // the check stood for a call that does move: redo it for real
        slot ^= 1;
        D2_HIP(hipMemsetAsync(ro()->cnt, 0, 8, s->stream));
        enqueue_shuffle(slot);
        nsh++;
        const bool last = nsh >= MAX_SHUFFLE;           // Rmain.cpp:321: the loop stops at MAX_SHUFFLE calls regardless
        int32_t *g2 = nullptr;
        if (!last) { g2 = ro()->cnt + (slot ^ 1); st.nshuffle++; nsh++; }
        enqueue_pupdate_bud(g2);
        apply_moves(nm_prev, prev_inline.data(), slot_prev);
        fetch_round_out();
        pending_slot = slot;
        if (last) break;                                // out of calls: the evaluation just fetched is valid
      }
    }
    st.ms_bookkeep += ms_since(t0);
  }
  void replay_pending() {
    if (pending_slot >= 0) apply_shuffle_result(pending_slot);
    pending_slot = -1;
  }

  // last round when max_clust stops the loop (Rmain.cpp:316): only the shuffles matter for the outputs
  void round_tail_no_bud() {
    auto t0 = clk::now();
    int nsh = 0;
    bool shuffled;
    do {
      if (shard) { shuffled = sharded_shuffle(); continue; }
      D2_HIP(hipMemsetAsync(ro()->cnt, 0, 8, s->stream));
      enqueue_shuffle(0);
      fetch_round_out();
      shuffled = h_rout.p->cnt[0] > 0;
      apply_shuffle_result(0);
    } while (shuffled && ++nsh < MAX_SHUFFLE);
    st.ms_bookkeep += ms_since(t0);
  }

  void push_slot0() {   // only reachable when a slot-0 unique is not its partition's centre (unsorted input: plain mode)
    if (lane_is_me()) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: a slot-0 member moved under the device-driven rounds"};
    std::vector<uint8_t> f(N, 0);
    for (auto &b : bi) if (!b.raw.empty()) f[b.raw[0]] = 1;
    D2_HIP(hipMemcpyAsync(P.slot0, f.data(), (size_t)N, hipMemcpyHostToDevice, s->stream));   // ordered with the evaluations
    D2_HIP(hipStreamSynchronize(s->stream));                                                    // `f` is a local
  }

  // ---- b_bud (cluster.cpp:274-350): device arg-min, host tie-break in (partition, slot) order ------
  struct Birth { bool yes = false; char type = 'A'; BudTie c{}; double pval = 0; int newi = 0; };

  // get_pA (pval.cpp:67-89) on the host, i.e. with the CPU reference's libm
  double host_get_pA(const BudTie &t) const {
    const uint32_t reads = s->h_reads[t.raw];
    const bool prior = s->h_prior[t.raw] != 0;
    if (reads == 1 && !prior && !o.detect_singletons) return 1.;
    if (t.comp_ham == 0) return 1.;
    if (t.comp_lam == 0) return 0.;
    return pp::calc_pA((int)reads, t.comp_lam * t.from_reads, prior || o.detect_singletons);
  }

  // Decide from the fetched evaluation.  The device lists the exact ties of its best key and every candidate whose
  // p-value is within BUD_NEAR of it (engine.h).  With a single listed candidate (the normal case) everything needed
  // comes from the device, so the next round can be launched before the host mirror is brought up to date.  With
  // several, the host recomputes their p-values with its own libm and applies b_bud's rule itself (cluster.cpp:284-308:
  // p ascending, reads descending, first in (partition, slot) scan order), which needs the slot order to be current.
  // The birth p-value reported is always the host's (bit-identical to the CPU reference given the same pgamma source).
  Birth decide_bud() { return decide_bud(h_rout.p->bud); }
  Birth decide_bud(const BudOut &h) {
    if (!h.valid) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: bud evaluation not valid"};
    check_errflag(h.err_flag);
    if (!use_v2) st.nstored = (uint64_t)h.node_count;
    if (!use_v2 && (size_t)h.node_count + (size_t)N > (size_t)P.node_cap)
      grow_nodes(std::max((size_t)P.node_cap * 2, (size_t)h.node_count + 2 * (size_t)N));
    auto pick = [&](int track, BudTie &out, double &p_out) -> bool {
      int n = h.nties[track];
      if (!shard) {
        if (!h.found[track] || n <= 0) return false;
        if (n == 1) { out = h.ties[track][0]; p_out = host_get_pA(out); return true; }
      } else if (!h.found[track] || n < 0) n = 0;         // (this rank has no candidate; others may)
      replay_pending();                                   // slot order and partition reads must be current
      std::vector<BudTie> cand;
      if (n == 0) { }
      else if (n <= BUD_TIES) cand.assign(h.ties[track], h.ties[track] + n);
      else if (use_v2 && n <= TIES_FULL) {   // the device keeps the full records (it is halted: nothing rewrites them)
        cand.resize(n);
        D2_HIP(hipMemcpyAsync(cand.data(), v2_tiesrec.p + (size_t)track * TIES_FULL, (size_t)n * sizeof(BudTie), hipMemcpyDeviceToHost, s->side));
        D2_HIP(hipStreamSynchronize(s->side));
      } else {   // long list (e.g. many p == 0 with equal reads): fetch it and the fields of its members
        std::vector<int32_t> t(n);
        sync_spin(s->stream);
        D2_HIP(hipMemcpy(t.data(), track ? d_ties1.p : d_ties0.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        std::vector<double> lam(N);
        std::vector<uint32_t> ham(N);
        std::vector<int32_t> ci(N);
        D2_HIP(hipMemcpy(lam.data(), P.comp_lam, (size_t)N * 8, hipMemcpyDeviceToHost));
        D2_HIP(hipMemcpy(ham.data(), P.comp_ham, (size_t)N * 4, hipMemcpyDeviceToHost));
        D2_HIP(hipMemcpy(ci.data(), P.comp_i, (size_t)N * 4, hipMemcpyDeviceToHost));
        cand.resize(n);
        for (int k = 0; k < n; k++) {
          BudTie &c = cand[k];
          c.raw = t[k]; c.from = clust_of[t[k]]; c.from_reads = bi[c.from].reads;
          c.comp_i = ci[t[k]]; c.comp_lam = lam[t[k]]; c.comp_ham = ham[t[k]]; c.p = 0; c.pad = 0;
        }
      }
      if (shard) {
        // every rank lists the exact and near ties of ITS best key; the union holds the ties of the best key of all (a worse
        // rank's window lies above the global one), and b_bud's rule below picks among them with the host's arithmetic
        std::vector<BudTie> all;
        for (auto &part : sh_gatherv(cand.data(), cand.size() * sizeof(BudTie))) {
          const size_t k = part.size() / sizeof(BudTie), o = all.size();
          all.resize(o + k);
          if (k) memcpy((void *)(all.data() + o), part.data(), k * sizeof(BudTie));
        }
        cand.swap(all);
        n = (int)cand.size();
        if (n == 0) return false;
      }
      auto before = [&](int a, int b) { return clust_of[a] < clust_of[b] || (clust_of[a] == clust_of[b] && slot_of[a] < slot_of[b]); };
      int bk = -1;
      double bp = 0;
      for (int k = 0; k < n; k++) {
        const double pk = host_get_pA(cand[k]);
        const uint32_t rk = s->h_reads[cand[k].raw];
        bool better = bk < 0;
        if (!better) {
          const uint32_t rb = s->h_reads[cand[bk].raw];
          better = pk < bp || (pk == bp && (rk > rb || (rk == rb && before(cand[k].raw, cand[bk].raw))));
        }
        if (better) { bk = k; bp = pk; }
      }
      out = cand[bk];
      p_out = bp;
      return true;
    };
    Birth b;
    BudTie c;
    double p0 = 1.0;
    const bool have = pick(0, c, p0);
    const double pA = (have ? p0 : 1.0) * N;                        // minraw stays the cluster-0 centre (p = 1) otherwise
    if (pA < o.omegaA && have) { b.yes = true; b.type = 'A'; b.c = c; b.pval = pA; }
    else {
      BudTie cp;
      double p1 = 1.0;
      const bool havep = pick(1, cp, p1);
      const double pP = havep ? p1 : 1.0;
      if (pP < o.omegaP && havep) { b.yes = true; b.type = 'P'; b.c = cp; b.pval = pP; }
    }
    if (b.yes) b.newi = nclust_dev;
    return b;
  }

  // device side of a birth (cluster.cpp:313-347) + the new centre's k-mer record; one launch
  void launch_birth(const Birth &b) {
    const int raw = b.c.raw;
    if (b.newi >= ccap) grow_clusters(std::max(ccap * 2, b.newi + 1));
    const int ev = ev_begin(EV_BIRTH, profile_all);
    launch_apply_bud(P, s->D, d_creads_snap.p, raw, b.newi, b.c.from, s->h_reads[raw], b.c.from_reads - s->h_reads[raw],
                     s->d_ctab.p, ro_next()->cnt, s->stream);
    ev_end(ev);
    nclust_dev = b.newi + 1;
    snap_fresh = true;                                   // k_apply_bud rewrote the whole snapshot
    ri ^= 1;
    if (shard) {
      if ((int)h_upd.size() <= b.newi) h_upd.resize(b.newi + 1, 0);
      h_upd[b.newi] = 1; h_upd[b.c.from] = 1;            // (cluster.cpp:341-346: both partitions' expected reads changed)
    }
  }

  // the same, already done by k_auto_birth (and the next round already launched behind it): just book it
  void confirm_auto_birth(const Birth &b) {
    if (b.type != 'A' || b.c.raw != h_rout.p->bud.ties[0][0].raw || !spec_launched)
      throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: device and host bud decisions differ"};
    nclust_dev = b.newi + 1;
    snap_fresh = true;
    ri ^= 1;
    confirm_spec(b.newi, b.c.raw);
  }

  // host mirror of the same birth: bi_pop_raw from its partition, new Bi with the unique as only member and centre
  void record_birth(const Birth &b) {
    const int raw = b.c.raw;
    const int from = clust_of[raw];
    if (from != b.c.from) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: host/device membership diverged"};
    const double expected = b.c.comp_lam * bi[from].reads;
    Bi &bf = bi[from];
    const int slot = slot_of[raw];
    const uint32_t last = bf.raw.back();
    bf.raw[slot] = last;
    slot_of[last] = slot;
    bf.raw.pop_back();
    bf.reads -= s->h_reads[raw];
    bi.emplace_back();
    const int newi = (int)bi.size() - 1;
    Bi &nb = bi[newi];
    nb.birth_type = b.type;
    nb.birth_from = b.type == 'A' ? (uint32_t)from : 0u;              // never assigned for "P" births (cluster.cpp:331-345)
    nb.birth_pval = b.pval; nb.birth_fold = s->h_reads[raw] / expected; nb.birth_e = expected;
    nb.birth_comp = Comp{(uint32_t)b.c.comp_i, (uint32_t)raw, b.c.comp_lam, b.c.comp_ham};
    nb.raw.push_back((uint32_t)raw);
    nb.reads = s->h_reads[raw];
    nb.center = (uint32_t)raw;                                       // bi_assign_center: the only member
    slot_of[raw] = 0;
    clust_of[raw] = newi;
    if (slot == 0) push_slot0();
    if (b.type == 'A') logf(", Division (naive): Raw %d from Bi %d, pA=%.2e", raw, from, b.pval);
    else logf(", Division (prior): Raw %d, pP=%.2e", raw, b.pval);
  }

  void check_errflag(int32_t f) {
    if (f & 1) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "Lambda out-of-range error."};
    if (f & 2) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: comparison store overflow"};
    if (f & 32) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: a work list of a batch compare overflowed (the compare ran twice?)"};
    if (f & 16) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: the persistent tail's LDS mirror disagrees with the state arrays"};   // (DADA2HIP_V3_MIRROR=2 checks)
  }

  // =================================================================================================================
  // Round engine v2 (engine.h): batched multi-centre compares, device-driven rounds, the host trailing the device.
  // Used when the rounds run on the cooperative NW kernel and nothing forces the plain loop; DADA2HIP_ENGINE=classic
  // keeps the round-1 loop (the parity tests run both).
  bool use_v2 = false;
  Eng2 E2{}, E2L{};               // E2L: the argument block of chains WITHOUT the batch compare (Eng2::has_compare = 0)
  DevBuf<double> v2_lam0, v2_lam1;
  DevBuf<uint32_t> v2_ham0, v2_ham1;
  DevBuf<int32_t> v2_i1;
  DevBuf<unsigned long long> v2_smask;
  DevBuf<int32_t> v2_head, v2_blkcount, v2_dlt, v2_movers, v2_slotc, v2_sig, v2_n0d, v2_blist, v2_blistn;
  DevBuf<unsigned long long> v2_retrytot;   // NwBatch::fast_ctl [4]
  DevBuf<int32_t> v2_bretry, v2_bretryn;   // retry lists of the aligner's pointer-free pass (NwBatch::retry_list / retry_n)
  DevBuf<double> v2_lamB;
  DevBuf<uint32_t> v2_hamB;
  DevBuf<uint8_t> v2_moved;
  DevBuf<uint32_t> v2_statpart;
  DevBuf<CompBlk> v2_blk;
  DevBuf<Ctl2> v2_ctl;
  DevBuf<BudTie> v2_tiesrec;
  DevBuf<Round2Out> v2_dblk;
  DevBuf<uint16_t> v2_bcls, v2_full, v2_ord;
  DevBuf<uint2> v2_tab8;
  DevBuf<uint32_t> v2_cbits, v3_pf_cbits;   // Cache2::cbits of the two compare sets
  PinBuf<Round2Out> v2_hblk;
  DevBuf<unsigned long long> v2_trace;   // phase stamps of one traced round (DADA2HIP_V2_TRACE=<sequence number>[:<file>])
  int v2_trace_seq = -1;
  int v2_nbuf = 64, v2_depth = 2, v2_chain = SH_CHAIN;
  bool v2_debug = false;
  // persistent round tail (k3_tail, rounds3.inc.hip): rounds run back to back inside one launch
  bool v3_on = false;                 // this run's rounds go through k3_tail (else: the launch chains)
  int v3_depth = 1;                   // persistent launches kept in flight (DADA2HIP_V2_DEPTH)
  int v3_grid = 1;                    // its blocks (co-resident: at most one per CU)
  int v3_bs = 1024;                   // threads per block
  long v3_enq = 0;                    // k3_tail launches enqueued (their ordinals are 1, 2, ...)
  long v3_ord_seen = 0;               // launch ordinal of the last consumed block
  DevBuf<PSync> v3_psync;
  DevBuf<int32_t> v3_lockbuf;         // Eng2::spec_lock_buf: per block, the locks an evaluation attempt has decided (published when it stands)
  int v3_lock_stride = 0;
  bool v3_pf_lowreg = false;          // prefetch screens on the 80-register build of the screen kernel (the tail shares every CU with them)
  // ---- the next batch's compare under the tail: a second set of everything a batch compare works with ----
  bool v3_overlap = false;
  Eng2 E2P{};                         // argument block of the prefetch compare's kernels (ctl = the prefetch descriptor, own tables / lists / aligner scratch)
  SampleDev S2{};                     // the sample with the second aligner scratch
  DevBuf<Ctl2> v3_pfctl;
  DevBuf<PfSync> v3_pfsync;
  DevBuf<uint2> v3_pf_tab8;
  DevBuf<uint16_t> v3_pf_full, v3_pf_ord, v3_pf_foff;
  DevBuf<int32_t> v3_pf_blist, v3_pf_blistn, v3_pf_bretry, v3_pf_bretryn;
  DevBuf<uint32_t> v3_pf_ad;
  DevBuf<AdDesc> v3_pf_fdesc;
  long v3_pf_launched = 0;            // highest prefetch number whose chain has been sent to the second stream (numbers are 1, 2, ...)
  long v3_pf_seen = 0;                // highest prefetch number seen planned in a result block
  long v3_pf_chains = 0;              // chains sent (a chain whose gate gave up is sent again)
  bool v3_pf_gate_on = true;          // chains go out one AHEAD of their plan behind a gate kernel (else: when the plan is seen)
  hipEvent_t v3_pf_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // end of prefetch compare k: v3_pf_ev[k & 3]
  DevBuf<unsigned long long> v3_ktime;
  PinBuf<int32_t> v3_hflags;          // [0] result blocks the host has finished with, [16] ordinal of the last launch that ended
  bool v3_slot_held = false;          // the device's persistent slot (held for the run's rounds)
  int v3_slot_blocks = 0;
  long v2_enq = 0, v2_cons = 0;
  uint64_t v2_miss_launches = 0;
  struct EnqRec { int ev_screen, ev_nw; bool compare; };
  std::vector<EnqRec> v2_enqrec;      // per enqueued block (index = sequence number - 1)

  bool want_v2() const {
    if (knobs().engine_classic) return false;
    if (plain || no_auto || !rounds_use_coop()) return false;
    if (s->D.N < 2) return false;
    return true;
  }
  void v2_bind() {   // (re)build the by-value kernel argument block after any (re)allocation
    E2.P = P; E2.S = s->D;
    E2.T.lam0 = v2_lam0.p; E2.T.ham0 = v2_ham0.p; E2.T.lam1 = v2_lam1.p; E2.T.ham1 = v2_ham1.p; E2.T.i1 = v2_i1.p; E2.T.smask = v2_smask.p; E2.T.head = v2_head.p; E2.T.blk = v2_blk.p; E2.T.blk_count = v2_blkcount.p;
    E2.T.blk_cap = (int32_t)std::min<size_t>(v2_blk.n, 0x7FFFFFF0u);
    E2.C.NBUF = v2_nbuf; E2.C.bcls = v2_bcls.p; E2.C.slot_centre = v2_slotc.p;
    E2.C.tab8 = v2_tab8.p; E2.C.full = v2_full.p; E2.C.ord = v2_ord.p; E2.C.cbits = v2_cbits.p; E2.C.Npad = ((size_t)N + 31) & ~(size_t)15;
    E2.C.lamB = v2_lamB.p; E2.C.hamB = v2_hamB.p; E2.blist = v2_blist.p; E2.blist_n = v2_blistn.p; E2.bretry = v2_bretry.p; E2.bretry_n = v2_bretryn.p; E2.fast_ctl = v2_retrytot.p;
    E2.ctl = v2_ctl.p; E2.dblk = v2_dblk.p; E2.hblk = v2_hblk.p; E2.dlt = v2_dlt.p; E2.movers = v2_movers.p;
    E2.partial = d_partial.p; E2.ties0 = d_ties0.p; E2.ties1 = d_ties1.p; E2.ccap = ccap;
    E2.sig_list = v2_sig.p + 4; E2.sig_n = v2_sig.p; E2.ties_rec = v2_tiesrec.p;
    E2.greedy = o.greedy; E2.detect_singletons = o.detect_singletons;
    E2.total_reads = (double)(uint32_t)s->total_reads; E2.omegaA = o.omegaA; E2.omegaP = o.omegaP;
    E2.bp = BudParams{o.min_fold, o.omegaA, o.omegaP, o.min_hamming, o.min_abund};
    E2.sp = sp; E2.thresh = d_thresh_round.p; E2.max_shuffle = MAX_SHUFFLE;
    E2.trace = v2_trace.p; E2.trace_seq = v2_trace_seq;
    E2.moved = v2_moved.p; E2.n0d = v2_n0d.p; E2.stat_part = v2_statpart.p; E2.stat_n = v2_n0d.p + 2 * SH_LEVELS;
    E2.psync = v3_psync.p; E2.hcons = v3_hflags.p; E2.hexit = v3_hflags.p ? v3_hflags.p + 16 : nullptr; E2.ktime = v3_ktime.p;
    E2.sh_filter = 1; E2.grid_shuffle = 2048; E2.grid_pupdate = 1024;
    const Knobs &K = knobs();
    if (K.v2_filter >= 0) E2.sh_filter = K.v2_filter;
    if (K.v2_grid_shuffle > 0) E2.grid_shuffle = std::min(8192, K.v2_grid_shuffle);
    // (a wave of the shuffle pass adds up per-thread counts in 16-bit halves: fewer than 1000 uniques per thread)
    E2.grid_shuffle = std::max<int>(E2.grid_shuffle, (int)std::min<long long>(8192, (long long)N / (256ll * 1000) + 1));
    if (K.v2_grid_pupdate > 0) E2.grid_pupdate = K.v2_grid_pupdate;
    E2.mov_inline = MOV_INLINE2; E2.ring_limit = RING2;
    if (K.v2_mov_inline > 0) E2.mov_inline = std::min(MOV_INLINE2, K.v2_mov_inline);   // test knob: long mover lists
    if (K.v3_ring > 0) E2.ring_limit = std::min(RING2, K.v3_ring);                     // test knob: a host that lags
    E2.fail_ordinal = K.v3_fail_entry > 0 ? K.v3_fail_entry : 0;
    E2.grid_wait_ticks = (unsigned long long)((2.0 + (double)N / 1e6) * 1e8);   // 100 MHz ticks: 2 s + 1 s per 10^6 uniques
    E2.spec_lock_buf = v3_lockbuf.p; E2.spec_lock_stride = v3_lock_stride;
    E2.spec_eval = K.v3_spec != 0 ? 1 : 0;
    {   // the tail's LDS mirror of the per-unique facts its sweeps ask for (Eng2::mirror_on): where the uniques one block sweeps fit it
      const long long per_group = 4096ll * std::max(1, v3_grid);
      const long long per_block = ((long long)N + per_group - 1) / per_group * 4096ll;
      E2.mirror_on = (v3_on && K.v3_mirror != 0 && N < (1 << 24) && per_block <= (long long)tail_mirror_cap(s->device, v3_bs)) ? (K.v3_mirror == 2 ? 2 : 1) : 0;   // (2: test knob - every round ends with a comparison of the mirror with the arrays)
    }
    // grid barriers inside a persistent launch: XCD-hierarchical from 48 blocks on (rounds3.inc.hip::grid_sync; a small grid is
    // faster on the flat one).  (v3_grid is v3_setup's, which runs in front of every bind)
    E2.xbar = K.v3_xbar >= 0 ? K.v3_xbar : (v3_grid >= 48 ? 1 : 0);
    // (behind a call that moved many uniques the next one mostly moves some too, and a void attempt costs what a standing one
    //  saves - 10^6 uniques: tail 72.7 ms without, 71.1 / 71.7 / 72.6 / 74.5 / 76.9 ms at <= 8 / 32 / 128 / 1024 / always, profiles/r07v)
    E2.spec_max_prev = K.v3_spec_max >= 0 ? K.v3_spec_max : 16;
    E2.pf_on = v3_overlap ? 1 : 0; E2.pf_min = 2; E2.pf_plan = E2.pf_on;
    E2.pf_early = K.v3_pf_early >= 0 ? K.v3_pf_early : 4;
    E2.pf_sync = K.v3_pf_sync != 0 ? 1 : 0;
    E2.pf_ctl = v3_pfctl.p; E2.pf_blist_n = v3_pf_blistn.p; E2.pfsync = v3_pfsync.p;
    {   // how long a round waits inside the launch for a prefetch compare in flight before the launch is left (the host then
        // orders the next launch behind the compare): a few compares' worth - a compare is ~0.6 ms per 10^6 uniques
      // (100 us: a tail that waits holds half of the CUs idle beside a compare that would finish sooner on all of them; leaving
      //  costs a launch and an entry barrier, ~20-30 us.  10^6 uniques, same box: 4 ms bound 128.8 ms / tail 93.0, 100 us 128.3 /
      //  88.9, none 137 before the bitmaps - profiles/r09d, r09e)
      const double us = K.v3_pf_wait_us >= 0 ? (double)K.v3_pf_wait_us : 100.0;
      E2.pf_wait_ticks = (unsigned long long)(us * 100.0);
      // the gate of a chain enqueued ahead of its plan (k2_pf_gate): plans come every millisecond or so while rounds run
      v3_pf_gate_on = K.v3_pf_gate_us != 0;
      // (5 ms: plans come every millisecond while rounds run, and a gate that gives up only sends its chain down the slower path -
      //  the host launches it when it sees the plan.  The bound was 500 ms until the last hour of round 5: a waiting gate is a
      //  kernel at the head of a hardware queue, and when the runtime has mapped the stream of the TAIL onto the same queue -
      //  which it does, now and then, once two samples' streams exist - the tail's next launch waits for the gate that waits for
      //  the tail's plan: configs[3] on one GPU read 234 ms or 574-1 250 ms, one sample of a slow run waiting 1 024 ms = two bounds,
      //  profiles/r08k)
      E2.pf_gate_ticks = (unsigned long long)((K.v3_pf_gate_us > 0 ? (double)K.v3_pf_gate_us : 5000.0) * 100.0);
    }
    E2.has_compare = 1;
    E2.align_at_commit = v2_align_commit ? 1 : 0;
    E2L = E2; E2L.has_compare = 0;
    if (v3_overlap) {
      S2 = s->D; S2.ad_ptr = v3_pf_ad.p; S2.ad_foff = v3_pf_foff.p; S2.ad_desc = v3_pf_fdesc.p;
      E2P = E2; E2P.S = S2; E2P.ctl = v3_pfctl.p; E2P.C.tab8 = v3_pf_tab8.p; E2P.C.full = v3_pf_full.p; E2P.C.ord = v3_pf_ord.p; E2P.C.cbits = v3_pf_cbits.p;
      E2P.blist = v3_pf_blist.p; E2P.blist_n = v3_pf_blistn.p; E2P.bretry = v3_pf_bretry.p; E2P.bretry_n = v3_pf_bretryn.p; E2P.pf_on = 0; E2P.has_compare = 1;
    }
    v2_drop_graph();                 // (captured launches hold the old argument block)
  }
  void v2_alloc(int max_clust) {
    const size_t n = (size_t)N;
    v2_nbuf = 64;   // 512 cached centres, 2 bytes x N each (r02l sweep: 64 buffers beat 32 by 2.5 % at 1M uniques)
    {   // a batch buffer holds class words and the aligner's results for KB_MAX centres: 98 bytes per unique.  At most an
        // eighth of the device's memory goes to the cache (64 buffers take 6.3 GB at 10^6 uniques).  hipDeviceTotalMem, not
        // hipMemGetInfo: the latter is refused while ANOTHER host thread captures its round graph (dada2hip_run_multi).
      size_t total_b = 0;
      int dev_ = 0;
      (void)hipGetDevice(&dev_);
      if (hipDeviceTotalMem(&total_b, dev_) != hipSuccess || !total_b) { (void)hipGetLastError(); total_b = (size_t)64 << 30; }
      const size_t per_buf = (((size_t)N + 31) & ~(size_t)15) * (2 + (size_t)KB_MAX * 12);
      v2_nbuf = (int)std::max<size_t>(2, std::min<size_t>(64, total_b / 8 / std::max<size_t>(per_buf, 1)));
    }
    const Knobs &K = knobs();
    if (K.v2_nbuf > 0) v2_nbuf = std::min(64, K.v2_nbuf);   // (k2_birth keeps the slot table in LDS)
    v2_depth = K.v2_depth > 0 ? std::min(MOV_RING - 1, K.v2_depth) : 2;
    // persistent launches in flight: ONE.  A second one queued behind a launch that leaves for a prefetch compare still in flight
    // starts at once, spins its bound for the same compare and leaves again (every such launch is a chain of compare kernels that
    // find nothing to do, an entry barrier and a mirror fill), while the launch the host sends once it has seen the exit is ordered
    // behind the compare's event.  10^6 uniques, same box: depth 1 / 2 / 3 / 4 = 124.3 / 134.8 / 139.5 / 137.1 ms per pass with
    // 154 / 412 / 541 / 548 launches (profiles/r09t_sweep_cfg3_launches_in_flight.jsonl); the host tops up between blocks (run_v3)
    v3_depth = K.v2_depth > 0 ? v2_depth : 1;
    v2_chain = K.v2_chain > 0 ? std::min(SH_CHAIN, K.v2_chain) : SH_CHAIN;   // test knob: shorter shuffle chains
    v2_debug = K.v2_debug;
    hipStream_t stq = s->stream;
    v2_lam0.alloc(n); v2_ham0.alloc(n); v2_lam1.alloc(n); v2_ham1.alloc(n); v2_i1.alloc(n); v2_smask.alloc(n); v2_head.alloc(n); v2_blkcount.alloc(1);
    {
      // (4 blocks per unique: a selfConsist pass with a half-converged error matrix stores more than the converged ones, and a
      //  halt for capacity is a drained device, a 2x allocation and a copy - two of them in pass 3 of configs[2]'s loop, profiles/r09a)
      size_t cap0 = std::max<size_t>(4 * n, (size_t)1 << 16);
      if (K.node_cap > 0) cap0 = std::max<size_t>((size_t)K.node_cap, n + 16);   // test knob: forces growth
      if (v2_blk.n < cap0) v2_blk.alloc(cap0);
    }
    v2_ctl.alloc(1); v2_dblk.alloc(RING2); v2_hblk.alloc(RING2);
    v2_dlt.alloc((size_t)SH_LEVELS * ccap);
    v2_movers.alloc((size_t)std::max(MOV_RING * SH_CHAIN, SH_LEVELS) * 3 * n);
    const size_t slots = (size_t)v2_nbuf * KB_MAX;
    v2_bcls.alloc((size_t)v2_nbuf * (((size_t)N + 31) & ~(size_t)15));
    v2_lamB.alloc(slots * (((size_t)N + 31) & ~(size_t)15)); v2_hamB.alloc(slots * (((size_t)N + 31) & ~(size_t)15));
    v2_blist.alloc((size_t)2 * KB_MAX * (((size_t)N + 31) & ~(size_t)15)); v2_blistn.alloc(2 * KB_MAX);
    v2_bretry.alloc((size_t)KB_MAX * (((size_t)N + 31) & ~(size_t)15)); v2_bretryn.alloc(KB_MAX);
    D2_HIP(hipMemsetAsync(v2_blistn.p, 0, 2 * KB_MAX * 4, stq));
    D2_HIP(hipMemsetAsync(v2_bretryn.p, 0, KB_MAX * 4, stq));
    v2_retrytot.alloc(4);
    D2_HIP(hipMemsetAsync(v2_retrytot.p, 0, 32, stq));
    v2_slotc.alloc(slots); v2_tab8.alloc(NKMER); v2_cbits.alloc(KB_MAX * 32); v2_full.alloc((size_t)KB_MAX * NKMER); v2_ord.alloc((size_t)KB_MAX * s->D.LK + 64);
    v2_sig.alloc(n + 4); v2_tiesrec.alloc((size_t)2 * TIES_FULL);
    v2_moved.alloc(n); v2_n0d.alloc(2 * SH_LEVELS + 4); v2_statpart.alloc((size_t)4 * 8192);
    D2_HIP(hipMemsetAsync(v2_moved.p, 0, n, stq));
    D2_HIP(hipMemsetAsync(v2_n0d.p, 0, (2 * SH_LEVELS + 4) * 4, stq));
    D2_HIP(hipMemsetAsync(v2_sig.p, 0, 16, stq));
    D2_HIP(hipMemsetAsync(v2_blkcount.p, 0, 4, stq));
    D2_HIP(hipMemsetAsync(v2_dblk.p, 0, sizeof(Round2Out) * RING2, stq));
    D2_HIP(hipMemsetAsync(v2_dlt.p, 0, (size_t)SH_LEVELS * ccap * 4, stq));
    D2_HIP(hipMemsetAsync(v2_slotc.p, 0xFF, slots * 4, stq));
    for (int k = 0; k < RING2; k++) v2_hblk.p[k].seq = 0;
    Ctl2 c;
    memset(&c, 0, sizeof c);
    c.nclust = 1; c.centre = (int32_t)bi[0].center; c.slot = 0; c.max_clust = max_clust;
    c.n0 = N; c.low0 = N; c.need_compare = 0; c.nalign = 0; c.abuf = 0; c.stable = 1; c.bfrom = 0;
    c.pf_bbuf = -1; c.last_bbuf = -1; c.prev_bbuf = -1;
    for (int k = 0; k < KB_MAX; k++) c.acentre[k] = -1;
    for (int k = 0; k < KB_MAX; k++) c.bcentre[k] = -1;
    D2_HIP(hipMemcpyAsync(v2_ctl.p, &c, sizeof c, hipMemcpyHostToDevice, stq));
    D2_HIP(hipStreamSynchronize(stq));   // `c` is a local
    v2_trace_seq = -1;
    if (K.v2_trace_on) {
      v2_trace_seq = K.v2_trace_seq;
      v2_trace.alloc((size_t)TRACE_KERNELS * TRACE_BLOCKS * 8);
      D2_HIP(hipMemsetAsync(v2_trace.p, 0, (size_t)TRACE_KERNELS * TRACE_BLOCKS * 64, stq));
    }
    v2_enq = v2_cons = 0;
    v2_next_full = 0;
    v2_lite_on = true;
    if (K.v2_lite >= 0) v2_lite_on = K.v2_lite != 0;
    // one alignment per wave (band windows > 65 cells: long reads): a round's own list fills the device, so its pairs are
    // aligned when the round commits and nothing is aligned in vain (engine.h, Eng2::align_at_commit)
    v2_align_commit = nw_ad_apw(s->D, ap) == 1;
    if (K.v2_align >= 0) v2_align_commit = K.v2_align == 1;
    if (v2_align_commit) v2_lite_on = false;                   // (every chain carries the aligner's launches)
    v2_plain_rounds = 0;
    v2_miss_launches = 0;
    v2_enqrec.clear();
    v3_setup(stq);
    v2_bind();
  }

  // ---- persistent round tail -----------------------------------------------------------------------------------------------
  // DADA2HIP_V2_TAIL=chain keeps the launch chains; DADA2HIP_V3_GRID=n forces the number of blocks (tests: several blocks on a
  // small sample)
  void v3_setup(hipStream_t stq) {
    const Knobs &K = knobs();
    v3_on = !K.v2_tail_chain;
    if (K.v2_graph) v3_on = false;                                      // (hipGraph replay is a property of the chains)
    if (v2_trace_seq >= 0 || K.v2_trace_on) v3_on = false;              // (the phase trace stamps the chains' kernels)

    v3_grid = tail_grid(N, s->device);
    if (K.v3_grid > 0) v3_grid = std::min(K.v3_grid, tail_grid(1 << 30, s->device));
    // refuse up front a grid the device cannot hold at once (the entry barrier would time out): the launch chains serve the run
    // The next batch's compare under the tail: on by default where batches are aligned ahead (not the long reads,
    // which align at commit time) and the cache is deep enough to give a prefetch a buffer of its own.  The tail then runs on
    // 512-thread blocks: half of every CU's registers stay free for the compare's kernels.
    v3_overlap = v3_on && !v2_align_commit && v2_nbuf >= 4 && K.v3_overlap != 0 && N >= 2 && (active_runs(s->device).load() <= 1 || K.v3_overlap == 1);
    // (1024-thread blocks: under the overlap the tail takes its CUs WHOLE - ceil(N / 4096) of them for a sample of up to half the
    //  device, half of the device beyond that (below) - and the compares' kernels have the others to themselves.  Round 5 put
    //  512-thread blocks beside the compares on every CU up to 5 10^5 uniques: the same time at 10^5 plain uniques (12.9 ms both),
    //  and 38.3 against 33.9 ms on the deep 10^5 workload, profiles/r09h; DADA2HIP_V3_BLOCK=512 keeps that form for the tests)
    v3_bs = K.v3_block == 512 ? 512 : 1024;
    {
      // A sample whose 512-thread blocks would sit on more than half of the CUs: the tail takes HALF of the CUs WHOLE instead
      // (1024-thread blocks at 128 registers fill a CU's register file) and the compare's kernels have the other half to
      // themselves at their full occupancy, rather than sharing every CU with a tail block.  10^6 uniques, one box, median of
      // nine with the XCD-hierarchical barrier: 128 x 1024 135.0-136.5 ms, 245 x 512 136.5-137.4, 96 x 1024 139.3-141.6
      // (profiles/r08b); with the flat barrier 96 x 1024 led (137.8 vs 144.5 for 245 x 512: r07s, r07t); 64 / 80 CUs are too few
      // for the tail (150 / 145 ms), 160 / 192 leave the compare too little (149 / 182 ms).
      const int ncu = tail_grid(1 << 30, s->device);
      if (v3_overlap && K.v3_block == 0 && K.v3_grid == 0 && v3_grid > ncu / 2) {
        v3_bs = 1024;
        v3_grid = std::max(1, std::min((N + 8191) / 8192, ncu / 2));
      }
    }
    // (cap: blocks the device can hold at once; 0 = the kernel cannot be resident at all on this part, -1 = the query failed and
    //  the bounded entry barrier is the only guard)
    if (v3_on) { const int cap = tail_resident_max(s->device, v3_bs); if (cap == 0 || (cap > 0 && v3_grid > cap)) v3_on = false; }
    if (!v3_on) v3_overlap = false;
    // prefetch screens: the 80-register build where a tail block sits on EVERY CU beside them (three waves per SIMD fit into the
    // registers the tail leaves, where two of the wide build do); the full build where the tail has taken half of the CUs whole
    // and the compares have the others to themselves - there the narrow build only spills (29 VGPRs) and moves 1.8x the bytes
    // (same box, 10^6 uniques: tail 102.7 -> 97.0 ms, pass 143.0 -> 138.7 ms, profiles/r09a_sweep_cfg3_lowreg.jsonl)
    v3_pf_lowreg = K.v3_pf_lowreg >= 0 ? K.v3_pf_lowreg != 0 : (v3_overlap && v3_bs == 512);
    v3_lock_stride = 0;
    if (v3_on) {
      // uniques one block sweeps: groups of 4096 (ShufLds::U x BS), dealt wave by wave over the blocks (sweep_unique)
      const long long per_group = 4096ll * v3_grid;
      v3_lock_stride = (int)(((long long)N + per_group - 1) / per_group * 4096ll);
      v3_lockbuf.alloc((size_t)v3_lock_stride * (size_t)v3_grid);
    }
    v3_pf_launched = 0; v3_pf_seen = 0; v3_pf_chains = 0;
    if (v3_overlap) {
      if (!s->cmp) D2_HIP(hipStreamCreateWithFlags(&s->cmp, hipStreamNonBlocking));
      for (auto &e : v3_pf_ev) if (!e) D2_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      const size_t npad = ((size_t)N + 31) & ~(size_t)15;
      v3_pfctl.alloc(1); v3_pfsync.alloc(1); v3_pf_tab8.alloc(NKMER); v3_pf_cbits.alloc(KB_MAX * 32); v3_pf_full.alloc((size_t)KB_MAX * NKMER);
      v3_pf_ord.alloc((size_t)KB_MAX * s->D.LK + 64); v3_pf_blist.alloc((size_t)2 * KB_MAX * npad); v3_pf_blistn.alloc(2 * KB_MAX);
      v3_pf_bretry.alloc((size_t)KB_MAX * npad); v3_pf_bretryn.alloc(KB_MAX);
      D2_HIP(hipMemsetAsync(v3_pf_bretryn.p, 0, KB_MAX * 4, stq));
      // the second aligner scratch: pointer ring + the factor-offset rows of k_ad_product, sized as the first (ensure_ad_ring)
      v3_pf_ad.alloc((size_t)s->D.ad_waves * s->D.ad_wpw);
      v3_pf_foff.alloc((size_t)s->D.ad_fcap * s->D.ad_fstride); v3_pf_fdesc.alloc((size_t)s->D.ad_fcap);
      D2_HIP(hipMemsetAsync(v3_pf_fdesc.p, 0xFF, (size_t)s->D.ad_fcap * sizeof(AdDesc), stq));
      D2_HIP(hipMemsetAsync(v3_pfctl.p, 0, sizeof(Ctl2), stq));
      D2_HIP(hipMemsetAsync(v3_pfsync.p, 0, sizeof(PfSync), stq));
      D2_HIP(hipMemsetAsync(v3_pf_blistn.p, 0, 2 * KB_MAX * 4, stq));
    }
    v3_psync.alloc(1); v3_hflags.alloc(64); v3_ktime.alloc(KT_N);   // ([32] = "run over" for the prefetch gates, [40..47] = their results)
    D2_HIP(hipMemsetAsync(v3_psync.p, 0, sizeof(PSync), stq));
    D2_HIP(hipMemsetAsync(v3_ktime.p, 0, KT_N * 8, stq));
    for (int k = 0; k < 64; k++) v3_hflags.p[k] = 0;
    v3_enq = 0; v3_ord_seen = 0;
  }
  // The device's persistent slot, taken for the rounds only (run_v3 releases it).  Threads of this process take turns - the
  // rounds of one sample fill the device, the uploads, round 0 and final passes of the others overlap them; another PROCESS on
  // the same GPU that holds the slot sends this run to the launch chains.
  bool v3_acquire() {
    if (v3_slot_held) return true;
    v3_slot_blocks = v3_grid;
    persistent_slot(s->device).acquire(std::max(1, knobs().v3_slots), v3_slot_blocks, tail_grid(1 << 30, s->device));
    if (!persistent_file(s->device).try_acquire(s->device)) { persistent_slot(s->device).release(v3_slot_blocks); return false; }
    v3_slot_held = true;
    return true;
  }
  void v3_release() {
    if (v3_slot_held) { persistent_file(s->device).release(); persistent_slot(s->device).release(v3_slot_blocks); v3_slot_held = false; }
  }
  // halt whatever this run still has queued or running on the device and wait for it (errors ignored: this runs while another
  // error unwinds).  A running k3_tail sees the halt at its next round boundary at the latest when the host stops consuming
  // (ring limit); queued launches return at their first instruction.
  void v3_quiesce() {
    if (!v2_ctl.p || !s) return;
    v3_hflags.p[24] = 1;
    *(volatile int32_t *)(v3_hflags.p + 32) = 1;   // (a prefetch gate still waiting for a plan gives up)
    (void)hipMemcpyAsync(&v2_ctl.p->state, v3_hflags.p + 24, 4, hipMemcpyHostToDevice, s->side);
    (void)hipStreamSynchronize(s->side);
    (void)hipStreamSynchronize(s->stream);
    if (s->cmp) (void)hipStreamSynchronize(s->cmp);
    (void)hipGetLastError();
    v3_pf_release();
  }
  // the second set of compare buffers goes back to the allocation cache when the rounds are over (nothing of this run is left on
  // the second stream: the callers have waited for it)
  void v3_pf_release() {
    v3_pf_ad.free(); v3_pf_foff.free(); v3_pf_fdesc.free(); v3_pf_blist.free(); v3_pf_bretry.free();
  }
  // the entry barrier of a persistent launch failed: clear the failure, give the slot back, continue on the launch chains
  void v3_fallback() {
    sync_spin(s->stream);
    if (v3_overlap) {                                          // (the launch chains plan no prefetches)
      *(volatile int32_t *)(v3_hflags.p + 32) = 1;             // (the gate of the chain sent ahead gives up)
      sync_spin(s->cmp);
      v3_pf_totals();
      v3_overlap = false;
      v2_bind();
      v3_pf_release();
    }
    D2_HIP(hipMemsetAsync(v3_psync.p, 0, sizeof(PSync), s->stream));
    launch2_resume(E2, s->stream, /*keep_list=*/true, /*compare_done=*/true);   // (every launch but the first has its compare in front of it)
    v3_release();
    v3_on = false;
    st.tail_fallbacks++;
    v2_enq = v2_cons;
    v2_enqrec.assign((size_t)v2_cons, EnqRec{-1, -1, false});
    v2_next_full = 0;                                          // (whether the coming centre is cached is not known here: full chains)
    if (knobs().v2_summary) fprintf(stderr, "[v3] entry barrier failed after %ld blocks: continuing on the launch chains\n", v2_cons);
  }
  bool v3_block_ready() const { return *(volatile int32_t *)&v2_hblk.p[v2_cons % RING2].seq == (int32_t)(v2_cons + 1); }
  long v3_ended() const { return (long)*(volatile int32_t *)(v3_hflags.p + 16); }
  // one super-chain: the launches of a batch compare (they find nothing to do unless the round in front of the device needs
  // one: Ctl2::nbatch / nalign), then the persistent tail, which runs rounds until the next compare is due
  std::vector<EnqRec> v3_rec;          // per k3_tail launch (index = ordinal - 1): its compare's profile events
  void v3_enqueue(bool first) {
    const auto t_enq = clk::now();
    hipStream_t stq = s->stream;
    EnqRec rec{-1, -1, !first};
    if (!first) {
      rec.ev_screen = ev_begin(EV_SCREEN, profile_all, /*spec=*/true);
      launch2_screen_multi(E2, stq);
      ev_end(rec.ev_screen);
      launch2_batch_lists(E2, stq);
      const NwBatch nb{&v2_ctl.p->nalign, v2_blistn.p, v2_blist.p, v2_ctl.p->acentre, &v2_ctl.p->abuf, E2.C.Npad, v2_bretry.p, v2_bretryn.p, v2_retrytot.p};
      launch_gapless_batch(s->D, nb, ap, s->d_err.p, v2_lamB.p, v2_hamB.p, &v2_ctl.p->state, stq);
      rec.ev_nw = ev_begin(EV_NW, profile_all, /*spec=*/true);
      launch_nw_ad(s->D, -1, nullptr, nullptr, nullptr, 0, nullptr, nullptr, ap, s->d_err.p, v2_lamB.p, v2_hamB.p, nullptr, 0, 0, nullptr, stq,
                   &v2_ctl.p->state, &nb);
      ev_end(rec.ev_nw);
    }
    const int ev = ev_begin(EV_TAIL, profile_all);
    Eng2 Ek = E2;
    if (!profile_all) Ek.ktime = nullptr;
    // (the rule of v3_setup - no prefetching beside another sample of this process - again for every launch: the other sample may
    //  have arrived since.  Its kernels share the runtime's hardware queues with this run's second stream, and a tail spinning
    //  for a compare queued behind them holds the device's persistent slot for nothing: configs[3] on one GPU 233 -> 603 ms
    //  when the first sample of each wave happened to start alone, profiles/r08g)
    Ek.pf_plan = (v3_overlap && (active_runs(s->device).load() <= 1 || knobs().v3_overlap == 1)) ? 1 : 0;
    launch3_tail(Ek, v3_grid, v3_bs, first, (int)(v3_enq + 1), s->h_reads[bi[0].center], stq);
    ev_end(ev);
    v3_rec.push_back(rec);
    v3_enq++;
    st.ms_enqueue += ms_since(t_enq);
  }

  // Prefetch compare number k (the tail planned it: Ctl2::pf_seq reached k in the block just read) on the second stream: tables of
  // the batch, screen, work lists, gapless pairs, aligner (+ product), completion word.  Everything it reads and writes is its
  // own (descriptor, tables, lists, aligner scratch, the batch buffer's rows) or constant (the sample) - except the greedy locks
  // the screen looks at, which only ever make it classify more pairs than the commit will use.
  void v3_pf_enqueue(long k) {
    const auto t_enq = clk::now();
    hipStream_t st2 = s->cmp;
    Ctl2 *pc = v3_pfctl.p;
    // the gate: waits for plan k (at once when the chain is sent for a plan already seen), opens or halts the descriptor
    *(volatile int32_t *)(v3_hflags.p + 40 + (k & 7)) = 0;
    launch2_pf_gate(E2P, (int)k, v3_hflags.p + 32, v3_hflags.p + 40 + (k & 7), st2);
    launch2_pf_tables(E2P, st2);
    int ev = ev_begin(EV_PF_SCREEN, profile_all, false, false, st2);
    launch2_screen_multi(E2P, st2, /*beside_tail=*/v3_pf_lowreg);
    ev_end(ev);
    launch2_batch_lists(E2P, st2);
    const NwBatch nb{&pc->nalign, v3_pf_blistn.p, v3_pf_blist.p, pc->acentre, &pc->abuf, E2.C.Npad, v3_pf_bretry.p, v3_pf_bretryn.p, v2_retrytot.p};
    launch_gapless_batch(S2, nb, ap, s->d_err.p, v2_lamB.p, v2_hamB.p, &pc->state, st2);
    ev = ev_begin(EV_PF_NW, profile_all, false, false, st2);
    launch_nw_ad(S2, -1, nullptr, nullptr, nullptr, 0, nullptr, nullptr, ap, s->d_err.p, v2_lamB.p, v2_hamB.p, nullptr, 0, 0, nullptr, st2,
                 &pc->state, &nb);
    ev_end(ev);
    launch2_pf_done(E2P, st2);
    D2_HIP(hipEventRecord(v3_pf_ev[k & 3], st2));
    v3_pf_launched = std::max(v3_pf_launched, k);
    v3_pf_chains++;
    st.ms_enqueue += ms_since(t_enq);
  }
  int v3_pf_gate_result(long k) const { return (int)*(volatile int32_t *)(v3_hflags.p + 40 + (k & 7)); }   // 0: waiting / not run yet, 1: passed, 2: gave up
  int32_t v3_pf_stat[4] = {0, 0, 0, 0};   // Ctl2::pf_hits / pf_spins / pf_exits / pf_centres as of the last block read
  void v3_pf_totals() {
    PfSync ps;
    D2_HIP(hipMemcpy(&ps, v3_pfsync.p, sizeof ps, hipMemcpyDeviceToHost));
    st.nnw_run += ps.nnw; st.ngapless_run += ps.ngapless;      // pairs the prefetch compares' aligner launches worked through
    st.pf_compares = (uint64_t)v3_pf_seen;
    st.pf_hits = (uint64_t)v3_pf_stat[0]; st.pf_waits = (uint64_t)v3_pf_stat[1]; st.pf_exits = (uint64_t)v3_pf_stat[2]; st.pf_centres = (uint64_t)v3_pf_stat[3];
    st.overlap_on = 1;
  }
  // what a published block says about the prefetch lane: a new plan goes to the second stream at once (before the block's moves
  // are replayed), and a round that has to wait for a compare still in flight gets the next persistent launch ordered behind it
  void v3_pf_serve(const Round2Out &b) {
    if (!v3_overlap) return;
    for (int k = 0; k < 4; k++) v3_pf_stat[k] = b.pf_stat[k];
    v3_pf_seen = std::max(v3_pf_seen, (long)b.pf_seq);
    // a plan whose chain has not been sent (no chain ahead), or whose chain's gate gave up before the plan came: send it (again)
    if (v3_pf_seen >= 1) {
      if (v3_pf_launched < v3_pf_seen) v3_pf_enqueue(v3_pf_seen);
      else if (v3_pf_gate_result(v3_pf_seen) == 2) v3_pf_enqueue(v3_pf_seen);
    }
    // ONE chain ahead: the next plan's chain goes out once the chain in front of it has passed its gate (two gates in the
    // stream would wait for each other: the second plan cannot be made before the first compare is done)
    // (... and not while another sample of the process is active: no plan will come - Eng2::pf_plan)
    if (v3_pf_gate_on && (active_runs(s->device).load() <= 1 || knobs().v3_overlap == 1) && v3_pf_launched == v3_pf_seen &&
        (v3_pf_seen == 0 || v3_pf_gate_result(v3_pf_seen) == 1))
      v3_pf_enqueue(v3_pf_seen + 1);
    if (b.pf_wait > 0 && (long)b.pf_wait <= v3_pf_launched) D2_HIP(hipStreamWaitEvent(s->stream, v3_pf_ev[b.pf_wait & 3], 0));
  }

  // (a fixed number per call: while the device is halted every launch ends at once, and "fewer than depth in flight" stays true
  //  however many are sent)
  bool v3_dev_halted = false;          // the block in the host's hands halted the device: nothing to keep fed until it is answered

  // ---- the replay lane (persistent tail) ---------------------------------------------------------------------------------------
  // The host's mirror of the partitions (member lists in the reference's order: b_bud's ties, the final clustering) trails the device
  // through the published moves and births.  That replay is 45-55 ms of a 10^6-unique pass and used to run between the waits of
  // the thread that also feeds the device: a persistent launch that left (a compare due) found nobody to send the next one until
  // the block in hand was replayed - with ONE launch in flight that is device time.  A block that asks nothing of the host (birth
  // applied on the device, movers inline) is now copied out of the ring and handed to a second host thread, which replays it,
  // re-derives the decision from the published candidates (decide_bud: must equal the device's) and books the birth, in block
  // order; the boundary thread drains the lane before it touches the mirror itself (a halt that wants the host's decision, a
  // paused block, the end of the rounds).  Off when the caller wants the verbose log (its callback runs on the calling thread).
  struct ReplayLane {
    std::thread th;
    std::mutex mu;                                       // guards q and err
    std::deque<std::unique_ptr<unsigned char[]>> q;
    std::atomic<long> pushed{0}, done{0};                // blocks handed over / worked through (the lane polls: a futex wake per
    std::atomic<bool> stop{false};                       //  block on the thread that feeds the device cost 2-3 ms of a pass)
    std::exception_ptr err;
    std::thread::id tid;
  } lane;
  bool lane_on = false;
  bool lane_is_me() const { return lane_on && std::this_thread::get_id() == lane.tid; }
  void lane_start() {
    lane.stop.store(false); lane.pushed.store(0); lane.done.store(0); lane.err = nullptr;
    const bool polite = wait_blocks();
    std::unique_lock<std::mutex> hold(lane.mu);                    // (the thread's first look at the queue waits until tid is written)
    lane.th = std::thread([this, polite]() {
      (void)hipSetDevice(s->device);
      for (unsigned idle = 0;;) {
        if (lane.done.load(std::memory_order_relaxed) == lane.pushed.load(std::memory_order_acquire)) {
          if (lane.stop.load(std::memory_order_acquire)) return;
          cpu_relax();
          if (polite || ++idle > 200000u) { struct timespec ts{0, 20000}; nanosleep(&ts, nullptr); }   // (a lane nobody feeds stops burning its core)
          continue;
        }
        idle = 0;
        std::unique_ptr<unsigned char[]> job;
        bool failed;
        {
          std::lock_guard<std::mutex> lk(lane.mu);
          job = std::move(lane.q.front());
          lane.q.pop_front();
          failed = (bool)lane.err;
        }
        try {
          if (!failed) {
            const Round2Out &b = *(const Round2Out *)job.get();
            v2_replay(b, 0);
            Birth bb = decide_bud(b.bud);
            if (!bb.yes || bb.type != 'A' || bb.c.raw != b.bud.ties[0][0].raw)
              throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: device and host bud decisions differ"};
            record_birth(bb);
          }
        } catch (...) {
          std::lock_guard<std::mutex> lk(lane.mu);
          if (!lane.err) lane.err = std::current_exception();
        }
        lane.done.fetch_add(1, std::memory_order_release);
      }
    });
    lane.tid = lane.th.get_id();
    lane_on = true;
  }
  // a block the host has nothing to answer: its used part copied out of the ring (header, candidates, inline movers)
  void lane_push(const Round2Out &b, int tot) {
    const size_t bytes = offsetof(Round2Out, mov) + (size_t)3 * (size_t)tot * 4;
    std::unique_ptr<unsigned char[]> job(new unsigned char[(bytes + 15) & ~(size_t)15]);
    memcpy(job.get(), &b, bytes);
    {
      std::lock_guard<std::mutex> lk(lane.mu);
      lane.q.push_back(std::move(job));
    }
    lane.pushed.fetch_add(1, std::memory_order_release);
  }
  // everything handed over has been replayed (its exception, if it raised one, surfaces here)
  void lane_drain() {
    if (!lane_on) return;
    while (lane.done.load(std::memory_order_acquire) != lane.pushed.load(std::memory_order_relaxed)) cpu_relax();
    std::lock_guard<std::mutex> lk(lane.mu);
    if (lane.err) { std::exception_ptr e = lane.err; lane.err = nullptr; std::rethrow_exception(e); }
  }
  bool lane_failed() { std::lock_guard<std::mutex> lk(lane.mu); return (bool)lane.err; }
  void lane_stop() noexcept {
    if (!lane.th.joinable()) { lane_on = false; return; }
    lane.stop.store(true, std::memory_order_release);
    lane.th.join();                                      // (it works through what is queued first: nothing it touches goes away before)
    { std::lock_guard<std::mutex> lk(lane.mu); lane.q.clear(); }
    lane_on = false;
  }
  void v3_topup() {
    if (v3_dev_halted) return;
    const long need = (long)v3_depth - (v3_enq - v3_ended());
    for (long k = 0; k < need; k++) v3_enqueue(false);
  }
  // Wait for the next result block, keeping v2_depth super-chains queued meanwhile.  The device reports the end of every launch
  // through a word of its own (v3_ended), which can reach the host a moment after the last block of that launch: a stream that
  // has gone idle without a block in sight is therefore no error as long as a further launch can still be sent.
  const Round2Out &v3_wait_block() {
    const int ring = (int)(v2_cons % RING2);
    const int32_t want = (int32_t)(v2_cons + 1);
    volatile int32_t *seqp = &v2_hblk.p[ring].seq;
    const auto tw = clk::now();
    const bool polite = wait_blocks();
    int burst = 0;
    for (unsigned spins = 0; *seqp != want; spins++) {
      if (v3_enq - v3_ended() < v3_depth && burst < 256) { v3_enqueue(false); burst++; continue; }   // (bounded: a device that ends every launch at once without a block is an error)
      cpu_relax();
      if (polite && spins > 64) { struct timespec ts{0, 20000}; nanosleep(&ts, nullptr); }
      if ((spins & 0xFFFF) == 0xFFFF || (polite && (spins & 0xFF) == 0xFF)) {
        hipError_t e = hipStreamQuery(s->stream);
        if (e != hipSuccess && e != hipErrorNotReady)
          throw d2::DeviceError{DADA2HIP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(e) + " (round result)"};
        if (e == hipSuccess && *seqp != want) {
          // everything queued has run (its writes are visible now) and the block is not there: every launch found the device
          // halted or the ring full and said so - send another one - or a launch ended without saying so (a barrier timed out)
          // ... in which case a barrier gave up.  The ENTRY barrier (not all blocks of the launch became resident: another tenant
          // holds CUs) has changed nothing - the run goes on on the launch chains (run_v3 catches this)
          if (v3_ended() < v3_enq || burst >= 2) {
            PSync ps;
            D2_HIP(hipMemcpy(&ps, v3_psync.p, sizeof ps, hipMemcpyDeviceToHost));
            if (ps.fail == 1u) throw TailEntryFailed{};
            if (ps.fail != 0u || v3_ended() < v3_enq || burst >= 512)
              throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: the persistent round tail ended without publishing its result (a grid barrier timed out?)"};
          }
          v3_enqueue(false); burst++; continue;
        }
        if (ms_since(tw) > wait_timeout_s() * 1e3)
          throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: timed out waiting for the round result"};
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    v2_cons++;
    st.ms_wait_device += ms_since(tw);
    return v2_hblk.p[ring];
  }

  // run_dada's loop (Rmain.cpp:312-331) with the rounds inside persistent launches: the host keeps v2_depth super-chains queued,
  // trails the device through the published result blocks exactly as with the launch chains, and answers the same halts
  void run_v3(int max_clust) {
    // the device's persistent slot is held for the rounds only.  An exception (abort hook, time-out, internal error) can leave
    // launches of this run queued or running: they are halted and waited for BEFORE the slot goes to the next run, which would
    // otherwise put its own persistent kernel beside a dying one
    struct SlotGuard { Run *r; ~SlotGuard() { r->lane_stop(); if (r->v3_running) r->v3_quiesce(); r->v3_release(); r->v3_running = false; } } slot_guard{this};
    v3_running = true;
    if (knobs().v3_lane != 0 && !(o.verbose && hooks && hooks->log) && !v2_debug) lane_start();
    auto t0 = clk::now();
    st.nstored = (uint64_t)N;                                  // round 0 keeps every comparison (E_minmax starts at -999)
    v3_rec.clear();
    if (v3_overlap && v3_pf_gate_on && (active_runs(s->device).load() <= 1 || knobs().v3_overlap == 1)) v3_pf_enqueue(1);         // the first prefetch compare's chain waits at its gate on the second stream
    v3_enqueue(true);                                          // b_p_update after round 0 + the first b_bud (+ the rounds that follow)
    bool done = false;
    long n_halt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_pause = 0, n_blocks = 0;
    while (!done) {
      // keep the device fed: super-chains in flight = enqueued - ended (the device reports the ordinal of every launch that
      // ends, whether it ran rounds or found the device halted); none is added while a result block waits to be consumed
      // (round 6: ALSO while blocks wait to be consumed - a host that trails the device by a few blocks never entered the wait that
      //  used to do the topping up, and the device then sat idle behind its last queued launch: 80 gaps of ~0.5 ms between a
      //  k3_tail and the next compare's screen over six passes at 10^6 uniques, profiles/r09k_cfg3_summary.md.  A launch queued
      //  behind a halt the host has not read yet ends at once without touching anything.)
      for (int burst = 0; !v3_dev_halted && v3_enq - v3_ended() < v3_depth && burst < 4; burst++) v3_enqueue(false);
      const long seq = v2_cons + 1;
      const Round2Out *bp = nullptr;
      try { bp = &v3_wait_block(); }
      catch (const TailEntryFailed &) {
        // every queued launch has ended and none of them has touched the state since the last consumed block: hand the rounds
        // to the launch chains from exactly here
        lane_drain();                                          // (the mirror is the boundary thread's again)
        lane_stop();
        v3_running = false;
        v3_fallback();
        st.ms_bookkeep += ms_since(t0);
        run_v2(max_clust, /*resume=*/true, /*first=*/v2_cons == 0);
        return;
      }
      const Round2Out &b = *bp;
      n_blocks++;
      v3_pf_serve(b);
      if (hooks && hooks->should_abort && hooks->should_abort(hooks->user))
        throw RuntimeErr{DADA2HIP_ERR_ABORTED, "dada2hip: aborted by caller"};
      if (v2_debug)
        fprintf(stderr, "[v3] blk %ld launch %d halt %d paused %d nclust %d nsh %d cnt %d %d %d %d nbatch %d slot %d birth %d found %d nties %d p %.3e blk %d\n", seq,
                b.kord, b.halt, b.paused, b.nclust, b.nsh, b.cnt[0], b.cnt[1], b.cnt[2], b.cnt[3], b.nbatch, b.slot, b.birth_applied, b.bud.found[0],
                b.bud.nties[0], b.bud.best_p[0], b.blk_count);
      v3_dev_halted = b.halt != H2_NONE;                       // (a paused block is resumed inside v2_replay, before its moves are replayed)
      int tot_mov = 0;
      for (int l = 0; l < b.nsh; l++) tot_mov += b.cnt[l];
      // the replay lane takes the block when the host has nothing to answer (the lane raises what the block's flags or its
      // decision would have raised here; a lane that has failed hands everything back so that the error surfaces at once)
      const bool to_lane = lane_on && b.halt == H2_NONE && b.paused == 0 && tot_mov <= E2.mov_inline && b.birth_applied != 0 && !lane_failed();
      if (to_lane) lane_push(b, tot_mov);
      else { lane_drain(); v2_replay(b, seq, /*resume_after_fetch=*/b.paused != 0 && b.halt == H2_NONE); }
      n_halt[b.halt & 7]++;
      if (b.kord > v3_ord_seen) {                              // first block of its launch: the compare in front of that launch served it
        v3_ord_seen = b.kord;
        const EnqRec &rec = v3_rec[(size_t)b.kord - 1];
        if (rec.compare && b.pad0[1] + b.pad0[2] > 0) {        // (the aligner launches had work)
          if (rec.ev_nw >= 0) evs[rec.ev_nw].ok = 1;
          st.nnw_run += (uint64_t)b.pad0[1]; st.ngapless_run += (uint64_t)b.pad0[2];
        }
        if (rec.compare && b.nbatch > 0) {                     // (a batch screen ran: a cache miss)
          v2_miss_launches++;
          if (rec.ev_screen >= 0) evs[rec.ev_screen].ok = 1;
        }
      }
      switch (b.halt) {
        case H2_NONE: {                                        // birth applied on the device: book it
          if (to_lane) {                                       // (the lane books it behind the block's moves)
            nclust_dev = b.nclust;
            st.ncompare += (uint64_t)N;
            break;
          }
          Birth bb = decide_bud(b.bud);
          if (!bb.yes || bb.type != 'A' || bb.c.raw != b.bud.ties[0][0].raw)
            throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: device and host bud decisions differ"};
          logf("\nNew Cluster C%i:", bb.newi);
          nclust_dev = bb.newi + 1;
          record_birth(bb);
          st.ncompare += (uint64_t)N;
          if (b.paused) n_pause++;                             // (its mover lists did not fit the block: v2_replay fetched them and resumed the device)
          break;
        }
        case H2_NO_BIRTH: {
          Birth bb = decide_bud(b.bud);
          if (bb.yes) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: device and host bud decisions differ"};
          done = true;
          break;
        }
        case H2_MAXCLUST: done = true; break;
        case H2_HOST_DECIDE:
        case H2_CAPACITY: {
          // the launches already queued behind the halt end at once (they neither touch state nor publish): the host's birth
          // is simply queued behind them, no drain needed unless buffers have to grow
          if (b.halt == H2_CAPACITY) v2_grow(b);
          Birth bb = decide_bud(b.bud);
          if (!bb.yes) { done = true; break; }
          if (bb.newi + 2 > ccap) v2_grow(b);
          logf("\nNew Cluster C%i:", bb.newi);
          launch2_host_birth(E2, bb.c.raw, bb.c.from, s->stream);
          nclust_dev = bb.newi + 1;
          record_birth(bb);
          st.ncompare += (uint64_t)N;
          break;
        }
        case H2_FAIL: throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: a grid barrier of the persistent round tail timed out (blocks not co-resident?)"};
        default: throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: unexpected halt code from the persistent round tail"};
      }
      v3_dev_halted = false;                                   // (answered: host birth / resume are in the stream)
      *(volatile int32_t *)v3_hflags.p = (int32_t)v2_cons;     // the device may reuse the ring slots of everything consumed
    }
    lane_drain();
    lane_stop();
    v3_running = false;
    sync_spin(s->stream);                                      // launches queued behind the final halt
    if (v3_overlap) {
      *(volatile int32_t *)(v3_hflags.p + 32) = 1;             // the gate of the chain sent ahead of a plan that never came gives up
      sync_spin(s->cmp);                                       // a last prefetch compare may still be running
      v3_pf_totals();
      v3_pf_release();
    }
    if (profile_all && v3_ktime.p) {                           // phase clocks of block 0 (100 MHz): what the tail's time went into
      unsigned long long kt[KT_N];
      D2_HIP(hipMemcpy(kt, v3_ktime.p, sizeof kt, hipMemcpyDeviceToHost));
      const double ms = 1e3 / 1e8;
      st.tail_ms_shuffle0 = kt[KT_S0] * ms; st.tail_ms_shuffle_more = kt[KT_SL] * ms; st.tail_ms_pupdate = kt[KT_P] * ms;
      st.tail_ms_barriers = (kt[KT_S0_BAR] + kt[KT_SL_BAR] + kt[KT_P_BAR]) * ms; st.tail_ms_birth = kt[KT_BIRTH] * ms;
      st.tail_ms_publish = kt[KT_PUBLISH] * ms; st.tail_ms_entry = kt[KT_LAUNCH] * ms;
      st.tail_levels = kt[KT_LEVELS]; st.tail_ms_release = kt[KT_RELEASE] * ms;
      st.tail_ms_pf_wait = kt[KT_PFWAIT] * ms; st.tail_ms_pf_plan = kt[KT_PLAN] * ms;
      if (knobs().v2_summary) {
        fprintf(stderr, "[v3] block-0 sub-phase ms (kid: 1 commit+shuffle0, 2 later shuffles, 5 p-update, 6 serial end):");
        for (int kid : {1, 2, 5, 6}) { fprintf(stderr, "  kid%d:", kid); for (int ph = 1; ph < 8; ph++) fprintf(stderr, " %.2f", kt[KT_SUB + 8 * kid + ph] * ms); }
        fprintf(stderr, "\n");
      }
    }
    st.tail_xcd_barrier = (E2.xbar && v3_grid > 1) ? 1u : 0u;
    st.tail_mirror = (v3_enq > 0 && E2.mirror_on) ? 1u : 0u;
    st.tail_launches = (uint64_t)v3_enq; st.tail_pauses = (uint64_t)n_pause; st.tail_blocks = (uint32_t)v3_grid; st.tail_threads = (uint32_t)v3_bs;
    if (knobs().v2_summary && v3_overlap)
      fprintf(stderr, "[v3] overlap: prefetch compares %ld in %ld chains (centres %d)  rounds served from a prefetched batch %d  waits inside the launch %d  launches left for one %d  threads per block %d\n",
              v3_pf_seen, v3_pf_chains, v3_pf_stat[3], v3_pf_stat[0], v3_pf_stat[1], v3_pf_stat[2], v3_bs);
    if (knobs().v2_summary)
      fprintf(stderr, "[v3] blocks %ld  launches %ld  grid %d  halts none/nobirth/host/more/cap/max %ld %ld %ld %ld %ld %ld  pauses %ld  ms: wait %.1f replay %.1f enqueue %.1f total %.1f  moves %llu misses %llu\n",
              n_blocks, v3_enq, v3_grid, n_halt[0], n_halt[1], n_halt[2], n_halt[3], n_halt[4], n_halt[5], n_pause, st.ms_wait_device, st.ms_replay,
              st.ms_enqueue, ms_since(t0), (unsigned long long)st.nmoves, (unsigned long long)v2_miss_launches);
    st.ms_bookkeep += ms_since(t0);
  }

  // one chain: [shuffle x nlev][b_p_update + b_bud arg-min + ties][birth / plan / publish]; a full round is a chain with the
  // batch compare in front (no-ops on a cache hit) and the round's store filter in its first shuffle
  // A full round is always the same launches with the same arguments (what differs lives in the control block), so it can be
  // captured once into a hipGraph and replayed: one API call per round (DADA2HIP_V2_GRAPH=1; off by default, see graph_off()),
  // ... in both forms: [0] with the batch compare in front, [1] without (Eng2::has_compare)
  hipGraphExec_t v2_graph[2] = {nullptr, nullptr};
  int v2_graph_state = 0;            // 0: not tried yet, 1: in use, -1: unavailable
  int v2_plain_rounds = 0;
  long v2_next_full = 0;             // first chain (sequence number) that is expected to need a batch compare again
  bool v2_lite_on = true;
  bool v2_align_commit = false;       // Eng2::align_at_commit
  void v2_drop_graph() {
    for (auto &g : v2_graph) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    if (v2_graph_state == 1) v2_graph_state = 0;
  }
  bool v2_try_graph() {
    if (graph_off() || profile_all || v2_graph_state < 0) return false;
    if (v2_graph_state == 1) return true;
    if (v2_plain_rounds < 1) return false;   // the first round goes out plainly (one-time function attributes are set by its launches)
    hipStream_t stq = s->stream;
    for (int lite = 0; lite < 2; lite++) {
      hipGraph_t g = nullptr;
      if (hipStreamBeginCapture(stq, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); v2_drop_graph(); v2_graph_state = -1; return false; }
      v2_round_launches(v2_chain, true, true, nullptr, lite != 0);
      if (hipStreamEndCapture(stq, &g) != hipSuccess || !g) { (void)hipGetLastError(); v2_drop_graph(); v2_graph_state = -1; return false; }
      const hipError_t e = hipGraphInstantiate(&v2_graph[lite], g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (e != hipSuccess) { (void)hipGetLastError(); v2_graph[lite] = nullptr; v2_drop_graph(); v2_graph_state = -1; return false; }
    }
    v2_graph_state = 1;
    return true;
  }
  // hipGraph replay is OFF by default: a graph launch leaves a 13 us bubble behind its last kernel (profiles/r03h, r03s), 10 ms
  // per pass at 10^6 uniques, while the six or ten plain launches of a chain cost the host 25 us per round - a third of what
  // it has (it trails the device anyway) - and run back to back: 173 against 183 ms per pass (profiles/r03x vs r03w).
  // DADA2HIP_V2_GRAPH=1 brings the graphs back (a host that is short of cycles: 8 instead of 25 ms of enqueue per pass).
  static bool graph_off() { return !knobs().v2_graph; }
  void v2_enqueue_chain(int nlev, bool with_compare, bool store) {
    const auto t_enq = clk::now();
    EnqRec rec{-1, -1, with_compare};
    // a round that is expected to find its centre cached goes out without the batch compare's launches (engine.h, Eng2::has_compare)
    const bool lite = with_compare && v2_lite_on && v2_plain_rounds >= 1 && v2_enq + 1 < v2_next_full;
    if (nlev == v2_chain && with_compare && store && v2_try_graph()) D2_HIP(hipGraphLaunch(v2_graph[lite ? 1 : 0], s->stream));
    else {
      v2_round_launches(nlev, with_compare, store, &rec, lite);
      if (with_compare) v2_plain_rounds++;
    }
    if (lite) st.lite_chains++;   // (sending the chains behind an expected refill without a compare as well was tried: as many
                                  //  more wrong guesses as it saved launches, profiles/README.md r03u)
    v2_enqrec.push_back(rec);
    v2_enq++;
    st.ms_enqueue += ms_since(t_enq);
  }
  void v2_round_launches(int nlev, bool with_compare, bool store, EnqRec *recp, bool lite = false) {
    hipStream_t stq = s->stream;
    EnqRec rec{-1, -1, with_compare};
    const Eng2 &E2 = lite ? this->E2L : this->E2;       // (chains without the compare say so to their kernels)
    if (with_compare && !lite) {
      // the round's comparisons: a batch screen if its centre is not cached (no-op otherwise), the work lists of the centre
      // with the greedy skip as of now, the aligner on them (centre read from the control block)
      rec.ev_screen = ev_begin(EV_SCREEN, profile_all, /*spec=*/true);
      launch2_screen_multi(E2, stq);
      ev_end(rec.ev_screen);
      // ... its survivors through the aligner, all batch positions in one launch (both no-ops on a cache hit) ...
      launch2_batch_lists(E2, stq);
      const NwBatch nb{&v2_ctl.p->nalign, v2_blistn.p, v2_blist.p, v2_ctl.p->acentre, &v2_ctl.p->abuf, E2.C.Npad, v2_bretry.p, v2_bretryn.p, v2_retrytot.p};
      launch_gapless_batch(s->D, nb, ap, s->d_err.p, v2_lamB.p, v2_hamB.p, &v2_ctl.p->state, stq);
      rec.ev_nw = ev_begin(EV_NW, profile_all, /*spec=*/true);
      launch_nw_ad(s->D, -1, nullptr, nullptr, nullptr, 0, nullptr, nullptr, ap, s->d_err.p, v2_lamB.p, v2_hamB.p, nullptr, 0, 0, nullptr, stq,
                   &v2_ctl.p->state, &nb);
      ev_end(rec.ev_nw);
    }
    int ev = ev_begin(EV_SHUFFLE, profile_all && nlev > 0);
    for (int l = 0; l < nlev; l++) launch2_shuffle(E2, l, store && l == 0, stq);
    ev_end(ev);
    ev = ev_begin(EV_PVAL, profile_all);
    launch2_eval(E2, nlev, s->h_reads[bi[0].center], stq);
    ev_end(ev);
    if (recp) *recp = rec;
  }

  const Round2Out &v2_wait_block() {
    const int ring = (int)(v2_cons % RING2);
    const int32_t want = (int32_t)(v2_cons + 1);
    volatile int32_t *seqp = &v2_hblk.p[ring].seq;
    const auto tw = clk::now();
    const bool polite = wait_blocks();
    for (unsigned spins = 0; *seqp != want; spins++) {
      cpu_relax();
      if (polite && spins > 64) { struct timespec ts{0, 20000}; nanosleep(&ts, nullptr); }
      if ((spins & 0xFFFF) == 0xFFFF || (polite && (spins & 0xFF) == 0xFF)) {
        hipError_t e = hipStreamQuery(s->stream);
        if (e != hipSuccess && e != hipErrorNotReady)
          throw d2::DeviceError{DADA2HIP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(e) + " (round result)"};
        if (e == hipSuccess && *seqp != want)
          throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: the device did not publish the round result"};
        if (ms_since(tw) > wait_timeout_s() * 1e3)
          throw d2::DeviceError{DADA2HIP_ERR_DEVICE, "dada2hip: timed out waiting for the round result"};
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    v2_cons++;
    st.ms_wait_device += ms_since(tw);
    return v2_hblk.p[ring];
  }

  // bring the host mirror up to date with one published block: the chain's moves in call order, then the counters
  // resume_after_fetch (persistent tail, a paused block): the device waits halted until the host has the full mover lists - it is
  // resumed as soon as they have been COPIED, the replay of (possibly 10^5) moves then runs beside it
  void v2_replay(const Round2Out &b, long seq, bool resume_after_fetch = false) {
    const auto t_rep = clk::now();
    int tot = 0;
    for (int l = 0; l < b.nsh; l++) tot += b.cnt[l];
    st.nmoves += (uint64_t)tot;
    if (tot <= E2.mov_inline) {
      int off = 0;
      for (int l = 0; l < b.nsh; l++) { replay_moves(b.mov + 3 * off, b.cnt[l]); off += b.cnt[l]; }
    } else {
      // (launch chains keep MOV_RING sets of lists; the persistent tail one set, and it stays halted until resumed)
      const int ring = (int)((seq - 1) % MOV_RING);
      h_big.alloc((size_t)3 * tot);
      size_t off = 0;
      for (int l = 0; l < b.nsh; l++) {
        const int nm = b.cnt[l];
        if (!nm) continue;
        const size_t slot = b.kord ? (size_t)l : (size_t)(ring * SH_CHAIN + l);
        D2_HIP(hipMemcpyAsync(h_big.p + 3 * off, v2_movers.p + slot * 3 * N, (size_t)3 * nm * 4, hipMemcpyDeviceToHost, s->side));
        off += (size_t)nm;
      }
      D2_HIP(hipStreamSynchronize(s->side));
      if (resume_after_fetch) {
        launch2_resume(E2, s->stream, /*keep_list=*/true);
        v3_enqueue(false);                                     // (what was queued behind the pause found the device halted)
      }
      off = 0;
      for (int l = 0; l < b.nsh; l++) { replay_moves(h_big.p + 3 * off, b.cnt[l]); off += (size_t)b.cnt[l]; }
    }
    st.nshuffle += (uint64_t)b.nsh;
    st.nnw += b.stat[0]; st.ngapless += b.stat[1]; st.nshroud += b.stat[2]; st.nskipped += b.stat[3];
    st.nnw_rounds += b.stat[0] - (uint64_t)b.pad0[3];          // (what the rounds COMMITTED: nnw_run - this = aligned in vain; pad0[3]: round 0's pairs, which ride in the first block)
    st.nstored += (uint64_t)b.pad0[0];                      // comparisons kept by the chain's store filter
    st.ms_replay += ms_since(t_rep);
    if (b.err_flag & 4) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "N-W Align out of range."};
    if (b.err_flag & 8) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: a cached compare lacks a comparison its round needs"};
    check_errflag(b.err_flag);
  }

  void v2_grow(const Round2Out &b) {
    sync_spin(s->stream);
    if (nclust_dev + 2 > ccap) {
      grow_clusters(std::max(ccap * 2, nclust_dev + 2));
      v2_dlt.alloc((size_t)SH_LEVELS * ccap);                      // (deltas are all zero between chains)
      D2_HIP(hipMemsetAsync(v2_dlt.p, 0, (size_t)SH_LEVELS * ccap * 4, s->stream));
    }
    if ((size_t)b.blk_count + (size_t)N > v2_blk.n) {
      const size_t cap = std::max(v2_blk.n * 2, (size_t)b.blk_count + 2 * (size_t)N);
      DevBuf<CompBlk> nb;
      nb.alloc(cap);
      D2_HIP(hipMemcpyAsync(nb.p, v2_blk.p, (size_t)b.blk_count * sizeof(CompBlk), hipMemcpyDeviceToDevice, s->stream));
      D2_HIP(hipStreamSynchronize(s->stream));
      std::swap(v2_blk.p, nb.p); std::swap(v2_blk.n, nb.n);
    }
    v2_bind();
  }

  // run_dada's loop (Rmain.cpp:312-331) with the device in charge of the rounds
  // resume: taking over from the persistent tail in mid-run (its entry barrier failed); first: nothing has been evaluated yet
  void run_v2(int max_clust, bool resume = false, bool first = true) {
    auto t0 = clk::now();
    if (!resume) st.nstored = (uint64_t)N;                     // round 0 keeps every comparison (E_minmax starts at -999)
    if (first) v2_enqueue_chain(0, false, false);             // b_p_update after round 0 + the first b_bud
    bool done = false;
    double t_decide = 0, t_halt = 0, t_top = 0;
    long n_halt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_big = 0, n_kind[5] = {0, 0, 0, 0, 0};
    while (!done) {
      const auto t_loop = clk::now();
      while (v2_enq - v2_cons < v2_depth) v2_enqueue_chain(v2_chain, true, true);
      t_top += ms_since(t_loop);
      const long seq = v2_cons + 1;
      const Round2Out &b = v2_wait_block();
      if (hooks && hooks->should_abort && hooks->should_abort(hooks->user))
        throw RuntimeErr{DADA2HIP_ERR_ABORTED, "dada2hip: aborted by caller"};
      // keep the device fed before the host mirror catches up with this block
      if (b.halt == H2_NONE) while (v2_enq - v2_cons < v2_depth) v2_enqueue_chain(v2_chain, true, true);
      if (v2_debug)
        fprintf(stderr, "[v2] blk %ld halt %d nclust %d nlev %d nsh %d cnt %d %d %d %d nbatch %d slot %d birth %d found %d nties %d p %.3e blk %d\n", seq,
                b.halt, b.nclust, b.nlev, b.nsh, b.cnt[0], b.cnt[1], b.cnt[2], b.cnt[3], b.nbatch, b.slot, b.birth_applied, b.bud.found[0],
                b.bud.nties[0], b.bud.best_p[0], b.blk_count);
      if (b.halt == H2_HOST_DECIDE && knobs().v2_summary) {   // what kind of decision came back to the host
        const int nt = b.bud.nties[0];
        bool zero = b.bud.found[0] && nt > 1 && nt <= BUD_TIES, in0 = false, pristine = true;
        int cmin = INT32_MAX, ncmin = 0;
        for (int k = 0; zero && k < nt; k++) {
          const BudTie &t = b.bud.ties[0][k];
          if (t.p != 0.0) zero = false;
          if (clust_of[t.raw] == 0) { in0 = true; if (slot_of[t.raw] != t.raw) pristine = false; }
          if (clust_of[t.raw] < cmin) { cmin = clust_of[t.raw]; ncmin = 1; } else if (clust_of[t.raw] == cmin) ncmin++;
        }
        n_kind[!zero ? 0 : (in0 ? (pristine ? 1 : 2) : (ncmin == 1 ? 3 : 4))]++;
      }
      v2_replay(b, seq);
      n_halt[b.halt & 7]++;
      if (b.cnt[0] + b.cnt[1] + b.cnt[2] + b.cnt[3] > MOV_INLINE2) n_big++;
      const auto t_dec = clk::now();
      const EnqRec &rec = v2_enqrec[seq - 1];
      if (rec.compare && b.pad0[1] + b.pad0[2] > 0) {          // (the aligner launches of this chain had work)
        if (rec.ev_nw >= 0) evs[rec.ev_nw].ok = 1;
        st.nnw_run += (uint64_t)b.pad0[1]; st.ngapless_run += (uint64_t)b.pad0[2];
      }
      if (rec.compare && b.nbatch > 0) {                       // (this chain's batch compare really ran: a cache miss)
        v2_miss_launches++;
        if (rec.ev_screen >= 0) evs[rec.ev_screen].ok = 1;
        v2_next_full = seq + b.nbatch;                         // the batch holds the centres of this round and, if the guesses hold, the next nbatch - 1
      }
      if (b.nlev > 0 && b.halt != H2_SHUFFLE_MORE) { }        // (a round's commit is complete)
      switch (b.halt) {
        case H2_NONE: {                                        // birth applied on the device: book it
          Birth bb = decide_bud(b.bud);
          if (!bb.yes || bb.type != 'A' || bb.c.raw != b.bud.ties[0][0].raw)
            throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: device and host bud decisions differ"};
          logf("\nNew Cluster C%i:", bb.newi);
          nclust_dev = bb.newi + 1;
          record_birth(bb);
          st.ncompare += (uint64_t)N;
          break;
        }
        case H2_NO_BIRTH: {
          Birth bb = decide_bud(b.bud);
          if (bb.yes) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: device and host bud decisions differ"};
          done = true;
          break;
        }
        case H2_MAXCLUST: done = true; break;
        case H2_HOST_DECIDE:
        case H2_CAPACITY: {
          // the launches already queued behind the halt are no-ops (they neither touch state nor publish): the host's birth
          // is simply queued behind them, no drain needed unless buffers have to grow
          v2_enq = v2_cons;
          v2_enqrec.resize((size_t)v2_cons);
          if (b.halt == H2_CAPACITY) v2_grow(b);
          Birth bb = decide_bud(b.bud);
          if (!bb.yes) { done = true; break; }
          if (bb.newi + 2 > ccap) v2_grow(b);
          logf("\nNew Cluster C%i:", bb.newi);
          launch2_host_birth(E2, bb.c.raw, bb.c.from, s->stream);
          v2_next_full = 0;                                    // (whether the new centre is cached is not known here: send a full chain)
          nclust_dev = bb.newi + 1;
          record_birth(bb);
          st.ncompare += (uint64_t)N;
          break;
        }
        case H2_NEED_COMPARE: {                                // a chain without the compare met a centre that is not cached
          v2_enq = v2_cons;
          v2_enqrec.resize((size_t)v2_cons);
          v2_next_full = 0;
          st.lite_misses++;
          launch2_resume(E2, s->stream, /*keep_list=*/true);
          break;
        }
        case H2_SHUFFLE_MORE: {                                // more than SH_CHAIN moving shuffles: continue the same round
          v2_enq = v2_cons;                                    // (the launches queued behind the halt are no-ops: no drain needed)
          v2_enqrec.resize((size_t)v2_cons);
          launch2_resume(E2, s->stream);
          v2_enqueue_chain(v2_chain, false, false);
          break;
        }
        default: throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: internal error: unknown halt code"};
      }
      if (b.halt == H2_NONE) t_decide += ms_since(t_dec); else t_halt += ms_since(t_dec);
    }
    if (knobs().v2_summary)
      fprintf(stderr, "[v2] blocks %ld  halts none/nobirth/host/more/cap/max/need-compare %ld %ld %ld %ld %ld %ld %ld  chains without compare %llu  big-mover blocks %ld  ms: wait %.1f replay %.1f "
                      "enqueue %.1f decide %.1f halt-handling %.1f top-up %.1f total %.1f  moves %llu misses %llu  host decisions other/zero-ties in partition 0 at "
                      "their first slots/... moved/elsewhere, one in the lowest partition/... several %ld %ld %ld %ld %ld\n", v2_cons, n_halt[0], n_halt[1],
              n_halt[2], n_halt[3], n_halt[4], n_halt[5], n_halt[6], (unsigned long long)st.lite_chains, n_big, st.ms_wait_device, st.ms_replay, st.ms_enqueue, t_decide, t_halt, t_top,
              ms_since(t0), (unsigned long long)st.nmoves, (unsigned long long)v2_miss_launches, n_kind[0], n_kind[1], n_kind[2], n_kind[3], n_kind[4]);
    sync_spin(s->stream);                                      // no-op launches queued behind the final halt
    if (v2_trace_seq >= 0 && v2_trace.p) {                     // dump the traced round's stamps (tools/trace_round.py reads them)
      std::vector<unsigned long long> h((size_t)TRACE_KERNELS * TRACE_BLOCKS * 8);
      D2_HIP(hipMemcpy(h.data(), v2_trace.p, h.size() * 8, hipMemcpyDeviceToHost));
      if (FILE *f = fopen(knobs().v2_trace_file.c_str(), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    st.ms_bookkeep += ms_since(t0);
  }

  uint64_t nw_cells_per_alignment() const {
    // algorithmic DP cells of one alignment: (L1 + L2 + 1) anti-diagonals x (band + 1)
    const int L = s->D.maxlen;
    if (o.band_size < 0) return (uint64_t)(L + 1) * (L + 1);
    return (uint64_t)(2 * L + 1) * (uint64_t)(o.band_size + 1);
  }
};

void upload_err(dada2hip_sample *s, const double *err_colmajor, int ncol, std::vector<double> &rowmajor) {
  rowmajor.resize((size_t)16 * ncol);
  for (int r = 0; r < 16; r++)
    for (int c = 0; c < ncol; c++) rowmajor[(size_t)r * ncol + c] = err_colmajor[(size_t)c * 16 + r];   // cluster.cpp:166-170
  s->d_err.alloc(rowmajor.size());
  D2_HIP(hipMemcpyAsync(s->d_err.p, rowmajor.data(), rowmajor.size() * 8, hipMemcpyHostToDevice, s->stream));
}

void alloc_round_buffers(dada2hip_sample *s) {
  const size_t N = (size_t)s->D.N;
  s->d_skip.alloc(N); s->d_cls.alloc(N); s->d_lambda.alloc(N); s->d_ham.alloc(N);
  s->d_nw_list.alloc(N); s->d_gl_list.alloc(N); s->d_counters.alloc(8); s->d_thresh.alloc(s->D.maxlen + 2);
  s->d_ctab.alloc(768 + (size_t)s->D.LK / 2 + 64);
  s->h_counters.alloc(8);
}

void check_opts(const dada2hip_opts &o, int qmax, int ncol) {
  if (ncol < 1) throw InputError{"Error matrix must have 16 rows."};
  // (checked whatever USE_QUALS says: the output tables index err's columns by quality, error.cpp:152-167)
  if (qmax > ncol - 1) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "Rounded quality exceeded range of err lookup table."};
}

void init_run(Run &run, dada2hip_sample *s, const double *err, int err_ncol, const dada2hip_opts *opts, double cutoff) {
  run.s = s; run.o = *opts; run.N = s->D.N; run.ncol = err_ncol;
  memset(&run.st, 0, sizeof run.st);
  upload_err(s, err, err_ncol, run.err_rowmajor);
  alloc_round_buffers(s);
  run.wclass = nw_class(opts->band_size, s->D.maxlen, s->D.minlen);   // (the lane kernel's scratch ring is allocated on first use)
  run.ap = AlignParams{opts->match, opts->mismatch, opts->gap, opts->band_size, nw_sentinel(*opts), opts->use_quals, err_ncol};
  // raw_align's choice (nwalign_endsfree.cpp:57-64): nwalign_endsfree_homo only with VECTORIZED_ALIGNMENT off and a homopolymer
  // penalty that differs from the gap penalty (R/dada.R:229-231 switches the vectorized aligner off in that case)
  run.ap.homo_gap = (!opts->vectorized_alignment && opts->homo_gap != opts->gap && opts->homo_gap <= 0) ? opts->homo_gap : opts->gap;
  run.ap.endsfree = 1;
  run.sp = ScreenParams{opts->use_kmers, opts->gapless, opts->band_size, opts->SSE};
  run.thresh_round = make_thresh(s->D.maxlen, cutoff);
  run.thresh_one = make_thresh(s->D.maxlen, 1.0);
}

// ---- dada_uniques proper: run_dada (Rmain.cpp:297-336) + outputs (Rmain.cpp:172-294, error.cpp) --
void sample_run(dada2hip_sample *s, const double *err, int err_ncol, const dada2hip_opts *opts,
                const dada2hip_hooks *hooks, dada2hip_result *R, const dada2hip_shard *shard = nullptr) {
  auto t_total = clk::now();
  select_device(s->device);
  if (!err || !opts) throw InputError{"Error matrix must have 16 rows."};
  check_opts(*opts, s->qmax, err_ncol);
  struct ActiveGuard { int dev; ActiveGuard(int d) : dev(d) { active_runs(dev)++; } ~ActiveGuard() { active_runs(dev)--; } } active_guard{s->device};
  SampleDev &D = s->D;
  const int N = D.N;
  if (shard && (shard->world < 1 || shard->rank < 0 || shard->rank >= shard->world || !shard->exchange))
    throw InputError{"dada2hip: invalid shard descriptor."};
  if (shard && shard->world == 1) shard = nullptr;      // one rank: the ordinary run
  // this rank's block of uniques (restored on every exit: the resident sample serves unsharded runs too)
  struct RangeGuard { SampleDev &D; ~RangeGuard() { D.r_lo = 0; D.r_hi = D.N; } } range_guard{D};
  D.r_lo = shard ? (int32_t)((int64_t)N * shard->rank / shard->world) : 0;
  D.r_hi = shard ? (int32_t)((int64_t)N * (shard->rank + 1) / shard->world) : N;
  double t_sub[5] = {0, 0, 0, 0, 0};                  // (DADA2HIP_V2_SUMMARY: where the time in front of round 0 goes)
  auto t_lap = clk::now();
  auto lap = [&](int k) { t_sub[k] = ms_since(t_lap); t_lap = clk::now(); };
  ensure_ad_ring(s);
  AdRingGuard ring_guard{s};
  if (!s->run_cache) s->run_cache = std::make_shared<Run>();
  Run &run = *static_cast<Run *>(s->run_cache.get());
  lap(0);
  run.hooks = hooks;
  run.shard = shard; run.lo = D.r_lo; run.hi = D.r_hi;
  run.sh_exchange_failed = false; run.sh_points_left = true;
  run.h_upd.assign(1, 1);
  init_run(run, s, err, err_ncol, opts, opts->kdist_cutoff);
  run.st.ms_upload = s->ms_upload;
  hipStream_t stq = s->stream;
  lap(1);
  run.alloc_state();
  lap(2);
  // b_init (containers.cpp:111-137): one partition holding every unique, centre = first max-reads member
  run.clust_of.assign(N, 0);
  run.slot_of.resize(N);
  run.bi.emplace_back();
  {
    Bi &b0 = run.bi[0];
    b0.birth_type = 'I'; b0.birth_fold = 1.0; b0.birth_e = (double)(uint32_t)s->total_reads;
    b0.raw.resize(N);
    uint32_t mx = 0;
    for (int i = 0; i < N; i++) {
      b0.raw[i] = (uint32_t)i;
      run.slot_of[i] = i;
      b0.reads += s->h_reads[i];
      if (s->h_reads[i] > mx) { b0.center = (uint32_t)i; mx = s->h_reads[i]; }
    }
    run.plain = knobs().no_speculation || b0.raw[0] != b0.center || shard != nullptr;
    run.no_auto = run.plain || knobs().no_autobirth;
    const uint8_t one = 1;
    D2_HIP(hipMemcpy(run.P.slot0, &one, 1, hipMemcpyHostToDevice));   // unique 0 sits in slot 0 of partition 0
    run.push_cluster(0, true, true);
    D2_HIP(hipMemcpy(run.d_creads_snap.p, run.P.creads, 4, hipMemcpyDeviceToDevice));
  }

  run.nclust_dev = 1;
  int max_clust = opts->max_clust < 1 ? N : opts->max_clust;
  run.max_clust_run = max_clust;
  run.use_v2 = max_clust > 1 && !shard && run.want_v2();
  lap(3);
  if (run.use_v2) run.v2_alloc(max_clust);
  lap(4);
  run.st.ms_setup = ms_since(t_total);
  if (knobs().v2_summary)
    fprintf(stderr, "[run] setup %.2f ms: aligner ring / run cache %.2f  init_run %.2f  alloc_state %.2f  b_init mirror %.2f  v2_alloc %.2f\n", run.st.ms_setup,
            t_sub[0], t_sub[1], t_sub[2], t_sub[3], t_sub[4]);
  const auto t_round0 = clk::now();
  run.compare_round(0, (int)run.bi[0].center, 1.0);   // Rmain.cpp:309-310: no k-mer screen in round 0
  run.st.ms_round0 = ms_since(t_round0);
  if (knobs().v2_summary) fprintf(stderr, "[run] round 0 (host side): %.2f ms\n", run.st.ms_round0);
  // run_dada's loop (Rmain.cpp:312-331), rotated: every iteration ends with b_p_update + the b_bud that opens
  // the reference's next iteration, so one device round trip serves both.  After a decision the next round's
  // kernels are launched first; the host mirror (moves replay, birth record) is updated while the GPU works.
  if (run.use_v2 && run.v3_on && run.v3_acquire()) run.run_v3(max_clust);
  else if (run.use_v2) {
    // the launch chains (asked for, or another run holds the device's persistent slot) plan no prefetch compares: nobody would
    // launch them, and a round must never take its comparisons from a batch that was only planned
    if (run.v3_overlap) { run.v3_overlap = false; run.v2_bind(); run.v3_pf_release(); }
    run.run_v2(max_clust);
  }
  else if (run.nclust_dev < max_clust) {
    run.round_tail(false);                            // b_p_update after round 0, then the first b_bud
    for (;;) {
      Run::Birth b = run.decide_bud();
      if (!b.yes) { run.replay_pending(); break; }
      run.logf("\nNew Cluster C%i:", b.newi);
      if (run.h_rout.p->bud.auto_applied) run.confirm_auto_birth(b);   // applied on the device, next round already running
      else {
        run.spec_launched = false;                     // (a speculative round, if any, found next = -1 and did nothing)
        run.launch_birth(b);
        run.compare_round(b.newi, b.c.raw, opts->kdist_cutoff);
      }
      const bool more = run.nclust_dev < max_clust;
      run.replay_pending();                           // host mirror catches up while the GPU runs the round
      run.record_birth(b);
      if (hooks && hooks->should_abort && hooks->should_abort(hooks->user))
        throw RuntimeErr{DADA2HIP_ERR_ABORTED, "dada2hip: aborted by caller"};
      if (!more) { run.round_tail_no_bud(); break; }  // max_clust reached: shuffle to stability, no further b_bud
      run.round_tail(true);
    }
  }
  run.st.rounds = (uint32_t)run.bi.size();
  if (run.use_v2 && run.v2_retrytot.p) {   // (the rounds are over and their streams drained: a plain copy)
    unsigned long long rt[4] = {0, 0, 0, 0};
    D2_HIP(hipMemcpy(rt, run.v2_retrytot.p, 32, hipMemcpyDeviceToHost));
    run.st.nnw_retry = rt[0]; run.st.nnw_fast = rt[1]; run.st.screen_stage2 = rt[3];
  }

  // ---- final alignments (Rmain.cpp:172-236): every member vs its centre, use_kmers = false ----------
  auto t_final = clk::now();
  const int C = (int)run.bi.size();
  const int LV = D.maxlen;
  // work slots grouped in chunks that share a centre: 64 (one wave of the lane kernel) or the cooperative kernel's
  // alignments per wave
  const int fk = knobs().nw_kernel;
  const bool coop_fin_ok = opts->band_size != 0 && nw_ad_lds_bytes(D, run.ap) > 0 && nw_ad_lds_bytes(D, run.ap) <= 150 * 1024;
  bool coop_fin = coop_fin_ok && N < COOP_MAX_BATCH;
  if (fk == NWK_LANE) coop_fin = false;
  if (fk == NWK_COOP && coop_fin_ok) coop_fin = true;
  // band windows too wide for the LDS-pointer kernel: the wide anti-diagonal kernel (same rule as compare_round)
  bool wide_fin = !coop_fin_ok && opts->band_size > 0 && nw_adw_ok(D, run.ap) && nw_adw_lds_bytes(D, run.ap) <= 150 * 1024 &&
                  run.wclass != 33 && run.wclass != 65 && N < (1 << 20);
  if (fk == NWK_LANE) wide_fin = false;
  if (fk == NWK_WIDE) wide_fin = opts->band_size > 0 && nw_adw_ok(D, run.ap) && nw_adw_lds_bytes(D, run.ap) <= 150 * 1024;
  if (wide_fin) { coop_fin = false; ensure_adw_scratch(s, run.ap); }
  const size_t per = coop_fin ? (size_t)nw_ad_apw(D, run.ap) : (wide_fin ? (size_t)nw_adw_apw(D, run.ap) : 64);
  std::vector<int32_t> work, chunk_centre, centre_of_cluster(C);
  work.reserve((size_t)N + per * (size_t)C);
  uint64_t n_final_local = 0;
  if (!shard) {
    // every member of every partition, a partition's members in list order, padded to whole chunks: the member lists ARE the
    // work list (one block copy per partition instead of a push per unique: 10^6 uniques cost 5 ms here)
    size_t tot = 0;
    for (int i = 0; i < C; i++) tot += (run.bi[i].raw.size() + per - 1) / per * per;
    work.assign(tot, -1);
    chunk_centre.resize(tot / per);
    size_t off = 0;
    for (int i = 0; i < C; i++) {
      centre_of_cluster[i] = (int32_t)run.bi[i].center;
      const auto &m = run.bi[i].raw;
      static_assert(sizeof(m[0]) == sizeof(int32_t), "member lists are 32-bit");
      if (!m.empty()) memcpy(work.data() + off, m.data(), m.size() * sizeof(int32_t));
      const size_t nch = (m.size() + per - 1) / per;
      std::fill(chunk_centre.begin() + off / per, chunk_centre.begin() + off / per + nch, (int32_t)run.bi[i].center);
      off += nch * per;
      n_final_local += m.size();
    }
  } else
  for (int i = 0; i < C; i++) {
    centre_of_cluster[i] = (int32_t)run.bi[i].center;
    const auto &m = run.bi[i].raw;
    size_t kk = 0;
    for (size_t k = 0; k < m.size(); k++) {
      if ((int)m[k] < D.r_lo || (int)m[k] >= D.r_hi) continue;   // (sharded run: the members of this rank's block)
      if (kk++ % per == 0) chunk_centre.push_back((int32_t)run.bi[i].center);
      work.push_back((int32_t)m[k]);
      n_final_local++;
    }
    while (work.size() % per) work.push_back(-1);
  }
  if (work.empty()) { work.assign(per, -1); chunk_centre.push_back((int32_t)run.bi[0].center); }
  s->d_work.alloc(work.size()); s->d_chunk_centre.alloc(chunk_centre.size());
  s->d_view.alloc((size_t)N * LV);
  s->d_correct.alloc(N);
  s->d_trans.alloc((size_t)16 * err_ncol); s->d_nsubs.alloc(N);
  s->d_qsum.alloc((size_t)C * D.maxlen); s->d_qn.alloc((size_t)C * D.maxlen);
  D2_HIP(hipMemcpyAsync(s->d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, stq));
  D2_HIP(hipMemcpyAsync(s->d_chunk_centre.p, chunk_centre.data(), chunk_centre.size() * 4, hipMemcpyHostToDevice, stq));
  if (opts->band_size == 0) {
    launch_gapless(D, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, (int)work.size(), run.ap, s->d_err.p, s->d_lambda.p,
                   s->d_ham.p, s->d_view.p, LV, 0, stq);
    run.st.ngapless += n_final_local;
  } else {
    const int evn_i = run.ev_begin(Run::EV_NW, true, false, /*big=*/true);
    if (coop_fin)
      launch_nw_ad(D, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, (int)work.size(), nullptr, nullptr, run.ap, s->d_err.p,
                   s->d_lambda.p, s->d_ham.p, s->d_view.p, LV, 0, nullptr, stq);
    else if (wide_fin)
      launch_nw_adw(D, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, (int)work.size(), run.ap, s->d_err.p, s->scr_adw.p,
                    s->scr_adw_wpw, s->scr_adw_waves, s->d_lambda.p, s->d_ham.p, s->d_view.p, LV, 0, stq);
    else {
      ensure_scratch(s, run.ap.band);
      launch_nw(D, run.wclass, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, (int)work.size(), run.ap, s->d_err.p, s->scr,
                s->d_lambda.p, s->d_ham.p, s->d_view.p, LV, 0, nullptr, 0, nullptr, stq);
    }
    run.ev_end(evn_i);
    run.st.nnw += n_final_local;
  }
  // final per-unique p and the OMEGA_C decision (Rmain.cpp:238-252), on the device
  const int ev_fin = run.ev_begin(Run::EV_FINAL, run.profile_all);
  launch_final_p(run.P, D, opts->omegaC, s->d_correct.p, stq);
  std::vector<uint8_t> correct(N);
  R->pval.assign(N, 0.0);
  D2_HIP(hipMemcpyAsync(R->pval.data(), run.P.p, (size_t)N * 8, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemcpyAsync(correct.data(), s->d_correct.p, (size_t)N, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemsetAsync(s->d_trans.p, 0, (size_t)16 * err_ncol * 4, stq));
  D2_HIP(hipMemsetAsync(s->d_qsum.p, 0, (size_t)C * D.maxlen * 8, stq));
  D2_HIP(hipMemsetAsync(s->d_qn.p, 0, (size_t)C * D.maxlen * 4, stq));
  launch_final_tables(D, s->d_view.p, LV, s->d_work.p, (int)work.size(), run.P.clust_of, run.P.centre_of, s->d_correct.p, err_ncol, 1,
                      s->d_trans.p, s->d_qsum.p, s->d_qn.p, s->d_nsubs.p, C, stq);
  run.ev_end(ev_fin);
  std::vector<int32_t> nsubs(N);
  std::vector<unsigned long long> qsum((size_t)C * D.maxlen);
  std::vector<uint32_t> qn((size_t)C * D.maxlen);
  R->subqual.assign((size_t)16 * err_ncol, 0);
  D2_HIP(hipMemcpyAsync(nsubs.data(), s->d_nsubs.p, (size_t)N * 4, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemcpyAsync(qsum.data(), s->d_qsum.p, qsum.size() * 8, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemcpyAsync(qn.data(), s->d_qn.p, qn.size() * 4, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemcpyAsync(R->subqual.data(), s->d_trans.p, R->subqual.size() * 4, hipMemcpyDeviceToHost, stq));

  // post-hoc partition p-values (error.cpp:101-119): comparisons of partition i against the centres of others
  std::vector<int32_t> ph_ji;
  std::vector<double> ph_lam;
  {
    std::vector<int32_t> cl_of_centre(N, -1);
    for (int i = 0; i < C; i++) cl_of_centre[run.bi[i].center] = i;
    const int cap = std::max(1 << 16, 4 * C * 64);
    run.d_cl_of_centre.alloc(N); run.d_ph_ji.alloc(2 * (size_t)cap); run.d_ph_lam.alloc(cap); run.d_ph_n.alloc(1);
    D2_HIP(hipMemcpyAsync(run.d_cl_of_centre.p, cl_of_centre.data(), (size_t)N * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemsetAsync(run.d_ph_n.p, 0, 4, stq));
    auto posthoc = [&](int capv) {
      if (run.use_v2) launch2_posthoc(run.E2, run.d_cl_of_centre.p, run.d_ph_ji.p, run.d_ph_lam.p, run.d_ph_n.p, capv, stq);
      else launch_posthoc(run.P, D, run.d_cl_of_centre.p, run.d_ph_ji.p, run.d_ph_lam.p, run.d_ph_n.p, capv, stq);
    };
    posthoc(cap);
    int32_t n = 0;
    D2_HIP(hipMemcpyAsync(&n, run.d_ph_n.p, 4, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipStreamSynchronize(stq));
    if (n > cap) {   // dense centre-vs-centre store: retry with the exact size
      run.d_ph_ji.alloc(2 * (size_t)n); run.d_ph_lam.alloc(n);
      D2_HIP(hipMemsetAsync(run.d_ph_n.p, 0, 4, stq));
      posthoc(n);
      D2_HIP(hipStreamSynchronize(stq));
    }
    ph_ji.resize(2 * (size_t)n); ph_lam.resize(n);
    if (n) {
      D2_HIP(hipMemcpy(ph_ji.data(), run.d_ph_ji.p, ph_ji.size() * 4, hipMemcpyDeviceToHost));
      D2_HIP(hipMemcpy(ph_lam.data(), run.d_ph_lam.p, ph_lam.size() * 8, hipMemcpyDeviceToHost));
    }
  }

  // birth substitutions (Rmain.cpp:209-215,231-234): parent centre vs new centre, k-mers on, cutoff 1.0.
  // One pair per wave chunk; the aligned views go to their own plane, one row per pair.
  std::vector<uint16_t> bview((size_t)C * LV, 0);
  if (C > 1) {
    const int nb = C - 1;
    std::vector<int32_t> bcc(nb), braw(nb);
    for (int i = 1; i < C; i++) { bcc[i - 1] = (int32_t)run.bi[run.bi[i].birth_comp.i].center; braw[i - 1] = (int32_t)run.bi[i].center; }
    DevBuf<int32_t> d_bcc, d_braw, d_wgl, d_wnw;
    DevBuf<uint8_t> d_bcls;
    d_bcc.alloc(nb); d_braw.alloc(nb); d_bcls.alloc(nb);
    D2_HIP(hipMemcpyAsync(d_bcc.p, bcc.data(), (size_t)nb * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(d_braw.p, braw.data(), (size_t)nb * 4, hipMemcpyHostToDevice, stq));
    launch_pair_class(D, d_bcc.p, d_braw.p, nb, run.sp, d_bcls.p, stq);
    std::vector<uint8_t> pair_cls(nb);
    D2_HIP(hipMemcpyAsync(pair_cls.data(), d_bcls.p, (size_t)nb, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipStreamSynchronize(stq));
    bool coop_b = coop_fin_ok && !wide_fin;          // one pair per chunk: the cooperative kernels need no 64-wide batch
    if (fk == NWK_LANE) coop_b = false;
    const size_t pern = coop_b ? (size_t)nw_ad_apw(D, run.ap) : (wide_fin ? (size_t)nw_adw_apw(D, run.ap) : 64);
    std::vector<int32_t> w_gl((size_t)nb * 64, -1), w_nw((size_t)nb * pern, -1);
    int n_gl = 0, n_nw = 0;
    for (int k = 0; k < nb; k++) {
      if (pair_cls[k] == CLS_GAPLESS) { w_gl[(size_t)k * 64] = braw[k]; n_gl++; }
      else { w_nw[(size_t)k * pern] = braw[k]; n_nw++; }
    }
    d_wgl.alloc(w_gl.size()); d_wnw.alloc(w_nw.size());
    s->d_view_b.alloc((size_t)nb * LV);
    D2_HIP(hipMemcpyAsync(d_wgl.p, w_gl.data(), w_gl.size() * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(d_wnw.p, w_nw.data(), w_nw.size() * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemsetAsync(s->d_view_b.p, 0, (size_t)nb * LV * 2, stq));
    if (n_gl) launch_gapless(D, 0, d_bcc.p, d_wgl.p, nullptr, (int)w_gl.size(), run.ap, s->d_err.p, s->d_lambda.p, s->d_ham.p,
                             s->d_view_b.p, LV, 1, stq);
    if (n_nw && coop_b)
      launch_nw_ad(D, 0, d_bcc.p, d_wnw.p, nullptr, (int)w_nw.size(), nullptr, nullptr, run.ap, s->d_err.p, s->d_lambda.p,
                   s->d_ham.p, s->d_view_b.p, LV, 1, nullptr, stq);
    else if (n_nw && wide_fin)
      launch_nw_adw(D, 0, d_bcc.p, d_wnw.p, nullptr, (int)w_nw.size(), run.ap, s->d_err.p, s->scr_adw.p, s->scr_adw_wpw,
                    s->scr_adw_waves, s->d_lambda.p, s->d_ham.p, s->d_view_b.p, LV, 1, stq);
    else if (n_nw) {
      ensure_scratch(s, run.ap.band);
      launch_nw(D, run.wclass, 0, d_bcc.p, d_wnw.p, nullptr, (int)w_nw.size(), run.ap, s->d_err.p, s->scr, s->d_lambda.p,
                s->d_ham.p, s->d_view_b.p, LV, 1, nullptr, 0, nullptr, stq);
    }
    D2_HIP(hipMemcpyAsync(&bview[(size_t)LV], s->d_view_b.p, (size_t)nb * LV * 2, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipStreamSynchronize(stq));
    if (!shard || shard->rank == 0) {                       // (every rank aligns the few birth pairs; counted once)
      run.st.ngapless += (uint64_t)n_gl;
      run.st.nnw += (uint64_t)n_nw;
    }
  }
  D2_HIP(hipStreamSynchronize(stq));
  D2_HIP(hipGetLastError());
  {   // error flag + run totals of the screen counters
    int32_t ef = 0;
    unsigned long long tot[4];
    D2_HIP(hipMemcpy(&ef, run.P.err_flag, 4, hipMemcpyDeviceToHost));
    D2_HIP(hipMemcpy(tot, run.P.totals, 32, hipMemcpyDeviceToHost));
    run.check_errflag(ef);
    check_nw_flag(s);
    run.st.nnw += tot[0]; run.st.ngapless += tot[1]; run.st.nshroud += tot[2]; run.st.nskipped += tot[3];
    if (!run.use_v2) {
      int32_t cnt = 0;
      D2_HIP(hipMemcpy(&cnt, run.P.node_count, 4, hipMemcpyDeviceToHost));
      run.st.nstored = (uint64_t)cnt;
    }
    // kernel times: event-timed launches, summed per kernel class.  Sampled mode: the per-round NW and screen launches
    // are extrapolated from the sampled ones; DADA2HIP_PROFILE=1: every launch was timed, the sums are exact.
    float ems;
    double cls_ms[Run::EV_NCLS] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double nw_big = 0, nw_small = 0, sc_sum = 0;
    int n_small = 0, n_sc = 0, n_big = 0;
    for (size_t k = 0; k < run.ev_used; k++) {
      const Run::EvRec &e = run.evs[k];
      if (!e.ok) continue;                                             // speculative launch that found nothing to do
      D2_HIP(hipEventElapsedTime(&ems, e.a, e.b));
      cls_ms[e.cls] += ems;
      // (the prefetch compares of the second stream are aligner / screen launches like the others - they run beside the persistent
      //  tail, which their durations show)
      if (e.cls == Run::EV_NW || e.cls == Run::EV_PF_NW) { if (e.big) { nw_big += ems; n_big++; } else { nw_small += ems; n_small++; } }
      if (e.cls == Run::EV_SCREEN || e.cls == Run::EV_PF_SCREEN) { sc_sum += ems; n_sc++; }
    }
    const int rounds = run.n_round_launches;
    if (run.profile_all) {
      run.st.nw_kernel_ms = nw_big + nw_small;
      run.st.nw_kernel_launches = (uint64_t)(n_big + n_small);
      run.st.screen_kernel_ms = sc_sum;
      run.st.screen_kernel_launches = (uint64_t)n_sc;
      run.st.kernel_times_sampled = 0;
      run.st.dev_ms_screen = cls_ms[Run::EV_SCREEN]; run.st.dev_ms_nw = cls_ms[Run::EV_NW];
      run.st.dev_ms_shuffle = cls_ms[Run::EV_SHUFFLE]; run.st.dev_ms_pval = cls_ms[Run::EV_PVAL];
      run.st.dev_ms_birth = cls_ms[Run::EV_BIRTH]; run.st.dev_ms_final = cls_ms[Run::EV_FINAL];
      run.st.dev_ms_tail = cls_ms[Run::EV_TAIL];
      run.st.dev_ms_pf_screen = cls_ms[Run::EV_PF_SCREEN]; run.st.dev_ms_pf_nw = cls_ms[Run::EV_PF_NW];
    } else {
      run.st.nw_kernel_ms = nw_big + (n_small ? nw_small / n_small * (rounds - 1) : 0.0);
      run.st.nw_kernel_launches = (uint64_t)rounds + 1;
      run.st.screen_kernel_ms = n_sc ? sc_sum / n_sc * rounds : 0.0;
      run.st.screen_kernel_launches = (uint64_t)rounds;
      run.st.kernel_times_sampled = 1;
    }
    run.st.nw_cells = run.st.nnw * run.nw_cells_per_alignment();
    {   // algorithmic bytes the screen launches had to read: a unique's ordered k-mer record + 6 B of scalars
      const uint64_t row = 2 * (uint64_t)(D.maxlen - KMER_SIZE + 1);
      if (run.use_v2) {
        const uint64_t passes = run.v2_miss_launches + run.st.pf_compares;                // batch compares: misses + prefetches
        run.st.screen_bytes = (uint64_t)N * (row + 6) * (1 + passes);                     // round 0 + one pass per batch compare
        run.st.batch_compares = passes;
        if (!run.profile_all) run.st.screen_kernel_launches = 1 + passes;
      } else run.st.screen_bytes = (run.st.ncompare - run.st.nskipped) * row + run.st.ncompare * 6;
    }
  }

  if (shard) {
    // ---- the blocks' results to every rank: per-unique values by block, sums by all-reduce -------------
    const int W = shard->world;
    auto gather_block = [&](void *base, size_t elem) {
      auto parts = run.sh_gatherv((const char *)base + (size_t)D.r_lo * elem, (size_t)(D.r_hi - D.r_lo) * elem);
      for (int w = 0; w < W; w++) {
        const size_t lo_w = (size_t)((int64_t)N * w / W);
        if (!parts[w].empty()) memcpy((char *)base + lo_w * elem, parts[w].data(), parts[w].size());
      }
    };
    gather_block(R->pval.data(), 8);
    gather_block(correct.data(), 1);
    gather_block(nsubs.data(), 4);
    {
      std::vector<int64_t> v(R->subqual.size() + qsum.size() + qn.size());
      size_t o = 0;
      for (int32_t x : R->subqual) v[o++] = x;
      for (unsigned long long x : qsum) v[o++] = (int64_t)x;
      for (uint32_t x : qn) v[o++] = x;
      run.sh_allreduce(v);
      o = 0;
      for (auto &x : R->subqual) x = (int32_t)v[o++];          // (int32 wrap-around as R's integer matrix, error.cpp:152-167)
      for (auto &x : qsum) x = (unsigned long long)v[o++];
      for (auto &x : qn) x = (uint32_t)v[o++];
    }
    {   // post-hoc comparisons (partition of a centre, partition compared, lambda) found in each block's store
      struct Ph { int32_t j, i; double lam; };
      std::vector<Ph> mine(ph_lam.size());
      for (size_t k = 0; k < mine.size(); k++) mine[k] = Ph{ph_ji[2 * k], ph_ji[2 * k + 1], ph_lam[k]};
      ph_ji.clear(); ph_lam.clear();
      for (auto &part : run.sh_gatherv(mine.data(), mine.size() * sizeof(Ph)))
        for (size_t k = 0; k + sizeof(Ph) <= part.size(); k += sizeof(Ph)) {
          Ph e;
          memcpy(&e, part.data() + k, sizeof(Ph));
          ph_ji.push_back(e.j); ph_ji.push_back(e.i); ph_lam.push_back(e.lam);
        }
    }
    {   // work counters: every rank counted its own block's comparisons
      std::vector<int64_t> v = {(int64_t)run.st.nnw, (int64_t)run.st.ngapless, (int64_t)run.st.nshroud, (int64_t)run.st.nskipped,
                                (int64_t)run.st.nstored};
      run.sh_allreduce(v);
      run.st.nnw = (uint64_t)v[0]; run.st.ngapless = (uint64_t)v[1]; run.st.nshroud = (uint64_t)v[2]; run.st.nskipped = (uint64_t)v[3];
      run.st.nstored = (uint64_t)v[4];
      run.st.nw_cells = run.st.nnw * run.nw_cells_per_alignment();
      run.st.screen_bytes = (run.st.ncompare - run.st.nskipped) * 2 * (uint64_t)(D.maxlen - KMER_SIZE + 1) + run.st.ncompare * 6;
    }
    run.sh_points_left = false;   // (what follows - the assembly of the outputs - exchanges nothing)
  }

  // ---- assemble the six outputs -------------------------------------------------------------------
  R->nclust = C; R->nraw = N; R->maxlen = D.maxlen; R->ncol = err_ncol;
  R->sequence.resize(C);
  R->abundance.assign(C, 0); R->n0.assign(C, 0); R->n1.assign(C, 0); R->nunq.assign(C, 0);
  R->birth_from.assign(C, 0); R->birth_ham.assign(C, 0); R->center.assign(C, 0);
  R->clust_pval.assign(C, 0); R->birth_pval.assign(C, 0); R->birth_fold.assign(C, 0); R->birth_qave.assign(C, 0);
  // b_make_clustering_df (error.cpp:9-127).  The sums over a partition's members are integer sums and the representative
  // sequence is the member with the most reads, the FIRST such in list order (error.cpp:19-34): one streaming pass over the
  // uniques in index order with the host mirror's partition / slot of each (the per-partition member loops were 10^6 random
  // reads into three arrays) gives the same values
  R->map.assign(N, DADA2HIP_NA_INTEGER);   // Rmain.cpp:268-279
  {
    std::vector<uint32_t> max_reads(C, 0);
    std::vector<int32_t> max_raw(C, -1), max_slot(C, 0);
    for (int r = 0; r < N; r++) {
      const int i = run.clust_of[r];
      const uint32_t rd = s->h_reads[r];
      if (rd > max_reads[i] || (rd == max_reads[i] && max_raw[i] >= 0 && run.slot_of[r] < max_slot[i])) {
        max_reads[i] = rd; max_raw[i] = r; max_slot[i] = run.slot_of[r];
      }
      if (!correct[r]) continue;
      R->map[r] = i + 1;
      R->abundance[i] += (int32_t)rd;
      R->nunq[i]++;
      if (nsubs[r] == 0) R->n0[i] += (int32_t)rd;
      if (nsubs[r] == 1) R->n1[i] += (int32_t)rd;
    }
    for (int i = 0; i < C; i++) R->sequence[i] = max_raw[i] >= 0 ? seq_string(s, max_raw[i]) : std::string("");
  }
  for (int i = 0; i < C; i++) {
    const Bi &b = run.bi[i];
    R->center[i] = (int32_t)b.center;
    if (i == 0) {
      R->birth_pval[i] = na_real(); R->birth_from[i] = DADA2HIP_NA_INTEGER; R->birth_fold[i] = na_real();
      R->birth_ham[i] = DADA2HIP_NA_INTEGER; R->birth_qave[i] = na_real();
    } else {
      R->birth_from[i] = (int32_t)b.birth_from + 1;
      R->birth_pval[i] = b.birth_pval; R->birth_fold[i] = b.birth_fold; R->birth_ham[i] = (int32_t)b.birth_comp.hamming;
    }
  }
  {   // post-hoc p-value (error.cpp:101-119): tot_e[j] += lambda * bi[i].reads in ascending i
    std::vector<double> tot_e(C, 0.0);
    std::vector<int32_t> order(ph_lam.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = (int32_t)k;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
      if (ph_ji[2 * a] != ph_ji[2 * b]) return ph_ji[2 * a] < ph_ji[2 * b];
      return ph_ji[2 * a + 1] < ph_ji[2 * b + 1];
    });
    for (int k : order) tot_e[ph_ji[2 * k]] += ph_lam[k] * run.bi[ph_ji[2 * k + 1]].reads;
    for (int i = 0; i < C; i++) R->clust_pval[i] = pp::calc_pA((int)s->h_reads[run.bi[i].center], tot_e[i], true);
  }
  // birth_subs data.frame (error.cpp:261-300) + birth_qave (error.cpp:83-92)
  for (int i = 1; i < C; i++) {
    const uint32_t pc = run.bi[run.bi[i].birth_comp.i].center;
    const std::string cs = seq_string(s, (int)pc);
    double q_ave = 0.0;
    int nsb = 0;
    for (int p0 = 0; p0 < (int)cs.size(); p0++) {
      const uint16_t v = bview[(size_t)i * LV + p0];
      if (!(v & 0x8000u)) continue;
      const char rb = "ACGT"[(v >> 8) & 3];
      if (rb != cs[p0]) {
        R->bs_pos.push_back(p0 + 1); R->bs_ref.push_back(cs[p0]); R->bs_sub.push_back(rb);
        R->bs_qual.push_back((double)(v & 255u)); R->bs_clust.push_back(i + 1);
        q_ave += (double)(v & 255u);
        nsb++;
      }
    }
    R->birth_qave[i] = q_ave / ((double)nsb);
  }
  // cluster quality matrix (error.cpp:225-258)
  R->clusterquals.assign((size_t)D.maxlen * C, 0.0);
  for (int i = 0; i < C; i++) {
    const int clen = s->h_len[run.bi[i].center];
    for (int p0 = 0; p0 < clen; p0++)
      R->clusterquals[(size_t)i * D.maxlen + p0] = ((double)qsum[(size_t)i * D.maxlen + p0]) / qn[(size_t)i * D.maxlen + p0];
    for (int p0 = clen; p0 < D.maxlen; p0++) R->clusterquals[(size_t)i * D.maxlen + p0] = na_real();
  }
  run.st.ms_final = ms_since(t_final);
  run.st.ms_total = ms_since(t_total);
  R->stats = run.st;
  run.logf("\nALIGN: %llu aligns, %llu shrouded (%d raw).\n", (unsigned long long)(run.st.ncompare - run.st.nskipped),
           (unsigned long long)run.st.nshroud, N);
}

}  // namespace

// =================================================================================================
extern "C" {

const char *dada2hip_version(void) { return "dada2hip 0.1.0 (gfx950)"; }

int dada2hip_sample_create(int32_t nraw, const char *const *seqs, const int32_t *abundances, const uint8_t *priors,
                           const double *quals, int32_t quals_nrow, int32_t device, dada2hip_sample **out, char *errbuf,
                           size_t errlen) {
  if (out) *out = nullptr;
  dada2hip_sample *s = new dada2hip_sample();
  int rc = guarded(errbuf, errlen, [&] { sample_create(s, nraw, seqs, abundances, priors, quals, quals_nrow, device); });
  if (rc != DADA2HIP_OK) { dada2hip_sample_free(s); return rc; }
  *out = s;
  return rc;
}

int dada2hip_sample_set_priors(dada2hip_sample *s, const uint8_t *priors, char *errbuf, size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    select_device(s->device);
    for (int i = 0; i < s->D.N; i++) s->h_prior[i] = priors && priors[i] ? 1 : 0;
    D2_HIP(hipMemcpy(s->D.prior, s->h_prior.data(), (size_t)s->D.N, hipMemcpyHostToDevice));
  });
}

int dada2hip_sample_run_sharded(dada2hip_sample *s, const double *err, int32_t err_ncol, const dada2hip_opts *opts,
                                const dada2hip_hooks *hooks, const dada2hip_shard *shard, dada2hip_result **out,
                                char *errbuf, size_t errlen) {
  if (out) *out = nullptr;
  dada2hip_result *R = new dada2hip_result();
  int rc = guarded(errbuf, errlen, [&] {
    if (!shard) throw InputError{"dada2hip: invalid shard descriptor."};
    try {
      sample_run(s, err, err_ncol, opts, hooks, R, shard);
    } catch (const PeerFailed &) {
      throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "dada2hip: another rank of the sharded run failed"};
    } catch (...) {
      // tell the other ranks at their next exchange point (each opens with an 8-byte all-gather of a size: -1 = failed), so
      // that they return an error too instead of blocking in a collective this rank will never join - unless the
      // run's last exchange point is behind this rank: its peers are done exchanging, a notification would wait alone (ADVICE
      // r4).  When the failure IS the exchange callback the notification is still tried, once: a callback that failed before it
      // entered the collective (the case the tests inject) leaves the peers waiting in it, and this is what releases them; a
      // callback whose transport is gone fails again, which is ignored.  Either way a collective that never completes is the
      // transport's to time out (include/dada2hip.h, dada2hip_shard).
      const Run *rr = s->run_cache ? static_cast<const Run *>(s->run_cache.get()) : nullptr;
      const bool notify = !rr || rr->sh_points_left;
      if (notify && shard->world > 1 && shard->exchange) {
        const int64_t failed = -1;
        std::vector<int64_t> all((size_t)shard->world);
        (void)shard->exchange(shard->user, 0, &failed, 8, all.data());
      }
      throw;
    }
  });
  if (rc != DADA2HIP_OK) { delete R; return rc; }
  *out = R;
  return rc;
}

void dada2hip_sample_free(dada2hip_sample *s) {
  if (!s) return;
  if (s->stream) { (void)hipSetDevice(s->device); (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
  if (s->cmp) { (void)hipStreamSynchronize(s->cmp); (void)hipStreamDestroy(s->cmp); }
  if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); }   // (its buffers return to the allocation cache)
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  delete s;
}

int32_t dada2hip_sample_nraw(const dada2hip_sample *s) { return s ? s->D.N : 0; }
int32_t dada2hip_sample_maxlen(const dada2hip_sample *s) { return s ? s->D.maxlen : 0; }

int dada2hip_sample_run(dada2hip_sample *s, const double *err, int32_t err_ncol, const dada2hip_opts *opts,
                        const dada2hip_hooks *hooks, dada2hip_result **out, char *errbuf, size_t errlen) {
  if (out) *out = nullptr;
  dada2hip_result *R = new dada2hip_result();
  int rc = guarded(errbuf, errlen, [&] { sample_run(s, err, err_ncol, opts, hooks, R); });
  if (rc != DADA2HIP_OK) { delete R; return rc; }
  *out = R;
  return rc;
}

int dada2hip_dada_uniques(int32_t nraw, const char *const *seqs, const int32_t *abundances, const uint8_t *priors,
                          const double *err, int32_t err_ncol, const double *quals, int32_t quals_nrow,
                          const dada2hip_opts *opts, int32_t device, const dada2hip_hooks *hooks, dada2hip_result **out,
                          char *errbuf, size_t errlen) {
  if (out) *out = nullptr;
  if (nraw > 0 && seqs && abundances && priors == nullptr) { /* priors optional */ }
  dada2hip_sample *s = nullptr;
  int rc = dada2hip_sample_create(nraw, seqs, abundances, priors, quals, quals_nrow, device, &s, errbuf, errlen);
  if (rc != DADA2HIP_OK) return rc;
  rc = dada2hip_sample_run(s, err, err_ncol, opts, hooks, out, errbuf, errlen);
  dada2hip_sample_free(s);
  return rc;
}

// ---- batch form: the sample loop of R/dada.R:266-366 spread over the GPUs of the node ------------------------
// One host thread per entry of device_ids (the same ordinal may appear twice: two streams of samples on one GPU);
// thread t runs samples t, t + n_devices, ... in order.  Samples are independent given err, so there is
// no exchange between the threads; the caller sums the $subqual matrices afterwards (accumulateTrans,
// R/errorModels.R:462-471).
int dada2hip_run_multi(int32_t n_samples, const dada2hip_sample_input *samples, const double *err, int32_t err_ncol,
                       const dada2hip_opts *opts, int32_t n_devices, const int32_t *device_ids, dada2hip_result **out,
                       char *errbuf, size_t errlen) {
  if (n_samples < 0 || (n_samples > 0 && (!samples || !out))) { set_err(errbuf, errlen, "dada2hip: bad sample list"); return DADA2HIP_ERR_INPUT; }
  for (int i = 0; i < n_samples; i++) out[i] = nullptr;
  if (n_samples == 0) return DADA2HIP_OK;
  const int32_t dev0 = 0;
  if (n_devices <= 0 || !device_ids) { n_devices = 1; device_ids = &dev0; }
  const int nthr = std::min(n_devices, n_samples);
  std::vector<int> rcs(n_samples, DADA2HIP_OK);
  std::vector<std::string> msgs(n_samples);
  std::atomic<int> failed{0};
  auto worker = [&](int t) {
    for (int i = t; i < n_samples; i += nthr) {
      if (failed.load()) return;                        // one sample failed: R would have stopped the whole call
      char eb[1024];
      eb[0] = 0;
      const dada2hip_sample_input &in = samples[i];
      rcs[i] = dada2hip_dada_uniques(in.nraw, in.seqs, in.abundances, in.priors, err, err_ncol, in.quals, in.quals_nrow,
                                     opts, device_ids[t], nullptr, &out[i], eb, sizeof eb);
      if (rcs[i] != DADA2HIP_OK) { msgs[i] = eb; failed.store(1); }
    }
  };
  // A device that several of the call's threads share: every sample on it runs without the prefetch compares of DESIGN.md 5c, from
  // its first round on - not just the samples that happen to start while another one is active (active_runs() is what v3_setup and
  // every persistent launch ask).  The first sample of each wave used to start alone, plan prefetches and park their gate kernels
  // on a second stream, which then shared the runtime's hardware queues with the next sample's streams: four samples of 60 k
  // uniques, two in flight, read 81-115 ms in eight of ten fresh processes and 157 / 306 ms in the other two
  // (tests/test_gpu_scale_and_edges.py::test_several_samples_on_one_gpu_take_the_same_time_run_after_run).
  std::vector<int> shared_devs;
  for (int t = 0; t < nthr; t++) {
    int same = 0;
    for (int u = 0; u < nthr; u++) same += device_ids[u] == device_ids[t];
    if (same >= 2 && std::find(shared_devs.begin(), shared_devs.end(), (int)device_ids[t]) == shared_devs.end()) shared_devs.push_back((int)device_ids[t]);
  }
  for (int d : shared_devs) active_runs(d)++;
  std::vector<std::thread> th;
  for (int t = 1; t < nthr; t++) th.emplace_back(worker, t);
  worker(0);
  for (auto &x : th) x.join();
  for (int d : shared_devs) active_runs(d)--;
  for (int i = 0; i < n_samples; i++)
    if (rcs[i] != DADA2HIP_OK) {
      char m[1200];
      snprintf(m, sizeof m, "sample %d: %s", i + 1, msgs[i].c_str());
      set_err(errbuf, errlen, m);
      for (int k = 0; k < n_samples; k++) { if (out[k]) dada2hip_result_free(out[k]); out[k] = nullptr; }
      return rcs[i];
    }
  return DADA2HIP_OK;
}

void dada2hip_trim_cache(void) { AllocCache::get().trim(); }

int dada2hip_sample_compare(dada2hip_sample *s, int32_t centre, const double *err, int32_t err_ncol,
                            const dada2hip_opts *opts, double kdist_cutoff, const uint8_t *skip, double *lambda,
                            uint32_t *hamming, uint8_t *cls, dada2hip_stats *stats, char *errbuf, size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    select_device(s->device);
    if (centre < 0 || centre >= s->D.N) throw InputError{"dada2hip: centre out of range"};
    check_opts(*opts, s->qmax, err_ncol);
    ensure_ad_ring(s);
    AdRingGuard ring_guard{s};
    Run run;
    run.hooks = nullptr;
    init_run(run, s, err, err_ncol, opts, kdist_cutoff);
    SampleDev &D = s->D;
    hipStream_t stq = s->stream;
    const int N = D.N;
    if (skip) D2_HIP(hipMemcpyAsync(s->d_skip.p, skip, (size_t)N, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(s->d_thresh.p, run.thresh_round.data(), run.thresh_round.size() * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemsetAsync(s->d_counters.p, 0, 8 * 4, stq));
    D2_HIP(hipEventRecord(s->ev0, stq));
    launch_screen(D, centre, run.sp, skip ? s->d_skip.p : nullptr, nullptr, 0, s->d_thresh.p, s->d_cls.p, s->d_lambda.p,
                  s->d_ham.p, s->d_nw_list.p, s->d_gl_list.p, s->d_counters.p, s->d_ctab.p, true, nullptr, stq);
    D2_HIP(hipEventRecord(s->ev1, stq));
    launch_gapless(D, centre, nullptr, s->d_gl_list.p, s->d_counters.p + 1, 0, run.ap, s->d_err.p, s->d_lambda.p, s->d_ham.p,
                   nullptr, 0, 0, stq);
    D2_HIP(hipMemcpyAsync(s->h_counters.p, s->d_counters.p, 16, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipStreamSynchronize(stq));
    float ems = 0;
    D2_HIP(hipEventElapsedTime(&ems, s->ev0, s->ev1));
    run.st.screen_kernel_ms = ems;
    run.st.screen_kernel_launches = 1;
    const int n_nw = s->h_counters.p[0];
    run.st.ncompare = (uint64_t)N; run.st.nnw = (uint64_t)n_nw; run.st.ngapless = (uint64_t)s->h_counters.p[1];
    {
      std::vector<uint8_t> hc(N);
      D2_HIP(hipMemcpy(hc.data(), s->d_cls.p, (size_t)N, hipMemcpyDeviceToHost));
      for (int i = 0; i < N; i++) { run.st.nshroud += hc[i] == CLS_SHROUD; run.st.nskipped += hc[i] == CLS_SKIP; }
    }
    if (n_nw > 0) {
      const int f = knobs().nw_kernel;
      const bool coop_ok = nw_ad_lds_bytes(D, run.ap) > 0 && nw_ad_lds_bytes(D, run.ap) <= 150 * 1024;
      bool coop = coop_ok && n_nw < 65536;
      if (f == NWK_LANE) coop = false;
      if (f == NWK_COOP && coop_ok) coop = true;
      D2_HIP(hipEventRecord(s->ev0, stq));
      if (coop)
        launch_nw_ad(D, centre, nullptr, s->d_nw_list.p, nullptr, n_nw, nullptr, nullptr, run.ap, s->d_err.p, s->d_lambda.p,
                     s->d_ham.p, nullptr, 0, 0, nullptr, stq);
      else {
        ensure_scratch(s, run.ap.band);
        launch_nw(D, run.wclass, centre, nullptr, s->d_nw_list.p, nullptr, n_nw, run.ap, s->d_err.p, s->scr, s->d_lambda.p,
                  s->d_ham.p, nullptr, 0, 0, nullptr, 0, nullptr, stq);
      }
      D2_HIP(hipEventRecord(s->ev1, stq));
      D2_HIP(hipStreamSynchronize(stq));
      D2_HIP(hipEventElapsedTime(&ems, s->ev0, s->ev1));
      run.st.nw_kernel_ms = ems;
      run.st.nw_kernel_launches = 1;
      run.st.nw_cells = (uint64_t)n_nw * run.nw_cells_per_alignment();
    }
    launch_fill_null(N, s->d_cls.p, s->d_lambda.p, s->d_ham.p, stq);
    D2_HIP(hipStreamSynchronize(stq));
    check_nw_flag(s);
    if (lambda) D2_HIP(hipMemcpy(lambda, s->d_lambda.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (hamming) D2_HIP(hipMemcpy(hamming, s->d_ham.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (cls) D2_HIP(hipMemcpy(cls, s->d_cls.p, (size_t)N, hipMemcpyDeviceToHost));
    D2_HIP(hipGetLastError());
    if (stats) *stats = run.st;
  });
}

int dada2hip_calc_pA(int32_t n, const int32_t *reads, const double *E_reads, const uint8_t *prior, int32_t device,
                     double *out, char *errbuf, size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    select_device(device);
    if (n <= 0) return;
    DevBuf<int32_t> dr; DevBuf<double> dE, dout; DevBuf<uint8_t> dp;
    dr.alloc(n); dE.alloc(n); dout.alloc(n); dp.alloc(n);
    D2_HIP(hipMemcpy(dr.p, reads, (size_t)n * 4, hipMemcpyHostToDevice));
    D2_HIP(hipMemcpy(dE.p, E_reads, (size_t)n * 8, hipMemcpyHostToDevice));
    if (prior) D2_HIP(hipMemcpy(dp.p, prior, (size_t)n, hipMemcpyHostToDevice));
    launch_calc_pA(n, dr.p, dE.p, prior ? dp.p : nullptr, dout.p, nullptr);
    D2_HIP(hipDeviceSynchronize());
    D2_HIP(hipGetLastError());
    D2_HIP(hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  });
}

// ---- pairwise alignment exports (C_nwalign evaluate.cpp:18 / C_nwvec nwalign_vectorized.cpp:321) ---
// vec_semantics: the call is C_nwvec (raw-byte comparison, any letters); else C_nwalign (nt2int codes: ACGT only)
static int nwvec_any(int32_t n, const char *const *s1, const char *const *s2, int32_t match, int32_t mismatch,
                     int32_t gap_p, int32_t homo_gap_p, int32_t band, int32_t endsfree, int32_t device, char *const *out, char *errbuf,
                     size_t errlen, bool vec_semantics = false) {
  return guarded(errbuf, errlen, [&] {
    if (n <= 0) return;
    // a throw-away resident sample holding the 2n strings; pair i = (centre 2i, raw 2i+1), one pair per LANE: the lane kernels
    // take a centre per work item (NwArgs::pair_centre), so 64 unrelated pairs share a wave and the move strings take
    // n x (2 maxlen + 2) bytes (round 2 gave every pair a whole 64-slot chunk: 64x the memory)
    std::vector<const char *> seqs(2 * (size_t)n);
    int maxlen = 0;
    bool acgt = true;
    for (int i = 0; i < n; i++) {
      seqs[2 * i] = s1[i]; seqs[2 * i + 1] = s2[i];
      maxlen = std::max<int>(maxlen, (int)std::max(strlen(s1[i]), strlen(s2[i])));
    }
    for (int i = 0; i < 2 * n && acgt; i++)
      for (const char *c = seqs[i]; *c; c++)
        if (*c != 'A' && *c != 'C' && *c != 'G' && *c != 'T') { acgt = false; break; }
    // Letters outside ACGT.  C_nwalign has no defined behaviour for them (nt2int maps N to 5 and the aligners index a 4 x 4
    // score table with it, evaluate.cpp:28-33); C_nwvec has: it compares the strings' raw BYTES (nwalign_vectorized.cpp:165), so
    // only the equality pattern of a pair's letters matters.  The letters of each pair are renumbered 0..15 in order of first
    // appearance and stored as two 2-bit planes (AlignParams::hi_off); more than 16 distinct bytes in one pair is not DNA.
    const bool planes = !acgt;
    if (planes && (!vec_semantics || homo_gap_p != gap_p))
      throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED, "dada2hip: nwalign on the device takes A/C/G/T only (the reference's own behaviour for N / IUPAC codes is undefined there: evaluate.cpp:28-33)."};
    std::vector<std::string> coded;          // planes: [0, 2n) low planes, [2n, 4n) high planes, as ACGT strings
    if (planes) {
      coded.resize(4 * (size_t)n);
      for (int i = 0; i < n; i++) {
        int code_of[256];
        for (int &c : code_of) c = -1;
        int ncodes = 0;
        for (int w = 0; w < 2; w++) {
          const char *str = seqs[2 * i + w];
          const size_t len = strlen(str);
          std::string &lo = coded[2 * i + w], &hi = coded[2 * (size_t)n + 2 * i + w];
          lo.resize(len); hi.resize(len);
          for (size_t p = 0; p < len; p++) {
            int &c = code_of[(unsigned char)str[p]];
            if (c < 0) {
              if (ncodes == 16) throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED, "dada2hip: nwvec: more than 16 distinct letters in one pair of strings."};
              c = ncodes++;
            }
            lo[p] = "ACGT"[c & 3]; hi[p] = "ACGT"[c >> 2];
          }
        }
      }
      seqs.resize(4 * (size_t)n);
      for (size_t k = 0; k < coded.size(); k++) seqs[k] = coded[k].c_str();
    }
    const int nrows = (int)seqs.size();
    std::vector<int32_t> ab((size_t)nrows, 1);
    dada2hip_sample *s = new dada2hip_sample();
    std::unique_ptr<dada2hip_sample, void (*)(dada2hip_sample *)> guard(s, dada2hip_sample_free);
    sample_create(s, nrows, seqs.data(), ab.data(), nullptr, nullptr, 0, device, /*lite=*/true);
    ensure_scratch(s, band, 0, /*force_generic=*/planes);
    std::vector<double> errm(16, 1.0), rowm;
    upload_err(s, errm.data(), 1, rowm);
    dada2hip_opts o;
    memset(&o, 0, sizeof o);
    o.match = match; o.mismatch = mismatch; o.gap = gap_p; o.vectorized_alignment = 1;
    AlignParams ap{match, mismatch, gap_p, band, nw_sentinel(o), 0, 1};
    // C_nwalign (evaluate.cpp:36-48): nwalign_endsfree[_homo] when endsfree, the global nwalign (one gap penalty) otherwise;
    // the scalar aligners mark the band edge with -9999 (nwalign_endsfree.cpp:113-119)
    ap.endsfree = endsfree ? 1 : 0;
    ap.homo_gap = endsfree ? homo_gap_p : gap_p;
    if (!ap.plain()) ap.sentinel = -9999;
    ap.hi_off = planes ? 2 * n : 0;
    std::vector<int32_t> work((size_t)n), cc(n);
    for (int i = 0; i < n; i++) { work[i] = 2 * i + 1; cc[i] = 2 * i; }
    const int stride = 2 * maxlen + 2;
    s->d_work.alloc(work.size()); s->d_chunk_centre.alloc(n); s->d_moves.alloc(work.size() * (size_t)stride);
    s->d_nmoves.alloc(work.size()); s->d_lambda.alloc(2 * (size_t)n); s->d_ham.alloc(2 * (size_t)n);
    D2_HIP(hipMemcpyAsync(s->d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, s->stream));
    D2_HIP(hipMemcpyAsync(s->d_chunk_centre.p, cc.data(), (size_t)n * 4, hipMemcpyHostToDevice, s->stream));
    if (band == 0) throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED, "dada2hip: band == 0 is the gapless pairing, not an NW call."};
    launch_nw(s->D, s->scr_class, 0, nullptr, s->d_work.p, nullptr, (int)work.size(), ap, s->d_err.p, s->scr,
              s->d_lambda.p, s->d_ham.p, nullptr, 0, 0, s->d_moves.p, stride, s->d_nmoves.p, s->stream, s->d_chunk_centre.p);
    std::vector<uint8_t> moves(work.size() * (size_t)stride);
    std::vector<int32_t> nm(work.size());
    D2_HIP(hipMemcpyAsync(moves.data(), s->d_moves.p, moves.size(), hipMemcpyDeviceToHost, s->stream));
    D2_HIP(hipMemcpyAsync(nm.data(), s->d_nmoves.p, nm.size() * 4, hipMemcpyDeviceToHost, s->stream));
    D2_HIP(hipStreamSynchronize(s->stream));
    D2_HIP(hipGetLastError());
    check_nw_flag(s);
    for (int i = 0; i < n; i++) {
      const uint8_t *mv = &moves[(size_t)i * stride];
      const int len = nm[i];
      int a = (int)strlen(s1[i]), b = (int)strlen(s2[i]);
      char *o0 = out[2 * i], *o1 = out[2 * i + 1];
      for (int t = 0; t < len; t++) {   // moves were recorded from the end of the alignment backwards
        const int pos = len - 1 - t;
        const uint8_t p = mv[t];
        if (p == 1) { o0[pos] = s1[i][--a]; o1[pos] = s2[i][--b]; }
        else if (p == 2) { o0[pos] = '-'; o1[pos] = s2[i][--b]; }
        else { o0[pos] = s1[i][--a]; o1[pos] = '-'; }
      }
      o0[len] = 0; o1[len] = 0;
    }
  });
}

int dada2hip_nwvec(int32_t n, const char *const *s1, const char *const *s2, int32_t match, int32_t mismatch,
                   int32_t gap_p, int32_t band, int32_t endsfree, int32_t device, char *const *out, char *errbuf,
                   size_t errlen) {
  return nwvec_any(n, s1, s2, match, mismatch, gap_p, gap_p, band, endsfree, device, out, errbuf, errlen, /*vec_semantics=*/true);
}

// ---- bimera identification (chimera.cpp): the step after dada() ---------------------------------------------------
namespace {
struct BimPair { int32_t left, right, left_oo, right_oo, ham; };

// Aligns every (query j, parent k) pair the table asks for on the device and returns, per query, its parents with the
// get_lr / get_ham_endsfree results.  parents[j] lists the k's (ascending); res[j][t] belongs to parents[j][t].
void bimera_pairs(int ncol, const char *const *seqs, const std::vector<std::vector<int32_t>> &parents, int match, int mismatch,
                  int gap_p, int max_shift, int allow_one_off, int device, std::vector<std::vector<BimPair>> &res) {
  const bool times = knobs().bimera_times;
  auto t0 = clk::now();
  double ms_dev = 0, ms_unpack = 0, ms_pack = 0;
  res.assign(ncol, {});
  size_t npairs = 0;
  for (int j = 0; j < ncol; j++) { res[j].resize(parents[j].size()); npairs += parents[j].size(); }
  if (!npairs) return;
  for (int i = 0; i < ncol; i++)
    for (const char *c = seqs[i]; *c; c++)
      if (*c != 'A' && *c != 'C' && *c != 'G' && *c != 'T')
        throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED, "dada2hip: bimera identification on the device takes A/C/G/T only."};
  std::vector<int32_t> ab(ncol, 1);
  dada2hip_sample *s = new dada2hip_sample();
  std::unique_ptr<dada2hip_sample, void (*)(dada2hip_sample *)> guard(s, dada2hip_sample_free);
  sample_create(s, ncol, seqs, ab.data(), nullptr, nullptr, 0, device, /*lite=*/true);
  if (max_shift == 0) throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED, "dada2hip: maxShift 0 is outside the implemented path."};
  std::vector<double> errm(16, 1.0), rowm;
  upload_err(s, errm.data(), 1, rowm);
  dada2hip_opts o;
  memset(&o, 0, sizeof o);
  o.match = match; o.mismatch = mismatch; o.gap = gap_p; o.vectorized_alignment = 1;   // nwalign_vectorized2 (chimera.cpp:26)
  AlignParams ap{match, mismatch, gap_p, max_shift, nw_sentinel(o), 0, 1};
  ap.homo_gap = gap_p;   // (plain ends-free alignment: chimera.cpp:26,122 call nwalign_vectorized2 / nwalign_endsfree)
  const int stride = 2 * s->D.maxlen + 2;
  // The aligner: the anti-diagonal kernel of the denoising path in its bimera mode (k_nw_ad<.., LR>: the alignment is reduced
  // to get_lr / get_ham_endsfree inside the kernel, nothing but five words per pair leaves it) wherever its geometry applies
  // (reads <= 2 047 nt, band window <= 127 cells); otherwise - and under DADA2HIP_NW_KERNEL=lane|wide, the tests' way of
  // running both - the lane kernel with its move strings kept and k_bimera_lr behind it.
  const bool coop = [&] {
    const int f = knobs().nw_kernel;
    if (f == NWK_LANE || f == NWK_WIDE) return false;
    const size_t b = nw_ad_lr_lds_bytes(s->D, ap);
    return b > 0 && b <= 150 * 1024;
  }();
  AdRingGuard ring{s};
  if (coop) ensure_ad_ring(s);
  else ensure_scratch(s, max_shift, npairs);
  // batches of whole queries, each query's parents padded to the kernel's chunks (alignments that share their query)
  const size_t per = coop ? (size_t)nw_ad_apw(s->D, ap) : 64;
  const size_t budget_slots = coop ? ((size_t)1 << 23) : std::max<size_t>(4096, ((size_t)256 << 20) / (size_t)stride);
  s->d_lambda.alloc(ncol); s->d_ham.alloc(ncol);
  DevBuf<int32_t> d_out;
  PinBuf<int32_t> h_out;
  std::vector<int32_t> work, cc;
  std::vector<std::pair<int, size_t>> where;   // (query, first slot) of the batch
  hipStream_t stq = s->stream;
  const double ms_setup = ms_since(t0);
  int j = 0, nbatch = 0;
  while (j < ncol) {
    auto tb = clk::now();
    // the batch's queries and their first slots, then the lists filled by the host pool (4.4 M slots at 3 000 sequences)
    where.clear();
    size_t nslots = 0;
    while (j < ncol && (nslots == 0 || nslots + parents[j].size() + per <= budget_slots)) {
      if (!parents[j].empty()) {
        where.push_back({j, nslots});
        nslots += (parents[j].size() + per - 1) / per * per;
      }
      j++;
    }
    work.resize(nslots); cc.resize(nslots / per);
    parallel_for(where.size(), 16, [&](size_t lo, size_t hi) {
      for (size_t w = lo; w < hi; w++) {
        const int q = where[w].first;
        const size_t first = where[w].second, np = parents[q].size(), padded = (np + per - 1) / per * per;
        memcpy(&work[first], parents[q].data(), np * sizeof(int32_t));
        for (size_t t = np; t < padded; t++) work[first + t] = -1;
        for (size_t c = 0; c < padded / per; c++) cc[first / per + c] = q;
      }
    });
    if (work.empty()) continue;
    const int nwork = (int)work.size();
    nbatch++;
    ms_pack += ms_since(tb); tb = clk::now();
    s->d_work.alloc(work.size()); s->d_chunk_centre.alloc(cc.size()); d_out.alloc(work.size() * 5);
    h_out.alloc(work.size() * 5);
    D2_HIP(hipMemcpyAsync(s->d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(s->d_chunk_centre.p, cc.data(), cc.size() * 4, hipMemcpyHostToDevice, stq));
    if (coop)
      launch_nw_ad_lr(s->D, s->d_chunk_centre.p, s->d_work.p, nwork, ap, s->d_err.p, allow_one_off, max_shift, d_out.p, stq);
    else {
      s->d_moves.alloc(work.size() * (size_t)stride); s->d_nmoves.alloc(work.size());
      launch_nw(s->D, s->scr_class, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, nwork, ap, s->d_err.p, s->scr, s->d_lambda.p, s->d_ham.p,
                nullptr, 0, 0, s->d_moves.p, stride, s->d_nmoves.p, stq);
      launch_bimera_lr(s->D, s->d_chunk_centre.p, s->d_work.p, nwork, s->d_moves.p, stride, s->d_nmoves.p, allow_one_off, max_shift,
                       d_out.p, stq);
    }
    D2_HIP(hipMemcpyAsync(h_out.p, d_out.p, work.size() * 5 * 4, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipStreamSynchronize(stq));
    D2_HIP(hipGetLastError());
    check_nw_flag(s);
    ms_dev += ms_since(tb); tb = clk::now();
    parallel_for(where.size(), 16, [&](size_t lo, size_t hi) {
      for (size_t w = lo; w < hi; w++) {
        const int q = where[w].first;
        for (size_t t = 0; t < parents[q].size(); t++) {
          const int32_t *o5 = h_out.p + (where[w].second + t) * 5;
          res[q][t] = BimPair{o5[0], o5[1], o5[2], o5[3], o5[4]};
        }
      }
    });
    ms_unpack += ms_since(tb);
  }
  if (times)
    fprintf(stderr, "[bimera] pairs %zu in %d batches (%s): setup %.1f ms, work lists %.1f ms, device (alloc + H2D + kernels + D2H) %.1f ms, unpack %.1f ms\n",
            npairs, nbatch, coop ? "k_nw_ad, bimera mode" : "k_nw + k_bimera_lr", ms_setup, ms_pack, ms_dev, ms_unpack);
}
}  // namespace

// C_table_bimera2 (chimera.cpp:61-208).  mat: nrow (samples) x ncol (sequences), column-major as R's IntegerMatrix.
int dada2hip_table_bimera2(int32_t nrow, int32_t ncol, const int32_t *mat, const char *const *seqs, double min_fold, int32_t min_abund,
                           int32_t allow_one_off, int32_t min_one_off_par_dist, int32_t match, int32_t mismatch, int32_t gap_p,
                           int32_t max_shift, int32_t device, int32_t *nflag, int32_t *nsam, char *errbuf, size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    if (nrow < 0 || ncol < 0 || (ncol > 0 && (!mat || !seqs || !nflag || !nsam))) throw InputError{"dada2hip: bad sequence table"};
    if (ncol == 0) return;
    select_device(device);
    // which (query, parent) pairs does the table ask for?  (chimera.cpp:117-118: a parent is more abundant than the query
    // by min_fold and at least min_abund in some sample where the query is present)
    const bool times = knobs().bimera_times;
    auto t0 = clk::now();
    std::vector<std::vector<int32_t>> parents(ncol);
    parallel_for((size_t)ncol, 8, [&](size_t lo, size_t hi) {
      std::vector<uint8_t> need(ncol);
      for (size_t j = lo; j < hi; j++) {
        std::fill(need.begin(), need.end(), 0);
        for (int i = 0; i < nrow; i++) {
          const int vj = mat[i + j * (size_t)nrow];
          if (vj <= 0) continue;
          for (int k = 0; k < ncol; k++) {
            const int vk = mat[i + (size_t)k * nrow];
            if (vk > (min_fold * vj) && vk >= min_abund) need[k] = 1;
          }
        }
        for (int k = 0; k < ncol; k++) if (need[k]) parents[j].push_back(k);
      }
    });
    const double ms_parents = ms_since(t0);
    auto t1 = clk::now();
    std::vector<std::vector<BimPair>> res;
    bimera_pairs(ncol, seqs, parents, match, mismatch, gap_p, max_shift, allow_one_off, device, res);
    const double ms_pairs = ms_since(t1);
    t1 = clk::now();
    // per sequence and sample: is there a two-parent model?  (chimera.cpp:103-161)
    parallel_for((size_t)ncol, 8, [&](size_t lo, size_t hi) {
      std::vector<int32_t> lefts(ncol), rights(ncol), lefts_oo(ncol), rights_oo(ncol);
      std::vector<uint8_t> allowed(ncol), have(ncol);
      for (size_t j = lo; j < hi; j++) {
        const int sqlen = (int)strlen(seqs[j]);
        std::fill(have.begin(), have.end(), 0);
        for (size_t t = 0; t < parents[j].size(); t++) {
          const int k = parents[j][t];
          const BimPair &b = res[j][t];
          have[k] = 1;
          allowed[k] = allow_one_off && b.ham >= min_one_off_par_dist;
          if (b.left + b.right < sqlen) { lefts[k] = b.left; rights[k] = b.right; lefts_oo[k] = b.left_oo; rights_oo[k] = b.right_oo; }
          else { lefts[k] = rights[k] = lefts_oo[k] = rights_oo[k] = 0; }   // id / pure-shift / internal-indel "parents"
        }
        int ns = 0, nf = 0;
        for (int i = 0; i < nrow; i++) {
          const int vj = mat[i + j * (size_t)nrow];
          if (vj <= 0) continue;
          ns++;
          int max_left = 0, max_right = 0, oml = 0, omr = 0, omlo = 0, omro = 0;
          for (int k = 0; k < ncol; k++) {
            const int vk = mat[i + (size_t)k * nrow];
            if (!(vk > (min_fold * vj) && vk >= min_abund) || !have[k]) continue;
            max_left = std::max(max_left, lefts[k]); max_right = std::max(max_right, rights[k]);
            if (allow_one_off && allowed[k]) {
              oml = std::max(oml, lefts[k]); omr = std::max(omr, rights[k]);
              omlo = std::max(omlo, lefts_oo[k]); omro = std::max(omro, rights_oo[k]);
            }
          }
          if (max_right + max_left >= sqlen) nf++;
          else if (allow_one_off && (oml + omro >= sqlen || omlo + omr >= sqlen)) nf++;
        }
        nflag[j] = nf; nsam[j] = ns;
      }
    });
    if (times) fprintf(stderr, "[bimera] table %d x %d: parent lists %.1f ms, pairs %.1f ms, flags %.1f ms\n", nrow, ncol, ms_parents, ms_pairs, ms_since(t1));
  });
}

// C_is_bimera (chimera.cpp:18-59): *out = 1 when sq is a two-parent bimera of `pars`
int dada2hip_is_bimera(const char *sq, int32_t npars, const char *const *pars, int32_t allow_one_off, int32_t min_one_off_par_dist,
                       int32_t match, int32_t mismatch, int32_t gap_p, int32_t max_shift, int32_t device, int32_t *out, char *errbuf,
                       size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    if (!sq || !out || (npars > 0 && !pars)) throw InputError{"dada2hip: bad arguments"};
    *out = 0;
    if (npars <= 0) return;
    select_device(device);
    std::vector<const char *> seqs(npars + 1);
    seqs[0] = sq;
    for (int i = 0; i < npars; i++) seqs[i + 1] = pars[i];
    std::vector<std::vector<int32_t>> parents(npars + 1);
    for (int i = 0; i < npars; i++) parents[0].push_back(i + 1);
    std::vector<std::vector<BimPair>> res;
    bimera_pairs(npars + 1, seqs.data(), parents, match, mismatch, gap_p, max_shift, allow_one_off, device, res);
    const int sqlen = (int)strlen(sq);
    int max_left = 0, max_right = 0, oml = 0, omr = 0, omlo = 0, omro = 0;
    // (the reference stops at the first parent that completes a model; the maxima only grow, so the answer is the same)
    for (const BimPair &b : res[0]) {
      if (b.left + b.right >= sqlen) continue;
      max_left = std::max(max_left, b.left); max_right = std::max(max_right, b.right);
      if (allow_one_off && b.ham >= min_one_off_par_dist) {
        oml = std::max(oml, b.left); omr = std::max(omr, b.right);
        omlo = std::max(omlo, b.left_oo); omro = std::max(omro, b.right_oo);
      }
    }
    if (max_right + max_left >= sqlen) *out = 1;
    if (allow_one_off && (oml + omro >= sqlen || omlo + omr >= sqlen)) *out = 1;
  });
}

// get_lr / get_ham_endsfree (chimera.cpp:211-293) of n (query, parent) alignments: what C_is_bimera / C_table_bimera2 look at
int dada2hip_bimera_pairs(int32_t n, const char *const *queries, const char *const *parents, int32_t allow_one_off, int32_t match,
                          int32_t mismatch, int32_t gap_p, int32_t max_shift, int32_t device, int32_t *out, char *errbuf,
                          size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    if (n < 0 || (n > 0 && (!queries || !parents || !out))) throw InputError{"dada2hip: bad arguments"};
    if (n == 0) return;
    select_device(device);
    std::vector<const char *> seqs(2 * (size_t)n);
    std::vector<std::vector<int32_t>> par(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
      if (!queries[i] || !parents[i]) throw InputError{"dada2hip: bad arguments"};
      seqs[2 * i] = queries[i]; seqs[2 * i + 1] = parents[i];
      par[2 * i].push_back(2 * i + 1);
    }
    std::vector<std::vector<BimPair>> res;
    bimera_pairs(2 * n, seqs.data(), par, match, mismatch, gap_p, max_shift, allow_one_off, device, res);
    for (int i = 0; i < n; i++) {
      const BimPair &b = res[2 * i][0];
      int32_t *o = out + 5 * (size_t)i;
      o[0] = b.left; o[1] = b.right; o[2] = b.left_oo; o[3] = b.right_oo; o[4] = b.ham;
    }
  });
}

int dada2hip_nwalign(const char *s1, const char *s2, int32_t match, int32_t mismatch, int32_t gap_p, int32_t homo_gap_p,
                     int32_t band, int32_t endsfree, int32_t device, char *out0, char *out1, char *errbuf, size_t errlen) {
  const char *a[1] = {s1}, *b[1] = {s2};
  char *o[2] = {out0, out1};
  return nwvec_any(1, a, b, match, mismatch, gap_p, homo_gap_p, band, endsfree, device, o, errbuf, errlen);
}

// ---- result getters ------------------------------------------------------------------------------
int32_t dada2hip_result_nclust(const dada2hip_result *r) { return r->nclust; }
int32_t dada2hip_result_nraw(const dada2hip_result *r) { return r->nraw; }
int32_t dada2hip_result_maxlen(const dada2hip_result *r) { return r->maxlen; }
int32_t dada2hip_result_ncol(const dada2hip_result *r) { return r->ncol; }
int32_t dada2hip_result_nbirth_subs(const dada2hip_result *r) { return (int32_t)r->bs_pos.size(); }
const char *dada2hip_result_sequence(const dada2hip_result *r, int32_t i) { return (i >= 0 && i < r->nclust) ? r->sequence[i].c_str() : nullptr; }
const int32_t *dada2hip_result_abundance(const dada2hip_result *r) { return r->abundance.data(); }
const int32_t *dada2hip_result_n0(const dada2hip_result *r) { return r->n0.data(); }
const int32_t *dada2hip_result_n1(const dada2hip_result *r) { return r->n1.data(); }
const int32_t *dada2hip_result_nunq(const dada2hip_result *r) { return r->nunq.data(); }
const double *dada2hip_result_clust_pval(const dada2hip_result *r) { return r->clust_pval.data(); }
const int32_t *dada2hip_result_birth_from(const dada2hip_result *r) { return r->birth_from.data(); }
const double *dada2hip_result_birth_pval(const dada2hip_result *r) { return r->birth_pval.data(); }
const double *dada2hip_result_birth_fold(const dada2hip_result *r) { return r->birth_fold.data(); }
const int32_t *dada2hip_result_birth_ham(const dada2hip_result *r) { return r->birth_ham.data(); }
const double *dada2hip_result_birth_qave(const dada2hip_result *r) { return r->birth_qave.data(); }
const int32_t *dada2hip_result_center(const dada2hip_result *r) { return r->center.data(); }
const int32_t *dada2hip_result_bs_pos(const dada2hip_result *r) { return r->bs_pos.data(); }
const char *dada2hip_result_bs_ref(const dada2hip_result *r) { return r->bs_ref.data(); }
const char *dada2hip_result_bs_sub(const dada2hip_result *r) { return r->bs_sub.data(); }
const double *dada2hip_result_bs_qual(const dada2hip_result *r) { return r->bs_qual.data(); }
const int32_t *dada2hip_result_bs_clust(const dada2hip_result *r) { return r->bs_clust.data(); }
const int32_t *dada2hip_result_subqual(const dada2hip_result *r) { return r->subqual.data(); }
const double *dada2hip_result_clusterquals(const dada2hip_result *r) { return r->clusterquals.data(); }
const int32_t *dada2hip_result_map(const dada2hip_result *r) { return r->map.data(); }
const double *dada2hip_result_pval(const dada2hip_result *r) { return r->pval.data(); }
void dada2hip_result_stats(const dada2hip_result *r, dada2hip_stats *out) { if (out) *out = r->stats; }
void dada2hip_result_free(dada2hip_result *r) { delete r; }

}  // extern "C"
