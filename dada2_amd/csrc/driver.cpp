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
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <unordered_map>

#include "engine.h"
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
struct RuntimeErr { int code; std::string msg; };

template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  void alloc(size_t count) {
    if (count <= n && p) return;
    if (p) { (void)hipFree(p); p = nullptr; }
    n = count;
    D2_HIP(hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)));
  }
  void zero(hipStream_t st) { D2_HIP(hipMemsetAsync(p, 0, n * sizeof(T), st)); }
};

template <typename T> struct PinBuf {
  T *p = nullptr;
  size_t n = 0;
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  void alloc(size_t count) {
    if (count <= n && p) return;
    if (p) { (void)hipHostFree(p); p = nullptr; }
    n = count;
    D2_HIP(hipHostMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault));
  }
};

}  // namespace d2

using namespace d2;

// =================================================================================================
struct dada2hip_sample {
  int device = 0;
  hipStream_t stream = nullptr;
  SampleDev D;
  DevBuf<uint32_t> seq2, heavy, reads;
  DevBuf<uint8_t> qual, nheavy, prior;
  DevBuf<uint16_t> kord;
  DevBuf<int32_t> len;
  std::vector<std::string> seqs;
  std::vector<int32_t> h_len;
  std::vector<uint32_t> h_reads;
  std::vector<uint8_t> h_prior;
  uint64_t total_reads = 0;
  int qmax = 0;
  double ms_upload = 0;
  // per-run device work buffers (kept across runs: selfConsist passes reuse them)
  DevBuf<uint8_t> d_skip, d_cls, d_correct, d_moves;
  DevBuf<double> d_lambda, d_err;
  DevBuf<uint32_t> d_ham, scr_ptr, scr_t, d_qn;
  DevBuf<int32_t> d_nw_list, d_gl_list, d_counters, d_thresh, scr_rows, d_work, d_chunk_centre, d_cluster_of,
      d_centre_of_cluster, d_trans, d_nsubs, d_nmoves;
  DevBuf<uint16_t> d_view, d_view_b;
  DevBuf<unsigned long long> d_qsum;
  PinBuf<double> h_lambda;
  PinBuf<uint32_t> h_ham;
  PinBuf<uint8_t> h_skip, h_cls;
  PinBuf<int32_t> h_counters;
  NwScratch scr;
  int scr_class = -1, scr_band = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
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

void set_err(char *errbuf, size_t errlen, const std::string &m) {
  if (errbuf && errlen) snprintf(errbuf, errlen, "%s", m.c_str());
}

template <typename F> int guarded(char *errbuf, size_t errlen, F &&f) {
  try {
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
void sample_create(dada2hip_sample *s, int32_t nraw, const char *const *seqs, const int32_t *abund,
                   const uint8_t *priors, const double *quals, int32_t quals_nrow, int device) {
  auto t0 = clk::now();
  if (nraw <= 0) throw InputError{"Zero input sequences."};
  if (!seqs || !abund) throw InputError{"Sequence and abundance vectors had different lengths."};
  int maxlen = 0, minlen = SEQLEN;
  s->seqs.resize(nraw);
  s->h_len.resize(nraw);
  s->h_reads.resize(nraw);
  s->h_prior.assign(nraw, 0);
  for (int i = 0; i < nraw; i++) {
    s->seqs[i] = seqs[i];
    int l = (int)s->seqs[i].size();
    s->h_len[i] = l;
    maxlen = std::max(maxlen, l);
    minlen = std::min(minlen, l);
    s->h_reads[i] = (uint32_t)abund[i];
    s->total_reads += (uint32_t)abund[i];
    if (priors) s->h_prior[i] = priors[i] ? 1 : 0;
  }
  if (maxlen >= SEQLEN) throw InputError{"Input sequences exceed the maximum allowed string length."};
  if (minlen <= KMER_SIZE) throw InputError{"Input sequences must all be longer than the kmer-size (5)."};
  if (!quals)
    throw InputError{"dada2hip: a quality matrix is required (the reference reads it unconditionally, src/error.cpp:158)."};
  if (quals_nrow != maxlen) throw InputError{"Sequence must have associated qualities for each nucleotide position."};

  select_device(device);
  s->device = device;
  D2_HIP(hipStreamCreate(&s->stream));
  D2_HIP(hipEventCreate(&s->ev0));
  D2_HIP(hipEventCreate(&s->ev1));
  SampleDev &D = s->D;
  D.N = nraw; D.maxlen = maxlen; D.minlen = minlen;
  D.W2 = (((maxlen + 15) / 16) + 3) & ~3;
  D.LQ = (maxlen + 15) & ~15;
  D.LK = (maxlen - KMER_SIZE + 1 + 7) & ~7;
  D.HMAX = (maxlen - KMER_SIZE + 1) / (RANK_SAT + 1);

  // 2-bit packing on the host (validates ACGT: R checks C_isACGT before the call, R/dada.R:269)
  std::vector<uint32_t> packed((size_t)nraw * D.W2, 0u);
  for (int i = 0; i < nraw; i++) {
    const std::string &q = s->seqs[i];
    uint32_t *row = &packed[(size_t)i * D.W2];
    for (int p = 0; p < (int)q.size(); p++) {
      uint32_t c;
      switch (q[p]) {
        case 'A': c = 0; break;
        case 'C': c = 1; break;
        case 'G': c = 2; break;
        case 'T': c = 3; break;
        default: throw InputError{"Invalid derep$uniques vector. Sequences must be made up only of A/C/G/T."};
      }
      row[p >> 4] |= c << ((p & 15) << 1);
    }
  }
  s->seq2.alloc(packed.size());
  s->len.alloc(nraw); s->reads.alloc(nraw); s->prior.alloc(nraw); s->nheavy.alloc(nraw);
  s->qual.alloc((size_t)nraw * D.LQ);
  s->kord.alloc((size_t)nraw * D.LK);
  s->heavy.alloc((size_t)nraw * std::max(D.HMAX, 1));
  D.seq2 = s->seq2.p; D.len = s->len.p; D.reads = s->reads.p; D.prior = s->prior.p; D.nheavy = s->nheavy.p;
  D.qual = s->qual.p; D.kord = s->kord.p; D.heavy = s->heavy.p;
  D2_HIP(hipMemcpyAsync(D.seq2, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, s->stream));
  D2_HIP(hipMemcpyAsync(D.len, s->h_len.data(), (size_t)nraw * 4, hipMemcpyHostToDevice, s->stream));
  D2_HIP(hipMemcpyAsync(D.reads, s->h_reads.data(), (size_t)nraw * 4, hipMemcpyHostToDevice, s->stream));
  D2_HIP(hipMemcpyAsync(D.prior, s->h_prior.data(), (size_t)nraw, hipMemcpyHostToDevice, s->stream));

  // qualities: stream the double matrix through a staging buffer, round on the device
  DevBuf<int32_t> flags;
  flags.alloc(2);
  flags.zero(s->stream);
  {
    const size_t rows_per = std::max<size_t>(1, (size_t)(256u << 20) / ((size_t)maxlen * 8));
    DevBuf<double> stage;
    stage.alloc(std::min<size_t>(rows_per, nraw) * maxlen);
    for (size_t r0 = 0; r0 < (size_t)nraw; r0 += rows_per) {
      size_t nr = std::min<size_t>(rows_per, nraw - r0);
      D2_HIP(hipMemcpyAsync(stage.p, quals + r0 * maxlen, nr * maxlen * 8, hipMemcpyHostToDevice, s->stream));
      launch_round_quals(stage.p, (int)nr, maxlen, D.len + r0, D.qual + r0 * D.LQ, D.LQ, flags.p, s->stream);
      D2_HIP(hipStreamSynchronize(s->stream));
    }
  }
  int32_t hf[2] = {0, 0};
  D2_HIP(hipMemcpy(hf, flags.p, 8, hipMemcpyDeviceToHost));
  if (hf[0]) throw InputError{"Invalid derep$quals matrix. Quality values must be positive integers."};
  s->qmax = hf[1];
  launch_build_kmers(D, s->stream);
  D2_HIP(hipStreamSynchronize(s->stream));
  D2_HIP(hipGetLastError());
  s->ms_upload = ms_since(t0);
}

void ensure_scratch(dada2hip_sample *s, int band) {
  SampleDev &D = s->D;
  int wc = nw_class(band, D.maxlen, D.minlen);
  if (s->scr_class == wc && s->scr_band == band && s->scr.ptr) return;
  size_t ppw = nw_ptr_words_per_wave(wc, band, D.maxlen, D.minlen);
  int nwaves = std::min(4096, ((D.N + 63) / 64 + 3) & ~3);
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

struct Bi {   // dada.h:85-105
  std::vector<uint32_t> raw;
  uint32_t reads = 0, center = 0xFFFFFFFFu;
  bool update_e = true, check_locks = true;
  double self = 0;
  char birth_type = 'I';
  uint32_t birth_from = 0;
  double birth_pval = 0, birth_fold = 1, birth_e = 0;
  Comp birth_comp{0, 0, 0, 0};
  std::vector<Comp> comp;
};

struct Run {
  dada2hip_sample *s;
  dada2hip_opts o;
  const dada2hip_hooks *hooks;
  int N, ncol;
  std::vector<double> err_rowmajor;
  std::vector<double> p, E_minmax;
  std::vector<Comp> comp;
  std::vector<uint8_t> lock, correct;
  std::vector<Bi> bi;
  dada2hip_stats st;
  std::vector<int32_t> thresh_round, thresh_one;
  AlignParams ap;
  ScreenParams sp;
  int wclass;

  void logf(const char *fmt, ...) {
    if (!o.verbose || !hooks || !hooks->log) return;
    char buf[512];
    va_list a;
    va_start(a, fmt);
    vsnprintf(buf, sizeof buf, fmt, a);
    va_end(a);
    hooks->log(buf, hooks->user);
  }

  // containers.cpp:150-197
  void bi_add_raw(int i, uint32_t r) { bi[i].raw.push_back(r); bi[i].reads += s->h_reads[r]; bi[i].update_e = true; }
  uint32_t bi_pop_raw(int i, uint32_t slot) {
    Bi &b = bi[i];
    uint32_t r = b.raw[slot];
    b.raw[slot] = b.raw.back();   // swap-with-last (containers.cpp:187)
    b.raw.pop_back();
    b.reads -= s->h_reads[r];
    b.update_e = true;
    return r;
  }
  // cluster.cpp:371-386
  void bi_assign_center(int i) {
    Bi &b = bi[i];
    uint32_t mx = 0;
    b.center = 0xFFFFFFFFu;
    for (uint32_t r : b.raw) {
      lock[r] = 0;
      if (s->h_reads[r] > mx) { b.center = r; mx = s->h_reads[r]; }
    }
    b.check_locks = true;
  }

  // one b_compare round on the device: dense lambda/hamming for every unique (cluster.cpp:90-149)
  void device_compare(int centre, double cutoff, const uint8_t *h_skip_or_null, bool count_stats) {
    SampleDev &D = s->D;
    hipStream_t stq = s->stream;
    auto t0 = clk::now();
    const std::vector<int32_t> &th = (cutoff == 1.0) ? thresh_one : thresh_round;
    D2_HIP(hipMemcpyAsync(s->d_thresh.p, th.data(), th.size() * 4, hipMemcpyHostToDevice, stq));
    if (h_skip_or_null) D2_HIP(hipMemcpyAsync(s->d_skip.p, h_skip_or_null, (size_t)N, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemsetAsync(s->d_counters.p, 0, 8 * 4, stq));
    D2_HIP(hipEventRecord(s->ev0, stq));
    launch_screen(D, centre, sp, h_skip_or_null ? s->d_skip.p : nullptr, s->d_thresh.p, s->d_cls.p, s->d_lambda.p,
                  s->d_ham.p, s->d_nw_list.p, s->d_gl_list.p, s->d_counters.p, stq);
    D2_HIP(hipEventRecord(s->ev1, stq));
    D2_HIP(hipMemcpyAsync(s->h_counters.p, s->d_counters.p, 4 * 4, hipMemcpyDeviceToHost, stq));
    launch_gapless(D, centre, nullptr, s->d_gl_list.p, s->d_counters.p + 1, 0, ap, s->d_err.p, s->d_lambda.p,
                   s->d_ham.p, nullptr, 0, 0, stq);
    D2_HIP(hipStreamSynchronize(stq));   // counters on the host: sizes the NW grid
    float ems = 0;
    D2_HIP(hipEventElapsedTime(&ems, s->ev0, s->ev1));
    st.screen_kernel_ms += ems;
    st.screen_kernel_launches++;
    st.ms_screen += ms_since(t0);
    const int n_nw = s->h_counters.p[0], n_gl = s->h_counters.p[1];
    if (count_stats) {
      st.ncompare += (uint64_t)N;
      st.nshroud += (uint64_t)s->h_counters.p[2];
      st.nskipped += (uint64_t)s->h_counters.p[3];
      st.ngapless += (uint64_t)n_gl;
      st.nnw += (uint64_t)n_nw;
    }
    auto t1 = clk::now();
    if (n_nw > 0) {
      D2_HIP(hipEventRecord(s->ev0, stq));
      if (use_coop(n_nw))
        launch_nw_ad(D, centre, nullptr, s->d_nw_list.p, nullptr, n_nw, ap, s->d_err.p, s->d_lambda.p, s->d_ham.p, stq);
      else
        launch_nw(D, wclass, centre, nullptr, s->d_nw_list.p, nullptr, n_nw, ap, s->d_err.p, s->scr, s->d_lambda.p,
                  s->d_ham.p, nullptr, 0, 0, nullptr, 0, nullptr, stq);
      D2_HIP(hipEventRecord(s->ev1, stq));
    }
    D2_HIP(hipMemcpyAsync(s->h_lambda.p, s->d_lambda.p, (size_t)N * 8, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipMemcpyAsync(s->h_ham.p, s->d_ham.p, (size_t)N * 4, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipStreamSynchronize(stq));
    D2_HIP(hipGetLastError());
    if (n_nw > 0) {
      D2_HIP(hipEventElapsedTime(&ems, s->ev0, s->ev1));
      st.nw_kernel_ms += ems;
      st.nw_kernel_launches++;
      st.nw_cells += (uint64_t)n_nw * nw_cells_per_alignment();
    }
    st.ms_nw += ms_since(t1);
  }

  // Kernel choice for a round's NW batch: the cooperative anti-diagonal kernel (k_nw_ad, low latency)
  // for batches that cannot fill the chip one-alignment-per-lane, the lane-per-alignment kernel (k_nw,
  // ~3x fewer instructions per alignment) for very large ones.  DADA2HIP_NW_KERNEL=lane|coop forces one.
  bool use_coop(int n_nw) const {
    const size_t lds = nw_ad_lds_bytes(s->D, ap);
    if (lds == 0 || lds > 150 * 1024) return false;
    const char *f = getenv("DADA2HIP_NW_KERNEL");
    if (f && !strcmp(f, "lane")) return false;
    if (f && !strcmp(f, "coop")) return true;
    return n_nw < 262144;
  }

  uint64_t nw_cells_per_alignment() const {
    // algorithmic DP cells of one alignment: (L1 + L2 + 1) anti-diagonals x (band + 1)
    const int L = s->D.maxlen;
    if (o.band_size < 0) return (uint64_t)(L + 1) * (L + 1);
    return (uint64_t)(2 * L + 1) * (uint64_t)(o.band_size + 1);
  }

  // serial store filter of b_compare_parallel (cluster.cpp:179-201)
  void store_round(int i) {
    auto t0 = clk::now();
    Bi &b = bi[i];
    const uint32_t c = b.center, creads = s->h_reads[c];
    const double *lamv = s->h_lambda.p;
    const uint32_t *hamv = s->h_ham.p;
    const double total = (double)0 + (double)s->total_reads;   // b->reads is unsigned int; lambda * b->reads in fp64
    for (uint32_t index = 0; index < (uint32_t)N; index++) {
      const double lambda = lamv[index];
      if (lambda < 0 || lambda > 1) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "Lambda out-of-range error."};
      if (index == c) b.self = lambda;
      if (lambda * total > E_minmax[index]) {
        if (lambda * creads > E_minmax[index]) E_minmax[index] = lambda * creads;
        Comp cp{(uint32_t)i, index, lambda, hamv[index]};
        b.comp.push_back(cp);
        st.nstored++;
        if (i == 0 || index == c) comp[index] = cp;
      }
    }
    st.ms_bookkeep += ms_since(t0);
  }

  // b_shuffle2 (cluster.cpp:210-266)
  bool shuffle() {
    auto t0 = clk::now();
    bool shuffled = false;
    const int C = (int)bi.size();
    std::vector<double> emax(N);
    std::vector<const Comp *> cmax(N);
    for (int idx = 0; idx < N; idx++) { cmax[idx] = &bi[0].comp[idx]; emax[idx] = cmax[idx]->lambda * bi[0].reads; }
    for (int i = 1; i < C; i++) {
      const double breads = bi[i].reads;
      for (const Comp &cp : bi[i].comp) {
        double e = cp.lambda * breads;
        if (e > emax[cp.index]) { cmax[cp.index] = &cp; emax[cp.index] = e; }
      }
    }
    for (int i = 0; i < C; i++) {
      for (int r = (int)bi[i].raw.size() - 1; r >= 0; r--) {
        uint32_t raw = bi[i].raw[r];
        if (cmax[raw]->i != (uint32_t)i) {
          if (raw == bi[i].center) continue;
          bi_pop_raw(i, (uint32_t)r);
          bi_add_raw((int)cmax[raw]->i, raw);
          comp[raw] = *cmax[raw];
          shuffled = true;
        }
      }
    }
    st.nshuffle++;
    st.ms_bookkeep += ms_since(t0);
    return shuffled;
  }

  // get_pA (pval.cpp:67-89)
  double get_pA(uint32_t raw, int i) {
    const double lambda = comp[raw].lambda;
    const uint32_t hamming = comp[raw].hamming;
    if (s->h_reads[raw] == 1 && !s->h_prior[raw] && !o.detect_singletons) return 1.;
    if (hamming == 0) return 1.;
    if (lambda == 0) return 0.;
    return pp::calc_pA((int)s->h_reads[raw], lambda * bi[i].reads, s->h_prior[raw] || o.detect_singletons);
  }

  // b_p_update (pval.cpp:14-40)
  void p_update() {
    auto t0 = clk::now();
    for (int i = 0; i < (int)bi.size(); i++) {
      Bi &b = bi[i];
      if (b.update_e) {
        for (uint32_t raw : b.raw) p[raw] = get_pA(raw, i);
        b.update_e = false;
      }
      if (o.greedy && b.check_locks) {
        for (uint32_t raw : b.raw) {
          double E_center = s->h_reads[b.center] * comp[raw].lambda;
          if (E_center > s->h_reads[raw]) lock[raw] = 1;
          if (raw == b.center) lock[raw] = 1;
        }
        b.check_locks = false;
      }
    }
    st.ms_pval += ms_since(t0);
  }

  // b_bud (cluster.cpp:274-350)
  int bud() {
    auto t0 = clk::now();
    int mini = -1, minr = -1, mini_p = -1, minr_p = -1;
    uint32_t minraw = bi[0].center, minraw_p = bi[0].center;
    for (int i = 0; i < (int)bi.size(); i++) {
      const Bi &b = bi[i];
      for (int r = 1; r < (int)b.raw.size(); r++) {
        const uint32_t raw = b.raw[r];
        if (s->h_reads[raw] < (uint32_t)o.min_abund) continue;
        if ((int)comp[raw].hamming >= o.min_hamming) {
          if (o.min_fold <= 1 || ((double)s->h_reads[raw]) >= o.min_fold * comp[raw].lambda * b.reads) {
            if (p[raw] < p[minraw] || (p[raw] == p[minraw] && s->h_reads[raw] > s->h_reads[minraw])) { mini = i; minr = r; minraw = raw; }
            if (s->h_prior[raw] && (p[raw] < p[minraw_p] || (p[raw] == p[minraw_p] && s->h_reads[raw] > s->h_reads[minraw_p]))) { mini_p = i; minr_p = r; minraw_p = raw; }
          }
        }
      }
    }
    const double pA = p[minraw] * N, pP = p[minraw_p];
    int newi = 0;
    if (pA < o.omegaA && mini >= 0) {
      const double expected = comp[minraw].lambda * bi[mini].reads;
      uint32_t raw = bi_pop_raw(mini, (uint32_t)minr);
      bi.emplace_back();
      newi = (int)bi.size() - 1;
      Bi &nb = bi[newi];
      nb.birth_type = 'A'; nb.birth_from = (uint32_t)mini; nb.birth_pval = pA; nb.birth_fold = s->h_reads[raw] / expected;
      nb.birth_e = expected; nb.birth_comp = comp[minraw];
      bi_add_raw(newi, raw);
      bi_assign_center(newi);
      logf(", Division (naive): Raw %u from Bi %d, pA=%.2e", raw, mini, pA);
    } else if (pP < o.omegaP && mini_p >= 0) {
      const double expected = comp[minraw_p].lambda * bi[mini_p].reads;
      uint32_t raw = bi_pop_raw(mini_p, (uint32_t)minr_p);
      bi.emplace_back();
      newi = (int)bi.size() - 1;
      Bi &nb = bi[newi];
      nb.birth_type = 'P'; nb.birth_from = 0;   // never assigned in the reference (cluster.cpp:331-345)
      nb.birth_pval = pP; nb.birth_fold = s->h_reads[raw] / expected; nb.birth_e = expected; nb.birth_comp = comp[minraw_p];
      bi_add_raw(newi, raw);
      bi_assign_center(newi);
      logf(", Division (prior): Raw %u from Bi %d, pP=%.2e", raw, mini_p, pP);
    }
    st.ms_bookkeep += ms_since(t0);
    return newi;
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
  s->h_lambda.alloc(N); s->h_ham.alloc(N); s->h_skip.alloc(N); s->h_cls.alloc(N); s->h_counters.alloc(8);
}

void check_opts(const dada2hip_opts &o, int qmax, int ncol) {
  if (o.homo_gap != o.gap && !o.vectorized_alignment && o.homo_gap <= 0 && o.band_size != 0)
    throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED,
                     "dada2hip: HOMOPOLYMER_GAP_PENALTY != GAP_PENALTY (nwalign_endsfree_homo) is outside the implemented path."};
  if (ncol < 1) throw InputError{"Error matrix must have 16 rows."};
  if (o.use_quals && qmax > ncol - 1) throw RuntimeErr{DADA2HIP_ERR_RUNTIME, "Rounded quality exceeded range of err lookup table."};
}

// ---- dada_uniques proper: run_dada (Rmain.cpp:297-336) + outputs (Rmain.cpp:172-294, error.cpp) --
void sample_run(dada2hip_sample *s, const double *err, int err_ncol, const dada2hip_opts *opts,
                const dada2hip_hooks *hooks, dada2hip_result *R) {
  auto t_total = clk::now();
  select_device(s->device);
  if (!err || !opts) throw InputError{"Error matrix must have 16 rows."};
  check_opts(*opts, s->qmax, err_ncol);
  SampleDev &D = s->D;
  const int N = D.N;
  Run run;
  run.s = s; run.o = *opts; run.hooks = hooks; run.N = N; run.ncol = err_ncol;
  memset(&run.st, 0, sizeof run.st);
  run.st.ms_upload = s->ms_upload;
  upload_err(s, err, err_ncol, run.err_rowmajor);
  alloc_round_buffers(s);
  ensure_scratch(s, opts->band_size);
  run.wclass = s->scr_class;
  run.ap = AlignParams{opts->match, opts->mismatch, opts->gap, opts->band_size, nw_sentinel(*opts), opts->use_quals, err_ncol};
  run.sp = ScreenParams{opts->use_kmers, opts->gapless, opts->band_size, opts->SSE};
  run.thresh_round = make_thresh(D.maxlen, opts->kdist_cutoff);
  run.thresh_one = make_thresh(D.maxlen, 1.0);
  run.p.assign(N, 0.0);
  run.E_minmax.assign(N, -999.0);                 // containers.cpp:39
  run.comp.assign(N, Comp{0, 0, 0, 0});
  run.lock.assign(N, 0);
  run.correct.assign(N, 1);
  // b_init (containers.cpp:111-137)
  run.bi.emplace_back();
  run.bi[0].birth_type = 'I'; run.bi[0].birth_fold = 1.0; run.bi[0].birth_e = (double)(uint32_t)s->total_reads;
  run.bi[0].raw.reserve(N);
  for (int i = 0; i < N; i++) run.bi_add_raw(0, (uint32_t)i);
  run.bi_assign_center(0);

  auto make_skip = [&](int i) -> const uint8_t * {
    if (!opts->greedy) return nullptr;
    const uint32_t creads = s->h_reads[run.bi[i].center];
    uint8_t *sk = s->h_skip.p;
    for (int r = 0; r < N; r++) sk[r] = (s->h_reads[r] > creads || run.lock[r]) ? 1 : 0;   // cluster.cpp:127-130
    return sk;
  };

  run.device_compare((int)run.bi[0].center, 1.0, make_skip(0), true);   // Rmain.cpp:309-310: no k-mer screen in round 0
  run.store_round(0);
  run.p_update();
  int max_clust = opts->max_clust < 1 ? N : opts->max_clust;
  int newi;
  while ((int)run.bi.size() < max_clust && (newi = run.bud())) {
    run.logf("\nNew Cluster C%i:", newi);
    run.device_compare((int)run.bi[newi].center, opts->kdist_cutoff, make_skip(newi), true);
    run.store_round(newi);
    int nshuffle = 0;
    bool shuffled;
    do { shuffled = run.shuffle(); } while (shuffled && ++nshuffle < MAX_SHUFFLE);
    run.p_update();
    if (hooks && hooks->should_abort && hooks->should_abort(hooks->user))
      throw RuntimeErr{DADA2HIP_ERR_ABORTED, "dada2hip: aborted by caller"};
  }
  run.st.rounds = (uint32_t)run.bi.size();

  // ---- final alignments (Rmain.cpp:172-236): every member vs its centre, use_kmers = false ----------
  auto t_final = clk::now();
  const int C = (int)run.bi.size();
  const int LV = D.maxlen;
  std::vector<int32_t> work, chunk_centre, cluster_of(N), centre_of_cluster(C);
  work.reserve((size_t)N + 64 * (size_t)C);
  for (int i = 0; i < C; i++) {
    centre_of_cluster[i] = (int32_t)run.bi[i].center;
    const auto &m = run.bi[i].raw;
    for (size_t k = 0; k < m.size(); k++) {
      if (k % 64 == 0) chunk_centre.push_back((int32_t)run.bi[i].center);
      work.push_back((int32_t)m[k]);
      cluster_of[m[k]] = i;
    }
    while (work.size() % 64) work.push_back(-1);
  }
  s->d_work.alloc(work.size()); s->d_chunk_centre.alloc(chunk_centre.size());
  s->d_view.alloc((size_t)N * LV);
  s->d_cluster_of.alloc(N); s->d_centre_of_cluster.alloc(C); s->d_correct.alloc(N);
  s->d_trans.alloc((size_t)16 * err_ncol); s->d_nsubs.alloc(N);
  s->d_qsum.alloc((size_t)C * D.maxlen); s->d_qn.alloc((size_t)C * D.maxlen);
  hipStream_t stq = s->stream;
  D2_HIP(hipMemcpyAsync(s->d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, stq));
  D2_HIP(hipMemcpyAsync(s->d_chunk_centre.p, chunk_centre.data(), chunk_centre.size() * 4, hipMemcpyHostToDevice, stq));
  D2_HIP(hipMemcpyAsync(s->d_cluster_of.p, cluster_of.data(), (size_t)N * 4, hipMemcpyHostToDevice, stq));
  D2_HIP(hipMemcpyAsync(s->d_centre_of_cluster.p, centre_of_cluster.data(), (size_t)C * 4, hipMemcpyHostToDevice, stq));
  if (opts->band_size == 0) {
    launch_gapless(D, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, (int)work.size(), run.ap, s->d_err.p, s->d_lambda.p,
                   s->d_ham.p, s->d_view.p, LV, 0, stq);
    run.st.ngapless += (uint64_t)N;
  } else {
    D2_HIP(hipEventRecord(s->ev0, stq));
    launch_nw(D, run.wclass, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, (int)work.size(), run.ap, s->d_err.p, s->scr,
              s->d_lambda.p, s->d_ham.p, s->d_view.p, LV, 0, nullptr, 0, nullptr, stq);
    D2_HIP(hipEventRecord(s->ev1, stq));
    D2_HIP(hipStreamSynchronize(stq));
    float ems = 0;
    D2_HIP(hipEventElapsedTime(&ems, s->ev0, s->ev1));
    run.st.nw_kernel_ms += ems;
    run.st.nw_kernel_launches++;
    run.st.nnw += (uint64_t)N;
    run.st.nw_cells += (uint64_t)N * run.nw_cells_per_alignment();
  }

  // final per-unique p and the OMEGA_C decision (Rmain.cpp:238-252)
  R->pval.assign(N, 0.0);
  for (int i = 0; i < C; i++)
    for (uint32_t raw : run.bi[i].raw) {
      if (run.bi[i].center == raw) run.p[raw] = 1.0;
      else {
        run.p[raw] = pp::calc_pA((int)s->h_reads[raw], run.comp[raw].lambda * run.bi[i].reads, true);
        if (run.p[raw] < opts->omegaC) run.correct[raw] = 0;
      }
      R->pval[raw] = run.p[raw];
    }
  D2_HIP(hipMemcpyAsync(s->d_correct.p, run.correct.data(), (size_t)N, hipMemcpyHostToDevice, stq));
  D2_HIP(hipMemsetAsync(s->d_trans.p, 0, (size_t)16 * err_ncol * 4, stq));
  D2_HIP(hipMemsetAsync(s->d_qsum.p, 0, (size_t)C * D.maxlen * 8, stq));
  D2_HIP(hipMemsetAsync(s->d_qn.p, 0, (size_t)C * D.maxlen * 4, stq));
  launch_final_tables(D, s->d_view.p, LV, s->d_cluster_of.p, s->d_centre_of_cluster.p, s->d_correct.p, err_ncol, 1,
                      s->d_trans.p, s->d_qsum.p, s->d_qn.p, s->d_nsubs.p, C, stq);
  std::vector<int32_t> nsubs(N);
  std::vector<unsigned long long> qsum((size_t)C * D.maxlen);
  std::vector<uint32_t> qn((size_t)C * D.maxlen);
  R->subqual.assign((size_t)16 * err_ncol, 0);
  D2_HIP(hipMemcpyAsync(nsubs.data(), s->d_nsubs.p, (size_t)N * 4, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemcpyAsync(qsum.data(), s->d_qsum.p, qsum.size() * 8, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemcpyAsync(qn.data(), s->d_qn.p, qn.size() * 4, hipMemcpyDeviceToHost, stq));
  D2_HIP(hipMemcpyAsync(R->subqual.data(), s->d_trans.p, R->subqual.size() * 4, hipMemcpyDeviceToHost, stq));

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
    std::vector<int32_t> w_gl((size_t)nb * 64, -1), w_nw((size_t)nb * 64, -1);
    int n_gl = 0, n_nw = 0;
    for (int k = 0; k < nb; k++) {
      if (pair_cls[k] == CLS_GAPLESS) { w_gl[(size_t)k * 64] = braw[k]; n_gl++; }
      else { w_nw[(size_t)k * 64] = braw[k]; n_nw++; }
    }
    d_wgl.alloc(w_gl.size()); d_wnw.alloc(w_nw.size());
    s->d_view_b.alloc((size_t)nb * LV);
    D2_HIP(hipMemcpyAsync(d_wgl.p, w_gl.data(), w_gl.size() * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemcpyAsync(d_wnw.p, w_nw.data(), w_nw.size() * 4, hipMemcpyHostToDevice, stq));
    D2_HIP(hipMemsetAsync(s->d_view_b.p, 0, (size_t)nb * LV * 2, stq));
    if (n_gl) launch_gapless(D, 0, d_bcc.p, d_wgl.p, nullptr, (int)w_gl.size(), run.ap, s->d_err.p, s->d_lambda.p, s->d_ham.p,
                             s->d_view_b.p, LV, 1, stq);
    if (n_nw) launch_nw(D, run.wclass, 0, d_bcc.p, d_wnw.p, nullptr, (int)w_nw.size(), run.ap, s->d_err.p, s->scr, s->d_lambda.p,
                        s->d_ham.p, s->d_view_b.p, LV, 1, nullptr, 0, nullptr, stq);
    D2_HIP(hipMemcpyAsync(&bview[(size_t)LV], s->d_view_b.p, (size_t)nb * LV * 2, hipMemcpyDeviceToHost, stq));
    D2_HIP(hipStreamSynchronize(stq));
    run.st.ngapless += (uint64_t)n_gl;
    run.st.nnw += (uint64_t)n_nw;
  }
  D2_HIP(hipStreamSynchronize(stq));
  D2_HIP(hipGetLastError());

  // ---- assemble the six outputs -------------------------------------------------------------------
  R->nclust = C; R->nraw = N; R->maxlen = D.maxlen; R->ncol = err_ncol;
  R->sequence.resize(C);
  R->abundance.assign(C, 0); R->n0.assign(C, 0); R->n1.assign(C, 0); R->nunq.assign(C, 0);
  R->birth_from.assign(C, 0); R->birth_ham.assign(C, 0); R->center.assign(C, 0);
  R->clust_pval.assign(C, 0); R->birth_pval.assign(C, 0); R->birth_fold.assign(C, 0); R->birth_qave.assign(C, 0);
  std::unordered_map<uint32_t, int> center_of;
  for (int i = 0; i < C; i++) {   // b_make_clustering_df (error.cpp:9-127)
    const Bi &b = run.bi[i];
    uint32_t max_reads = 0;
    int max_raw = -1;
    for (uint32_t raw : b.raw) if (s->h_reads[raw] > max_reads) { max_raw = (int)raw; max_reads = s->h_reads[raw]; }
    R->sequence[i] = max_raw >= 0 ? s->seqs[max_raw] : std::string("");
    R->center[i] = (int32_t)b.center;
    for (uint32_t raw : b.raw) {
      if (!run.correct[raw]) continue;
      R->abundance[i] += (int32_t)s->h_reads[raw];
      R->nunq[i]++;
      if (nsubs[raw] == 0) R->n0[i] += (int32_t)s->h_reads[raw];
      if (nsubs[raw] == 1) R->n1[i] += (int32_t)s->h_reads[raw];
    }
    if (i == 0) {
      R->birth_pval[i] = na_real(); R->birth_from[i] = DADA2HIP_NA_INTEGER; R->birth_fold[i] = na_real();
      R->birth_ham[i] = DADA2HIP_NA_INTEGER; R->birth_qave[i] = na_real();
    } else {
      R->birth_from[i] = (int32_t)b.birth_from + 1;
      R->birth_pval[i] = b.birth_pval; R->birth_fold[i] = b.birth_fold; R->birth_ham[i] = (int32_t)b.birth_comp.hamming;
    }
    center_of[b.center] = i;
  }
  {   // post-hoc p-value (error.cpp:101-119)
    std::vector<double> tot_e(C, 0.0);
    for (int i = 0; i < C; i++)
      for (const Comp &cp : run.bi[i].comp) {
        auto it = center_of.find(cp.index);
        if (it != center_of.end() && it->second != i) tot_e[it->second] += cp.lambda * run.bi[i].reads;
      }
    for (int i = 0; i < C; i++) R->clust_pval[i] = pp::calc_pA((int)s->h_reads[run.bi[i].center], tot_e[i], true);
  }
  // birth_subs data.frame (error.cpp:261-300) + birth_qave (error.cpp:83-92)
  for (int i = 1; i < C; i++) {
    const uint32_t pc = run.bi[run.bi[i].birth_comp.i].center;
    const std::string &cs = s->seqs[pc];
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
  R->map.assign(N, DADA2HIP_NA_INTEGER);   // Rmain.cpp:268-279
  for (int i = 0; i < C; i++)
    for (uint32_t raw : run.bi[i].raw) R->map[raw] = run.correct[raw] ? i + 1 : DADA2HIP_NA_INTEGER;
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

void dada2hip_sample_free(dada2hip_sample *s) {
  if (!s) return;
  if (s->stream) { (void)hipSetDevice(s->device); (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
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

int dada2hip_sample_compare(dada2hip_sample *s, int32_t centre, const double *err, int32_t err_ncol,
                            const dada2hip_opts *opts, double kdist_cutoff, const uint8_t *skip, double *lambda,
                            uint32_t *hamming, uint8_t *cls, dada2hip_stats *stats, char *errbuf, size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    select_device(s->device);
    if (centre < 0 || centre >= s->D.N) throw InputError{"dada2hip: centre out of range"};
    check_opts(*opts, s->qmax, err_ncol);
    Run run;
    run.s = s; run.o = *opts; run.hooks = nullptr; run.N = s->D.N; run.ncol = err_ncol;
    memset(&run.st, 0, sizeof run.st);
    upload_err(s, err, err_ncol, run.err_rowmajor);
    alloc_round_buffers(s);
    ensure_scratch(s, opts->band_size);
    run.wclass = s->scr_class;
    run.ap = AlignParams{opts->match, opts->mismatch, opts->gap, opts->band_size, nw_sentinel(*opts), opts->use_quals, err_ncol};
    run.sp = ScreenParams{opts->use_kmers, opts->gapless, opts->band_size, opts->SSE};
    run.thresh_round = make_thresh(s->D.maxlen, kdist_cutoff);
    run.thresh_one = make_thresh(s->D.maxlen, 1.0);
    if (skip) memcpy(s->h_skip.p, skip, (size_t)s->D.N);
    run.device_compare(centre, kdist_cutoff == 1.0 ? 1.0 : kdist_cutoff, skip ? s->h_skip.p : nullptr, true);
    if (lambda) memcpy(lambda, s->h_lambda.p, (size_t)s->D.N * 8);
    if (hamming) memcpy(hamming, s->h_ham.p, (size_t)s->D.N * 4);
    if (cls) D2_HIP(hipMemcpy(cls, s->d_cls.p, (size_t)s->D.N, hipMemcpyDeviceToHost));
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
int dada2hip_nwvec(int32_t n, const char *const *s1, const char *const *s2, int32_t match, int32_t mismatch,
                   int32_t gap_p, int32_t band, int32_t endsfree, int32_t device, char *const *out, char *errbuf,
                   size_t errlen) {
  return guarded(errbuf, errlen, [&] {
    if (!endsfree)
      throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED, "dada2hip: endsfree=FALSE (global nwalign) is outside the denoising path."};
    if (n <= 0) return;
    // a throw-away resident sample holding the 2n strings; pair i = (centre 2i, raw 2i+1), one pair per wave
    std::vector<const char *> seqs(2 * (size_t)n);
    std::vector<int32_t> ab(2 * (size_t)n, 1);
    int maxlen = 0;
    for (int i = 0; i < n; i++) {
      seqs[2 * i] = s1[i]; seqs[2 * i + 1] = s2[i];
      maxlen = std::max<int>(maxlen, (int)std::max(strlen(s1[i]), strlen(s2[i])));
    }
    std::vector<double> q((size_t)2 * n * maxlen, 0.0);
    dada2hip_sample *s = new dada2hip_sample();
    std::unique_ptr<dada2hip_sample, void (*)(dada2hip_sample *)> guard(s, dada2hip_sample_free);
    sample_create(s, 2 * n, seqs.data(), ab.data(), nullptr, q.data(), maxlen, device);
    ensure_scratch(s, band);
    std::vector<double> errm(16, 1.0), rowm;
    upload_err(s, errm.data(), 1, rowm);
    dada2hip_opts o;
    memset(&o, 0, sizeof o);
    o.match = match; o.mismatch = mismatch; o.gap = gap_p; o.vectorized_alignment = 1;
    AlignParams ap{match, mismatch, gap_p, band, nw_sentinel(o), 0, 1};
    std::vector<int32_t> work((size_t)n * 64, -1), cc(n);
    for (int i = 0; i < n; i++) { work[(size_t)i * 64] = 2 * i + 1; cc[i] = 2 * i; }
    const int stride = 2 * maxlen + 2;
    s->d_work.alloc(work.size()); s->d_chunk_centre.alloc(n); s->d_moves.alloc(work.size() * (size_t)stride);
    s->d_nmoves.alloc(work.size()); s->d_lambda.alloc(2 * (size_t)n); s->d_ham.alloc(2 * (size_t)n);
    D2_HIP(hipMemcpyAsync(s->d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, s->stream));
    D2_HIP(hipMemcpyAsync(s->d_chunk_centre.p, cc.data(), (size_t)n * 4, hipMemcpyHostToDevice, s->stream));
    if (band == 0) throw RuntimeErr{DADA2HIP_ERR_UNSUPPORTED, "dada2hip: band == 0 is the gapless pairing, not an NW call."};
    launch_nw(s->D, s->scr_class, 0, s->d_chunk_centre.p, s->d_work.p, nullptr, (int)work.size(), ap, s->d_err.p, s->scr,
              s->d_lambda.p, s->d_ham.p, nullptr, 0, 0, s->d_moves.p, stride, s->d_nmoves.p, s->stream);
    std::vector<uint8_t> moves(work.size() * (size_t)stride);
    std::vector<int32_t> nm(work.size());
    D2_HIP(hipMemcpyAsync(moves.data(), s->d_moves.p, moves.size(), hipMemcpyDeviceToHost, s->stream));
    D2_HIP(hipMemcpyAsync(nm.data(), s->d_nmoves.p, nm.size() * 4, hipMemcpyDeviceToHost, s->stream));
    D2_HIP(hipStreamSynchronize(s->stream));
    D2_HIP(hipGetLastError());
    for (int i = 0; i < n; i++) {
      const uint8_t *mv = &moves[(size_t)i * 64 * stride];
      const int len = nm[(size_t)i * 64];
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

int dada2hip_nwalign(const char *s1, const char *s2, int32_t match, int32_t mismatch, int32_t gap_p, int32_t homo_gap_p,
                     int32_t band, int32_t endsfree, int32_t device, char *out0, char *out1, char *errbuf, size_t errlen) {
  if (gap_p != homo_gap_p) {
    set_err(errbuf, errlen, "dada2hip: homo_gap_p != gap_p (nwalign_endsfree_homo) is outside the implemented path.");
    return DADA2HIP_ERR_UNSUPPORTED;
  }
  const char *a[1] = {s1}, *b[1] = {s2};
  char *o[2] = {out0, out1};
  return dada2hip_nwvec(1, a, b, match, mismatch, gap_p, band, endsfree, device, o, errbuf, errlen);
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
