/*
 * dada2hip.h — C ABI of libdada2hip.so, the MI355X (gfx950) implementation of DADA2's
 * divisive-denoising hot path.  This is the drop-in boundary: these entry points are what
 * the reference's Rcpp glue for the path would bind (INTEGRATION.md shows the 40-line
 * Rcpp translation unit that re-exports `_dada2_dada_uniques`, `_dada2_C_nwalign` and
 * `_dada2_C_nwvec` on top of them, so R/dada.R and R/misc.R call the new path unchanged).
 *
 * Reference interfaces replaced (all paths relative to /root/reference):
 *   dada2hip_dada_uniques   <- dada_uniques()      src/Rmain.cpp:30-295  (.Call `_dada2_dada_uniques`,
 *                                                   src/RcppExports.cpp:18-53, R/RcppExports.R:8-10)
 *   dada2hip_sample_* / _run<- the same call split so reads/qualities/k-mer tables stay resident in
 *                              HBM across the selfConsist passes of R/dada.R:256-405 (only `err` changes)
 *   dada2hip_nwalign        <- C_nwalign()         src/evaluate.cpp:18-62    (`_dada2_C_nwalign`, RcppExports.cpp:94)
 *   dada2hip_nwvec          <- C_nwvec()           src/nwalign_vectorized.cpp:321-343 (`_dada2_C_nwvec`, :227)
 *   dada2hip_result_*       <- the Rcpp::List of six objects built at src/Rmain.cpp:254-294 and src/error.cpp
 *   dada2hip_table_bimera2  <- C_table_bimera2()   src/chimera.cpp:192-208   (`_dada2_C_table_bimera2`; R/chimeras.R:236)
 *   dada2hip_is_bimera      <- C_is_bimera()       src/chimera.cpp:18-59     (`_dada2_C_is_bimera`; R/chimeras.R:44)
 *   dada2hip_merge_pairs    <- mergePairs()        R/paired.R:92-201 on C_nwalign / C_eval_pair / C_pair_consensus
 *                                                   (src/evaluate.cpp:18-62, :73-114, :124-174)
 *   dada2hip_derep_*        <- derepFastq() / qtables2()  R/sequenceIO.R:45-124, :150-183 (host-side C++, zlib)
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every call returns 0 on success or a
 * non-zero code with a NUL-terminated message in `errbuf` (the reference's Rcpp::stop texts are
 * kept where one exists).  Inputs are borrowed for the duration of the call only.  A result is
 * owned by the library until dada2hip_result_free().  There is NO CPU fallback: every entry point
 * that computes fails with DADA2HIP_ERR_DEVICE when no gfx950 device is usable.
 *
 * Environment.  The library reads its DADA2HIP_* variables in ONE place (dada2_amd/csrc/knobs.h), once per boundary call,
 * into an immutable snapshot; none of them changes a result.  Supported (the parity tests sweep them):
 *   DADA2HIP_ENGINE=classic            one centre per round, a host round trip per decision (round 1's engine)
 *   DADA2HIP_V2_TAIL=chain             the round tail as launch chains instead of the persistent kernel
 *   DADA2HIP_V3_OVERLAP=0|1            the next batch's compare under the persistent tail on a second stream (default: on)
 *   DADA2HIP_V3_SPEC=0|1               the evaluation of a round rides on its shuffle calls (default: on; 0 = a phase of its own)
 *   DADA2HIP_V3_MIRROR=0|1             the persistent tail keeps the per-unique facts its sweeps ask for in LDS (default: on)
 *   DADA2HIP_V3_SLOTS=<n>              persistent launches of this process side by side on a device (default 3; 1 = the rounds of
 *                                      several samples in flight take turns)
 * The library SETS one variable when it is loaded, unless the caller has: GPU_MAX_HW_QUEUES=8 (the HIP runtime's hardware queues
 * per process, 4 by default; a run uses three streams, several samples in flight three each).
 *   DADA2HIP_NW_KERNEL=lane|coop|wide  force one aligner family;  DADA2HIP_AD_HOMO=0  homopolymer gaps on the lane kernels
 *   DADA2HIP_WAIT=block, DADA2HIP_WAIT_TIMEOUT_S=<s>   sleep instead of spin while waiting; bound of every device wait
 *   DADA2HIP_HOST_THREADS=<n>, DADA2HIP_ALLOC_CACHE=0, DADA2HIP_ALLOC_CACHE_GB=<n>   marshalling pool, allocation cache
 *   DADA2HIP_DEREP_INFLATE=zlib        dada2hip_derep_fastq: .gz files through zlib's streaming inflate even where libdeflate is installed
 *   DADA2HIP_PROFILE=1, DADA2HIP_V2_SUMMARY, DADA2HIP_V2_DEBUG   per-launch device times in the stats; traces on stderr
 * Test / tuning knobs (sizes of rings and grids, forced growth paths, injected failures) are listed with their meaning in
 * knobs.h and DESIGN.md §10b; they are not part of the interface.
 */
#ifndef DADA2HIP_H
#define DADA2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DADA2HIP_OK            0
#define DADA2HIP_ERR_INPUT     1   /* validation failure (messages of src/Rmain.cpp:52-78) */
#define DADA2HIP_ERR_DEVICE    2   /* no usable GPU / HIP runtime error */
#define DADA2HIP_ERR_RUNTIME   3   /* runtime error of the algorithm ("Lambda out-of-range error." ...) */
#define DADA2HIP_ERR_UNSUPPORTED 4 /* input outside what the device path represents (N / IUPAC codes in nwalign, band 0 in nwvec) */
#define DADA2HIP_ERR_ABORTED   5   /* should_abort() returned non-zero (Rcpp::checkUserInterrupt analogue) */

#define DADA2HIP_NA_INTEGER (-2147483647 - 1)   /* R's NA_integer_ */

/* The 23 scalars dada_uniques takes positionally (src/Rmain.cpp:33-47); doubles first so the
 * struct has no padding (112 bytes).  Defaults and normalisation: R/dada.R:1-27, :222-237. */
typedef struct dada2hip_opts {
  double kdist_cutoff, omegaA, omegaP, omegaC, min_fold;
  int32_t match, mismatch, gap, homo_gap, band_size, max_clust, min_hamming, min_abund;
  int32_t use_kmers, detect_singletons, use_quals, final_consensus, vectorized_alignment;
  int32_t multithread, verbose, SSE, gapless, greedy;
} dada2hip_opts;

/* Optional host callbacks (replace Rprintf and Rcpp::checkUserInterrupt, src/Rmain.cpp:317-333). */
typedef struct dada2hip_hooks {
  void (*log)(const char *msg, void *user);
  int (*should_abort)(void *user);   /* polled once per divisive round */
  void *user;
} dada2hip_hooks;

typedef struct dada2hip_sample dada2hip_sample;   /* uniques of one sample, resident in HBM */
typedef struct dada2hip_result dada2hip_result;

/* Counters of one run (the reference's nalign/nshroud, src/dada.h:113-114, plus device timings). */
typedef struct dada2hip_stats {
  uint64_t ncompare;     /* raws offered to b_compare over all rounds (nraw x rounds)            */
  uint64_t nskipped;     /* greedy skips: reads > centre reads, or locked (cluster.cpp:127-130)  */
  uint64_t nshroud;      /* k-mer screened out (nwalign_endsfree.cpp:51-53)                      */
  uint64_t ngapless;     /* gapless pairings (:54-55)                                            */
  uint64_t nnw;          /* banded NW alignments incl. the final-subs pass (:57-64)              */
  uint64_t nshuffle;     /* b_shuffle2 calls                                                     */
  uint64_t nstored;      /* Comparisons kept (cluster.cpp:189-199)                               */
  uint32_t rounds;       /* b_compare rounds = final number of partitions                        */
  uint32_t kernel_times_sampled;  /* 1: nw/screen_kernel_ms are EXTRAPOLATED from sampled launches (round 0, every 8th
                                     round, the final pass); 0: every launch was event-timed (DADA2HIP_PROFILE=1)  */
  /* host wall-clock split of the call: upload = sample creation (marshalling + H2D + k-mer build), screen = enqueue
   * of the compare kernels, bookkeep = round tails incl. waiting for the device, final = final pass + outputs */
  double ms_total, ms_upload, ms_screen, ms_nw, ms_gapless, ms_bookkeep, ms_pval, ms_final;
  double nw_kernel_ms;   /* HIP-event time of the NW kernel launches (see kernel_times_sampled)  */
  uint64_t nw_kernel_launches;
  uint64_t nw_cells;     /* DP cells those launches filled (algorithmic work)      */
  double screen_kernel_ms;
  uint64_t screen_kernel_launches;
  uint64_t screen_bytes; /* algorithmic bytes the screen launches had to read                     */
  /* device time per kernel class, summed over every launch; filled only under DADA2HIP_PROFILE=1 (else 0) */
  double dev_ms_screen, dev_ms_nw, dev_ms_shuffle, dev_ms_pval, dev_ms_birth, dev_ms_final;
  /* host side of the device-driven round loop: waiting for published round results (= the device is the bottleneck),
   * replaying moves / births on the host mirror, enqueuing launches; uniques moved by b_shuffle2 over the run */
  double ms_wait_device, ms_replay, ms_enqueue;
  uint64_t nmoves, batch_compares;
  /* pairs the aligner processed for the rounds' batch compares (device-driven rounds align a whole batch - up to eight
   * coming centres - in one launch, as of the screen: a pair the greedy rule skips when its round commits, or whose batch
   * position is never used, is aligned in vain; nnw / ngapless above stay the reference's counts) */
  uint64_t nnw_run, ngapless_run;
  /* rounds enqueued without the launches of a batch compare because their centre was expected to be cached, and how many
   * of those guesses were wrong (the device then halts and the host sends the full chain) */
  uint64_t lite_chains, lite_misses;
  /* persistent round tail (one launch runs rounds back to back until the next batch compare is due): launches, blocks of the
   * launch, pauses (a round whose mover lists did not fit its result block), shuffle calls; under DADA2HIP_PROFILE=1 also the
   * device time of those launches and what block 0 of them spent in each phase (barriers = waiting for the other blocks,
   * which includes the serial end of the round run by the last arriver) */
  uint64_t tail_launches, tail_pauses, tail_levels;
  uint32_t tail_blocks;
  uint32_t tail_fallbacks;   /* persistent launches whose entry barrier failed (blocks not co-resident): the run went on on the launch chains */
  double dev_ms_tail, tail_ms_entry, tail_ms_shuffle0, tail_ms_shuffle_more, tail_ms_pupdate, tail_ms_barriers, tail_ms_birth,
      tail_ms_publish, tail_ms_release;
  /* the NEXT batch's compare under the persistent tail (second stream): prefetch compares launched; rounds whose centre came out
   * of a prefetched batch; how often the tail had to spin for a prefetch still in flight / left its launch for one; centres
   * prefetched; threads per block of the tail; 1 if the overlap was on.  Under DADA2HIP_PROFILE=1: device time of the
   * prefetch compares' screen and aligner launches (they run INSIDE the interval of dev_ms_tail) */
  uint64_t pf_compares, pf_hits, pf_waits, pf_exits, pf_centres;
  uint32_t tail_threads, overlap_on;
  double dev_ms_pf_screen, dev_ms_pf_nw;
  uint32_t tail_xcd_barrier;   /* 1: the persistent launches of the run used the XCD-hierarchical grid barrier */
  uint32_t tail_mirror;        /* 1: ... and kept the per-unique facts their sweeps ask for in LDS for the life of a launch (DADA2HIP_V3_MIRROR) */
  /* host wall of what stands in front of the rounds: setup = state buffers, memsets, the host mirror of b_init; round0 = the
   * comparison of every unique with the first centre (Rmain.cpp:309-310) up to the first enqueue of the rounds */
  double ms_setup, ms_round0;
  /* under DADA2HIP_PROFILE=1: what the serial end of the rounds spent spinning for a prefetch compare still in flight, and
   * planning the next prefetch (both are part of tail_ms_birth) */
  double tail_ms_pf_wait, tail_ms_pf_plan;
  /* batch compares on the default scores run a pointer-free first pass of the aligner (k_nw_ad<.., FAST>): the pairs it could
   * not finish - walks with an interior gap - and handed to the full kernel, and the pairs the pass looked at (of nnw_run: the pass
   * switches itself off for the rest of a run once more than a quarter came back) */
  uint64_t nnw_retry, nnw_fast;
  /* batch screens with the 5-mer presence bitmaps: uniques (summed over the batch screens, each up to 8 centres) whose class
   * word the 128-byte bound did not settle, i.e. that went through the exact k-mer walk */
  uint64_t screen_stage2;
  /* NW pairs the rounds committed (nnw without round 0, the final pass and the birth pairs): nnw_run - nnw_rounds were aligned in
   * vain - their batch position was never used, or a greedy skip dropped the pair when its round committed */
  uint64_t nnw_rounds;
} dada2hip_stats;

/* ---- whole-call form: exactly dada_uniques (src/Rmain.cpp:30) ---------------------------------
 * seqs       nraw NUL-terminated ACGT strings (abundance-sorted, R/sequenceIO.R:98)
 * abundances nraw ints; priors nraw bytes (0/1) or NULL
 * err        16 x err_ncol doubles, COLUMN-major (R NumericMatrix), rows A2A,A2C,..,T2T
 * quals      quals_nrow x nraw doubles, column-major: positions are rows, uniques are columns
 *            (src/Rmain.cpp:69,113); quals_nrow must equal the longest sequence.  Required: the
 *            reference dereferences it unconditionally (src/error.cpp:158) and R always passes it.
 * device     HIP device ordinal (the reference has no such notion; the R glue passes 0). */
int dada2hip_dada_uniques(int32_t nraw, const char *const *seqs, const int32_t *abundances,
                          const uint8_t *priors, const double *err, int32_t err_ncol, const double *quals,
                          int32_t quals_nrow, const dada2hip_opts *opts, int32_t device,
                          const dada2hip_hooks *hooks, dada2hip_result **out, char *errbuf, size_t errlen);

/* ---- batch form: the per-sample loop of dada() (R/dada.R:266-366 calls dada_uniques once per sample, serially)
 * spread over the GPUs of one node.  Samples only share `err`, so sample i simply runs on
 * device_ids[i % n_devices], one host thread per entry of device_ids, the samples of a thread in order; out[i]
 * receives sample i's result (all NULL on failure; errbuf then names the first failing sample).  The caller sums
 * the results' $subqual matrices (accumulateTrans, R/errorModels.R:462-471).  n_devices <= 0: device 0 only. */
typedef struct dada2hip_sample_input {
  int32_t nraw;
  int32_t quals_nrow;
  const char *const *seqs;
  const int32_t *abundances;
  const uint8_t *priors;      /* may be NULL */
  const double *quals;
} dada2hip_sample_input;
int dada2hip_run_multi(int32_t n_samples, const dada2hip_sample_input *samples, const double *err, int32_t err_ncol,
                       const dada2hip_opts *opts, int32_t n_devices, const int32_t *device_ids, dada2hip_result **out,
                       char *errbuf, size_t errlen);

/* Device and pinned-host allocations are cached per process between calls (the "per-device context cache" of a
 * drop-in replacement: back-to-back dada_uniques calls do not pay hipMalloc / hipFree again); this hands every cached
 * block back to the runtime.  DADA2HIP_ALLOC_CACHE=0 disables the cache. */
void dada2hip_trim_cache(void);

/* ---- resident form --------------------------------------------------------------------------- */
int dada2hip_sample_create(int32_t nraw, const char *const *seqs, const int32_t *abundances,
                           const uint8_t *priors, const double *quals, int32_t quals_nrow, int32_t device,
                           dada2hip_sample **out, char *errbuf, size_t errlen);
int dada2hip_sample_set_priors(dada2hip_sample *s, const uint8_t *priors, char *errbuf, size_t errlen);
int dada2hip_sample_run(dada2hip_sample *s, const double *err, int32_t err_ncol, const dada2hip_opts *opts,
                        const dada2hip_hooks *hooks, dada2hip_result **out, char *errbuf, size_t errlen);
void dada2hip_sample_free(dada2hip_sample *s);

/* ---- one sample over several GPUs (last row: "intra-sample sharding") --------------
 * The reference has no counterpart: a sample is one serial dada_uniques call (R/dada.R:266).  Here every rank (one
 * process per GPU) holds the whole sample's sequences, qualities and k-mer records, and does the per-unique work of
 * run_dada - b_compare (src/cluster.cpp:90-204), b_shuffle2's arg-max (:229-239), b_p_update (src/pval.cpp:14-40), the
 * first stage of b_bud (src/cluster.cpp:284-308), the final alignments (src/Rmain.cpp:172-236) - for ONE contiguous
 * block of the uniques, [nraw * rank / world, nraw * (rank + 1) / world).  What the reference's serial bookkeeping
 * shares between uniques is exchanged through `exchange` at fixed points, the same sequence on every rank:
 *   per b_shuffle2 call   the (unique, from, to) triples that moved  (all ranks replay all of them in the reference's order:
 *                         partition reads, member slots and the arg-max snapshot of the next call stay identical everywhere)
 *   per b_bud             the candidate records of each rank's best key and its near ties (48 B each)
 *   at the end            p-value / correction / substitution count of every unique, the transition and quality sums
 * and every rank returns the complete, identical result.  The round loop runs host-driven (one centre per round, every
 * decision on the host): the exchanges sit between its kernels.
 *   kind 0  all-gather: every rank contributes send_bytes bytes (the same count on all ranks); recv receives
 *           world * send_bytes bytes in rank order
 *   kind 1  all-reduce(sum) of send_bytes / 8 int64 values; send == recv (in place)
 * `exchange` returns 0 on success; anything else aborts the run with DADA2HIP_ERR_RUNTIME on that rank.
 * Failure protocol: every exchange point opens with an 8-byte kind-0 gather of a size; a rank that has failed between two points
 * contributes -1 at the next one (one extra call of `exchange`, not made once the run's last exchange point is behind it), and
 * its peers return DADA2HIP_ERR_RUNTIME there.  A collective that never completes - a rank that died, a transport that is gone -
 * is `exchange`'s own to time out: the library never waits for a peer by itself. */
typedef struct dada2hip_shard {
  int32_t rank, world;
  int (*exchange)(void *user, int32_t kind, const void *send, int64_t send_bytes, void *recv);
  void *user;
} dada2hip_shard;
int dada2hip_sample_run_sharded(dada2hip_sample *s, const double *err, int32_t err_ncol, const dada2hip_opts *opts,
                                const dada2hip_hooks *hooks, const dada2hip_shard *shard, dada2hip_result **out,
                                char *errbuf, size_t errlen);
int32_t dada2hip_sample_nraw(const dada2hip_sample *s);
int32_t dada2hip_sample_maxlen(const dada2hip_sample *s);

/* ---- result access (library-owned arrays, valid until dada2hip_result_free) -------------------
 * $clustering (src/error.cpp:121-126): one row per partition. */
int32_t dada2hip_result_nclust(const dada2hip_result *r);
int32_t dada2hip_result_nraw(const dada2hip_result *r);
int32_t dada2hip_result_maxlen(const dada2hip_result *r);
int32_t dada2hip_result_ncol(const dada2hip_result *r);          /* columns of $subqual */
int32_t dada2hip_result_nbirth_subs(const dada2hip_result *r);
const char *dada2hip_result_sequence(const dada2hip_result *r, int32_t i);
const int32_t *dada2hip_result_abundance(const dada2hip_result *r);
const int32_t *dada2hip_result_n0(const dada2hip_result *r);
const int32_t *dada2hip_result_n1(const dada2hip_result *r);
const int32_t *dada2hip_result_nunq(const dada2hip_result *r);
const double *dada2hip_result_clust_pval(const dada2hip_result *r);
const int32_t *dada2hip_result_birth_from(const dada2hip_result *r);   /* 1-based, NA for row 0 */
const double *dada2hip_result_birth_pval(const dada2hip_result *r);
const double *dada2hip_result_birth_fold(const dada2hip_result *r);
const int32_t *dada2hip_result_birth_ham(const dada2hip_result *r);
const double *dada2hip_result_birth_qave(const dada2hip_result *r);
const int32_t *dada2hip_result_center(const dada2hip_result *r);       /* 0-based unique index of each centre (extra) */
/* $birth_subs (src/error.cpp:299) */
const int32_t *dada2hip_result_bs_pos(const dada2hip_result *r);       /* 1-based */
const char *dada2hip_result_bs_ref(const dada2hip_result *r);          /* one char per row */
const char *dada2hip_result_bs_sub(const dada2hip_result *r);
const double *dada2hip_result_bs_qual(const dada2hip_result *r);
const int32_t *dada2hip_result_bs_clust(const dada2hip_result *r);     /* 1-based */
/* $subqual 16 x ncol int32 column-major; $clusterquals maxlen x nclust double column-major (NA past the
 * centre's length); $map nraw int32 (1-based or NA); $pval nraw double. */
const int32_t *dada2hip_result_subqual(const dada2hip_result *r);
const double *dada2hip_result_clusterquals(const dada2hip_result *r);
const int32_t *dada2hip_result_map(const dada2hip_result *r);
const double *dada2hip_result_pval(const dada2hip_result *r);
void dada2hip_result_stats(const dada2hip_result *r, dada2hip_stats *out);
void dada2hip_result_free(dada2hip_result *r);

/* ---- pairwise alignment exports ---------------------------------------------------------------
 * dada2hip_nwalign == C_nwalign(s1, s2, match, mismatch, gap_p, homo_gap_p, band, endsfree)
 * (src/evaluate.cpp:18): ACGT in, two gapped strings out (capacity >= len1+len2+1 each).
 * dada2hip_nwvec == C_nwvec (src/nwalign_vectorized.cpp:321): n pairs at once; out[2*i], out[2*i+1]
 * are caller buffers of capacity >= len1_i+len2_i+1.  Both run the device NW kernel, and all of C_nwalign's aligners
 * are there: nwalign_endsfree (src/nwalign_endsfree.cpp:76-216), nwalign_endsfree_homo when homo_gap_p != gap_p
 * (:220-396: a gap opposite a base of a homopolymer run of >= 3 costs homo_gap_p) and, with endsfree = 0, the global
 * nwalign (:403-537; C_nwvec: nwalign_vectorized2 with end_gap_p = gap_p, src/nwalign_vectorized.cpp:333). */
int dada2hip_nwalign(const char *s1, const char *s2, int32_t match, int32_t mismatch, int32_t gap_p,
                     int32_t homo_gap_p, int32_t band, int32_t endsfree, int32_t device, char *out0, char *out1,
                     char *errbuf, size_t errlen);
int dada2hip_nwvec(int32_t n, const char *const *s1, const char *const *s2, int32_t match, int32_t mismatch,
                   int32_t gap_p, int32_t band, int32_t endsfree, int32_t device, char *const *out,
                   char *errbuf, size_t errlen);

/* ---- bimera identification: the step after dada() ------------------------------------------
 * dada2hip_table_bimera2 == C_table_bimera2(mat, seqs, min_fold, min_abund, allow_one_off, min_one_off_par_dist, match,
 * mismatch, gap_p, max_shift) (src/chimera.cpp:192): mat is the nrow (samples) x ncol (sequences) integer table, column-major
 * as R holds it; nflag[ncol] / nsam[ncol] receive the two columns of the returned data.frame.  Every (sequence, more abundant
 * parent) alignment the table asks for runs on the device NW kernel (band = max_shift, ends-free: the denoising path's
 * aligner, chimera.cpp:122), get_lr / get_ham_endsfree (:211-293) on the device too.
 * dada2hip_is_bimera == C_is_bimera (src/chimera.cpp:18): *out = 1 / 0. */
int dada2hip_table_bimera2(int32_t nrow, int32_t ncol, const int32_t *mat, const char *const *seqs, double min_fold,
                           int32_t min_abund, int32_t allow_one_off, int32_t min_one_off_par_dist, int32_t match,
                           int32_t mismatch, int32_t gap_p, int32_t max_shift, int32_t device, int32_t *nflag, int32_t *nsam,
                           char *errbuf, size_t errlen);
int dada2hip_is_bimera(const char *sq, int32_t npars, const char *const *pars, int32_t allow_one_off,
                       int32_t min_one_off_par_dist, int32_t match, int32_t mismatch, int32_t gap_p, int32_t max_shift,
                       int32_t device, int32_t *out, char *errbuf, size_t errlen);
/* dada2hip_bimera_pairs: the quantities the two entry points above reduce to a decision, for n (query, parent) pairs:
 * out[5 i .. 5 i + 4] = left, right, left_oo, right_oo of get_lr (src/chimera.cpp:243-293) and get_ham_endsfree (:211-239) on the
 * alignment nwalign_vectorized2(query, parent, match, mismatch, gap_p, 0, max_shift) (chimera.cpp:26,122).  The reference keeps
 * them inside C_is_bimera / BimeraTableParallel; this entry exists so that parity can be checked per alignment and not only
 * per flagged sequence (tests/, oracle_bimera_pairs). */
int dada2hip_bimera_pairs(int32_t n, const char *const *queries, const char *const *parents, int32_t allow_one_off,
                          int32_t match, int32_t mismatch, int32_t gap_p, int32_t max_shift, int32_t device, int32_t *out,
                          char *errbuf, size_t errlen);

/* ---- dereplication front-end: the step before dada() ----------------------------------------
 * dada2hip_derep_fastq == derepFastq(fl, n = chunk_reads, qualityType = "Auto" | offset) (R/sequenceIO.R:45-124 on top of
 * qtables2, :150-183): reads one FASTQ file (plain or gzip), dereplicates it and returns the `derep-class` content:
 *   seqs[nuniques]        the unique sequences, by decreasing abundance (ties: first chunk seen, then C-locale lexical)
 *   abundances[nuniques]  $uniques
 *   quals                 $quals as nuniques rows of maxlen doubles (row u = mean quality per position of unique u, NA past
 *                         its length) — which IS the column-major maxlen x nuniques matrix dada2hip_dada_uniques takes
 *   map[nreads]           $map, 0-BASED unique index per read (DADA2HIP_NA_INTEGER for zero-length reads)
 * chunk_reads <= 0 means the reference's default 1e6; qual_offset 33 / 64, or 0 for "Auto" (ShortRead's rule: a character
 * below ';' anywhere in the first chunk => Phred+33, else +64).  Host-side: no device is touched.
 * dada2hip_sample_from_derep uploads the object as a resident sample without another host copy. */
typedef struct dada2hip_derep dada2hip_derep;
int dada2hip_derep_fastq(const char *path, int64_t chunk_reads, int32_t qual_offset, dada2hip_derep **out, char *errbuf,
                         size_t errlen);
int32_t dada2hip_derep_nuniques(const dada2hip_derep *d);
int64_t dada2hip_derep_nreads(const dada2hip_derep *d);
int32_t dada2hip_derep_maxlen(const dada2hip_derep *d);
const char *const *dada2hip_derep_seqs(const dada2hip_derep *d);
const int32_t *dada2hip_derep_abundances(const dada2hip_derep *d);
const double *dada2hip_derep_quals(const dada2hip_derep *d);
const int32_t *dada2hip_derep_map(const dada2hip_derep *d);
void dada2hip_derep_free(dada2hip_derep *d);
int dada2hip_sample_from_derep(const dada2hip_derep *d, const uint8_t *priors, int32_t device, dada2hip_sample **out,
                               char *errbuf, size_t errlen);

/* ---- mergePairs: denoised forward + reverse reads -> merged amplicons -------------------------
 * dada2hip_merge_pairs == the per-sample body of mergePairs(dadaF, derepF, dadaR, derepR, minOverlap, maxMismatch,
 * returnRejects=TRUE, justConcatenate, trimOverhang) (R/paired.R:113-190).  fwd[i] / rev[i] = dadaF$map[derepF$map][i] /
 * dadaR$map[derepR$map][i]: the 1-BASED denoised-sequence index of read pair i, or DADA2HIP_NA_INTEGER.  seqsF / n0F
 * (seqsR / n0R) are dadaF$clustering$sequence / $n0.  Every unique (forward, reverse) pair is aligned on the device as
 * R's nwalign(F, rc(R), band=-1) does (C_nwalign -> nwalign_endsfree, unbanded, scores 1/-64/-64 when maxMismatch == 0
 * else 1/-8/-8, paired.R:152-159), then C_eval_pair / C_pair_consensus (src/evaluate.cpp:73,124) on the host.
 * Rows come back in the reference's order (first appearance, stably sorted by decreasing abundance) INCLUDING the rejects
 * (accept == 0, sequence ""): returnRejects=FALSE is a filter on `accept`.  prefer is NA with just_concatenate. */
typedef struct dada2hip_mergers dada2hip_mergers;
int dada2hip_merge_pairs(int64_t nreads, const int32_t *fwd, const int32_t *rev, int32_t nF, const char *const *seqsF,
                         const int32_t *n0F, int32_t nR, const char *const *seqsR, const int32_t *n0R, int32_t min_overlap,
                         int32_t max_mismatch, int32_t trim_overhang, int32_t just_concatenate, int32_t device,
                         dada2hip_mergers **out, char *errbuf, size_t errlen);
int32_t dada2hip_mergers_nrow(const dada2hip_mergers *m);
const char *dada2hip_mergers_sequence(const dada2hip_mergers *m, int32_t i);
const int32_t *dada2hip_mergers_abundance(const dada2hip_mergers *m);
const int32_t *dada2hip_mergers_forward(const dada2hip_mergers *m);
const int32_t *dada2hip_mergers_reverse(const dada2hip_mergers *m);
const int32_t *dada2hip_mergers_nmatch(const dada2hip_mergers *m);
const int32_t *dada2hip_mergers_nmismatch(const dada2hip_mergers *m);
const int32_t *dada2hip_mergers_nindel(const dada2hip_mergers *m);
const int32_t *dada2hip_mergers_prefer(const dada2hip_mergers *m);
const int32_t *dada2hip_mergers_accept(const dada2hip_mergers *m);
void dada2hip_mergers_free(dada2hip_mergers *m);

/* One b_compare round exposed for kernel-level parity tests and for bench.py's roofline leg:
 * compares every unique of `s` against unique `centre` exactly as CompareParallel does
 * (src/cluster.cpp:90-149) with the given k-mer cutoff and greedy-skip mask (skip[i] != 0 => NULL
 * sub); writes lambda[nraw], hamming[nraw] (0xFFFFFFFF for NULL subs) and cls[nraw]
 * (0 skipped, 1 shrouded, 2 gapless, 3 NW). */
int dada2hip_sample_compare(dada2hip_sample *s, int32_t centre, const double *err, int32_t err_ncol,
                            const dada2hip_opts *opts, double kdist_cutoff, const uint8_t *skip,
                            double *lambda, uint32_t *hamming, uint8_t *cls, dada2hip_stats *stats,
                            char *errbuf, size_t errlen);

/* Poisson tail used for the abundance p-value: calc_pA (src/pval.cpp:44-64) evaluated by the
 * device kernel for n (reads, E) pairs — for parity tests against the oracle. */
int dada2hip_calc_pA(int32_t n, const int32_t *reads, const double *E_reads, const uint8_t *prior, int32_t device,
                     double *out, char *errbuf, size_t errlen);

const char *dada2hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DADA2HIP_H */
