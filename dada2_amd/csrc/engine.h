// engine.h — shared declarations between the HIP kernels (kernels.hip) and the host driver
// (driver.cpp / capi.cpp) of libdada2hip.so.  Host language is C++ because the reference's
// host side for this path is compiled C++ (Rcpp); see include/dada2hip.h for the boundary.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/dada2hip.h"

namespace d2 {

constexpr int KMER_SIZE = 5;       // /root/reference/src/dada.h:27
constexpr int NKMER = 1024;        // 4^5
constexpr int RANK_SAT = 63;       // 6-bit occurrence rank packed above the 10-bit k-mer id
constexpr int MAX_SHUFFLE = 10;    // dada.h:30
constexpr int SEQLEN = 9999;       // dada.h:24
constexpr double TAIL_APPROX_CUTOFF = 1e-7;  // dada.h:25

// comparison classes written by the screen kernel
enum : uint8_t { CLS_SKIP = 0, CLS_SHROUD = 1, CLS_GAPLESS = 2, CLS_NW = 3 };

struct DeviceError {
  int code;
  std::string msg;
};

#define D2_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      throw d2::DeviceError{DADA2HIP_ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(_e) +  \
                                                     " at " __FILE__ ":" + std::to_string(__LINE__)}; \
  } while (0)

// Uniques of one sample, resident in HBM (layout: DESIGN.md §3).
struct SampleDev {
  int32_t N = 0, maxlen = 0, minlen = 0;
  // the uniques this process works on: [r_lo, r_hi) = [0, N) normally; one contiguous block of them when the sample is
  // sharded over several GPUs (dada2hip_sample_run_sharded: every rank holds all sequences, each does its block's
  // comparisons, shuffles and p-values - the host-driven round loop)
  int32_t r_lo = 0, r_hi = 0;
  int32_t W2 = 0;   // u32 words per 2-bit packed sequence row (multiple of 4 -> 16 B aligned rows)
  int32_t LQ = 0;   // bytes per quality row (multiple of 16)
  int32_t LK = 0;   // u16 entries per ordered-k-mer row (multiple of 8 -> 16 B)
  int32_t HMAX = 0; // heavy-k-mer slots per unique ((maxlen-4)/64)
  uint32_t *seq2 = nullptr;   // [N][W2]   base p: word p/16, bits 2*(p%16); codes A,C,G,T = 0..3
  uint8_t *qual = nullptr;    // [N][LQ]   (uint8) round(mean quality)  (containers.cpp:34)
  uint16_t *kord = nullptr;   // [N][LK]   k-mer id (10 bits) | min(occurrence rank, 63) << 10
  uint32_t *heavy = nullptr;  // [N][HMAX] k-mers occurring > 63 times: id | count << 16
  uint8_t *nheavy = nullptr;  // [N]
  // which of the 1024 5-mers occur in the unique at all (one bit each, 128 B per unique) and how many of its k-mer positions
  // repeat an earlier one (kmult = positions - distinct 5-mers): popcount(bits_raw & bits_centre) + kmult is an UPPER bound of the
  // k-mer overlap sum_k min(a_k, b_k) of kmers.cpp:29-48, which lets the batch screen call most pairs "shrouded" from 128 bytes
  // without walking the unique's k-mer record (k2_screen_multi).  nullptr: no prefilter (long reads: the bitmaps saturate)
  uint32_t *kbits = nullptr;  // [N][32]
  uint16_t *kmult = nullptr;  // [N]
  int32_t *len = nullptr;     // [N]
  uint32_t *reads = nullptr;  // [N]
  uint8_t *prior = nullptr;   // [N]
  int32_t *nw_flag = nullptr; // [1] set by the NW kernels when a traceback leaves its bounds (never silently wrong)
  // traceback-pointer ring of the anti-diagonal aligner k_nw_ad: one slot per wave of its grid, [16-step block][lane]
  // (every flush is one coalesced 256-byte store per wave)
  uint32_t *ad_ptr = nullptr;
  int32_t ad_waves = 0;       // wave slots (a multiple of 4)
  int32_t ad_wpw = 0;         // words per wave slot = 64 * ceil((2 * maxlen + 1) / 16)
  // lambda of k_nw_ad's alignments is multiplied up by a kernel of its own, one LANE per alignment (k_ad_product): the aligner
  // leaves each alignment's per-position factor offsets (u16 byte offsets into err) in row `id` of ad_foff and a descriptor
  // (where lambda goes, how many positions) in ad_desc[id]; id = the alignment's work slot in the launch, < ad_fcap
  uint16_t *ad_foff = nullptr;
  struct AdDesc *ad_desc = nullptr;
  int32_t ad_fcap = 0, ad_fstride = 0;
};

struct AdDesc { long long dest; int32_t L2, pad; };   // dest < 0: nothing to do (consumed / never written)

struct AlignParams {
  int32_t match, mismatch, gap, band, sentinel;
  int32_t use_quals, ncol;
  // the two aligners beside nwalign_endsfree / nwalign_vectorized2 (lane kernels k_nw / k_nw_gen only):
  int32_t homo_gap = 0;   // gap opposite a base of a homopolymer run of >= 3 (nwalign_endsfree_homo, nwalign_endsfree.cpp:220-396);
                          // equal to `gap` = no homopolymer gapping
  int32_t endsfree = 1;   // 0: global nwalign (nwalign_endsfree.cpp:403-537): end gaps cost `gap`, no free moves on the last row / column
  // C_nwvec on strings with letters outside ACGT (it compares raw bytes, nwalign_vectorized.cpp:165): every pair's letters are
  // renumbered 0..15 by the host and each string stored as TWO 2-bit rows - row r holds the low, row r + hi_off the high two
  // bits of its codes; two positions match when both planes do.  0: plain 2-bit rows.  Honoured by k_nw_gen<PAIRS> only.
  int32_t hi_off = 0;
  bool plain() const { return endsfree && homo_gap == gap; }   // (whoever fills the struct sets homo_gap = gap for the plain aligner)
};

struct ScreenParams {
  int32_t use_kmers, gapless, band, sse;
};

// Device-resident partition state (B / Bi / Raw bookkeeping fields of dada.h:65-123 as SoA).
struct PartState {
  double *E_minmax = nullptr;     // [N]  Raw::E_minmax
  uint8_t *lock = nullptr;        // [N]  Raw::lock
  int32_t *clust_of = nullptr;    // [N]  partition the unique currently belongs to
  int32_t *comp_i = nullptr;      // [N]  Raw::comp (i, lambda, hamming)
  double *comp_lam = nullptr;
  uint32_t *comp_ham = nullptr;
  double *p = nullptr;            // [N]  Raw::p
  uint8_t *slot0 = nullptr;       // [N]  occupies slot 0 of its partition (b_bud skips r = 0, cluster.cpp:285)
  int32_t *head = nullptr;        // [N]  head of the unique's list of stored comparisons
  int32_t *node_i = nullptr;      // [cap] Comparison nodes (Bi::comp entries), appended round by round
  double *node_lam = nullptr;
  uint32_t *node_ham = nullptr;
  int32_t *node_next = nullptr;
  int32_t *node_count = nullptr;
  int32_t node_cap = 0;
  uint32_t *creads = nullptr;     // [C]  Bi::reads
  int32_t *centre_of = nullptr;   // [C]  Bi::center
  uint8_t *update_e = nullptr;    // [C]
  uint8_t *check_locks = nullptr; // [C]
  int32_t *err_flag = nullptr;
  unsigned long long *totals = nullptr;   // [4] run totals of the screen counters
};

struct BudParams {
  double min_fold, omegaA, omegaP;
  int32_t min_hamming, min_abund;
};

// (uint8) round(x) of one quality row, branch-free (hostsimd.cpp: plain C++ with an AVX2 clone); false = redo the row by the scalar rule
bool round_quality_row(const double *src, uint8_t *dst, int L, int *mx_out);

void launch_fill_f64(double *d_p, size_t n, double v, hipStream_t st);
void launch_fill_null(int n, const uint8_t *d_cls, double *d_lam, uint32_t *d_ham, hipStream_t st);
void launch_store(const PartState &P, const SampleDev &S, int ci, int centre, double total_reads, const double *d_lam,
                  const uint32_t *d_ham, const int32_t *d_round_counters, const uint8_t *d_cls, int32_t *d_zero2, hipStream_t st);
// The store filter of a round (ci >= 1) rides in front of the round's first shuffle: pass its arguments here.
struct StoreRound {
  int ci, centre;
  double total_reads;
  const double *lam;
  const uint32_t *ham;
  const int32_t *round_counters;
  const uint8_t *cls;
};
// check_only: count would-be movers, apply nothing (d_creads_snap may then be the live reads)
void launch_shuffle(const PartState &P, const SampleDev &S, const uint32_t *d_creads_snap, int32_t *d_movers,
                    int32_t *d_nmovers, int32_t *d_inline, const StoreRound *store, int check_only, int nclust, hipStream_t st);
// result block of one b_bud evaluation, fetched by the host in a single copy
constexpr int BUD_TIES = 64;     // tie records published inline with the round result
constexpr int TIES_FULL = 4096;  // ... and kept in full on the device (engine v2) for the host to fetch when there are more
struct BudTie { int32_t raw, comp_i; uint32_t comp_ham; int32_t from; uint32_t from_reads, pad; double comp_lam; double p; };
// Candidates listed by k_bud_ties: the exact (p, reads) ties of the device's best key AND every other candidate whose
// device p-value lies within BUD_NEAR (relative) of the best one.  Device and host libm differ in the last ulp of
// exp / log / lgamma, so the order of two DIFFERENT candidates that close is not trusted: the host re-evaluates the
// listed candidates with its own libm (the CPU reference's arithmetic) and takes b_bud's decision itself.
constexpr double BUD_NEAR = 1e-9;
struct BudOut {
  double best_p[2];
  uint32_t best_reads[2];
  int32_t found[2], nties[2];
  int32_t err_flag, node_count;
  int32_t valid;               // 0 when a speculative evaluation was cancelled on the device
  int32_t auto_applied;        // 1 when k_auto_birth applied the (unambiguous) birth on the device
  BudTie ties[2][BUD_TIES];
};
// Everything the host needs from one round tail, fetched with a single copy.
constexpr int MOVERS_INLINE = 512;
struct RoundOut {
  int32_t cnt[2];              // [slot of the real shuffle] = its movers, [other] = uniques one more call would move (zeroed by
                               // k_apply_bud / k_auto_birth ahead of the round)
  int32_t seq;                 // host copy only: sequence number k_auto_birth writes last when it publishes the block
  int32_t pad;
  BudOut bud;
  int32_t mov[2][3 * MOVERS_INLINE];   // first movers of each shuffle (raw, from, to); the full lists stay on the device
  int32_t pad_tail[2];         // (size is a multiple of 16: the block is published as uint4s)
};
// fused [would one more shuffle move anything? ->] b_p_update + first stage of b_bud, then the reduction + tie listing.
// d_check_cnt (optional, zeroed): receives the number of uniques one more b_shuffle2 call would move; non-zero voids
// the evaluation on the device.
void launch_pupdate_bud(const PartState &P, const SampleDev &S, int greedy, int detect_singletons, const BudParams &bp,
                        double init_p, uint32_t init_reads, void *d_partial, BudOut *d_out, int32_t *d_over0, int32_t *d_over1,
                        int nclust, uint8_t *d_lock_tmp, int32_t *d_check_cnt, hipStream_t st);
// birth + the new centre's k-mer record for the coming round (one launch)
void launch_apply_bud(const PartState &P, const SampleDev &S, uint32_t *d_creads_snap, int raw, int newi, int from,
                      uint32_t reads_new, uint32_t reads_from, uint32_t *d_ctab, int32_t *d_zero2, hipStream_t st);
// the unambiguous birth applied on the device (d_next[0] = its unique, or -1): lets the next round start before the host looks
// It finally publishes the round's result block to pinned host memory (h_block, seq written last): no copy, no sync.
void launch_auto_birth(const PartState &P, const SampleDev &S, uint32_t *d_creads_snap, RoundOut *d_block, double omegaA, int newi,
                       uint32_t *d_ctab, int32_t *d_zero2, int32_t *d_next, RoundOut *h_block, int seq, hipStream_t st);
void launch_centre_table(const SampleDev &S, int centre, uint32_t *d_ctab, hipStream_t st);
void launch_final_p(const PartState &P, const SampleDev &S, double omegaC, uint8_t *d_correct, hipStream_t st);
void launch_posthoc(const PartState &P, const SampleDev &S, const int32_t *d_cluster_of_centre, int32_t *d_out_ji, double *d_out_lam,
                    int32_t *d_nout, int cap, hipStream_t st);

// launch wrappers implemented in kernels.hip -----------------------------------------------------
void launch_build_kmers(const SampleDev &S, hipStream_t st);
// counters: [0]=#NW work items, [1]=#gapless work items, [2]=#shrouded, [3]=#skipped
void launch_screen(const SampleDev &S, int centre, const ScreenParams &sp, const uint8_t *d_skip, const uint8_t *d_lock,
                   int greedy, const int32_t *d_thresh, uint8_t *d_cls, double *d_lambda, uint32_t *d_ham, int32_t *d_nw_list,
                   int32_t *d_gl_list, int32_t *d_counters, uint32_t *d_ctab, bool build_table, const int32_t *d_centre_dev,
                   hipStream_t st);
void launch_gapless(const SampleDev &S, int centre, const int32_t *d_chunk_centre, const int32_t *d_work,
                    const int32_t *d_nwork, int nwork_host, const AlignParams &ap, const double *d_err,
                    double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV, int view_by_chunk, hipStream_t st);
void launch_pair_class(const SampleDev &S, const int32_t *d_pc, const int32_t *d_pr, int n, const ScreenParams &sp,
                       uint8_t *d_out, hipStream_t st);

struct NwScratch {
  uint32_t *ptr = nullptr;  // traceback pointers  [wave][row][NPW][64]
  uint32_t *tcode = nullptr;  // transition nibbles  [wave][word][64]
  int32_t *rows = nullptr;    // generic kernel: DP row [wave][Wgen][64]
  size_t ptr_words_per_wave = 0, t_words_per_wave = 0, row_words_per_wave = 0;
  int nwaves = 0;
};
// Returns the WMAX class that will be used for (band, maxlen, minlen): 33, 65, 129 or 0 (generic).
int nw_class(int band, int maxlen, int minlen);
size_t nw_ptr_words_per_wave(int wclass, int band, int maxlen, int minlen);
void launch_nw(const SampleDev &S, int wclass, int centre, const int32_t *d_chunk_centre, const int32_t *d_work,
               const int32_t *d_nwork, int nwork_host, const AlignParams &ap, const double *d_err, const NwScratch &scr,
               double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV, int view_by_chunk, uint8_t *d_moves,
               int moves_stride, int32_t *d_nmoves, hipStream_t st, const int32_t *d_pair_centre = nullptr);   // d_pair_centre: a centre per work item

// batch mode of k_nw_ad (round engine v2): the comparisons of a whole batch compare in one launch, everything read on the device
struct NwBatch {
  const int32_t *on;        // centres in the batch compare in flight (0: the launch has nothing to do)
  const int32_t *n;         // [2 KB_MAX] lengths of the lists below
  const int32_t *list;      // [2 KB_MAX][stride]: uniques to align with batch centre k (row k), its gapless ones (row KB_MAX + k)
  const int32_t *centre;    // [KB_MAX]
  const int32_t *bbuf;      // batch buffer: results go to row (*bbuf * KB_MAX + k) of d_lambda / d_ham, rows of `stride` entries
  size_t stride;
  // the pointer-free pass (k_nw_ad<.., FAST>) hands the pairs it cannot finish to the full kernel through these: row k of
  // retry_list ([KB_MAX][stride]) / retry_n[k], zeroed by k2_batch_lists.  nullptr: one launch of the full kernel as before
  int32_t *retry_list = nullptr;
  int32_t *retry_n = nullptr;
  unsigned long long *fast_ctl = nullptr;   // [0] pairs handed over so far, [1] pairs the pass has looked at, [2] != 0: the pass is off (Eng2::fast_ctl)
};
void launch_gapless_batch(const SampleDev &S, const NwBatch &b, const AlignParams &ap, const double *d_err, double *d_lambda,
                          uint32_t *d_ham, const int32_t *d_stop_dev, hipStream_t st);
// d_gl_work/d_gl_nwork (optional): the round's gapless comparisons, processed by the same kernel
// d_view (optional): aligned views, row = unique (or chunk when view_by_chunk); chunks are nw_ad_apw() work slots
void launch_nw_ad(const SampleDev &S, int centre, const int32_t *d_chunk_centre, const int32_t *d_work,
                  const int32_t *d_nwork, int nwork_host, const int32_t *d_gl_work, const int32_t *d_gl_nwork,
                  const AlignParams &ap, const double *d_err, double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV,
                  int view_by_chunk, const int32_t *d_centre_dev, hipStream_t st, const int32_t *d_stop_dev = nullptr,
                  const NwBatch *batch = nullptr);
int nw_ad_apw(const SampleDev &S, const AlignParams &ap);
size_t nw_ad_lds_bytes(const SampleDev &S, const AlignParams &ap);
// bimera mode (chimera.cpp): pairs (query = chunk centre, parent = work item) reduced in the kernel to get_lr / get_ham_endsfree
void launch_nw_ad_lr(const SampleDev &S, const int32_t *d_chunk_centre, const int32_t *d_work, int nwork, const AlignParams &ap,
                     const double *d_err, int allow_one_off, int max_shift, int32_t *d_out, hipStream_t st);
size_t nw_ad_lr_lds_bytes(const SampleDev &S, const AlignParams &ap);

// wide-band / long-read anti-diagonal kernel (8 band cells per lane, pointers in an HBM ring of `scr_waves` wave slots)
bool nw_adw_ok(const SampleDev &S, const AlignParams &ap);
size_t nw_adw_ptr_words_per_wave(const SampleDev &S, const AlignParams &ap);
int nw_adw_waves(const SampleDev &S, const AlignParams &ap, int nwork);
size_t nw_adw_lds_bytes(const SampleDev &S, const AlignParams &ap);
void launch_nw_adw(const SampleDev &S, int centre, const int32_t *d_chunk_centre, const int32_t *d_work, const int32_t *d_nwork,
                   int nwork_host, const AlignParams &ap, const double *d_err, uint32_t *d_ptr_scr, size_t ptr_wpw,
                   int scr_waves, double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV, int view_by_chunk,
                   hipStream_t st);
int nw_adw_apw(const SampleDev &S, const AlignParams &ap);

// =================================================================================================================
// Round engine v2: batched multi-centre compares + device-driven rounds.
//
//  * A compare is no longer tied to its round: when the centre of the coming round has no comparisons yet, ONE pass
//    over the k-mer records screens every unique against up to KB_MAX centres at once - the coming centre plus the
//    candidates that the bud ordering makes likely to be born next - and ONE aligner launch aligns all surviving pairs
//    (about eight rounds' worth: the launch runs in the aligner's saturated regime, where an alignment costs 0.6 of what it
//    costs in a launch of one round's work).
//    The results (class, lambda, hamming per unique) stay in a cache of NBUF batches; the round of a cached centre
//    commits them without touching the k-mer records or the aligner again (HBM bytes per comparison / KB_MAX, aligner
//    launches with KB_MAX rounds of work).  Exactness: a comparison depends only on the two uniques, err and the
//    options; the greedy skip (cluster.cpp:127-130) is re-applied at commit time with the lock state of that moment
//    (locks only grow between a compare and its commit, except for the centre itself).
//  * A round is a fixed sequence of launches whose parameters (centre, partition count, cache slot, which shuffles
//    still have to run) live in a device control block: the host enqueues rounds AHEAD of the results it has seen and
//    trails the device, replaying the published moves / births on its mirror.  Whenever the decision is not the
//    device's to take (exact or near ties, prior births, no birth, more than SH_CHAIN shuffles, capacity) the device
//    halts - every later launch becomes a no-op - and the host takes over as in the classic loop.
constexpr int KB_MAX = 8;          // centres per batch compare (one byte lane each in the packed count table)
constexpr int SH_CHAIN = 4;        // b_shuffle2 calls enqueued per chain (the first unconditional, the rest guarded)
constexpr int SH_LEVELS = MAX_SHUFFLE;   // b_shuffle2 calls of one round (Rmain.cpp:321): what the persistent tail kernel runs in one go
#ifndef D2_RING2
#define D2_RING2 16
#endif
constexpr int RING2 = D2_RING2;    // result blocks in flight (device copies + pinned host copies)
constexpr int MOV_RING = 4;        // full mover lists in flight (launch chains; the persistent tail keeps one set and pauses on overflow)
constexpr int TRACE_BLOCKS = 4096, TRACE_KERNELS = 8;   // (slot 0 unused since k2_lists went into the store pass; shuffle 0..3, p-update, birth, spare)
constexpr int MOV_INLINE2 = 32768; // movers published inline per chain (all its shuffles, concatenated); only the used part is copied

// stored comparisons of one unique (Bi::comp entries that name it): the round-0 entry lives in lam0/ham0 (every
// unique has one, containers.cpp:39 + cluster.cpp:189), the next one - for most uniques of a large sample the only other
// one: the comparison with their own partition's centre - in i1/lam1/ham1 (read coalesced), any further ones in a chain of
// 64-byte blocks, newest block first
struct alignas(64) CompBlk {
  int32_t next, cnt;
  int32_t i[3];
  uint32_t ham[3];
  double lam[3];
  int32_t pad[2];
};
static_assert(sizeof(CompBlk) == 64, "CompBlk is one 64-byte line");
struct Store2 {
  double *lam0 = nullptr, *lam1 = nullptr;
  uint32_t *ham0 = nullptr, *ham1 = nullptr;
  int32_t *i1 = nullptr;          // partition of the second entry, -1 = none
  // bit (k & 63) is set for every partition k the unique holds a stored comparison with: lets a shuffle call tell "none of this
  // unique's partitions gained reads" without walking its chain (false positives only cost a look)
  unsigned long long *smask = nullptr;
  int32_t *head = nullptr;
  CompBlk *blk = nullptr;
  int32_t *blk_count = nullptr;
  int32_t blk_cap = 0;
};

enum : int32_t { H2_NONE = 0, H2_NO_BIRTH, H2_HOST_DECIDE, H2_SHUFFLE_MORE, H2_CAPACITY, H2_MAXCLUST, H2_NEED_COMPARE, H2_FAIL };

struct Ctl2 {
  int32_t state;        // 0 = running, 1 = halted (every launch returns at once)
  int32_t halt;         // H2_* of the last halt
  int32_t nclust;       // partitions that exist
  int32_t centre;       // centre (unique index) of the round in flight = centre_of[nclust - 1]
  int32_t slot;         // cache slot holding the comparisons against `centre`
  int32_t nbatch;       // centres the round's batch compare has to process (0 = cache hit, nothing to launch)
  int32_t bbuf;         // batch buffer it fills (slot = bbuf * KB_MAX + position)
  int32_t next_bbuf;    // FIFO cursor over the batch buffers
  int32_t pub_seq;      // result blocks published so far
  int32_t nsh_base;     // shuffles of the round in flight executed by earlier chains
  int32_t max_clust;
  int32_t nalign_ran;   // nalign of a compare that ran in front of a persistent launch which left at its entry (ring full, prefetch awaited) and was marked done there (k3_tail); consumed by the round's result block
  // what the aligner launches of the coming chain work on (NwBatch): positions [0, nalign) of batch buffer abuf, centre of
  // position k = acentre[k] (-1: none).  Batch mode: the batch just planned (nalign = nbatch, 0 on a cache hit).  Commit mode
  // (Eng2::align_at_commit): always the ONE position of the coming round's centre, whether it was screened just now or long ago.
  int32_t nalign, abuf;
  int32_t acentre[KB_MAX];
  int32_t need_compare; // the coming round's centre is not cached: its chain must carry the batch compare (Eng2::has_compare)
  int32_t n0, low0;     // members of partition 0 now / a lower bound of the fewest it has ever had (tie rule of k2_birth)
  int32_t bcentre[KB_MAX];
  uint32_t breads[KB_MAX];
  int32_t blen[KB_MAX];
  // persistent tail (k3_tail): leave the launch after the round in flight (a compare is due, the host's ring is nearly full, a
  // pause); what the host has consumed as far as the device knows
  int32_t kexit, hcons_seen;
  // what the first shuffle call of a round may assume: the previous round's calls ended with one that moved nothing (every
  // unique sits in its arg-max), and since then only the birth has changed reads - partition bfrom lost its new centre's
  int32_t stable, bfrom;
  // ---- the NEXT batch's compare under the persistent tail (Eng2::pf_on) ----
  // pf_seq: prefetch compares planned so far (the host launches number k on the second stream when it sees pf_seq >= k in a
  // published block; PfSync::done follows when the compare has run).  pf_bbuf: the batch buffer the latest one fills.
  // last_bbuf: the batch buffer planned last, by a miss or by a prefetch - the round whose centre is found in it plans the next
  // prefetch.  pf_wait: the coming round's centre sits in prefetch number pf_wait whose compare had not finished when the round
  // was decided: no round may start before PfSync::done >= pf_wait (0: nothing to wait for).
  int32_t pf_seq, pf_bbuf, last_bbuf, pf_wait;
  int32_t prev_bbuf;            // the batch buffer planned before last_bbuf (-1: none since the last miss)
  // statistics of the run so far: rounds whose centre came out of a prefetched batch; spins for a prefetch in flight that
  // ended in time / that ended with the launch left; centres prefetched
  int32_t pf_hits, pf_spins, pf_exits, pf_centres;
  unsigned long long pf_mask;   // bit b: batch buffer b was filled by a prefetch
};

// Results of the batch compares, kept until their centre's round comes (or the batch buffer is recycled): the class, 2 bits
// per (unique, batch position), and lambda / hamming of the pairs the aligner saw.  A wrong guess costs its share of one
// pass over the k-mer records and its alignments (about one batch position in ten is never used; profiles/README.md).
struct Cache2 {
  int32_t NBUF = 0;               // batch buffers
  uint16_t *bcls = nullptr;       // [NBUF][Npad]  2 bits per batch position: CLS_* as of the compare
  int32_t *slot_centre = nullptr; // [NBUF * KB_MAX]  unique index or -1
  uint2 *tab8 = nullptr;          // [1024]  byte k = min(count of the 5-mer in batch centre k, 63) + 0x7F
  uint16_t *full = nullptr;       // [KB_MAX][1024] full counts (heavy k-mer correction)
  uint16_t *ord = nullptr;        // [KB_MAX][LK] ordered 5-mers, 0xFFFF past the end
  uint32_t *cbits = nullptr;      // [KB_MAX][32] presence bitmaps of the batch's centres (SampleDev::kbits rows; zeros past the batch)
  // lambda / hamming of the batch's alignments, row = cache slot (batch buffer * KB_MAX + position): written by the ONE
  // aligner launch that follows a batch screen, read (sparsely: NW / gapless classes only) when the slot's round commits
  double *lamB = nullptr;         // [NBUF * KB_MAX][Npad]
  uint32_t *hamB = nullptr;       // [NBUF * KB_MAX][Npad]
  size_t Npad = 0;
};

struct Round2Out {
  int32_t seq;                    // host copy only: written last by k2_birth when it publishes the block
  int32_t halt;                   // H2_*
  int32_t nclust;                 // partitions after this block's birth, if any
  int32_t birth_applied;          // 1: bud.ties[0][0] became partition nclust - 1 on the device
  int32_t nlev;                   // shuffle launches of the chain
  int32_t nsh;                    // ... of which executed
  int32_t cnt[SH_LEVELS];         // movers of each
  int32_t nbatch;                 // centres compared by this round's batch launch (0 = hit)
  int32_t slot;
  int32_t err_flag, blk_count;
  int32_t pad0[4];
  int32_t pf_seq;                 // persistent tail with overlap: prefetch compares planned so far (Ctl2::pf_seq as of this block)
  int32_t pf_wait;                // ... and the prefetch this block's NEXT round has to wait for (Ctl2::pf_wait; 0: none)
  int32_t pf_stat[4];             // Ctl2::pf_hits / pf_spins / pf_exits / pf_centres as of this block
  int32_t pad1[2];
  int32_t kord;                   // persistent tail: ordinal of the k3_tail launch that ran this round (0: a launch chain)
  int32_t paused;                 // persistent tail: the device halted BEHIND this block's decision because its mover lists did not fit
                                  // the block (they stay in Eng2::movers until the host has fetched them and resumes)
  unsigned long long stat[4];     // commit-time classes of the round's comparisons: NW, gapless, shrouded, greedy-skipped
  BudOut bud;
  int32_t mov[3 * MOV_INLINE2];   // (unique, from, to) of the chain's movers, shuffles concatenated
};
static_assert(sizeof(Round2Out) % 16 == 0, "Round2Out is published as uint4s");

struct Eng2 {   // everything the v2 kernels share, passed by value
  PartState P;
  SampleDev S;
  Store2 T;
  Cache2 C;
  Ctl2 *ctl;
  Round2Out *dblk;                // [RING2] device-side result blocks
  Round2Out *hblk;                // [RING2] pinned host copies
  int32_t *dlt;                   // [SH_LEVELS][ccap] partition-read deltas of the chain's shuffles
  // b_bud takes the FIRST of several equal keys in partition order, then in the order of the partition's member list
  // (cluster.cpp:284-308).  The member lists live on the host (bi_pop_raw moves the LAST member into the hole,
  // containers.cpp:177-194, so a list's order is the history of every move), but one case needs no list: a unique that has
  // never moved still sits in partition 0 at the slot of its index as long as partition 0 has never been short enough to
  // make it the last member.  moved[] and the two counters per shuffle call are what k2_birth needs to recognise that case
  // and settle such a tie itself (they are most of the ties of a deep sample: p underflowed to 0, equal reads, neighbours
  // in the abundance order) instead of halting for the host.
  uint8_t *moved;                 // [N] the unique has been moved by a shuffle or a birth
  int32_t *n0d;                   // [SH_LEVELS][2] members partition 0 lost / gained in each of the chain's shuffle calls
  int32_t *movers;                // [MOV_RING][SH_CHAIN][3 N] (launch chains) / [SH_LEVELS][3 N] (persistent tail)
  uint32_t *stat_part;            // [grid of the store pass][4] per-block class counts of the round (NW, gapless, shrouded, skipped)
  int32_t *stat_n;                // number of entries in stat_part (0: the chain had no store pass), consumed by k2_birth
  int32_t *blist, *blist_n;       // [2 KB_MAX][Npad] / [2 KB_MAX]: work lists of the batch compare (NwBatch)
  int32_t *bretry, *bretry_n;     // [KB_MAX][Npad] / [KB_MAX]: NwBatch::retry_list / retry_n of that compare (nullptr: no pointer-free pass)
  unsigned long long *fast_ctl;   // [4] the run's NwBatch::fast_ctl words (shared by the compares of both streams)
  void *partial;                  // block partials of the bud arg-min
  int32_t *ties0, *ties1;         // full tie lists
  BudTie *ties_rec;               // [2][TIES_FULL] full records of the first TIES_FULL listed candidates per track
  int32_t *sig_list, *sig_n;      // significant bud candidates of the last evaluation (k2_pupdate)
  int32_t ccap;
  int32_t greedy, detect_singletons;
  double total_reads, omegaA, omegaP;
  BudParams bp;
  ScreenParams sp;
  const int32_t *thresh;
  int32_t max_shuffle;
  // phase trace of ONE round (tools/trace_round.py; DADA2HIP_V2_TRACE=<block sequence number>): every block of the round's
  // tail kernels stamps the shader clock at its phase boundaries into trace[(kernel * TRACE_BLOCKS + block) * 8 + phase]
  unsigned long long *trace;
  int32_t trace_seq;
  // A chain comes in two forms: with the four launches of a batch compare in front (screen, work lists, gapless pairs,
  // aligner) or without.  Seven rounds in eight find their centre cached, and each of those launches costs 4.6 us even when
  // it has nothing to do, so the host leaves them out of the chains it expects to be cache hits (it knows how many centres the
  // last batch held).  When it guessed wrong - the prediction of the coming centres failed early - the chain's kernels see
  // need_compare without has_compare, do nothing, and k2_birth reports H2_NEED_COMPARE: the host sends a full chain.
  int32_t has_compare;
  // When are the pairs of a cached screen aligned?  0: all positions of a batch at once, right behind its screen - a launch
  // of eight rounds' work runs in the aligner's saturated regime, at the price of the pairs a later greedy skip or an unused
  // position wastes (10 % at 250 nt).  1: each centre's pairs when its round commits, with the greedy skip of that moment -
  // nothing is wasted; chosen when one round's list fills the device on its own (one alignment per wave: long reads, where
  // aligning ahead cost 20 % more alignments and saved nothing, profiles/r03z_bench_cfg5.json vs r04a).
  int32_t align_at_commit;
  int32_t sh_filter;                                // later shuffle calls of a chain visit only the uniques the previous call can have unsettled
  int32_t grid_shuffle, grid_pupdate;               // host side: block caps of the per-round launches (tuning knobs)
  int32_t mov_inline;                               // movers published inline with a round's result block (<= MOV_INLINE2; test knob)
  int32_t ring_limit;                               // result blocks the device may be ahead of the host (<= RING2; test knob)
  // ---- persistent round tail (k3_tail, rounds3.inc.hip) ----
  struct PSync *psync;                              // grid barrier of the launch
  volatile int32_t *hcons;                          // pinned host word: result blocks the host has finished with
  volatile int32_t *hexit;                          // pinned host word: ordinal of the last k3_tail launch that has ended
  unsigned long long *ktime;                        // [KT_N] phase clocks of block 0 (DADA2HIP_PROFILE=1), else nullptr
  int32_t xbar;                                     // 1: XCD-hierarchical barriers inside a launch (PSync; DADA2HIP_V3_XBAR, default: from 48 blocks on)
  int32_t spec_max_prev;                            // ... when the call before moved at most this many uniques (DADA2HIP_V3_SPEC_MAX)
  int32_t spec_eval;                                // 1: the shuffle calls behind the commit's carry the round's evaluation (shuffle_body<.., SPEC>; DADA2HIP_V3_SPEC)
  int32_t fail_ordinal;                             // test knob (DADA2HIP_V3_FAIL_ENTRY): the k3_tail launch of this ordinal fails its entry barrier (0: none)
  unsigned long long grid_wait_ticks;               // bound of a grid-barrier wait (100 MHz ticks; the host scales it with the sample: 2 s + 1 s per 10^6 uniques)
  // The locks an evaluation ATTEMPT on a shuffle call decides (pval.cpp:29-36) are not written to PartState::lock while it is unknown
  // whether the attempt stands: the prefetch compare of the second stream reads lock[] at any time, and a lock that is published
  // and then taken back breaks "locks only grow between a compare and its commit" (ADVICE r5).  Block b collects them in
  // spec_lock_buf[b * spec_lock_stride ...] and writes them out behind the barrier that made the attempt stand (k3_tail).
  int32_t *spec_lock_buf;
  int32_t spec_lock_stride;                         // entries per block = uniques one block sweeps
  // 1: every block of the persistent tail keeps, for the life of a launch, one word per unique it sweeps in LDS (the dynamic LDS
  // behind TailLds: rounds3.inc.hip, "the mirror") - partition, "holds a second stored comparison", lock, "p != 1": what PASS A of
  // the shuffle and evaluation sweeps asks of every unique.  A unique is swept, and those facts written, by ONE block only (the
  // birth's new centre excepted, which its owner patches behind the round's last barrier), so the global arrays stay the truth
  // (every write goes to both) and the mirror is refilled at every launch entry.  DADA2HIP_V3_MIRROR=0 turns it off.
  int32_t mirror_on;
  // ---- the next batch's compare under the persistent tail: what a prefetch compare works with.  The compare
  //      kernels of the second stream get a copy of this block whose `ctl`, `C.tab8 / full / ord`, `blist / blist_n` and aligner
  //      scratch ARE these (so they run unchanged); the tail's planner fills pf_ctl and clears pf_blist_n ----
  int32_t pf_on;                                    // 1: the tail plans prefetch compares
  int32_t pf_plan;                                  // 1: this LAUNCH may plan further prefetches (0: another sample of the process is in flight on the device now; what is planned already is still waited for and used)
  int32_t pf_min;                                   // fewest uncached candidates worth a prefetch pass
  int32_t pf_sync;                                  // measurement knob: wait for every prefetch at the next serial end (DADA2HIP_V3_PF_SYNC)
  int32_t pf_early;                                 // the next prefetch is planned when the rounds are this many positions into the batch BEFORE the one planned last (KB_MAX: only when they reach the last one)
  Ctl2 *pf_ctl;                                     // descriptor of the prefetch compare (nbatch, bbuf, bcentre / breads / blen, nalign, abuf, acentre; state stays 0)
  int32_t *pf_blist_n;                              // [2 KB_MAX] lengths of its work lists
  struct PfSync *pfsync;                            // PfSync::done = prefetch compares that have run
  unsigned long long pf_wait_ticks;                 // how long (100 MHz ticks) a round waits inside the launch for a prefetch in flight
  unsigned long long pf_gate_ticks;                 // how long the gate of a chain enqueued ahead waits for its plan (k2_pf_gate)
};

// Written by the last kernel of a prefetch compare (k2_pf_done, second stream), polled by the persistent tail.
struct PfSync {
  uint32_t done, pad0[31];
  unsigned long long nnw, ngapless;                 // pairs the prefetch compares' aligner launches worked through (run totals)
};

// Grid barrier of the persistent tail kernel: one monotonic arrival counter and one generation word, each on a cache line of
// its own (rounds3.inc.hip).  Zeroed when a run starts; a launch continues where the previous one left them.
// From 48 blocks on the barriers INSIDE a launch are XCD-hierarchical (Eng2::xbar; the guide's barrier-xcd): the blocks are grouped
// by the XCC they run on (hardware register, not blockIdx), each group has an arrival counter and a generation word of its own,
// and only a group's last arriver writes the XCC's L2 back and arrives on `top`.  These words are per launch: the entry barrier
// (always the flat one) counts the blocks of each XCC in xcount, and its last arriver turns that into xn / ngrp and clears the rest.
struct PSync {
  uint32_t arrive, pad0[31];
  uint32_t gen, pad1[31];
  uint32_t fail, pad2[31];
  uint32_t top, pad3[31];
  uint32_t ngrp, pad4[31];
  uint32_t xarr[8][32], xgen[8][32], xcount[8][32], xn[8][32];
};
enum { KT_S0 = 0, KT_S0_BAR, KT_SL, KT_SL_BAR, KT_P, KT_P_BAR, KT_BIRTH, KT_PUBLISH, KT_ROUNDS, KT_LEVELS, KT_LAUNCH, KT_RELEASE,
       KT_PFWAIT /* serial end: spinning for a prefetch compare in flight */, KT_PLAN /* planning the next prefetch */, KT_SUB_LAST = 14, KT_SUB = 16, KT_N = 80 };

void launch2_store0(const Eng2 &E, const double *d_lam, const uint32_t *d_ham, const uint8_t *d_cls, const int32_t *d_round_counters,
                    hipStream_t st);
void launch2_screen_multi(const Eng2 &E, hipStream_t st, bool beside_tail = false);   // beside_tail: the 80-register build (prefetch compares)
// prefetch compare (second stream): the k-mer tables of the batch pf_ctl describes (the planner only chose its centres) in front
// of the screen, the completion word behind the aligner.  E = the prefetch's argument block (see Eng2::pf_on)
void launch2_pf_gate(const Eng2 &E, int k, const int32_t *h_quit, int32_t *h_result, hipStream_t st);   // first kernel of chain k: waits for plan k (pinned quit / result words)
void launch2_pf_tables(const Eng2 &E, hipStream_t st);
void launch2_pf_done(const Eng2 &E, hipStream_t st);
void launch2_batch_lists(const Eng2 &E, hipStream_t st);                              // classes of a batch screen -> the aligner's work lists
void launch2_shuffle(const Eng2 &E, int level, bool store, hipStream_t st);
// b_p_update + b_bud arg-min (grid) ; ties, decision, birth, plan of the coming round, publication (one block)
void launch2_eval(const Eng2 &E, int nlev, uint32_t init_reads, hipStream_t st);
void launch2_host_birth(const Eng2 &E, int raw, int from, hipStream_t st);            // the host's decision applied + plan; resumes
void launch2_resume(const Eng2 &E, hipStream_t st, bool keep_list = false, bool compare_done = false);   // keep_list: the candidates k2_pupdate listed stay valid
void launch2_posthoc(const Eng2 &E, const int32_t *d_cluster_of_centre, int32_t *d_out_ji, double *d_out_lam, int32_t *d_nout,
                     int cap, hipStream_t st);
// the persistent round tail: rounds run back to back inside ONE launch of `grid` co-resident blocks until a compare is due, the
// device halts or the host's ring fills up.  first: the evaluation behind round 0 (no shuffle).  ordinal: this launch's number.
int tail_grid(int N, int device);
int tail_resident_max(int device, int bs);                  // blocks of k3_tail the device can hold at once (occupancy query; 0 = unknown)
int tail_mirror_cap(int device, int bs);                    // uniques per block the tail's LDS mirror holds (Eng2::mirror_on; 0: its LDS does not fit this part)
void launch3_tail(const Eng2 &E, int grid, int bs, bool first, int ordinal, uint32_t init_reads, hipStream_t st);   // bs: 1024 or 512 threads per block

// get_lr + get_ham_endsfree (chimera.cpp:211-293) on the move strings k_nw left behind: out[slot] = {left, right, left_oo, right_oo, ham}
void launch_bimera_lr(const SampleDev &S, const int32_t *d_chunk_centre, const int32_t *d_work, int nwork, const uint8_t *d_moves,
                      int stride, const int32_t *d_nmoves, int allow_one_off, int max_shift, int32_t *d_out, hipStream_t st);

void launch_calc_pA(int n, const int32_t *d_reads, const double *d_E, const uint8_t *d_prior, double *d_out,
                    hipStream_t st);

// final tables from the per-unique aligned views (error.cpp:131-172, :225-258)
void launch_final_tables(const SampleDev &S, const uint16_t *d_view, int LV, const int32_t *d_work, int nslots,
                         const int32_t *d_cluster_of, const int32_t *d_centre_of_cluster, const uint8_t *d_correct, int ncol,
                         int has_quals, int32_t *d_trans, unsigned long long *d_qsum, uint32_t *d_qn, int32_t *d_nsubs,
                         int nclust, hipStream_t st);

}  // namespace d2
