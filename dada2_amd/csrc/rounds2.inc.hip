// rounds2.inc.hip — kernels of round engine v2 (engine.h); included by kernels.hip inside namespace d2.
//   k2_screen_multi   kmer_dist_SSEi_8 / kord_dist_SSEi + raw_align's dispatch against up to KB_MAX centres per pass
//                     (kmers.cpp:29-150, nwalign_endsfree.cpp:10-73)
//   k2_store0         store filter of round 0 (cluster.cpp:179-201, every unique is kept: E_minmax starts at -999)
//   k2_batch_lists    the aligner's work lists of a batch compare (NwBatch)
//   k2_shuffle        commit of a later round's cached comparisons (greedy skip of cluster.cpp:127-130 as of now, store filter
//                     :179-201) + b_shuffle2 (:210-266)
//   k2_pupdate/_ties  b_p_update + b_bud's arg-min (pval.cpp:14-40, cluster.cpp:274-310)
//   k2_birth          the unambiguous birth (cluster.cpp:313-347), the plan of the coming round's compare, publication
// Every launch reads what it has to do from the device control block (Ctl2): the host enqueues rounds ahead of the
// results it has seen.

#ifndef D2_PUPD_INLINE
#define D2_PUPD_INLINE __forceinline__
#endif
#ifndef D2_SHUF_INLINE
#define D2_SHUF_INLINE __forceinline__
#endif
// phase stamps of the traced round (Eng2::trace; nullptr in every normal run)
#define D2_TRACE2(KID, SUBKID, PHASE)                                                                              \
  do {                                                                                                             \
    if (E.trace && threadIdx.x == 0 && (int)blockIdx.x < TRACE_BLOCKS && E.ctl->pub_seq == E.trace_seq)           \
      E.trace[((size_t)(KID) * TRACE_BLOCKS + blockIdx.x) * 8 + (PHASE)] = gcn_clock();                            \
    D2_KSUB(SUBKID, PHASE, blockIdx.x == 0);                                                                       \
  } while (0)
#define D2_TRACE(KID, PHASE) D2_TRACE2(KID, KID, PHASE)
// (a shuffle call: trace slot 1 + level; block 0's sub-phase clocks go to kid 1 for the commit's call and kid 2 for every later one)
#define D2_TRACE_SH(LEVEL, PHASE) D2_TRACE2(1 + (LEVEL), (LEVEL) == 0 ? 1 : 2, PHASE)
// persistent tail under DADA2HIP_PROFILE=1: what block 0 (the deciding block for the serial section) spends between the stamps
// of a phase body, summed over the run: ktime[KT_SUB + 8 kid + phase] (kid 1 = commit + first shuffle call, 2 = later calls,
// 5 = p-update, 6 = serial end of the round)
// (the persistent tail accumulates in LDS - D2_KT = s_ktime, flushed to Eng2::ktime once per round by k3_tail: a stamp that is a
//  read-modify-write of global memory costs the stamping lane a round trip, which the rest of its block then waits for at the
//  next block barrier - with seven stamps per shuffle call the clocks measured mostly themselves)
#ifdef D2_TAIL_TU
static __shared__ unsigned long long s_ktime[KT_N];
#define D2_KT s_ktime
#else
#define D2_KT E.ktime
#endif
#define D2_KSUB(KID, PHASE, WHO)                                                                                   \
  do {                                                                                                             \
    if (E.ktime && threadIdx.x == 0 && (WHO)) {                                                                    \
      const unsigned long long now_ = gcn_wall_clock();                                                            \
      unsigned long long *last_ = D2_KT + KT_SUB_LAST + ((KID) == 6 ? 1 : 0);                                      \
      if ((PHASE) > 0) D2_KT[KT_SUB + 8 * (KID) + (PHASE)] += now_ - *last_;                                        \
      *last_ = now_;                                                                                               \
    }                                                                                                              \
  } while (0)

// Which unique thread t of block b looks at in slice u of its group grp of a sweep: waves of 64 consecutive uniques are dealt
// round-robin over the blocks.  The input is sorted by abundance, and the abundant uniques are the expensive ones (long chains
// of stored comparisons, p-values below 1, bud candidates): dealt in contiguous runs they all landed in the first blocks, and in
// the persistent tail every other block waited for those at the barrier.
template <int BS, int U>
static __device__ __forceinline__ int sweep_unique(int grp, int u) {
  // (the thread index goes through an opaque move: inside the persistent kernel the indices - and every address derived from
  //  them - are invariants of the round loop, and hoisted out of it they were a hundred spilled registers)
  const int t = gcn_opaque_lane((int)threadIdx.x);
  const long long chunk = ((long long)(grp * U + u) * (BS / 64) + (t >> 6)) * gridDim.x + blockIdx.x;
  const long long r = chunk * 64 + (t & 63);
  return r > 0x7FFFFFF0ll ? 0x7FFFFFF0 : (int)r;
}

// ---- the persistent tail's LDS mirror (Eng2::mirror_on): one word per unique a block sweeps, at the place the sweeps visit it
// (slot = (grp * U + u) * BS + thread, which is mir_slot(r) of the unique r = sweep_unique(grp, u) whatever BS is).  nullptr in
// the launch chains and wherever the mirror is off: every reader falls back to the global arrays, which stay the truth.
constexpr uint32_t MIR_I1 = 1u << 31;      // T.i1[r] >= 0: the unique holds a second stored comparison
constexpr uint32_t MIR_LOCK = 1u << 30;    // P.lock[r]
constexpr uint32_t MIR_PNE1 = 1u << 29;    // P.p[r] != 1.0
constexpr uint32_t MIR_CL = 0xFFFFFFu;     // P.clust_of[r] (the host leaves the mirror off where partitions could number 2^24)
constexpr int MIR_CAP = 8192;              // uniques per block: two groups of 4096 (10^6 uniques on 123 blocks)
// A second word per unique sits MIR_CAP words behind the first: Store2::smask folded to 32 bits (bit k & 31 for a stored comparison
// with partition k) - what a filtered shuffle call asks of a unique that holds several (a coarser filter only lists more uniques
// for PASS B, which decides exactly).
// With the mirror a sweep is ONE pass over all the slots of the block instead of one per group of 4096 uniques (PASS A costs LDS
// reads, so nothing is gained by overlapping its loads group by group, and every group costs PASS B's round trips again); the
// work lists then hold 16-bit slots (| class << 13) instead of 32-bit unique indices, in the same LDS.
constexpr uint32_t MIR_INVALID = 0x1F000000u;   // the word of a slot behind the last unique (bits no valid word has)
// A lane's position in an LDS list it appends to, with ONE atomic per wave (every lane of the wave calls it, converged): a sweep
// appends hundreds of entries per block, and same-address LDS atomics are served one lane at a time
static __device__ __forceinline__ int wave_push(int *counter, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return 0;
  const int lane = (int)(threadIdx.x & 63u);
  int base = 0;
  if (lane == 0) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, 0, 64);
  return base + __popcll(m & ((1ull << lane) - 1ull));
}
static __device__ __forceinline__ int mir_slot(int r) { return ((r >> 6) / (int)gridDim.x) * 64 + (r & 63); }
static __device__ __forceinline__ int mir_unique(int slot) { return ((((slot >> 6) * (int)gridDim.x) + (int)blockIdx.x) << 6) | (slot & 63); }
static __device__ __forceinline__ uint32_t mir_fold(unsigned long long m) { return (uint32_t)m | (uint32_t)(m >> 32); }
static __device__ __forceinline__ void mir_upd(uint32_t *mir, int slot, uint32_t clear, uint32_t set) { mir[slot] = (mir[slot] & ~clear) | set; }
static __device__ __forceinline__ void mir_set(uint32_t *mir, int r, uint32_t clear, uint32_t set) {
  if (mir) mir_upd(mir, mir_slot(r), clear, set);
}

// ---- chain bookkeeping: which shuffle launches of the chain ran, and whether the evaluation after them stands -------
struct Chain2 { int nexec; bool eval_ok; };
static __device__ __forceinline__ Chain2 chain_state(const Ctl2 *ctl, const Round2Out *out, int nlev, int max_shuffle) {
  int nexec = 0;
  for (int j = 0; j < nlev; j++) {
    if (ctl->nsh_base + j >= max_shuffle) break;          // Rmain.cpp:321: at most MAX_SHUFFLE calls per round
    if (j > 0 && out->cnt[j - 1] == 0) break;             // the previous call moved nothing: the loop has ended
    nexec = j + 1;
  }
  bool ok = nlev == 0;                                     // (the chain after round 0 has no shuffle)
  if (nexec > 0 && out->cnt[nexec - 1] == 0) ok = true;
  if (nlev > 0 && ctl->nsh_base + nexec >= max_shuffle) ok = true;
  return Chain2{nexec, ok};
}
// partition reads as of the start of shuffle `nlv` of the chain (= after its first nlv calls)
static __device__ __forceinline__ uint32_t reads_at(const Eng2 &E, int i, int nlv) {
  uint32_t v = E.P.creads[i];
  for (int l = 0; l < nlv; l++) v += (uint32_t)E.dlt[(size_t)l * E.ccap + i];
  return v;
}

#ifndef D2_TAIL_TU
// ---- round 0: every unique keeps its comparison with the first centre -----------------------------------------------
__global__ __launch_bounds__(256) void k2_store0(Eng2 E, const double *__restrict__ lam, const uint32_t *__restrict__ ham,
                                                 const uint8_t *__restrict__ cls, const int32_t *__restrict__ round_counters) {
  const PartState &P = E.P;
  const SampleDev &S = E.S;
  const int centre = E.ctl->centre;
  const uint32_t creads = S.reads[centre];
  if (blockIdx.x == 0 && threadIdx.x < 2) {
    Round2Out *out = E.dblk + (E.ctl->pub_seq % RING2);
    atomicAdd(&out->stat[threadIdx.x], (unsigned long long)round_counters[threadIdx.x]);
    if (threadIdx.x == 0) out->pad0[3] = round_counters[0];          // (round 0's NW pairs: not pairs the rounds' batch compares ran, dada2hip_stats::nnw_rounds)
  }
  for (int r = blockIdx.x * 256 + threadIdx.x; r < S.N; r += gridDim.x * 256) {
    const uint8_t cl = cls[r];
    double l = 0.0;
    uint32_t h = 0xFFFFFFFFu;                                          // NULL sub (cluster.cpp:139-143)
    if (cl == CLS_GAPLESS || cl == CLS_NW) { l = lam[r]; h = ham[r]; }
    if (!(l >= 0.0 && l <= 1.0)) atomicOr(P.err_flag, 1);              // "Lambda out-of-range error." (cluster.cpp:184)
    const double em = P.E_minmax[r];
    if (l * E.total_reads > em) {                                      // always: E_minmax starts at -999
      if (l * creads > em) P.E_minmax[r] = l * creads;
      P.comp_i[r] = 0; P.comp_lam[r] = l; P.comp_ham[r] = h;           // i == 0: Raw::comp is refreshed (cluster.cpp:197)
    }
    E.T.lam0[r] = l; E.T.ham0[r] = h;
    E.T.smask[r] = 1ull;                                               // (one stored comparison: partition 0's)
    E.T.i1[r] = -1;
    E.T.head[r] = -1;
  }
}

#endif  // D2_TAIL_TU
// ---- the evaluation's pieces (b_p_update + b_bud's first stage; their sweep is pupdate_body below) ------------------------
// Besides the block minima, the candidates whose p-value is significant (within a factor 2 of the thresholds) are listed:
// k2_birth looks for the ties / near ties of the best key and for the likely next centres among those few, not among all
// uniques.
constexpr int PUPD_TAB = 1024;    // partitions whose per-partition facts k2_pupdate keeps in LDS
constexpr int SIG_CAP = 1024;
template <int BS>
struct PupdLds {
  static constexpr int U = BS >= 512 ? 4096 / BS : 2;                        // uniques per thread per group of the sweep
  int s_nwork;
  int32_t s_work[U * BS];                                                // the group's uniques whose p-value / candidacy has to be looked at
  BudKey s_k[2][BS / 64];
  int32_t s_sig[SIG_CAP];
  int s_nsig, s_sbase;
  int s_nlock;                                                           // locks an evaluation ATTEMPT has decided so far (Eng2::spec_lock_buf)
  // what a unique needs from ITS PARTITION (reads, update / lock flags, the centre and its reads) sits in LDS: the loads
  // behind clust_of[r] were a chain of three global round trips per unique in a latency-bound kernel
  uint32_t s_prd[PUPD_TAB], s_cread[PUPD_TAB];
  int32_t s_cen[PUPD_TAB];
  uint8_t s_upd[PUPD_TAB], s_chk[PUPD_TAB];
};
// b_p_update + the first stage of b_bud's arg-min by a grid of BS-thread blocks, after `nexec` shuffle calls of the round: the
// body of k2_pupdate (BS = 256) and of the evaluation phase of the persistent tail (BS = 1024).  partial[2 b], [2 b + 1]: block
// b's best keys.
// The pieces of the evaluation, so that the persistent tail can also run them INSIDE a shuffle call (shuffle_body<.., SPEC>):
//   pupd_tables   the per-partition facts in LDS (reads after the round's first `nexec` shuffle calls); the caller synchronises
//   pupd_wanted   PASS A's test of one unique
//   pupd_pass_b   PASS B over the work list of a group
//   pupd_finish   the block's minima and its listed candidates leave the block
template <int BS>
static __device__ __forceinline__ int pupd_tables(const Eng2 &E, PupdLds<BS> &L, int nexec) {
  const PartState &P = E.P;
  const int ntab = min(E.ctl->nclust, PUPD_TAB);
  for (int k = threadIdx.x; k < ntab; k += BS) {
    const int c = P.centre_of[k];
    L.s_prd[k] = reads_at(E, k, nexec);
    L.s_cen[k] = c;
    L.s_cread[k] = E.S.reads[c];
    L.s_upd[k] = P.update_e[k];
    L.s_chk[k] = P.check_locks[k];
  }
  if (threadIdx.x == 0) { L.s_nsig = 0; L.s_nlock = 0; }
  return ntab;
}
// a unique whose partition has not changed and whose p is exactly 1 (nine in ten of a large sample: singletons, pval.cpp:69) can
// neither be re-evaluated nor be a bud candidate - init is (p = 1, reads of the most abundant unique), and 1 is never below a
// threshold (omegaA < 1 <= N / 2; omegaP <= 1 / 2 is checked)
static __device__ __forceinline__ bool pupd_p1_skip(const Eng2 &E) { return 2.0 * E.bp.omegaP <= 1.0 && 2.0 * E.bp.omegaA <= (double)E.S.N; }
template <int BS>
static __device__ __forceinline__ bool pupd_wanted(const Eng2 &E, const PupdLds<BS> &L, int ntab, bool p1_skip, int cl, double p) {
  const bool intab = cl < ntab;
  const bool touched = (intab ? L.s_upd[cl] : E.P.update_e[cl]) || (E.greedy && (intab ? L.s_chk[cl] : E.P.check_locks[cl]));
  return touched || !(p1_skip && p == 1.0);
}
// DEFER (an evaluation attempt riding on a shuffle call, shuffle_body<.., SPEC>): the locks it decides go to the block's list in
// Eng2::spec_lock_buf instead of PartState::lock - k3_tail writes them out once the attempt is known to stand (spec_locks_flush)
template <int BS, bool DEFER = false>
static __device__ __forceinline__ void pupd_pass_b(const Eng2 &E, PupdLds<BS> &L, int nexec, int ntab, int nwork, BudKey &b0, BudKey &b1, uint32_t *mir = nullptr) {
  const PartState &P = E.P;
  const SampleDev &S = E.S;
  const int32_t *s_work = L.s_work;
  for (int w = threadIdx.x; w < nwork; w += BS) {
    const int slot = mir ? (int)((const uint16_t *)s_work)[w] : 0;
    const int r = mir ? mir_unique(slot) : s_work[w];
    const int cl = P.clust_of[r];
    const double l = P.comp_lam[r];
    const uint32_t reads = S.reads[r];
    const uint32_t ham = P.comp_ham[r];
    const bool pr = S.prior[r] != 0;
    const bool s0 = P.slot0[r] != 0;
    double p = P.p[r];
    const bool intab = cl < ntab;
    const uint32_t prd = intab ? L.s_prd[cl] : reads_at(E, cl, nexec);
    if (intab ? L.s_upd[cl] : P.update_e[cl]) {
      p = dev_get_pA(reads, pr, E.detect_singletons != 0, l, ham, prd);
      P.p[r] = p;
      if (mir) mir_upd(mir, slot, MIR_PNE1, p != 1.0 ? MIR_PNE1 : 0u);
    }
    if (E.greedy && (intab ? L.s_chk[cl] : P.check_locks[cl])) {          // pval.cpp:29-36
      const int c = intab ? L.s_cen[cl] : P.centre_of[cl];
      const uint32_t cr = intab ? L.s_cread[cl] : S.reads[c];
      if ((cr * l > reads) || r == c) {
        if (DEFER) {
          const int k = atomicAdd(&L.s_nlock, 1);                          // (a block lists a unique at most once per attempt: k < stride)
          if (k < E.spec_lock_stride) E.spec_lock_buf[(size_t)blockIdx.x * E.spec_lock_stride + k] = r;
        } else { P.lock[r] = 1; if (mir) mir_upd(mir, slot, 0u, MIR_LOCK); }
      }
    }
    // bud_candidate2 with the values at hand
    if (s0) continue;                                                    // r = 0 is skipped as "the centre" (cluster.cpp:285)
    if (reads < (uint32_t)E.bp.min_abund) continue;
    if ((int)ham < E.bp.min_hamming) continue;
    if (!(E.bp.min_fold <= 1 || ((double)reads) >= E.bp.min_fold * l * prd)) continue;
    if (bud_better(p, reads, b0)) { b0.p = p; b0.reads = reads; }
    if (pr && bud_better(p, reads, b1)) { b1.p = p; b1.reads = reads; }
    if (p * S.N < 2.0 * E.bp.omegaA || (pr && p < 2.0 * E.bp.omegaP)) {
      const int q = atomicAdd(&L.s_nsig, 1);
      if (q < SIG_CAP) L.s_sig[q] = r; else E.sig_list[atomicAdd(E.sig_n, 1)] = r;
    }
  }
}
// The attempt stood (the barrier behind it saw that its shuffle call moved nothing, and the round's decision has been taken): the
// locks it decided become visible now.  A new centre the decision has just made is skipped: the birth unlocked it
// (bi_assign_center, cluster.cpp:377; apply_birth_and_plan) BEFORE this runs, and a lock of its own must not land behind that.
template <int BS>
static __device__ __forceinline__ void spec_locks_flush(const Eng2 &E, const PupdLds<BS> &L, const Round2Out *out, uint32_t *mir = nullptr) {
  const int n = min(L.s_nlock, E.spec_lock_stride);
  const int skip = out->birth_applied ? E.ctl->centre : -1;
  const int32_t *buf = E.spec_lock_buf + (size_t)blockIdx.x * E.spec_lock_stride;
  if (n == 0) return;                                                     // (uniform: an LDS word)
  for (int k = threadIdx.x; k < n; k += BS) {
    const int r = buf[k];
    if (r != skip) { E.P.lock[r] = 1; mir_set(mir, r, 0u, MIR_LOCK); }
  }
  // the coming round's commit reads lock[] of these very uniques, in other waves of this block: the stores have left the wave
  // before any of them goes on
  gcn_drain_stores();
  __syncthreads();
}
template <int BS>
static __device__ __forceinline__ void pupd_finish(const Eng2 &E, PupdLds<BS> &L, BudKey b0, BudKey b1, BudKey *__restrict__ partial) {
  BudKey (&s_k)[2][BS / 64] = L.s_k;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    BudKey t;
    t.p = __shfl_xor(b0.p, o, 64); t.reads = __shfl_xor(b0.reads, o, 64);
    if (bud_better(t.p, t.reads, b0)) b0 = t;
    t.p = __shfl_xor(b1.p, o, 64); t.reads = __shfl_xor(b1.reads, o, 64);
    if (bud_better(t.p, t.reads, b1)) b1 = t;
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_k[0][w] = b0; s_k[1][w] = b1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BS / 64; k++) {
      if (bud_better(s_k[0][k].p, s_k[0][k].reads, b0)) b0 = s_k[0][k];
      if (bud_better(s_k[1][k].p, s_k[1][k].reads, b1)) b1 = s_k[1][k];
    }
    partial[2 * blockIdx.x] = b0;
    partial[2 * blockIdx.x + 1] = b1;
    const int n = min(L.s_nsig, SIG_CAP);
    L.s_sbase = n ? atomicAdd(E.sig_n, n) : 0;                         // one global atomic per block
  }
  __syncthreads();
  for (int i = threadIdx.x; i < min(L.s_nsig, SIG_CAP); i += BS) E.sig_list[L.s_sbase + i] = L.s_sig[i];
}


// ---- store filter of a round + b_shuffle2 ------------------------------------------------------------------------------
// STORE (first shuffle of a round): the commit of the round's cached comparisons - class from the batch's class word with
// the greedy skip (cluster.cpp:127-130) applied with the lock state of the commit, lambda / hamming from the rows the
// batch's aligner launch filled - and the store filter of cluster.cpp:179-201 on them.
// Most uniques of a large sample never get a second stored comparison: for them the arg-max is partition 0 whatever the
// reads are, and the pass touches 8 bytes of their state.
// halted, or a chain without the batch compare its round needs (Eng2::has_compare): nothing to do
static __device__ __forceinline__ bool v2_idle(const Eng2 &E) {
  const Ctl2 *ctl = E.ctl;
  return ctl->state != 0 || (!E.has_compare && ctl->need_compare != 0);
}
// LDS of one block of the shuffle pass (BS threads): movers and new store blocks are buffered per block and written out once
// at the end - ONE device atomic per block for each of the two counters (thousands of same-address atomics per launch were
// most of this kernel's time)
template <int BS>
struct ShufLds {
  static constexpr int MOVCAP = BS, NEWCAP = BS / 2;
  static constexpr int U = BS >= 512 ? 4096 / BS : 2;                        // uniques per thread per group of the sweep (4096 per block in the persistent tail)
  int s_n, s_base, s_an, s_abase, s_keep, s_anyinc, s_nwork;
  unsigned long long s_incmask;                                          // bit (k & 63) of every partition k whose reads rose in the previous call
  // (16-byte aligned: k3_tail lends the 32 KB from here on - every array behind it is rewritten at the start of a call - to the
  //  serial end of a round, birth_body<EXT_CNT>)
  alignas(16) int32_t s_work[U * BS];                                    // the group's uniques that have work to do: index | class << 30
  int32_t s_mov[3 * MOVCAP];
  int32_t s_newr[NEWCAP], s_newhead[NEWCAP];
  uint32_t s_newh[NEWCAP];
  double s_newl[NEWCAP];
  int32_t s_delta[DELTA_TAB];
  uint32_t s_reads[DELTA_TAB];                                           // partition reads as of the start of this call
  int8_t s_sgn[DELTA_TAB];
  uint32_t s_st[BS / 64][4];
};
// One b_shuffle2 call (STORE: preceded by the commit of the round's comparisons) by a grid of BS-thread blocks: the body of
// k2_shuffle (a launch of its own, BS = 256) and of the shuffle phases of the persistent tail k3_tail (BS = 1024).
// out: the round's result block; mv: where the call's full mover list goes; moved_before: movers of the round's earlier calls.
//
// SPEC (the persistent tail, calls after the commit's): the call also runs the round's EVALUATION (b_p_update + the block minima
// of b_bud, the pieces above) on the assumption that it moves nothing - which is how every round's shuffle loop ends
// (Rmain.cpp:320-325), so the last call of a round IS its evaluation: one sweep over the uniques and one grid barrier less per
// round.  The two PASS A loads of the evaluation travel with the shuffle's own; a block evaluates only while it has moved nothing
// itself.  If any block moved something the attempt is void and costs nothing but its time: the p-values it wrote belong to
// members of touched partitions, which the next attempt rewrites (a touched partition stays touched until the round's
// evaluation stands, and a unique that moves lands in a touched partition); the candidate list is emptied by the barrier's last
// arriver; the block minima are overwritten; and the locks it decided were never published: they wait in the block's list
// (Eng2::spec_lock_buf) for the barrier that makes the attempt stand - the prefetch compare of the second stream reads lock[] at
// any time and relies on locks only growing between a compare and its commit.
template <bool STORE, int BS, bool SPEC = false>
static __device__ D2_SHUF_INLINE void shuffle_body(const Eng2 &E, ShufLds<BS> &L, int level, int moved_before, int32_t *mv, Round2Out *out,
                                                   PupdLds<BS> *LP = nullptr, BudKey init = BudKey{1.0, 0u}, BudKey *partial = nullptr, uint32_t *mir = nullptr) {
  static_assert(!(SPEC && STORE), "the speculative evaluation rides on the calls after the commit's");
  const Ctl2 *ctl = E.ctl;
  constexpr int MOVCAP = ShufLds<BS>::MOVCAP, NEWCAP = ShufLds<BS>::NEWCAP;
  int &s_n = L.s_n, &s_base = L.s_base, &s_an = L.s_an, &s_abase = L.s_abase, &s_keep = L.s_keep, &s_anyinc = L.s_anyinc;
  int32_t *s_mov = L.s_mov, *s_newr = L.s_newr, *s_newhead = L.s_newhead, *s_delta = L.s_delta;
  uint32_t *s_newh = L.s_newh, *s_reads = L.s_reads;
  double *s_newl = L.s_newl;
  int8_t *s_sgn = L.s_sgn;
  D2_TRACE_SH(level, 0);
  if (threadIdx.x == 0) { s_n = 0; s_an = 0; s_keep = 0; }
  const PartState &P = E.P;
  const SampleDev &S = E.S;
  const Store2 &T = E.T;
  const int N = S.N;
  const int nclust = ctl->nclust, ci = nclust - 1, centre = ctl->centre;
  const int ntab = nclust < DELTA_TAB ? nclust : DELTA_TAB;
  // A call after the chain's first one only has to look at the uniques the PREVIOUS call can have unsettled.  After a call
  // every unique that is not a centre sits in the arg-max of lambda * reads over its stored comparisons (it was moved there,
  // or it was there already), so it can only want to move now if the reads of its own partition went DOWN in that call or
  // the reads of another partition it holds a comparison with went UP (ties go to the lowest partition before and after:
  // a falling rival or a rising home cannot change the winner).  s_sgn[k] = sign of partition k's net reads delta of the
  // previous call; everybody else leaves after reading 12 bytes.
  const bool filt = !STORE && level >= 1 && E.sh_filter;
  const int32_t *dlp = E.dlt + (size_t)(level >= 1 ? level - 1 : 0) * E.ccap;
  if (filt && threadIdx.x == 0) { s_anyinc = 0; L.s_incmask = nclust > ntab ? ~0ull : 0ull; }
  int ptab = 0;
  if (SPEC) ptab = pupd_tables<BS>(E, *LP, level);                       // (reads after `level` calls = after this one, if it moves nothing)
  const bool p1_skip = SPEC && pupd_p1_skip(E);
  BudKey eb0 = init, eb1 = init;
  __syncthreads();
  for (int k = threadIdx.x; k < ntab; k += BS) {
    s_delta[k] = 0; s_reads[k] = reads_at(E, k, level);
    if (filt) {
      const int32_t d = dlp[k];
      s_sgn[k] = d < 0 ? -1 : (d > 0 ? 1 : 0);
      if (d > 0) { s_anyinc = 1; atomicOr(&L.s_incmask, 1ull << (k & 63)); }
    }
  }
  __syncthreads();
  const bool anyinc = filt ? (s_anyinc != 0 || nclust > ntab) : true;
  const unsigned long long incmask = filt ? (mir ? (unsigned long long)mir_fold(L.s_incmask) : L.s_incmask) : ~0ull;   // (the mirror's masks are folded to 32 bits)
  // the commit of a round: who can move in the round's FIRST call?  Whoever stores a comparison with the new centre now; and,
  // if the previous round ended stable, only the members of the partition the birth took the centre from (its reads fell) -
  // nobody else's home lost reads, and the only partition that gained is the new one
  const bool stable = STORE && ctl->stable != 0;
  const int bfrom = ctl->bfrom;
  auto sgn_of = [&](int i) __attribute__((always_inline)) -> int { if (i < ntab) return s_sgn[i]; const int32_t d = dlp[i]; return d < 0 ? -1 : (d > 0 ? 1 : 0); };
  auto rd_at = [&](int i) __attribute__((always_inline)) -> uint32_t { return i < ntab ? s_reads[i] : reads_at(E, i, level); };
  int32_t *dl = E.dlt + (size_t)level * E.ccap;
  const uint32_t creads_c = S.reads[centre];
  const double *lam_row = E.C.lamB + (size_t)ctl->slot * E.C.Npad;       // the round's comparisons: its cache slot's rows
  const uint32_t *ham_row = E.C.hamB + (size_t)ctl->slot * E.C.Npad;
  // ... and their classes: the cached class word of the slot's batch, with the greedy skip of cluster.cpp:127-130 evaluated NOW
  // (lock state of the commit, not of the compare).  This was a launch of its own (k2_lists, 10 us per round).
  const uint16_t *cls_row = E.C.bcls + (size_t)(ctl->slot / KB_MAX) * E.C.Npad;
  const int kpos2 = 2 * (ctl->slot % KB_MAX);
  uint32_t st01 = 0, st23 = 0;                                           // this thread's NW | gapless and shrouded | skipped pairs (16 bits each)
  const uint32_t reads_ci = STORE ? rd_at(ci) : 0u;
  const uint32_t reads_0 = rd_at(0);
  int my_keep = 0;                                                       // comparisons this thread stored (STORE)
  int my_n0 = 0;                                                         // members partition 0 lost (low half) / gained (high half)
  D2_TRACE_SH(level, 1);
  // The sweep in two passes per U * BS uniques of the block.  PASS A, a few instructions per unique with the U uniques of a
  // thread requested together: does the unique hold a second stored comparison (only such a unique can ever move), and - the
  // commit of the round - the class of its comparison with the new centre after the greedy skip, counted.  The rest (most
  // uniques of a large sample: one stored comparison, shrouded or skipped now) is done with after 4 / 11 bytes.  The others go
  // to a work list in LDS.  PASS B takes the list one unique per thread: everything the decision needs in ONE round trip, then
  // the store filter, the arg-max, the move.  The long code exists once and runs on full waves (round 3 ran it in whatever
  // lanes happened to need it, one unique after the other, with a chain of two to three round trips each: 14-20 us per sweep
  // of 10^6 uniques for 12-50 MB of traffic).
  constexpr int U = ShufLds<BS>::U;
  int32_t *s_work = L.s_work;
  uint16_t *s_work16 = (uint16_t *)L.s_work;                             // (under the mirror: slot | class << 13)
  int &s_nwork = L.s_nwork;
  const int ngrp = (int)(((long long)N + (long long)U * BS * gridDim.x - 1) / ((long long)U * BS * gridDim.x));
  const int gspan = mir ? ngrp : 1;                                      // groups per pass (the mirror: all of them at once)
  for (int grp0 = 0; grp0 < ngrp; grp0 += gspan) {
    if (threadIdx.x == 0) { s_nwork = 0; if (SPEC) LP->s_nwork = 0; }
    __syncthreads();
    for (int grp = grp0; grp < grp0 + gspan; grp++) {
      int i1s[U], froms[U];
      uint32_t clw[U], rds[U];
      uint8_t lks[U], oks[U];
      unsigned long long sms[U];
      double pps[U];
      const bool need_r = STORE || !mir;                                 // (a sweep over the mirror that reads nothing else needs no index)
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = need_r ? sweep_unique<BS, U>(grp, u) : 0;
        i1s[u] = -1; froms[u] = 0; clw[u] = 0; rds[u] = 0; lks[u] = 0; sms[u] = 0; pps[u] = 1.0; oks[u] = 0;
        if (mir) {
          // (the mirror: what is left for memory is the round's class word and the reads at the commit)
          const uint32_t mw = mir[(grp * U + u) * BS + (int)threadIdx.x];
          if (!(mw & MIR_INVALID)) {
            oks[u] = 1;
            i1s[u] = (mw & MIR_I1) ? 0 : -1;                            // (PASS A only asks whether there is one)
            froms[u] = (int)(mw & MIR_CL);
            if (STORE) { clw[u] = cls_row[r]; rds[u] = S.reads[r]; lks[u] = (mw & MIR_LOCK) ? 1 : 0; }
            if (filt && (mw & MIR_I1)) sms[u] = (unsigned long long)mir[MIR_CAP + (grp * U + u) * BS + (int)threadIdx.x];
            if (SPEC) pps[u] = (mw & MIR_PNE1) ? 0.0 : 1.0;             // (... and whether p is 1)
          }
        } else if (r < N) {
          oks[u] = 1;
          i1s[u] = T.i1[r];
          if (STORE || filt || SPEC) froms[u] = P.clust_of[r];
          if (STORE) { clw[u] = cls_row[r]; rds[u] = S.reads[r]; lks[u] = P.lock[r]; }
          if (filt) sms[u] = T.smask[r];
          if (SPEC) pps[u] = P.p[r];
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = need_r ? sweep_unique<BS, U>(grp, u) : 0;
        const bool ok = oks[u] != 0;
        uint32_t cl = 0;
        if (STORE && ok) {
          cl = (clw[u] >> kpos2) & 3u;
          const bool skip = E.greedy && (rds[u] > creads_c || lks[u] != 0);
          if (skip) cl = CLS_SKIP;
          else if (cl == CLS_SKIP) atomicOr(P.err_flag, 8);            // the cache lacks a comparison the round needs
          if (cl >= CLS_GAPLESS) st01 += cl == CLS_NW ? 1u : 0x10000u; else st23 += cl == CLS_SHROUD ? 1u : 0x10000u;
        }
        // a unique with ONE stored comparison (partition 0's) that gets no second one now sits in partition 0 or is a centre;
        // one with several can only want to move if its home lost reads or a partition it holds a comparison with gained
        bool work = cl >= CLS_GAPLESS;
        if (i1s[u] >= 0) {
          if (STORE) work = work || !stable || froms[u] == bfrom;
          else if (filt) work = sgn_of(froms[u]) < 0 || (anyinc && (sms[u] & incmask) != 0ull);
          else work = true;
        }
        work = work && ok;
        const int slot = (grp * U + u) * BS + (int)threadIdx.x;
        {
          const int k = wave_push(&s_nwork, work);
          if (work) { if (mir) s_work16[k] = (uint16_t)((uint32_t)slot | (cl << 13)); else s_work[k] = (int32_t)((uint32_t)r | (cl << 30)); }
        }
        if (SPEC) {
          const bool want = ok && pupd_wanted<BS>(E, *LP, ptab, p1_skip, froms[u], pps[u]);
          const int k = wave_push(&LP->s_nwork, want);
          if (want) { if (mir) ((uint16_t *)LP->s_work)[k] = (uint16_t)slot; else LP->s_work[k] = r; }
        }
      }
    }
    __syncthreads();
    D2_TRACE_SH(level, 5);
    const int nwork = s_nwork;
    for (int w = threadIdx.x; w < nwork; w += BS) {
      const uint32_t item = mir ? (uint32_t)s_work16[w] : (uint32_t)s_work[w];
      const int slot = (int)(item & 0x1FFFu);                            // (the mirror's lists only)
      const int r = mir ? mir_unique(slot) : (int)(item & 0x3FFFFFFFu);
      const uint32_t cl = mir ? item >> 13 : item >> 30;
      // one round trip: everything the decision can need (a unique on this list is one in five to ten)
      const int i1 = T.i1[r];
      const int from = P.clust_of[r];
      const int head_raw = T.head[r];
      const double lam0_r = T.lam0[r], lam1_r = T.lam1[r];
      const double l_raw = (STORE && cl >= CLS_GAPLESS) ? lam_row[r] : 0.0, em = (STORE && cl >= CLS_GAPLESS) ? P.E_minmax[r] : 0.0;
      const uint32_t h_raw = (STORE && cl >= CLS_GAPLESS) ? ham_row[r] : 0u;
      bool keep = false, need_new = false, move = false;
      double l = 0.0, best_l = 0.0;
      uint32_t h = 0, best_h = 0;
      int head = -1, hcnt = 3, apos = 0, pos = 0, to = 0;
      {
      head = i1 >= 0 ? head_raw : -1;                                    // (a chain only exists behind a used second entry)
      if (STORE) {
        if (cl >= CLS_GAPLESS) {
          l = l_raw; h = h_raw;
          if (!(l >= 0.0 && l <= 1.0)) atomicOr(P.err_flag, 1);        // "Lambda out-of-range error." (cluster.cpp:184)
          keep = l * E.total_reads > em;                               // this partition could attract this unique
          if (keep) {
            my_keep++;
            if (l * creads_c > em) P.E_minmax[r] = l * creads_c;
            if (r == centre) { P.comp_i[r] = ci; P.comp_lam[r] = l; P.comp_ham[r] = h; }
            T.smask[r] |= 1ull << (ci & 63);
            if (mir) mir[MIR_CAP + slot] |= 1u << (ci & 31);
          }
        }
      }
      // (the commit of a stable round: a unique that stores nothing now and whose home did not lose reads stays where it is)
      const bool reeval = !STORE || keep || !stable || from == bfrom;
      // arg-max of lambda * reads over the stored comparisons; ties go to the lowest partition (cluster.cpp:229-239)
      const CompBlk *best_cb = nullptr;
      int best_k = 0;
      int best_i = from, best_src = 0;                                   // 0: round-0 entry, 1: second entry, 2: chain block, 3: this round's
      if (reeval && (i1 >= 0 || keep)) {
        best_i = 0;
        best_l = lam0_r;
        double best_e = best_l * reads_0;
        if (i1 >= 0) {
          const double nl = lam1_r, e = nl * rd_at(i1);
          if (e > best_e || (e == best_e && i1 < best_i)) { best_e = e; best_i = i1; best_l = nl; best_src = 1; }
        }
        for (int b = head, first = 1, hops = 0; b >= 0 && hops < (1 << 22); first = 0, hops++) {   // (bounded: never spin on a bad link)
          const CompBlk *cb = T.blk + b;
          const int cnt = cb->cnt;
          if (first) hcnt = cnt;
#pragma unroll
          for (int k = 0; k < 3; k++)
            if (k < cnt) {
              const int i = cb->i[k];
              const double nl = cb->lam[k], e = nl * rd_at(i);
              if (e > best_e || (e == best_e && i < best_i)) { best_e = e; best_i = i; best_l = nl; best_cb = cb; best_k = k; best_src = 2; }
            }
          b = cb->next;
        }
        if (keep) {
          if (i1 < 0) { T.i1[r] = ci; T.lam1[r] = l; T.ham1[r] = h; if (mir) mir[slot] |= MIR_I1; }  // the unique's second stored comparison: inline
          else {
            need_new = head < 0 || hcnt >= 3;
            if (!need_new) {                                           // room in the newest block: append in place
              CompBlk *cb = T.blk + head;
              cb->i[hcnt] = ci; cb->ham[hcnt] = h; cb->lam[hcnt] = l; cb->cnt = hcnt + 1;
            }
          }
          const double e = l * reads_ci;
          if (e > best_e) { best_e = e; best_i = ci; best_l = l; best_src = 3; }   // (ci is the highest index: only strictly)
        }
      }
      if (best_i != from && r != P.centre_of[from]) {
        move = true;
        to = best_i;
        if (best_src == 0) { best_l = lam0_r; best_h = T.ham0[r]; }
        else if (best_src == 1) best_h = T.ham1[r];
        else if (best_src == 2) best_h = best_cb->ham[best_k];
        else best_h = h;
        P.clust_of[r] = to;
        if (mir) mir_upd(mir, slot, MIR_CL, (uint32_t)to);
        P.comp_i[r] = to; P.comp_lam[r] = best_l; P.comp_ham[r] = best_h;
        E.moved[r] = 1;
        my_n0 += (from == 0 ? 1 : 0) + (to == 0 ? 0x10000 : 0);
        const uint32_t rd = S.reads[r];
        if (to < ntab) atomicAdd(&s_delta[to], (int32_t)rd); else atomicAdd(&dl[to], (int32_t)rd);
        if (from < ntab) atomicSub(&s_delta[from], (int32_t)rd); else atomicSub(&dl[from], (int32_t)rd);
        P.update_e[to] = 1; P.update_e[from] = 1;
      }
      }
      if (need_new) {
        apos = atomicAdd(&s_an, 1);
        if (apos < NEWCAP) { s_newr[apos] = r; s_newhead[apos] = head; s_newh[apos] = h; s_newl[apos] = l; }
        else {   // (more new blocks in one thread block than the buffer holds: straight to the device counter)
          const int nb = atomicAdd(T.blk_count, 1);
          if (nb < T.blk_cap) {
            CompBlk *cb = T.blk + nb;
            cb->next = head; cb->cnt = 1; cb->i[0] = ci; cb->ham[0] = h; cb->lam[0] = l;
            T.head[r] = nb;
          } else atomicOr(P.err_flag, 2);
        }
      }
      if (move) {
        pos = atomicAdd(&s_n, 1);
        if (pos < MOVCAP) { s_mov[3 * pos] = r; s_mov[3 * pos + 1] = from; s_mov[3 * pos + 2] = to; }
        else {
          const int k = atomicAdd(&out->cnt[level], 1);
          int32_t *m = mv + 3 * (size_t)k;
          m[0] = r; m[1] = from; m[2] = to;
          const int ki = moved_before + k;
          if (ki < E.mov_inline) { out->mov[3 * ki] = r; out->mov[3 * ki + 1] = from; out->mov[3 * ki + 2] = to; }
        }
      }
    }
    __syncthreads();                                                     // (the list is rewritten by the next group)
    D2_TRACE_SH(level, 6);
    if (SPEC) {
      if (s_n == 0) pupd_pass_b<BS, true>(E, *LP, level, ptab, LP->s_nwork, eb0, eb1, mir);   // (s_n: uniform, the block is behind a barrier)
      __syncthreads();
      D2_TRACE_SH(level, 7);
    }
  }
  __syncthreads();                                                       // the block's movers / new blocks are all buffered
  if (SPEC && s_n == 0) pupd_finish<BS>(E, *LP, eb0, eb1, partial);
  D2_TRACE_SH(level, 2);
  const int nmov = min(s_n, MOVCAP), nnew = min(s_an, NEWCAP);
  if (threadIdx.x == 0) {
    s_base = nmov ? atomicAdd(&out->cnt[level], nmov) : 0;
    s_abase = nnew ? atomicAdd(T.blk_count, nnew) : 0;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < nnew; q += BS) {
    const int nb = s_abase + q;
    if (nb < T.blk_cap) {
      CompBlk *cb = T.blk + nb;
      cb->next = s_newhead[q]; cb->cnt = 1; cb->i[0] = ci; cb->ham[0] = s_newh[q]; cb->lam[0] = s_newl[q];
      T.head[s_newr[q]] = nb;
    } else atomicOr(P.err_flag, 2);
  }
  for (int q = threadIdx.x; q < nmov; q += BS) {
    const int k = s_base + q;
    int32_t *m = mv + 3 * (size_t)k;
    m[0] = s_mov[3 * q]; m[1] = s_mov[3 * q + 1]; m[2] = s_mov[3 * q + 2];
    const int ki = moved_before + k;
    if (ki < E.mov_inline) { out->mov[3 * ki] = s_mov[3 * q]; out->mov[3 * ki + 1] = s_mov[3 * q + 1]; out->mov[3 * ki + 2] = s_mov[3 * q + 2]; }
  }
  __syncthreads();                                                       // every delta of the block is in the table
  if (__any(my_n0 != 0)) {                                               // (a thread moves a handful of uniques at most: no carry)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_n0 += __shfl_xor(my_n0, o, 64);
    if ((threadIdx.x & 63) == 0) {
      if (my_n0 & 0xFFFF) atomicAdd(&E.n0d[2 * level], my_n0 & 0xFFFF);
      if (my_n0 >> 16) atomicAdd(&E.n0d[2 * level + 1], my_n0 >> 16);
    }
  }
  if (STORE) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_keep += __shfl_xor(my_keep, o, 64);
    if ((threadIdx.x & 63) == 0 && my_keep) atomicAdd(&s_keep, my_keep);
    __syncthreads();
    if (threadIdx.x == 0 && s_keep) atomicAdd(&out->pad0[0], s_keep);   // Comparisons kept this round (cluster.cpp:189-199)
    // the round's class statistics (the reference's nalign / nshroud counters): a thread sees a handful of uniques, a wave
    // at most 64 x that - no carry between the 16-bit halves.  Every block has some, so they leave as one plain 16-byte
    // store per block and k2_birth adds them up (an atomic per wave on the four counters of the result block tripled this
    // kernel's time: 32 000 atomics on one cache line)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { st01 += __shfl_xor(st01, o, 64); st23 += __shfl_xor(st23, o, 64); }
    if ((threadIdx.x & 63) == 0) {
      uint32_t *w = L.s_st[threadIdx.x >> 6];
      w[0] = st01 & 0xFFFFu; w[1] = st01 >> 16; w[2] = st23 & 0xFFFFu; w[3] = st23 >> 16;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      uint32_t v = 0;
#pragma unroll
      for (int w = 0; w < BS / 64; w++) v += L.s_st[w][threadIdx.x];
      E.stat_part[(size_t)blockIdx.x * 4 + threadIdx.x] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *E.stat_n = (int32_t)gridDim.x;
  }
  D2_TRACE_SH(level, 3);
  for (int k = threadIdx.x; k < ntab; k += BS) {
    const int32_t d = s_delta[k];
    if (d) atomicAdd(&dl[k], d);
  }
  D2_TRACE_SH(level, 4);
}

#ifndef D2_TAIL_TU
template <bool STORE>
__global__ __launch_bounds__(256, 8) GCN_SGPR_BUDGET(80) void k2_shuffle(Eng2 E, int level) {
  const Ctl2 *ctl = E.ctl;
  if (v2_idle(E)) return;
  Round2Out *out = E.dblk + (ctl->pub_seq % RING2);
  if (ctl->nsh_base + level >= E.max_shuffle) return;
  int moved_before = 0;
  for (int l = 0; l < level; l++) {
    const int c = out->cnt[l];
    if (c == 0) return;                                                // an earlier call moved nothing: the loop has ended
    moved_before += c;
  }
  __shared__ ShufLds<256> L;
  shuffle_body<STORE, 256>(E, L, level, moved_before, E.movers + ((size_t)((ctl->pub_seq % MOV_RING) * SH_CHAIN + level)) * 3 * (size_t)E.S.N, out);
}

constexpr int LISTS_PER_THREAD = 8;

// ---- work lists of a batch compare: for every batch position the uniques whose pair goes to the aligner (class NW) and the
// gapless ones, as of the screen (a pair the greedy rule skips at ITS commit is aligned in vain; locks only grow, so no pair
// a commit needs is ever missing).  Streaming shape: one thread takes EIGHT consecutive uniques with one 16-byte load of their
// class words; most class words have no such pair at all. ----
__global__ __launch_bounds__(256) void k2_batch_lists(Eng2 E) {
  const Ctl2 *ctl = E.ctl;
  const int nb = ctl->nalign;
  if (ctl->state != 0 || nb == 0) return;
  // (the retry lists of the aligner's pointer-free pass, which follows in the stream, start empty)
  if (blockIdx.x == 0 && threadIdx.x < KB_MAX && E.bretry_n) E.bretry_n[threadIdx.x] = 0;
  // ... and the pass is switched off for the rest of the run once it has handed more than a quarter of its pairs to the full
  // kernel (a sweep without pointers costs about 0.8 of one with: beyond that two sweeps lose to one)
  if (blockIdx.x == 0 && threadIdx.x == 0 && E.fast_ctl && E.fast_ctl[1] >= 2048ull && 4ull * E.fast_ctl[0] > E.fast_ctl[1]) E.fast_ctl[2] = 1ull;
  __shared__ int s_cnt[2 * KB_MAX], s_base[2 * KB_MAX];
  const SampleDev &S = E.S;
  const uint16_t *bcls = E.C.bcls + (size_t)ctl->abuf * E.C.Npad;
  // commit mode: only the position of the coming round's centre, and only the pairs the greedy rule (cluster.cpp:127-130) does
  // not skip NOW - the lock state of the commit, which follows in the same chain
  const bool commit = E.align_at_commit != 0;
  const int kpos = ctl->slot % KB_MAX;
  const uint32_t fmask = commit ? (2u << (2 * kpos)) : 0xAAAAu;        // bit 2k+1 of a class word: NW or gapless (CLS_GAPLESS = 2, CLS_NW = 3)
  const uint32_t creads_c = commit ? S.reads[ctl->centre] : 0u;
  if (threadIdx.x < 2 * KB_MAX) s_cnt[threadIdx.x] = 0;
  const int r0 = (blockIdx.x * 256 + threadIdx.x) * LISTS_PER_THREAD;
  uint4 cw = make_uint4(0, 0, 0, 0), rd0 = cw, rd1 = cw;
  uint2 lk = make_uint2(0, 0);
  if (r0 < S.N) {
    cw = *(const uint4 *)(bcls + r0);
    if (commit && E.greedy) {
      lk = *(const uint2 *)(E.P.lock + r0);
      rd0 = *(const uint4 *)(S.reads + r0);
      rd1 = *(const uint4 *)(S.reads + r0 + 4);
    }
  }
  __syncthreads();
  const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w}, rds[8] = {rd0.x, rd0.y, rd0.z, rd0.w, rd1.x, rd1.y, rd1.z, rd1.w};
  const bool any = ((cw.x | cw.y | cw.z | cw.w) & (fmask | (fmask << 16))) != 0;   // (positions >= the batch's size hold class 0)
  auto fields = [&](int q) __attribute__((always_inline)) -> uint32_t {   // class word of unique r0 + q, and which of its fields to list
    const uint32_t w = (cws[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu;
    uint32_t m = w & fmask;
    if (commit && E.greedy) {
      const bool locked = ((q < 4 ? lk.x >> (8 * q) : lk.y >> (8 * (q - 4))) & 0xFFu) != 0;
      if (rds[q] > creads_c || locked) m = 0;
    }
    return w | (m << 16);
  };
  uint32_t n_lo = 0, n_hi = 0;                                          // per-list counts of this thread, 4 bits each (<= 8)
  if (any) {
#pragma unroll
    for (int q = 0; q < LISTS_PER_THREAD; q++) {
      if (r0 + q >= S.N) break;
      const uint32_t wf = fields(q), w = wf & 0xFFFFu;
      for (uint32_t m = wf >> 16; m; m &= m - 1) {
        const int k = __builtin_ctz(m) >> 1;
        const int list = ((w >> (2 * k)) & 1u) ? k : KB_MAX + k;        // NW : gapless
        if (list < 8) n_lo += 1u << (4 * list); else n_hi += 1u << (4 * (list - 8));
      }
    }
  }
  int mybase[2 * KB_MAX];                                               // ... and where they start in the block's share of each list
#pragma unroll
  for (int l = 0; l < 2 * KB_MAX; l++) {
    const int n = (int)(((l < 8 ? n_lo : n_hi) >> (4 * (l & 7))) & 15u);
    mybase[l] = n ? atomicAdd(&s_cnt[l], n) : 0;
  }
  __syncthreads();
  if (threadIdx.x < 2 * KB_MAX) {
    const int n = s_cnt[threadIdx.x];
    s_base[threadIdx.x] = n ? atomicAdd(&E.blist_n[threadIdx.x], n) : 0;   // one device atomic per list per block
  }
  __syncthreads();
  if (any) {
#pragma unroll
    for (int q = 0; q < LISTS_PER_THREAD; q++) {
      if (r0 + q >= S.N) break;
      const uint32_t wf = fields(q), w = wf & 0xFFFFu;
      for (uint32_t m = wf >> 16; m; m &= m - 1) {
        const int k = __builtin_ctz(m) >> 1;
        const int list = ((w >> (2 * k)) & 1u) ? k : KB_MAX + k;
        int pos = 0;
#pragma unroll
        for (int l = 0; l < 2 * KB_MAX; l++) if (l == list) pos = s_base[l] + mybase[l]++;
        // (a list holds a unique at most once per compare: pos < N <= Npad - unless the same compare ran twice, which the entry of
        //  the persistent launch rules out since round 6; a write past the list's end would land in the next list)
        if ((size_t)pos < E.C.Npad) E.blist[(size_t)list * E.C.Npad + pos] = r0 + q; else atomicOr(E.P.err_flag, 32);
      }
    }
  }
}

#endif  // D2_TAIL_TU
// ---- b_p_update + first stage of b_bud (no "would another shuffle move" pass: the chain's shuffles are real calls) -----
static __device__ __forceinline__ bool bud_candidate2(const Eng2 &E, int r, int nlv) {
  const PartState &P = E.P;
  if (P.slot0[r]) return false;                                          // r = 0 is skipped as "the centre" (cluster.cpp:285)
  const uint32_t reads = E.S.reads[r];
  if (reads < (uint32_t)E.bp.min_abund) return false;
  if ((int)P.comp_ham[r] < E.bp.min_hamming) return false;
  if (!(E.bp.min_fold <= 1 || ((double)reads) >= E.bp.min_fold * P.comp_lam[r] * reads_at(E, P.clust_of[r], nlv))) return false;
  return true;
}

template <int BS>
static __device__ D2_PUPD_INLINE void pupdate_body(const Eng2 &E, PupdLds<BS> &L, int nexec, BudKey init, BudKey *__restrict__ partial, uint32_t *mir = nullptr) {
  const PartState &P = E.P;
  const SampleDev &S = E.S;
  D2_TRACE(5, 0);
  const int ntab = pupd_tables<BS>(E, L, nexec);
  __syncthreads();
  BudKey b0 = init, b1 = init;
  D2_TRACE(5, 1);
  // Two passes per U * BS uniques, as in the shuffle sweep.  PASS A reads a unique's partition and p-value: most uniques are
  // done with after 12 bytes (pupd_wanted).  The others go to a work list in LDS.
  // PASS B, one listed unique per thread: the reference's b_p_update / lock / candidate code, the special functions of the
  // p-value on full waves instead of in the odd lane.
  constexpr int U = PupdLds<BS>::U;
  const bool p1_skip = pupd_p1_skip(E);
  int32_t *s_work = L.s_work;
  int &s_nwork = L.s_nwork;
  const int ngrp = (int)(((long long)S.N + (long long)U * BS * gridDim.x - 1) / ((long long)U * BS * gridDim.x));
  const int gspan = mir ? ngrp : 1;                                      // (the mirror: one pass over all the block's slots, shuffle_body)
  for (int grp0 = 0; grp0 < ngrp; grp0 += gspan) {
    if (threadIdx.x == 0) s_nwork = 0;
    __syncthreads();
    for (int grp = grp0; grp < grp0 + gspan; grp++) {
      int cls_[U];
      double ps_[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        cls_[u] = -1; ps_[u] = 1.0;
        if (mir) {
          const uint32_t mw = mir[(grp * U + u) * BS + (int)threadIdx.x];
          if (!(mw & MIR_INVALID)) { cls_[u] = (int)(mw & MIR_CL); ps_[u] = (mw & MIR_PNE1) ? 0.0 : 1.0; }
        } else {
          const int r = sweep_unique<BS, U>(grp, u);
          if (r < S.N) { cls_[u] = P.clust_of[r]; ps_[u] = P.p[r]; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int cl = cls_[u];
        const bool want = cl >= 0 && pupd_wanted<BS>(E, L, ntab, p1_skip, cl, ps_[u]);
        const int k = wave_push(&s_nwork, want);
        if (want) {
          if (mir) ((uint16_t *)s_work)[k] = (uint16_t)((grp * U + u) * BS + (int)threadIdx.x); else s_work[k] = sweep_unique<BS, U>(grp, u);
        }
      }
    }
    __syncthreads();
    D2_TRACE(5, 4);
    pupd_pass_b<BS>(E, L, nexec, ntab, s_nwork, b0, b1, mir);
    __syncthreads();                                                     // (the list is rewritten by the next group)
    D2_TRACE(5, 5);
  }
  D2_TRACE(5, 2);
  pupd_finish<BS>(E, L, b0, b1, partial);
  D2_TRACE(5, 3);
}
#ifndef D2_TAIL_TU
__global__ __launch_bounds__(256) void k2_pupdate(Eng2 E, int nlev, BudKey init, BudKey *__restrict__ partial) {
  const Ctl2 *ctl = E.ctl;
  if (v2_idle(E)) return;
  Round2Out *out = E.dblk + (ctl->pub_seq % RING2);
  const Chain2 cs = chain_state(ctl, out, nlev, E.max_shuffle);
  if (!cs.eval_ok) return;
  __shared__ PupdLds<256> L;
  pupdate_body<256>(E, L, cs.nexec, init, partial);
}

#endif  // D2_TAIL_TU
// ---- the birth, the plan of the coming round's compare, and the publication of the round's result block -----------------
constexpr int NBUF_MAX = 64;      // batch buffers (NBUF_MAX x KB_MAX cached centres at most)
constexpr int PLAN_PER = 8;       // significant candidates each thread of k2_birth looks at when it predicts (8192 in all)
constexpr int PLAN_BITS = 65536;  // uniques below this index are looked up in a bitmap of the cached centres

// block-wide arg-min of (p, reads, r) keys over the 1024 threads; returns the winning unique (or -1) to every thread
static __device__ __forceinline__ int block_best(double p, uint32_t reads, int r, double *s_p, uint32_t *s_rd, int *s_r) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double p2 = __shfl_xor(p, o, 64);
    const uint32_t rd2 = __shfl_xor(reads, o, 64);
    const int r2 = __shfl_xor(r, o, 64);
    const bool take = r2 >= 0 && (r < 0 || p2 < p || (p2 == p && (rd2 > reads || (rd2 == reads && r2 < r))));
    if (take) { p = p2; reads = rd2; r = r2; }
  }
  const int w = threadIdx.x >> 6, nwv = (int)blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_p[w] = p; s_rd[w] = reads; s_r[w] = r; }
  __syncthreads();
  p = s_p[0]; reads = s_rd[0]; r = s_r[0];
  for (int k = 1; k < nwv; k++) {
    const double p2 = s_p[k];
    const uint32_t rd2 = s_rd[k];
    const int r2 = s_r[k];
    const bool take = r2 >= 0 && (r < 0 || p2 < p || (p2 == p && (rd2 > reads || (rd2 == reads && r2 < r))));
    if (take) { p = p2; reads = rd2; r = r2; }
  }
  return r;
}

// (one block of 1024 threads)  Applies the birth of `raw` out of partition `from`, then plans the compare of the round that
// follows: a cache hit needs nothing; a miss takes the next batch buffer and fills it with `raw` plus the best bud
// candidates of the last evaluation (the significant ones k2_pupdate listed, in b_bud's own order: p ascending, reads
// descending) that are not cached yet - they are the likely next centres, and a wrong guess only costs its share of one
// pass over the k-mer records - and builds the batch's k-mer tables.
// what the aligner launches of the coming chain work on (Ctl2::nalign / abuf / acentre); `slot` is the new centre's cache slot,
// nb the size of the batch just planned (0: the centre was cached)
static __device__ __forceinline__ void plan_aligner(const Eng2 &E, int centre, int slot, int nb) {
  Ctl2 *ctl = E.ctl;
  const int tid = threadIdx.x;
  __syncthreads();                                                      // (bcentre[] of a fresh batch is written)
  if (E.align_at_commit) {
    if (tid < KB_MAX) ctl->acentre[tid] = tid == slot % KB_MAX ? centre : -1;
    if (tid == 0) { ctl->nalign = KB_MAX; ctl->abuf = slot / KB_MAX; }
    if (tid < 2 * KB_MAX) E.blist_n[tid] = 0;
  } else {
    if (tid < KB_MAX) ctl->acentre[tid] = tid < nb ? ctl->bcentre[tid] : -1;
    if (tid == 0) { ctl->nalign = nb; ctl->abuf = ctl->bbuf; }
    if (nb > 0 && tid < 2 * KB_MAX) E.blist_n[tid] = 0;
  }
}
// The best listed candidates of the last evaluation (the significant ones k2_pupdate listed, in b_bud's own order: p ascending,
// reads descending) that are neither cached (s_tab / s_bits: the slot table and a bitmap of its low indices) nor `raw`:
// positions [first, KB_MAX) of s_bc, *s_nb = number of positions filled.  They are the likely next centres.
static __device__ __forceinline__ void plan_select(const Eng2 &E, int raw, int first, int *s_bc, int *s_nb, const int *s_tab, const uint32_t *s_bits,
                                                   double *s_p, uint32_t *s_rd, int *s_r, int nslots) {
  const PartState &P = E.P;
  const SampleDev &S = E.S;
  const int tid = threadIdx.x;
  // EVERY listed candidate is looked at (a deep sample lists tens of thousands, in no order: the first 8 blockDim of them - all
  // that round 5 looked at - need not hold the best ones, and with 512-thread blocks held half as many: the deep 100 k workload of
  // bench.py missed the cache 55 times in 256 rounds that way, 38 times with 1024-thread blocks, profiles/r09h).  A thread keeps the
  // PLAN_PER best keys it has seen - the KB_MAX best of all are among the threads' own KB_MAX best.
  const int M = *E.sig_n;
  auto before = [](double p, uint32_t rd, int r, double p2, uint32_t rd2, int r2) __attribute__((always_inline)) -> bool {   // b_bud's order (+ index)
    return r2 < 0 || p < p2 || (p == p2 && (rd > rd2 || (rd == rd2 && r < r2)));
  };
  double kp[PLAN_PER];
  uint32_t krd[PLAN_PER];
  int kr[PLAN_PER];
#pragma unroll
  for (int j = 0; j < PLAN_PER; j++) { kr[j] = -1; kp[j] = 0.0; krd[j] = 0; }
  for (int base = 0; base < M; base += PLAN_PER * (int)blockDim.x) {
    double np[PLAN_PER];
    uint32_t nrd[PLAN_PER];
    int nr[PLAN_PER];
#pragma unroll
    for (int j = 0; j < PLAN_PER; j++) {                       // (the chunk's candidates requested together)
      const int q = base + tid + j * (int)blockDim.x;
      nr[j] = -1; np[j] = 0.0; nrd[j] = 0;
      if (q < M) {
        const int r = E.sig_list[q];
        bool ok = r != raw && !P.slot0[r];
        if (ok) {
          if (r < PLAN_BITS) ok = !((s_bits[r >> 5] >> (r & 31)) & 1u);
          else for (int t = 0; ok && t < nslots; t++) if (s_tab[t] == r) ok = false;
        }
        if (ok) { nr[j] = r; np[j] = P.p[r]; nrd[j] = S.reads[r]; }
      }
    }
    if (base == 0) {
#pragma unroll
      for (int j = 0; j < PLAN_PER; j++) { kr[j] = nr[j]; kp[j] = np[j]; krd[j] = nrd[j]; }
      continue;
    }
#pragma unroll
    for (int j = 0; j < PLAN_PER; j++) {                       // a further chunk: each of its candidates against the worst key kept
      if (nr[j] < 0) continue;
      int w = 0;
#pragma unroll
      for (int k = 1; k < PLAN_PER; k++) if (kr[w] >= 0 && before(kp[w], krd[w], kr[w], kp[k], krd[k], kr[k])) w = k;   // w: an empty place, else the worst key
      if (before(np[j], nrd[j], nr[j], kp[w], krd[w], kr[w])) {
#pragma unroll
        for (int k = 0; k < PLAN_PER; k++) if (k == w) { kr[k] = nr[j]; kp[k] = np[j]; krd[k] = nrd[j]; }
      }
    }
  }
  for (int sel = first; sel < KB_MAX; sel++) {
    double bp = 0.0;
    uint32_t brd = 0;
    int br = -1;
#pragma unroll
    for (int j = 0; j < PLAN_PER; j++)
      if (kr[j] >= 0 && (br < 0 || kp[j] < bp || (kp[j] == bp && (krd[j] > brd || (krd[j] == brd && kr[j] < br))))) {
        bp = kp[j]; brd = krd[j]; br = kr[j];
      }
    const int win = block_best(bp, brd, br, s_p, s_rd, s_r);
    if (win < 0) break;
#pragma unroll
    for (int j = 0; j < PLAN_PER; j++) if (kr[j] == win) kr[j] = -1;
    if (tid == 0) { s_bc[sel] = win; *s_nb = sel + 1; }
  }
  __syncthreads();
}

// k-mer tables of a batch of nb centres (bc[], in LDS or global memory): byte k of tab8[id] = min(count of 5-mer id in centre k, 63) + 0x7F
// (the screen's compare then is one subtraction: bit 7 of byte - rank is set exactly when rank < count), the full counts and
// the ordered 5-mers.  s_cnt: [KB_MAX][1024] words of LDS.
static __device__ __forceinline__ void build_batch_tables(const SampleDev &S, const Cache2 &C, int nb, const int *bc, uint32_t *s_cnt) {
  const int tid = threadIdx.x;
  for (int k = tid; k < KB_MAX * NKMER; k += blockDim.x) s_cnt[k] = 0;
  __syncthreads();
  for (int k = 0; k < nb; k++) {
    const int c = bc[k];
    const int nkc = S.len[c] - KMER_SIZE + 1;
    const uint16_t *crow = S.kord + (size_t)c * S.LK;
    uint16_t *ko = C.ord + (size_t)k * S.LK;
    for (int i = tid; i < S.LK; i += blockDim.x) {
      const uint32_t km = crow[i] & 1023u;
      if (i < nkc) atomicAdd(&s_cnt[k * NKMER + km], 1u);
      ko[i] = i < nkc ? (uint16_t)km : (uint16_t)0xFFFF;
    }
  }
  __syncthreads();
  for (int id = tid; id < NKMER; id += blockDim.x) {
    uint32_t lo = 0, hi = 0;
    for (int k = 0; k < KB_MAX; k++) {
      const uint32_t c = k < nb ? s_cnt[k * NKMER + id] : 0u;
      const uint32_t sat = (c < RANK_SAT ? c : RANK_SAT) + 0x7Fu;
      if (k < 4) lo |= sat << (8 * k); else hi |= sat << (8 * (k - 4));
      C.full[(size_t)k * NKMER + id] = (uint16_t)c;
    }
    C.tab8[id] = make_uint2(lo, hi);
  }
  if (C.cbits && S.kbits)   // presence bitmaps of the batch's centres (the screen's prefilter): their rows of SampleDev::kbits
    for (int q = tid; q < KB_MAX * 32; q += blockDim.x) C.cbits[q] = (q >> 5) < nb ? S.kbits[(size_t)bc[q >> 5] * 32 + (q & 31)] : 0u;
}

// The next batch's compare under the persistent tail (Eng2::pf_on): choose up to KB_MAX of the best candidates
// that are not cached, give them the next batch buffer (never `keepbuf`, whose rows the coming round's commit reads) and describe
// the compare in E.pf_ctl.  The host sees Ctl2::pf_seq move in the round's result block and launches the compare on its second
// stream; the rounds go on meanwhile.  s_tab / s_bits as for plan_select, current as of now.
static __device__ __forceinline__ void plan_prefetch(const Eng2 &E, int raw, int keepbuf, int *s_misc, int *s_tab, uint32_t *s_bits, double *s_p, uint32_t *s_rd) {
  const SampleDev &S = E.S;
  Ctl2 *ctl = E.ctl;
  const Cache2 &C = E.C;
  const int tid = threadIdx.x;
  const int nslots = C.NBUF * KB_MAX;
  int *s_nb = s_misc, *s_bc = s_misc + 4, *s_r = s_misc + 12, *s_pb = s_misc + 2;
  if (tid == 0) {
    int bbuf = ctl->next_bbuf;
    if (bbuf == keepbuf) bbuf = (bbuf + 1) % C.NBUF;
    *s_pb = bbuf;
    *s_nb = 0;
  }
  __syncthreads();
  const int bbuf = *s_pb;
  // what the buffer held is gone (whether or not the plan goes through: a batch buffer is a cache)
  if (tid < KB_MAX) {
    const int c = s_tab[bbuf * KB_MAX + tid];
    if (c >= 0 && c < PLAN_BITS) atomicAnd(&s_bits[c >> 5], ~(1u << (c & 31)));
    s_tab[bbuf * KB_MAX + tid] = -1;
  }
  __syncthreads();
  plan_select(E, raw, 0, s_bc, s_nb, s_tab, s_bits, s_p, s_rd, s_r, nslots);
  const int nb = *s_nb;
  if (nb < E.pf_min) {
    // too few candidates left for a pass over the k-mer records to be worth it: no further attempt until something is planned
    if (tid == 0) ctl->last_bbuf = -2;
    return;
  }
  Ctl2 *pc = E.pf_ctl;
  if (tid < KB_MAX) {
    const int k = tid;
    const int c = k < nb ? s_bc[k] : -1;
    pc->bcentre[k] = c; pc->acentre[k] = c;
    pc->breads[k] = c >= 0 ? S.reads[c] : 0u;
    pc->blen[k] = c >= 0 ? S.len[c] : 0;
    C.slot_centre[bbuf * KB_MAX + k] = c;
  }
  if (tid < 2 * KB_MAX) E.pf_blist_n[tid] = 0;
  if (tid == 0) { pc->nbatch = nb; pc->bbuf = bbuf; pc->nalign = nb; pc->abuf = bbuf; }
  gcn_drain_stores();
  __syncthreads();
  if (tid == 0) {
    // the descriptor is complete: its sequence number goes out last, behind an agent-scope release - the gate kernel of the
    // compare's chain, already waiting on the second stream (k2_pf_gate), takes it from there
    const int seq = ctl->pf_seq + 1;
    gcn_release_agent();
    gcn_store_agent((uint32_t *)&pc->pf_seq, (uint32_t)seq);
    ctl->next_bbuf = (bbuf + 1) % C.NBUF;
    ctl->pf_seq = seq; ctl->pf_bbuf = bbuf; ctl->prev_bbuf = ctl->last_bbuf; ctl->last_bbuf = bbuf;
    ctl->pf_mask |= 1ull << (bbuf & 63);
    ctl->pf_centres += nb;
  }
}

// The plan itself, given the LDS layout of apply_birth_and_plan (s_misc + 32: the slot table, if have_tab says it is there already)
static __device__ __forceinline__ void prefetch_plan_now(const Eng2 &E, int raw, int keepbuf, uint32_t *s_cnt, int *s_misc, bool have_tab) {
  const Cache2 &C = E.C;
  const int tid = threadIdx.x;
  const int nslots = C.NBUF * KB_MAX;
  int *s_tab = s_misc + 32;
  uint32_t *s_bits = s_cnt;
  double *s_p = (double *)(s_cnt + PLAN_BITS / 32);
  uint32_t *s_rd = (uint32_t *)(s_p + 16);
  const unsigned long long tpl = E.ktime ? gcn_wall_clock() : 0ull;
  for (int q = tid; q < PLAN_BITS / 32; q += blockDim.x) s_bits[q] = 0;
  __syncthreads();
  for (int q = tid; q < nslots; q += blockDim.x) {
    const int c = have_tab ? s_tab[q] : C.slot_centre[q];
    s_tab[q] = c;
    if (c >= 0 && c < PLAN_BITS) atomicOr(&s_bits[c >> 5], 1u << (c & 31));
  }
  __syncthreads();
  plan_prefetch(E, raw, keepbuf, s_misc, s_tab, s_bits, s_p, s_rd);
  if (E.ktime && tid == 0) E.ktime[KT_PLAN] += gcn_wall_clock() - tpl;
}

// hint (optional, the serial end of a round has them at hand; nullptr: read from memory): hint[0] = reads of `raw`,
// hint[1], hint[2] = Ctl2::n0 / low0 as of now, hint[3] != 0: s_misc + 32 already holds a copy of Cache2::slot_centre
// defer_prefetch: the plan of the next prefetch compare - nobody but the second stream waits for it - is left to the caller
// (prefetch_plan_deferred below, behind the release of the other blocks); returns true when one is due
static __device__ __forceinline__ bool apply_birth_and_plan(const Eng2 &E, int raw, int from, uint32_t *s_cnt /*[KB_MAX][1024]*/, int *s_misc,
                                                            const int32_t *hint = nullptr, bool defer_prefetch = false) {
  const PartState &P = E.P;
  const SampleDev &S = E.S;
  Ctl2 *ctl = E.ctl;
  const Cache2 &C = E.C;
  const int tid = threadIdx.x;
  const int nslots = C.NBUF * KB_MAX;
  int *s_nb = s_misc, *s_hit = s_misc + 1, *s_trig = s_misc + 3;
  int *s_bc = s_misc + 4;                                           // [KB_MAX]
  int *s_r = s_misc + 12;                                           // [16]
  int *s_tab = s_misc + 32;                                         // [nslots] copy of slot_centre
  uint32_t *s_bits = s_cnt;                                         // [PLAN_BITS / 32] cached uniques among the low indices
  double *s_p = (double *)(s_cnt + PLAN_BITS / 32);                 // [16]
  uint32_t *s_rd = (uint32_t *)(s_p + 16);                          // [16]
  if (tid == 0) {
    const int newi = ctl->nclust;
    const uint32_t reads_new = hint ? (uint32_t)hint[0] : S.reads[raw];
    P.clust_of[raw] = newi;
    P.lock[raw] = 0;                                                // bi_assign_center unlocks the members (cluster.cpp:377)
    P.slot0[raw] = 1;
    E.moved[raw] = 1;
    if (from == 0) { const int n0 = (hint ? hint[1] : ctl->n0) - 1; ctl->n0 = n0; if (n0 < (hint ? hint[2] : ctl->low0)) ctl->low0 = n0; }
    P.creads[newi] = reads_new;
    P.creads[from] -= reads_new;
    P.centre_of[newi] = raw;
    P.update_e[newi] = 1; P.check_locks[newi] = 1;
    P.update_e[from] = 1;
    ctl->nclust = newi + 1;
    ctl->centre = raw;
    ctl->bfrom = from;
    ctl->nsh_base = 0;
    *s_hit = -1; *s_trig = 0;
  }
  __syncthreads();
  const bool have_tab = hint && hint[3];
  if (have_tab) { for (int q = tid; q < nslots; q += blockDim.x) if (s_tab[q] == raw) *s_hit = q; }
  else for (int q = tid; q < nslots; q += blockDim.x) if (C.slot_centre[q] == raw) *s_hit = q;   // (at most one slot holds it)
  __syncthreads();
  const int hit = *s_hit;
  if (hit >= 0) {
    if (tid == 0) {
      ctl->slot = hit; ctl->nbatch = 0; ctl->need_compare = 0;
      if (E.pf_on) {
        // a centre out of a prefetched batch whose compare is still running: give it a moment inside the launch, else leave the
        // launch - no round may read the batch's rows before PfSync::done says they are there (Ctl2::pf_wait)
        const int hb = hit / KB_MAX;
        int wait = 0;
        if ((ctl->pf_mask >> (hb & 63)) & 1ull) ctl->pf_hits += 1;
        uint32_t done = gcn_load_agent(&E.pfsync->done);
        const uint32_t seq = (uint32_t)ctl->pf_seq;
        if (hb == ctl->pf_bbuf && (int32_t)(done - seq) < 0) {
          const unsigned long long t0 = gcn_wall_clock();
          while ((int32_t)(done - seq) < 0 && gcn_wall_clock() - t0 < E.pf_wait_ticks) { gcn_poll_pause(); done = gcn_load_agent(&E.pfsync->done); }
          if ((int32_t)(done - seq) < 0) { wait = (int)seq; ctl->pf_exits += 1; } else ctl->pf_spins += 1;
          if (E.ktime) E.ktime[KT_PFWAIT] += gcn_wall_clock() - t0;
        }
        ctl->pf_wait = wait;
        // Time to plan the next prefetch (the second stream being free)?  When the rounds reach the batch planned last - or already
        // when they are pf_early positions into the one before it: a compare beside the tail takes longer than the rounds of one
        // batch, so it has to start before the batch in front of it is used up
        // (... and only while the run has the device to itself: Eng2::pf_plan, per launch)
        const bool due = E.pf_plan && (hb == ctl->last_bbuf || (hb == ctl->prev_bbuf && (hit % KB_MAX) >= E.pf_early));
        *s_trig = (due && (int32_t)(done - seq) >= 0) ? 1 : 0;
      }
    }
    plan_aligner(E, raw, hit, 0);
    if (!E.pf_on) return false;
    __syncthreads();
    if (!*s_trig) return false;
    if (defer_prefetch) return true;
    prefetch_plan_now(E, raw, hit / KB_MAX, s_cnt, s_misc, have_tab);
    return false;
  }
  if (tid == 0) {
    const int bbuf = ctl->next_bbuf;
    ctl->next_bbuf = (bbuf + 1) % C.NBUF;
    for (int k = 0; k < KB_MAX; k++) C.slot_centre[bbuf * KB_MAX + k] = -1;
    ctl->bbuf = bbuf;
    ctl->slot = bbuf * KB_MAX;
    ctl->prev_bbuf = -1; ctl->last_bbuf = bbuf;
    ctl->pf_mask &= ~(1ull << (bbuf & 63));
    ctl->pf_wait = 0;
    s_bc[0] = raw;
    *s_nb = 1;
  }
  for (int q = tid; q < PLAN_BITS / 32; q += blockDim.x) s_bits[q] = 0;
  __syncthreads();
  for (int q = tid; q < nslots; q += blockDim.x) {
    const int c = C.slot_centre[q];
    s_tab[q] = c;
    if (c >= 0 && c < PLAN_BITS) atomicOr(&s_bits[c >> 5], 1u << (c & 31));
  }
  __syncthreads();
  // ---- prediction: the best keys among the listed candidates ----
  plan_select(E, raw, 1, s_bc, s_nb, s_tab, s_bits, s_p, s_rd, s_r, nslots);
  const int nb = *s_nb;
  const int bbuf_now = ctl->bbuf;
  // (no prefetch is planned here: the compare of THIS batch has the device to itself, and the round that follows - its centre is
  //  this batch's first position - plans the next one, which then runs under the rounds of this batch)
  if (tid < KB_MAX) {
    const int k = tid;
    if (k < nb) {
      const int c = s_bc[k];
      ctl->bcentre[k] = c; ctl->breads[k] = S.reads[c]; ctl->blen[k] = S.len[c];
      C.slot_centre[bbuf_now * KB_MAX + k] = c;
    } else { ctl->bcentre[k] = -1; ctl->breads[k] = 0; ctl->blen[k] = 0; }
  }
  __syncthreads();
  build_batch_tables(S, C, nb, s_bc, s_cnt);
  if (tid == 0) { ctl->nbatch = nb; ctl->need_compare = 1; }
  plan_aligner(E, raw, bbuf_now * KB_MAX, nb);
  return false;
}

static __device__ __forceinline__ void publish_copy(const Eng2 &E, Round2Out *out, int ring, int seq) {
  // plain stores to pinned host memory, then the sequence number: the host polls it instead of copying and synchronising
  __syncthreads();
  int tot = 0;
  for (int l = 0; l < SH_LEVELS; l++) tot += out->cnt[l];
  if (tot > E.mov_inline) tot = E.mov_inline;
  const int used16 = (int)((offsetof(Round2Out, mov) + (size_t)12 * tot + 15) / 16);
  const uint4 *src = (const uint4 *)out;
  uint4 *dst = (uint4 *)(E.hblk + ring);
  for (int i = threadIdx.x; i < used16; i += blockDim.x) {
    uint4 v = src[i];
    if (i == 0) v.x = (uint32_t)(seq - 1);               // (word 0 of the block is `seq`: not yet)
    dst[i] = v;
  }
  // the waves that copied wait for their own stores at the barrier; ONE system-scope release then orders the whole block
  // before the sequence number (a system fence in each of the 16 waves cost 4 of this kernel's 17.7 us: profiles/r02t vs r02q)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(&E.hblk[ring].seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static __device__ void publish_block(const Eng2 &E, Round2Out *out, int ring) {
  const int seq = E.ctl->pub_seq + 1;
  publish_copy(E, out, ring, seq);
  if (threadIdx.x == 0) E.ctl->pub_seq = seq;
}

static __device__ __forceinline__ void clear_block(Round2Out *nx) {
  if (threadIdx.x < SH_LEVELS) nx->cnt[threadIdx.x] = 0;
  if (threadIdx.x < 4) { nx->stat[threadIdx.x] = 0; nx->pad0[threadIdx.x] = 0; }
  if (threadIdx.x == 0) {
    nx->bud.nties[0] = 0; nx->bud.nties[1] = 0; nx->bud.valid = 0; nx->bud.found[0] = 0; nx->bud.found[1] = 0;
    nx->bud.auto_applied = 0; nx->halt = H2_NONE; nx->birth_applied = 0; nx->nsh = 0; nx->nlev = 0; nx->nbatch = 0;
    nx->kord = 0; nx->paused = 0;
  }
}

// The serial end of a round (ONE block of 1024 threads): folds the reads deltas and class counts of the round's shuffle calls,
// finishes b_bud's arg-min over the block minima, lists ties / near ties, takes the decision that is the device's to take,
// applies the birth and plans the coming round's compare (apply_birth_and_plan), clears the next result block.  The caller
// publishes.  cs: which of the chain's `nlev` shuffle calls ran and whether the evaluation behind them stands.
// kord: 0 in a launch chain (k2_birth); the launch ordinal in the persistent tail (k3_tail), which also wants to know whether to
// leave the launch after this round (Ctl2::kexit) and pauses when the round's movers do not fit the block.
// go (the persistent tail): called once everything the OTHER blocks read in their next phase is written - it lets them go; what
// follows it is this block's and the host's alone (the plan of the next prefetch compare, the prefetch fields of the result block)
// EXT_CNT / ext_cnt (the persistent tail): the 32 KB of LDS the plan's k-mer tables and bitmaps want are the caller's - k3_tail
// lends the work lists of its sweeps, which are dead while the serial section runs - instead of a static array of this function
struct BirthNoGo { __device__ void operator()() const {} };
template <bool EXT_CNT = false, typename GO = BirthNoGo>
static __device__ __forceinline__ void birth_body(const Eng2 &E, int nlev, const Chain2 cs, BudKey init, const BudKey *__restrict__ partial, int nblocks, int kord,
                                                  GO &&go = GO(), uint32_t *ext_cnt = nullptr) {
  // The section is one block's work while (in the persistent tail) every other block waits: it is written as few dependent
  // memory round trips as the logic allows.  STAGE 1 requests everything that depends on nothing computed here - the scalars
  // thread 0 will want (one lane each, into LDS), the reads deltas, the blocks' statistics and minima, the head of the
  // candidate list, the cache's slot table - in one go; STAGE 2 the p-values of the listed candidates; the ties then sit in
  // LDS for the decision.  (Round 3's form took about fifteen trips, 24 us a round at 10^6 uniques.)
  Ctl2 *ctl = E.ctl;
  uint32_t *s_cnt;
  if constexpr (EXT_CNT) s_cnt = ext_cnt;
  else {
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt_own[KB_MAX * NKMER];
    s_cnt = s_cnt_own;
  }
  __shared__ int s_misc[32 + NBUF_MAX * KB_MAX];
  __shared__ int s_halt, s_raw, s_from, s_evalok, s_nt[2], s_nnear;
  __shared__ BudKey s_k[2][16];
  __shared__ int32_t s_pre[64];
  __shared__ int32_t s_hint[4];
  __shared__ BudTie s_tie[BUD_TIES];                     // the first ties of track 0, where the decision reads them
  enum { PRE_N0 = 0, PRE_LOW0, PRE_NALIGN, PRE_SLOT, PRE_NBATCH, PRE_MAXCLUST, PRE_ERR, PRE_NWFLAG, PRE_BLKCNT, PRE_STATN, PRE_SIGN,
         PRE_NEEDCMP, PRE_HCONS, PRE_NALRAN, PRE_N0D = 14 /* 2 SH_LEVELS */, PRE_BLIST = 34 /* 2 KB_MAX */, PRE_CNT = 50 /* SH_LEVELS */ };
  static_assert(PRE_N0D + 2 * SH_LEVELS <= PRE_BLIST && PRE_BLIST + 2 * KB_MAX <= PRE_CNT && PRE_CNT + SH_LEVELS <= 64, "scalar slots");
  const PartState &P = E.P;
  const SampleDev &S = E.S;
  const int tid = threadIdx.x;
  const int ring = ctl->pub_seq % RING2;
  Round2Out *out = E.dblk + ring;
  const int nclust = ctl->nclust;
  const bool tr = E.trace && ctl->pub_seq == E.trace_seq && threadIdx.x == 0;
#define D2_TRB(PHASE) do { if (tr) E.trace[((size_t)6 * TRACE_BLOCKS) * 8 + (PHASE)] = gcn_clock(); D2_KSUB(6, PHASE, true); } while (0)
  D2_TRB(0);
  if (kord && E.pf_on && E.pf_sync) {
    // (measurement knob, DADA2HIP_V3_PF_SYNC=1: every prefetch compare is waited for at the next round's serial end, with the whole
    //  tail resident and idle - what the compare's kernels cost beside a tail that does nothing)
    if (tid == 0) {
      const uint32_t seq = (uint32_t)ctl->pf_seq;
      const unsigned long long t0 = gcn_wall_clock();
      while ((int32_t)(gcn_load_agent(&E.pfsync->done) - seq) < 0 && gcn_wall_clock() - t0 < 4 * E.pf_wait_ticks) gcn_poll_pause();
    }
    __syncthreads();
  }
  // ---- stage 1: requests ----
  const int32_t *pa = nullptr;
  switch (tid) {
    case PRE_N0: pa = &ctl->n0; break;
    case PRE_LOW0: pa = &ctl->low0; break;
    case PRE_NALIGN: pa = &ctl->nalign; break;
    case PRE_SLOT: pa = &ctl->slot; break;
    case PRE_NBATCH: pa = &ctl->nbatch; break;
    case PRE_MAXCLUST: pa = &ctl->max_clust; break;
    case PRE_ERR: pa = P.err_flag; break;
    case PRE_NWFLAG: pa = S.nw_flag; break;
    case PRE_BLKCNT: pa = E.T.blk_count; break;
    case PRE_STATN: pa = E.stat_n; break;
    case PRE_SIGN: pa = E.sig_n; break;
    case PRE_NEEDCMP: pa = &ctl->need_compare; break;
    case PRE_HCONS: pa = &ctl->hcons_seen; break;
    case PRE_NALRAN: pa = &ctl->nalign_ran; break;
    default:
      if (tid >= PRE_N0D && tid < PRE_N0D + 2 * cs.nexec) pa = E.n0d + (tid - PRE_N0D);
      else if (tid >= PRE_BLIST && tid < PRE_BLIST + 2 * KB_MAX) pa = E.blist_n + (tid - PRE_BLIST);
      else if (tid >= PRE_CNT && tid < PRE_CNT + SH_LEVELS) pa = &out->cnt[tid - PRE_CNT];
  }
  const int32_t pv = pa ? *pa : 0;
  uint4 sp = make_uint4(0, 0, 0, 0);                      // this thread's share of the store pass's class counts (masked below)
  if (tid < 8192) sp = ((const uint4 *)E.stat_part)[tid];   // (the buffer holds 8192 entries; what lies past the pass's grid is dropped below)
  BudKey pk0 = init, pk1 = init;                          // ... of the blocks' minima
  if (cs.eval_ok && tid < nblocks) { pk0 = partial[2 * tid]; pk1 = partial[2 * tid + 1]; }
  int sq[2] = {-1, -1};                                   // ... of the candidate list (entries past its end are dropped below)
  if (cs.eval_ok) {
    if (tid < S.N) sq[0] = E.sig_list[tid];
    if (tid + (int)blockDim.x < S.N) sq[1] = E.sig_list[tid + blockDim.x];
  }
  const int nslots = E.C.NBUF * KB_MAX;
  int *s_tab = s_misc + 32;
  for (int q = tid; q < nslots; q += blockDim.x) s_tab[q] = E.C.slot_centre[q];
  // fold the chain's partition-read deltas into the reads
  for (int i = tid; i < nclust; i += blockDim.x) {
    int32_t d = 0;
    for (int l = 0; l < cs.nexec; l++) { d += E.dlt[(size_t)l * E.ccap + i]; E.dlt[(size_t)l * E.ccap + i] = 0; }   // (the rows of calls that did not run hold zeros)
    if (d) P.creads[i] += (uint32_t)d;
  }
  if (tid < 64) s_pre[tid] = pv;
  if (tid < 2) s_nt[tid] = 0;
  __syncthreads();
  {   // class statistics of the chain's store pass, if it had one: the blocks' partial counts (k2_shuffle<true>)
    const int np = s_pre[PRE_STATN];
    if (np > 0) {
      uint32_t v[4] = {0, 0, 0, 0};
      if (tid < np) { v[0] = sp.x; v[1] = sp.y; v[2] = sp.z; v[3] = sp.w; }
      for (int b = tid + blockDim.x; b < np; b += blockDim.x) {
        const uint4 q = ((const uint4 *)E.stat_part)[b];
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if ((tid & 63) == 0 && v[k]) atomicAdd(&out->stat[k], (unsigned long long)v[k]);
      }
    }
  }
  if (tid == 0) {
    *E.stat_n = 0;
    s_nnear = 0;
    // partition 0's member count through the chain's shuffle calls: within one call only the number it lost is known, not
    // the order of losses and gains, so the running minimum is taken as if all losses came first (a lower bound)
    int n0 = s_pre[PRE_N0], low0 = s_pre[PRE_LOW0];
    for (int l = 0; l < cs.nexec; l++) {
      const int lost = s_pre[PRE_N0D + 2 * l], gained = s_pre[PRE_N0D + 2 * l + 1];
      if (lost | gained) { E.n0d[2 * l] = 0; E.n0d[2 * l + 1] = 0; }
      if (n0 - lost < low0) low0 = n0 - lost;
      n0 += gained - lost;
    }
    ctl->n0 = n0; ctl->low0 = low0;
    s_hint[1] = n0; s_hint[2] = low0; s_hint[3] = 1;
  }
  D2_TRB(1);
  // ---- second stage of b_bud's arg-min (cluster.cpp:284-308): the block minima of k2_pupdate, then the exact ties of the
  //      best key and every other listed candidate whose non-zero p is within BUD_NEAR of it (engine.h) ----
  if (cs.eval_ok) {
    BudKey b0 = init, b1 = init;
    if (tid < nblocks) { if (bud_better(pk0.p, pk0.reads, b0)) b0 = pk0; if (bud_better(pk1.p, pk1.reads, b1)) b1 = pk1; }
    for (int k = tid + blockDim.x; k < nblocks; k += blockDim.x) {
      if (bud_better(partial[2 * k].p, partial[2 * k].reads, b0)) b0 = partial[2 * k];
      if (bud_better(partial[2 * k + 1].p, partial[2 * k + 1].reads, b1)) b1 = partial[2 * k + 1];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      BudKey t;
      t.p = __shfl_xor(b0.p, o, 64); t.reads = __shfl_xor(b0.reads, o, 64);
      if (bud_better(t.p, t.reads, b0)) b0 = t;
      t.p = __shfl_xor(b1.p, o, 64); t.reads = __shfl_xor(b1.reads, o, 64);
      if (bud_better(t.p, t.reads, b1)) b1 = t;
    }
    if ((tid & 63) == 0) { s_k[0][tid >> 6] = b0; s_k[1][tid >> 6] = b1; }
    __syncthreads();
    b0 = s_k[0][0]; b1 = s_k[1][0];
    for (int k = 1; k < ((int)blockDim.x >> 6); k++) {
      if (bud_better(s_k[0][k].p, s_k[0][k].reads, b0)) b0 = s_k[0][k];
      if (bud_better(s_k[1][k].p, s_k[1][k].reads, b1)) b1 = s_k[1][k];
    }
    const bool found0 = bud_better(b0.p, b0.reads, init), found1 = bud_better(b1.p, b1.reads, init);
    // (no window when the best p-value is clearly not significant: b_bud then gives no birth whatever the order; candidates
    //  that are not significant were not listed, and an exact tie of a significant key is significant itself)
    const bool sig0 = b0.p * S.N < 2.0 * E.bp.omegaA, sig1 = b1.p < 2.0 * E.bp.omegaP;
    const double thr0 = sig0 ? b0.p * (1.0 + BUD_NEAR) + 2e-323 : -1.0, thr1 = sig1 ? b1.p * (1.0 + BUD_NEAR) + 2e-323 : -1.0;
    const int M = s_pre[PRE_SIGN];
    BudOut *bo = &out->bud;
    // ---- stage 2: the listed candidates' keys, the first two per thread requested together ----
    double ps_[2] = {0.0, 0.0};
    uint32_t rds_[2] = {0, 0};
    uint8_t prs_[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int q = tid + j * (int)blockDim.x;
      if (q >= M) sq[j] = -1;
      if (sq[j] >= 0) { ps_[j] = P.p[sq[j]]; rds_[j] = S.reads[sq[j]]; prs_[j] = S.prior[sq[j]]; }
    }
    auto consider = [&](int r, double p, uint32_t reads, bool prior) __attribute__((always_inline)) {
      for (int track = 0; track < 2; track++) {
        if (track == 1 && !prior) continue;
        const BudKey &bk = track ? b1 : b0;
        if (!(track ? (found1 && sig1) : (found0 && sig0))) continue;
        const bool exact = p == bk.p && reads == bk.reads;
        const bool near = !exact && p != 0.0 && p <= (track ? thr1 : thr0);
        if (!exact && !near) continue;
        if (near && track == 0) s_nnear = 1;
        const int k = atomicAdd(&s_nt[track], 1);
        if (k < TIES_FULL) {
          BudTie t;
          t.raw = r; t.comp_i = P.comp_i[r]; t.comp_ham = P.comp_ham[r]; t.comp_lam = P.comp_lam[r];
          t.from = P.clust_of[r]; t.from_reads = P.creads[t.from]; t.p = p; t.pad = 0;
          if (k < BUD_TIES) { bo->ties[track][k] = t; if (track == 0) s_tie[k] = t; }
          E.ties_rec[(size_t)track * TIES_FULL + k] = t;
        }
        (track ? E.ties1 : E.ties0)[k] = r;
      }
    };
#pragma unroll
    for (int j = 0; j < 2; j++) if (sq[j] >= 0) consider(sq[j], ps_[j], rds_[j], prs_[j] != 0);
    for (int q = tid + 2 * (int)blockDim.x; q < M; q += blockDim.x) {
      const int r = E.sig_list[q];
      consider(r, P.p[r], S.reads[r], S.prior[r] != 0);
    }
    __syncthreads();
    if (tid == 0) {
      bo->best_p[0] = b0.p; bo->best_p[1] = b1.p;
      bo->best_reads[0] = b0.reads; bo->best_reads[1] = b1.reads;
      bo->found[0] = found0; bo->found[1] = found1;
      bo->nties[0] = s_nt[0]; bo->nties[1] = s_nt[1];
      bo->valid = 1;
      // ---- the decision (thread 0; everything it reads is in registers or LDS) ----
      s_k[0][0] = b0; s_k[1][0] = b1;                     // (kept for the decision block below)
      s_misc[2] = found0 ? 1 : 0; s_misc[3] = found1 ? 1 : 0;
    }
  }
  __syncthreads();
  D2_TRB(2);
  if (tid == 0) {
    out->nlev = nlev; out->nsh = cs.nexec; out->slot = s_pre[PRE_SLOT]; out->nbatch = s_pre[PRE_NBATCH];
    if (s_pre[PRE_NALRAN] > 0) ctl->nalign_ran = 0;
    if ((s_pre[PRE_NALIGN] > 0 || s_pre[PRE_NALRAN] > 0) && nlev > 0) {   // alignments / gapless pairs the aligner ran for this chain
      int nn = 0, ng = 0;
      for (int k = 0; k < KB_MAX; k++) { nn += s_pre[PRE_BLIST + k]; ng += s_pre[PRE_BLIST + KB_MAX + k]; }
      out->pad0[1] = nn; out->pad0[2] = ng;
    }
    const int errf = s_pre[PRE_ERR] | (s_pre[PRE_NWFLAG] ? 4 : 0), blkc = s_pre[PRE_BLKCNT];
    out->err_flag = errf;
    out->blk_count = blkc;
    int halt = H2_NONE, raw = -1, from = 0;
    if (!cs.eval_ok) { halt = H2_SHUFFLE_MORE; ctl->nsh_base += cs.nexec; }
    else if (nclust >= s_pre[PRE_MAXCLUST]) halt = H2_MAXCLUST;
    else {
      const BudKey b0 = s_k[0][0], b1 = s_k[1][0];
      const bool found0 = s_misc[2] != 0, found1 = s_misc[3] != 0;
      const int nt0 = s_nt[0];
      // the unambiguous case of b_bud (cluster.cpp:300-330) is the device's; the margins keep every decision that depends
      // on the last ulp of a p-value on the host
      const double pA = b0.p * S.N;
      // Several candidates with the same key: b_bud keeps the first one it meets, partitions in index order and each
      // partition's members in list order (cluster.cpp:284-308).  Settled here when the key is p = 0 exactly (an underflow
      // on any libm; nothing near it is listed) and the order follows without the member lists: the lowest partition holds
      // one candidate, or it is partition 0 and its candidates have never moved and never been its last member - their
      // slots are their indices (Eng2::moved).  Everything else stays the host's.
      int win = nt0 == 1 ? 0 : -1;
      if (found0 && nt0 > 1 && nt0 <= BUD_TIES && !s_nnear && b0.p == 0.0) {
        int cmin = 0x7FFFFFFF, ncmin = 0, kmin = -1, rmin = 0x7FFFFFFF;
        bool ok = true;
        const int low0 = s_hint[2];
        for (int k = 0; k < nt0; k++) {
          const int fr = s_tie[k].from, rw = s_tie[k].raw;
          if (fr < cmin) { cmin = fr; ncmin = 0; rmin = 0x7FFFFFFF; }
          if (fr == cmin) {
            ncmin++;
            if (fr == 0 && (E.moved[rw] || rw + 1 >= low0)) ok = false;
            if (rw < rmin) { rmin = rw; kmin = k; }
          }
        }
        if (ok && (ncmin == 1 || cmin == 0)) win = kmin;
      }
      if (win > 0) {
        const BudTie t = s_tie[0]; s_tie[0] = s_tie[win]; s_tie[win] = t;
        out->bud.ties[0][0] = s_tie[0]; out->bud.ties[0][win] = s_tie[win];
        win = 0;
      }
      const bool birthA = found0 && win == 0 && pA < E.omegaA * (1.0 - 1e-9);
      const bool noA = !found0 || pA >= E.omegaA * (1.0 + 1e-9);
      const bool noP = !found1 || b1.p >= E.omegaP * (1.0 + 1e-9);
      if (birthA) { raw = s_tie[0].raw; from = s_tie[0].from; s_hint[0] = (int32_t)b0.reads; }
      else if (noA && noP) halt = H2_NO_BIRTH;
      else halt = H2_HOST_DECIDE;
      if (halt == H2_NONE && (nclust + 2 > E.ccap || blkc + S.N > E.T.blk_cap)) halt = H2_CAPACITY;
    }
    // did the round's shuffles end with a call that moved nothing?  (not when MAX_SHUFFLE cut them short, Rmain.cpp:321: the
    // next round's first call then looks at everybody)
    if (cs.eval_ok) ctl->stable = (nlev == 0 || (cs.nexec > 0 && s_pre[PRE_CNT + cs.nexec - 1] == 0)) ? 1 : 0;
    out->halt = halt;
    out->birth_applied = halt == H2_NONE ? 1 : 0;
    out->nclust = nclust + (halt == H2_NONE ? 1 : 0);
    s_halt = halt; s_raw = raw; s_from = from; s_evalok = cs.eval_ok ? 1 : 0;
    if (halt != H2_NONE) { ctl->state = 1; ctl->halt = halt; }
  }
  __syncthreads();
  D2_TRB(3);
  if (s_evalok)   // b_p_update has consumed the flags (pval.cpp:24,37)
    for (int k = tid; k < nclust; k += blockDim.x) { P.update_e[k] = 0; P.check_locks[k] = 0; }
  __syncthreads();
  bool plan_due = false;
  if (s_halt == H2_NONE) plan_due = apply_birth_and_plan(E, s_raw, s_from, s_cnt, s_misc, s_hint, /*defer_prefetch=*/kord != 0);
  D2_TRB(4);
  if (kord && tid == 0) {
    int tot = 0;
    for (int l = 0; l < SH_LEVELS; l++) tot += s_pre[PRE_CNT + l];
    const bool pause = s_halt == H2_NONE && tot > E.mov_inline;      // the full lists stay in Eng2::movers until the host has them
    if (pause) { out->paused = 1; ctl->state = 1; ctl->halt = H2_NONE; }
    out->kord = kord;
    // leave the launch after this round?  a halt, a pause, a compare (or, aligning at commit time, the aligner) is due, or the
    // host's ring of result blocks would not take the NEXT block: publishing sequence number q reuses the slot of q - RING2
    const int seq = ctl->pub_seq + 1;
    // (... or the coming round's centre sits in a prefetched batch whose compare is still running)
    int ex = (s_halt != H2_NONE || pause || ctl->need_compare != 0 || E.align_at_commit != 0 || ctl->pf_wait != 0) ? 1 : 0;
    if (!ex && seq + 1 - s_pre[PRE_HCONS] > E.ring_limit) {
      const int hc = gcn_load_system((const int32_t *)E.hcons);
      ctl->hcons_seen = hc;
      if (seq + 1 - hc > E.ring_limit) ex = 1;
    }
    ctl->kexit = ex;
  }
  clear_block(E.dblk + ((ring + 1) % RING2));
  go();
  // ---- behind the release of the other blocks (persistent tail): they are in the coming round's commit, which touches none of
  //      what the plan reads (the candidate list, p-values, the slot table) - the next evaluation, which rewrites the list, lies
  //      behind a barrier this block has yet to arrive at ----
  if (plan_due) prefetch_plan_now(E, s_raw, s_misc[1] / KB_MAX, s_cnt, s_misc, /*have_tab=*/true);   // (s_misc[1]: the new centre's cache slot, apply_birth_and_plan's s_hit)
  __syncthreads();
  if (tid == 0) {
    if (s_halt == H2_NONE) *E.sig_n = 0;                     // consumed: the next evaluation lists afresh
    out->pf_seq = ctl->pf_seq; out->pf_wait = ctl->pf_wait;
    out->pf_stat[0] = ctl->pf_hits; out->pf_stat[1] = ctl->pf_spins; out->pf_stat[2] = ctl->pf_exits; out->pf_stat[3] = ctl->pf_centres;
  }
#undef D2_TRB
}

#ifndef D2_TAIL_TU
__global__ __launch_bounds__(1024) void k2_birth(Eng2 E, int nlev, BudKey init, const BudKey *__restrict__ partial, int nblocks) {
  Ctl2 *ctl = E.ctl;
  const bool halted = ctl->state != 0, starved = !E.has_compare && ctl->need_compare != 0;
  __syncthreads();                                                       // every thread has read both before thread 0 changes either
  if (halted) return;
  if (starved) {
    // the chain came without the batch compare its round needs: nothing has run, say so and halt (the block was cleared by
    // the previous chain's k2_birth)
    const int ring0 = ctl->pub_seq % RING2;
    Round2Out *o = E.dblk + ring0;
    if (threadIdx.x == 0) {
      o->halt = H2_NEED_COMPARE; o->nclust = ctl->nclust; o->birth_applied = 0; o->nlev = 0; o->nsh = 0; o->nbatch = 0; o->slot = ctl->slot;
      o->err_flag = *E.P.err_flag | (*E.S.nw_flag ? 4 : 0); o->blk_count = *E.T.blk_count;
      ctl->state = 1; ctl->halt = H2_NEED_COMPARE;
    }
    __syncthreads();
    clear_block(E.dblk + ((ring0 + 1) % RING2));
    publish_block(E, o, ring0);
    return;
  }
  if (threadIdx.x == 0) ctl->need_compare = 0;                           // the round's compare, if it needed one, has run
  const int ring = ctl->pub_seq % RING2;
  Round2Out *out = E.dblk + ring;
  const Chain2 cs = chain_state(ctl, out, nlev, E.max_shuffle);
  birth_body(E, nlev, cs, init, partial, nblocks, 0);
  publish_block(E, out, ring);
}

// the host's own b_bud decision (ties, near ties, prior births, capacity) applied, the coming round planned, the device resumed
__global__ __launch_bounds__(1024) void k2_host_birth(Eng2 E, int raw, int from) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[KB_MAX * NKMER];
  __shared__ int s_misc[32 + NBUF_MAX * KB_MAX];
  if (threadIdx.x == 0) { E.ctl->state = 0; E.ctl->halt = H2_NONE; }
  __syncthreads();
  apply_birth_and_plan(E, raw, from, s_cnt, s_misc);
  __syncthreads();
  if (threadIdx.x == 0) *E.sig_n = 0;
}
// compare_done: the batch compare the control block still announces has run already (it stood in front of a persistent launch
// that gave up at its entry barrier): the launch chains that take over must not run it again - k2_batch_lists would append every
// pair to the work lists a second time
__global__ void k2_resume(Eng2 E, int keep_list, int compare_done) {
  E.ctl->state = 0; E.ctl->halt = H2_NONE;
  if (!keep_list) *E.sig_n = 0;
  if (compare_done) { E.ctl->nbatch = 0; E.ctl->need_compare = 0; if (!E.align_at_commit) E.ctl->nalign = 0; }
}

// ---- a prefetch compare on the second stream (Eng2::pf_on): E is ITS argument block - ctl = the prefetch descriptor, C.tab8 /
//      full / ord and blist / blist_n its own copies.  The tables of the batch in front of the screen (the planner, inside the
//      persistent tail's serial section, only chose the centres); the completion word behind the aligner. ----
// The GATE in front of a prefetch compare's chain.  The host enqueues the chain of prefetch number k on the second stream BEFORE
// the tail has planned it (one chain ahead), with this kernel first: one lane that waits until the descriptor carries sequence
// number k - then the chain's kernels, next in the stream, start within microseconds of the plan instead of a host round trip
// later (the host sees a plan only when it consumes the round's result block, behind the replay of the blocks before it:
// 0.3-0.5 ms at 10^6 uniques, profiles/r07o).  It gives up when the host says the run is over (`quit`, pinned) or after its bound;
// it then marks the descriptor halted, which turns the rest of the chain into launches that find nothing to do, and the host
// sends the chain again when (if) the plan shows up in a block.  result (pinned): 1 = passed, 2 = gave up.
__global__ void k2_pf_gate(Eng2 E, int k, const int32_t *quit, int32_t *result) {
  if (threadIdx.x != 0) return;
  Ctl2 *pc = E.ctl;
  int res = 2;
  const unsigned long long t0 = gcn_wall_clock();
  for (unsigned n = 0;; n++) {
    if ((int32_t)(gcn_load_agent((const uint32_t *)&pc->pf_seq) - (uint32_t)k) >= 0) { res = 1; break; }
    if (gcn_wall_clock() - t0 > E.pf_gate_ticks || ((n & 15u) == 0u && gcn_load_system(quit) != 0)) break;
    gcn_poll_pause();
  }
  gcn_acquire_agent();
  pc->state = res == 1 ? 0 : 1;
  __hip_atomic_store(result, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(1024) void k2_pf_tables(Eng2 E) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[KB_MAX * NKMER];
  const int nb = E.ctl->nbatch;
  if (E.ctl->state != 0 || nb <= 0) return;
  build_batch_tables(E.S, E.C, nb, E.ctl->bcentre, s_cnt);
}
__global__ void k2_pf_done(Eng2 E) {
  if (threadIdx.x != 0 || E.ctl->state != 0) return;   // (a chain whose gate gave up has compared nothing)
  unsigned long long nn = 0, ng = 0;
  for (int k = 0; k < KB_MAX; k++) { nn += (unsigned long long)E.blist_n[k]; ng += (unsigned long long)E.blist_n[KB_MAX + k]; }
  E.pfsync->nnw += nn; E.pfsync->ngapless += ng;
  // everything the compare's kernels wrote is in memory (they ended); the word the persistent tail polls follows it
  gcn_release_agent();
  gcn_store_agent(&E.pfsync->done, (uint32_t)E.ctl->pf_seq);
}

// ---- k-mer screen against the batch's centres ---------------------------------------------------------------------------
// 16 lanes per unique as in k_screen, but one pass over the unique's k-mer record serves up to KB_MAX centres: the centre
// tables are interleaved (byte k of tab8[id] = min(count_k[id], 63) + 0x7F), so one 8-byte LDS read + two SWAR subtractions give
// "rank < count" for all centres.  A wave owns 16 consecutive uniques per macro-iteration (lane group g takes uniques
// 4g..4g+3 in turn): the four rows and their scalars are requested together before any of them is used, and the class
// words of the 16 uniques (2 bits per centre) leave as one 32-byte store.  Only classes are produced: the aligner runs
// when (and if) a centre's round comes.
struct ScrIn { uint4 c0, c1; int Lr, nh; uint32_t rd; bool lk; };

// CORES: waves per SIMD the kernel is compiled for.  4 (123 registers) when it has the device to itself; 6 (80 registers, a few
// spilled) for the prefetch compares that run BESIDE the persistent tail, which holds half of every CU's registers: the screen's
// throughput is its resident blocks (every block waits for its rows most of the time), and three waves per SIMD fit into the
// other half where two of the wide build do (profiles/r07c: 485 us beside the tail against 246 us alone).
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k2_screen_multi(Eng2 E) {
  const Ctl2 *ctl = E.ctl;
  const int nb = ctl->nbatch;
  if (ctl->state != 0 || nb == 0) return;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
  const SampleDev &S = E.S;
  const Cache2 &C = E.C;
  const ScreenParams sp = E.sp;
  uint2 *tab = (uint2 *)s_mem;                               // [1024]
  uint16_t *cord = (uint16_t *)(tab + NKMER);                // [KB_MAX][LK]
  int32_t *thr = (int32_t *)(cord + (size_t)KB_MAX * S.LK);  // [maxlen + 2] kdist > cutoff <=> dot < thr[d]
  __shared__ int cL[KB_MAX], cC[KB_MAX];
  __shared__ uint32_t cR[KB_MAX];
  const int tid = threadIdx.x;
  for (int i = tid; i < NKMER; i += 256) tab[i] = C.tab8[i];
  {
    const int n16 = (nb * S.LK * 2 + 15) / 16;
    const uint4 *src = (const uint4 *)C.ord;
    for (int i = tid; i < n16; i += 256) ((uint4 *)cord)[i] = src[i];
  }
  for (int i = tid; i < S.maxlen + 2; i += 256) thr[i] = E.thresh[i];
  if (tid < KB_MAX) { cL[tid] = ctl->blen[tid]; cC[tid] = ctl->bcentre[tid]; cR[tid] = ctl->breads[tid]; }
  __syncthreads();
  uint16_t *bcls = C.bcls + (size_t)ctl->bbuf * C.Npad;
  const int lane = tid & 63, sub = tid & 15, g = lane >> 4;
  const int gwave = blockIdx.x * 4 + (tid >> 6), nwaves = gridDim.x * 4;
  const int nchunk = (S.maxlen - KMER_SIZE + 1 + 7) >> 3;
  const uint4 pad4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  const bool heavy_on = sp.use_kmers && S.HMAX > 0;
  // after the unordered pass every lane of a group holds all eight overlaps: lane k (< 8) then takes the decisions of centre k
  const int kc = sub & 7;
  const bool kvalid = sub < 8 && kc < nb;
  const int Lc = cL[kc], Cc = cC[kc];
  const uint32_t Rc = cR[kc];
  auto load = [&](int r) __attribute__((always_inline)) {
    ScrIn in;
    const bool on = r < S.N;
    const uint4 *row = (const uint4 *)(S.kord + (size_t)(on ? r : 0) * S.LK);
    in.c0 = (on && sp.use_kmers && sub < nchunk) ? row[sub] : pad4;
    in.c1 = (on && sp.use_kmers && sub + 16 < nchunk) ? row[sub + 16] : pad4;
    in.Lr = on ? S.len[r] : 0;
    in.rd = on ? S.reads[r] : 0u;
    in.lk = on && E.greedy && E.P.lock[r];
    in.nh = (on && heavy_on) ? S.nheavy[r] : 0;
    return in;
  };
  auto one = [&](int r, const ScrIn in) __attribute__((always_inline)) -> uint32_t {   // class word (2 bits per centre) of unique r, valid in the group's lane 0
    if (r >= S.N) return 0u;
    const uint4 *row = (const uint4 *)(S.kord + (size_t)r * S.LK);
    // ---- pass 1: unordered overlap with every centre of the batch ----
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;                 // 16-bit fields: centres (0,2) (1,3) (4,6) (5,7)
    if (sp.use_kmers) {
      for (int ch = sub, j = 0; ch < nchunk; ch += 16, j++) {
        const uint4 v = j == 0 ? in.c0 : (j == 1 ? in.c1 : row[ch]);
        uint32_t ax = 0, ay = 0;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
          for (int hlf = 0; hlf < 2; hlf++) {
            const uint32_t x = hlf ? (w[e] >> 16) : (w[e] & 0xFFFFu);
            const uint2 t = tab[x & 1023u];
            const uint32_t rk = x >> 10;                               // rank 0..63; want rank < count  <=>  count + 0x7F - rank >= 0x80
            const uint32_t rr = gcn_bcast_byte0(rk);                   // (bytes stay within 0x40 .. 0xBE: no borrow between them)
            ax += ((t.x - rr) >> 7) & 0x01010101u;                     // byte k: rank < min(count_k, 63)
            ay += ((t.y - rr) >> 7) & 0x01010101u;
          }
        }
        w0 += ax & 0x00FF00FFu; w1 += (ax >> 8) & 0x00FF00FFu;
        w2 += ay & 0x00FF00FFu; w3 += (ay >> 8) & 0x00FF00FFu;
      }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        w0 += __shfl_xor(w0, o, 16); w1 += __shfl_xor(w1, o, 16);
        w2 += __shfl_xor(w2, o, 16); w3 += __shfl_xor(w3, o, 16);
      }
    }
    // ---- lane k of the group: the dispatch of raw_align against centre k (nwalign_endsfree.cpp:10-73) ----
    uint32_t dot = (((kc & 4) ? ((kc & 1) ? w3 : w2) : ((kc & 1) ? w1 : w0)) >> ((kc & 2) ? 16 : 0)) & 0xFFFFu;
    const bool skipped = E.greedy && (in.rd > Rc || (in.lk && r != Cc));
    if (in.nh > 0 && kvalid && !skipped) {                   // k-mers occurring > 63 times: exact correction
      for (int hh = 0; hh < in.nh; hh++) {
        const uint32_t e = S.heavy[(size_t)r * S.HMAX + hh], cr = e >> 16, cc = C.full[(size_t)kc * NKMER + (e & 1023u)];
        const uint32_t m = cr < cc ? cr : cc;
        if (m > RANK_SAT) dot += m - RANK_SAT;
      }
      dot &= 0xFFFFu;                                        // the reference accumulates in uint16_t (kmers.cpp:16,34,69)
    }
    const int d = (Lc < in.Lr ? Lc : in.Lr) - KMER_SIZE + 1;
    const bool shroud = sp.use_kmers && (int)dot < thr[d > 0 ? d : 0];   // kdist > kdist_cutoff
    const bool gl_ok = sp.gapless && sp.use_kmers && (sp.sse >= 1 || in.Lr == Lc);
    bool gapless = sp.band == 0;
    // ---- pass 2 (surviving pairs only): ordered overlap over the first d positions, the whole group per pair ----
    const bool need2 = kvalid && !skipped && !shroud && !gapless && gl_ok;
    uint32_t mask = (uint32_t)((__ballot(need2) >> (16 * g)) & 0xFFull);
    while (mask) {
      const int kk = __builtin_ctz(mask);
      mask &= mask - 1;
      const int dk = __shfl(d, kk, 16);                      // (lane kk of this lane group)
      const uint32_t dotk = (uint32_t)__shfl((int)dot, kk, 16);
      const uint16_t *ck = cord + (size_t)kk * S.LK;
      uint32_t ord = 0;
      for (int ch = sub, j = 0; ch < nchunk; ch += 16, j++) {
        const uint4 v = j == 0 ? in.c0 : (j == 1 ? in.c1 : row[ch]);
        const uint4 ck4 = ((const uint4 *)ck)[ch];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w}, cw[4] = {ck4.x, ck4.y, ck4.z, ck4.w};
        const int i0 = ch << 3;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t t = (w[e] ^ cw[e]) & 0x03FF03FFu;   // k-mer ids only (rank bits masked off)
          ord += ((t & 0xFFFFu) == 0 && i0 + 2 * e < dk);
          ord += ((t >> 16) == 0 && i0 + 2 * e + 1 < dk);
        }
      }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) ord += __shfl_xor(ord, o, 16);
      if (kc == kk && sub < 8) gapless = (ord & 0xFFFFu) == dotk;      // kodist == kdist
    }
    uint32_t c = !kvalid ? 0u : (skipped ? (uint32_t)CLS_SKIP : (shroud ? (uint32_t)CLS_SHROUD : (gapless ? (uint32_t)CLS_GAPLESS : (uint32_t)CLS_NW)));
    uint32_t code = c << (2 * kc);
    code |= __shfl_xor(code, 1, 16); code |= __shfl_xor(code, 2, 16); code |= __shfl_xor(code, 4, 16);
    return code & 0xFFFFu;
  };
  const bool pre = sp.use_kmers && S.kbits != nullptr && S.kmult != nullptr && C.cbits != nullptr;
  if (!pre) {
    for (int base16 = gwave * 16; base16 < S.N; base16 += nwaves * 16) {
      // request everything the four uniques of this lane group need, then work through them
      const int r0 = base16 + 4 * g;
      const ScrIn i0 = load(r0), i1 = load(r0 + 1);
      unsigned long long clsacc = one(r0, i0);
      const ScrIn i2 = load(r0 + 2);
      clsacc |= (unsigned long long)one(r0 + 1, i1) << 16;
      const ScrIn i3 = load(r0 + 3);
      clsacc |= (unsigned long long)one(r0 + 2, i2) << 32;
      clsacc |= (unsigned long long)one(r0 + 3, i3) << 48;
      if (sub == 0 && r0 < S.N) *(unsigned long long *)(bcls + r0) = clsacc;
    }
    return;
  }
  // ---- with the presence bitmaps (SampleDev::kbits): two stages.  STAGE 1 reads 128 bytes per unique - lane `sub` of its group
  //      two of the 32 bitmap words - and bounds the overlap with every centre of the batch from above:
  //          sum_k min(a_k, b_k)  <=  |{k : a_k > 0 and b_k > 0}| + sum_k (a_k - 1)+  =  popcount(bits_r & bits_c) + kmult_r,
  //      so where that bound is below the threshold of kmers.cpp:47 the pair is shrouded whatever its exact overlap is (and a greedy
  //      skip is decided from reads and lock alone): most uniques get their whole class word here, 55 instructions per lane
  //      instead of 180 + 16 gathers.  The uniques with a centre the bound does not settle go to a list in LDS.  STAGE 2 takes that
  //      list through the exact code above, every lane group busy.  The class words are the same either way: stage 1 only ever
  //      says "shrouded" where the exact overlap, being no larger than the bound, says so too.
  constexpr int SCR_ITERS = 8, SCR_LIST = SCR_ITERS * 64;      // uniques a block looks at between two flushes of its list
  __shared__ int s_list[SCR_LIST];
  __shared__ int s_nlist;
  __shared__ uint2 s_cbits[KB_MAX][16];
  if (tid < KB_MAX * 16) s_cbits[tid >> 4][tid & 15] = ((const uint2 *)C.cbits)[tid];
  if (tid == 0) s_nlist = 0;
  __syncthreads();
  const int wv = tid >> 6, grp = tid >> 4;
  auto flush = [&]() __attribute__((always_inline)) {
    __syncthreads();
    const int n = s_nlist;
    for (int q = grp; q < ((n + 15) & ~15); q += 32) {          // (wave-uniform trip count: the exact code ballots over the wave)
      // (a group without an entry of its own walks the list's first unique again and stores nothing: every lane of the wave goes
      //  through the same cross-lane operations)
      const bool ha = q < n, hb = q + 16 < n;
      const int ra = ha ? s_list[q] : s_list[0], rb = hb ? s_list[q + 16] : s_list[0];
      const ScrIn ia = load(ra), ib = load(rb);
      const uint32_t ca = one(ra, ia), cb = one(rb, ib);
      if (sub == 0 && ha) bcls[ra] = (uint16_t)ca;
      if (sub == 0 && hb) bcls[rb] = (uint16_t)cb;
    }
    __syncthreads();
    if (tid == 0) { if (n && E.fast_ctl) atomicAdd(&E.fast_ctl[3], (unsigned long long)n); s_nlist = 0; }   // (statistics: uniques that needed stage 2)
    __syncthreads();
  };
  int iter = 0;
  for (long long bb = (long long)blockIdx.x * 64; bb < S.N; bb += (long long)gridDim.x * 64, iter++) {
    const int r0 = (int)bb + 16 * wv + 4 * g;
    uint2 bw[4];
    int Ls[4], kms[4];
    uint32_t rds[4];
    bool lks[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + u;
      const bool on = r < S.N;
      bw[u] = on ? ((const uint2 *)(S.kbits + (size_t)r * 32))[sub] : make_uint2(0u, 0u);
      Ls[u] = on ? S.len[r] : 0;
      rds[u] = on ? S.reads[r] : 0u;
      lks[u] = on && E.greedy && E.P.lock[r];
      kms[u] = on ? (int)S.kmult[r] : 0;
    }
    unsigned long long clsacc = 0;
    uint32_t survs = 0;                                         // bit u: unique r0 + u goes to stage 2
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + u;
      uint32_t p[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const uint2 c0 = s_cbits[2 * kk][sub], c1 = s_cbits[2 * kk + 1][sub];
        const uint32_t n0 = (uint32_t)__popc(bw[u].x & c0.x) + (uint32_t)__popc(bw[u].y & c0.y);
        const uint32_t n1 = (uint32_t)__popc(bw[u].x & c1.x) + (uint32_t)__popc(bw[u].y & c1.y);
        p[kk] = n0 | (n1 << 16);
      }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        p[0] += __shfl_xor(p[0], o, 16); p[1] += __shfl_xor(p[1], o, 16);
        p[2] += __shfl_xor(p[2], o, 16); p[3] += __shfl_xor(p[3], o, 16);
      }
      const uint32_t pk = (kc & 4) ? ((kc & 2) ? p[3] : p[2]) : ((kc & 2) ? p[1] : p[0]);
      const int bound = (int)((pk >> ((kc & 1) ? 16 : 0)) & 0xFFFFu) + kms[u];
      const int d = (Lc < Ls[u] ? Lc : Ls[u]) - KMER_SIZE + 1;
      const bool skipped = E.greedy && (rds[u] > Rc || (lks[u] && r != Cc));
      const bool open = kvalid && r < S.N && !skipped && !(bound < thr[d > 0 ? d : 0]);
      uint32_t code = (kvalid ? (skipped ? (uint32_t)CLS_SKIP : (uint32_t)CLS_SHROUD) : 0u) << (2 * kc);
      code |= __shfl_xor(code, 1, 16); code |= __shfl_xor(code, 2, 16); code |= __shfl_xor(code, 4, 16);
      const bool any_open = ((__ballot(open) >> (16 * g)) & 0xFFFFull) != 0ull;
      if (any_open) survs |= 1u << u;
      clsacc |= (unsigned long long)(code & 0xFFFFu) << (16 * u);
    }
    if (sub == 0 && r0 < S.N) {
      if (survs == 0) *(unsigned long long *)(bcls + r0) = clsacc;   // (the row is padded: Npad >= N + 15)
      else {
        int at = atomicAdd(&s_nlist, __popc(survs));
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (r0 + u >= S.N) continue;
          if ((survs >> u) & 1u) s_list[at++] = r0 + u;
          else bcls[r0 + u] = (uint16_t)(clsacc >> (16 * u));
        }
      }
    }
    if ((iter % SCR_ITERS) == SCR_ITERS - 1) flush();
  }
  flush();
}

// post-hoc partition p-value inputs (error.cpp:101-119) from the v2 store
__global__ __launch_bounds__(256) void k2_posthoc(Eng2 E, const int32_t *__restrict__ cluster_of_centre, int32_t *__restrict__ out_ji,
                                                  double *__restrict__ out_lam, int32_t *__restrict__ nout, int cap) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= E.S.N) return;
  const int j = cluster_of_centre[r];
  if (j < 0) return;
  if (j != 0) {
    const int k = atomicAdd(nout, 1);
    if (k < cap) { out_ji[2 * k] = j; out_ji[2 * k + 1] = 0; out_lam[k] = E.T.lam0[r]; }
  }
  const int i1 = E.T.i1[r];
  if (i1 >= 0 && i1 != j) {
    const int k = atomicAdd(nout, 1);
    if (k < cap) { out_ji[2 * k] = j; out_ji[2 * k + 1] = i1; out_lam[k] = E.T.lam1[r]; }
  }
  for (int b = E.T.head[r], hops = 0; b >= 0 && hops < (1 << 22); b = E.T.blk[b].next, hops++) {
    const CompBlk *cb = E.T.blk + b;
    for (int q = 0; q < cb->cnt; q++) {
      const int i = cb->i[q];
      if (i == j) continue;
      const int k = atomicAdd(nout, 1);
      if (k < cap) { out_ji[2 * k] = j; out_ji[2 * k + 1] = i; out_lam[k] = cb->lam[q]; }
    }
  }
}

// ---- launch wrappers ------------------------------------------------------------------------------------------------------
void launch2_store0(const Eng2 &E, const double *d_lam, const uint32_t *d_ham, const uint8_t *d_cls, const int32_t *d_round_counters,
                    hipStream_t st) {
  const int grid = std::min((E.S.N + 255) / 256, 2048);
  hipLaunchKernelGGL(k2_store0, dim3(grid), dim3(256), 0, st, E, d_lam, d_ham, d_cls, d_round_counters);
}
void launch2_batch_lists(const Eng2 &E, hipStream_t st) {
  const int grid = (E.S.N + 256 * LISTS_PER_THREAD - 1) / (256 * LISTS_PER_THREAD);
  hipLaunchKernelGGL(k2_batch_lists, dim3(grid), dim3(256), 0, st, E);
}
void launch2_pf_gate(const Eng2 &E, int k, const int32_t *h_quit, int32_t *h_result, hipStream_t st) {
  hipLaunchKernelGGL(k2_pf_gate, dim3(1), dim3(64), 0, st, E, k, h_quit, h_result);
}
void launch2_pf_tables(const Eng2 &E, hipStream_t st) { hipLaunchKernelGGL(k2_pf_tables, dim3(1), dim3(1024), 0, st, E); }
void launch2_pf_done(const Eng2 &E, hipStream_t st) { hipLaunchKernelGGL(k2_pf_done, dim3(1), dim3(64), 0, st, E); }
void launch2_screen_multi(const Eng2 &E, hipStream_t st, bool beside_tail) {
  const size_t lds = (size_t)NKMER * 8 + (size_t)KB_MAX * E.S.LK * 2 + (size_t)(E.S.maxlen + 2) * 4 + 16;
  static size_t attr_set[64] = {0};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (lds > attr_set[dev_ & 63]) {
    (void)hipFuncSetAttribute((const void *)k2_screen_multi<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)k2_screen_multi<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev_ & 63] = lds;
  }
  const int grid = std::min((E.S.N + 63) / 64, 2048);
  if (beside_tail) hipLaunchKernelGGL(k2_screen_multi<6>, dim3(grid), dim3(256), lds, st, E);
  else hipLaunchKernelGGL(k2_screen_multi<4>, dim3(grid), dim3(256), lds, st, E);
}
void launch2_shuffle(const Eng2 &E, int level, bool store, hipStream_t st) {
  const int grid = std::min((E.S.N + 255) / 256, (int)E.grid_shuffle);   // one device atomic per counter per block
  if (store) hipLaunchKernelGGL(k2_shuffle<true>, dim3(grid), dim3(256), 0, st, E, level);
  else hipLaunchKernelGGL(k2_shuffle<false>, dim3(grid), dim3(256), 0, st, E, level);
}
void launch2_eval(const Eng2 &E, int nlev, uint32_t init_reads, hipStream_t st) {
  BudKey init{1.0, init_reads};
  const int grid = std::min((E.S.N + 255) / 256, std::min(8192, (int)E.grid_pupdate));   // (Eng2::partial holds 8192 key pairs)
  hipLaunchKernelGGL(k2_pupdate, dim3(grid), dim3(256), 0, st, E, nlev, init, (BudKey *)E.partial);
  hipLaunchKernelGGL(k2_birth, dim3(1), dim3(1024), 0, st, E, nlev, init, (const BudKey *)E.partial, grid);
}
void launch2_host_birth(const Eng2 &E, int raw, int from, hipStream_t st) {
  hipLaunchKernelGGL(k2_host_birth, dim3(1), dim3(1024), 0, st, E, raw, from);
}
void launch2_resume(const Eng2 &E, hipStream_t st, bool keep_list, bool compare_done) {
  hipLaunchKernelGGL(k2_resume, dim3(1), dim3(1), 0, st, E, keep_list ? 1 : 0, compare_done ? 1 : 0);
}
void launch2_posthoc(const Eng2 &E, const int32_t *d_cluster_of_centre, int32_t *d_out_ji, double *d_out_lam, int32_t *d_nout,
                     int cap, hipStream_t st) {
  hipLaunchKernelGGL(k2_posthoc, dim3((E.S.N + 255) / 256), dim3(256), 0, st, E, d_cluster_of_centre, d_out_ji, d_out_lam, d_nout, cap);
}
#endif  // D2_TAIL_TU
