// rounds3.inc.hip — the persistent round tail of engine v2 (engine.h); included by kernels.hip inside namespace d2
// behind rounds2.inc.hip, whose phase bodies it runs.
//
// A round of run_dada (Rmain.cpp:316-331) behind its compare is  b_shuffle2 x (1..MAX_SHUFFLE)  ->  b_p_update  ->  b_bud  ->
// birth.  As a chain of launches that is 4-8 kernels of 13-20 us each whose work is worth 2-4 us of HBM traffic: they are bound
// by grid ramp-up, control-block loads, dependent gathers and the guarded launches that find nothing to do.  k3_tail runs the
// same phase bodies inside ONE launch of G co-resident blocks (1024 threads, at most one per CU):
//
//   round:  commit + shuffle call 0 | barrier | shuffle call 1 + evaluation as if it moved nothing | barrier | ... until a call
//           moves nothing: ITS barrier's LAST ARRIVER runs the serial end of the round (birth_body: arg-min, ties, decision,
//           birth, plan) before it releases the others | the same block then publishes the round's result block to the host
//           (a round whose commit call moves nothing, or whose calls MAX_SHUFFLE cuts short: evaluation | barrier as a phase
//           of its own, as in round 4)
//
// and goes on to the next round for as long as the new centre's comparisons are cached.  It leaves the launch when a compare
// is due (the launches of the batch compare follow in the stream, then the next k3_tail), when the device halts (the same
// halts as the launch chains, minus SHUFFLE_MORE and NEED_COMPARE which cannot occur), when the round's movers did not fit the
// result block (pause) or when the host's ring of result blocks would not take another one.
//
// Inter-block visibility (MI355X: eight XCDs with private L2s, per-CU L1s never refreshed by other CUs' stores) follows
// /opt/skills/guides: every block drains its stores, one lane releases at agent scope and arrives on a monotonic counter; the
// last arriver acquires, works, releases and bumps the generation word; the others poll it relaxed, acquire once, and only
// then read.  Every spin is bounded (GRID_WAIT_S): a launch that cannot become co-resident fails loudly instead of hanging.
// Nothing here depends on which XCD or in which order blocks run.

// bound of a barrier wait: Eng2::grid_wait_ticks (100 MHz ticks; the host scales it with the sample - 2 s + 1 s per 10^6 uniques -
// the slowest phase of 10^6 uniques being tens of microseconds)

template <int BS>
struct TailLds {
  ShufLds<BS> sh;                                       // (both at once: a shuffle call after the commit's also evaluates, shuffle_body<.., SPEC>)
  PupdLds<BS> pu;
  int last, ok;
  int xtmp[8];
  int xcc, xn, ngrp;                                    // XCD-hierarchical barriers: this block's XCC, the launch's blocks on it, XCCs in use
};
// the XCD-hierarchical barriers' state of a launch (engine.h, PSync): which of them a barrier call uses
struct XBar { int on; uint32_t epoch; };

// code: TAIL_FAIL_ENTRY = the launch's entry barrier (not every block became resident: nothing has changed yet, the host sends the
// run to the launch chains), TAIL_FAIL_LATE = a barrier inside a round (the run is lost)
enum : uint32_t { TAIL_FAIL_ENTRY = 1u, TAIL_FAIL_LATE = 2u };
static __device__ __forceinline__ void tail_fail(const Eng2 &E, uint32_t code) {
  // a barrier gave up: halt the device-side progression so that the launches queued behind this one do nothing
  gcn_store_agent(&E.psync->fail, code);
  E.ctl->state = 1; E.ctl->halt = H2_FAIL;
}

// Grid barrier with a serial section: every block arrives; the last one runs `serial` (the whole block, block barriers allowed)
// between its acquire and the release of the others.  Returns false when the wait ran into its bound (every block then leaves).
//
// xb (XBar::on): the XCD-hierarchical form (MI355X_MICROARCH.md, barrier-xcd; tools/micro/grid_barrier.hip measured both on this
// device: 7.1 -> 3.7 us at 245 blocks, 3.4 -> 2.5 at 96, and 1.8 -> 2.0 at 25, which is why small grids keep the flat one).  The
// blocks of one XCC share its L2: every wave has drained its stores into it before its block arrives on the XCC's counter, so ONE
// write-back by the XCC's last arriver covers them all (245 blocks each writing an L2 back that 30 others are writing back too
// was most of the flat barrier's time); that block arrives on the top counter, the last one there runs `serial` and then bumps
// every XCC's generation word.  Every block acquires behind its wait, as before.
template <int BS, typename F>
static __device__ __forceinline__ bool grid_sync(const Eng2 &E, TailLds<BS> &L, uint32_t &epoch, int G, F &&serial, uint32_t fail_code = TAIL_FAIL_LATE,
                                                 XBar *xb = nullptr) {
  PSync *ps = E.psync;
  gcn_drain_stores();                                   // every wave: its own stores have left the CU
  __syncthreads();
  const bool xmode = xb != nullptr && xb->on != 0 && G > 1;
  if (threadIdx.x == 0 && xmode) {
    int last = 0, ok = 1;
    const uint32_t e1 = xb->epoch + 1u;
    const int x = L.xcc;
    const uint32_t t = gcn_add_agent(&ps->xarr[x][0], 1u);
    if (t == e1 * (uint32_t)L.xn - 1u) {                // the XCC's last arriver: its L2 holds the stores of all the XCC's blocks
      gcn_release_agent();
      const uint32_t t2 = gcn_add_agent(&ps->top, 1u);
      last = t2 == e1 * (uint32_t)L.ngrp - 1u;
    }
    if (!last) {
      const unsigned long long t0 = gcn_wall_clock();
      for (unsigned n = 1;; n++) {
        if ((int32_t)(gcn_load_agent(&ps->xgen[x][0]) - e1) >= 0) break;
        gcn_poll_pause();
        if ((n & 63u) == 0u && (gcn_load_agent(&ps->fail) != 0u || gcn_wall_clock() - t0 > E.grid_wait_ticks)) { ok = 0; break; }
      }
      if (!ok) tail_fail(E, fail_code);
    }
    gcn_acquire_agent();
    L.last = last; L.ok = ok;
  }
  if (threadIdx.x == 0 && !xmode) {
    int last = 1, ok = 1;
    if (G > 1) {
      const bool tm = E.ktime != nullptr && blockIdx.x == 0;
      const unsigned long long tr0 = tm ? gcn_wall_clock() : 0ull;
      gcn_release_agent();
      if (tm) s_ktime[KT_RELEASE] += gcn_wall_clock() - tr0;
      const uint32_t t = gcn_add_agent(&ps->arrive, 1u);
      last = t == (epoch + 1u) * (uint32_t)G - 1u;
      if (!last) {
        const unsigned long long t0 = gcn_wall_clock();
        for (unsigned n = 1;; n++) {
          if ((int32_t)(gcn_load_agent(&ps->gen) - (epoch + 1u)) >= 0) break;
          gcn_poll_pause();
          if ((n & 63u) == 0u && (gcn_load_agent(&ps->fail) != 0u || gcn_wall_clock() - t0 > E.grid_wait_ticks)) { ok = 0; break; }
        }
        if (!ok) tail_fail(E, fail_code);
      } else if (fail_code == TAIL_FAIL_ENTRY && gcn_load_agent(&ps->fail) != 0u) ok = 0;   // (the others gave up waiting for this block: leave with them)
      gcn_acquire_agent();
    } else {
      // one block: no other block to wait for, but the phase that follows reads through this CU's L1 what device-scope atomics
      // of the phase before left in L2 (reads deltas, counters): the L1 lines have to go, as at a kernel boundary
      gcn_release_agent();
      gcn_acquire_agent();
    }
    L.last = last; L.ok = ok;
  }
  __syncthreads();
  if (!L.ok) return false;
  if (L.last) {
    // `serial` gets the release as a callable: a section with work that nobody waits for (the plan of the next prefetch compare,
    // birth_body) lets the other blocks go BEFORE that work and does it while they are in their next phase; a section that does
    // not call it is released behind its last statement as before
    bool released = false;
    auto release = [&]() __attribute__((always_inline)) {
      gcn_drain_stores();
      __syncthreads();                                  // the whole block is through what the others will read
      if (xmode) {
        if (threadIdx.x == 0) gcn_release_agent();
        __syncthreads();
        if (threadIdx.x < 8) gcn_store_agent(&ps->xgen[threadIdx.x][0], xb->epoch + 1u);   // (the write-back is done: lanes of the same wave)
      } else if (G > 1) {
        if (threadIdx.x == 0) { gcn_release_agent(); gcn_store_agent(&ps->gen, epoch + 1u); }
      } else {
        // one block: its own waves read next what the serial section just wrote (the control block, the new centre's state)
        if (threadIdx.x == 0) { gcn_release_agent(); gcn_acquire_agent(); }
        __syncthreads();
      }
      released = true;
    };
    serial(release);
    if (!released) release();
  }
  if (xmode) xb->epoch++; else epoch++;
  return true;
}

// the launch's dynamic LDS: TailLds, and behind it the mirror's 2 x MIR_CAP words where Eng2::mirror_on says so
template <int BS>
constexpr size_t tail_lds_bytes(bool mirror) { return ((sizeof(TailLds<BS>) + 15) & ~(size_t)15) + (mirror ? (size_t)MIR_CAP * 8 : 0); }

// Fill the mirror from the global arrays (launch entry, behind the entry barrier's acquire): the same deal of uniques as every sweep
template <int BS>
static __device__ __forceinline__ void mir_fill(const Eng2 &E, uint32_t *mir) {
  constexpr int U = ShufLds<BS>::U;
  const int N = E.S.N;
  for (int grp = 0; (long long)grp * U * BS * gridDim.x < N; grp++) {
    int i1s[U], cls_[U];
    uint8_t lks[U];
    double ps_[U];
    unsigned long long sms[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = sweep_unique<BS, U>(grp, u);
      i1s[u] = -1; cls_[u] = 0; lks[u] = 0; ps_[u] = 1.0; sms[u] = 0ull;
      if (r < N) { i1s[u] = E.T.i1[r]; cls_[u] = E.P.clust_of[r]; lks[u] = E.P.lock[r]; ps_[u] = E.P.p[r]; sms[u] = E.T.smask[r]; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int slot = (grp * U + u) * BS + (int)threadIdx.x;
      mir[slot] = sweep_unique<BS, U>(grp, u) < N ? ((uint32_t)cls_[u] & MIR_CL) | (i1s[u] >= 0 ? MIR_I1 : 0u) | (lks[u] ? MIR_LOCK : 0u) | (ps_[u] != 1.0 ? MIR_PNE1 : 0u) : MIR_INVALID;
      mir[MIR_CAP + slot] = mir_fold(sms[u]);
    }
  }
  __syncthreads();
}

// test knob (Eng2::mirror_on == 2): the mirror against the arrays it shadows, at the end of a round - a difference is an internal error
template <int BS>
static __device__ __forceinline__ void mir_verify(const Eng2 &E, const uint32_t *mir) {
  constexpr int U = ShufLds<BS>::U;
  const int N = E.S.N;
  __syncthreads();
  for (int grp = 0; (long long)grp * U * BS * gridDim.x < N; grp++)
    for (int u = 0; u < U; u++) {
      const int r = sweep_unique<BS, U>(grp, u);
      if (r >= N) { if (mir[(grp * U + u) * BS + (int)threadIdx.x] != MIR_INVALID) atomicOr(E.P.err_flag, 16); continue; }
      const uint32_t want = ((uint32_t)E.P.clust_of[r] & MIR_CL) | (E.T.i1[r] >= 0 ? MIR_I1 : 0u) | (E.P.lock[r] ? MIR_LOCK : 0u) | (E.P.p[r] != 1.0 ? MIR_PNE1 : 0u);
      const int slot = (grp * U + u) * BS + (int)threadIdx.x;
      if (mir[slot] != want || mir[MIR_CAP + slot] != mir_fold(E.T.smask[r])) atomicOr(E.P.err_flag, 16);
    }
}

template <int BS>
__global__ __launch_bounds__(BS, BS == 512 ? 2 : 4) void k3_tail(Eng2 E, BudKey init, int first, int ordinal) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn3[];
  TailLds<BS> &L = *(TailLds<BS> *)s_dyn3;
  uint32_t *const mir = E.mirror_on ? (uint32_t *)(s_dyn3 + tail_lds_bytes<BS>(false)) : nullptr;
  gcn_raise_priority();                                  // (its waves are few and wait most of the time: when they can issue, they go first)
  Ctl2 *ctl = E.ctl;
  const int G = (int)gridDim.x;
  const bool timer = E.ktime != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tk = timer ? gcn_wall_clock() : 0ull;
#define KT_LAP(SLOT) do { if (timer) { const unsigned long long now_ = gcn_wall_clock(); s_ktime[SLOT] += now_ - tk; tk = now_; } } while (0)
  // the block's phase clocks of this launch collect in LDS (rounds2.inc.hip, D2_KT) and are added to Eng2::ktime once per round
  if (E.ktime) { for (int k = threadIdx.x; k < KT_N; k += BS) s_ktime[k] = 0ull; __syncthreads(); }
#define KT_FLUSH() do { if (E.ktime) { __syncthreads(); for (int k_ = threadIdx.x; k_ < KT_N; k_ += BS) if (k_ != KT_SUB_LAST && k_ != KT_SUB_LAST + 1) { const unsigned long long v_ = s_ktime[k_]; if (v_) { atomicAdd(&E.ktime[k_], v_); s_ktime[k_] = 0ull; } } } } while (0)
  if (ctl->state != 0) {                                 // halted before the launch (written by an earlier kernel: every block sees it)
    if (blockIdx.x == 0 && threadIdx.x == 0) *E.hexit = ordinal;
    return;
  }
  if (ordinal == E.fail_ordinal) {                       // test knob (DADA2HIP_V3_FAIL_ENTRY): what a launch that cannot become co-resident leaves behind
    if (blockIdx.x == 0 && threadIdx.x == 0) tail_fail(E, TAIL_FAIL_ENTRY);
    return;
  }
  uint32_t epoch = E.psync->gen;                         // barriers completed by earlier launches (arrive == gen * G between launches)
  XBar xb{E.xbar && G > 1 ? 1 : 0, 0u};
  if (xb.on && threadIdx.x == 0) { L.xcc = gcn_xcc_id(); gcn_add_agent(&E.psync->xcount[L.xcc][0], 1u); }   // (counted before the entry barrier)
  // entry: every block is resident before any state changes, and ONE block decides whether the host's ring takes the first
  // result block of this launch (the host may still be reading the slot it goes to)
  if (!grid_sync<BS>(E, L, epoch, G, [&](auto &&) {
        if (xb.on && threadIdx.x < 8) {
          // every block of the launch has counted itself on its XCC: the launch's group sizes, and a clean slate for its barriers
          PSync *ps = E.psync;
          const uint32_t n = gcn_load_agent(&ps->xcount[threadIdx.x][0]);
          gcn_store_agent(&ps->xn[threadIdx.x][0], n);
          gcn_store_agent(&ps->xcount[threadIdx.x][0], 0u);
          gcn_store_agent(&ps->xarr[threadIdx.x][0], 0u);
          gcn_store_agent(&ps->xgen[threadIdx.x][0], 0u);
          L.xtmp[threadIdx.x] = (int)n;
        }
        if (xb.on) {
          __syncthreads();
          if (threadIdx.x == 0) {
            uint32_t ng = 0;
            for (int g = 0; g < 8; g++) ng += L.xtmp[g] != 0;
            gcn_store_agent(&E.psync->ngrp, ng);
            gcn_store_agent(&E.psync->top, 0u);
          }
        }
        if (threadIdx.x == 0) {
          const int seq = ctl->pub_seq + 1;
          int ex = 0;
          if (seq - ctl->hcons_seen > E.ring_limit) {
            const int hc = gcn_load_system((const int32_t *)E.hcons);
            ctl->hcons_seen = hc;
            if (seq - hc > E.ring_limit) ex = 1;
          }
          if (ctl->pf_wait) {
            // the coming round's centre sits in a prefetched batch whose compare was still running when the last launch left:
            // it has to be there before any block reads the batch's rows (the host orders the launch behind it when it can)
            const uint32_t seq = (uint32_t)ctl->pf_wait;
            uint32_t done = gcn_load_agent(&E.pfsync->done);
            const unsigned long long t0 = gcn_wall_clock();
            while ((int32_t)(done - seq) < 0 && gcn_wall_clock() - t0 < E.pf_wait_ticks) { gcn_poll_pause(); done = gcn_load_agent(&E.pfsync->done); }
            if ((int32_t)(done - seq) < 0) ex = 1; else ctl->pf_wait = 0;
          }
          ctl->kexit = ex;
          ctl->need_compare = 0;                          // the compare of the coming round, if it needed one, ran in front of this launch
          if (ex) {
            // ... and this launch leaves without running the round: the NEXT launch's compare kernels must not run the same batch
            // again (k2_batch_lists would append every pair to the work lists a second time - past their end, for a long list).
            // What the compare wrote stays valid until a round runs; the round's block still reports the pairs (Ctl2::nalign_ran)
            if (ctl->nalign > 0) ctl->nalign_ran = ctl->nalign;
            ctl->nbatch = 0; ctl->nalign = 0;
          }
        }
      }, TAIL_FAIL_ENTRY))
    return;
  if (ctl->kexit) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *E.hexit = ordinal;
    return;
  }
  if (xb.on) {
    if (threadIdx.x == 0) { L.xn = (int)gcn_load_agent(&E.psync->xn[L.xcc][0]); L.ngrp = (int)gcn_load_agent(&E.psync->ngrp); }
    __syncthreads();
  }
  if (mir) mir_fill<BS>(E, mir);
  KT_LAP(KT_LAUNCH);
  for (int rnd = 0;; rnd++) {
    const int ring = ctl->pub_seq % RING2;
    Round2Out *out = E.dblk + ring;
    int level = 0;
    const bool shuffles = !(first && rnd == 0);
    if (shuffles) {
      // ---- commit of the round's cached comparisons + b_shuffle2 until a call moves nothing (Rmain.cpp:320-325) ----
      shuffle_body<true, BS>(E, L.sh, 0, 0, E.movers, out, nullptr, BudKey{1.0, 0u}, nullptr, mir);
      KT_LAP(KT_S0);
      if (!grid_sync<BS>(E, L, epoch, G, [](auto &&) {}, TAIL_FAIL_LATE, &xb)) return;
      KT_LAP(KT_S0_BAR);
      level = 1;
    }
    int moved = shuffles ? out->cnt[0] : 0;
    for (;;) {
      const bool more = shuffles && level < E.max_shuffle && out->cnt[level - 1] > 0;   // another b_shuffle2 call is due
      int32_t *mv = E.movers + (size_t)level * 3 * (size_t)E.S.N;
      // (the attempt is worth its time when the call is likely to be the round's last: the call before it moved few uniques)
      const bool attempt = more && E.spec_eval && out->cnt[level - 1] <= E.spec_max_prev;
      if (more && !attempt) {
        shuffle_body<false, BS>(E, L.sh, level, moved, mv, out, nullptr, BudKey{1.0, 0u}, nullptr, mir);
        KT_LAP(KT_SL);
        if (!grid_sync<BS>(E, L, epoch, G, [](auto &&) {}, TAIL_FAIL_LATE, &xb)) return;
        KT_LAP(KT_SL_BAR);
        moved += out->cnt[level];
        level++;
        continue;
      }
      // The round's evaluation (b_p_update + the block minima of b_bud): riding on the shuffle call that is due, on the
      // assumption that the call moves nothing - or, when no call is due any more, as a phase of its own.  The last block to
      // arrive behind an evaluation that stands takes the round's decision.
      if (more) { shuffle_body<false, BS, true>(E, L.sh, level, moved, mv, out, &L.pu, init, (BudKey *)E.partial, mir); KT_LAP(KT_SL); }
      else { pupdate_body<BS>(E, L.pu, level, init, (BudKey *)E.partial, mir); KT_LAP(KT_P); }
      const int lv = level;
      if (!grid_sync<BS>(E, L, epoch, G, [&](auto &&release) {
            if (more && out->cnt[lv] != 0) {              // (every thread of the block reads the same settled word)
              if (threadIdx.x == 0) *E.sig_n = 0;         // the attempt is void: its listed candidates go
              return;
            }
            const int nlev = more ? lv + 1 : lv;
            const unsigned long long tb = E.ktime ? gcn_wall_clock() : 0ull;
            // the decision and everything the other blocks read next, THEN their release, then what only this block and the host
            // need (birth_body calls `go` in between)
            // (its 32 KB of table scratch: the sweeps' work lists, dead until this block's next phase)
            static_assert(sizeof(ShufLds<BS>) - offsetof(ShufLds<BS>, s_work) >= sizeof(uint32_t) * KB_MAX * NKMER && offsetof(ShufLds<BS>, s_work) % 16 == 0, "ShufLds lends 32 KB to birth_body");
            birth_body<true>(E, nlev, Chain2{nlev, true}, init, (const BudKey *)E.partial, G, ordinal, [&]() {
              if (threadIdx.x == 0) {
                ctl->pub_seq = ctl->pub_seq + 1;          // (the others find the NEXT round's block through it)
                if (E.ktime) { atomicAdd(&E.ktime[KT_BIRTH], gcn_wall_clock() - tb); atomicAdd(&E.ktime[KT_ROUNDS], 1ull); atomicAdd(&E.ktime[KT_LEVELS], (unsigned long long)nlev); }
              }
              release();
            }, (uint32_t *)L.sh.s_work);
          }, TAIL_FAIL_LATE, &xb))
        return;
      if (!more) { KT_LAP(KT_P_BAR); break; }
      KT_LAP(KT_SL_BAR);
      moved += out->cnt[level];
      level++;
      if (out->cnt[level - 1] == 0) {                     // the call moved nothing: the attempt stood, the round is decided
        if (E.greedy) spec_locks_flush<BS>(E, L.pu, out, mir); // (before the block can leave the launch: the locks belong to this round)
        break;
      }
    }
    const bool leave = ctl->kexit != 0;
    if (mir && out->birth_applied) {
      // the round's birth (apply_birth_and_plan, by the deciding block): the new centre sits in partition nclust - 1 now, unlocked -
      // the block that sweeps it brings its mirror word up to date before the coming commit reads it (shuffle_body synchronises
      // the block in front of its sweep)
      const int raw = ctl->centre;
      if (threadIdx.x == 0 && (raw >> 6) % G == (int)blockIdx.x) mir_set(mir, raw, MIR_CL | MIR_LOCK, (uint32_t)(ctl->nclust - 1) & MIR_CL);
    }
    if (E.mirror_on == 2) mir_verify<BS>(E, mir);
    KT_FLUSH();
    if (L.last) {
      // the block that took the decision publishes it while the others are already in the next round's first phase
      const unsigned long long tp = E.ktime ? gcn_wall_clock() : 0ull;
      publish_copy(E, out, ring, ctl->pub_seq);
      if (threadIdx.x == 0) {
        if (leave) *E.hexit = ordinal;
        if (E.ktime) atomicAdd(&E.ktime[KT_PUBLISH], gcn_wall_clock() - tp);
      }
    }
    if (leave) return;
  }
#undef KT_LAP
#undef KT_FLUSH
}

// Per-device facts of the persistent tail, established ONCE per device and process (std::call_once: dada2hip_run_multi's host threads
// come through here side by side): the CU count, the dynamic-LDS attribute of both instances, and how many blocks of each the
// device can hold at once.  cap: > 0 blocks, 0 = the kernel cannot be resident at all (its LDS does not fit this part, or the
// attribute was refused: the run then goes to the launch chains instead of failing at its first launch), -1 = the query failed
// (unknown: the bounded entry barrier is what stands between a grid that cannot be co-resident and a hang).
struct TailDev {
  std::once_flag once;
  int ncu = 64;
  int cap[2] = {-1, -1};            // [0]: 1024-thread blocks, [1]: 512
  bool mirror[2] = {false, false};  // the launch may carry the mirror's LDS behind TailLds (tail_lds_bytes(true) fits a workgroup of this part)
};
static TailDev &tail_dev(int device) {
  static TailDev devs[64];
  TailDev &d = devs[device & 63];
  std::call_once(d.once, [&]() {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (have_cur && cur != device) (void)hipSetDevice(device);
    hipDeviceProp_t prop;
    const bool have_prop = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0;
    if (have_prop) d.ncu = prop.multiProcessorCount;
    for (int w = 0; w < 2; w++) {
      const void *fn = w ? (const void *)k3_tail<512> : (const void *)k3_tail<1024>;
      const int bs = w ? 512 : 1024;
      const size_t lds_m = w ? tail_lds_bytes<512>(true) : tail_lds_bytes<1024>(true), lds_0 = w ? tail_lds_bytes<512>(false) : tail_lds_bytes<1024>(false);
      // with the mirror if the part's LDS takes it (160 KB per workgroup on gfx950), else without
      size_t lds = lds_m;
      d.mirror[w] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m) == hipSuccess;
      if (d.mirror[w]) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, bs, lds_m) == hipSuccess && per_cu <= 0) d.mirror[w] = false;   // (accepted but not resident)
      }
      if (!d.mirror[w]) {
        (void)hipGetLastError();
        lds = lds_0;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_0) != hipSuccess) { (void)hipGetLastError(); d.cap[w] = 0; continue; }
      }
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, bs, lds) != hipSuccess || !have_prop) { (void)hipGetLastError(); d.cap[w] = -1; continue; }
      d.cap[w] = std::max(0, per_cu) * d.ncu;
    }
    if (have_cur && cur != device) (void)hipSetDevice(cur);
  });
  return d;
}
int tail_resident_max(int device, int bs) { return tail_dev(device).cap[bs == 512 ? 1 : 0]; }
int tail_mirror_cap(int device, int bs) { return tail_dev(device).mirror[bs == 512 ? 1 : 0] ? MIR_CAP : 0; }
int tail_grid(int N, int device) {
  // one block (of 1024 or 512 threads) per CU at most (they have to be co-resident); 4096 uniques per block at 10^6 uniques
  const int want = (N + 4095) / 4096;
  return std::max(1, std::min(want, tail_dev(device).ncu));
}
void launch3_tail(const Eng2 &E, int grid, int bs, bool first, int ordinal, uint32_t init_reads, hipStream_t st) {
  BudKey init{1.0, init_reads};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  (void)tail_dev(dev_);                                    // (the dynamic-LDS attribute of both instances is set there)
  if (bs == 512) hipLaunchKernelGGL(k3_tail<512>, dim3(grid), dim3(512), tail_lds_bytes<512>(E.mirror_on != 0), st, E, init, first ? 1 : 0, ordinal);
  else hipLaunchKernelGGL(k3_tail<1024>, dim3(grid), dim3(1024), tail_lds_bytes<1024>(E.mirror_on != 0), st, E, init, first ? 1 : 0, ordinal);
}
