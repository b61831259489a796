// knobs.h — the library's run-time knobs: ONE place that reads the environment.
//
// Every DADA2HIP_* variable is parsed here, into one immutable snapshot, when a boundary call (include/dada2hip.h) is entered
// (knobs_reload(), called by the C-ABI entry points); everything below the boundary reads `knobs()`.  Nothing else in the
// library calls getenv.  The supported set is listed in include/dada2hip.h ("Environment"); all default to "off" / automatic,
// and none changes a result - they choose engines, kernel families and buffer sizes, which the parity tests sweep.
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

namespace d2 {

enum { NWK_AUTO = 0, NWK_LANE = 1, NWK_COOP = 2, NWK_WIDE = 3 };   // Knobs::nw_kernel

struct Knobs {
  // ---- which engine / kernel family (tests cross-check them) ----
  bool engine_classic = false;        // DADA2HIP_ENGINE=classic        round 1's engine: one centre per round, host round trip per decision
  int nw_kernel = 0;                  // DADA2HIP_NW_KERNEL=lane|coop|wide   (1 / 2 / 3; 0 = automatic)
  bool ad_homo = true;                // DADA2HIP_AD_HOMO=0             homopolymer-gap aligner back on the lane kernels + classic engine
  bool no_speculation = false;        // DADA2HIP_NO_SPECULATION        classic engine: the reference's plain loop
  bool no_autobirth = false;          // DADA2HIP_NO_AUTOBIRTH          classic engine: every bud decision on the host
  // ---- round engine v2 ----
  int v2_nbuf = 0;                    // DADA2HIP_V2_NBUF               cached batch buffers (0 = automatic: <= 64, <= 1/8 of the device)
  int v2_depth = 0;                   // DADA2HIP_V2_DEPTH              rounds / super-chains enqueued ahead (0 = automatic: 2)
  int v2_chain = 0;                   // DADA2HIP_V2_CHAIN              shuffle calls per launch chain (0 = automatic: 4)
  bool v2_graph = false;              // DADA2HIP_V2_GRAPH=1            hipGraph replay of the launch chains
  int v2_lite = -1;                   // DADA2HIP_V2_LITE=0|1           chains expected to hit the cache go out without the compare launches
  int v2_align = -1;                  // DADA2HIP_V2_ALIGN=commit|batch when the pairs of a batch are aligned (1 / 0; -1 = automatic)
  int v2_filter = -1;                 // DADA2HIP_V2_FILTER=0|1         later shuffle calls visit only the uniques the previous call can have unsettled
  int v2_grid_shuffle = 0;            // DADA2HIP_V2_GRID_SHUFFLE       block cap of k2_shuffle (0 = 2048)
  int v2_grid_pupdate = 0;            // DADA2HIP_V2_GRID_PUPDATE       block cap of k2_pupdate (0 = 1024)
  int v2_mov_inline = 0;              // DADA2HIP_V2_MOV_INLINE         movers that ride inline with a result block (tests: pauses / long lists)
  bool v2_tail_chain = false;         // DADA2HIP_V2_TAIL=chain         the round tail as launch chains instead of the persistent kernel
  int v3_grid = 0;                    // DADA2HIP_V3_GRID               blocks of the persistent launch (tests: several blocks on a small sample)
  int v3_ring = 0;                    // DADA2HIP_V3_RING               result blocks the device may be ahead of the host (tests: a host that lags)
  int v3_block = 0;                   // DADA2HIP_V3_BLOCK=512|1024     threads per block of the persistent tail (0 = automatic)
  int v3_overlap = -1;                // DADA2HIP_V3_OVERLAP=0|1        the next batch's compare under the persistent tail, on a second stream (-1 = automatic)
  int v3_pf_wait_us = -1;             // DADA2HIP_V3_PF_WAIT_US         how long the tail spins for a prefetched compare before it leaves the launch
  int v3_pf_early = -1;               // DADA2HIP_V3_PF_EARLY=n         (tuning) plan the next prefetch n positions into the batch before the newest (0-8)
  int v3_pf_gate_us = -1;             // DADA2HIP_V3_PF_GATE_US         how long the gate of a prefetch chain enqueued ahead waits for its plan (0 = no chain ahead: the host launches a compare when it sees the plan)
  int v3_xbar = -1;                   // DADA2HIP_V3_XBAR=0|1           XCD-hierarchical grid barriers inside a persistent launch (default: from 48 blocks on)
  int v3_spec = 1;                     // DADA2HIP_V3_SPEC=0|1           the round's evaluation rides on its shuffle calls (0: a phase of its own behind them, round 4's form)
  int v3_spec_max = -1;               // DADA2HIP_V3_SPEC_MAX=n         (tuning) ... only behind a call that moved at most n uniques
  int v3_pf_sync = 0;                 // DADA2HIP_V3_PF_SYNC=1          (measurement) every prefetch is waited for at the next serial end, the tail resident and idle
  int v3_pf_lowreg = -1;              // DADA2HIP_V3_PF_LOWREG=0|1      (tuning) prefetch screens on the 80-register build of the screen kernel (-1 = automatic: where the tail shares every CU)
  int replay_radix_min = 4096;        // DADA2HIP_REPLAY_RADIX_MIN=n    test knob: mover lists from n entries on are ordered by the radix sort of the host's replay (1 = always)
  int v3_slots = 3;                   // DADA2HIP_V3_SLOTS=n            persistent launches of this process side by side on one device (several samples in flight; 1 = their rounds take turns)
  int v3_lane = 0;                    // DADA2HIP_V3_LANE=1             the host's replay of moves and births on a second host thread (run_v3's replay lane; measured 3-4 ms SLOWER per 10^6-unique pass while the device is the bound: not the default)
  int v3_mirror = 1;                  // DADA2HIP_V3_MIRROR=0           the persistent tail reads every unique's partition / flags from global memory in each sweep (no LDS mirror)
  int v3_fail_entry = 0;              // DADA2HIP_V3_FAIL_ENTRY=n       test knob: the n-th persistent launch fails its entry barrier (-> launch chains)
  bool v2_debug = false;              // DADA2HIP_V2_DEBUG              per-block trace on stderr
  bool v2_summary = false;            // DADA2HIP_V2_SUMMARY            per-pass summary on stderr
  bool v2_trace_on = false;           // DADA2HIP_V2_TRACE=<round>[:file]   in-kernel phase stamps of one round (tools/trace_round.py)
  int v2_trace_seq = -1;
  std::string v2_trace_file;
  // ---- measurement / plumbing ----
  bool profile = false;               // DADA2HIP_PROFILE=1             event-time every launch (dev_ms_* in the stats)
  long long node_cap = 0;             // DADA2HIP_NODE_CAP              first allocation of the comparison store (tests: the growth path)
  bool wait_block = false;            // DADA2HIP_WAIT=block            blocking synchronisation instead of spinning
  double wait_timeout_s = 600.0;      // DADA2HIP_WAIT_TIMEOUT_S        bound on every host-side device wait
  int coop_max = 4000000;             // DADA2HIP_COOP_MAX              (experiment) uniques above which round 0 leaves k_nw_ad
  bool kord_align = false;            // DADA2HIP_KORD_ALIGN=1          (experiment) k-mer rows padded to 64 bytes
  int screen_grid = 2048;             // DADA2HIP_SCREEN_GRID           (experiment) block cap of k_screen
  long long ad_fcap = 0;              // DADA2HIP_AD_FCAP               rows of k_ad_product's offset buffer (tests: the in-kernel product)
  int screen_bits = 1;                // DADA2HIP_SCREEN_BITS=0         no 5-mer presence bitmaps: the batch screen walks every unique's k-mer record
  int ad_fast = 1;                    // DADA2HIP_AD_FAST=0             batch compares on the full aligner only (no pointer-free first pass)
  int ad_debug = 0;                   // DADA2HIP_AD_DEBUG              profiling build only (make prof): skips phases of k_nw_ad, results void
  bool bimera_times = false;          // DADA2HIP_BIMERA_TIMES=1        stderr: host / device split of a bimera call
  bool derep_zlib = false;            // DADA2HIP_DEREP_INFLATE=zlib    .gz files through zlib's streaming inflate even where libdeflate is installed
  bool derep_times = false;           // DADA2HIP_DEREP_TIMES=1         stderr: phases of a dada2hip_derep_fastq call
  // (read once per process, when the host pool / the allocation cache are created: DADA2HIP_HOST_THREADS, DADA2HIP_ALLOC_CACHE,
  //  DADA2HIP_ALLOC_CACHE_GB)
  int host_threads = 0;
  int alloc_cache = -1;
  long long alloc_cache_gb = -1;

  static Knobs from_env() {
    Knobs k;
    auto S = [](const char *n) -> const char * { return std::getenv(n); };
    auto I = [&](const char *n, int dflt) -> int { const char *e = S(n); return e ? std::atoi(e) : dflt; };
    auto B = [&](const char *n) -> bool { const char *e = S(n); return e && std::atoi(e) != 0; };
    auto T = [&](const char *n) -> int { const char *e = S(n); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; };   // tri-state
    if (const char *e = S("DADA2HIP_ENGINE")) k.engine_classic = !std::strcmp(e, "classic");
    if (const char *e = S("DADA2HIP_NW_KERNEL")) k.nw_kernel = !std::strcmp(e, "lane") ? 1 : (!std::strcmp(e, "coop") ? 2 : (!std::strcmp(e, "wide") ? 3 : 0));
    if (const char *e = S("DADA2HIP_AD_HOMO")) k.ad_homo = std::strcmp(e, "0") != 0;
    k.no_speculation = S("DADA2HIP_NO_SPECULATION") != nullptr;
    k.no_autobirth = S("DADA2HIP_NO_AUTOBIRTH") != nullptr;
    k.v2_nbuf = I("DADA2HIP_V2_NBUF", 0); k.v2_depth = I("DADA2HIP_V2_DEPTH", 0); k.v2_chain = I("DADA2HIP_V2_CHAIN", 0);
    k.v2_graph = B("DADA2HIP_V2_GRAPH");
    k.v2_lite = T("DADA2HIP_V2_LITE");
    if (const char *e = S("DADA2HIP_V2_ALIGN")) k.v2_align = !std::strcmp(e, "commit") ? 1 : 0;
    k.v2_filter = T("DADA2HIP_V2_FILTER");
    k.v2_grid_shuffle = I("DADA2HIP_V2_GRID_SHUFFLE", 0); k.v2_grid_pupdate = I("DADA2HIP_V2_GRID_PUPDATE", 0);
    k.v2_mov_inline = I("DADA2HIP_V2_MOV_INLINE", 0);
    if (const char *e = S("DADA2HIP_V2_TAIL")) k.v2_tail_chain = !std::strcmp(e, "chain");
    k.v3_grid = I("DADA2HIP_V3_GRID", 0); k.v3_ring = I("DADA2HIP_V3_RING", 0); k.v3_block = I("DADA2HIP_V3_BLOCK", 0);
    k.v3_overlap = T("DADA2HIP_V3_OVERLAP"); k.v3_pf_wait_us = I("DADA2HIP_V3_PF_WAIT_US", -1);
    k.v3_fail_entry = I("DADA2HIP_V3_FAIL_ENTRY", 0); k.v3_spec = I("DADA2HIP_V3_SPEC", 1); k.v3_xbar = T("DADA2HIP_V3_XBAR"); k.v3_spec_max = I("DADA2HIP_V3_SPEC_MAX", -1);
    k.v3_pf_early = I("DADA2HIP_V3_PF_EARLY", -1); k.v3_pf_lowreg = T("DADA2HIP_V3_PF_LOWREG"); k.v3_pf_sync = I("DADA2HIP_V3_PF_SYNC", 0); k.v3_pf_gate_us = I("DADA2HIP_V3_PF_GATE_US", -1);
    k.v3_mirror = I("DADA2HIP_V3_MIRROR", 1);
    k.v3_lane = I("DADA2HIP_V3_LANE", 0);
    k.v3_slots = I("DADA2HIP_V3_SLOTS", 3);
    k.replay_radix_min = I("DADA2HIP_REPLAY_RADIX_MIN", 4096);
    k.v2_debug = S("DADA2HIP_V2_DEBUG") != nullptr; k.v2_summary = S("DADA2HIP_V2_SUMMARY") != nullptr;
    if (const char *e = S("DADA2HIP_V2_TRACE")) {
      k.v2_trace_on = true; k.v2_trace_seq = std::atoi(e);
      const char *colon = std::strchr(e, ':');
      k.v2_trace_file = colon ? colon + 1 : "dada2hip_trace.bin";
    }
    k.profile = B("DADA2HIP_PROFILE");
    if (const char *e = S("DADA2HIP_NODE_CAP")) k.node_cap = std::atoll(e);
    if (const char *e = S("DADA2HIP_WAIT")) k.wait_block = !std::strcmp(e, "block");
    if (const char *e = S("DADA2HIP_WAIT_TIMEOUT_S")) k.wait_timeout_s = std::atof(e);
    k.coop_max = I("DADA2HIP_COOP_MAX", 4000000);
    if (const char *e = S("DADA2HIP_KORD_ALIGN")) k.kord_align = !std::strcmp(e, "1");
    k.screen_grid = I("DADA2HIP_SCREEN_GRID", 2048);
    if (const char *e = S("DADA2HIP_AD_FCAP")) k.ad_fcap = std::atoll(e);
    k.ad_debug = I("DADA2HIP_AD_DEBUG", 0); k.ad_fast = I("DADA2HIP_AD_FAST", 1); k.screen_bits = I("DADA2HIP_SCREEN_BITS", 1);
    if (const char *e = S("DADA2HIP_BIMERA_TIMES")) k.bimera_times = !std::strcmp(e, "1");
    if (const char *e = S("DADA2HIP_DEREP_INFLATE")) k.derep_zlib = !std::strcmp(e, "zlib");
    if (const char *e = S("DADA2HIP_DEREP_TIMES")) k.derep_times = !std::strcmp(e, "1");
    k.host_threads = I("DADA2HIP_HOST_THREADS", 0);
    k.alloc_cache = T("DADA2HIP_ALLOC_CACHE");
    if (const char *e = S("DADA2HIP_ALLOC_CACHE_GB")) k.alloc_cache_gb = std::atoll(e);
    return k;
  }
};

namespace knobs_detail {
inline std::mutex &mu() { static std::mutex m; return m; }          // serialises reloads only
inline std::atomic<const Knobs *> &cur() { static std::atomic<const Knobs *> p{nullptr}; return p; }
inline bool same(const Knobs &a, const Knobs &b) {
  // (field-wise: the struct holds a std::string)
  return a.engine_classic == b.engine_classic && a.nw_kernel == b.nw_kernel && a.ad_homo == b.ad_homo && a.no_speculation == b.no_speculation &&
         a.no_autobirth == b.no_autobirth && a.v2_nbuf == b.v2_nbuf && a.v2_depth == b.v2_depth && a.v2_chain == b.v2_chain && a.v2_graph == b.v2_graph &&
         a.v2_lite == b.v2_lite && a.v2_align == b.v2_align && a.v2_filter == b.v2_filter && a.v2_grid_shuffle == b.v2_grid_shuffle &&
         a.v2_grid_pupdate == b.v2_grid_pupdate && a.v2_mov_inline == b.v2_mov_inline && a.v2_tail_chain == b.v2_tail_chain && a.v3_grid == b.v3_grid &&
         a.v3_ring == b.v3_ring && a.v3_block == b.v3_block && a.v3_overlap == b.v3_overlap && a.v3_pf_wait_us == b.v3_pf_wait_us &&
         a.v3_fail_entry == b.v3_fail_entry && a.v3_spec == b.v3_spec && a.v3_xbar == b.v3_xbar && a.v3_spec_max == b.v3_spec_max && a.v3_pf_early == b.v3_pf_early && a.v3_pf_lowreg == b.v3_pf_lowreg && a.v3_pf_sync == b.v3_pf_sync && a.v3_pf_gate_us == b.v3_pf_gate_us && a.v3_mirror == b.v3_mirror && a.v3_lane == b.v3_lane && a.v3_slots == b.v3_slots && a.replay_radix_min == b.replay_radix_min && a.v2_debug == b.v2_debug && a.v2_summary == b.v2_summary && a.v2_trace_on == b.v2_trace_on &&
         a.v2_trace_seq == b.v2_trace_seq && a.v2_trace_file == b.v2_trace_file && a.profile == b.profile && a.node_cap == b.node_cap &&
         a.wait_block == b.wait_block && a.wait_timeout_s == b.wait_timeout_s && a.coop_max == b.coop_max && a.kord_align == b.kord_align &&
         a.screen_grid == b.screen_grid && a.ad_fcap == b.ad_fcap && a.ad_debug == b.ad_debug && a.ad_fast == b.ad_fast && a.screen_bits == b.screen_bits && a.bimera_times == b.bimera_times && a.derep_times == b.derep_times && a.derep_zlib == b.derep_zlib &&
         a.host_threads == b.host_threads && a.alloc_cache == b.alloc_cache && a.alloc_cache_gb == b.alloc_cache_gb;
}
}  // namespace knobs_detail

// Re-read the environment (every C-ABI entry point does, once, before anything else).  A snapshot that differs from the current
// one replaces it; old snapshots are never freed (a few hundred bytes each, and only when the environment changed), so a
// reference obtained from knobs() by another thread stays valid.
// The ONE variable the library SETS (once, when it is loaded, and only if the caller has not): the HIP runtime maps a process's
// streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) when it initialises.  A run uses three streams (rounds, side copies,
// prefetch compares) and several samples in flight use three each; streams that share a queue run one behind the other - a
// persistent launch at the head of a queue holds back whatever else was mapped onto it.  With 8: configs[3] on one GPU 204 -> 190 ms
// at two slots (profiles/r10i_*, r10j_*); the single-sample call reads the same with 4, 8 or 16 (151-158 ms, profiles/r11b_*).  No effect if the runtime
// is already up (torch initialised first): dada2_amd/_lib.py and bench.py set it before either is.
inline void knobs_process_defaults() { (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0); }

inline void knobs_reload() {
  Knobs k = Knobs::from_env();
  std::lock_guard<std::mutex> g(knobs_detail::mu());
  const Knobs *c = knobs_detail::cur().load(std::memory_order_relaxed);
  if (!c || !knobs_detail::same(*c, k)) knobs_detail::cur().store(new Knobs(std::move(k)), std::memory_order_release);
}
// (an acquire load: knobs() is called from polling loops and per-round paths of several host threads - no lock on the read side)
inline const Knobs &knobs() {
  const Knobs *c = knobs_detail::cur().load(std::memory_order_acquire);
  if (c) return *c;
  knobs_reload();
  return *knobs_detail::cur().load(std::memory_order_acquire);
}

}  // namespace d2
