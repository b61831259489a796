// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for DADA2's divisive-denoising
// hot path.  No MFMA: the work is integer DP, byte/halfword streaming and sequential fp64 products
// (DESIGN.md §4 gives the roofline of each kernel).  Reference behaviour followed, file:line under
// /root/reference/src:
//   k_build_kmers   assign_kmer / assign_kmer8 / assign_kmer_order      kmers.cpp:158-279
//   k_screen        kmer_dist_SSEi_8 / kmer_dist_SSEi / kord_dist_SSEi  kmers.cpp:29-150
//                   + the dispatch of raw_align                          nwalign_endsfree.cpp:10-73
//   k_gapless       nwalign_gapless + al2subs + compute_lambda_ts       nwalign_endsfree.cpp:539-639, pval.cpp:144-197
//   k_nw / k_nw_gen nwalign_vectorized2 == nwalign_endsfree + al2subs + compute_lambda_ts
//                                                                        nwalign_vectorized.cpp:71-318, nwalign_endsfree.cpp:76-216
//   k_calc_pA       calc_pA                                              pval.cpp:44-64
//   k_final_*       b_make_transition_by_quality_matrix / b_make_cluster_quality_matrix   error.cpp:131-172, :225-258
#include <cstring>

#include "engine.h"
#include "gcn.h"
#include "knobs.h"
#include "rounds_common.h"
#include "ppois.h"

namespace d2 {

static __device__ __forceinline__ uint32_t base_at(const uint32_t *__restrict__ row, int p) {
  return (row[p >> 4] >> ((p & 15) << 1)) & 3u;
}

// position p of a 2-bit row of length len lies in a run of >= 3 equal bases (nwalign_endsfree.cpp:227-256)
static __device__ __forceinline__ uint32_t homo_at(const uint32_t *__restrict__ row, int len, int p) {
  if (p < 0 || p >= len) return 0u;
  const uint32_t b = base_at(row, p);
  const int l = (p >= 1 && base_at(row, p - 1) == b) ? ((p >= 2 && base_at(row, p - 2) == b) ? 2 : 1) : 0;
  const int r = (p + 1 < len && base_at(row, p + 1) == b) ? ((p + 2 < len && base_at(row, p + 2) == b) ? 2 : 1) : 0;
  return (l + r >= 2) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------------
// k-mer records, one thread per unique with a private 1024-entry u16 count table in LDS
// (table[km][lane], 128 KiB per 64-thread block).  For every position i < len-4 it emits the
// ordered k-mer id (kmers.cpp:246-279) together with its occurrence rank = number of earlier
// positions holding the same k-mer, saturated at 63.  With ranks, the unordered overlap of the
// count tables (kmers.cpp:13-93) is   sum_k min(a_k, b_k) = #{ i : rank_i < count_centre[kmer_i] },
// so a round streams ~2 B/position instead of the 1 KiB count table per unique; k-mers that occur
// more than 63 times go to a small per-unique "heavy" list and are corrected exactly.
__global__ __launch_bounds__(64) void k_build_kmers(SampleDev S) {
  __shared__ uint16_t tbl[NKMER * 64];
  const int lane = threadIdx.x;
  uint16_t *t = tbl + lane;
  for (int base = blockIdx.x * 64; base < S.N; base += gridDim.x * 64) {
    const int r = base + lane;
    for (int k = 0; k < NKMER; k++) t[k * 64] = 0;
    if (r < S.N) {
      const uint32_t *row = S.seq2 + (size_t)r * S.W2;
      uint16_t *ko = S.kord + (size_t)r * S.LK;
      const int L = S.len[r], nk = L - KMER_SIZE + 1;
      uint32_t km = 0;
      for (int p = 0; p < L; p++) {
        km = ((km << 2) | base_at(row, p)) & (NKMER - 1);   // first base most significant (kmers.cpp:176-184)
        if (p >= KMER_SIZE - 1) {
          uint32_t rk = t[km * 64];
          t[km * 64] = (uint16_t)(rk + 1);
          ko[p - (KMER_SIZE - 1)] = (uint16_t)(km | ((rk < RANK_SAT ? rk : RANK_SAT) << 10));
        }
      }
      for (int i = nk; i < S.LK; i++) ko[i] = 0xFFFFu;
      int nh = 0;
      if (S.HMAX > 0 || S.kbits) {
        int distinct = 0;
        for (int w = 0; w < NKMER / 32; w++) {
          uint32_t bits = 0;
          for (int b = 0; b < 32; b++) {
            const int k = 32 * w + b;
            const uint32_t c = t[k * 64];
            if (c) { bits |= 1u << b; distinct++; }
            if (S.HMAX > 0 && c > RANK_SAT) S.heavy[(size_t)r * S.HMAX + nh++] = (uint32_t)k | (c << 16);
          }
          if (S.kbits) S.kbits[(size_t)r * 32 + w] = bits;
        }
        if (S.kmult) S.kmult[r] = (uint16_t)min(max(nk, 0) - distinct, 65535);
      }
      S.nheavy[r] = (uint8_t)nh;
    }
  }
}

void launch_build_kmers(const SampleDev &S, hipStream_t st) {
  int grid = std::min((S.N + 63) / 64, 1024);
  hipLaunchKernelGGL(k_build_kmers, dim3(grid), dim3(64), 0, st, S);
}

// ------------------------------------------------------------------------------------------------
// Centre record of a round, built once by one block so the screen blocks only copy ~3.5 KB out of L2:
//   ctab[0..255]    : u8[1024]  min(count, 63) per 5-mer  (rank_i < min(count,63) <=> rank_i < 63 && rank_i < count)
//   ctab[256..767]  : u16[1024] full counts (only read for the rare "heavy" k-mer correction)
//   ctab[768.. ]    : u16[LK]   ordered k-mers, 0xFFFF past the end
constexpr int CTAB_SAT = 0, CTAB_CNT = 256, CTAB_ORD = 768;
static __device__ __forceinline__ void centre_table_body(const SampleDev &S, int centre, uint32_t *__restrict__ ctab, uint32_t *cnt) {
  const int tid = threadIdx.x;
  for (int k = tid; k < NKMER; k += 256) cnt[k] = 0;
  __syncthreads();
  const int nkc = S.len[centre] - KMER_SIZE + 1;
  const uint16_t *crow = S.kord + (size_t)centre * S.LK;
  uint16_t *ko = (uint16_t *)(ctab + CTAB_ORD);
  for (int i = tid; i < S.LK; i += 256) {
    const uint32_t km = crow[i] & 1023u;
    if (i < nkc) atomicAdd(&cnt[km], 1u);
    ko[i] = i < nkc ? (uint16_t)km : (uint16_t)0xFFFF;
  }
  __syncthreads();
  uint8_t *sat = (uint8_t *)(ctab + CTAB_SAT);
  uint16_t *full = (uint16_t *)(ctab + CTAB_CNT);
  for (int k = tid; k < NKMER; k += 256) {
    const uint32_t c = cnt[k];
    sat[k] = (uint8_t)(c < RANK_SAT ? c : RANK_SAT);
    full[k] = (uint16_t)c;
  }
}
__global__ __launch_bounds__(256) void k_centre_table(SampleDev S, int centre, uint32_t *__restrict__ ctab) {
  __shared__ uint32_t cnt[NKMER];
  centre_table_body(S, centre, ctab, cnt);
}
void launch_centre_table(const SampleDev &S, int centre, uint32_t *d_ctab, hipStream_t st) {
  hipLaunchKernelGGL(k_centre_table, dim3(1), dim3(256), 0, st, S, centre, d_ctab);
}

// ------------------------------------------------------------------------------------------------
// k-mer screen of one b_compare round: every unique against the partition centre.  16 lanes per
// unique (4 uniques per wave); the centre's count table and ordered k-mers live in LDS.  HBM-bound:
// algorithmic bytes per unique = 2*(len-4) (k-mer records) + 4 (len) + 1 (skip) + 1 (class out).
__global__ __launch_bounds__(256) void k_screen(SampleDev S, int centre, ScreenParams sp,
                                                const uint8_t *__restrict__ skip, const uint8_t *__restrict__ lock,
                                                int greedy, const int32_t *__restrict__ thresh,
                                                uint8_t *__restrict__ cls, double *__restrict__ lam,
                                                uint32_t *__restrict__ ham, int32_t *__restrict__ nw_list,
                                                int32_t *__restrict__ gl_list, int32_t *__restrict__ counters,
                                                int cap, const uint32_t *__restrict__ ctab,
                                                const int32_t *__restrict__ centre_dev) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
  // speculative launch (enqueued before the host has seen the bud decision): the centre comes from the descriptor
  // k_auto_birth wrote, -1 = no birth was applied on the device -> nothing to do
  if (centre_dev) { centre = *centre_dev; if (centre < 0) return; }
  const uint8_t *csat = (const uint8_t *)s_mem;            // [1024] min(count, 63)
  const uint16_t *cfull = (const uint16_t *)(s_mem + CTAB_CNT);   // [1024] full counts
  int32_t *s_cnt = (int32_t *)(s_mem + CTAB_ORD);           // [8]
  int32_t *s_nw = s_cnt + 8;                                // [cap] this block's NW work items
  int32_t *s_gl = s_nw + cap;                               // [cap] this block's gapless work items
  const uint16_t *ckord = (const uint16_t *)(s_gl + cap);   // [LK] centre ordered k-mers
  const int tid = threadIdx.x;
  const int Lc = S.len[centre];
  const uint32_t creads = S.reads[centre];
  {   // centre record built once for the round by k_centre_table
    const uint4 *src = (const uint4 *)ctab;
    for (int i = tid; i < CTAB_ORD / 4; i += 256) ((uint4 *)s_mem)[i] = src[i];
    const int nk4 = (S.LK * 2 + 15) / 16;
    const uint4 *srck = (const uint4 *)(ctab + CTAB_ORD);
    for (int i = tid; i < nk4; i += 256) ((uint4 *)ckord)[i] = srck[i];
  }
  if (tid < 8) s_cnt[tid] = 0;
  __syncthreads();
  const int sub = tid & 15, grp = tid >> 4;
  const int nchunk = (S.maxlen - KMER_SIZE + 1 + 7) >> 3;     // 16-byte chunks (8 k-mer records) per row; rows may be padded to 128 B
  const bool two = nchunk <= 32;                              // each lane owns <= 2 chunks: keep them in registers
  for (int base = S.r_lo + blockIdx.x * 16; base < S.r_hi; base += gridDim.x * 16) {
    const int r = base + grp;
    if (r >= S.r_hi) continue;
    // issue the row's chunks and the per-unique scalars together (independent loads)
    const uint4 *row = (const uint4 *)(S.kord + (size_t)r * S.LK);
    const uint4 pad4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // rank 63: never counted
    uint4 c0 = (sp.use_kmers && sub < nchunk) ? row[sub] : pad4;
    uint4 c1 = (sp.use_kmers && two && sub + 16 < nchunk) ? row[sub + 16] : pad4;
    const int Lr = S.len[r];
    // greedy skip (cluster.cpp:127-130): more reads than the centre, or locked to its partition
    const bool skipped = (skip && skip[r]) || (greedy && (S.reads[r] > creads || (lock && lock[r])));
    const int d = (Lc < Lr ? Lc : Lr) - KMER_SIZE + 1;
    // ---- pass 1: unordered overlap  dot = #{ i : rank_i < min(count_centre[kmer_i], 63) }
    // (rows are padded with 0xFFFF = rank 63 past len-4, so no bounds test is needed)
    uint32_t dot = 0;
    auto dot8 = [&](const uint4 &v) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t x = w[e];
        dot += (((x >> 10) & 63u) < csat[x & 1023u]);
        dot += ((x >> 26) < csat[(x >> 16) & 1023u]);
      }
    };
    if (!skipped && sp.use_kmers) {
      dot8(c0);
      if (two) dot8(c1);
      else for (int ch = sub + 16; ch < nchunk; ch += 16) dot8(row[ch]);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, 16);
    if (!skipped && sp.use_kmers && S.HMAX > 0) {            // k-mers occurring > 63 times: exact correction
      const int nh = S.nheavy[r];
      for (int h = 0; h < nh; h++) {
        const uint32_t e = S.heavy[(size_t)r * S.HMAX + h], cr = e >> 16, cc = cfull[e & 1023u];
        const uint32_t m = cr < cc ? cr : cc;
        if (m > RANK_SAT) dot += m - RANK_SAT;
      }
    }
    dot &= 0xFFFFu;                                           // the reference accumulates in uint16_t (kmers.cpp:16,34,69)
    const bool shroud = !skipped && sp.use_kmers && (int)dot < thresh[d];   // kdist > kdist_cutoff
    // ---- pass 2 (survivors only, ~5 % of a round): ordered overlap over the first d positions
    const bool gl_ok = sp.gapless && sp.use_kmers && (sp.sse >= 1 || Lr == Lc);   // kord_dist: -1 on unequal lengths when SSE==0
    uint32_t ord = 0;
    if (!skipped && !shroud && gl_ok && sp.band != 0) {
      auto ord8 = [&](const uint4 &v, int ch) {
        const uint4 ck4 = ((const uint4 *)ckord)[ch];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w}, cw[4] = {ck4.x, ck4.y, ck4.z, ck4.w};
        const int i0 = ch << 3;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t t = (w[e] ^ cw[e]) & 0x03FF03FFu;    // k-mer ids only (rank bits masked off)
          ord += ((t & 0xFFFFu) == 0 && i0 + 2 * e < d);
          ord += ((t >> 16) == 0 && i0 + 2 * e + 1 < d);
        }
      };
      if (sub < nchunk) ord8(c0, sub);
      if (two) { if (sub + 16 < nchunk) ord8(c1, sub + 16); }
      else for (int ch = sub + 16; ch < nchunk; ch += 16) ord8(row[ch], ch);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) ord += __shfl_xor(ord, o, 16);
      ord &= 0xFFFFu;
    }
    if (sub == 0) {
      uint8_t c;
      if (skipped) c = CLS_SKIP;
      else if (shroud) c = CLS_SHROUD;
      else if (sp.band == 0 || (gl_ok && ord == dot)) c = CLS_GAPLESS;                // kodist == kdist
      else c = CLS_NW;
      cls[r] = c;
      // skipped / shrouded uniques get lambda 0, hamming -1 (cluster.cpp:139-143): implied by cls[], not written here
      // (8-byte scattered stores cost a 64-byte HBM write each); k_store / k_fill_null materialise them.
      if (c == CLS_GAPLESS) s_gl[atomicAdd(&s_cnt[1], 1)] = r;
      else if (c == CLS_NW) s_nw[atomicAdd(&s_cnt[0], 1)] = r;
    }
  }
  __syncthreads();
  if (tid < 2) {
    const int n = s_cnt[tid];
    s_cnt[4 + tid] = n ? atomicAdd(&counters[tid], n) : 0;   // one global atomic per list per block
  }
  __syncthreads();
  for (int i = tid; i < s_cnt[0]; i += 256) nw_list[s_cnt[4] + i] = s_nw[i];
  for (int i = tid; i < s_cnt[1]; i += 256) gl_list[s_cnt[5] + i] = s_gl[i];
}

void launch_screen(const SampleDev &S, int centre, const ScreenParams &sp, const uint8_t *d_skip, const uint8_t *d_lock,
                   int greedy, const int32_t *d_thresh, uint8_t *d_cls, double *d_lambda, uint32_t *d_ham, int32_t *d_nw_list,
                   int32_t *d_gl_list, int32_t *d_counters, uint32_t *d_ctab, bool build_table, const int32_t *d_centre_dev,
                   hipStream_t st) {
  if (build_table) hipLaunchKernelGGL(k_centre_table, dim3(1), dim3(256), 0, st, S, centre, d_ctab);
  const int grid_cap = std::max(1, knobs().screen_grid);
  int grid = std::min((S.N + 15) / 16, grid_cap);
  int iters = ((S.N + 15) / 16 + grid - 1) / grid;
  int cap = iters * 16;
  size_t lds = (size_t)(CTAB_ORD + 8) * 4 + (size_t)cap * 8 + (size_t)S.LK * 2 + 32;
  while (lds > 64 * 1024) {   // very large N: more blocks, shorter per-block lists
    grid *= 2;
    iters = ((S.N + 15) / 16 + grid - 1) / grid;
    cap = iters * 16;
    lds = (size_t)(CTAB_ORD + 8) * 4 + (size_t)cap * 8 + (size_t)S.LK * 2 + 32;
  }
  hipLaunchKernelGGL(k_screen, dim3(grid), dim3(256), lds, st, S, centre, sp, d_skip, d_lock, greedy, d_thresh, d_cls, d_lambda,
                     d_ham, d_nw_list, d_gl_list, d_counters, cap, (const uint32_t *)d_ctab, d_centre_dev);
}

// ------------------------------------------------------------------------------------------------
// Class of isolated (centre, raw) pairs — the birth substitutions of Rmain.cpp:209-215 are
// sub_new(parent centre, new centre, use_kmers, cutoff 1.0): never shrouded (kdist <= 1), gapless iff
// band == 0 or ordered == unordered k-mer overlap (nwalign_endsfree.cpp:54).  One wave per pair.
__global__ __launch_bounds__(256) void k_pair_class(SampleDev S, const int32_t *__restrict__ pc,
                                                    const int32_t *__restrict__ pr, int n, ScreenParams sp,
                                                    uint8_t *__restrict__ out) {
  __shared__ uint32_t s_cnt[4][NKMER];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t *ccnt = s_cnt[w];
  for (int base = blockIdx.x * 4; base < n; base += gridDim.x * 4) {
    const int pair = base + w;
    const bool on = pair < n;
    const int c = on ? pc[pair] : 0, r = on ? pr[pair] : 0;
    const int Lc = S.len[c], Lr = S.len[r], nkc = Lc - KMER_SIZE + 1, nkr = Lr - KMER_SIZE + 1;
    const int d = (Lc < Lr ? Lc : Lr) - KMER_SIZE + 1;
    const uint16_t *crow = S.kord + (size_t)c * S.LK, *rrow = S.kord + (size_t)r * S.LK;
    for (int k = lane; k < NKMER; k += 64) ccnt[k] = 0;
    __syncthreads();
    for (int i = lane; i < nkc; i += 64) atomicAdd(&ccnt[crow[i] & 1023u], 1u);
    __syncthreads();
    uint32_t dot = 0, ord = 0;
    for (int i = lane; i < nkr; i += 64) {
      const uint32_t x = rrow[i], km = x & 1023u, rk = x >> 10;
      if (rk < RANK_SAT) dot += (rk < ccnt[km]);
      if (i < d) ord += (km == (crow[i] & 1023u));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { dot += __shfl_xor(dot, o, 64); ord += __shfl_xor(ord, o, 64); }
    if (on && lane == 0) {
      if (S.HMAX > 0) {
        const int nh = S.nheavy[r];
        for (int h = 0; h < nh; h++) {
          const uint32_t e = S.heavy[(size_t)r * S.HMAX + h], cr = e >> 16, cc = ccnt[e & 1023u];
          const uint32_t m = cr < cc ? cr : cc;
          if (m > RANK_SAT) dot += m - RANK_SAT;
        }
      }
      dot &= 0xFFFFu; ord &= 0xFFFFu;
      const bool gl_ok = sp.gapless && sp.use_kmers && (sp.sse >= 1 || Lr == Lc);
      out[pair] = (sp.band == 0 || (gl_ok && ord == dot)) ? CLS_GAPLESS : CLS_NW;
    }
    __syncthreads();
  }
}

void launch_pair_class(const SampleDev &S, const int32_t *d_pc, const int32_t *d_pr, int n, const ScreenParams &sp,
                       uint8_t *d_out, hipStream_t st) {
  if (n <= 0) return;
  int grid = std::min((n + 3) / 4, 1024);
  hipLaunchKernelGGL(k_pair_class, dim3(grid), dim3(256), 0, st, S, d_pc, d_pr, n, sp, d_out);
}

// ------------------------------------------------------------------------------------------------
// Gapless comparison: position-wise pairing (nwalign_gapless), substitutions where both bases
// exist and differ (al2subs), lambda as the sequential fp64 product over raw positions
// (compute_lambda_ts).  One thread per unique; the centre is wave-uniform.
__global__ __launch_bounds__(256) void k_gapless(SampleDev S, int centre, const int32_t *__restrict__ chunk_centre,
                                                 const int32_t *__restrict__ work, const int32_t *__restrict__ nwork_dev,
                                                 int nwork_host, AlignParams ap, const double *__restrict__ err,
                                                 double *__restrict__ lam, uint32_t *__restrict__ ham,
                                                 uint16_t *__restrict__ view, int LV, int view_by_chunk) {
  extern __shared__ double s_err[];
  for (int i = threadIdx.x; i < 16 * ap.ncol; i += blockDim.x) s_err[i] = err[i];
  __syncthreads();
  const int nwork = nwork_dev ? *nwork_dev : nwork_host;
  const int lane = threadIdx.x & 63, gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  for (int chunk = gwave; chunk * 64 < nwork; chunk += nwaves) {
    const int idx = chunk * 64 + lane;
    const int r = idx < nwork ? work[idx] : -1;
    if (r < 0) continue;
    const int c = chunk_centre ? chunk_centre[chunk] : centre;
    const uint32_t *crow = S.seq2 + (size_t)c * S.W2, *rrow = S.seq2 + (size_t)r * S.W2;
    const uint8_t *qrow = S.qual + (size_t)r * S.LQ;
    const int L1 = S.len[c], L2 = S.len[r];
    const size_t vr = view_by_chunk ? (size_t)chunk : (size_t)r;
    double l = 1.0;
    uint32_t h = 0, cw = 0, rw = 0, qw = 0;
    for (int p = 0; p < L2; p++) {
      if ((p & 15) == 0) { rw = rrow[p >> 4]; cw = p < L1 ? crow[p >> 4] : 0; }
      if ((p & 3) == 0) qw = *(const uint32_t *)(qrow + p);
      const uint32_t rb = (rw >> ((p & 15) << 1)) & 3u, q = ap.use_quals ? ((qw >> ((p & 3) << 3)) & 255u) : 0u;
      uint32_t t = 5u * rb;
      if (p < L1) {
        const uint32_t cb = (cw >> ((p & 15) << 1)) & 3u;
        t = 4u * cb + rb;
        h += (cb != rb);
        if (view) view[vr * LV + p] = (uint16_t)(0x8000u | (rb << 8) | q);
      }
      l = l * s_err[t * ap.ncol + q];
    }
    if (view) for (int p = L2; p < L1; p++) view[vr * LV + p] = 0;   // centre positions opposite the end gap
    lam[r] = l;
    ham[r] = h;
  }
}

void launch_gapless(const SampleDev &S, int centre, const int32_t *d_chunk_centre, const int32_t *d_work,
                    const int32_t *d_nwork, int nwork_host, const AlignParams &ap, const double *d_err,
                    double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV, int view_by_chunk, hipStream_t st) {
  int maxwork = d_nwork ? S.N : nwork_host;
  if (maxwork <= 0) return;
  int grid = std::min((maxwork + 255) / 256, 2048);
  hipLaunchKernelGGL(k_gapless, dim3(grid), dim3(256), (size_t)16 * ap.ncol * sizeof(double), st, S, centre,
                     d_chunk_centre, d_work, d_nwork, nwork_host, ap, d_err, d_lambda, d_ham, d_view, LV, view_by_chunk);
}

// The gapless pairs of a batch compare (round engine v2, NwBatch): rows KB_MAX .. 2 KB_MAX - 1 of the batch lists, one thread per
// pair as above, a wave works on one batch position (its centre is wave-uniform).  They used to ride along in the aligner's
// launch, three to a wave; 64 to a wave they take a tenth of that time.
__global__ __launch_bounds__(256) void k_gapless_batch(SampleDev S, NwBatch b, AlignParams ap, const double *__restrict__ err,
                                                       double *__restrict__ lam, uint32_t *__restrict__ ham, const int32_t *__restrict__ stop_dev) {
  if (stop_dev && *stop_dev != 0) return;
  const int nb = *b.on;
  if (nb <= 0) return;
  int wk[KB_MAX + 1];                                        // first 64-pair slice of every batch position
  wk[0] = 0;
#pragma unroll
  for (int k = 0; k < KB_MAX; k++) wk[k + 1] = wk[k] + (k < nb ? (b.n[KB_MAX + k] + 63) / 64 : 0);
  if ((int)blockIdx.x * 4 >= wk[KB_MAX]) return;
  extern __shared__ double s_err[];
  for (int i = threadIdx.x; i < 16 * ap.ncol; i += blockDim.x) s_err[i] = err[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  for (int w = gwave; w < wk[KB_MAX]; w += nwaves) {
    int k = 0, w0 = 0;
#pragma unroll
    for (int q = 1; q < KB_MAX; q++) if (w >= wk[q]) { k = q; w0 = wk[q]; }
    const int idx = (w - w0) * 64 + lane;
    if (idx >= b.n[KB_MAX + k]) continue;
    const int r = b.list[(size_t)(KB_MAX + k) * b.stride + idx], c = b.centre[k];
    const uint32_t *crow = S.seq2 + (size_t)c * S.W2, *rrow = S.seq2 + (size_t)r * S.W2;
    const uint8_t *qrow = S.qual + (size_t)r * S.LQ;
    const int L1 = S.len[c], L2 = S.len[r];
    double l = 1.0;
    uint32_t h = 0, cw = 0, rw = 0, qw = 0;
    for (int p = 0; p < L2; p++) {                           // nwalign_gapless + compute_lambda_ts, exactly as k_gapless
      if ((p & 15) == 0) { rw = rrow[p >> 4]; cw = p < L1 ? crow[p >> 4] : 0; }
      if ((p & 3) == 0) qw = *(const uint32_t *)(qrow + p);
      const uint32_t rb = (rw >> ((p & 15) << 1)) & 3u, q = ap.use_quals ? ((qw >> ((p & 3) << 3)) & 255u) : 0u;
      uint32_t t = 5u * rb;
      if (p < L1) {
        const uint32_t cb = (cw >> ((p & 15) << 1)) & 3u;
        t = 4u * cb + rb;
        h += (cb != rb);
      }
      l = l * s_err[t * ap.ncol + q];
    }
    const size_t o = ((size_t)*b.bbuf * KB_MAX + k) * b.stride + r;
    lam[o] = l;
    ham[o] = h;
  }
}
void launch_gapless_batch(const SampleDev &S, const NwBatch &b, const AlignParams &ap, const double *d_err, double *d_lambda,
                          uint32_t *d_ham, const int32_t *d_stop_dev, hipStream_t st) {
  const int grid = std::min((S.N + 255) / 256 + KB_MAX, 2048);
  hipLaunchKernelGGL(k_gapless_batch, dim3(grid), dim3(256), (size_t)16 * ap.ncol * sizeof(double), st, S, b, ap, d_err, d_lambda, d_ham,
                     d_stop_dev);
}

// ------------------------------------------------------------------------------------------------
// Banded ends-free Needleman-Wunsch, ONE ALIGNMENT PER LANE (64 per wave, the centre is
// wave-uniform).  Band coordinates: cell (i, j) lives at k = j - i + lband, so in row i
//   diag (i-1,j-1) = d[k] of the previous row, up (i-1,j) = d[k+1], left (i,j-1) = d[k-1] of this row;
// the row is updated in place in registers (d[WMAX]).  Tie-break up > left > diag
// (nwalign_endsfree.cpp:146-156).  First row/column are 0 (ends-free), moves along the last
// row/column are free (:121-134), neighbours outside the band read `sentinel` (:113-119).
// 2-bit traceback pointers go to an HBM scratch ring laid out [row][word][lane] so every store is a
// coalesced 256 B; the traceback emits the transition code of every raw position (4 bits) and the
// lambda product then runs forward over raw positions 0..len-1 exactly as pval.cpp:188-192 does.
struct NwArgs {
  SampleDev S;
  int centre;
  const int32_t *centre_dev;   // speculative round: centre read from the device descriptor (-1 = nothing to do)
  const int32_t *chunk_centre;
  const int32_t *pair_centre;  // lane kernels (k_nw, k_nw_gen): the centre of EVERY work item (pairwise alignments: C_nwvec, mergePairs) -
                               // 64 unrelated pairs to a wave instead of one; the row loops then run to each lane's own length
  const int32_t *work;
  const int32_t *nwork_dev;
  int nwork_host;
  AlignParams ap;
  const double *err;
  uint32_t *ptr_scr;
  uint32_t *t_scr;
  int32_t *row_scr;
  size_t ptr_wpw, t_wpw, row_wpw;
  double *lam;
  uint32_t *ham;
  uint16_t *view;
  int LV;
  int view_by_chunk;
  uint8_t *moves;
  int moves_stride;
  int32_t *nmoves;
  const int32_t *stop_dev;     // round engine v2: non-zero = the device has halted, nothing to do
  int32_t *lr_out;             // bimera mode of k_nw_ad: five words per work slot (left, right, left_oo, right_oo, hamming)
  int lr_one_off, lr_max_shift;
  // round engine v2, batch mode (k_nw_ad only): the alignments of a whole batch compare - up to KB_MAX centres - in ONE launch.
  // List k (k < KB_MAX) holds the uniques to align with batch centre k (list KB_MAX + k its gapless ones: k_gapless_batch);
  // blocks work on one centre at a time (it is staged once per block), results go to row (bbuf * KB_MAX + k) of lam / ham.
  const int32_t *batch_on;     // number of centres of the batch compare in flight (0: nothing to do), or nullptr: not batch mode
  const int32_t *batch_n;      // [2 KB_MAX] list lengths
  const int32_t *batch_list;   // [2 KB_MAX][batch_stride]
  const int32_t *batch_centre; // [KB_MAX] centre of each batch position
  const int32_t *batch_bbuf;   // batch buffer the results belong to
  size_t batch_stride;         // row length of batch_list and of lam / ham
  // k_nw_ad<.., FAST> (batch mode): the pairs whose walk back is NOT "free end run + one whole diagonal + border run" - the pass
  // keeps no pointers, so it cannot finish them - go to row k of retry_list / retry_n[k], which the full kernel works through next
  int32_t *retry_list;
  int32_t *retry_n;
  // fast_ctl[0] = pairs handed over so far in this run, [1] = pairs the pointer-free pass has looked at, [2] != 0: the pass is OFF
  // for the rest of the run - it handed more than a quarter of its pairs over (reads with indels everywhere: PacBio-style), so two
  // sweeps cost more than one.  Decided by k2_batch_lists in front of each compare; while off the FAST launch returns at once and
  // the full kernel, launched on the retry lists, works through the batch's OWN lists instead (alt_list / alt_n)
  unsigned long long *fast_ctl;
  const int32_t *alt_list;
  const int32_t *alt_n;
};

// shared tail: traceback + lambda.  NPW = pointer words per row.
template <int NPW>
static __device__ __forceinline__ void nw_traceback_lambda(const NwArgs &a, const double *s_err, int lane, int idx,
                                                           bool active, int r, int c, int L1, int L2, int lband,
                                                           const uint32_t *ptr, uint32_t *tsc, int npw_rt, int chunk) {
  const SampleDev &S = a.S;
  const size_t vr = a.view_by_chunk ? (size_t)chunk : (size_t)r;
  const uint32_t *crow = S.seq2 + (size_t)c * S.W2, *rrow = S.seq2 + (size_t)r * S.W2;
  const uint8_t *qrow = S.qual + (size_t)r * S.LQ;
  const int npw = NPW > 0 ? NPW : npw_rt;
  int i = L1, j = L2, k = L2 - L1 + lband;
  uint32_t h = 0, tw = 0;
  int step = 0;
  while (i > 0 || j > 0) {
    uint32_t p;
    if (i == 0) p = 2;
    else if (j == 0) p = 3;
    else p = (ptr[((size_t)i * npw + (k >> 4)) * 64 + lane] >> ((k & 15) << 1)) & 3u;
    if (a.moves && active) a.moves[(size_t)idx * a.moves_stride + step] = (uint8_t)p;
    step++;
    if (p == 3) {
      i--; k++;
      if (a.view && active) a.view[vr * a.LV + i] = 0;
    } else {
      const int pj = j - 1;
      const uint32_t rb = base_at(rrow, pj);
      uint32_t t = 5u * rb;
      if (p != 2) {
        const uint32_t cb = base_at(crow, i - 1);
        t = 4u * cb + rb;
        h += (cb != rb);
        i--;
        if (a.view && active) a.view[vr * a.LV + i] = (uint16_t)(0x8000u | (rb << 8) | (a.ap.use_quals ? qrow[pj] : 0));
      } else {
        k--;
      }
      j--;
      tw |= t << ((pj & 7) << 2);
      if ((pj & 7) == 0) { tsc[(size_t)(pj >> 3) * 64 + lane] = tw; tw = 0; }
    }
  }
  if (a.nmoves && active) a.nmoves[idx] = step;
  // lambda: sequential fp64 product over raw positions (pval.cpp:188-192)
  double l = 1.0;
  uint32_t qw = 0;
  for (int pj = 0; pj < L2; pj++) {
    if ((pj & 7) == 0) tw = tsc[(size_t)(pj >> 3) * 64 + lane];
    if ((pj & 3) == 0) qw = *(const uint32_t *)(qrow + pj);
    const uint32_t t = (tw >> ((pj & 7) << 2)) & 15u, q = a.ap.use_quals ? ((qw >> ((pj & 3) << 3)) & 255u) : 0u;
    l = l * s_err[t * a.ap.ncol + q];
  }
  if (active) { a.lam[r] = l; a.ham[r] = h; }
}

// PAIRS: a centre per work item (NwArgs::pair_centre) instead of one per wave.  PLAIN: the default aligner (ends-free, one gap
// penalty) with the switches of the other two compiled out - they cost the inner loop a quarter of its speed when they are
// run-time flags (bimera table 3 000 x 8: 1.25 s against 0.99 s)
template <int WMAX, bool PAIRS, bool PLAIN>
__global__ __launch_bounds__(256) void k_nw(NwArgs a) {
  constexpr int NPW = (2 * WMAX + 31) / 32;   // pointer words per row
  constexpr int NW32 = (WMAX + 15) / 16;      // raw-window words (2-bit codes)
  extern __shared__ double s_err[];
  for (int i = threadIdx.x; i < 16 * a.ap.ncol; i += blockDim.x) s_err[i] = a.err[i];
  __syncthreads();
  const SampleDev &S = a.S;
  const int lane = threadIdx.x & 63, gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  uint32_t *ptr = a.ptr_scr + (size_t)gwave * a.ptr_wpw;
  uint32_t *tsc = a.t_scr + (size_t)gwave * a.t_wpw;
  const int nwork = a.nwork_dev ? *a.nwork_dev : a.nwork_host;
  const int SENT = a.ap.sentinel, MATCH = a.ap.match, MISMATCH = a.ap.mismatch, GAP = a.ap.gap, B = a.ap.band;
  for (int chunk = gwave; chunk * 64 < nwork; chunk += nwaves) {
    const int idx = chunk * 64 + lane;
    constexpr bool pairs = PAIRS;
    const int c = pairs ? a.pair_centre[idx < nwork ? idx : nwork - 1]
                        : gcn_readfirstlane(a.chunk_centre ? a.chunk_centre[chunk] : a.centre);
    int r = idx < nwork ? a.work[idx] : -1;
    const bool active = r >= 0;
    if (!active) r = c;                       // idle lanes align the centre to itself, results dropped
    const int L1 = pairs ? S.len[c] : gcn_readfirstlane(S.len[c]);
    const int L2 = S.len[r];
    const int lband = B + (L1 > L2 ? L1 - L2 : 0), rband = B + (L2 > L1 ? L2 - L1 : 0);
    const int W = lband + rband + 1;          // <= WMAX (host picks the kernel class)
    const uint32_t *crow = S.seq2 + (size_t)c * S.W2, *rrow = S.seq2 + (size_t)r * S.W2;

    // the other two scalar aligners of the reference ride on the same sweep (wave-uniform switches): EF = ends-free (first
    // row / column 0, free moves along the last row / column), HOMO = a gap opposite a homopolymer base costs HG
    const bool EF = PLAIN ? true : a.ap.endsfree != 0, HOMO = PLAIN ? false : (EF && a.ap.homo_gap != GAP);
    const int HG = a.ap.homo_gap;
    int d[WMAX];
    // row 0: D[0][j] = 0 (global: j * gap) for 0 <= j <= min(rband, L2)
#pragma unroll
    for (int k = 0; k < WMAX; k++) {
      const int j = k - lband;
      d[k] = (j >= 0 && j <= rband && j <= L2) ? (EF ? 0 : j * GAP) : SENT;
    }
    // raw window: code at k = base of raw position (0-based) i - lband + k - 1, for the row about to be computed
    uint32_t win[NW32], hwin[NW32];            // hwin: homopolymer flag of the raw base of each column (same fields)
#pragma unroll
    for (int w = 0; w < NW32; w++) { win[w] = 0; hwin[w] = 0; }
    for (int k = 0; k < WMAX; k++) {          // prologue for row i = 1
      const int p = 1 - lband + k - 1;
      const uint32_t code = (p >= 0 && p < L2) ? base_at(rrow, p) : 0u;
      // shift the whole window right by one code and insert at the top
#pragma unroll
      for (int w = 0; w < NW32 - 1; w++) win[w] = (win[w] >> 2) | (win[w + 1] << 30);
      win[NW32 - 1] = (win[NW32 - 1] >> 2) | (code << (((WMAX - 1) & 15) << 1));
      if (HOMO) {
#pragma unroll
        for (int w = 0; w < NW32 - 1; w++) hwin[w] = (hwin[w] >> 2) | (hwin[w + 1] << 30);
        hwin[NW32 - 1] = (hwin[NW32 - 1] >> 2) | (homo_at(rrow, L2, p) << (((WMAX - 1) & 15) << 1));
      }
    }
    int kzero = lband - 1;                    // k of column j == 0 in row i (row 1 here)
    int kend = L2 - 1 + lband;                // k of column j == L2 in row i
    uint32_t cw = 0;
    for (int i = 1; i <= L1; i++) {
      if (((i - 1) & 15) == 0) cw = crow[(i - 1) >> 4];   // wave-uniform -> scalar load (per lane with pair_centre)
      const uint32_t cb = (cw >> (((i - 1) & 15) << 1)) & 3u;
      const uint32_t crep = cb * 0x55555555u;
      uint32_t m[NW32];
#pragma unroll
      for (int w = 0; w < NW32; w++) {
        const uint32_t x = win[w] ^ crep;
        m[w] = ~(x | (x >> 1)) & 0x55555555u;             // bit 2k' set where raw base == centre base
      }
      const int gapL = (EF && i == L1) ? 0 : GAP;         // free moves along the last row
      const int gapU = (HOMO && homo_at(crow, L1, i - 1)) ? HG : GAP;   // up move: a gap opposite centre base i - 1
      const int khi = kend < W - 1 ? kend : W - 1;
      uint32_t pw[NPW];
#pragma unroll
      for (int w = 0; w < NPW; w++) pw[w] = 0;
      int leftv = SENT;
#pragma unroll
      for (int k = 0; k < WMAX; k++) {
        const uint32_t mbit = (m[k >> 4] >> ((k & 15) << 1)) & 1u;
        const int diag = d[k] + (mbit ? MATCH : MISMATCH);
        const int upn = (k + 1 < WMAX) ? d[k + 1] : SENT;
        const int up = upn + ((EF && k == kend) ? 0 : gapU);   // free moves along the last column
        const int left = leftv + ((HOMO && !(EF && i == L1) && ((hwin[k >> 4] >> ((k & 15) << 1)) & 1u)) ? HG : gapL);
        const bool t1 = left >= diag;
        const int e1 = t1 ? left : diag;
        const uint32_t p1 = t1 ? 2u : 1u;
        const bool t2 = up >= e1;
        const int e = t2 ? up : e1;
        const uint32_t p = t2 ? 3u : p1;
        const bool valid = (k > kzero) && (k <= khi);
        const int v = valid ? e : ((k == kzero) ? (EF ? 0 : i * GAP) : SENT);
        d[k] = v;
        leftv = v;
        pw[k >> 4] |= p << ((k & 15) << 1);
      }
#pragma unroll
      for (int w = 0; w < NPW; w++) ptr[((size_t)i * NPW + w) * 64 + lane] = pw[w];
      // advance the raw window to row i+1
      {
        const int p = (i + 1) - lband + (WMAX - 1) - 1;
        const uint32_t code = (p >= 0 && p < L2) ? base_at(rrow, p) : 0u;
#pragma unroll
        for (int w = 0; w < NW32 - 1; w++) win[w] = (win[w] >> 2) | (win[w + 1] << 30);
        win[NW32 - 1] = (win[NW32 - 1] >> 2) | (code << (((WMAX - 1) & 15) << 1));
        if (HOMO) {
#pragma unroll
          for (int w = 0; w < NW32 - 1; w++) hwin[w] = (hwin[w] >> 2) | (hwin[w + 1] << 30);
          hwin[NW32 - 1] = (hwin[NW32 - 1] >> 2) | (homo_at(rrow, L2, p) << (((WMAX - 1) & 15) << 1));
        }
      }
      kzero--;
      kend--;
    }
    nw_traceback_lambda<NPW>(a, s_err, lane, idx, active, r, c, L1, L2, lband, ptr, tsc, NPW, chunk);
  }
}

// Generic variant for any band (incl. unbanded, band < 0) and any length difference: same recurrence,
// DP row kept in an HBM/L2 scratch row [k][lane] instead of registers.  Used only when the band
// window does not fit the register classes (default nwalign() calls, exotic BAND_SIZE).
template <bool PAIRS>
__global__ __launch_bounds__(256) void k_nw_gen(NwArgs a, int Wgen) {
  extern __shared__ double s_err[];
  for (int i = threadIdx.x; i < 16 * a.ap.ncol; i += blockDim.x) s_err[i] = a.err[i];
  __syncthreads();
  const SampleDev &S = a.S;
  const int lane = threadIdx.x & 63, gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int npw = (2 * Wgen + 31) / 32;
  uint32_t *ptr = a.ptr_scr + (size_t)gwave * a.ptr_wpw;
  uint32_t *tsc = a.t_scr + (size_t)gwave * a.t_wpw;
  int32_t *drow = a.row_scr + (size_t)gwave * a.row_wpw;
  const int nwork = a.nwork_dev ? *a.nwork_dev : a.nwork_host;
  const int SENT = a.ap.sentinel, MATCH = a.ap.match, MISMATCH = a.ap.mismatch, GAP = a.ap.gap;
  for (int chunk = gwave; chunk * 64 < nwork; chunk += nwaves) {
    const int idx = chunk * 64 + lane;
    const int c = PAIRS ? a.pair_centre[idx < nwork ? idx : nwork - 1] : (a.chunk_centre ? a.chunk_centre[chunk] : a.centre);
    int r = idx < nwork ? a.work[idx] : -1;
    const bool active = r >= 0;
    if (!active) r = c;
    const int L1 = S.len[c], L2 = S.len[r];
    const int B = a.ap.band < 0 ? (L1 > L2 ? L1 : L2) : a.ap.band;   // band < 0: full matrix
    const int lband = B + (L1 > L2 ? L1 - L2 : 0), rband = B + (L2 > L1 ? L2 - L1 : 0);
    const int W = lband + rband + 1;
    const uint32_t *crow = S.seq2 + (size_t)c * S.W2, *rrow = S.seq2 + (size_t)r * S.W2;
    // (C_nwvec on letters outside ACGT: the high two bits of each position's code in a second plane of rows, AlignParams::hi_off)
    const int HI = PAIRS ? a.ap.hi_off : 0;
    const uint32_t *crowh = crow + (size_t)HI * S.W2, *rrowh = rrow + (size_t)HI * S.W2;
    const bool EF = a.ap.endsfree != 0, HOMO = EF && a.ap.homo_gap != GAP;   // (see k_nw)
    const int HG = a.ap.homo_gap;
    for (int k = 0; k < Wgen; k++) {
      const int j = k - lband;
      drow[(size_t)k * 64 + lane] = (j >= 0 && j <= rband && j <= L2) ? (EF ? 0 : j * GAP) : SENT;
    }
    for (int i = 1; i <= L1; i++) {
      const uint32_t cb = base_at(crow, i - 1) | (HI ? base_at(crowh, i - 1) << 2 : 0u);
      const int gapL = (EF && i == L1) ? 0 : GAP;
      const int gapU = (HOMO && homo_at(crow, L1, i - 1)) ? HG : GAP;
      const int kzero = lband - i, kend = L2 - i + lband;
      const int khi = kend < W - 1 ? kend : W - 1;
      int leftv = SENT;
      uint32_t pwv = 0;
      int dk = drow[lane];
      for (int k = 0; k < Wgen; k++) {
        const int upn = (k + 1 < Wgen) ? drow[(size_t)(k + 1) * 64 + lane] : SENT;
        const int j = i - lband + k;
        const bool valid = (k > kzero) && (k <= khi);
        int v = (k == kzero) ? (EF ? 0 : i * GAP) : SENT;
        uint32_t p = 0;
        if (valid) {
          const uint32_t rb = base_at(rrow, j - 1) | (HI ? base_at(rrowh, j - 1) << 2 : 0u);
          const int diag = dk + (rb == cb ? MATCH : MISMATCH);
          const int up = upn + ((EF && k == kend) ? 0 : gapU);
          const int left = leftv + ((HOMO && !(EF && i == L1) && homo_at(rrow, L2, j - 1)) ? HG : gapL);
          const bool t1 = left >= diag;
          const int e1 = t1 ? left : diag;
          const bool t2 = up >= e1;
          v = t2 ? up : e1;
          p = t2 ? 3u : (t1 ? 2u : 1u);
        }
        drow[(size_t)k * 64 + lane] = v;
        leftv = v;
        dk = upn;
        pwv |= p << ((k & 15) << 1);
        if ((k & 15) == 15 || k == Wgen - 1) { ptr[((size_t)i * npw + (k >> 4)) * 64 + lane] = pwv; pwv = 0; }
      }
    }
    nw_traceback_lambda<0>(a, s_err, lane, idx, active, r, c, L1, L2, lband, ptr, tsc, npw, chunk);
  }
}

// ------------------------------------------------------------------------------------------------
// Low-latency variant for the per-round batches (a few thousand alignments): the same recurrence
// swept by ANTI-DIAGONALS with GL lanes per alignment (64/GL alignments per wave).  Lane g owns
// band cells k = 2g, 2g+1; on anti-diagonal t = i + j the cells with k = t + lband (mod 2) are
// live, so every lane updates exactly one cell per step: diag = its own cell two steps ago, left /
// up = the neighbouring cells one step ago — one of them its own other cell, the other fetched
// from the neighbour lane with a DPP wave shift.  Pointers (2 bits per lane per step) stay in LDS.
// The traceback (one lane per alignment) skips whole diagonal runs per LDS word and emits run
// descriptors; all lanes then turn runs into per-position error-model factors in LDS, and one lane
// multiplies them in raw-position order (pval.cpp:188-192) — bit-identical to k_nw.
// guard entries either side of the staged sequences (cell indices run about -GL .. len+GL)
static inline int ad_pad(int GL) { return GL + 8; }
// guard words in front of a staged sequence in k_nw_ad (>= GL + 11, a multiple of 4: position 0 is 16-byte aligned)
static __host__ __device__ constexpr int ad_guard(int GL) { return (GL + 11 + 3) & ~3; }
constexpr int AD_RCAP = 64;   // run descriptors buffered per alignment between traceback chunks

// Traceback pointers of k_nw_ad: every step shifts a TWO-bit move code into the TOP of the lane's pointer word,
//   0 : the cell came from above (up; ties go to up first)      1 : from the left (it wins its tie with the diagonal)      2 : diagonal,
// so after the 16 steps of a block step s sits at bits 2s, 2s + 1.  The codes are in the reference's tie order, which lets
// the steady state of the default scores carry them as TAGS in the two low bits of the three candidates: the minimum of the
// tagged candidates is the cell value AND the move (ad_step_k) - no compare, no select, no subtraction.
constexpr uint32_t AD_UP = 0u, AD_LEFT = 1u, AD_DIAG = 2u;
static __device__ __forceinline__ uint32_t ad_ptr_code(bool from_up, bool left_ge_diag) { return from_up ? AD_UP : (left_ge_diag ? AD_LEFT : AD_DIAG); }

// one anti-diagonal step of one lane's live cell.  PAR is the cell parity (k = 2g + PAR).
// LEAN: steady-state step — every in-band cell of the wave is an interior cell away from the last
// row/column, so the matrix-edge logic (axis cells, free end gaps) is compiled out.
// Out-of-band cells: a LEAN step does not select the sentinel into them, it ADDS AD_OOB instead of the gap penalty (gsel,
// a per-lane register: GAP for an in-band cell, AD_OOB for the others), so they sit about 10^6 below the band and never
// win a max in an in-band neighbour — one select less on every step.
constexpr int AD_OOB = -(1 << 20);
// (round 2 kept three steady-state formulations behind a knob - sentinel select, additive mask, additive mask + v_max3:
// 69.7 / 66.9 / 63.3 us at 8 700 alignments, profiles/r02o - only the last one is left.)  vnext = the one base that changes for the
// next step (raw base after an even cell, centre base after an odd one), loaded by the caller.
// HOMO (nwalign_endsfree_homo, nwalign_endsfree.cpp:220-396): a gap opposite a base of a homopolymer run of >= 3 costs HG
// instead of GAP.  The staged base words carry that flag in bit 31 (ad_base_word): the up move consumes centre base i - 1 (cb),
// the left move raw base j - 1 (rb).
template <int GL, int PAR, bool DEF, bool LEAN, bool EDGE, bool HOMO = false>
static __device__ __forceinline__ void ad_step(int &d0, int &d1, int &i, int &j, uint32_t &cb, uint32_t &rb, uint32_t &pw,
                                               uint32_t vnext, int fs, bool g_first, bool g_last,
                                               bool kok, int gsel, int L1, int L2, int SENT_, int MATCH_, int MISMATCH_, int GAP_,
                                               int HG_ = 0) {
  // DEF: the reference's default scoring (MATCH 5, MISMATCH -4, GAP -8, vectorized sentinel) as literals
  const int SENT = DEF ? -32760 : SENT_, MATCH = DEF ? 5 : MATCH_, MISMATCH = DEF ? -4 : MISMATCH_, GAP = DEF ? -8 : GAP_;
  if (HOMO) {
    // the same cell with the two gap addends chosen per base; equality looks past the flag bit
    const bool same = ((cb ^ rb) << 1) == 0u;
    const bool hu = (cb >> 31) != 0u, hl = (rb >> 31) != 0u;
    int left_src, up_src, own;
    if (PAR == 0) {
      const int lft = gcn_wave_shr1<false>(SENT, d1);
      own = d0; left_src = (EDGE && g_first) ? SENT : lft; up_src = d1;
    } else {
      const int upn = gcn_wave_shl1<false>(SENT, d0);
      own = d1; left_src = d0; up_src = (EDGE && g_last) ? SENT : upn;
    }
    const int diag = own + (same ? MATCH : MISMATCH);
    const int up = up_src + ((!LEAN && j == L2) ? 0 : (hu ? HG_ : GAP));      // free moves along the last column
    const int left = left_src + ((!LEAN && i == L1) ? 0 : (hl ? HG_ : GAP));  // ... and the last row
    const bool t1 = left >= diag;
    const int e1 = max(left, diag);
    const bool t2 = up >= e1;
    const int e = max(up, e1);
    int val;
    uint32_t p = t2 ? 3u : (t1 ? 2u : 1u);
    if (LEAN) {
      val = kok ? e : SENT;
    } else {
      const bool interior = kok && ((unsigned)(i - 1) < (unsigned)L1) && ((unsigned)(j - 1) < (unsigned)L2);
      val = interior ? e : (kok ? 0 : SENT);
      if (!interior) p = (i <= 0 ? 2u : 3u);
    }
    if (PAR == 0) { d0 = val; rb = vnext; j++; } else { d1 = val; cb = vnext; i++; }
    pw = gcn_push_low2(pw, 3u - p);
    return;
  }
  if (LEAN && !EDGE) {
    // steady state, band inside the lane group: the DPP neighbour needs no masking (lanes without a source read 0, they
    // are out of band), the fetch folds into the add, the two max into one v_max3
    const int nb = PAR == 0 ? gcn_wave_shr1<true>(0, d1)     // lane-1's odd cell (wave_shr:1)
                            : gcn_wave_shl1<true>(0, d0);    // lane+1's even cell (wave_shl:1)
    const int own = PAR == 0 ? d0 : d1, other = PAR == 0 ? d1 : d0;
    const int diag = own + (cb == rb ? MATCH : MISMATCH);
    const int left = (PAR == 0 ? nb : other) + gsel, up = (PAR == 0 ? other : nb) + gsel;
    const int e = gcn_max3(left, diag, up);                  // (spelled out: the compiler would keep the inner max for t2)
    const bool t2 = up == e;                                  // up >= max(left, diag)  <=>  the maximum IS up
    const bool t1 = left >= diag;
    if (PAR == 0) { d0 = e; rb = vnext; j++; } else { d1 = e; cb = vnext; i++; }
    pw = gcn_push_low2(pw, ad_ptr_code(t2, t1));
    return;
  }
  int left_src, up_src, own;
  if (PAR == 0) {
    const int lft = gcn_wave_shr1<false>(SENT, d1);   // lane-1's odd cell (wave_shr:1)
    own = d0; left_src = (EDGE && g_first) ? SENT : lft; up_src = d1;
  } else {
    const int upn = gcn_wave_shl1<false>(SENT, d0);   // lane+1's even cell (wave_shl:1)
    own = d1; left_src = d0; up_src = (EDGE && g_last) ? SENT : upn;
  }
  const int diag = own + (cb == rb ? MATCH : MISMATCH);
  const int up = up_src + ((!LEAN && j == L2) ? 0 : GAP);      // free moves along the last column
  const int left = left_src + ((!LEAN && i == L1) ? 0 : GAP);  // ... and the last row
  const bool t1 = left >= diag;
  const int e1 = max(left, diag);
  const bool t2 = up >= e1;
  const int e = max(up, e1);
  int val;
  uint32_t p = t2 ? 3u : (t1 ? 2u : 1u);
  if (LEAN) {
    val = kok ? e : SENT;
  } else {
    const bool interior = kok && ((unsigned)(i - 1) < (unsigned)L1) && ((unsigned)(j - 1) < (unsigned)L2);
    // cells before/after the matrix are never read by a live cell, so only the band edge needs the sentinel
    val = interior ? e : (kok ? 0 : SENT);
    if (!interior) p = (i <= 0 ? 2u : 3u);                     // first row: left, first column: up
  }
  if (PAR == 0) { d0 = val; rb = vnext; j++; } else { d1 = val; cb = vnext; i++; }
  pw = gcn_push_low2(pw, 3u - p);                           // (up / left / diagonal as ad_ptr_code)
}

// The same step for the reference's DEFAULT scores (match 5, mismatch -4, gap -8) in the COST domain K = 4 (5 t - 2 H) on
// anti-diagonal t = i + j: every path into a cell has the same t, so arg-max and ties are exactly those of H, and
//   a match costs 0, a mismatch 72, a gap 84, a free move along the last row / column 20, an axis cell is 20 t, out of band is BIG.
// Bases are staged as one word each, 36 << (8 * code): the sum of absolute byte differences of two such words is 0 for equal
// bases and 72 otherwise, so diag = own + substitution cost is ONE v_sad_u8 (it was compare + select + add), and the
// three-way minimum is one v_min3.  Out-of-band cells take BIG instead of the gap cost (additive mask, as above).
// All costs are multiples of 4, which leaves the two low bits of a candidate for its move code (AD_UP < AD_LEFT < AD_DIAG,
// the reference's tie order): cells are kept as value + AD_DIAG, so the diagonal candidate is tagged by the v_sad_u8
// itself and the two gap addends carry (code - AD_DIAG); v_min3 of the three tagged candidates then yields value and move
// at once - equal values are separated by their tags exactly as the reference breaks the tie.  One v_alignbit_b32 moves
// the tag into the pointer word, one v_and_or_b32 re-tags the cell: 6 vector instructions per cell (round 2: 11.25).
constexpr int ADK_BIG = 1 << 24, ADK_MIS = 72, ADK_GAP = 84, ADK_FREE = 20, ADK_HOT = 36;
// FAST (the pointer-free pass, k_nw_ad<.., FAST>): no pointer word.  `pw` is then the cell's DIAGONAL ACCUMULATOR - the AND of the
// move codes of every interior cell this lane has computed on the diagonal its cell of this parity runs along (a lane's cell
// k' = 2g + PAR keeps its j - i for the whole sweep): bit 1 stays set exactly while every one of them was AD_DIAG (0b10; the
// other codes are 0b00 and 0b01) - and `ep` receives the move code of the diagonal's LAST interior cell, the one on the last row
// or the last column.  That is all the traceback of an alignment without interior gaps needs (fast_walk below).
template <int GL, int PAR, bool LEAN, bool EDGE, bool FAST = false>
static __device__ __forceinline__ void ad_step_k(int &d0, int &d1, int &i, int &j, uint32_t &cb, uint32_t &rb, uint32_t &pw,
                                                 uint32_t vnext, int fs, bool g_first, bool g_last, bool kok, int gsel, int gsel_nb,
                                                 int L1, int L2, uint32_t *ep = nullptr) {
  if (LEAN) {
    // gsel_nb is the addend of the neighbour LANE's cell: the gap cost, or BIG where that neighbour must not be seen (out of
    // band, or - when the band fills the group's cells, EDGE - the last lane of the group before / the first of the next one;
    // a lane without any source reads 0 there).  No select, no masking instruction: just a second per-lane constant.
    const int nb = PAR == 0 ? gcn_wave_shr1<true>(0, d1)     // lane-1's odd cell (wave_shr:1)
                            : gcn_wave_shl1<true>(0, d0);    // lane+1's even cell (wave_shl:1)
    const int own = PAR == 0 ? d0 : d1, other = PAR == 0 ? d1 : d0;
    const int diag = gcn_sad_u8(cb, rb, own);
    const int left = PAR == 0 ? nb + gsel_nb : other + gsel, up = PAR == 0 ? other + gsel : nb + gsel_nb;
    const int et = gcn_min3(left, diag, up);                  // value | move code
    if (FAST) pw &= (uint32_t)et;                             // (one v_and in place of the v_alignbit; nothing is ever flushed)
    else pw = gcn_push_low2(pw, (uint32_t)et);
    const int e = (et & ~3) | (int)AD_DIAG;
    if (PAR == 0) { d0 = e; rb = vnext; j++; } else { d1 = e; cb = vnext; i++; }
    return;
  }
  int left_src, up_src, own;
  if (PAR == 0) {
    const int lft = gcn_wave_shr1<false>(ADK_BIG, d1);        // (every source carries the same + AD_DIAG: the compares do not see it)
    own = d0; left_src = (EDGE && g_first) ? ADK_BIG : lft; up_src = d1;
  } else {
    const int upn = gcn_wave_shl1<false>(ADK_BIG, d0);
    own = d1; left_src = d0; up_src = (EDGE && g_last) ? ADK_BIG : upn;
  }
  const int diag = own + (cb == rb ? 0 : ADK_MIS);
  const int up = up_src + (j == L2 ? ADK_FREE : ADK_GAP);      // free moves along the last column
  const int left = left_src + (i == L1 ? ADK_FREE : ADK_GAP);  // ... and the last row
  const bool t1 = left <= diag;
  const int e1 = min(left, diag);
  const bool t2 = up <= e1;
  const int e = min(up, e1);
  int val;
  uint32_t p = t2 ? 3u : (t1 ? 2u : 1u);
  {
    const bool interior = kok && ((unsigned)(i - 1) < (unsigned)L1) && ((unsigned)(j - 1) < (unsigned)L2);
    val = interior ? e : (kok ? ADK_FREE * (i + j) + (int)AD_DIAG : ADK_BIG);  // axis cells: H = 0
    if (!interior) p = (i <= 0 ? 2u : 3u);                     // first row: left, first column: up
    if (FAST && interior) {
      pw &= 3u - p;
      if (i == L1 || j == L2) *ep = 3u - p;                    // the diagonal's last cell: where the walk back enters or leaves it
    }
  }
  if (PAR == 0) { d0 = val; rb = vnext; j++; } else { d1 = val; cb = vnext; i++; }
  if (!FAST) pw = gcn_push_low2(pw, 3u - p);                 // (up / left / diagonal as ad_ptr_code)
}

// LDS geometry of k_nw_ad, shared by host and device.  Per WAVE: the staged centre (every alignment of a wave has the same
// centre), one word per base with guard words either side.  Per alignment: run descriptors and the staged raw (one word per
// base).  The expansion of the traceback overwrites each raw base IN PLACE with the byte offset of that position's error-model
// factor in the LDS copy of err (a position belongs to exactly one run, and nothing reads the bases after the DP), and reads
// the qualities from global memory where it needs them: round 3 kept a u16 offset row and a staged quality row per alignment
// besides the words - 40 % of the per-alignment LDS, which at 1 500 nt held the kernel at ONE block per CU (one wave per SIMD:
// 6.85 cycles per dependent VALU instruction instead of ~4.2 at four; profiles/r05a_cfg5_summary.md).
// The 2-bit traceback pointers are NOT here: they go to the HBM ring SampleDev::ad_ptr, one coalesced 256-byte store per wave
// per 16 steps (they took 2.3 of the 4.1 KB per alignment of round 2's kernel and, with the fp64 factor array that aliased
// them, held the kernel at three blocks per CU).
struct AdGeom {
  int GL, APW, NCOL;        // lanes per alignment, alignments per wave, pointer columns per alignment
  int edge;                 // 1: group-boundary lanes must mask their DPP neighbour (band fills the group's cells)
  int nwords;               // 16-step blocks of a sweep (pointer words per lane)
  int seqwords;             // words per staged sequence incl. guards (multiple of 4)
  int per_al_bytes, per_wave_bytes;   // per_wave_bytes includes the wave's own centre words unless shared_c
  int shared_c;             // the launch has one centre for all its work (no per-chunk centres): staged once per block
  int block_bytes;          // LDS behind the err table
  int bmw;                  // bimera mode (k_nw_ad<.., LR>): words per column bitmap (three of them behind the raw words), else 0
};
// Lane g of a group owns cells k' = 2g, 2g+1; an alignment's band cell k sits at k' = k + o.  The origin shift o makes
// lband + o even, so every alignment of a wave is in phase (even cells live on even steps) whatever its length
// difference.  When the group has room (W + 4 <= 2 GL) o is 2 or 3: the first two cells and the last cell of every
// group are then never in band, always hold the sentinel, and the cross-group DPP reads need no masking.
static __host__ __device__ inline AdGeom ad_geom(int band, int maxlen, int minlen, int shared_c = 0, int lr = 0) {
  AdGeom G;
  const int W = 2 * band + (maxlen - minlen) + 1;
  G.GL = W + 1 <= 42 ? 21 : (W + 1 <= 64 ? 32 : 64);   // 21 lanes x 2 cells cover the default band (W = 33): 3 alignments per wave
  G.APW = 64 / G.GL;
  G.edge = (W + 4 > 2 * G.GL) ? 1 : 0;
  G.NCOL = (W + (G.edge ? 1 : 3) + 1) / 2;
  G.nwords = (2 * maxlen + 1 + 15) / 16;
  G.seqwords = (maxlen + 2 * ad_guard(G.GL) + 3) & ~3;
  G.bmw = lr ? (((2 * maxlen + 2 + 31) / 32 + 3) & ~3) : 0;   // one bit per alignment column (at most 2 maxlen of them)
  G.per_al_bytes = AD_RCAP * 4 + 4 * G.seqwords + 12 * G.bmw;
  G.shared_c = shared_c;
  G.per_wave_bytes = (shared_c ? 0 : 4 * G.seqwords) + G.APW * G.per_al_bytes;
  G.block_bytes = (shared_c ? 4 * G.seqwords : 0) + 4 * G.per_wave_bytes;
  return G;
}

// a staged base: ADK_HOT << (8 * code); HOMO: bit 31 = the position lies in a homopolymer run of >= 3 (homo_at)
template <bool HOMO>
static __device__ __forceinline__ uint32_t ad_base_word(const uint32_t *__restrict__ row, int len, int p) {
  const uint32_t w = (uint32_t)ADK_HOT << (base_at(row, p) << 3);
  return HOMO ? (w | (homo_at(row, len, p) << 31)) : w;
}

// ---- bimera mode of k_nw_ad (chimera.cpp:211-293): the alignment as three bitmaps over its columns, bit t = column t counted
//      from the END of the alignment (the order the traceback finds them in) ----
static __device__ __forceinline__ bool bm_bit(const uint32_t *bm, int t) { return ((bm[t >> 5] >> (t & 31)) & 1u) != 0u; }
// smallest t' in [t, len) whose bit equals `want`, or len
static __device__ __forceinline__ int bm_next(const uint32_t *bm, int t, int len, bool want) {
  while (t < len) {
    uint32_t w = bm[t >> 5];
    if (!want) w = ~w;
    w >>= (t & 31);
    if (w) { const int r = t + __builtin_ctz(w); return r < len ? r : len; }
    t = (t | 31) + 1;
  }
  return len;
}
// largest t' in [0, t] whose bit equals `want`, or -1
static __device__ __forceinline__ int bm_prev(const uint32_t *bm, int t, bool want) {
  while (t >= 0) {
    uint32_t w = bm[t >> 5];
    if (!want) w = ~w;
    w <<= 31 - (t & 31);
    if (w) return t - __builtin_clz(w);
    t = (t & ~31) - 1;
  }
  return -1;
}
// set bits at lo..hi (inclusive)
static __device__ __forceinline__ int bm_count(const uint32_t *bm, int lo, int hi) {
  int n = 0;
  for (int t = lo; t <= hi;) {
    uint32_t w = bm[t >> 5] >> (t & 31);
    const int span = min(32 - (t & 31), hi - t + 1);
    if (span < 32) w &= (1u << span) - 1u;
    n += __builtin_popcount(w);
    t += span;
  }
  return n;
}
// get_lr (chimera.cpp:243-293) and get_ham_endsfree (:211-239) of one alignment from its bitmaps: NE = the column is not a pair
// of equal bases, B2 = gap in the query (the chunk's centre), B3 = gap in the parent; len columns.  Forward column c is bit
// len - 1 - c.  The statement-by-statement form of the same rules is k_bimera_lr below (on move strings).
static __device__ __forceinline__ void bimera_lr_bits(const uint32_t *NE, const uint32_t *B2, const uint32_t *B3, int len, int one_off,
                                                      int ms, int32_t *__restrict__ o) {
  // left: scan in over the columns before the query starts, ends-free until the parent starts, covered until a mismatch
  int pos = len - 1 - bm_prev(B2, len - 1, false), left = 0;
  if (pos < ms) {
    const int p3 = min(len - 1 - bm_prev(B3, len - 1 - pos, false), ms);
    left = p3 - pos; pos = p3;
  }
  { const int pn = len - 1 - bm_prev(NE, len - 1 - pos, true); left += pn - pos; pos = pn; }
  int left_oo = 0;
  if (one_off) {
    left_oo = left;
    pos++;
    if (pos < len && !bm_bit(B2, len - 1 - pos)) left_oo++;
    if (pos < len) left_oo += (len - 1 - bm_prev(NE, len - 1 - pos, true)) - pos;
  }
  // right: the same from the end (the reference compares `pos > len - max_shift` in size_t: never true when len < max_shift)
  int t = bm_next(B2, 0, len, false), right = 0;
  if (len >= ms && t < ms - 1) {
    const int t3 = min(bm_next(B3, t, len, false), ms - 1);
    right = t3 - t; t = t3;
  }
  { const int tn = bm_next(NE, t, len, true); right += tn - t; t = tn; }
  int right_oo = 0;
  if (one_off) {
    right_oo = right;
    t++;
    if (t < len && !bm_bit(B2, t)) right_oo++;
    if (t < len) right_oo += bm_next(NE, t, len, true) - t;
  }
  // hamming distance between the two end-gap runs
  int is = 0, ntrail = 0;
  if (bm_bit(B2, len - 1)) is = len - 1 - bm_prev(B2, len - 1, false);
  else if (bm_bit(B3, len - 1)) is = len - 1 - bm_prev(B3, len - 1, false);
  if (bm_bit(B2, 0)) ntrail = bm_next(B2, 0, len, false);
  else if (bm_bit(B3, 0)) ntrail = bm_next(B3, 0, len, false);
  o[0] = left; o[1] = right; o[2] = left_oo; o[3] = right_oo;
  o[4] = len - 1 - is >= ntrail ? bm_count(NE, ntrail, len - 1 - is) : 0;
}

// FAST: the pointer-free pass of a batch compare (launch_nw_ad).  Most pairs a round aligns differ by substitutions only (or by
// that and a free end gap): their walk back from (L1, L2) - the reference takes the diagonal only where it is STRICTLY best
// (nwalign_endsfree.cpp:146-156) and walks pointer by pointer (:164-188) - is a run of free moves along the last column or row, then
// ONE diagonal from its last cell all the way to the first row / column, then the forced moves along that border.  Whether a
// diagonal is walked whole is the AND of its cells' move codes, and where the walk enters it is decided by the move codes of the
// diagonals' LAST cells: two registers per lane instead of 2 bits per cell in memory (the pointer ring took 286 MB of stores and
// 2 x 93 MB of fetches per batch launch at 10^6 uniques - 19x the launch's algorithmic bytes, profiles/r08c_traffic_cfg3.json),
// no traceback, and the expansion / product work on the same run descriptors as ever.  A pair that is not of this form is
// appended to the retry list of its batch position (NwArgs::retry_list) and aligned by the full kernel in a second launch.
template <int GL, bool DEF, bool EDGE, bool LR = false, bool HOMO = false, bool FAST = false>
__global__ __launch_bounds__(256, 4) void k_nw_ad(NwArgs a, const int32_t *__restrict__ gl_work, const int32_t *__restrict__ gl_nwork_dev,
                                               AdGeom G) {
  static_assert(!FAST || (DEF && !LR && !HOMO), "the pointer-free pass exists for the default scores of the denoising path");
  constexpr int APW = 64 / GL;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double *s_err = s_dyn;
  const int nerr = 16 * a.ap.ncol;
  const bool batch = a.batch_on != nullptr;
  int bk[KB_MAX + 1];                                        // batch mode: first block-sized work slice of every batch position
  {   // the per-round grid is sized for the worst case (the batch size is only known on the device): blocks past the
      // work leave before touching anything, also when a speculative round turned out to have no centre
    if (a.stop_dev && *a.stop_dev != 0) return;
    bk[0] = 0;
    if (batch) {
      const int nb = *a.batch_on;
      if (nb <= 0) return;
      if (a.fast_ctl) {
        const bool off = a.fast_ctl[2] != 0ull;
        if (FAST) {
          if (off) return;
          if (blockIdx.x == 0 && threadIdx.x == 0) {
            unsigned long long tot = 0;
            for (int k = 0; k < KB_MAX; k++) tot += k < nb ? (unsigned long long)a.batch_n[k] : 0ull;
            atomicAdd(&a.fast_ctl[1], tot);
          }
        } else if (off && a.alt_list) { a.batch_list = a.alt_list; a.batch_n = a.alt_n; }
      }
#pragma unroll
      for (int k = 0; k < KB_MAX; k++) bk[k + 1] = bk[k] + (k < nb ? (a.batch_n[k] + 4 * APW - 1) / (4 * APW) : 0);   // (the gapless rows: k_gapless_batch)
      if ((int)blockIdx.x >= bk[KB_MAX]) return;
    } else {
      const int n_all = (a.nwork_dev ? *a.nwork_dev : a.nwork_host) + (gl_work ? *gl_nwork_dev : 0);
      if ((int)blockIdx.x * 4 * APW >= n_all) return;
      if (a.centre_dev && *a.centre_dev < 0) return;
    }
  }
  for (int i = threadIdx.x; i < nerr; i += blockDim.x) s_err[i] = a.err[i];
  const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // alignment slot in the wave / lane in the group.  With GL = 21 lane 63 is a ghost: it rides along the DP as an
  // extra, always-out-of-band lane of the last group and is excluded from everything else.
  const bool ghost = lane / GL >= APW;
  const int al = ghost ? APW - 1 : lane / GL, g = ghost ? GL : lane % GL;
  // LDS behind err: [centre words, once per block when the launch has ONE centre (a round) | once per wave (final pass,
  // births: a centre per chunk)]; per alignment: [run descriptors][raw words][factor offsets u16][quals]
  uint8_t *blk0 = (uint8_t *)(s_dyn + nerr);
  const int cw_bytes = 4 * G.seqwords;
  uint8_t *wbase = blk0 + (G.shared_c ? cw_bytes : 0) + (size_t)wib * G.per_wave_bytes;
  uint32_t *cwd = (uint32_t *)(G.shared_c ? blk0 : wbase) + ad_guard(GL); // centre base p as ADK_HOT << (8 * code)
  uint8_t *abase = wbase + (G.shared_c ? 0 : cw_bytes) + (size_t)al * G.per_al_bytes;
  uint32_t *runs = (uint32_t *)abase;
  uint32_t *rwd = (uint32_t *)(abase + AD_RCAP * 4) + ad_guard(GL);      // raw base p, same encoding; after the expansion: byte offset
                                                                        // into s_err of the position's error-model factor
  uint32_t *bm = (uint32_t *)(abase + AD_RCAP * 4 + 4 * G.seqwords);     // bimera mode: the column bitmaps NE | B2 | B3
  const SampleDev &S = a.S;
  if (G.shared_c && !batch) {                                          // the launch's one centre, staged by the whole block
    const int cv = a.centre_dev ? *a.centre_dev : a.centre;
    if (cv >= 0) {
      const int Lc = S.len[cv];
      for (int p = threadIdx.x; p < Lc; p += 256) cwd[p] = ad_base_word<HOMO>(S.seq2 + (size_t)cv * S.W2, Lc, p);
    }
  }
  __syncthreads();
  const int gwave = blockIdx.x * 4 + wib;
  uint32_t *pg = S.ad_ptr + (size_t)gwave * S.ad_wpw;      // this wave's slot of the pointer ring: [16-step block][lane]
  int n_nw = batch ? 0 : (a.nwork_dev ? *a.nwork_dev : a.nwork_host);
  const int n_gl = batch ? 0 : (gl_work ? *gl_nwork_dev : 0);   // gapless items ride along: same factors/product tail
  int nwork = n_nw + n_gl;
  const int SENT = a.ap.sentinel, MATCH = a.ap.match, MISMATCH = a.ap.mismatch, GAP = a.ap.gap, B = a.ap.band, HGAP = a.ap.homo_gap;
  const int centre_v = batch ? 0 : (a.centre_dev ? *a.centre_dev : a.centre);
  if (centre_v < 0 && !a.chunk_centre) return;
  const int32_t *wl = a.work, *gll = gl_work;
  size_t out_off = 0;
  int kcur = -1;
  // one iteration = one work slice of the block (4 waves x APW alignments); it = blockIdx.x + j gridDim.x, so a wave's
  // chunk index it * 4 + wib runs over gwave + j nwaves
  for (int it = blockIdx.x;; it += gridDim.x) {
    int chunk, c;
    if (batch) {
      if (it >= bk[KB_MAX]) break;                           // (block-uniform: the barriers below are safe)
      int k = 0, b0 = 0;                                     // the last position whose first slice is <= it (bk is nondecreasing)
#pragma unroll
      for (int q = 1; q < KB_MAX; q++) if (it >= bk[q]) { k = q; b0 = bk[q]; }
      c = a.batch_centre[k];
      if (k != kcur) {                                       // next batch position: its centre replaces the staged one
        __syncthreads();
        const int Lc = S.len[c];
        for (int p = threadIdx.x; p < Lc; p += 256) cwd[p] = ad_base_word<HOMO>(S.seq2 + (size_t)c * S.W2, Lc, p);
        __syncthreads();
        kcur = k;
        wl = a.batch_list + (size_t)k * a.batch_stride; gll = a.batch_list + (size_t)(KB_MAX + k) * a.batch_stride;
        n_nw = a.batch_n[k]; nwork = n_nw;
        out_off = ((size_t)*a.batch_bbuf * KB_MAX + k) * a.batch_stride;
      }
      chunk = (it - b0) * 4 + wib;
    } else {
      chunk = it * 4 + wib;
      if (chunk * APW >= nwork) break;
      c = a.chunk_centre ? a.chunk_centre[chunk] : centre_v;
    }
    const int idx = chunk * APW + al;
    int r = idx < n_nw ? wl[idx] : (idx < nwork ? gll[idx - n_nw] : -1);
    const bool gapless = idx >= n_nw;
    const bool active = r >= 0;
    if (!active) r = c;
    const int L1 = S.len[c], L2 = S.len[r];
    const int lband = B + (L1 > L2 ? L1 - L2 : 0), rband = B + (L2 > L1 ? L2 - L1 : 0);
    const int W = lband + rband + 1;                       // <= 2*NCOL
    const int T = (gapless || !active) ? -1 : L1 + L2;     // idle / gapless slots run no DP steps of their own
    // stage both sequences (one base per byte, guard bytes either side) and the raw's qualities
    if (!G.shared_c)                                       // the chunk's centre, by all lanes of the wave
      for (int p = lane; p < L1; p += 64) cwd[p] = ad_base_word<HOMO>(S.seq2 + (size_t)c * S.W2, L1, p);
    if (!ghost)
      for (int p = g; p < L2; p += GL) rwd[p] = ad_base_word<HOMO>(S.seq2 + (size_t)r * S.W2, L2, p);
    if (LR && !ghost)
      for (int w = g; w < 3 * G.bmw; w += GL) bm[w] = 0;
    int ecol = 0;                                          // bimera mode: alignment columns found so far (from the end)
    const uint8_t *qrow = S.qual + (size_t)r * S.LQ;        // the raw's qualities (read where the expansion needs them)
    // aligned view of the unique on its centre (final pass / birth substitutions): centre positions facing a gap stay 0
    const size_t vr = a.view_by_chunk ? (size_t)chunk : (size_t)r;
    if (a.view && active && !ghost)
      for (int p = g; p < L1; p += GL) a.view[vr * a.LV + p] = 0;
    int Tmax = T;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));
    const int dbg = a.moves_stride;
    Tmax = gcn_readfirstlane(Tmax);
    const int org = (EDGE ? 0 : 2) + (lband & 1), lbo = lband + org;   // origin shift: lbo is even
    uint32_t acc0 = ~0u, acc1 = ~0u, ep0 = 3u, ep1 = 3u;   // FAST: diagonal accumulators / last-cell move codes of the lane's two cells
    if (Tmax >= 0 && !(dbg & 1)) {
      int d0 = SENT, d1 = SENT;
      uint32_t pw = 0;
      int i = (lbo >> 1) - g, j = -i;                        // the lane's even cell on step 0
      uint32_t cb = cwd[i - 1], rb = rwd[j - 1];
      const bool g_first = g == 0, g_last = ghost || g == GL - 1;
      const bool kok0 = !ghost && 2 * g >= org && 2 * g < W + org, kok1 = !ghost && 2 * g + 1 >= org && 2 * g + 1 < W + org;
      // (cost domain: an even cell's own other cell is its UP source and the neighbour lane's its LEFT one, an odd cell's the
      //  other way round; the addends carry the move code relative to the cells' + AD_DIAG)
      constexpr int KUP = ADK_GAP + (int)AD_UP - (int)AD_DIAG, KLEFT = ADK_GAP + (int)AD_LEFT - (int)AD_DIAG;
      const int gs0 = DEF ? (kok0 ? KUP : ADK_BIG) : (kok0 ? GAP : AD_OOB), gs1 = DEF ? (kok1 ? KLEFT : ADK_BIG) : (kok1 ? GAP : AD_OOB);
      // ... and for the cell fetched from the neighbour lane: hidden across a group boundary when the band fills the group
      const int gn0 = DEF ? ((EDGE && g_first) || !kok0 ? ADK_BIG : KLEFT) : gs0, gn1 = DEF ? ((EDGE && g_last) || !kok1 ? ADK_BIG : KUP) : gs1;
      if (DEF) { d0 = ADK_BIG + (int)AD_DIAG; d1 = ADK_BIG + (int)AD_DIAG; }                 // (the default scores run in the cost domain: ad_step_k)
      // Steady state [tA, tB): every in-band cell of every alignment in the wave is interior and off the
      // last row / column (i >= 1, j >= 1, i < L1, j < L2 for all k in the band).
      int tA = (lband > rband ? lband : rband) + 2, tB = min(2 * L1 - lband, 2 * L2 - rband);
      if (T < 0) { tA = 0; tB = 0x3FFFFFFF; }               // idle / gapless slot: no constraint
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { tA = max(tA, __shfl_xor(tA, o, 64)); tB = min(tB, __shfl_xor(tB, o, 64)); }
      tA = gcn_readfirstlane(tA);
      tB = gcn_readfirstlane(tB);
#define AD_FLUSH(TT) { if (!FAST) { pg[(size_t)((TT) >> 4) * 64 + lane] = pw; pw = 0; } }   /* every lane, unconditionally: one 256-byte store */
#define AD_STEP(PARV, LEANV, VNEXT, FS, KOK, GS)                                                                                   \
  {                                                                                                                             \
    if (FAST) ad_step_k<GL, PARV, LEANV, EDGE, true>(d0, d1, i, j, cb, rb, (PARV) ? acc1 : acc0, (VNEXT), (FS), g_first, g_last, (KOK), (GS), (PARV) ? gn1 : gn0, L1, L2, (PARV) ? &ep1 : &ep0); \
    else if (DEF) ad_step_k<GL, PARV, LEANV, EDGE>(d0, d1, i, j, cb, rb, pw, (VNEXT), (FS), g_first, g_last, (KOK), (GS), (PARV) ? gn1 : gn0, L1, L2); \
    else ad_step<GL, PARV, DEF, LEANV, EDGE, HOMO>(d0, d1, i, j, cb, rb, pw, (VNEXT), (FS), g_first, g_last, (KOK), (GS), L1, L2, SENT, MATCH, MISMATCH, GAP, HGAP); \
  }
#define AD_FULL_STEP(TT)                                                                                                        \
  {                                                                                                                             \
    if (((TT) & 1) == 0) AD_STEP(0, false, rwd[j], ((TT) & 15) << 1, kok0, gs0)                                                 \
    else AD_STEP(1, false, cwd[i], ((TT) & 15) << 1, kok1, gs1)                                                                 \
    if (((TT) & 15) == 15) AD_FLUSH(TT)                                                                                         \
  }
#define AD_LEAN_PAIR(TT)                                                                                                        \
  {                                                                                                                             \
    AD_STEP(0, true, rwd[j], ((TT) & 15) << 1, kok0, gs0)                                                                       \
    AD_STEP(1, true, cwd[i], (((TT) + 1) & 15) << 1, kok1, gs1)                                                                 \
    if ((((TT) + 1) & 15) == 15) AD_FLUSH((TT) + 1)                                                                             \
  }
      int t = 0;
      // leading steps with the matrix-edge logic up to the steady state, then steady-state pairs up to a 16-step boundary
      const int tA2 = (tA + 1) & ~1;
      for (; t <= Tmax && t < tA2; t++) AD_FULL_STEP(t)
      for (; (t & 15) != 0 && t + 2 <= tB && t + 1 <= Tmax; t += 2) AD_LEAN_PAIR(t)
      // steady state in blocks of 16 steps = one pointer word per column: constant field shifts, one flush per block
      for (; t + 16 <= tB && t + 15 <= Tmax; t += 16) {
        // the block consumes eight raw and eight centre bases: two 32-byte LDS reads each
        uint32_t rwv[8], cwv[8];
        __builtin_memcpy(rwv, rwd + j, 32);
        __builtin_memcpy(cwv, cwd + i, 32);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          AD_STEP(0, true, rwv[u], 4 * u, kok0, gs0)
          AD_STEP(1, true, cwv[u], 4 * u + 2, kok1, gs1)
        }
        AD_FLUSH(t)
      }
      for (; t + 2 <= tB && t + 1 <= Tmax; t += 2) AD_LEAN_PAIR(t)
      for (; t <= Tmax; t++) AD_FULL_STEP(t)
#undef AD_LEAN_PAIR
#undef AD_FULL_STEP
#undef AD_STEP
#undef AD_FLUSH
      if (!FAST && ((t - 1) & 15) != 15) pg[(size_t)((t - 1) >> 4) * 64 + lane] = pw >> (2 * (15 - ((t - 1) & 15)));   // (the last, partial block: step s at bits 2s too)
    }
    // ---- FAST: the walk back without pointers.  Every lane of a group follows it redundantly (what it reads comes from the lane
    //      that owns the band cell in question, so all of them see the same): from (L1, L2) - the last cell of the diagonal
    //      j - i = L2 - L1, band cell k' = L2 - L1 + lbo - along the last column (free up moves: the next diagonal's last cell) or
    //      the last row (free left moves: the previous one's) until a last cell says "diagonal"; that diagonal must then have been
    //      walked whole.  fres: 1 = the alignment is (dir, fm free moves, diagonal fk), 2 = not of this form (retry list).
    int fres = 0, fk = 0, fm = 0, fdir = -1;
    if (FAST) {
      fk = L2 - L1 + lbo;
      if (T < 0) fres = 2;                                  // (idle slots: nothing to decide)
      for (int itw = 0; itw <= 2 * GL + 1; itw++) {         // (a walk visits each band cell at most once)
        if (__all(fres != 0)) break;
        const bool inband = fk >= org && fk < W + org;
        const int src = al * GL + (inband ? (fk >> 1) : 0);
        const uint32_t e0 = (uint32_t)__shfl((int)ep0, src, 64), e1 = (uint32_t)__shfl((int)ep1, src, 64);
        const uint32_t a0 = (uint32_t)__shfl((int)acc0, src, 64), a1 = (uint32_t)__shfl((int)acc1, src, 64);
        if (fres != 0) continue;
        const uint32_t pe = (fk & 1) ? e1 : e0, whole = (((fk & 1) ? a1 : a0) >> 1) & 1u;
        if (!inband || pe > AD_DIAG) fres = 2;              // (pe == 3: a cell the sweep never reached - cannot happen on a walk, but never trusted)
        else if (pe == AD_DIAG) fres = whole ? 1 : 2;
        else if (fdir < 0 || (int)pe == fdir) { fdir = (int)pe; fm++; fk += pe == AD_UP ? 1 : -1; }
        else fres = 2;                                      // the free run turned: an interior gap
      }
      if (fres == 0) fres = 2;
    }
    // ---- traceback (first lane of each group) in chunks of <= AD_RCAP merged runs, expanded by all lanes into one
    //      transition code per raw position.  Run: pj_lo (12 b) | n (12 b) << 12 | (delta + 128) << 24, delta = pi - pj;
    //      255 << 24 = gap in the centre (self transition).
    int ti = L1, tj = L2;
    bool done = !active || (dbg & 2);
    uint32_t h = 0;
    int guard = L1 + L2 + 2;                               // bounded: never spin on bad pointers
    while (true) {
      int nruns = 0;
      uint32_t last = 0;                                   // the leader's pending (mergeable) run, 0 = none
      auto push = [&](int lo, int n, int dl) {
        if (last) {
          const int llo = last & 4095, ln = (last >> 12) & 4095, ldl = (int)(last >> 24);
          if (ldl == dl && lo + n == llo) { last = (uint32_t)lo | ((uint32_t)(ln + n) << 12) | ((uint32_t)dl << 24); return; }
          runs[nruns++] = last;
        }
        last = (uint32_t)lo | ((uint32_t)n << 12) | ((uint32_t)dl << 24);
      };
      const bool lead = g == 0 && !ghost;
      if (lead && !done && gapless) {
        // nwalign_gapless (nwalign_endsfree.cpp:539-555): position-wise pairing, a longer raw's tail faces gaps
        const int n = L1 < L2 ? L1 : L2;
        runs[nruns++] = 0u | ((uint32_t)n << 12) | (128u << 24);
        if (L2 > n) runs[nruns++] = (uint32_t)n | ((uint32_t)(L2 - n) << 12) | (255u << 24);
        done = true;
      }
      if (FAST && lead && !done) {
        // the alignment fast_walk found, as the run descriptors the pointer walk would have pushed: fm raw positions facing the
        // free end gap (left moves along the last row; up moves consume centre positions only), the diagonal j - i = dd from the
        // border to its last cell, the raw positions in front of it on the first row
        if (fres == 1) {
          const int dd = fk - lbo;
          const int n = dd >= 0 ? min(L1, L2 - dd) : min(L1 + dd, L2);
          if (fdir == (int)AD_LEFT && fm > 0) runs[nruns++] = (uint32_t)(L2 - fm) | ((uint32_t)fm << 12) | (255u << 24);
          if (n > 0) runs[nruns++] = (uint32_t)(dd >= 0 ? dd : 0) | ((uint32_t)n << 12) | ((uint32_t)(128 - dd) << 24);
          if (dd > 0) runs[nruns++] = 0u | ((uint32_t)dd << 12) | (255u << 24);
        } else if (active && !gapless) {
          const int q = atomicAdd(&a.retry_n[kcur], 1);
          a.retry_list[(size_t)kcur * a.batch_stride + q] = r;
          if (a.fast_ctl) atomicAdd(&a.fast_ctl[0], 1ull);
        }
        done = true; ti = 0; tj = 0;
      }
      // The path is walked by the group's first lane, but every diagonal stretch is measured by the whole group at once:
      // lane q looks at the pointer word q blocks of 16 steps further back in the path's column, so one round finds the
      // next non-diagonal move up to 16 GL steps away (a gap-free 250-nt alignment takes 2 rounds instead of 32).
      const int gl0 = al * GL;
      if (!FAST)
      for (;;) {
        const bool act = lead && !done && (ti > 0 || tj > 0) && nruns < AD_RCAP - 2 && guard > 0;
        if (!__any(act)) break;
        bool gapmove = false;
        const int gact = __shfl((int)act, gl0, 64);
        const int tt = __shfl(ti + tj, gl0, 64), col = __shfl((tj - ti + lbo) >> 1, gl0, 64);
        const int f0 = tt & 15, widx = (tt >> 4) - g;
        uint32_t word = 0xAAAAAAAAu;                       // (before the matrix: never reached, the axis cells stop the run)
        if (gact && !ghost && widx >= 0) word = pg[(size_t)widx * 64 + gl0 + col];
        const int ftop = g == 0 ? f0 : 14 + (f0 & 1);
        // fields of the path cell's parity at positions <= ftop that are NOT diagonal (11)
        const uint32_t x = word ^ 0xAAAAAAAAu;              // step s of the block at bits 2s, 2s + 1; a zero field = diagonal
        uint32_t nz = (x | (x >> 1)) & 0x55555555u;
        nz &= (f0 & 1) ? 0x44444444u : 0x11111111u;
        nz &= (ftop == 15) ? 0xFFFFFFFFu : ((1u << ((ftop + 1) << 1)) - 1u);
        const bool st = nz != 0;
        const int fb = st ? (31 - __clz(nz)) >> 1 : 0;
        const int ng = st ? (ftop - fb) >> 1 : (ftop >> 1) + 1;   // diagonal moves inside this word
        const uint32_t pst = (word >> (fb << 1)) & 3u;            // the pointer that ends the stretch
        const unsigned long long bal = (__ballot(st && !ghost) >> gl0) & (GL == 64 ? ~0ull : ((1ull << (GL & 63)) - 1ull));
        const int qs = bal ? __builtin_ctzll(bal) : GL;           // first word (counting back) holding a stop
        const int srcl = gl0 + (qs < GL ? qs : 0);
        const int n0 = __shfl(ng, gl0, 64), nq = __shfl(ng, srcl, 64);
        const uint32_t pq = (uint32_t)__shfl((int)pst, srcl, 64);
        if (act) {
          guard--;
          int n = qs == 0 ? n0 : n0 + 8 * (qs - 1) + (qs < GL ? nq : 0);
          const int room = ti < tj ? ti : tj;
          const bool clamped = n > room;                   // (cannot happen: cells on the first row/column carry pointers 2/3)
          if (clamped) { n = room; atomicOr(S.nw_flag, 1); }
          if (n > 0) {
            push(tj - n, n, ti - tj + 128);
            ti -= n; tj -= n;
          }
          gapmove = qs < GL && !clamped && (ti > 0 || tj > 0);   // (0,0) carries an axis pointer too: the path ends there
        }
        // The stretch ended at a gap move.  A RUN of gap moves in one direction - the free end gaps of a pair of unequal
        // lengths are 20-60 of them at 1.5 kb - is measured by the whole group in one round trip: lane s looks at the pointer
        // of the cell s moves further along the row (from the left) or the column (from above).  Taking them one per round
        // was three quarters of the long reads' traceback time (profiles/r05v_nw_phases_cfg5.jsonl).
        const int gdir = __shfl(gapmove ? (int)pq : -1, gl0, 64);
        const int gi = __shfl(ti, gl0, 64), gj = __shfl(tj, gl0, 64);
        bool cont = false;
        if (gdir >= 0 && !ghost) {
          const int ci = gdir == (int)AD_LEFT ? gi : gi - g, cj = gdir == (int)AD_LEFT ? gj - g : gj;
          if (gdir == (int)AD_LEFT ? (cj >= 1 && ci >= 0) : (ci >= 1 && cj >= 0)) {
            const int t2 = ci + cj, k2 = cj - ci + lbo;
            if (k2 >= org && k2 < W + org)                 // (in band: the path never leaves it)
              cont = ((pg[(size_t)(t2 >> 4) * 64 + gl0 + (k2 >> 1)] >> ((t2 & 15) << 1)) & 3u) == (uint32_t)gdir;
          }
        }
        const unsigned long long runmask = (__ballot(cont) >> gl0) & (GL == 64 ? ~0ull : ((1ull << (GL & 63)) - 1ull));
        if (gapmove) {
          int m = runmask == ~0ull ? 64 : __builtin_ctzll(~runmask);   // leading lanes whose cell goes on in the same direction
          if (m > GL) m = GL;
          if (m < 1) m = 1;                                // (lane 0 looks at the cell the stretch ended on: never 0)
          if (pq == AD_LEFT) { tj -= m; push(tj, m, 255); }   // from the left: m raw positions facing gaps
          else {                                           // from above (the diagonal code cannot be here)
            ti -= m;
            if (LR) push(ti, m, 254);                      // bimera mode: centre positions facing gaps are columns too
          }
        }
      }
      if (lead && !gapless && active) {
        if (last) runs[nruns++] = last;
        if (!(ti > 0 || tj > 0)) done = true;
        else if (guard <= 0) { done = true; atomicOr(S.nw_flag, 1); }   // bounded walk tripped: report it
      }
      nruns = __shfl(nruns, al * GL, 64);
      done = __shfl((int)done, al * GL, 64) != 0;
      int nrmax = nruns;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) nrmax = max(nrmax, __shfl_xor(nrmax, o, 64));
      for (int ri = 0; ri < nrmax; ri++) {
        if (ri < nruns && !ghost && !(dbg & 4)) {
          const uint32_t dsc = runs[ri];
          const int lo = dsc & 4095, n = (dsc >> 12) & 4095, dl = (int)(dsc >> 24);
          if (LR) {
            // bimera mode: one bit per column instead of a factor per raw position.  254 / 255 = gap in the parent / the query
            for (int p = lo + g; p < lo + n; p += GL) {
              const int tc = ecol + (lo + n - 1 - p);
              const uint32_t bit = 1u << (tc & 31);
              bool ne = true;
              if (dl < 254) ne = __builtin_ctz(rwd[p]) != __builtin_ctz(cwd[p + dl - 128]);
              else atomicOr(&bm[(dl == 255 ? 1 : 2) * G.bmw + (tc >> 5)], bit);
              if (ne) atomicOr(&bm[tc >> 5], bit);
            }
            ecol += n;
          } else
          for (int pj = lo + g; pj < lo + n; pj += GL) {
            const uint32_t rb = (uint32_t)__builtin_ctz(rwd[pj]) >> 3;   // base code back from its word 9 << (8 * code)
            const uint32_t q = a.ap.use_quals ? qrow[pj] : 0u;
            uint32_t tc = 5u * rb;
            if (dl != 255) {
              const uint32_t cb = (uint32_t)__builtin_ctz(cwd[pj + dl - 128]) >> 3;
              tc = 4u * cb + rb;
              h += (cb != rb);
              if (a.view && active)
                a.view[vr * a.LV + pj + dl - 128] = (uint16_t)(0x8000u | (rb << 8) | q);
            }
            rwd[pj] = (tc * (uint32_t)a.ap.ncol + q) << 3;   // &err[t(pj)][q(pj)] - err, in bytes: replaces the base (each position is in ONE run)
          }
        }
      }
      if (__all(done)) break;
    }
    if (LR) {
      gcn_wave_sync();                                     // every lane's bits are in LDS before one lane reads them
      if (g == 0 && !ghost && active)
        bimera_lr_bits(bm, bm + G.bmw, bm + 2 * G.bmw, ecol, a.lr_one_off, a.lr_max_shift, a.lr_out + (size_t)idx * 5);
      continue;
    }
    // hamming: group sum through LDS (group sizes are not powers of two); the run buffer is free by now
    if (g == 0 && !ghost) runs[0] = 0;
    gcn_wave_sync();
    if (!ghost && h) atomicAdd(&runs[0], h);
    gcn_wave_sync();                                       // (also: every lane's factors are in LDS before one lane multiplies them)
    h = runs[0];
    // ---- lambda: sequential product in raw-position order (pval.cpp:188-192), one lane per alignment.  The factors are
    //      fetched straight from the LDS copy of err through the offsets the expansion left: eight offsets per two 16-byte reads,
    //      the next eight factors on their way while the current eight are multiplied (the product itself stays strictly
    //      sequential) ---------
    // Since round 4 the product normally runs in k_ad_product, 64 alignments to a wave: here it kept ONE lane of a wave busy
    // for 2.5 instructions per position - a sixth of the launch's issued wave-instructions on an issue-bound kernel.  The
    // group copies its offsets out (u16, coalesced) and leaves a descriptor; alignments past the buffer's capacity are still
    // multiplied up here.
    const long long fid = ((long long)it * 4 + wib) * APW + al;
    const bool offload = S.ad_foff != nullptr && fid < (long long)S.ad_fcap && !(dbg & 8);
    if (FAST && fres != 1) continue;                       // (on the retry list: the full kernel will write this pair's results)
    if (offload) {
      if (active && !ghost) {
        uint16_t *fo = S.ad_foff + (size_t)fid * S.ad_fstride;
        for (int pj = g; pj < L2; pj += GL) fo[pj] = (uint16_t)rwd[pj];
        if (g == 0) {
          AdDesc d;
          d.dest = (long long)(out_off + r); d.L2 = L2; d.pad = 0;
          S.ad_desc[fid] = d;
          a.ham[out_off + r] = h;
        }
      }
    } else
    if (g == 0 && active && !(dbg & 8)) {
      const char *eb = (const char *)s_err;
      auto fetch8 = [&](int pj, double (&f)[8]) __attribute__((always_inline)) {
        uint32_t o[8];
        __builtin_memcpy(o, (const uint32_t *)__builtin_assume_aligned(rwd + pj, 16), 32);   // (pj is a multiple of 8, the guard a multiple of 4 words)
#pragma unroll
        for (int k = 0; k < 8; k++) f[k] = *(const double *)(eb + o[k]);
      };
      double l = 1.0;
      int pj = 0;
      if (L2 >= 8) {
        double f[8], n[8];
        fetch8(0, f);
        for (pj = 8; pj + 8 <= L2; pj += 8) {
          fetch8(pj, n);
#pragma unroll
          for (int k = 0; k < 8; k++) l = l * f[k];
#pragma unroll
          for (int k = 0; k < 8; k++) f[k] = n[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) l = l * f[k];
      }
      for (; pj < L2; pj++) l = l * *(const double *)(eb + rwd[pj]);
      a.lam[out_off + r] = l;
      a.ham[out_off + r] = h;
    }
  }
}

// lambda = the sequential product of the per-position error-model factors in raw-position order (pval.cpp:188-192) for the
// alignments k_nw_ad left offsets and a descriptor for: ONE LANE per alignment (the product of one alignment cannot be split -
// fp64 multiplication is not associative and the reference's order is part of the result - but 64 alignments multiply side by
// side), eight offsets per 16-byte load, the factors from the LDS copy of err.  A descriptor is consumed (dest = -1) so that
// the slots a later, smaller launch does not write are not multiplied again.
__global__ __launch_bounds__(256) void k_ad_product(const uint16_t *__restrict__ foff, int stride, AdDesc *__restrict__ desc, int nscan,
                                                    const double *__restrict__ err, int nerr, double *__restrict__ lam,
                                                    const int32_t *__restrict__ stop_dev, const int32_t *__restrict__ batch_on,
                                                    const int32_t *__restrict__ batch_n, int slots_per_slice,
                                                    const unsigned long long *__restrict__ fast_ctl = nullptr, const int32_t *__restrict__ alt_n = nullptr) {
  extern __shared__ double s_err[];
  if (stop_dev && *stop_dev != 0) return;
  if (fast_ctl && alt_n && fast_ctl[2] != 0ull) batch_n = alt_n;   // (the pointer-free pass is off: the launch in front worked on the batch's own lists)
  if (batch_on) {   // batch mode: the launch's work slots are known on the device only (the lists' lengths, as k_nw_ad counts them)
    const int nb = *batch_on;
    if (nb <= 0) return;
    int slices = 0;
    for (int k = 0; k < KB_MAX; k++) slices += k < nb ? (batch_n[k] + slots_per_slice - 1) / slots_per_slice : 0;
    nscan = min(nscan, slices * slots_per_slice);
  }
  if ((int)(blockIdx.x * blockDim.x) >= nscan) return;
  for (int i = threadIdx.x; i < nerr; i += blockDim.x) s_err[i] = err[i];
  __syncthreads();
  const char *eb = (const char *)s_err;
  for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < nscan; id += gridDim.x * blockDim.x) {
    const AdDesc d = desc[id];
    if (d.dest < 0) continue;
    desc[id].dest = -1;
    const uint16_t *row = foff + (size_t)id * stride;
    double l = 1.0;
    int pj = 0;
    for (; pj + 8 <= d.L2; pj += 8) {
      const uint4 o = *(const uint4 *)(row + pj);
      const uint32_t w[4] = {o.x, o.y, o.z, o.w};
      double f[8];
#pragma unroll
      for (int k = 0; k < 4; k++) { f[2 * k] = *(const double *)(eb + (w[k] & 0xFFFFu)); f[2 * k + 1] = *(const double *)(eb + (w[k] >> 16)); }
#pragma unroll
      for (int k = 0; k < 8; k++) l = l * f[k];
    }
    for (; pj < d.L2; pj++) l = l * *(const double *)(eb + row[pj]);
    lam[d.dest] = l;
  }
}

void launch_nw_ad(const SampleDev &S, int centre, const int32_t *d_chunk_centre, const int32_t *d_work,
                  const int32_t *d_nwork, int nwork_host, const int32_t *d_gl_work, const int32_t *d_gl_nwork,
                  const AlignParams &ap, const double *d_err, double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV,
                  int view_by_chunk, const int32_t *d_centre_dev, hipStream_t st, const int32_t *d_stop_dev, const NwBatch *batch) {
  int maxwork = d_nwork ? S.N : nwork_host;
  if (batch) maxwork = S.N;
  if (maxwork <= 0 && !d_gl_work) return;
  NwArgs a;
  memset(&a, 0, sizeof a);
  a.S = S; a.centre = centre; a.chunk_centre = d_chunk_centre; a.work = d_work; a.nwork_dev = d_nwork;
  a.nwork_host = nwork_host; a.ap = ap; a.err = d_err; a.lam = d_lambda; a.ham = d_ham;
  a.view = d_view; a.LV = LV; a.view_by_chunk = view_by_chunk; a.centre_dev = d_centre_dev; a.stop_dev = d_stop_dev;
#ifdef DADA2HIP_PROFILING   // `make prof` only (libdada2hip_prof.so, tools/nw_phases.py): skipping phases voids the results
  a.moves_stride = knobs().ad_debug;
#endif
  if (batch) {
    a.batch_on = batch->on; a.batch_n = batch->n; a.batch_list = batch->list; a.batch_centre = batch->centre; a.batch_bbuf = batch->bbuf;
    a.batch_stride = batch->stride;
  }
  const AdGeom G = ad_geom(ap.band, S.maxlen, S.minlen, d_chunk_centre ? 0 : 1);
  const size_t lds = (size_t)16 * ap.ncol * 8 + (size_t)G.block_bytes;
  int waves = (std::max(maxwork, 1) + G.APW - 1) / G.APW;
  if (d_gl_work || batch) waves = (S.N + G.APW - 1) / G.APW;
  int grid = std::max(1, std::min((waves + 3) / 4, std::min(256 * 8, S.ad_waves / 4)));   // one slot of the pointer ring per wave
  const bool homo = ap.endsfree && ap.homo_gap != ap.gap;   // nwalign_endsfree_homo: the generic-score step with per-base gap costs
  const bool def = !homo && ap.match == 5 && ap.mismatch == -4 && ap.gap == -8 && ap.sentinel == -32760;
  // a batch compare on the default scores: first the pointer-free pass over the batch's lists (k_nw_ad<.., FAST>), then the full
  // kernel over the pairs that pass could not finish (its retry lists: usually a few per cent of the pairs, often none)
  const bool fast = batch && def && batch->retry_list && batch->retry_n && knobs().ad_fast != 0 && a.moves_stride == 0;
  // per (instance, device): the dynamic-LDS attribute belongs to the function ON a device
  auto set_lds = [&](const void *fn, std::atomic<size_t> (&done)[64]) {
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    std::atomic<size_t> &d = done[dev_ & 63];
    if (lds > d.load(std::memory_order_acquire)) {
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      d.store(lds, std::memory_order_release);
    }
  };
#define D2_LAUNCH_AD(GLV, DEFV, EDGEV, HOMOV, FASTV)                                                                     \
  do {                                                                                                                   \
    static std::atomic<size_t> attr_set[64];                                                                             \
    set_lds((const void *)k_nw_ad<GLV, DEFV, EDGEV, false, HOMOV, FASTV>, attr_set);                                     \
    hipLaunchKernelGGL((k_nw_ad<GLV, DEFV, EDGEV, false, HOMOV, FASTV>), dim3(grid), dim3(256), lds, st, a, d_gl_work, d_gl_nwork, G);  \
  } while (0)
#define D2_LAUNCH_AD2(GLV, FASTV)                                                                                        \
  do {                                                                                                                   \
    if (FASTV) { if (G.edge) D2_LAUNCH_AD(GLV, true, true, false, FASTV); else D2_LAUNCH_AD(GLV, true, false, false, FASTV); } \
    else if (homo) { if (G.edge) D2_LAUNCH_AD(GLV, false, true, true, false); else D2_LAUNCH_AD(GLV, false, false, true, false); } \
    else if (def) { if (G.edge) D2_LAUNCH_AD(GLV, true, true, false, false); else D2_LAUNCH_AD(GLV, true, false, false, false); } \
    else { if (G.edge) D2_LAUNCH_AD(GLV, false, true, false, false); else D2_LAUNCH_AD(GLV, false, false, false, false); }   \
  } while (0)
  auto product = [&](const int32_t *list_n, const unsigned long long *fctl = nullptr, const int32_t *alt_n = nullptr) {   // the products of what a launch aligned (its work slots: ids below the bound)
    if (!(S.ad_foff && a.moves_stride == 0)) return;
    const int nerr = 16 * ap.ncol;
    long long bound = batch ? (long long)S.ad_fcap : (long long)((maxwork + (d_gl_work ? S.N : 0) + 4 * G.APW - 1) / (4 * G.APW) + 1) * 4 * G.APW;
    const int nscan = (int)std::min<long long>(bound, S.ad_fcap);
    const int pgrid = std::max(1, std::min((nscan + 255) / 256, 2048));
    hipLaunchKernelGGL(k_ad_product, dim3(pgrid), dim3(256), (size_t)nerr * 8, st, (const uint16_t *)S.ad_foff, (int)S.ad_fstride, S.ad_desc, nscan, d_err, nerr,
                       d_lambda, d_stop_dev, batch ? batch->on : nullptr, list_n, 4 * G.APW, fctl, alt_n);
  };
  if (fast) {
    a.retry_list = batch->retry_list; a.retry_n = batch->retry_n; a.fast_ctl = batch->fast_ctl;
    if (G.GL == 21) D2_LAUNCH_AD2(21, true);
    else if (G.GL == 32) D2_LAUNCH_AD2(32, true);
    else D2_LAUNCH_AD2(64, true);
    product(batch->n);
    a.alt_list = a.batch_list; a.alt_n = a.batch_n;                 // (what the full kernel works through while the pass is off)
    a.batch_list = batch->retry_list; a.batch_n = batch->retry_n;   // (rows k < KB_MAX are all either kernel reads)
    a.retry_list = nullptr; a.retry_n = nullptr;
  }
  if (G.GL == 21) D2_LAUNCH_AD2(21, false);
  else if (G.GL == 32) D2_LAUNCH_AD2(32, false);
  else D2_LAUNCH_AD2(64, false);
#undef D2_LAUNCH_AD2
#undef D2_LAUNCH_AD
  if (fast) product(batch->retry_n, batch->fast_ctl, batch->n);
  else product(batch ? batch->n : nullptr);
}

// Bimera mode: the pairs (query = chunk centre, parent = work item; chunks of nw_ad_apw() slots, -1 = empty slot) aligned by
// k_nw_ad<.., LR> with band = max_shift (chimera.cpp:26,122), each reduced in the kernel to get_lr / get_ham_endsfree:
// d_out[5 slot .. 5 slot + 4] = left, right, left_oo, right_oo, hamming.  Needs the pointer ring (SampleDev::ad_ptr).
void launch_nw_ad_lr(const SampleDev &S, const int32_t *d_chunk_centre, const int32_t *d_work, int nwork, const AlignParams &ap,
                     const double *d_err, int allow_one_off, int max_shift, int32_t *d_out, hipStream_t st) {
  if (nwork <= 0) return;
  NwArgs a;
  memset(&a, 0, sizeof a);
  a.S = S; a.chunk_centre = d_chunk_centre; a.work = d_work; a.nwork_host = nwork; a.ap = ap; a.err = d_err;
  a.lr_out = d_out; a.lr_one_off = allow_one_off; a.lr_max_shift = max_shift;
  const AdGeom G = ad_geom(ap.band, S.maxlen, S.minlen, 0, 1);
  const size_t lds = (size_t)16 * ap.ncol * 8 + (size_t)G.block_bytes;
  const int waves = (nwork + G.APW - 1) / G.APW;
  const int grid = std::max(1, std::min((waves + 3) / 4, std::min(256 * 8, S.ad_waves / 4)));   // one slot of the pointer ring per wave
  const bool def = ap.match == 5 && ap.mismatch == -4 && ap.gap == -8 && ap.sentinel == -32760;
#define D2_LAUNCH_LR(GLV, DEFV, EDGEV)                                                                                   \
  do {                                                                                                                   \
    static size_t attr_set[64] = {0};                                                                                    \
    int dev_ = 0;                                                                                                        \
    (void)hipGetDevice(&dev_);                                                                                           \
    if (lds > attr_set[dev_ & 63]) {                                                                                     \
      (void)hipFuncSetAttribute((const void *)k_nw_ad<GLV, DEFV, EDGEV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_set[dev_ & 63] = lds;                                                                                         \
    }                                                                                                                    \
    hipLaunchKernelGGL((k_nw_ad<GLV, DEFV, EDGEV, true>), dim3(grid), dim3(256), lds, st, a, (const int32_t *)nullptr,   \
                       (const int32_t *)nullptr, G);                                                                     \
  } while (0)
#define D2_LAUNCH_LR2(GLV)                                                                                               \
  do {                                                                                                                   \
    if (def) { if (G.edge) D2_LAUNCH_LR(GLV, true, true); else D2_LAUNCH_LR(GLV, true, false); }                         \
    else { if (G.edge) D2_LAUNCH_LR(GLV, false, true); else D2_LAUNCH_LR(GLV, false, false); }                           \
  } while (0)
  if (G.GL == 21) D2_LAUNCH_LR2(21);
  else if (G.GL == 32) D2_LAUNCH_LR2(32);
  else D2_LAUNCH_LR2(64);
#undef D2_LAUNCH_LR2
#undef D2_LAUNCH_LR
}
size_t nw_ad_lr_lds_bytes(const SampleDev &S, const AlignParams &ap) {
  if (nw_ad_lds_bytes(S, ap) == 0) return 0;
  return (size_t)16 * ap.ncol * 8 + (size_t)ad_geom(ap.band, S.maxlen, S.minlen, 0, 1).block_bytes;
}

// LDS needed by k_nw_ad for this sample/band, or 0 when the cooperative kernel does not apply.
int nw_ad_apw(const SampleDev &S, const AlignParams &ap) { return ad_geom(ap.band, S.maxlen, S.minlen).APW; }
size_t nw_ad_lds_bytes(const SampleDev &S, const AlignParams &ap) {
  // (factor offsets are u16: 16 * ncol * 8 < 65 536.)  The global aligner (endsfree = 0: C_nwalign only) stays on the lane
  // kernels; the homopolymer-gap aligner of dada() runs here unless DADA2HIP_AD_HOMO=0 sends it back to them.
  const bool homo_ok = knobs().ad_homo;
  if (ap.band <= 0 || S.maxlen > 2047 || ap.ncol > 500 || S.ad_waves <= 0 || !ap.endsfree || (!ap.plain() && !homo_ok)) return 0;
  const int W = 2 * ap.band + (S.maxlen - S.minlen) + 1;
  if (W > 127) return 0;
  const AdGeom G = ad_geom(ap.band, S.maxlen, S.minlen);   // (a centre per wave: the larger of the two layouts)
  return (size_t)16 * ap.ncol * 8 + (size_t)G.block_bytes;
}

// ------------------------------------------------------------------------------------------------
// Wide-band / long-read variant of the anti-diagonal kernel (BASELINE.json configs[4]: ~1 500 nt, band 32,
// ragged lengths -> band windows of 100-500 cells).  Same sweep as k_nw_ad, but every lane owns EIGHT
// consecutive band cells (four live per step, all in registers), so 21 / 32 / 64 lanes cover windows of 168 /
// 256 / 512 cells, and the 2-bit pointers go to an HBM scratch ring ([16-step block][cell pair][lane]: each flush
// is four coalesced 256-B stores per wave) instead of LDS, which leaves LDS for 8 waves per CU.  The band origin
// of each alignment is shifted by one cell when its left band is odd, so every alignment of a wave is in phase
// (even cells on even steps) whatever the pair's length difference.
// Cell k' = 8g + c of lane g maps to the pair's band cell k = k' - s (s = lband & 1); lbs = lband + s.
// On an even step t the live cells c = 2m have i = I - m, j = J + m with I = (t - 8g + lbs) / 2, J = t - I; on the
// following odd step the cells c = 2m + 1 have i = I - m, j = J + 1 + m.  The four centre bases and four raw bases
// a step needs are two byte-packed shift registers fed by one LDS byte load per step.
struct AdwGeom {
  int GL, APW, pad, seqbytes, tbytes, fch, per_al_bytes, nblk16;
};
constexpr uint32_t ADW_DL0 = 8192, ADW_GAP = 0xFFFFFFFFu;   // run descriptor word y: (pi - pj) + ADW_DL0, or "gap in the centre"
static __host__ __device__ inline AdwGeom adw_geom(int band, int maxlen, int minlen) {
  AdwGeom G;
  const int We = 2 * band + (maxlen - minlen) + 2;        // widest window + the phase cell
  G.GL = We <= 168 ? 21 : (We <= 256 ? 32 : 64);
  G.APW = 64 / G.GL;
  G.pad = 8 * G.GL + 8;                                   // guard bytes either side of a staged sequence
  G.seqbytes = (maxlen + 2 * G.pad + 7) & ~7;
  G.tbytes = (maxlen + 7) & ~7;
  G.fch = (G.seqbytes / 8) & ~7;                          // fp64 factors per chunk (they alias the centre bytes)
  G.per_al_bytes = AD_RCAP * 8 + 2 * G.seqbytes + G.tbytes;
  G.nblk16 = (2 * maxlen + 1 + 15) / 16 + 1;
  return G;
}

// one step of a lane's four live cells.  PAR = parity of the live cells; FS = 2 * (t & 15) (field shift).
template <int PAR, bool LEAN, bool DEF>
static __device__ __forceinline__ void adw_step(int (&d)[8], uint32_t (&pw)[4], int fs, uint32_t x, int nb, uint32_t kmask,
                                                const int (&gs)[8], int I, int Jr, int L1, int L2, int SENT_, int MATCH_,
                                                int MISMATCH_, int GAP_) {
  const int SENT = DEF ? -32760 : SENT_, MATCH = DEF ? 5 : MATCH_, MISMATCH = DEF ? -4 : MISMATCH_, GAP = DEF ? -8 : GAP_;
  if (LEAN) {
    // steady state as in k_nw_ad: out-of-band cells are pushed AD_OOB below the band by their own gap addend gs[c]
    // (GAP in band) instead of being reset to the sentinel, and the three-way maximum is one v_max3
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int c = 2 * m + PAR;
      const int left_src = PAR ? d[c - 1] : (m == 0 ? nb : d[c - 1]);
      const int up_src = PAR ? (m == 3 ? nb : d[c + 1]) : d[c + 1];
      const int diag = d[c] + (((x >> (8 * m)) & 0xFFu) == 0 ? MATCH : MISMATCH);
      const int left = left_src + gs[c], up = up_src + gs[c];
      const int e = gcn_max3(left, diag, up);
      const bool t1 = left >= diag, t2 = up == e;            // up >= max(left, diag)  <=>  the maximum IS up
      d[c] = e;
      pw[m] |= (t2 ? 3u : (t1 ? 2u : 1u)) << fs;
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int c = 2 * m + PAR;
    const int left_src = PAR ? d[c - 1] : (m == 0 ? nb : d[c - 1]);
    const int up_src = PAR ? (m == 3 ? nb : d[c + 1]) : d[c + 1];
    const bool kin = (kmask >> c) & 1u;
    const int diag = d[c] + (((x >> (8 * m)) & 0xFFu) == 0 ? MATCH : MISMATCH);
    const int i = I - m, j = Jr + m;
    const int up = up_src + ((!LEAN && j == L2) ? 0 : GAP);      // free moves along the last column
    const int left = left_src + ((!LEAN && i == L1) ? 0 : GAP);  // ... and the last row
    const bool t1 = left >= diag;
    const int e1 = max(left, diag);
    const bool t2 = up >= e1;
    const int e = max(up, e1);
    uint32_t p = t2 ? 3u : (t1 ? 2u : 1u);
    int val;
    if (LEAN) {
      val = kin ? e : SENT;
    } else {
      const bool interior = kin && ((unsigned)(i - 1) < (unsigned)L1) && ((unsigned)(j - 1) < (unsigned)L2);
      val = interior ? e : (kin ? 0 : SENT);
      if (!interior) p = (i <= 0 ? 2u : 3u);                     // first row: left, first column: up
    }
    d[c] = val;
    pw[m] |= p << fs;
  }
}

template <int GL, bool DEF>
__global__ __launch_bounds__(256, 3) void k_nw_adw(NwArgs a, AdwGeom G) {
  constexpr int APW = 64 / GL;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double *s_err = s_dyn;
  const int nerr = 16 * a.ap.ncol;
  if ((int)blockIdx.x * 4 * APW >= (a.nwork_dev ? *a.nwork_dev : a.nwork_host)) return;   // grid sized for the worst case
  for (int i = threadIdx.x; i < nerr; i += blockDim.x) s_err[i] = a.err[i];
  const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool ghost = lane / GL >= APW;                     // GL = 21: lane 63 rides along with no cells of its own
  const int al = ghost ? APW - 1 : lane / GL, g = ghost ? GL : lane % GL;
  uint8_t *abase = (uint8_t *)(s_dyn + nerr) + ((size_t)wib * APW + al) * G.per_al_bytes;
  // run descriptors are two words here: x = pj_lo | n << 12, y = (pi - pj) + ADW_DL0, ADW_GAP = gap in the centre.  Band
  // windows reach 512 cells, so |pi - pj| reaches ~500 (a 1 300-nt read that is a suffix of a 1 500-nt centre): the
  // one-word format of k_nw_ad (8-bit delta, windows <= 127 cells) does not hold it.
  uint2 *runs = (uint2 *)abase;
  double *fac = (double *)(abase + AD_RCAP * 8);           // aliases the centre bytes once the transition codes exist
  uint8_t *cbytes = abase + AD_RCAP * 8 + G.pad;
  uint8_t *rbytes = cbytes + G.seqbytes;
  uint8_t *tcode = abase + AD_RCAP * 8 + 2 * G.seqbytes;
  __syncthreads();
  const SampleDev &S = a.S;
  const int gwave = blockIdx.x * 4 + wib, nwaves = gridDim.x * 4;
  uint32_t *pg = a.ptr_scr + (size_t)gwave * a.ptr_wpw;    // [16-step block][cell pair 0..3][lane]
  const int nwork = a.nwork_dev ? *a.nwork_dev : a.nwork_host;
  const int SENT = a.ap.sentinel, MATCH = a.ap.match, MISMATCH = a.ap.mismatch, GAP = a.ap.gap, B = a.ap.band;
  for (int chunk = gwave; chunk * APW < nwork; chunk += nwaves) {
    const int idx = chunk * APW + al;
    const int c = a.chunk_centre ? a.chunk_centre[chunk] : a.centre;
    int r = idx < nwork ? a.work[idx] : -1;
    const bool active = r >= 0;
    if (!active) r = c;
    const int L1 = S.len[c], L2 = S.len[r];
    const int lband = B + (L1 > L2 ? L1 - L2 : 0), rband = B + (L2 > L1 ? L2 - L1 : 0);
    const int W = lband + rband + 1;
    const int sft = lband & 1, lbs = lband + sft;          // phase cell
    const int T = active ? L1 + L2 : -1;
    if (!ghost) {
      for (int p = g; p < L1; p += GL) cbytes[p] = (uint8_t)base_at(S.seq2 + (size_t)c * S.W2, p);
      for (int p = g; p < L2; p += GL) rbytes[p] = (uint8_t)base_at(S.seq2 + (size_t)r * S.W2, p);
    }
    const uint8_t *qrow = S.qual + (size_t)r * S.LQ;
    // aligned view of the unique on its centre (final pass / birth substitutions): centre positions facing a gap stay 0
    const size_t vr = a.view_by_chunk ? (size_t)chunk : (size_t)r;
    if (a.view && active && !ghost)
      for (int p = g; p < L1; p += GL) a.view[vr * a.LV + p] = 0;
    int Tmax = T;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));
    if (Tmax >= 0) {
      int d[8];
      uint32_t pw[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 8; q++) d[q] = SENT;
      uint32_t kmask = 0;
      if (!ghost)
        for (int q = 0; q < 8; q++) kmask |= (uint32_t)(8 * g + q >= sft && 8 * g + q < W + sft) << q;
      int gs[8];
#pragma unroll
      for (int q = 0; q < 8; q++) gs[q] = ((kmask >> q) & 1u) ? (DEF ? -8 : GAP) : AD_OOB;
      const bool g_first = g == 0, g_last = ghost || g == GL - 1;
      int I = (lbs >> 1) - 4 * g, J = -I;
      uint32_t cwin = 0, rwin = 0;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        cwin |= (uint32_t)cbytes[I - m - 1] << (8 * m);
        rwin |= (uint32_t)rbytes[J + m - 1] << (8 * m);
      }
      // steady state [tA, tB): every in-band cell of every alignment of the wave is interior and off the last row / column
      int tA = (lband > rband ? lband : rband) + 2, tB = min(2 * L1 - lband, 2 * L2 - rband);
      if (T < 0) { tA = 0; tB = 0x3FFFFFFF; }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { tA = max(tA, __shfl_xor(tA, o, 64)); tB = min(tB, __shfl_xor(tB, o, 64)); }
      tA = gcn_readfirstlane(tA);
      tB = gcn_readfirstlane(tB);
      Tmax = gcn_readfirstlane(Tmax);
#define ADW_EVEN(LEANV, FS)                                                                                              \
  {                                                                                                                      \
    const uint32_t nxt = rbytes[J + 3];                                                                                  \
    int nb = gcn_wave_shr1<false>(SENT, d[7]);   /* lane-1's last cell (wave_shr:1) */                                   \
    if (g_first) nb = SENT;                                                                                              \
    adw_step<0, LEANV, DEF>(d, pw, (FS), cwin ^ rwin, nb, kmask, gs, I, J, L1, L2, SENT, MATCH, MISMATCH, GAP);              \
    rwin = (rwin >> 8) | (nxt << 24);                                                                                    \
  }
#define ADW_ODD(LEANV, FS)                                                                                               \
  {                                                                                                                      \
    const uint32_t nxt = cbytes[I];                                                                                      \
    int nb = gcn_wave_shl1<false>(SENT, d[0]);   /* lane+1's first cell (wave_shl:1) */                                  \
    if (g_last) nb = SENT;                                                                                               \
    adw_step<1, LEANV, DEF>(d, pw, (FS), cwin ^ rwin, nb, kmask, gs, I, J + 1, L1, L2, SENT, MATCH, MISMATCH, GAP);          \
    cwin = (cwin << 8) | nxt;                                                                                            \
    I++; J++;                                                                                                            \
  }
#define ADW_FLUSH(TT)                                                                                                    \
  {                                                                                                                      \
    uint32_t *dst = pg + (size_t)((TT) >> 4) * 256 + lane;                                                               \
    dst[0] = pw[0]; dst[64] = pw[1]; dst[128] = pw[2]; dst[192] = pw[3];                                                 \
    pw[0] = pw[1] = pw[2] = pw[3] = 0;                                                                                   \
  }
      int t = 0;
      // leading steps with the matrix-edge logic, up to the first 16-step boundary inside the steady state
      for (; t <= Tmax && (t < tA || (t & 15) != 0); t++) {
        if ((t & 1) == 0) ADW_EVEN(false, (t & 15) << 1) else ADW_ODD(false, (t & 15) << 1)
        if ((t & 15) == 15) ADW_FLUSH(t)
      }
      for (; t + 16 <= tB && t + 15 <= Tmax; t += 16) {
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          ADW_EVEN(true, 2 * u)
          ADW_ODD(true, 2 * u + 2)
        }
        ADW_FLUSH(t)
      }
      for (; t <= Tmax; t++) {
        if ((t & 1) == 0) ADW_EVEN(false, (t & 15) << 1) else ADW_ODD(false, (t & 15) << 1)
        if ((t & 15) == 15) ADW_FLUSH(t)
      }
      if (((t - 1) & 15) != 15) ADW_FLUSH(t - 1)
#undef ADW_EVEN
#undef ADW_ODD
#undef ADW_FLUSH
    }
    // ---- traceback (first lane of each group) in chunks of <= AD_RCAP merged runs; all lanes expand the runs
    //      into one transition code per raw position (run format as in k_nw_ad)
    int ti = L1, tj = L2;
    bool done = !active;
    uint32_t h = 0;
    int guard = L1 + L2 + 2;
    const int gl0 = al * GL;
    while (true) {
      int nruns = 0;
      uint2 last = make_uint2(0u, 0u);                     // the leader's pending (mergeable) run, n == 0: none
      auto push = [&](int lo, int n, uint32_t dl) {
        const int llo = last.x & 4095, ln = (last.x >> 12) & 4095;
        if (ln) {
          if (last.y == dl && lo + n == llo) { last.x = (uint32_t)lo | ((uint32_t)(ln + n) << 12); return; }
          runs[nruns++] = last;
        }
        last = make_uint2((uint32_t)lo | ((uint32_t)n << 12), dl);
      };
      // as in k_nw_ad: the path is walked by the group's first lane, every diagonal stretch is measured by the whole group
      // (lane q reads the pointer word q blocks of 16 steps further back in the path's cell pair) - here each of those
      // reads is an HBM/L2 access, so one round replaces up to GL dependent memory round trips
      const bool lead = g == 0 && !ghost;
      for (;;) {
        const bool act = lead && !done && (ti > 0 || tj > 0) && nruns < AD_RCAP - 2 && guard > 0;
        if (!__any(act)) break;
        const int gact = __shfl((int)act, gl0, 64);
        const int tt = __shfl(ti + tj, gl0, 64), kkb = __shfl(tj - ti + lbs, gl0, 64);
        const int f0 = tt & 15, widx = (tt >> 4) - g;
        uint32_t word = 0x55555555u;
        if (gact && !ghost && widx >= 0) word = pg[(size_t)widx * 256 + ((kkb >> 1) & 3) * 64 + gl0 + (kkb >> 3)];
        const int ftop = g == 0 ? f0 : 14 + (f0 & 1);
        const uint32_t x = word ^ 0x55555555u;
        uint32_t nz = (x | (x >> 1)) & 0x55555555u;
        nz &= (f0 & 1) ? 0x44444444u : 0x11111111u;
        nz &= (ftop == 15) ? 0xFFFFFFFFu : ((1u << ((ftop + 1) << 1)) - 1u);
        const bool st = nz != 0;
        const int fb = st ? (31 - __clz(nz)) >> 1 : 0;
        const int ng = st ? (ftop - fb) >> 1 : (ftop >> 1) + 1;
        const uint32_t pst = (word >> (fb << 1)) & 3u;
        const unsigned long long bal = (__ballot(st && !ghost) >> gl0) & (GL == 64 ? ~0ull : ((1ull << (GL & 63)) - 1ull));
        const int qs = bal ? __builtin_ctzll(bal) : GL;
        const int srcl = gl0 + (qs < GL ? qs : 0);
        const int n0 = __shfl(ng, gl0, 64), nq = __shfl(ng, srcl, 64);
        const uint32_t pq = (uint32_t)__shfl((int)pst, srcl, 64);
        if (act) {
          guard--;
          int n = qs == 0 ? n0 : n0 + 8 * (qs - 1) + (qs < GL ? nq : 0);
          const int room = ti < tj ? ti : tj;
          const bool clamped = n > room;
          if (clamped) n = room;
          if (clamped) atomicOr(S.nw_flag, 1);             // (cannot happen: first row / column cells carry axis pointers)
          if (n > 0) {
            push(tj - n, n, (uint32_t)(ti - tj + ADW_DL0));
            ti -= n; tj -= n;
          }
          if (qs < GL && !clamped && (ti > 0 || tj > 0)) {
            if (pq == 2u) { tj--; push(tj, 1, ADW_GAP); }
            else ti--;
          }
        }
      }
      if (lead && active) {
        if ((last.x >> 12) & 4095) runs[nruns++] = last;
        if (!(ti > 0 || tj > 0)) done = true;
        else if (guard <= 0) { done = true; atomicOr(S.nw_flag, 1); }   // bounded walk tripped: report, never return a wrong lambda silently
      }
      nruns = __shfl(nruns, gl0, 64);
      done = __shfl((int)done, gl0, 64) != 0;
      int nrmax = nruns;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) nrmax = max(nrmax, __shfl_xor(nrmax, o, 64));
      for (int ri = 0; ri < nrmax; ri++) {
        if (ri < nruns && !ghost) {
          const uint2 dsc = runs[ri];
          const int lo = dsc.x & 4095, n = (dsc.x >> 12) & 4095;
          const int dl = (int)dsc.y - ADW_DL0;
          for (int pj = lo + g; pj < lo + n; pj += GL) {
            const uint32_t rb = rbytes[pj];
            uint32_t tc = 5u * rb;
            if (dsc.y != ADW_GAP) {
              const uint32_t cb = cbytes[pj + dl];
              tc = 4u * cb + rb;
              h += (cb != rb);
              if (a.view && active)
                a.view[vr * a.LV + pj + dl] = (uint16_t)(0x8000u | (rb << 8) | (a.ap.use_quals ? qrow[pj] : 0));
            }
            tcode[pj] = (uint8_t)tc;
          }
        }
      }
      if (__all(done)) break;
    }
    uint32_t *hsum = (uint32_t *)runs;                      // the run buffer is free by now
    gcn_wave_sync();                                        // (the last run descriptors have been read by every lane)
    if (g == 0 && !ghost) hsum[0] = 0;
    gcn_wave_sync();
    if (!ghost && h) atomicAdd(&hsum[0], h);
    gcn_wave_sync();
    h = hsum[0];
    // ---- lambda: factors in chunks of G.fch positions (all lanes), multiplied in raw-position order by one lane ----
    int L2max = active ? L2 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) L2max = max(L2max, __shfl_xor(L2max, o, 64));
    double l = 1.0;
    for (int base = 0; base < L2max; base += G.fch) {
      const int hi = min(L2, base + G.fch);
      if (!ghost && active)
        for (int pj = base + g; pj < hi; pj += GL) {
          const uint32_t q = a.ap.use_quals ? qrow[pj] : 0u;
          fac[pj - base] = s_err[(uint32_t)tcode[pj] * a.ap.ncol + q];
        }
      gcn_wave_sync();
      if (g == 0 && active && !ghost) {
        const int n = hi - base;
        int x = 0;
        for (; x + 8 <= n; x += 8) {
          const double f0 = fac[x], f1 = fac[x + 1], f2 = fac[x + 2], f3 = fac[x + 3];
          const double f4 = fac[x + 4], f5 = fac[x + 5], f6 = fac[x + 6], f7 = fac[x + 7];
          l = l * f0; l = l * f1; l = l * f2; l = l * f3; l = l * f4; l = l * f5; l = l * f6; l = l * f7;
        }
        for (; x < n; x++) l = l * fac[x];
      }
      gcn_wave_sync();                                      // (the next chunk's factors overwrite these)
    }
    if (g == 0 && active && !ghost) { a.lam[r] = l; a.ham[r] = h; }
  }
}

// Does the wide anti-diagonal kernel apply to this sample / band?  (run descriptors hold 12-bit positions)
bool nw_adw_ok(const SampleDev &S, const AlignParams &ap) {
  return ap.plain() && ap.band > 0 && S.maxlen <= 4095 && 2 * ap.band + (S.maxlen - S.minlen) + 2 <= 512;
}
size_t nw_adw_ptr_words_per_wave(const SampleDev &S, const AlignParams &ap) {
  return (size_t)adw_geom(ap.band, S.maxlen, S.minlen).nblk16 * 256;
}
int nw_adw_waves(const SampleDev &S, const AlignParams &ap, int nwork) {
  const AdwGeom G = adw_geom(ap.band, S.maxlen, S.minlen);
  return ((nwork + G.APW - 1) / G.APW + 3) & ~3;
}
void launch_nw_adw(const SampleDev &S, int centre, const int32_t *d_chunk_centre, const int32_t *d_work, const int32_t *d_nwork,
                   int nwork_host, const AlignParams &ap, const double *d_err, uint32_t *d_ptr_scr, size_t ptr_wpw,
                   int scr_waves, double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV, int view_by_chunk,
                   hipStream_t st) {
  const int maxwork = d_nwork ? S.N : nwork_host;
  if (maxwork <= 0) return;
  NwArgs a;
  memset(&a, 0, sizeof a);
  a.S = S; a.centre = centre; a.chunk_centre = d_chunk_centre; a.work = d_work; a.nwork_dev = d_nwork;
  a.nwork_host = nwork_host; a.ap = ap; a.err = d_err; a.lam = d_lambda; a.ham = d_ham;
  a.ptr_scr = d_ptr_scr; a.ptr_wpw = ptr_wpw;
  a.view = d_view; a.LV = LV; a.view_by_chunk = view_by_chunk;
  const AdwGeom G = adw_geom(ap.band, S.maxlen, S.minlen);
  const size_t lds = (size_t)16 * ap.ncol * 8 + (size_t)4 * G.APW * G.per_al_bytes;
  const int waves = (maxwork + G.APW - 1) / G.APW;
  const int grid = std::max(1, std::min((waves + 3) / 4, scr_waves / 4));
  const bool def = ap.match == 5 && ap.mismatch == -4 && ap.gap == -8 && ap.sentinel == -32760;
#define D2_LAUNCH_ADW(GLV, DEFV)                                                                                          \
  do {                                                                                                                    \
    static size_t attr_set[64] = {0};                                                                                     \
    int dev_ = 0;                                                                                                         \
    (void)hipGetDevice(&dev_);                                                                                            \
    if (lds > attr_set[dev_ & 63]) {                                                                                      \
      (void)hipFuncSetAttribute((const void *)k_nw_adw<GLV, DEFV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_set[dev_ & 63] = lds;                                                                                          \
    }                                                                                                                     \
    hipLaunchKernelGGL((k_nw_adw<GLV, DEFV>), dim3(grid), dim3(256), lds, st, a, G);                                      \
  } while (0)
  if (G.GL == 21) { if (def) D2_LAUNCH_ADW(21, true); else D2_LAUNCH_ADW(21, false); }
  else if (G.GL == 32) { if (def) D2_LAUNCH_ADW(32, true); else D2_LAUNCH_ADW(32, false); }
  else { if (def) D2_LAUNCH_ADW(64, true); else D2_LAUNCH_ADW(64, false); }
#undef D2_LAUNCH_ADW
}
int nw_adw_apw(const SampleDev &S, const AlignParams &ap) { return adw_geom(ap.band, S.maxlen, S.minlen).APW; }
size_t nw_adw_lds_bytes(const SampleDev &S, const AlignParams &ap) {
  const AdwGeom G = adw_geom(ap.band, S.maxlen, S.minlen);
  return (size_t)16 * ap.ncol * 8 + (size_t)4 * G.APW * G.per_al_bytes;
}

int nw_class(int band, int maxlen, int minlen) {
  if (band < 0) return 0;
  const int W = 2 * band + (maxlen - minlen) + 1;
  if (W <= 33) return 33;
  if (W <= 65) return 65;
  if (W <= 129) return 129;
  if (W <= 193) return 193;   // ragged long reads (configs[4]): one wave per SIMD, the band row still in registers
  if (W <= 257) return 257;
  return 0;
}

size_t nw_ptr_words_per_wave(int wclass, int band, int maxlen, int minlen) {
  int W = wclass;
  if (wclass == 0) W = (band < 0) ? (2 * maxlen + 1) : (2 * band + (maxlen - minlen) + 1);
  const int npw = (2 * W + 31) / 32;
  return (size_t)(maxlen + 1) * npw * 64;
}

void launch_nw(const SampleDev &S, int wclass, int centre, const int32_t *d_chunk_centre, const int32_t *d_work,
               const int32_t *d_nwork, int nwork_host, const AlignParams &ap, const double *d_err, const NwScratch &scr,
               double *d_lambda, uint32_t *d_ham, uint16_t *d_view, int LV, int view_by_chunk, uint8_t *d_moves,
               int moves_stride, int32_t *d_nmoves, hipStream_t st, const int32_t *d_pair_centre) {
  int maxwork = d_nwork ? S.N : nwork_host;
  if (maxwork <= 0) return;
  int waves = std::min((maxwork + 63) / 64, scr.nwaves);
  int grid = (waves + 3) / 4;
  if (grid * 4 > scr.nwaves) grid = scr.nwaves / 4;
  if (grid < 1) grid = 1;
  NwArgs a;
  memset(&a, 0, sizeof a);
  a.S = S; a.centre = centre; a.chunk_centre = d_chunk_centre; a.pair_centre = d_pair_centre; a.work = d_work; a.nwork_dev = d_nwork;
  a.nwork_host = nwork_host; a.ap = ap; a.err = d_err; a.ptr_scr = scr.ptr; a.t_scr = scr.tcode; a.row_scr = scr.rows;
  a.ptr_wpw = scr.ptr_words_per_wave; a.t_wpw = scr.t_words_per_wave; a.row_wpw = scr.row_words_per_wave;
  a.lam = d_lambda; a.ham = d_ham; a.view = d_view; a.LV = LV; a.view_by_chunk = view_by_chunk; a.moves = d_moves; a.moves_stride = moves_stride;
  a.nmoves = d_nmoves;
  size_t lds = (size_t)16 * ap.ncol * sizeof(double);
#define D2_NW_CLASS(W)                                                                                              \
  case W:                                                                                                           \
    if (d_pair_centre) hipLaunchKernelGGL((k_nw<W, true, false>), dim3(grid), dim3(256), lds, st, a);               \
    else if (ap.plain()) hipLaunchKernelGGL((k_nw<W, false, true>), dim3(grid), dim3(256), lds, st, a);             \
    else hipLaunchKernelGGL((k_nw<W, false, false>), dim3(grid), dim3(256), lds, st, a);                            \
    break;
  switch (ap.hi_off ? 0 : wclass) {   // (two-plane letters: the generic kernel)
    D2_NW_CLASS(33) D2_NW_CLASS(65) D2_NW_CLASS(129) D2_NW_CLASS(193) D2_NW_CLASS(257)
    default: {
      int Wgen = (ap.band < 0) ? (2 * S.maxlen + 1) : (2 * ap.band + (S.maxlen - S.minlen) + 1);
      if (d_pair_centre) hipLaunchKernelGGL(k_nw_gen<true>, dim3(grid), dim3(256), lds, st, a, Wgen);
      else hipLaunchKernelGGL(k_nw_gen<false>, dim3(grid), dim3(256), lds, st, a, Wgen);
    }
  }
#undef D2_NW_CLASS
}

// ================================================================================================
// Device-resident partition state.  Everything that is O(nraw) per round in the
// reference's serial bookkeeping runs here; the host only replays the (few) membership moves to keep
// the reference's slot order, and takes the C-sized decisions.
//
// Stored comparisons (Bi::comp, dada.h:104) are kept per UNIQUE as a linked list of nodes
// (cluster i, lambda, hamming): a round appends at most one node per unique, so the list is in
// descending cluster order and b_shuffle2's arg-max (cluster.cpp:229-239) is one walk per unique.

// store filter of b_compare_parallel (cluster.cpp:179-201) for the round of cluster `ci`
__global__ __launch_bounds__(256) void k_store(PartState P, SampleDev S, int ci, int centre, double total_reads,
                                               const double *__restrict__ lam, const uint32_t *__restrict__ ham,
                                               const int32_t *__restrict__ round_counters, const uint8_t *__restrict__ cls,
                                               int32_t *__restrict__ zero2) {
  __shared__ int s_n, s_base, s_cls[2];
  if (blockIdx.x == 0 && threadIdx.x < 2 && zero2) zero2[threadIdx.x] = 0;   // this round's shuffle counters
  const uint32_t creads = S.reads[centre];
  if (blockIdx.x == 0 && threadIdx.x < 2)   // fold this round's work-list sizes into the run totals
    atomicAdd((unsigned long long *)&P.totals[threadIdx.x], (unsigned long long)round_counters[threadIdx.x]);
  if (threadIdx.x < 2) s_cls[threadIdx.x] = 0;
  __syncthreads();
  int my_shroud = 0, my_skip = 0;
  for (int base = S.r_lo + blockIdx.x * 256; base < S.r_hi; base += gridDim.x * 256) {
    const int r = base + threadIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bool keep = false;
    double l = 0.0;
    uint32_t h = 0;
    int pos = 0;
    if (r < S.r_hi) {
      const uint8_t cl = cls[r];
      my_shroud += (cl == CLS_SHROUD);
      my_skip += (cl == CLS_SKIP);
      if (cl == CLS_SHROUD || cl == CLS_SKIP) { l = 0.0; h = 0xFFFFFFFFu; }   // NULL sub (cluster.cpp:139-143)
      else { l = lam[r]; h = ham[r]; }
      if (!(l >= 0.0 && l <= 1.0)) atomicOr(P.err_flag, 1);          // "Lambda out-of-range error." (cluster.cpp:184)
      const double em = P.E_minmax[r];
      keep = l * total_reads > em;                                     // this cluster could attract this raw
      if (keep) {
        if (l * creads > em) P.E_minmax[r] = l * creads;
        pos = atomicAdd(&s_n, 1);
        if (ci == 0 || r == centre) { P.comp_i[r] = ci; P.comp_lam[r] = l; P.comp_ham[r] = h; }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_n ? atomicAdd(P.node_count, s_n) : 0;
    __syncthreads();
    if (keep) {
      const int n = s_base + pos;
      if (n < P.node_cap) {
        P.node_i[n] = ci; P.node_lam[n] = l; P.node_ham[n] = h;
        P.node_next[n] = P.head[r];
        P.head[r] = n;
      } else atomicOr(P.err_flag, 2);
    }
    __syncthreads();
  }
  // class statistics of the round (nshroud / greedy skips, dada.h:113-114)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { my_shroud += __shfl_xor(my_shroud, o, 64); my_skip += __shfl_xor(my_skip, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cls[0], my_shroud); atomicAdd(&s_cls[1], my_skip); }
  __syncthreads();
  if (threadIdx.x < 2 && s_cls[threadIdx.x])
    atomicAdd((unsigned long long *)&P.totals[2 + threadIdx.x], (unsigned long long)s_cls[threadIdx.x]);
}

// Arguments of the store filter when it rides in front of the round's first shuffle (k_shuffle<true>)
struct StoreArgs {
  int ci, centre;
  double total_reads;
  const double *lam;
  const uint32_t *ham;
  const int32_t *round_counters;
  const uint8_t *cls;
};

// b_shuffle2 (cluster.cpp:210-266): per unique, the stored comparison with the largest expected
// reads lambda * bi[i].reads (reads as of the start of the call; ties go to the lowest cluster),
// then the move unless the unique is its partition's centre.  Movers are reported to the host,
// which replays them in the reference's order to maintain slots.
// STORE: the round's store filter (k_store's body, cluster.cpp:179-201) runs first on the same unique - the new
// comparison is the head of the unique's list, so the arg-max starts from it in registers.
// (No "last block refreshes the snapshot" here: a device-scope fence per block writes back and invalidates the
// XCD's L2 on this part - measured 1.7x slower at 1e6 uniques - so the snapshot stays a copy between launches.)
// check_only: count the uniques that WOULD move, apply nothing (then the live partition reads can stand in for the
// snapshot: nothing changes them during the kernel).  The speculative second shuffle of a round is such a check.
// Reads deltas of the movers are accumulated per block in LDS (partitions < DELTA_TAB, rounds_common.h) and flushed with one
// global atomic per touched partition: thousands of movers join the same new partition in a round, and same-address
// device atomics serialise.
template <bool STORE>
__global__ __launch_bounds__(256) void k_shuffle(PartState P, SampleDev S, const uint32_t *__restrict__ creads_snap,
                                                 int32_t *__restrict__ movers, int32_t *__restrict__ nmovers,
                                                 int32_t *__restrict__ inl, int check_only, int nclust, StoreArgs sa) {
  __shared__ int s_n, s_base, s_sn, s_sbase, s_cls[2];
  __shared__ int32_t s_delta[DELTA_TAB];
  const int ntab = nclust < DELTA_TAB ? nclust : DELTA_TAB;
  if (!check_only) for (int k = threadIdx.x; k < ntab; k += 256) s_delta[k] = 0;
  uint32_t screads = 0;
  int my_shroud = 0, my_skip = 0;
  if (STORE) {
    screads = S.reads[sa.centre];
    if (blockIdx.x == 0 && threadIdx.x < 2)   // fold this round's work-list sizes into the run totals
      atomicAdd((unsigned long long *)&P.totals[threadIdx.x], (unsigned long long)sa.round_counters[threadIdx.x]);
    if (threadIdx.x < 2) s_cls[threadIdx.x] = 0;
  }
  for (int base = S.r_lo + blockIdx.x * 256; base < S.r_hi; base += gridDim.x * 256) {
    const int r = base + threadIdx.x;
    if (threadIdx.x == 0) { s_n = 0; s_sn = 0; }
    __syncthreads();
    double best_e = -1.0, best_l = 0.0;
    int best_i = 0x7FFFFFFF;
    uint32_t best_h = 0;
    int head = -1;
    if (r < S.r_hi) head = P.head[r];
    // ---- store filter of this round's comparison ----
    bool keep = false;
    double l = 0.0;
    uint32_t h = 0;
    int spos = 0;
    if (STORE && r < S.r_hi) {
      const uint8_t cl = sa.cls[r];
      my_shroud += (cl == CLS_SHROUD);
      my_skip += (cl == CLS_SKIP);
      if (cl == CLS_SHROUD || cl == CLS_SKIP) { l = 0.0; h = 0xFFFFFFFFu; }   // NULL sub (cluster.cpp:139-143)
      else { l = sa.lam[r]; h = sa.ham[r]; }
      if (!(l >= 0.0 && l <= 1.0)) atomicOr(P.err_flag, 1);          // "Lambda out-of-range error." (cluster.cpp:184)
      const double em = P.E_minmax[r];
      keep = l * sa.total_reads > em;                                  // this cluster could attract this raw
      if (keep) {
        if (l * screads > em) P.E_minmax[r] = l * screads;
        spos = atomicAdd(&s_sn, 1);
        if (r == sa.centre) { P.comp_i[r] = sa.ci; P.comp_lam[r] = l; P.comp_ham[r] = h; }
        best_e = l * creads_snap[sa.ci]; best_i = sa.ci; best_l = l; best_h = h;
      }
    }
    if (STORE) {
      __syncthreads();
      if (threadIdx.x == 0) s_sbase = s_sn ? atomicAdd(P.node_count, s_sn) : 0;
      __syncthreads();
      if (keep) {
        const int n = s_sbase + spos;
        if (n < P.node_cap) {
          P.node_i[n] = sa.ci; P.node_lam[n] = l; P.node_ham[n] = h;
          P.node_next[n] = head;
          P.head[r] = n;
        } else atomicOr(P.err_flag, 2);
      }
    }
    // ---- arg-max over the stored comparisons and the move ----
    bool move = false;
    int from = 0, to = 0, pos = 0;
    if (r < S.r_hi) {
      for (int n = head; n >= 0; n = P.node_next[n]) {
        const int i = P.node_i[n];
        const double nl = P.node_lam[n], e = nl * creads_snap[i];
        if (e > best_e || (e == best_e && i < best_i)) { best_e = e; best_i = i; best_l = nl; best_h = P.node_ham[n]; }
      }
      from = P.clust_of[r];
      if (best_i != 0x7FFFFFFF && best_i != from && r != P.centre_of[from]) {
        move = !check_only;
        to = best_i;
        pos = atomicAdd(&s_n, 1);
        if (move) {
          P.clust_of[r] = to;
          P.comp_i[r] = to; P.comp_lam[r] = best_l; P.comp_ham[r] = best_h;
          const uint32_t rd = S.reads[r];
          if (to < ntab) atomicAdd(&s_delta[to], (int32_t)rd); else atomicAdd(&P.creads[to], rd);
          if (from < ntab) atomicSub(&s_delta[from], (int32_t)rd); else atomicSub(&P.creads[from], rd);
          P.update_e[to] = 1; P.update_e[from] = 1;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_n ? atomicAdd(nmovers, s_n) : 0;
    __syncthreads();
    if (move) {
      const int k = s_base + pos;
      int32_t *m = movers + 3 * (size_t)k;
      m[0] = r; m[1] = from; m[2] = to;
      if (inl && k < MOVERS_INLINE) { inl[3 * k] = r; inl[3 * k + 1] = from; inl[3 * k + 2] = to; }
    }
    __syncthreads();
  }
  if (!check_only) {   // (the loop ends on a barrier: every delta of the block is in the table)
    for (int k = threadIdx.x; k < ntab; k += 256) {
      const int32_t dlt = s_delta[k];
      if (dlt) atomicAdd(&P.creads[k], (uint32_t)dlt);
    }
  }
  if (STORE) {   // class statistics of the round (nshroud / greedy skips, dada.h:113-114)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { my_shroud += __shfl_xor(my_shroud, o, 64); my_skip += __shfl_xor(my_skip, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cls[0], my_shroud); atomicAdd(&s_cls[1], my_skip); }
    __syncthreads();
    if (threadIdx.x < 2 && s_cls[threadIdx.x])
      atomicAdd((unsigned long long *)&P.totals[2 + threadIdx.x], (unsigned long long)s_cls[threadIdx.x]);
  }
}

// (dev_get_pA, BudKey, bud_better: rounds_common.h)
static __device__ __forceinline__ bool bud_candidate(const PartState &P, const SampleDev &S, int r, BudParams bp) {
  if (P.slot0[r]) return false;                                          // r = 0 is skipped as "the centre" (:285)
  const uint32_t reads = S.reads[r];
  if (reads < (uint32_t)bp.min_abund) return false;
  if ((int)P.comp_ham[r] < bp.min_hamming) return false;
  if (!(bp.min_fold <= 1 || ((double)reads) >= bp.min_fold * P.comp_lam[r] * P.creads[P.clust_of[r]])) return false;
  return true;
}

// b_p_update fused with the first stage of b_bud: every thread refreshes p / lock of its uniques (pval.cpp:14-40)
// and folds them straight into the block's (p, reads) minimum.  Block 0 also resets the result block's tie counters.
// check_cnt (optional): the same pass first asks, per unique, whether one more b_shuffle2 call would move it (arg-max
// over its stored comparisons with the live partition reads - nothing moves in this kernel) and counts those uniques.
// A non-zero count cancels the evaluation (k_bud_ties and k_auto_birth test it); what this kernel wrote is then either
// recomputed by the evaluation that follows the real shuffle (p: the partitions' update flags are still set) or was
// never committed (locks go to lock_tmp and are committed by k_bud_ties).
__global__ __launch_bounds__(256) void k_pupdate_budmin(PartState P, SampleDev S, int greedy, int detect_singletons, BudParams bp,
                                                        BudKey init, BudKey *__restrict__ partial, BudOut *__restrict__ out,
                                                        uint8_t *__restrict__ lock_tmp, int32_t *__restrict__ check_cnt,
                                                        const int32_t *__restrict__ guard) {
  __shared__ BudKey s_k[2][4];
  __shared__ int s_would;
  const bool cancelled = guard && *guard != 0;   // speculative launch: a later shuffle still moved uniques, redo after it
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out->nties[0] = 0; out->nties[1] = 0;
    if (cancelled) { out->valid = 0; out->found[0] = 0; out->found[1] = 0; }
  }
  if (cancelled) return;
  if (threadIdx.x == 0) s_would = 0;
  __syncthreads();
  BudKey b0 = init, b1 = init;
  int would = 0;
  for (int r = S.r_lo + blockIdx.x * 256 + threadIdx.x; r < S.r_hi; r += gridDim.x * 256) {
    const int cl = P.clust_of[r];
    if (check_cnt) {
      double best_e = -1.0;
      int best_i = 0x7FFFFFFF;
      for (int n = P.head[r]; n >= 0; n = P.node_next[n]) {
        const int i = P.node_i[n];
        const double e = P.node_lam[n] * P.creads[i];
        if (e > best_e || (e == best_e && i < best_i)) { best_e = e; best_i = i; }
      }
      would += (best_i != 0x7FFFFFFF && best_i != cl && r != P.centre_of[cl]);
    }
    const double l = P.comp_lam[r];
    const uint32_t reads = S.reads[r];
    double p = P.p[r];
    if (P.update_e[cl]) { p = dev_get_pA(reads, S.prior[r] != 0, detect_singletons != 0, l, P.comp_ham[r], P.creads[cl]); P.p[r] = p; }
    uint8_t lk = 0;
    if (greedy && P.check_locks[cl]) {
      const int c = P.centre_of[cl];
      const double E_center = S.reads[c] * l;
      lk = (E_center > reads) || (r == c);
    }
    lock_tmp[r] = lk;
    if (!bud_candidate(P, S, r, bp)) continue;
    if (bud_better(p, reads, b0)) { b0.p = p; b0.reads = reads; }
    if (S.prior[r] && bud_better(p, reads, b1)) { b1.p = p; b1.reads = reads; }
  }
  if (check_cnt) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) would += __shfl_xor(would, o, 64);
    if ((threadIdx.x & 63) == 0 && would) atomicAdd(&s_would, would);
  }
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
    for (int k = 1; k < 4; k++) {
      if (bud_better(s_k[0][k].p, s_k[0][k].reads, b0)) b0 = s_k[0][k];
      if (bud_better(s_k[1][k].p, s_k[1][k].reads, b1)) b1 = s_k[1][k];
    }
    partial[2 * blockIdx.x] = b0;
    partial[2 * blockIdx.x + 1] = b1;
    if (check_cnt && s_would) atomicAdd(check_cnt, s_would);
  }
}

// second stage: EVERY block reduces the block partials to the best keys (a few KB from L2 - cheaper than a launch of
// its own), then lists the candidates of its uniques whose key equals the best one (normally exactly one in the
// whole grid) as tie records.  Block 0 publishes the keys, gathers the error flag and the comparison-store fill
// level into the result block and clears the per-partition flags k_pupdate_budmin has consumed (pval.cpp:24,37).
__global__ __launch_bounds__(256) void k_bud_ties(PartState P, SampleDev S, BudParams bp, const BudKey *__restrict__ partial,
                                                  int nblocks, BudKey init, int nclust, BudOut *__restrict__ out,
                                                  int32_t *__restrict__ overflow0, int32_t *__restrict__ overflow1,
                                                  const uint8_t *__restrict__ lock_tmp, const int32_t *__restrict__ guard) {
  __shared__ BudKey s_k[2][4];
  if (guard && *guard != 0) {   // the evaluation is void (see k_pupdate_budmin): nothing is committed
    if (blockIdx.x == 0 && threadIdx.x == 0) { out->valid = 0; out->found[0] = 0; out->found[1] = 0; }
    return;
  }
  BudKey b0 = init, b1 = init;
  for (int k = threadIdx.x; k < nblocks; k += 256) {
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
  if ((threadIdx.x & 63) == 0) { s_k[0][threadIdx.x >> 6] = b0; s_k[1][threadIdx.x >> 6] = b1; }
  __syncthreads();
  b0 = s_k[0][0]; b1 = s_k[1][0];
  for (int k = 1; k < 4; k++) {
    if (bud_better(s_k[0][k].p, s_k[0][k].reads, b0)) b0 = s_k[0][k];
    if (bud_better(s_k[1][k].p, s_k[1][k].reads, b1)) b1 = s_k[1][k];
  }
  const bool found0 = bud_better(b0.p, b0.reads, init), found1 = bud_better(b1.p, b1.reads, init);
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < nclust; k += 256) { P.update_e[k] = 0; P.check_locks[k] = 0; }
    if (threadIdx.x == 0) {
      out->best_p[0] = b0.p; out->best_p[1] = b1.p;
      out->best_reads[0] = b0.reads; out->best_reads[1] = b1.reads;
      out->found[0] = found0; out->found[1] = found1;
      out->err_flag = *P.err_flag;
      out->node_count = *P.node_count;
      out->valid = 1;
    }
  }
  // candidates to list: exact ties of the best key, plus any other candidate whose non-zero p is within BUD_NEAR of it
  // (their order against the best one is the host's call, engine.h).  p == 0 (lambda == 0, pval.cpp:78, or a Poisson
  // tail below the smallest denormal) is taken as exact: a zero only ties with zeros, by reads.
  // (no window when the best p-value is clearly not significant: b_bud then gives no birth whatever the order)
  const bool sig0 = b0.p * S.N < 2.0 * bp.omegaA, sig1 = b1.p < 2.0 * bp.omegaP;
  const double thr0 = sig0 ? b0.p * (1.0 + BUD_NEAR) + 2e-323 : -1.0, thr1 = sig1 ? b1.p * (1.0 + BUD_NEAR) + 2e-323 : -1.0;
  for (int r = S.r_lo + blockIdx.x * 256 + threadIdx.x; r < S.r_hi; r += gridDim.x * 256) {
    if (lock_tmp[r]) P.lock[r] = 1;                                    // b_p_update's greedy locks (pval.cpp:26-36)
    if (!bud_candidate(P, S, r, bp)) continue;
    const double p = P.p[r];
    const uint32_t reads = S.reads[r];
    for (int track = 0; track < 2; track++) {
      if (track == 1 && !S.prior[r]) continue;
      const BudKey &bk = track ? b1 : b0;
      if (!(track ? found1 : found0)) continue;
      const bool exact = p == bk.p && reads == bk.reads;
      const bool near = !exact && p != 0.0 && p <= (track ? thr1 : thr0);
      if (!exact && !near) continue;
      const int k = atomicAdd(&out->nties[track], 1);
      if (k < BUD_TIES) {
        BudTie &t = out->ties[track][k];
        t.raw = r; t.comp_i = P.comp_i[r]; t.comp_ham = P.comp_ham[r]; t.comp_lam = P.comp_lam[r];
        t.from = P.clust_of[r]; t.from_reads = P.creads[t.from]; t.p = p;
      }
      (track ? overflow1 : overflow0)[k] = r;
    }
  }
}

// Applies a birth (cluster.cpp:313-347): the unique leaves `from`, becomes the only member and centre of the new
// partition; bi_assign_center unlocks it; both partitions are flagged.  Also: the coming round's shuffle counters are
// zeroed, the reads snapshot is refreshed, the new centre's k-mer record is built.  One block of 256 threads.
static __device__ __forceinline__ void apply_bud_body(PartState &P, SampleDev &S, uint32_t *creads_snap, int raw, int newi, int from,
                                                      uint32_t reads_new, uint32_t reads_from, uint32_t *ctab, int32_t *zero2,
                                                      uint32_t *cnt) {
  if (threadIdx.x < 2 && zero2) zero2[threadIdx.x] = 0;   // the coming round's shuffle counters
  // the coming round's first shuffle reads the partition reads as of its start: refresh the whole snapshot here
  for (int k = threadIdx.x; k < newi; k += 256) if (k != from) creads_snap[k] = P.creads[k];
  if (threadIdx.x == 0) {
    P.clust_of[raw] = newi;
    P.lock[raw] = 0;
    P.slot0[raw] = 1;
    P.creads[newi] = reads_new; creads_snap[newi] = reads_new;
    P.creads[from] = reads_from; creads_snap[from] = reads_from;
    P.centre_of[newi] = raw;
    P.update_e[newi] = 1; P.check_locks[newi] = 1;
    P.update_e[from] = 1;
  }
  centre_table_body(S, raw, ctab, cnt);   // the new centre's k-mer record for the round that follows
}
// ... decided by the host
__global__ __launch_bounds__(256) void k_apply_bud(PartState P, SampleDev S, uint32_t *creads_snap, int raw, int newi, int from,
                                                   uint32_t reads_new, uint32_t reads_from, uint32_t *__restrict__ ctab,
                                                   int32_t *__restrict__ zero2) {
  __shared__ uint32_t cnt[NKMER];
  apply_bud_body(P, S, creads_snap, raw, newi, from, reads_new, reads_from, ctab, zero2, cnt);
}
// ... decided on the device: the unambiguous case of b_bud (cluster.cpp:300-330) - exactly one best candidate and
// pA = p * nraw < OMEGA_A, evaluated with the host's own expression - so the next round's screen and alignments,
// enqueued behind this kernel, can start without waiting for the host.  Anything else (no birth, exact ties, prior
// births) is left to the host: next[0] = -1 turns the speculative kernels behind it into no-ops.
__global__ __launch_bounds__(256) void k_auto_birth(PartState P, SampleDev S, uint32_t *creads_snap, RoundOut *__restrict__ blk,
                                                    double omegaA, int newi, uint32_t *__restrict__ ctab,
                                                    int32_t *__restrict__ zero2, int32_t *__restrict__ next,
                                                    RoundOut *__restrict__ host_blk, int seq) {
  __shared__ uint32_t cnt[NKMER];
  __shared__ int s_ok;
  BudOut *out = &blk->bud;
  if (threadIdx.x == 0) {
    // (the margin keeps the decision on the host whenever the device p-value is within libm noise of OMEGA_A)
    const bool ok = out->valid && out->found[0] && out->nties[0] == 1 && (out->best_p[0] * S.N < omegaA * (1.0 - 1e-9));
    s_ok = ok;
    out->auto_applied = ok ? 1 : 0;
    next[0] = ok ? out->ties[0][0].raw : -1;
  }
  __syncthreads();
  if (s_ok) {
    const BudTie t = out->ties[0][0];
    const uint32_t reads_new = S.reads[t.raw];
    apply_bud_body(P, S, creads_snap, t.raw, newi, t.from, reads_new, t.from_reads - reads_new, ctab, zero2, cnt);
  }
  // publish the round's result block (mover counts + first movers, bud evaluation) to the host: plain stores to pinned
  // memory, then the sequence number - the host polls it instead of copying and synchronising
  __syncthreads();
  static_assert(sizeof(RoundOut) % 16 == 0, "RoundOut is copied as uint4");
  const uint4 *src = (const uint4 *)blk;
  uint4 *dst = (uint4 *)host_blk;
  for (int i = threadIdx.x; i < (int)(sizeof(RoundOut) / 16); i += 256) {
    uint4 v = src[i];
    if (i == 0) v.z = (uint32_t)(seq - 1);              // (word 2 of the block is `seq`: not yet)
    dst[i] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&host_blk->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// final per-unique p and the OMEGA_C decision (Rmain.cpp:238-252)
__global__ __launch_bounds__(256) void k_final_p(PartState P, SampleDev S, double omegaC, uint8_t *__restrict__ correct) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < S.r_lo || r >= S.r_hi) return;
  const int cl = P.clust_of[r];
  double p = 1.0;
  uint8_t ok = 1;
  if (r != P.centre_of[cl]) {
    p = pp::calc_pA((int)S.reads[r], P.comp_lam[r] * P.creads[cl], true);
    if (p < omegaC) ok = 0;
  }
  P.p[r] = p;
  correct[r] = ok;
}

// post-hoc partition p-value inputs (error.cpp:101-119): stored comparisons of partition i whose unique
// is the centre of another partition j -> (j, i, lambda); the host sums them in ascending i.
__global__ __launch_bounds__(256) void k_posthoc(PartState P, SampleDev S, int nnodes, const int32_t *__restrict__ cluster_of_centre,
                                                 const int32_t *__restrict__ node_raw, int32_t *__restrict__ out_ji,
                                                 double *__restrict__ out_lam, int32_t *__restrict__ nout, int cap) {
  (void)nnodes; (void)node_raw;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < S.r_lo || r >= S.r_hi) return;
  const int j = cluster_of_centre[r];
  if (j < 0) return;
  for (int n = P.head[r]; n >= 0; n = P.node_next[n]) {
    const int i = P.node_i[n];
    if (i == j) continue;
    const int k = atomicAdd(nout, 1);
    if (k < cap) { out_ji[2 * k] = j; out_ji[2 * k + 1] = i; out_lam[k] = P.node_lam[n]; }
  }
}

// dense (lambda, hamming) view for dada2hip_sample_compare: materialise the NULL-sub entries implied by cls[]
__global__ void k_fill_null(int n, const uint8_t *__restrict__ cls, double *__restrict__ lam, uint32_t *__restrict__ ham) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && (cls[r] == CLS_SHROUD || cls[r] == CLS_SKIP)) { lam[r] = 0.0; ham[r] = 0xFFFFFFFFu; }
}
__global__ void k_fill_f64(double *__restrict__ p, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_f64(double *d_p, size_t n, double v, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, d_p, n, v);
}
void launch_fill_null(int n, const uint8_t *d_cls, double *d_lam, uint32_t *d_ham, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_null, dim3((n + 255) / 256), dim3(256), 0, st, n, d_cls, d_lam, d_ham);
}

void launch_store(const PartState &P, const SampleDev &S, int ci, int centre, double total_reads, const double *d_lam,
                  const uint32_t *d_ham, const int32_t *d_round_counters, const uint8_t *d_cls, int32_t *d_zero2, hipStream_t st) {
  int grid = std::min((S.N + 255) / 256, 2048);
  hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, st, P, S, ci, centre, total_reads, d_lam, d_ham, d_round_counters, d_cls,
                     d_zero2);
}
void launch_shuffle(const PartState &P, const SampleDev &S, const uint32_t *d_creads_snap, int32_t *d_movers,
                    int32_t *d_nmovers, int32_t *d_inline, const StoreRound *store, int check_only, int nclust, hipStream_t st) {
  int grid = std::min((S.N + 255) / 256, 2048);
  StoreArgs sa{};
  if (store) {
    sa = StoreArgs{store->ci, store->centre, store->total_reads, store->lam, store->ham, store->round_counters, store->cls};
    hipLaunchKernelGGL(k_shuffle<true>, dim3(grid), dim3(256), 0, st, P, S, d_creads_snap, d_movers, d_nmovers, d_inline, check_only,
                       nclust, sa);
  } else
    hipLaunchKernelGGL(k_shuffle<false>, dim3(grid), dim3(256), 0, st, P, S, d_creads_snap, d_movers, d_nmovers, d_inline,
                       check_only, nclust, sa);
}
void launch_pupdate_bud(const PartState &P, const SampleDev &S, int greedy, int detect_singletons, const BudParams &bp,
                        double init_p, uint32_t init_reads, void *d_partial, BudOut *d_out, int32_t *d_over0, int32_t *d_over1,
                        int nclust, uint8_t *d_lock_tmp, int32_t *d_check_cnt, hipStream_t st) {
  BudKey init{init_p, init_reads};
  int grid = std::min((S.N + 255) / 256, 1024);
  hipLaunchKernelGGL(k_pupdate_budmin, dim3(grid), dim3(256), 0, st, P, S, greedy, detect_singletons, bp, init, (BudKey *)d_partial,
                     d_out, d_lock_tmp, d_check_cnt, (const int32_t *)nullptr);
  hipLaunchKernelGGL(k_bud_ties, dim3(std::min((S.N + 255) / 256, 512)), dim3(256), 0, st, P, S, bp, (const BudKey *)d_partial, grid,
                     init, nclust, d_out, d_over0, d_over1, (const uint8_t *)d_lock_tmp, (const int32_t *)d_check_cnt);
}
void launch_apply_bud(const PartState &P, const SampleDev &S, uint32_t *d_creads_snap, int raw, int newi, int from,
                      uint32_t reads_new, uint32_t reads_from, uint32_t *d_ctab, int32_t *d_zero2, hipStream_t st) {
  hipLaunchKernelGGL(k_apply_bud, dim3(1), dim3(256), 0, st, P, S, d_creads_snap, raw, newi, from, reads_new, reads_from, d_ctab,
                     d_zero2);
}
void launch_auto_birth(const PartState &P, const SampleDev &S, uint32_t *d_creads_snap, RoundOut *d_block, double omegaA, int newi,
                       uint32_t *d_ctab, int32_t *d_zero2, int32_t *d_next, RoundOut *h_block, int seq, hipStream_t st) {
  hipLaunchKernelGGL(k_auto_birth, dim3(1), dim3(256), 0, st, P, S, d_creads_snap, d_block, omegaA, newi, d_ctab, d_zero2, d_next,
                     h_block, seq);
}
void launch_final_p(const PartState &P, const SampleDev &S, double omegaC, uint8_t *d_correct, hipStream_t st) {
  hipLaunchKernelGGL(k_final_p, dim3((S.N + 255) / 256), dim3(256), 0, st, P, S, omegaC, d_correct);
}
void launch_posthoc(const PartState &P, const SampleDev &S, const int32_t *d_cluster_of_centre, int32_t *d_out_ji, double *d_out_lam,
                    int32_t *d_nout, int cap, hipStream_t st) {
  hipLaunchKernelGGL(k_posthoc, dim3((S.N + 255) / 256), dim3(256), 0, st, P, S, 0, d_cluster_of_centre, nullptr, d_out_ji,
                     d_out_lam, d_nout, cap);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_calc_pA(int n, const int32_t *__restrict__ reads, const double *__restrict__ E,
                          const uint8_t *__restrict__ prior, double *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pp::calc_pA(reads[i], E[i], prior ? prior[i] != 0 : false);
}

void launch_calc_pA(int n, const int32_t *d_reads, const double *d_E, const uint8_t *d_prior, double *d_out,
                    hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_calc_pA, dim3((n + 127) / 128), dim3(128), 0, st, n, d_reads, d_E, d_prior, d_out);
}

// ------------------------------------------------------------------------------------------------
// Output tables from the aligned views.  view[r][pos0] = 0x8000 | rawbase<<8 | q for centre positions
// aligned to a raw base, 0 for centre positions opposite a gap.
//   $subqual[t = 4*centre_base + raw_base][q] += reads    (error.cpp:152-167; int32 wrap-around as R's int)
//   $clusterquals: sum(q*reads), sum(reads) per (cluster, pos0)  (error.cpp:241-252); integer sums are
//   exact and order-independent, the host divides (the reference's fp64 sum of integers is exact < 2^53).
//   nsubs[r] = substitutions of the final alignment (for n0/n1, error.cpp:58-61).
__global__ __launch_bounds__(256) void k_final_tables(SampleDev S, const uint16_t *__restrict__ view, int LV,
                                                      const int32_t *__restrict__ cluster_of,
                                                      const int32_t *__restrict__ centre_of_cluster,
                                                      const uint8_t *__restrict__ correct, int ncol, int has_quals,
                                                      int32_t *__restrict__ trans, unsigned long long *__restrict__ qsum,
                                                      uint32_t *__restrict__ qn, int32_t *__restrict__ nsubs) {
  extern __shared__ uint32_t s_hist[];   // [16*ncol]
  const int nb = 16 * ncol;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s_hist[i] = 0;
  __syncthreads();
  // one wave per unique, lanes stride over centre positions
  const int lane = threadIdx.x & 63, gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  for (int r = S.r_lo + gwave; r < S.r_hi; r += nwaves) {
    const int cl = cluster_of[r], c = centre_of_cluster[cl];
    const int Lc = S.len[c];
    const uint32_t *crow = S.seq2 + (size_t)c * S.W2;
    const uint32_t reads = S.reads[r];
    const bool corr = correct[r] != 0;
    uint32_t ns = 0;
    for (int p = lane; p < Lc; p += 64) {
      const uint32_t v = view[(size_t)r * LV + p];
      if (v & 0x8000u) {
        const uint32_t rb = (v >> 8) & 3u, q = v & 255u, cb = base_at(crow, p);
        ns += (cb != rb);
        if (corr) {
          atomicAdd(&s_hist[(has_quals ? q : 0u) * 16 + 4u * cb + rb], reads);
          if (has_quals) {
            atomicAdd(&qsum[(size_t)cl * S.maxlen + p], (unsigned long long)(q * reads));
            atomicAdd(&qn[(size_t)cl * S.maxlen + p], reads);
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ns += __shfl_xor(ns, o, 64);
    if (lane == 0) nsubs[r] = (int32_t)ns;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x)
    if (s_hist[i]) atomicAdd((uint32_t *)&trans[i], s_hist[i]);
}

// Same tables for reads of up to 256 nt, walking the final pass's work list (ordered by partition): a wave keeps the
// per-position quality sums of the partition it is in in registers (4 positions per lane) and flushes them with one
// atomic per position when the partition changes - the per-unique atomics of the kernel above all land on the few
// hundred addresses of the largest partitions and serialise (1.07 ms at 100 k uniques).
__global__ __launch_bounds__(256) void k_final_tables_seg(SampleDev S, const uint16_t *__restrict__ view, int LV,
                                                          const int32_t *__restrict__ work, int nslots,
                                                          const int32_t *__restrict__ cluster_of,
                                                          const int32_t *__restrict__ centre_of_cluster,
                                                          const uint8_t *__restrict__ correct, int ncol, int has_quals,
                                                          int32_t *__restrict__ trans, unsigned long long *__restrict__ qsum,
                                                          uint32_t *__restrict__ qn, int32_t *__restrict__ nsubs) {
  extern __shared__ uint32_t s_hist[];   // [16*ncol]
  const int nb = 16 * ncol;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s_hist[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int span = (nslots + nwaves - 1) / nwaves;
  const int lo = gwave * span, hi = min(nslots, lo + span);
  int cur = -1, Lc = 0;
  const uint32_t *crow = nullptr;
  unsigned long long qs[4] = {0, 0, 0, 0};
  uint32_t qc[4] = {0, 0, 0, 0};
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int p = lane + 64 * k;
      if (qc[k]) {
        atomicAdd(&qsum[(size_t)cur * S.maxlen + p], qs[k]);
        atomicAdd(&qn[(size_t)cur * S.maxlen + p], qc[k]);
      }
      qs[k] = 0; qc[k] = 0;
    }
  };
  for (int slot = lo; slot < hi; slot++) {
    const int r = work[slot];
    if (r < 0) continue;
    const int cl = cluster_of[r];
    if (cl != cur) {
      flush();
      cur = cl;
      const int c = centre_of_cluster[cl];
      Lc = S.len[c];
      crow = S.seq2 + (size_t)c * S.W2;
    }
    const uint32_t reads = S.reads[r];
    const bool corr = correct[r] != 0;
    uint32_t ns = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int p = lane + 64 * k;
      if (p >= Lc) continue;
      const uint32_t v = view[(size_t)r * LV + p];
      if (v & 0x8000u) {
        const uint32_t rb = (v >> 8) & 3u, q = v & 255u, cb = base_at(crow, p);
        ns += (cb != rb);
        if (corr) {
          atomicAdd(&s_hist[(has_quals ? q : 0u) * 16 + 4u * cb + rb], reads);
          if (has_quals) { qs[k] += (unsigned long long)(q * reads); qc[k] += reads; }
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ns += __shfl_xor(ns, o, 64);
    if (lane == 0) nsubs[r] = (int32_t)ns;
  }
  flush();
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x)
    if (s_hist[i]) atomicAdd((uint32_t *)&trans[i], s_hist[i]);
}

void launch_final_tables(const SampleDev &S, const uint16_t *d_view, int LV, const int32_t *d_work, int nslots,
                         const int32_t *d_cluster_of, const int32_t *d_centre_of_cluster, const uint8_t *d_correct, int ncol,
                         int has_quals, int32_t *d_trans, unsigned long long *d_qsum, uint32_t *d_qn, int32_t *d_nsubs,
                         int nclust, hipStream_t st) {
  (void)nclust;
  int grid = std::min((S.N + 3) / 4, 2048);
  if (d_work && S.maxlen <= 256)
    hipLaunchKernelGGL(k_final_tables_seg, dim3(grid), dim3(256), (size_t)16 * ncol * 4, st, S, d_view, LV, d_work, nslots,
                       d_cluster_of, d_centre_of_cluster, d_correct, ncol, has_quals, d_trans, d_qsum, d_qn, d_nsubs);
  else
    hipLaunchKernelGGL(k_final_tables, dim3(grid), dim3(256), (size_t)16 * ncol * 4, st, S, d_view, LV, d_cluster_of,
                       d_centre_of_cluster, d_correct, ncol, has_quals, d_trans, d_qsum, d_qn, d_nsubs);
}

// ------------------------------------------------------------------------------------------------
// Bimera identification (the step after dada(): src/chimera.cpp).  The alignment of a query against a candidate
// parent is the denoising path's own banded ends-free NW with band = max_shift (chimera.cpp:26,122), done by k_nw with
// its move strings kept; this kernel turns one alignment into what C_is_bimera / C_table_bimera2 consume:
// get_lr (chimera.cpp:243-293: left / right coverage, and the one-off variants) and get_ham_endsfree (:211-239).
// One thread per work slot; query = the slot's chunk centre, parent = work[slot].  Moves were recorded from the END of
// the alignment backwards: 1 = both bases, 2 = gap in the query, 3 = gap in the parent.
__global__ __launch_bounds__(256) void k_bimera_lr(SampleDev S, const int32_t *__restrict__ chunk_centre, const int32_t *__restrict__ work,
                                                   int nwork, const uint8_t *__restrict__ moves, int stride,
                                                   const int32_t *__restrict__ nmoves, int allow_one_off, int max_shift,
                                                   int32_t *__restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nwork) return;
  const int k = work[idx];
  if (k < 0) return;
  const int q = chunk_centre[idx >> 6];
  const uint32_t *qrow = S.seq2 + (size_t)q * S.W2, *prow = S.seq2 + (size_t)k * S.W2;
  const uint8_t *mv = moves + (size_t)idx * stride;
  const int len = nmoves[idx];
  auto col = [&](int c) -> int { return mv[len - 1 - c]; };                       // move of forward column c
  // ---- left (forward) ----
  int pos = 0, i = 0, j = 0, left = 0, left_oo = 0;
  auto eq_f = [&](int c) -> bool { return col(c) == 1 && base_at(qrow, i) == base_at(prow, j); };
  auto adv_f = [&](int c) { const int p = col(c); if (p != 2) i++; if (p != 3) j++; };
  while (pos < len && col(pos) == 2) { adv_f(pos); pos++; }                       // scan in until the query starts
  while (pos < len && col(pos) == 3 && pos < max_shift) { adv_f(pos); pos++; left++; }   // ends-free until the parent starts
  while (pos < len && eq_f(pos)) { adv_f(pos); pos++; left++; }                   // covered until a mismatch
  if (allow_one_off) {
    left_oo = left;
    if (pos < len) adv_f(pos);
    pos++;
    if (pos < len && col(pos) != 2) left_oo++;
    while (pos < len && eq_f(pos)) { adv_f(pos); pos++; left_oo++; }
  }
  // ---- right (backward): before column c is consumed the bases are (i - 1, j - 1) ----
  int right = 0, right_oo = 0;
  i = S.len[q]; j = S.len[k];
  pos = len - 1;
  auto eq_b = [&](int c) -> bool { return col(c) == 1 && base_at(qrow, i - 1) == base_at(prow, j - 1); };
  auto adv_b = [&](int c) { const int p = col(c); if (p != 2) i--; if (p != 3) j--; };
  while (pos >= 0 && col(pos) == 2) { adv_b(pos); pos--; }
  // (the reference compares `pos > len - max_shift` in size_t: never true when the alignment is shorter than max_shift)
  while (pos >= 0 && col(pos) == 3 && len >= max_shift && pos > len - max_shift) { adv_b(pos); pos--; right++; }
  while (pos >= 0 && eq_b(pos)) { adv_b(pos); pos--; right++; }
  if (allow_one_off) {
    right_oo = right;
    if (pos >= 0) adv_b(pos);
    pos--;
    if (pos >= 0 && col(pos) != 2) right_oo++;
    while (pos >= 0 && eq_b(pos)) { adv_b(pos); pos--; right_oo++; }
  }
  // ---- get_ham_endsfree: mismatching columns between the two end-gap runs ----
  int is = 0, je = len - 1;
  {
    bool g1 = col(0) == 2, g2 = col(0) == 3;
    while (g1 || g2) { is++; g1 = g1 && col(is) == 2; g2 = g2 && col(is) == 3; }
    g1 = col(je) == 2; g2 = col(je) == 3;
    while (g1 || g2) { je--; g1 = g1 && col(je) == 2; g2 = g2 && col(je) == 3; }
  }
  int ham = 0;
  i = 0; j = 0;
  for (int c = 0; c < len; c++) {
    const int p = col(c);
    if (c >= is && c <= je && !(p == 1 && base_at(qrow, i) == base_at(prow, j))) ham++;
    if (p != 2) i++;
    if (p != 3) j++;
  }
  int32_t *o = out + (size_t)idx * 5;
  o[0] = left; o[1] = right; o[2] = left_oo; o[3] = right_oo; o[4] = ham;
}
void launch_bimera_lr(const SampleDev &S, const int32_t *d_chunk_centre, const int32_t *d_work, int nwork, const uint8_t *d_moves,
                      int stride, const int32_t *d_nmoves, int allow_one_off, int max_shift, int32_t *d_out, hipStream_t st) {
  if (nwork <= 0) return;
  hipLaunchKernelGGL(k_bimera_lr, dim3((nwork + 255) / 256), dim3(256), 0, st, S, d_chunk_centre, d_work, nwork, d_moves, stride,
                     d_nmoves, allow_one_off, max_shift, d_out);
}

#include "rounds2.inc.hip"   // (the persistent round tail, rounds3.inc.hip, is the translation unit tail.hip)

}  // namespace d2
