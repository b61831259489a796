// nwpair.inc.hip — k_nw_ad2: the per-round aligner of engine v2 with TWO alignments per lane in packed int16 halves.
// Included by kernels.hip inside namespace d2.  Same recurrence, tie-breaks and outputs as k_nw_ad (nwalign_vectorized2 ==
// nwalign_endsfree + al2subs + compute_lambda_ts, /root/reference/src/nwalign_vectorized.cpp:71-318, pval.cpp:144-199).
//
// Preconditions (checked by nw_ad2_ok on the host; otherwise k_nw_ad runs): every read of the sample has the same length
// L <= 500, default scoring (5 / -4 / -8), band 1..18.  With equal lengths every alignment has the same band geometry,
// so the two alignments that share a lane sit at the same (i, j) on every step and the matrix-edge tests are per lane,
// not per half.
//
// Scores are kept as V = 2 H - 5 t (t = i + j, the anti-diagonal): a match adds 0, a mismatch -18, a gap -21, a free end
// move -5, the H = 0 border is -5 t.  Cells compared with each other always share t, so every max / tie is the one of H;
// |V| <= 18 L + 21 fits int16 with room for the out-of-band offset (-16 000, added instead of the gap penalty).
// Pointers: two bit planes per 16 steps and column, bit (16 h + step) of T1 = "left < diag", of T2 = "up < max(left, diag)"
// for the alignment in half h: up if !T2, else left if !T1, else diagonal (the reference's up > left > diag order).
//
// A wave holds 6 alignments: lane group al (21 lanes, as in k_nw_ad) carries the pair (2 al, 2 al + 1) through the DP; in
// the tail the group splits, lanes 0..9 serve the first alignment, lanes 10..20 the second.  Blocks have 2 waves so that
// three blocks (36 alignments) fit a CU's LDS, as with k_nw_ad.

typedef short pk_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
}
static __device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
}
static __device__ __forceinline__ uint32_t pk_maxs(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
}
// (spelled out: from the vector expressions the compiler builds min(x, 1) * -18 out of two compares, two selects and a
//  permute per step)
static __device__ __forceinline__ uint32_t pk_minu(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
static __device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

constexpr uint32_t P2_ONE = 0x00010001u;
constexpr uint32_t P2_MIS = 0xFFEEFFEEu;    // -18: 2 (mismatch - match)
constexpr uint32_t P2_GAP = 0xFFEBFFEBu;    // -21: 2 gap - match
constexpr uint32_t P2_FREE = 0xFFFBFFFBu;   //  -5: a free end move
constexpr uint32_t P2_OOBADD = 0xC180C180u; // -16 000: the "gap penalty" of an out-of-band cell
constexpr uint32_t P2_OOBV = 0xB1E0B1E0u;   // -20 000: what an out-of-band cell holds outside the steady state
constexpr int P2_GUARD = 32;                // guard entries either side of the staged bases (cell indices run about -21 .. L + 21)

struct Ad2Geom {
  int L, B, W, org, lbo, NCOL, nblk;
  int pair_words;      // pointer planes of one lane group: nblk * NCOL * 2 (later: the fp64 factors of its two alignments)
  int rpk_words;       // packed raw bases of one lane group
  int slot_bytes;      // runs + transition codes + qualities of one alignment
  int wave_bytes, block_bytes;
};
static __host__ __device__ inline Ad2Geom ad2_geom(int band, int L, int ncol) {
  Ad2Geom G;
  G.L = L; G.B = band; G.W = 2 * band + 1;
  G.org = 2 + (band & 1); G.lbo = band + G.org;
  G.NCOL = (G.W + 4) / 2;
  G.nblk = (2 * L + 1 + 15) / 16;
  G.pair_words = G.nblk * G.NCOL * 2;
  if (G.pair_words < 4 * L) G.pair_words = 4 * L;            // two alignments x L doubles
  G.pair_words = (G.pair_words + 3) & ~3;                     // the second alignment's factors start at its middle, 8-byte aligned
  G.rpk_words = (L + 2 * P2_GUARD + 1) & ~1;
  G.slot_bytes = AD_RCAP * 4 + 2 * ((L + 7) & ~7);
  G.wave_bytes = 3 * (G.pair_words + G.rpk_words) * 4 + 6 * G.slot_bytes;
  G.block_bytes = 16 * ncol * 8 + G.rpk_words * 4 + 2 * G.wave_bytes;
  return G;
}
bool nw_ad2_ok(const SampleDev &S, const AlignParams &ap) {
  if (S.minlen != S.maxlen || S.maxlen > 500 || S.maxlen < 16) return false;
  if (ap.band < 1 || ap.band > 18) return false;
  if (!(ap.match == 5 && ap.mismatch == -4 && ap.gap == -8)) return false;
  return ad2_geom(ap.band, S.maxlen, ap.ncol).block_bytes <= 54 * 1024;     // three blocks per CU
}

// one step of the packed sweep.  PAR: parity of the live cell; FULL: with the matrix-edge logic.
template <int PAR, bool FULL>
static __device__ __forceinline__ void ad2_step(uint32_t &d0, uint32_t &d1, int &i, int &j, uint32_t &cbw, uint32_t &rbw,
                                                uint32_t &acc1, uint32_t &acc2, uint32_t vnext, int f, bool kok, uint32_t gsel,
                                                int L, int t) {
  const uint32_t nb = PAR == 0 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d1, 0x138, 0xF, 0xF, true)     // lane-1's odd cell
                               : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d0, 0x130, 0xF, 0xF, true);    // lane+1's even cell
  const uint32_t own = PAR == 0 ? d0 : d1, other = PAR == 0 ? d1 : d0;
  const uint32_t m = pk_minu(cbw ^ rbw, P2_ONE);
  const uint32_t diag = pk_mad(m, P2_MIS, own);
  uint32_t gl = gsel, gu = gsel;
  if (FULL) {
    gl = i == L ? P2_FREE : P2_GAP;                           // free moves along the last row
    gu = j == L ? P2_FREE : P2_GAP;                           // ... and the last column
  }
  const uint32_t left = pk_adds(PAR == 0 ? nb : other, gl), up = pk_adds(PAR == 0 ? other : nb, gu);
  const uint32_t e1 = pk_maxs(left, diag), e = pk_maxs(up, e1);
  const uint32_t y1 = pk_subs(left, diag), y2 = pk_subs(up, e1);      // sign bits: left < diag, up < max(left, diag)
  uint32_t val = e, b1, b2;
  if (FULL) {
    const bool interior = kok && ((unsigned)(i - 1) < (unsigned)L) && ((unsigned)(j - 1) < (unsigned)L);
    const uint32_t bound = (uint32_t)((-5 * t) & 0xFFFF) * P2_ONE;     // H = 0 on the first row / column
    val = interior ? e : (kok ? bound : P2_OOBV);
    b1 = interior ? ((y1 >> 15) & P2_ONE) : 0u;                        // first row: left (T1 = 0, T2 = 1); first column: up (T2 = 0)
    b2 = interior ? ((y2 >> 15) & P2_ONE) : (i <= 0 ? P2_ONE : 0u);
    acc1 |= b1 << f;
    acc2 |= b2 << f;
  } else {
    acc1 |= (y1 >> (15 - f)) & (P2_ONE << f);
    acc2 |= (y2 >> (15 - f)) & (P2_ONE << f);
  }
  if (PAR == 0) { d0 = val; rbw = vnext; j++; } else { d1 = val; cbw = vnext; i++; }
}

__global__ __launch_bounds__(128) void k_nw_ad2(NwArgs a, const int32_t *__restrict__ gl_work, const int32_t *__restrict__ gl_nwork_dev,
                                                Ad2Geom G) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn2[];
  double *s_err = s_dyn2;
  const int nerr = 16 * a.ap.ncol;
  if (a.stop_dev && *a.stop_dev != 0) return;
  const int n_nw = a.nwork_dev ? *a.nwork_dev : a.nwork_host;
  const int n_gl = gl_work ? *gl_nwork_dev : 0;
  const int nwork = n_nw + n_gl;
  if ((int)blockIdx.x * 12 >= nwork) return;
  const int c = a.centre_dev ? *a.centre_dev : a.centre;
  if (c < 0) return;
  const SampleDev &S = a.S;
  const int L = G.L, NCOL = G.NCOL;
  uint32_t *cpk = (uint32_t *)(s_dyn2 + nerr) + P2_GUARD;                  // [L] centre bases, both halves
  for (int k = threadIdx.x; k < nerr; k += blockDim.x) s_err[k] = a.err[k];
  for (int p = threadIdx.x; p < L; p += blockDim.x) cpk[p] = base_at(S.seq2 + (size_t)c * S.W2, p) * P2_ONE;
  const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool ghost = lane >= 63;
  const int al = ghost ? 2 : lane / 21, g = ghost ? 21 : lane % 21;
  // tail geometry: the lane group splits in two
  const int h = g >= 10 ? 1 : 0;
  const int sg = ghost ? 99 : (h ? g - 10 : g), SG = h ? 11 : 10, sl0 = al * 21 + (h ? 10 : 0);
  const int slot = 2 * al + h;
  uint8_t *wbase = (uint8_t *)(cpk - P2_GUARD + G.rpk_words) + (size_t)wib * G.wave_bytes;
  uint32_t *pair = (uint32_t *)wbase + (size_t)al * (G.pair_words + G.rpk_words);   // pointer planes [nblk][NCOL][2]
  uint32_t *rpk = pair + G.pair_words + P2_GUARD;                                    // packed raw bases of the pair
  uint8_t *sbase = wbase + (size_t)3 * (G.pair_words + G.rpk_words) * 4 + (size_t)slot * G.slot_bytes;
  uint32_t *runs = (uint32_t *)sbase;
  uint8_t *tcode = sbase + AD_RCAP * 4;
  uint8_t *qlds = tcode + ((L + 7) & ~7);
  double *fac = (double *)pair + (size_t)h * (G.pair_words / 4);                     // after the traceback: L doubles per alignment
  __syncthreads();
  const int gwave = blockIdx.x * 2 + wib, nwaves = gridDim.x * 2;
  const int B = G.B, W = G.W, org = G.org, lbo = G.lbo;
  for (int chunk = gwave; chunk * 6 < nwork; chunk += nwaves) {
    const int idx = chunk * 6 + slot;
    int r = idx < n_nw ? a.work[idx] : (idx < nwork ? gl_work[idx - n_nw] : -1);
    const bool gapless = idx >= n_nw;
    const bool active = r >= 0 && !ghost;
    if (r < 0) r = c;
    // stage the raw's bases into its half of the pair's words, and its qualities
    if (!ghost) {
      uint16_t *rh = (uint16_t *)rpk + h;
      for (int p = sg; p < L; p += SG) rh[2 * p] = (uint16_t)base_at(S.seq2 + (size_t)r * S.W2, p);
      const uint32_t *qsrc = (const uint32_t *)(S.qual + (size_t)r * S.LQ);
      for (int w = sg; w * 4 < L; w += SG) ((uint32_t *)qlds)[w] = qsrc[w];
    }
    int T = (gapless || !active) ? -1 : 2 * L;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) T = max(T, __shfl_xor(T, o, 64));
    const int Tmax = __builtin_amdgcn_readfirstlane(T);
    const int dbg = a.moves_stride;                                      // profiling knob (DADA2HIP_AD_DEBUG): skip phases, results void
    if (Tmax >= 0 && !(dbg & 1)) {
      uint32_t d0 = P2_OOBV, d1 = P2_OOBV, acc1 = 0, acc2 = 0;
      int i = (lbo >> 1) - g, j = -i;
      uint32_t cbw = cpk[i - 1], rbw = rpk[j - 1];
      const bool kok0 = !ghost && 2 * g >= org && 2 * g < W + org, kok1 = !ghost && 2 * g + 1 >= org && 2 * g + 1 < W + org;
      const bool colok = !ghost && g < NCOL;
      const uint32_t gs0 = kok0 ? P2_GAP : P2_OOBADD, gs1 = kok1 ? P2_GAP : P2_OOBADD;
      // steady state [tA, tB): every in-band cell is interior and off the last row / column
      const int tA = B + 2, tB = 2 * L - B;
      const int t0 = (tA + 15) & ~15;
#define AD2_FLUSH(TT) { if (colok) { uint2 w2 = make_uint2(acc1, acc2); *(uint2 *)(pair + ((size_t)((TT) >> 4) * NCOL + g) * 2) = w2; } acc1 = 0; acc2 = 0; }
#define AD2_FULL(TT)                                                                                                      \
  {                                                                                                                       \
    if (((TT) & 1) == 0) ad2_step<0, true>(d0, d1, i, j, cbw, rbw, acc1, acc2, rpk[j], (TT) & 15, kok0, gs0, L, (TT));     \
    else ad2_step<1, true>(d0, d1, i, j, cbw, rbw, acc1, acc2, cpk[i], (TT) & 15, kok1, gs1, L, (TT));                    \
    if (((TT) & 15) == 15) AD2_FLUSH(TT)                                                                                  \
  }
      int t = 0;
      for (; t <= Tmax && t < t0; t++) AD2_FULL(t)
      for (; t + 16 <= tB && t + 15 <= Tmax; t += 16) {
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          ad2_step<0, false>(d0, d1, i, j, cbw, rbw, acc1, acc2, rpk[j], u, kok0, gs0, L, 0);
          ad2_step<1, false>(d0, d1, i, j, cbw, rbw, acc1, acc2, cpk[i], u + 1, kok1, gs1, L, 0);
        }
        AD2_FLUSH(t)
      }
      for (; t <= Tmax; t++) AD2_FULL(t)
      if (((t - 1) & 15) != 15) AD2_FLUSH(t - 1)
#undef AD2_FULL
#undef AD2_FLUSH
    }
    // ---- traceback: the first lane of each half-group walks, the half-group measures every diagonal stretch at once ----
    int ti = L, tj = L;
    bool done = !active || (dbg & 2);
    uint32_t hs = 0;
    int guard = 2 * L + 2;
    const bool lead = sg == 0;
    const int hsh = 16 * h;
    while (true) {
      int nruns = 0;
      uint32_t last = 0;
      auto push = [&](int lo, int n, int dl) {
        if (last) {
          const int llo = last & 4095, ln = (last >> 12) & 4095, ldl = (int)(last >> 24);
          if (ldl == dl && lo + n == llo) { last = (uint32_t)lo | ((uint32_t)(ln + n) << 12) | ((uint32_t)dl << 24); return; }
          runs[nruns++] = last;
        }
        last = (uint32_t)lo | ((uint32_t)n << 12) | ((uint32_t)dl << 24);
      };
      if (lead && !done && gapless) {
        runs[nruns++] = 0u | ((uint32_t)L << 12) | (128u << 24);          // nwalign_gapless: position-wise pairing (equal lengths)
        done = true;
      }
      for (;;) {
        const bool act = lead && !done && (ti > 0 || tj > 0) && nruns < AD_RCAP - 2 && guard > 0;
        if (!__any(act)) break;
        const int gact = __shfl((int)act, sl0, 64);
        const int tt = __shfl(ti + tj, sl0, 64), col = __shfl((tj - ti + lbo) >> 1, sl0, 64);
        const int f0 = tt & 15, widx = (tt >> 4) - sg;
        uint32_t nd = 0, t2w = 0;                                          // (before the matrix: never reached, the axis cells stop the run)
        if (gact && !ghost && widx >= 0) {
          const uint2 w2 = *(const uint2 *)(pair + ((size_t)widx * NCOL + col) * 2);
          nd = (~(w2.x & w2.y) >> hsh) & 0xFFFFu;
          t2w = (w2.y >> hsh) & 0xFFFFu;
        }
        const int ftop = sg == 0 ? f0 : 14 + (f0 & 1);
        uint32_t nz = nd & ((f0 & 1) ? 0xAAAAu : 0x5555u) & ((2u << ftop) - 1u);
        const bool st = nz != 0;
        const int fb = st ? 31 - __clz(nz) : 0;
        const int ng = st ? (ftop - fb) >> 1 : (ftop >> 1) + 1;           // diagonal moves inside this word
        const uint32_t pleft = (t2w >> fb) & 1u;                           // the stop is a left move (else up)
        const unsigned long long bal = (__ballot(st && !ghost) >> sl0) & ((1ull << SG) - 1ull);
        const int qs = bal ? __builtin_ctzll(bal) : SG;
        const int srcl = sl0 + (qs < SG ? qs : 0);
        const int n0 = __shfl(ng, sl0, 64), nq = __shfl(ng, srcl, 64);
        const uint32_t pq = (uint32_t)__shfl((int)pleft, srcl, 64);
        if (act) {
          guard--;
          int n = qs == 0 ? n0 : n0 + 8 * (qs - 1) + (qs < SG ? nq : 0);
          const int room = ti < tj ? ti : tj;
          const bool clamped = n > room;
          if (clamped) { n = room; atomicOr(S.nw_flag, 1); }
          if (n > 0) { push(tj - n, n, ti - tj + 128); ti -= n; tj -= n; }
          if (qs < SG && !clamped && (ti > 0 || tj > 0)) {
            if (pq) { tj--; push(tj, 1, 255); }
            else ti--;
          }
        }
      }
      if (lead && !gapless && active) {
        if (last) runs[nruns++] = last;
        if (!(ti > 0 || tj > 0)) done = true;
        else if (guard <= 0) { done = true; atomicOr(S.nw_flag, 1); }
      }
      nruns = __shfl(nruns, sl0, 64);
      done = __shfl((int)done, sl0, 64) != 0;
      int nrmax = nruns;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) nrmax = max(nrmax, __shfl_xor(nrmax, o, 64));
      for (int ri = 0; ri < nrmax; ri++) {
        if (ri < nruns && !ghost) {
          const uint32_t dsc = runs[ri];
          const int lo = dsc & 4095, n = (dsc >> 12) & 4095, dl = (int)(dsc >> 24);
          for (int pj = lo + sg; pj < lo + n; pj += SG) {
            const uint32_t rb = (rpk[pj] >> hsh) & 0xFFu;
            uint32_t tc = 5u * rb;
            if (dl != 255) {
              const uint32_t cb = cpk[pj + dl - 128] & 0xFFu;
              tc = 4u * cb + rb;
              hs += (cb != rb);
            }
            tcode[pj] = (uint8_t)tc;
          }
        }
      }
      if (__all(done)) break;
    }
    // ---- factors (over the pointer planes, no longer needed by either alignment of the pair), hamming, product ----
    if (!ghost && !(dbg & 4))
      for (int pj = sg; pj < L; pj += SG) {
        const uint32_t q = a.ap.use_quals ? qlds[pj] : 0u;
        fac[pj] = s_err[(uint32_t)tcode[pj] * a.ap.ncol + q];
      }
    if (lead) runs[0] = 0;
    if (!ghost && hs) atomicAdd(&runs[0], hs);
    hs = runs[0];
    if (lead && active && !(dbg & 8)) {
      double l = 1.0;
      int pj = 0;
      if (L >= 8) {
        double f0 = fac[0], f1 = fac[1], f2 = fac[2], f3 = fac[3], f4 = fac[4], f5 = fac[5], f6 = fac[6], f7 = fac[7];
        for (pj = 8; pj + 8 <= L; pj += 8) {
          const double n0 = fac[pj], n1 = fac[pj + 1], n2 = fac[pj + 2], n3 = fac[pj + 3];
          const double n4 = fac[pj + 4], n5 = fac[pj + 5], n6 = fac[pj + 6], n7 = fac[pj + 7];
          l = l * f0; l = l * f1; l = l * f2; l = l * f3; l = l * f4; l = l * f5; l = l * f6; l = l * f7;
          f0 = n0; f1 = n1; f2 = n2; f3 = n3; f4 = n4; f5 = n5; f6 = n6; f7 = n7;
        }
        l = l * f0; l = l * f1; l = l * f2; l = l * f3; l = l * f4; l = l * f5; l = l * f6; l = l * f7;
      }
      for (; pj < L; pj++) l = l * fac[pj];
      a.lam[r] = l;
      a.ham[r] = hs;
    }
  }
}

void launch_nw_ad2(const SampleDev &S, const int32_t *d_work, const int32_t *d_nwork, const int32_t *d_gl_work,
                   const int32_t *d_gl_nwork, const AlignParams &ap, const double *d_err, double *d_lambda, uint32_t *d_ham,
                   const int32_t *d_centre_dev, hipStream_t st, const int32_t *d_stop_dev, int centre_host, int nwork_host) {
  NwArgs a;
  memset(&a, 0, sizeof a);
  a.S = S; a.centre = centre_host; a.work = d_work; a.nwork_dev = d_nwork; a.nwork_host = nwork_host; a.ap = ap; a.err = d_err;
  a.lam = d_lambda; a.ham = d_ham; a.centre_dev = d_centre_dev; a.stop_dev = d_stop_dev;
  { const char *e = getenv("DADA2HIP_AD_DEBUG"); a.moves_stride = e ? atoi(e) : 0; }
  const Ad2Geom G = ad2_geom(ap.band, S.maxlen, ap.ncol);
  const size_t lds = (size_t)G.block_bytes;
  static size_t attr_set[64] = {0};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (lds > attr_set[dev_ & 63]) {
    (void)hipFuncSetAttribute((const void *)k_nw_ad2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev_ & 63] = lds;
  }
  const int maxwork = d_nwork ? S.N : std::max(nwork_host, 1);
  const int grid = std::min((maxwork + 11) / 12, 256 * 3 * 4);
  hipLaunchKernelGGL(k_nw_ad2, dim3(grid), dim3(128), lds, st, a, d_gl_work, d_gl_nwork, G);
}
