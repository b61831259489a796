// tools/micro/grid_barrier.hip - what ONE grid barrier of the persistent tail costs on this device, by itself.
// The barrier of rounds3.inc.hip::grid_sync (flat: every block arrives on one counter) against a two-level one (blocks arrive on
// one of eight counters - their XCD's, blocks being dealt to XCDs round-robin - and the last of each group arrives on the
// global one), for the grids the tail runs on; with and without 1 MB of stores per block before the barrier (the write-back
// the release has to wait for).  Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ; run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../dada2_amd/csrc/gcn.h"
using namespace d2;

struct Sync { uint32_t arrive, pad0[31], gen, pad1[31], grp[8][32], ggen[256][32]; };

template <bool HIER, int FAN>   // FAN: 0 one generation word, 8 one per group of blocks (blockIdx & 7), 256 one per block
__global__ __launch_bounds__(512) void k_bar(Sync *ps, int iters, uint32_t *scratch, int store_words, unsigned long long *out) {
  const int G = gridDim.x;
  uint32_t epoch = 0;
  __shared__ int s_dummy, s_last;
  const unsigned long long t0 = gcn_wall_clock();
  for (int it = 0; it < iters; it++) {
    for (int w = threadIdx.x; w < store_words; w += blockDim.x) scratch[(size_t)blockIdx.x * store_words + w] = (uint32_t)(it + w);
    gcn_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
      gcn_release_agent();
      bool last;
      if (HIER) {
        const int x = blockIdx.x & 7, nx = (G - x + 7) / 8;           // blocks of this group
        const uint32_t t = gcn_add_agent(&ps->grp[x][0], 1u);
        last = false;
        if (t == (epoch + 1u) * (uint32_t)nx - 1u) {
          const int ngrp = G < 8 ? G : 8;
          const uint32_t t2 = gcn_add_agent(&ps->arrive, 1u);
          last = t2 == (epoch + 1u) * (uint32_t)ngrp - 1u;
        }
      } else {
        const uint32_t t = gcn_add_agent(&ps->arrive, 1u);
        last = t == (epoch + 1u) * (uint32_t)G - 1u;
      }
      uint32_t *mine = FAN == 0 ? &ps->gen : (FAN == 8 ? &ps->ggen[blockIdx.x & 7][0] : &ps->ggen[blockIdx.x][0]);
      if (!last) {
        const unsigned long long w0 = gcn_wall_clock();   // (bounded: a grid that is not co-resident ends, wrong but ended)
        while ((int32_t)(gcn_load_agent(mine) - (epoch + 1u)) < 0 && gcn_wall_clock() - w0 < 2 * GCN_WALL_HZ) gcn_poll_pause();
      }
      gcn_acquire_agent();
      s_last = last ? 1 : 0;
      if (last) { s_dummy = it; gcn_release_agent(); if (FAN == 0) gcn_store_agent(&ps->gen, epoch + 1u); }
    }
    __syncthreads();
    if (FAN != 0 && s_last) {   // the releasing block fans the generation out, one lane per word
      const int n = FAN == 8 ? 8 : G;
      if ((int)threadIdx.x < n) gcn_store_agent(&ps->ggen[threadIdx.x][0], epoch + 1u);
    }
    epoch++;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *out = gcn_wall_clock() - t0;
}

// barrier-xcd of the guide (MI355X_MICROARCH.md, price list): blocks grouped by the XCC they RUN on (hardware register), a
// counter and a generation word per XCC; only the last arriver of an XCC writes its L2 back (the others' stores are in that
// L2 already: every wave drained them before its block arrived) and arrives on the top counter.
struct XSync { uint32_t top, p0[31], flat, p1[31], fgen, p2[31], xarr[8][32], xgen[8][32], xcount[8][32]; };
static __device__ __forceinline__ int xcc_id() { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x)); return x & 7; }
__global__ __launch_bounds__(512) void k_bar_xcd(XSync *ps, int iters, uint32_t *scratch, int store_words, unsigned long long *out) {
  const int G = gridDim.x;
  __shared__ int s_nxb, s_ngrp, s_x, s_dummy;
  if (threadIdx.x == 0) {
    const int x = xcc_id();
    s_x = x;
    gcn_add_agent(&ps->xcount[x][0], 1u);
    // a flat barrier once: everybody has counted itself
    gcn_release_agent();
    const uint32_t t = gcn_add_agent(&ps->flat, 1u);
    if (t == (uint32_t)G - 1u) gcn_store_agent(&ps->fgen, 1u);
    const unsigned long long w0 = gcn_wall_clock();
    while (gcn_load_agent(&ps->fgen) == 0u && gcn_wall_clock() - w0 < 2 * GCN_WALL_HZ) gcn_poll_pause();
    gcn_acquire_agent();
    int ng = 0;
    for (int g = 0; g < 8; g++) ng += gcn_load_agent(&ps->xcount[g][0]) != 0u;
    s_ngrp = ng; s_nxb = (int)gcn_load_agent(&ps->xcount[x][0]);
  }
  __syncthreads();
  const int x = s_x, nxb = s_nxb, ngrp = s_ngrp;
  uint32_t epoch = 0;
  const unsigned long long t0 = gcn_wall_clock();
  for (int it = 0; it < iters; it++) {
    for (int w = threadIdx.x; w < store_words; w += blockDim.x) scratch[(size_t)blockIdx.x * store_words + w] = (uint32_t)(it + w);
    gcn_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
      bool last = false;
      const uint32_t t = gcn_add_agent(&ps->xarr[x][0], 1u);
      if (t == (epoch + 1u) * (uint32_t)nxb - 1u) {
        gcn_release_agent();
        const uint32_t t2 = gcn_add_agent(&ps->top, 1u);
        last = t2 == (epoch + 1u) * (uint32_t)ngrp - 1u;
      }
      if (!last) {
        const unsigned long long w0 = gcn_wall_clock();
        while ((int32_t)(gcn_load_agent(&ps->xgen[x][0]) - (epoch + 1u)) < 0 && gcn_wall_clock() - w0 < 2 * GCN_WALL_HZ) gcn_poll_pause();
      }
      gcn_acquire_agent();
      if (last) {
        s_dummy = it;
        gcn_release_agent();
        for (int g = 0; g < 8; g++) gcn_store_agent(&ps->xgen[g][0], epoch + 1u);
      }
    }
    __syncthreads();
    epoch++;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = gcn_wall_clock() - t0; out[1] = (unsigned long long)ngrp * 1000 + nxb; }
}

int main() {
  Sync *ps; uint32_t *scr; unsigned long long *out;
  (void)hipMalloc(&ps, sizeof(Sync)); (void)hipMalloc(&scr, (size_t)256 * 262144 * 4); (void)hipMalloc(&out, 16); XSync *xs; (void)hipMalloc(&xs, sizeof(XSync));
  const int iters = 2000;
  for (int words : {0, 4096, 65536})
    for (int G : {8, 25, 64, 96, 128, 245}) {
      double us[6];
      for (int h = 0; h < 6; h++) {
        for (int rep = 0; rep < 2; rep++) {
          (void)hipMemset(ps, 0, sizeof(Sync));
          switch (h) {
            case 0: hipLaunchKernelGGL((k_bar<false, 0>), dim3(G), dim3(512), 0, 0, ps, iters, scr, words, out); break;
            case 1: hipLaunchKernelGGL((k_bar<true, 0>), dim3(G), dim3(512), 0, 0, ps, iters, scr, words, out); break;
            case 2: hipLaunchKernelGGL((k_bar<false, 8>), dim3(G), dim3(512), 0, 0, ps, iters, scr, words, out); break;
            case 3: hipLaunchKernelGGL((k_bar<true, 8>), dim3(G), dim3(512), 0, 0, ps, iters, scr, words, out); break;
            case 4: hipLaunchKernelGGL((k_bar<false, 256>), dim3(G), dim3(512), 0, 0, ps, iters, scr, words, out); break;
            default: hipLaunchKernelGGL((k_bar<true, 256>), dim3(G), dim3(512), 0, 0, ps, iters, scr, words, out); break;
          }
          (void)hipDeviceSynchronize();
        }
        unsigned long long t = 0; (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
        us[h] = (double)t / 100.0 / iters;
      }
      double xus = 0; unsigned long long grp = 0;
      for (int rep = 0; rep < 2; rep++) {
        (void)hipMemset(xs, 0, sizeof(XSync));
        hipLaunchKernelGGL(k_bar_xcd, dim3(G), dim3(512), 0, 0, xs, iters, scr, words, out);
        (void)hipDeviceSynchronize();
        unsigned long long t2[2] = {0, 0}; (void)hipMemcpy(t2, out, 16, hipMemcpyDeviceToHost);
        xus = (double)t2[0] / 100.0 / iters; grp = t2[1];
      }
      printf("{\"xcd_us\": %.2f, \"xcd_groups_x1000_plus_blocks_in_block0s\": %llu, ", xus, grp);
      printf("\"blocks\": %d, \"store_bytes_per_block\": %d, \"flat_us\": %.2f, \"two_level_us\": %.2f, \"flat_gen8_us\": %.2f, \"two_level_gen8_us\": %.2f, \"flat_genblock_us\": %.2f, \"two_level_genblock_us\": %.2f}\n", G, words * 4, us[0], us[1], us[2], us[3], us[4], us[5]);
    }
  return 0;
}
