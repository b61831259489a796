// rounds_common.h — the few device-side definitions the round kernels of kernels.hip and the persistent round tail (tail.hip, a
// translation unit of its own) share.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.h"
#include "ppois.h"

namespace d2 {

// partitions whose reads deltas / reads a block of the shuffle pass keeps in LDS (kernels.hip: k_shuffle, rounds2.inc.hip: shuffle_body)
constexpr int DELTA_TAB = 1024;

// get_pA (pval.cpp:67-89)
static __device__ __forceinline__ double dev_get_pA(uint32_t reads, bool prior, bool detect_singletons, double lambda,
                                                    uint32_t hamming, uint32_t bi_reads) {
  if (reads == 1 && !prior && !detect_singletons) return 1.;
  if (hamming == 0) return 1.;
  if (lambda == 0) return 0.;
  return pp::calc_pA((int)reads, lambda * bi_reads, prior || detect_singletons);
}

// b_bud (cluster.cpp:274-350), arg-min part.  Key order: p ascending, then reads descending; exact
// ties are resolved by the host in (partition, slot) scan order.  track 0 = all candidates, 1 = priors.
struct BudKey { double p; uint32_t reads; };
static __device__ __forceinline__ bool bud_better(double p, uint32_t reads, const BudKey &b) {
  return p < b.p || (p == b.p && reads > b.reads);
}

}  // namespace d2
