// hostsimd.cpp — the one host loop of the boundary that is worth a vector unit: the R-side quality matrix (8 bytes per base, 2 GB
// at 10^6 uniques x 250 nt) turned into the byte matrix the device keeps.  A translation unit of its own because it is plain
// C++ (function multiversioning does not exist in a HIP compilation).
#include <cstdint>

namespace d2 {

// (uint8) round(x) of one quality row, round() = half away from zero (raw_new, /root/reference/src/containers.cpp:34), for values
// in [0, 255.5): no branch in the loop, so the compiler vectorises it; the AVX2 clone is picked at load time on hosts that have
// it (every host of an MI355X does).  Returns false - the caller (driver.cpp, sample_create) redoes the row by the exact scalar
// rule - when a value lies outside that range or is NaN.
__attribute__((target_clones("avx2", "default")))
bool round_quality_row(const double *__restrict__ src, uint8_t *__restrict__ dst, int L, int *mx_out) {
  int ok = 1, mx = 0;
  for (int p = 0; p < L; p++) {
    const double x = src[p];
    ok &= (int)(x >= 0.0) & (int)(x < 255.5);
    const double xc = x >= 0.0 ? (x < 255.5 ? x : 0.0) : 0.0;   // (keeps the conversion defined for the values the caller will redo)
    const int t = (int)xc;
    const int v = t + (int)((xc - (double)t) >= 0.5);
    mx = v > mx ? v : mx;
    dst[p] = (uint8_t)v;
  }
  *mx_out = mx;
  return ok != 0;
}

}  // namespace d2
