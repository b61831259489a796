// microbench.hip — pins the two peaks bench.py's roofline fractions are quoted against, on the box itself
// ("confirm peaks with a micro-benchmark on the box"), and provides known-byte-count store patterns for
// calibrating rocprofv3's WRITE_SIZE counter (MI355X_MICROARCH.md §HBM: "WRITE_SIZE uncalibrated").
//
//   microbench peaks            -> one JSON object: int32 VALU lane-ops/s (add / max / cndmask / the NW step mix),
//                                  HBM read GB/s (uint4 stream over 4 GiB), HBM copy GB/s
//   microbench occ              -> cycles per instruction of a dependent DP-like chain at 1..8 waves per SIMD
//   microbench launch           -> cost of an (empty / tiny) dependent launch by grid size, eager and from a hipGraph
//   microbench store_bytes N    -> the k_screen class-byte pattern: one lane in 16 stores ONE byte, N bytes in all
//   microbench store_wide N     -> fully coalesced uint4 stores, N bytes in all
//   microbench read_wide N      -> fully coalesced uint4 loads, N bytes in all   (FETCH_SIZE check: expect 1/2)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip   (done by __graft_entry__.build()).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                    \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)

// ---- integer VALU: 16 independent accumulators per lane, ITER x 16 instructions of one kind, written in asm so the
// instruction count is exactly what is stated.  8 waves per SIMD resident (256-thread blocks, 8 blocks per CU).
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(int *out, int a, int b, int iters) {
  int x[16];
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = threadIdx.x + k * a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      // (second source in an SGPR: two VGPR sources of one instruction can collide on a register bank)
      if (KIND == 0) asm volatile("v_add_u32 %0, %2, %1" : "=v"(x[k]) : "v"(x[k]), "s"(b));
      if (KIND == 1) asm volatile("v_max_i32 %0, %2, %1" : "=v"(x[k]) : "v"(x[k]), "s"(b));
      if (KIND == 2) asm volatile("v_cndmask_b32 %0, 7, %1, vcc" : "=v"(x[k]) : "v"(x[k]));           // (vcc + an SGPR source would exceed the constant bus)
      if (KIND == 4) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(b));            // both sources VGPRs
      if (KIND == 5) asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(b));         // 2 x int16 per lane
      if (KIND == 6) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(b));
      if (KIND == 7) asm volatile("v_add3_u32 %0, %1, %2, %2" : "=v"(x[k]) : "v"(x[k]), "s"(b));
      if (KIND == 3) {   // the shape of one NW cell: 3 adds, 2 max, 2 compare+select pairs -> 9 VALU, counted as 9
        int d, u, l, e1, e;
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(x[k]), "v"(a));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(u) : "v"(x[(k + 1) & 15]), "v"(b));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(l) : "v"(x[(k + 15) & 15]), "v"(b));
        asm volatile("v_max_i32 %0, %1, %2" : "=v"(e1) : "v"(l), "v"(d));
        asm volatile("v_max_i32 %0, %1, %2" : "=v"(e) : "v"(u), "v"(e1));
        asm volatile("v_cmp_ge_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(d) : "v"(l), "v"(d), "v"(a), "v"(b) : "vcc");
        asm volatile("v_cmp_ge_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(u) : "v"(u), "v"(e1), "v"(d), "v"(b) : "vcc");
        x[k] = e + u;
      }
    }
  }
  int s = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += x[k];
  if (s == 0x7fffffff) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- per-instruction issue rates: 16 independent chains of ONE instruction form each (mode `rates`) -----------------
// The DP step of the NW kernels is a dependent chain of adds, max, compares and selects; which of them the SIMD issues
// at full rate decides how the step should be written.
#define RATE_KERNEL(NAME, ASM, ...)                                                                    \
  __global__ __launch_bounds__(256) void NAME(int *out, int a, int b, int iters) {                     \
    int x[16];                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 16; k++) x[k] = threadIdx.x + k * a;                         \
    int y = threadIdx.x ^ b;                                                                           \
    for (int it = 0; it < iters; it++) {                                                               \
      _Pragma("unroll") for (int k = 0; k < 16; k++) asm volatile(ASM : "+v"(x[k]) : "v"(y), "s"(b) __VA_ARGS__); \
    }                                                                                                  \
    int s = 0;                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 16; k++) s += x[k];                                          \
    if (s == 0x7fffffff) out[blockIdx.x * 256 + threadIdx.x] = s;                                      \
  }
RATE_KERNEL(r_add_vv, "v_add_u32 %0, %0, %1")
RATE_KERNEL(r_sub_vv, "v_sub_u32 %0, %0, %1")
RATE_KERNEL(r_max_vv, "v_max_i32 %0, %0, %1")
RATE_KERNEL(r_min_vv, "v_min_i32 %0, %0, %1")
RATE_KERNEL(r_minu_vv, "v_min_u32 %0, %0, %1")
RATE_KERNEL(r_and_vv, "v_and_b32 %0, %0, %1")
RATE_KERNEL(r_xor_vv, "v_xor_b32 %0, %0, %1")
RATE_KERNEL(r_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
RATE_KERNEL(r_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
RATE_KERNEL(r_add3, "v_add3_u32 %0, %0, %1, %1")
RATE_KERNEL(r_max3, "v_max3_i32 %0, %0, %1, %1")
RATE_KERNEL(r_mad_i24, "v_mad_i32_i24 %0, %0, 3, %1")
RATE_KERNEL(r_bfe_i32, "v_bfe_i32 %0, %0, 2, 6")
RATE_KERNEL(r_mov_dpp_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
RATE_KERNEL(r_mov_dpp_rowshr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
RATE_KERNEL(r_cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
RATE_KERNEL(r_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", : "s20", "s21")
RATE_KERNEL(r_cmp_vcc, "v_cmp_ge_i32 vcc, %0, %1\n\tv_add_u32 %0, %0, %1", : "vcc")
RATE_KERNEL(r_cmp_sgpr, "v_cmp_ge_i32_e64 s[20:21], %0, %1\n\tv_add_u32 %0, %0, %1", : "s20", "s21")
RATE_KERNEL(r_cmp_cndmask, "v_cmp_ge_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc", : "vcc")
RATE_KERNEL(r_addc, "v_addc_co_u32 %0, vcc, %0, %0, vcc", : "vcc")
RATE_KERNEL(r_pk_add_i16, "v_pk_add_i16 %0, %0, %1")
RATE_KERNEL(r_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
RATE_KERNEL(r_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
RATE_KERNEL(r_cmp_eq_u16_sdwa, "v_cmp_eq_u16_sdwa vcc, %0, %1 src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_add_u32 %0, %0, %1", : "vcc")
RATE_KERNEL(r_sad_u8, "v_sad_u8 %0, %0, %1, %0")
RATE_KERNEL(r_perm, "v_perm_b32 %0, %0, %1, %1")

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
  uint32_t acc = 0;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n; i += stride) { const uint4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ p, uint4 *__restrict__ q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
__global__ __launch_bounds__(256) void k_store_wide(uint4 *__restrict__ q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    q[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// k_screen's class bytes: 16 lanes per unique, lane 0 of each group stores one byte; consecutive groups -> consecutive bytes
__global__ __launch_bounds__(256) void k_store_bytes(uint8_t *__restrict__ q, size_t n) {
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (size_t base = (size_t)blockIdx.x * 16; base < n; base += (size_t)gridDim.x * 16) {
    const size_t r = base + grp;
    if (r < n && sub == 0) q[r] = (uint8_t)(r & 3);
  }
}


// ---- occupancy curve of a dependent NW-like step (mode `occ`): ONE dependent chain per lane (as a DP lane has), w waves
// per SIMD for w = 1..8: what a SIMD issues per cycle as a function of the waves it holds.
__global__ __launch_bounds__(256) void k_chain(int *out, int a, int b, int iters) {
  int x = threadIdx.x, y = threadIdx.x ^ b, z = a;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
      asm volatile("v_cmp_ge_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(z) : "v"(x), "v"(y) : "vcc");
    }
  }
  if (x + z == 0x7fffffff) out[blockIdx.x * 256 + threadIdx.x] = x;
}
// the same with FOUR independent chains per lane (does instruction-level parallelism inside a wave buy issue slots?)
__global__ __launch_bounds__(256) void k_chain4(int *out, int a, int b, int iters) {
  int x[4], z[4];
  const int y = threadIdx.x ^ b;
#pragma unroll
  for (int q = 0; q < 4; q++) { x[q] = threadIdx.x + q; z[q] = a + q; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
      for (int q = 0; q < 4; q++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[q]) : "v"(y));
#pragma unroll
      for (int q = 0; q < 4; q++) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(y), "v"(z[q]));
#pragma unroll
      for (int q = 0; q < 4; q++) asm volatile("v_cmp_ge_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(z[q]) : "v"(x[q]), "v"(y) : "vcc");
    }
  }
  int s = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) s += x[q] + z[q];
  if (s == 0x7fffffff) out[blockIdx.x * 256 + threadIdx.x] = s;
}
// ---- launch costs (mode `launch`): a kernel whose blocks leave at their first instruction
__global__ __launch_bounds__(256) void k_noop(const int *flag) { if (*flag) return; }
__global__ __launch_bounds__(256) void k_touch(int *buf, const int *flag, int n) {   // one coalesced 4-byte read-modify-write per thread
  if (*flag) return;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) buf[i] += 1;
}

static float time_ms(void (*launch)(void *), void *ctx, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(ctx);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(a, 0));
    launch(ctx);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return best;
}

struct ValuCtx { int *out; int kind, iters, grid; };
static void launch_valu(void *c) {
  ValuCtx *v = (ValuCtx *)c;
  switch (v->kind) {
    case 0: hipLaunchKernelGGL(k_valu<0>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters); break;
    case 1: hipLaunchKernelGGL(k_valu<1>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters); break;
    case 2: hipLaunchKernelGGL(k_valu<2>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters); break;
    case 3: hipLaunchKernelGGL(k_valu<3>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters); break;
    case 4: hipLaunchKernelGGL(k_valu<4>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters); break;
    case 5: hipLaunchKernelGGL(k_valu<5>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters); break;
    case 6: hipLaunchKernelGGL(k_valu<6>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters); break;
    default: hipLaunchKernelGGL(k_valu<7>, dim3(v->grid), dim3(256), 0, 0, v->out, 3, 5, v->iters);
  }
}
struct MemCtx { uint4 *p, *q; size_t n; uint32_t *out; int grid; };
static void launch_read(void *c) { MemCtx *m = (MemCtx *)c; hipLaunchKernelGGL(k_read, dim3(m->grid), dim3(256), 0, 0, m->p, m->n, m->out); }
static void launch_copy(void *c) { MemCtx *m = (MemCtx *)c; hipLaunchKernelGGL(k_copy, dim3(m->grid), dim3(256), 0, 0, m->p, m->q, m->n); }

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "peaks";
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  if (!strcmp(mode, "peaks")) {
    int *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    const int iters = 4096;
    double tops[8];
    const int per_iter[8] = {16, 16, 16, 16 * 9, 16, 16, 16, 16};
    for (int kind = 0; kind < 8; kind++) {
      double best = 0;
      for (int wps : {4, 8}) {      // resident waves per SIMD
        ValuCtx v{out, kind, iters, cus * wps};
        const float ms = time_ms(launch_valu, &v, 5);
        const double t = (double)v.grid * 256.0 * iters * per_iter[kind] / (ms * 1e-3) / 1e12;
        if (t > best) best = t;
      }
      tops[kind] = best;
    }
    double vmax = tops[0];
    for (int k : {1, 2, 4}) if (tops[k] > vmax) vmax = tops[k];
    const size_t bytes = (size_t)4 << 30;   // 4 GiB >> 256 MiB Infinity Cache
    MemCtx m{};
    CK(hipMalloc(&m.p, bytes)); CK(hipMalloc(&m.q, bytes)); CK(hipMalloc(&m.out, 4));
    CK(hipMemset(m.p, 1, bytes)); CK(hipMemset(m.q, 0, bytes));
    m.n = bytes / 16;
    double best_read = 0, best_copy = 0;
    int best_grid = 0;
    for (int mult : {4, 8, 16, 32}) {
      m.grid = cus * mult;
      const float ms = time_ms(launch_read, &m, 5);
      const double gbs = (double)bytes / (ms * 1e-3) / 1e9;
      if (gbs > best_read) { best_read = gbs; best_grid = m.grid; }
      const float mc = time_ms(launch_copy, &m, 3);
      const double cg = 2.0 * (double)bytes / (mc * 1e-3) / 1e9;
      if (cg > best_copy) best_copy = cg;
    }
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"valu_int32_tops\": %.2f, \"valu_add_tops\": %.2f, "
           "\"valu_max_tops\": %.2f, \"valu_cndmask_tops\": %.2f, \"valu_add_vv_tops\": %.2f, \"valu_add3_tops\": %.2f, "
           "\"valu_pk_add_i16_insts_tops\": %.2f, \"valu_pk_max_i16_insts_tops\": %.2f, \"valu_nw_cell_mix_tops\": %.2f, "
           "\"hbm_read_gbs\": %.1f, \"hbm_copy_gbs\": %.1f, \"hbm_read_grid\": %d, \"note\": \"32-bit lane-ops/s (instructions x "
           "64 lanes; pk_* count one per lane, i.e. two int16 results each); valu_int32_tops = best of add / max / cndmask, 16 "
           "independent chains per lane, best of 4 and 8 waves per SIMD; hbm_read = uint4 stream over 4 GiB, best of 5\"}\n",
           prop.gcnArchName, cus, prop.clockRate / 1000, vmax, tops[0], tops[1], tops[2], tops[4], tops[7], tops[5], tops[6], tops[3],
           best_read, best_copy, best_grid);
    return 0;
  }
  if (!strcmp(mode, "rates")) {
    int *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    struct R { const char *name; void (*k)(int *, int, int, int); int per; };
    const R tab[] = {
        {"v_add_u32 (vv)", r_add_vv, 1}, {"v_sub_u32", r_sub_vv, 1}, {"v_max_i32", r_max_vv, 1}, {"v_min_i32", r_min_vv, 1},
        {"v_min_u32", r_minu_vv, 1}, {"v_and_b32", r_and_vv, 1}, {"v_xor_b32", r_xor_vv, 1}, {"v_lshl_or_b32", r_lshl_or, 1},
        {"v_lshl_add_u32", r_lshl_add, 1}, {"v_add3_u32", r_add3, 1}, {"v_max3_i32", r_max3, 1}, {"v_mad_i32_i24", r_mad_i24, 1},
        {"v_bfe_i32", r_bfe_i32, 1}, {"v_mov_b32_dpp wave_shr:1", r_mov_dpp_shr, 1}, {"v_mov_b32_dpp row_shr:1", r_mov_dpp_rowshr, 1},
        {"v_cndmask_b32 (vcc)", r_cndmask_vcc, 1}, {"v_cndmask_b32_e64 (sgpr pair)", r_cndmask_sgpr, 1},
        {"v_cmp_ge_i32 vcc + v_add", r_cmp_vcc, 2}, {"v_cmp_ge_i32 sgpr + v_add", r_cmp_sgpr, 2}, {"v_cmp vcc + v_cndmask", r_cmp_cndmask, 2},
        {"v_addc_co_u32", r_addc, 1}, {"v_pk_add_i16", r_pk_add_i16, 1}, {"v_pk_max_i16", r_pk_max_i16, 1}, {"v_pk_min_i16", r_pk_min_i16, 1},
        {"v_cmp_eq_u16_sdwa + v_add", r_cmp_eq_u16_sdwa, 2}, {"v_sad_u8", r_sad_u8, 1}, {"v_perm_b32", r_perm, 1}};
    const int iters = 2048;
    printf("{\"device\": \"%s\", \"unit\": \"SIMD cycles per wave64 instruction at the nominal %d MHz (8 and 1 waves per SIMD)\", \"rates\": {", prop.gcnArchName,
           prop.clockRate / 1000);
    bool first = true;
    for (const R &r : tab) {
      double cyc[2];
      int q = 0;
      for (int wps : {8, 1}) {
        const int grid = cus * wps;
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(r.k, dim3(grid), dim3(256), 0, 0, out, 3, 5, iters);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          CK(hipEventRecord(a, 0));
          hipLaunchKernelGGL(r.k, dim3(grid), dim3(256), 0, 0, out, 3, 5, iters);
          CK(hipEventRecord(b, 0));
          CK(hipEventSynchronize(b));
          float ms;
          CK(hipEventElapsedTime(&ms, a, b));
          if (ms < best) best = ms;
        }
        // wave-instructions per SIMD = (waves per SIMD) x iters x 16 x per ; cycles = time x clock
        const double winstr = (double)wps * iters * 16.0 * r.per;
        cyc[q++] = best * 1e-3 * (prop.clockRate * 1e3) / winstr;
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
      }
      printf("%s\"%s\": [%.2f, %.2f]", first ? "" : ", ", r.name, cyc[0], cyc[1]);
      first = false;
    }
    printf("}}\n");
    return 0;
  }
  if (!strcmp(mode, "occ")) {
    int *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    const int iters = 1024;
    printf("{\"device\": \"%s\", \"unit\": \"cycles at the nominal %d MHz; per_simd = SIMD cycles per wave-instruction, per_wave = cycles between two instructions of one wave\", \"instr_per_iter\": 64, \"curve\": {", prop.gcnArchName, prop.clockRate / 1000);
    for (int variant = 0; variant < 2; variant++) {
      printf("%s\"%s\": {", variant ? ", " : "", variant ? "four_chains_per_lane" : "one_chain_per_lane");
      for (int wps = 1; wps <= 8; wps++) {
        const int grid = cus * wps;
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
          CK(hipEventRecord(a, 0));
          if (variant) hipLaunchKernelGGL(k_chain4, dim3(grid), dim3(256), 0, 0, out, 3, 5, iters);
          else hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, 0, out, 3, 5, iters);
          CK(hipEventRecord(b, 0));
          CK(hipEventSynchronize(b));
          float ms;
          CK(hipEventElapsedTime(&ms, a, b));
          if (rep && ms < best) best = ms;
        }
        const double cyc = best * 1e-3 * (prop.clockRate * 1e3);
        const double per_wave_instr = (double)iters * 64.0;
        printf("%s\"%d\": {\"per_simd\": %.2f, \"per_wave\": %.2f, \"us\": %.1f}", wps > 1 ? ", " : "", wps, cyc / (per_wave_instr * wps), cyc / per_wave_instr, best * 1e3);
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
      }
      printf("}");
    }
    printf("}}\n");
    return 0;
  }
  if (!strcmp(mode, "launch")) {
    int *flag, *buf;
    const int n = 1 << 20;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&buf, (size_t)n * 4));
    CK(hipMemset(buf, 0, (size_t)n * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("{\"device\": \"%s\", \"unit\": \"us per launch, 400 back-to-back launches on one stream between two events\", \"rows\": [", prop.gcnArchName);
    bool first = true;
    for (int work = 0; work < 2; work++)
      for (int graph = 0; graph < 2; graph++)
        for (int grid : {1, 64, 256, 512, 1024, 2048, 4096}) {
          const int one = work ? 0 : 1;
          CK(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
          const int NL = 400;
          hipGraphExec_t ge = nullptr;
          if (graph) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            for (int k = 0; k < 8; k++) {
              if (work) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, buf, flag, n);
              else hipLaunchKernelGGL(k_noop, dim3(grid), dim3(256), 0, st, flag);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
          }
          float best = 1e30f;
          for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, st));
            if (graph) for (int k = 0; k < NL / 8; k++) CK(hipGraphLaunch(ge, st));
            else for (int k = 0; k < NL; k++) {
              if (work) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, st, buf, flag, n);
              else hipLaunchKernelGGL(k_noop, dim3(grid), dim3(256), 0, st, flag);
            }
            CK(hipEventRecord(b, st));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
          }
          if (ge) CK(hipGraphExecDestroy(ge));
          printf("%s{\"kernel\": \"%s\", \"graph\": %d, \"grid\": %d, \"us\": %.2f}", first ? "" : ", ", work ? "touch_4MB" : "noop", graph, grid, best * 1e3 / NL);
          first = false;
        }
    printf("]}\n");
    return 0;
  }
  const size_t n = argc > 2 ? (size_t)atoll(argv[2]) : ((size_t)1 << 28);
  if (!strcmp(mode, "store_bytes")) {
    uint8_t *q;
    CK(hipMalloc(&q, n));
    hipLaunchKernelGGL(k_store_bytes, dim3(2048), dim3(256), 0, 0, q, n);
    CK(hipDeviceSynchronize());
    printf("{\"mode\": \"store_bytes\", \"algorithmic_bytes\": %zu}\n", n);
  } else if (!strcmp(mode, "store_wide")) {
    uint4 *q;
    CK(hipMalloc(&q, n));
    hipLaunchKernelGGL(k_store_wide, dim3(cus * 8), dim3(256), 0, 0, q, n / 16);
    CK(hipDeviceSynchronize());
    printf("{\"mode\": \"store_wide\", \"algorithmic_bytes\": %zu}\n", n);
  } else if (!strcmp(mode, "read_wide")) {
    uint4 *p;
    uint32_t *out;
    CK(hipMalloc(&p, n)); CK(hipMalloc(&out, 4));
    CK(hipMemset(p, 1, n));
    hipLaunchKernelGGL(k_read, dim3(cus * 8), dim3(256), 0, 0, p, n / 16, out);
    CK(hipDeviceSynchronize());
    printf("{\"mode\": \"read_wide\", \"algorithmic_bytes\": %zu}\n", n);
  } else {
    fprintf(stderr, "usage: microbench peaks | store_bytes N | store_wide N | read_wide N\n");
    return 1;
  }
  return 0;
}
