// tests/glue/occupy.hip — TEST INFRASTRUCTURE ONLY (tests/test_gpu_scale_and_edges.py): another tenant of the GPU inside the
// test process.  occupy_start(blocks, ms) launches, on a stream of its own, `blocks` blocks that each claim a whole CU's LDS and
// sleep-spin for `ms` milliseconds, and returns at once; occupy_wait() waits for them.  The persistent round tail of
// libdada2hip.so needs all its blocks resident at the same time: with half of the CUs held like this a launch of 250 blocks
// cannot be, and the test checks that the run then continues on the launch chains instead of failing.
#include <hip/hip_runtime.h>

__global__ void k_occupy(unsigned long long ticks, int *out) {
  extern __shared__ int s_hog[];
  s_hog[threadIdx.x] = (int)threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (out && s_hog[threadIdx.x] == -1) *out = 1;
}

static hipStream_t g_stream = nullptr;

extern "C" int occupy_start(int blocks, double ms) {
  const int lds = 160 * 1024;
  if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void *)k_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 2;
  hipLaunchKernelGGL(k_occupy, dim3(blocks), dim3(64), lds, g_stream, (unsigned long long)(ms * 1e5), (int *)nullptr);   // wall clock: 100 MHz
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
extern "C" int occupy_wait(void) { return g_stream && hipStreamSynchronize(g_stream) == hipSuccess ? 0 : 1; }
