// tests/emu/hip/hip_runtime.h — TEST INFRASTRUCTURE ONLY.  Never part of libdada2hip.so, never loaded by dada2_amd/.
//
// A functional stand-in for <hip/hip_runtime.h> that lets the REAL kernel and driver sources of dada2_amd/csrc be compiled
// for the host and executed lane by lane (tests/emu/emu.cpp): one fiber per GPU thread, blocks of a grid run one after
// the other, the cross-lane operations of a wave (DPP shifts, shuffles, ballots, readfirstlane) and __syncthreads() are
// rendezvous points of the fibers.  It exists because the authoring container has no GPU: kernel LOGIC (indexing, band
// geometry, traceback, list building) is debugged here; memory-model behaviour and performance only exist on the MI355X,
// where the -m gpu parity tests run the real thing.  The emulated library (tests/emu/build/libdada2hip_emu.so) is opened
// explicitly by tests/test_emu.py and by nothing else.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <cstdio>
#include <unistd.h>

// ---- qualifiers ------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

// ---- vector types -----------------------------------------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) int2 { int32_t x, y; };
struct alignas(16) int4 { int32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int32_t x, int32_t y) { return int2{x, y}; }
static inline int4 make_int4(int32_t x, int32_t y, int32_t z, int32_t w) { return int4{x, y, z, w}; }

// ---- the emulator's state of the running thread ---------------------------------------------------------------------------
namespace emu {
struct Idx { unsigned x, y, z; };
struct Cur {
  Idx tid, bid, bdim, gdim;
  void *dyn_lds;
};
extern Cur cur;   // (blocks and fibers run on ONE host thread at a time: a plain global, swapped by the scheduler)

enum Op { OP_SHFL, OP_SHFL_XOR, OP_DPP_SHR1, OP_DPP_SHL1, OP_BALLOT, OP_READFIRST };
// rendezvous of the lanes [base, base + width) of the calling lane's wave; returns this lane's result
uint64_t wave_op(Op op, int width, uint64_t value, int param, uint64_t old, bool bound_ctrl, bool pred);
void block_barrier();
void grid_yield();   // the polling lane of a grid-level wait (the persistent round-tail kernel's barrier): run the other blocks
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);
inline void *dyn_lds() { return cur.dyn_lds; }

template <typename T> inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "cross-lane values are at most 8 bytes");
  uint64_t b = 0;
  std::memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T> inline T from_bits(uint64_t b) {
  T v;
  std::memcpy(&v, &b, sizeof(T));
  return v;
}
}  // namespace emu

#define threadIdx (emu::cur.tid)
#define blockIdx (emu::cur.bid)
#define blockDim (emu::cur.bdim)
#define gridDim (emu::cur.gdim)
#define warpSize 64

// ---- cross-lane operations -----------------------------------------------------------------------------------------------
template <typename T> inline T __shfl(T v, int src, int width = 64) {
  return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL, width, emu::to_bits(v), src, emu::to_bits(v), false, false));
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
  return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL_XOR, width, emu::to_bits(v), mask, emu::to_bits(v), false, false));
}
inline unsigned long long __ballot(int pred) { return emu::wave_op(emu::OP_BALLOT, 64, 0, 0, 0, false, pred != 0); }
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { return __ballot(!pred) == 0; }
inline void __syncthreads() { emu::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline void __threadfence_system() {}
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline unsigned __brev(unsigned x) {
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  return __builtin_bswap32(x);
}
// (__hip_atomic_load / __hip_atomic_store are clang builtins in every language mode; only the scope names are HIP's)
#ifndef __HIP_MEMORY_SCOPE_SYSTEM
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
// HIP puts min / max overloads into the global namespace
#define EMU_MINMAX(T) \
  static inline T min(T a, T b) { return b < a ? b : a; } \
  static inline T max(T a, T b) { return a < b ? b : a; }
EMU_MINMAX(int) EMU_MINMAX(unsigned) EMU_MINMAX(long) EMU_MINMAX(unsigned long) EMU_MINMAX(long long) EMU_MINMAX(unsigned long long)
EMU_MINMAX(float) EMU_MINMAX(double)
#undef EMU_MINMAX

// ---- atomics (one host thread: plain read-modify-write) ----------------------------------------------------------------
template <typename T, typename U> inline T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <typename T, typename U> inline T atomicSub(T *p, U v) { T o = *p; *p = (T)(o - (T)v); return o; }
template <typename T, typename U> inline T atomicOr(T *p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <typename T, typename U> inline T atomicAnd(T *p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <typename T, typename U> inline T atomicMax(T *p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U> inline T atomicMin(T *p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U> inline T atomicExch(T *p, U v) { T o = *p; *p = (T)v; return o; }
template <typename T, typename U, typename V> inline T atomicCAS(T *p, U c, V v) { T o = *p; if (o == (T)c) *p = (T)v; return o; }

// ---- runtime API: device memory is host memory, streams execute at enqueue time -----------------------------------------
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorNotReady = 600, hipErrorNotSupported = 801, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct emuStream *hipStream_t;
struct emuEvent { std::chrono::steady_clock::time_point t; };
typedef emuEvent *hipEvent_t;
typedef void *hipGraph_t;
typedef void *hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipStreamCaptureModeRelaxed = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; int clockRate; char gcnArchName[64]; };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipDeviceGetPCIBusId(char *b, int n, int d) {   // (EMU_PCI_ID: a test holds the device's persistent-slot lock file from outside)
  if (const char *e = getenv("EMU_PCI_ID")) snprintf(b, n, "%s", e); else snprintf(b, n, "emu%d_%d", d, (int)getpid());
  return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { std::memset(p, 0, sizeof *p); p->multiProcessorCount = 8; p->clockRate = 1000000; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = *tot = (size_t)16 << 30; return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceTotalMem(size_t *tot, int) { *tot = (size_t)16 << 30; return hipSuccess; }
// EMU_GUARD=1: every "device" allocation ends at an inaccessible page, so the first store past a buffer faults where it
// happens (the fatal-signal handler of emu.cpp prints the frame) instead of corrupting the host heap
namespace emu { void *guard_alloc(size_t n); bool guard_free(void *p); }
inline hipError_t hipMalloc(void **p, size_t n) {
  static const bool guard = getenv("EMU_GUARD") != nullptr;
  *p = guard ? emu::guard_alloc(n) : std::aligned_alloc(256, (n + 255) & ~(size_t)255);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { if (!emu::guard_free(p)) std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emuEvent(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emuEvent(); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // (everything has run when it is enqueued)
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
// graphs: not emulated - the driver falls back to plain launches when capture is refused
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, void *, void *, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }

// kernels are called as functions by every fiber of every block
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  emu::launch(dim3(grid), dim3(block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); })
