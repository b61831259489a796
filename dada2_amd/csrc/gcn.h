// gcn.h — the handful of gfx950 instructions the kernels spell out themselves (wave64, CDNA4).  Everything here is a thin
// name over ONE machine instruction, so that the kernels read as algorithms and the functional emulator under tests/emu
// (test infrastructure: it runs the kernels lane by lane on the CPU) can supply the same names in plain C++.
#pragma once
#include <hip/hip_runtime.h>

// Cap on the scalar registers of a kernel (the excess is kept in lanes of a vector register): 256-thread blocks are admitted
// per CU up to floor(800 / (ceil(sgpr / 16) * 16 + 16)), so a kernel that wants 8 blocks per CU has to stay at or below 80.
#define GCN_SGPR_BUDGET(n) __attribute__((amdgpu_num_sgpr(n)))

namespace d2 {

// max(a, b, c) in one VOP3 instruction (the compiler keeps an inner max when one of its results is also compared)
static __device__ __forceinline__ int gcn_max3(int a, int b, int c) {
  int e;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(e) : "v"(a), "v"(b), "v"(c));
  return e;
}
static __device__ __forceinline__ int gcn_min3(int a, int b, int c) {
  int e;
  asm("v_min3_i32 %0, %1, %2, %3" : "=v"(e) : "v"(a), "v"(b), "v"(c));
  return e;
}
// acc + sum over the four bytes of |a.byte - b.byte| in one instruction (v_sad_u8)
static __device__ __forceinline__ int gcn_sad_u8(uint32_t a, uint32_t b, int acc) {
  int e;
  asm("v_sad_u8 %0, %1, %2, %3" : "=v"(e) : "v"(a), "v"(b), "v"(acc));
  return e;
}
// (acc >> 2) | (x << 30) in one instruction (v_alignbit_b32 takes bits 33..2 of {x, acc}): the two LOW bits of x are pushed
// into the top of a bit string - no mask, no shift, no or
static __device__ __forceinline__ uint32_t gcn_push_low2(uint32_t acc, uint32_t x) {
  return __builtin_amdgcn_alignbit(x, acc, 2);
}
// the low byte of x in all four bytes, one full-rate instruction (v_perm_b32 with selector 0; x * 0x01010101 would be a
// quarter-rate v_mul_lo_u32)
static __device__ __forceinline__ uint32_t gcn_bcast_byte0(uint32_t x) { return __builtin_amdgcn_perm(x, x, 0u); }
// DPP wave shifts by one lane: lane L reads `src` of lane L-1 (shr) / L+1 (shl); a lane without a source keeps `old`
// (bound_ctrl = false) or reads 0 (bound_ctrl = true)
template <bool BOUND_CTRL>
static __device__ __forceinline__ int gcn_wave_shr1(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, BOUND_CTRL);
}
template <bool BOUND_CTRL>
static __device__ __forceinline__ int gcn_wave_shl1(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, 0x130, 0xF, 0xF, BOUND_CTRL);
}
static __device__ __forceinline__ int gcn_readfirstlane(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Values the optimiser must not see through (no instruction): a uniform pointer / a per-lane integer that come out of an empty
// volatile asm are not invariants of an enclosing loop any more, so nothing derived from them is hoisted out of it and kept
// alive (spilled) across it - the persistent round-tail kernel's outer loop spans every phase of a round
template <typename T> static __device__ __forceinline__ const T *gcn_opaque_uniform(const T *p) { asm volatile("" : "+s"(p)); return p; }
static __device__ __forceinline__ int gcn_opaque_lane(int v) { asm volatile("" : "+v"(v)); return v; }
// The lanes of a wave execute in lockstep, so data one lane leaves in LDS is there for the others at the next instruction;
// this marks the places where a kernel relies on that.  No instruction: it only stops the compiler from moving memory
// operations across the point (and gives the lane-by-lane emulator its rendezvous).
// shader-clock timestamp (s_memtime) for the in-kernel phase traces of tools/trace_round.py
static __device__ __forceinline__ unsigned long long gcn_clock() { return __builtin_amdgcn_s_memtime(); }
// (the bare wave barrier is IntrNoMem - it pins the machine schedule, not LLVM's IR passes; the wavefront-scope
// fences either side emit no instruction beyond waitcnts but keep the optimiser from moving LDS accesses across the hand-off)
static __device__ __forceinline__ void gcn_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- inter-workgroup hand-offs inside one launch (the persistent round-tail kernel; MI355X: 8 XCDs with private L2s, a CU's
// L1 is never refreshed by other CUs' stores).  Producer: its own stores drained, block barrier, ONE lane releases at agent
// scope (writes the XCD's dirty L2 lines back) and then touches the flag; consumer: ONE lane polls relaxed, acquires at agent
// scope once (drops its CU's stale L1 lines), block barrier, plain loads.
static __device__ __forceinline__ void gcn_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
static __device__ __forceinline__ void gcn_release_agent() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (restated where the compiler cannot drop it behind the write-back)
}
static __device__ __forceinline__ void gcn_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
static __device__ __forceinline__ uint32_t gcn_load_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void gcn_store_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ uint32_t gcn_add_agent(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ int32_t gcn_load_system(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// the XCC (XCD) this wave runs on, 0-7
static __device__ __forceinline__ int gcn_xcc_id() { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x)); return x & 7; }
// wave priority 3 (s_setprio): a latency-bound kernel sharing its CUs with throughput kernels issues ahead of them
static __device__ __forceinline__ void gcn_raise_priority() { __builtin_amdgcn_s_setprio(3); }
// what a polling lane does between two looks at a word another workgroup (or the host) will write
static __device__ __forceinline__ void gcn_poll_pause() { __builtin_amdgcn_s_sleep(2); }   // (0 and 12 measured: no difference, profiles/r07f)
// constant-rate clock (100 MHz) for the bounds of those spins
static __device__ __forceinline__ unsigned long long gcn_wall_clock() { return wall_clock64(); }
constexpr unsigned long long GCN_WALL_HZ = 100000000ull;

}  // namespace d2
