// tests/emu/gcn.h — TEST INFRASTRUCTURE ONLY: the emulator's spelling of dada2_amd/csrc/gcn.h (same names, plain C++ over
// the fiber rendezvous of tests/emu/emu.cpp).  tests/emu/build.py puts it in place of the real header in the build copy.
#pragma once
#include <hip/hip_runtime.h>

#define GCN_SGPR_BUDGET(n)

namespace d2 {

static inline int gcn_max3(int a, int b, int c) { return std::max(a, std::max(b, c)); }
static inline int gcn_min3(int a, int b, int c) { return std::min(a, std::min(b, c)); }
static inline int gcn_sad_u8(uint32_t a, uint32_t b, int acc) {
  uint32_t s = (uint32_t)acc;
  for (int k = 0; k < 4; k++) {
    const int x = (int)((a >> (8 * k)) & 0xFFu), y = (int)((b >> (8 * k)) & 0xFFu);
    s += (uint32_t)(x > y ? x - y : y - x);
  }
  return (int)s;
}
static inline uint32_t gcn_bcast_byte0(uint32_t x) { return (x & 0xFFu) * 0x01010101u; }
static inline uint32_t gcn_push_low2(uint32_t acc, uint32_t x) { return (acc >> 2) | (x << 30); }
template <bool BOUND_CTRL> static inline int gcn_wave_shr1(int old, int src) {
  return (int)(uint32_t)emu::wave_op(emu::OP_DPP_SHR1, 64, (uint32_t)src, 0, (uint32_t)old, BOUND_CTRL, false);
}
template <bool BOUND_CTRL> static inline int gcn_wave_shl1(int old, int src) {
  return (int)(uint32_t)emu::wave_op(emu::OP_DPP_SHL1, 64, (uint32_t)src, 0, (uint32_t)old, BOUND_CTRL, false);
}
template <typename T> static inline const T *gcn_opaque_uniform(const T *p) { return p; }
static inline int gcn_opaque_lane(int v) { return v; }
static inline unsigned long long gcn_clock() { return 0; }
static inline void gcn_wave_sync() { (void)emu::wave_op(emu::OP_BALLOT, 64, 0, 0, 0, false, false); }
static inline void gcn_drain_stores() {}
static inline void gcn_release_agent() {}
static inline void gcn_acquire_agent() {}
static inline uint32_t gcn_load_agent(const uint32_t *p) { return *(const volatile uint32_t *)p; }
static inline void gcn_store_agent(uint32_t *p, uint32_t v) { *(volatile uint32_t *)p = v; }
static inline uint32_t gcn_add_agent(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
static inline int32_t gcn_load_system(const int32_t *p) { return *(const volatile int32_t *)p; }
static inline void gcn_raise_priority() {}
static inline int gcn_xcc_id() { return (int)(((blockIdx.x * 3u) % 7u) % 3u); }   // (uneven groups, not blockIdx order: what the barrier must not depend on)
static inline void gcn_poll_pause() { emu::grid_yield(); }   // the other blocks of the launch run while this lane waits
// a counter instead of a clock: every look at it is one "tick", so a wait that can never end still runs into its bound
static inline unsigned long long gcn_wall_clock() { static unsigned long long t = 0; return t += 64; }
constexpr unsigned long long GCN_WALL_HZ = 100000000ull;
static inline int gcn_readfirstlane(int v) { return (int)(uint32_t)emu::wave_op(emu::OP_READFIRST, 64, (uint32_t)v, 0, 0, false, false); }

}  // namespace d2
