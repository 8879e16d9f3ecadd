// tests/native/poa_quad_backend_emu.h -- TEST INFRASTRUCTURE ONLY: the CPU back end of svdss_amd/csrc/poa_quad_core.h,
// where the cross-lane primitives are meetings of the 64 fibres of wave_emu.h.  Included by poa_quad_emu.cpp in front
// of the core.
#pragma once
#define PQ_BACKEND 1
#include <cstdint>
#include "../../svdss_amd/csrc/poa_quad_defs.h"
// ------------------------------------------------------------------------------------------ emulator back end
#include "wave_emu.h"
#define PQ_DEV static inline
namespace pq {
inline int lane_id() { return wemu::lane_id(); }
#define PQ_SITE __LINE__
template <int GW>
struct Grp {
  static int g() { return wemu::lane_id() / GW; }
  static int l() { return wemu::lane_id() % GW; }
  static int shr1(int x, int fill, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    return l() == 0 ? fill : (int)(uint32_t)d[wemu::lane_id() - 1];
  }
  static int shl1(int x, int fill, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    return l() == GW - 1 ? fill : (int)(uint32_t)d[wemu::lane_id() + 1];
  }
  static int shr1z(int x, int site) { return shr1(x, 0, site); }
  static int shl1z(int x, int site) { return shl1(x, 0, site); }
  static int scan_max(int x, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    int m = PQ_INT_MIN;
    for (int i = g() * GW; i <= wemu::lane_id(); ++i) { const int v = (int)(uint32_t)d[i]; if (v > m) m = v; }
    return m;
  }
  static int scan_add(int x, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    int s = 0;
    for (int i = g() * GW; i <= wemu::lane_id(); ++i) s += (int)(uint32_t)d[i];
    return s;
  }
  static int all_max(int x, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    int m = PQ_INT_MIN;
    for (int i = g() * GW; i < g() * GW + GW; ++i) { const int v = (int)(uint32_t)d[i]; if (v > m) m = v; }
    return m;
  }
  static int all_min(int x, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    int m = 0x7fffffff;
    for (int i = g() * GW; i < g() * GW + GW; ++i) { const int v = (int)(uint32_t)d[i]; if (v < m) m = v; }
    return m;
  }
  static int last(int x, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    return (int)(uint32_t)d[g() * GW + GW - 1];
  }
  static int from(int x, int src_l, int site) {
    const uint64_t* d = wemu::meet((uint32_t)x, site);
    return (int)(uint32_t)d[g() * GW + (src_l & (GW - 1))];
  }
  static uint64_t bits(bool p, int site) {
    const uint64_t* d = wemu::meet(p ? 1 : 0, site);
    uint64_t m = 0;
    for (int i = 0; i < GW; ++i) if (d[g() * GW + i]) m |= 1ull << i;
    return m;
  }
  // first / last lane of the group with p (1 << 20 / -1 if none)
  static void first_last(bool p, int& first, int& last, int site) {
    const uint64_t m = bits(p, site);
    first = m ? __builtin_ctzll(m) : (1 << 20);
    last = m ? 63 - __builtin_clzll(m) : -1;
  }
};
inline bool wave_any(bool p, int site) {
  const uint64_t* d = wemu::meet(p ? 1 : 0, site);
  for (int i = 0; i < 64; ++i) if (d[i]) return true;
  return false;
}
// maximum over the wavefront of a value that is uniform within every group -> a wave-uniform value
template <int GW>
inline int wave_gmax(int x, int site) {
  const uint64_t* d = wemu::meet((uint32_t)x, site);
  int m = PQ_INT_MIN;
  for (int i = 0; i < 64; ++i) { const int v = (int)(uint32_t)d[i]; if (v > m) m = v; }
  return m;
}
inline void force_ready(uint32_t&) {}
inline void force_ready_i(int32_t&) {}
inline void vm_drain() {}
inline unsigned long long prof_clock() { return 0; }
inline void prof_out(const unsigned long long*) {}
inline void lds_sync(int site) { (void)wemu::meet(0, site); }
inline void mem_sync(int site) { (void)wemu::meet(0, site); }
inline int atomic_add(int32_t* p, int v) { const int o = *p; *p = o + v; return o; }
inline void atomic_max(int32_t* p, int v) { if (v > *p) *p = v; }
inline void atomic_add64(unsigned long long* p, unsigned long long v) { *p += v; }
inline int ctz64(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }
inline uint64_t load_u64(const uint8_t* p) { uint64_t x; __builtin_memcpy(&x, p, 8); return x; }
// the low bytes of four values side by side
inline uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return (a & 255u) | ((b & 255u) << 8) | ((c & 255u) << 16) | ((d & 255u) << 24); }
// signed 3-bit field of x at bit `at`
inline int sbfe3(uint32_t x, uint32_t at) { return (int)((int32_t)(x << (29u - (at & 31u))) >> 29); }
}  // namespace pq
