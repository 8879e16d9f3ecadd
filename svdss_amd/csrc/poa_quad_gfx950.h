// poa_quad_gfx950.h -- the gfx950 back end of poa_quad_core.h: a group's cross-lane primitives as DPP / ds_bpermute /
// ballot code.  Included by poa_quad.hip in front of the core.
#pragma once
#define PQ_BACKEND 1
#include <cstdint>
#include "poa_quad_defs.h"
// ------------------------------------------------------------------------------------------ gfx950 back end
#include <hip/hip_runtime.h>
#define PQ_DEV __device__ __forceinline__
#define PQ_SITE 0
namespace pq {
PQ_DEV int lane_id() { return (int)threadIdx.x; }
template <int CTRL, int RMASK>
PQ_DEV int dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, RMASK, 0xf, false); }
PQ_DEV int imax_(int a, int b) { return a > b ? a : b; }
PQ_DEV int imin_(int a, int b) { return a < b ? a : b; }
// DPP controls (gfx9): row_shl:n 0x100+n, row_shr:n 0x110+n, row_ror:n 0x120+n, wave_shl:1 0x130, wave_shr:1 0x138,
// row_bcast:15 0x142, row_bcast:31 0x143.  A lane whose source lies outside its 16-lane row keeps `old`.
template <int GW>
struct Grp {
  static_assert(GW == 16 || GW == 32 || GW == 64, "group width");
  PQ_DEV static int g() { return (int)threadIdx.x / GW; }
  PQ_DEV static int l() { return (int)threadIdx.x % GW; }
  PQ_DEV static int shr1(int x, int fill, int) {
    if (GW == 16) return dpp<0x111, 0xf>(fill, x);
    const int r = dpp<0x138, 0xf>(fill, x);
    return GW == 64 ? r : (l() == 0 ? fill : r);
  }
  PQ_DEV static int shl1(int x, int fill, int) {
    if (GW == 16) return dpp<0x101, 0xf>(fill, x);
    const int r = dpp<0x130, 0xf>(fill, x);
    return GW == 64 ? r : (l() == GW - 1 ? fill : r);
  }
  // the same with 0 for the lane that has no neighbour: one instruction (bound_ctrl), no register to preload with the fill
  PQ_DEV static int shr1z(int x, int) {
    if (GW == 16) return __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    const int r = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);
    return GW == 64 ? r : (l() == 0 ? 0 : r);
  }
  PQ_DEV static int shl1z(int x, int) {
    if (GW == 16) return __builtin_amdgcn_update_dpp(0, x, 0x101, 0xf, 0xf, true);
    const int r = __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true);
    return GW == 64 ? r : (l() == GW - 1 ? 0 : r);
  }
  PQ_DEV static int scan_max(int x, int) {
    x = imax_(x, dpp<0x111, 0xf>(PQ_INT_MIN, x));
    x = imax_(x, dpp<0x112, 0xf>(PQ_INT_MIN, x));
    x = imax_(x, dpp<0x114, 0xf>(PQ_INT_MIN, x));
    x = imax_(x, dpp<0x118, 0xf>(PQ_INT_MIN, x));
    if (GW >= 32) x = imax_(x, dpp<0x142, 0xa>(PQ_INT_MIN, x));
    if (GW >= 64) x = imax_(x, dpp<0x143, 0xc>(PQ_INT_MIN, x));
    return x;
  }
  PQ_DEV static int scan_add(int x, int) {
    x += dpp<0x111, 0xf>(0, x);
    x += dpp<0x112, 0xf>(0, x);
    x += dpp<0x114, 0xf>(0, x);
    x += dpp<0x118, 0xf>(0, x);
    if (GW >= 32) x += dpp<0x142, 0xa>(0, x);
    if (GW >= 64) x += dpp<0x143, 0xc>(0, x);
    return x;
  }
  PQ_DEV static int last(int x, int) {
    if (GW == 64) return __builtin_amdgcn_readlane(x, 63);
    if (GW == 32) { const int a = __builtin_amdgcn_readlane(x, 31), b = __builtin_amdgcn_readlane(x, 63); return g() ? b : a; }
    return __builtin_amdgcn_ds_bpermute((int)(threadIdx.x | 15u) << 2, x);
  }
  PQ_DEV static int all_max(int x, int s) {
    if (GW == 16) {   // rotations inside the DPP row: every lane ends with the maximum of the 16
      x = imax_(x, dpp<0x128, 0xf>(x, x));
      x = imax_(x, dpp<0x124, 0xf>(x, x));
      x = imax_(x, dpp<0x122, 0xf>(x, x));
      x = imax_(x, dpp<0x121, 0xf>(x, x));
      return x;
    }
    return last(scan_max(x, s), s);
  }
  PQ_DEV static int all_min(int x, int s) {
    if (GW == 16) {
      x = imin_(x, dpp<0x128, 0xf>(x, x));
      x = imin_(x, dpp<0x124, 0xf>(x, x));
      x = imin_(x, dpp<0x122, 0xf>(x, x));
      x = imin_(x, dpp<0x121, 0xf>(x, x));
      return x;
    }
    return -all_max(-x, s);   // (callers pass values far from INT_MIN)
  }
  PQ_DEV static int from(int x, int src_l, int) {
    // (one group = the wavefront: what is uniform in the group is uniform, and a scalar lane select does it)
    if (GW == 64) return __builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(src_l) & 63);
    return __builtin_amdgcn_ds_bpermute((g() * GW + (src_l & (GW - 1))) << 2, x);
  }
  PQ_DEV static uint64_t bits(bool p, int) {
    const uint64_t m = __ballot(p);
    if (GW == 64) return m;
    return (m >> (g() * GW)) & ((1ull << GW) - 1ull);
  }
  // first / last lane of the group with p (1 << 20 / -1 if none)
  PQ_DEV static void first_last(bool p, int& first, int& last, int s) {
    const uint64_t m = bits(p, s);
    first = m ? (int)__builtin_ctzll(m) : (1 << 20);
    last = m ? 63 - (int)__builtin_clzll(m) : -1;
  }
};
PQ_DEV bool wave_any(bool p, int) { return __ballot(p) != 0ull; }
template <int GW>
PQ_DEV int wave_gmax(int x, int) {
  int m = __builtin_amdgcn_readlane(x, 0);
#pragma unroll
  for (int i = GW; i < 64; i += GW) m = imax_(m, __builtin_amdgcn_readlane(x, i));
  return m;
}
// the value of a load is needed HERE (the compiler would otherwise wait for it at its first use, inside the row loop)
PQ_DEV void force_ready(uint32_t& x) { asm volatile("" : "+v"(x)); }
PQ_DEV void force_ready_i(int32_t& x) { asm volatile("" : "+v"(x)); }
// s_waitcnt vmcnt(0) (expcnt / lgkmcnt left alone) as an instruction the compiler's wait-count pass sees: behind it no
// vector-memory result is pending, so code after the join of a rare branch that loads is not made to wait
PQ_DEV void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }
PQ_DEV unsigned long long prof_clock() { return wall_clock64(); }
__device__ unsigned long long g_poaq_prof[8];   // SVDSS_DEBUG: 100 MHz ticks in prepare, forward, traceback, update; steps, general steps
PQ_DEV void prof_out(const unsigned long long* p) { if (threadIdx.x == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_poaq_prof[k], p[k]); }
// LDS accesses of one wavefront are served in program order: nothing to wait for, the compiler must only keep the order
PQ_DEV void lds_sync(int) { __builtin_amdgcn_wave_barrier(); }
// global memory written by other lanes of this wavefront: the stores have to have left the wavefront's queue
PQ_DEV void mem_sync(int) { __syncthreads(); }
PQ_DEV int atomic_add(int32_t* p, int v) { return atomicAdd(p, v); }
PQ_DEV void atomic_max(int32_t* p, int v) { atomicMax(p, v); }
PQ_DEV void atomic_add64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
PQ_DEV int ctz64(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }
PQ_DEV uint64_t load_u64(const uint8_t* p) { uint64_t x; __builtin_memcpy(&x, p, 8); return x; }
// the low bytes of four values side by side: two byte permutes and an or
PQ_DEV uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return __builtin_amdgcn_perm(b, a, 0x0c0c0400u) | __builtin_amdgcn_perm(d, c, 0x04000c0cu);
}
// signed 3-bit field of x at bit `at` (v_bfe_i32)
PQ_DEV int sbfe3(uint32_t x, uint32_t at) { return __builtin_amdgcn_sbfe(x, at, 3u); }
}  // namespace pq
