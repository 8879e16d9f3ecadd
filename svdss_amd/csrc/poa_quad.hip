// poa_quad.hip -- POA consensus, several sub-clusters per wavefront: kernels and launcher around poa_quad_core.h
// (which holds the device code and says why).  First stage of svdss_poa_consensus_batch (poa.hip); what it hands back
// goes to poa_wave.hip's rounds.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "poa_quad.h"
#include "poa_quad_gfx950.h"
#include "poa_quad_core.h"

// (launch bounds: one wavefront per block; the LDS ring, not the registers, decides how many stay on a CU)
#ifndef PQ_MIN_WAVES
#define PQ_MIN_WAVES 1
#endif
template <int GW, int C>
__global__ void __launch_bounds__(64, PQ_MIN_WAVES) poa_quad_kernel(const PoaWaveTask* tasks, int n_tasks, const uint8_t* seqs, const int64_t* seq_off,
                                                      int32_t* ws32, int32_t* cons_len, int32_t* status, unsigned long long* cells, int qcap) {
  extern __shared__ __align__(16) int32_t poaq_lds[];
  pq::poaq_run<GW, C>(tasks, n_tasks, seqs, seq_off, ws32, cons_len, status, cells, poaq_lds, (int)blockIdx.x, qcap);
}

template <int GW, int C>
static hipError_t launch_gc(const PoaWaveTask* d_tasks, int n_tasks, int max_len, const uint8_t* d_seqs, const int64_t* d_seq_off, int32_t* ws32,
                            int32_t* d_len, int32_t* d_status, unsigned long long* d_cells, hipStream_t stream) {
  constexpr int G = 64 / GW;
  const size_t lds = pq::Geom<GW, C>::lds_bytes(max_len);
  if (lds > 160 * 1024 - 512) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute((const void*)poa_quad_kernel<GW, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((poa_quad_kernel<GW, C>), dim3((unsigned)((n_tasks + G - 1) / G)), dim3(64), lds, stream, d_tasks, n_tasks, d_seqs,
                     d_seq_off, ws32, d_len, d_status, d_cells, pq::Geom<GW, C>::qcap(max_len));
  return hipGetLastError();
}

bool poa_quad_supported(int gw, int cols) {
  for (int k = 0; k < kPoaQuadNVariants; ++k)
    if (kPoaQuadVariants[k][0] == gw && kPoaQuadVariants[k][1] == cols) return true;
  return false;
}

size_t poa_quad_lds_bytes(int gw, int cols, int max_len) {
  const size_t rw = (size_t)(gw * cols + 2 * PQ_GD), g = (size_t)(64 / gw);
  return sizeof(int32_t) * (g * PQ_RING * 3 * rw + 3 * rw + 16) + g * (size_t)(((max_len + gw * cols + 24 + 15) & ~15));
}

hipError_t poa_quad_launch(int gw, int cols, const PoaWaveTask* d_tasks, int n_tasks, int max_len, const uint8_t* d_seqs, const int64_t* d_seq_off,
                           int32_t* ws32, int32_t* d_len, int32_t* d_status, unsigned long long* d_cells, hipStream_t stream) {
#define PQ_CASE(GW_, C_) \
  if (gw == GW_ && cols == C_) return launch_gc<GW_, C_>(d_tasks, n_tasks, max_len, d_seqs, d_seq_off, ws32, d_len, d_status, d_cells, stream)
  PQ_CASE(16, 3); PQ_CASE(16, 4); PQ_CASE(16, 5); PQ_CASE(16, 6); PQ_CASE(16, 7);
  PQ_CASE(32, 2); PQ_CASE(32, 3); PQ_CASE(32, 4);
  PQ_CASE(64, 1); PQ_CASE(64, 2);
#undef PQ_CASE
  return hipErrorInvalidValue;
}

// SVDSS_DEBUG: in-kernel phase timers summed over the wavefronts run since the last call
void poa_quad_debug_report() {
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(pq::g_poaq_prof), sizeof h) != hipSuccess) return;
  fprintf(stderr, "[poa_quad] 100MHz ticks (sum over wavefronts): prepare %llu forward %llu traceback %llu update %llu | row steps %llu, %llu the general way\n",
          h[0], h[1], h[2], h[3], h[4], h[5]);
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(pq::g_poaq_prof), h, sizeof h);
}

// ---------------------------------------------------------------------------------------------- primitives self-test
// The emulator of the tests checks the kernel's logic in terms of the group primitives; this kernel checks that the DPP /
// permute implementations of the primitives do what the emulator's do (tests/test_poa_quad_gpu.py compares with numpy).
template <int GW>
__device__ void selftest_gw(const int32_t* in, int32_t* out) {
  typedef pq::Grp<GW> GR;
  const int lane = (int)threadIdx.x;
  const int x = in[lane];
  int32_t* o = out + lane * 12;
  o[0] = GR::shr1(x, -7, 0);
  o[1] = GR::shl1(x, -9, 0);
  o[2] = GR::scan_max(x, 0);
  o[3] = GR::scan_add(x, 0);
  o[4] = GR::all_max(x, 0);
  o[5] = GR::all_min(x, 0);
  o[6] = GR::last(x, 0);
  o[7] = GR::from(x, (GR::last(x, 0) >> 3) & (GW - 1), 0);   // (the source lane is uniform within the group)
  const uint64_t b = GR::bits((x & 1) != 0, 0);
  o[8] = (int32_t)(uint32_t)b;
  o[9] = (int32_t)(uint32_t)(b >> 32);
  o[10] = pq::wave_any(x == 12345, 0) ? 1 : 0;
  o[11] = pq::ctz64(~(b >> (lane % GW)));
}

__global__ void __launch_bounds__(64) poa_quad_selftest_kernel(const int32_t* in, int32_t* out) {
  selftest_gw<16>(in, out);
  selftest_gw<32>(in, out + 64 * 12);
  selftest_gw<64>(in, out + 2 * 64 * 12);
}

// in: 64 values; out: 3 x 64 x 12 values (GW = 16, 32, 64)
extern "C" int svdss_poa_quad_selftest(const int32_t* in, int32_t* out, int32_t device) {
  if (!in || !out || device < 0) return -1;
  if (hipSetDevice(device) != hipSuccess) return -2;
  int32_t *d_in = nullptr, *d_out = nullptr;
  if (hipMalloc(&d_in, 64 * 4) != hipSuccess || hipMalloc(&d_out, 3 * 64 * 12 * 4) != hipSuccess) return -3;
  int rc = 0;
  if (hipMemcpy(d_in, in, 64 * 4, hipMemcpyHostToDevice) != hipSuccess) rc = -4;
  if (!rc) {
    hipLaunchKernelGGL(poa_quad_selftest_kernel, dim3(1), dim3(64), 0, 0, d_in, d_out);
    if (hipDeviceSynchronize() != hipSuccess) rc = -5;
  }
  if (!rc && hipMemcpy(out, d_out, 3 * 64 * 12 * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = -6;
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc;
}
