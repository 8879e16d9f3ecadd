// Developer probe: how many VALU / SALU / mixed instructions per cycle does one SIMD of this GPU retire with 1, 2, 4, 8
// wavefronts resident?  (POA's wavefronts are chains of dependent VALU and SALU instructions, 4 per SIMD: DESIGN 4.)
//   hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// kind 0: dependent VALU chain; 1: dependent SALU chain; 2: VALU and SALU chains interleaved 1:1; 3: four independent
// VALU chains; 4: SALU pairs that do not depend on each other
template <int KIND>
__global__ void __launch_bounds__(64) spin(int iters, uint32_t* out) {
  uint32_t v = threadIdx.x, v2 = v + 1, v3 = v + 2, v4 = v + 3;
  uint32_t s = blockIdx.x, s2 = s + 1;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) asm volatile(REP64("v_add_u32 %0, %0, %0\n") : "+v"(v));
    if (KIND == 1) asm volatile(REP64("s_add_u32 %0, %0, %0\n") : "+s"(s) : : "scc");
    if (KIND == 2) asm volatile(REP64("v_add_u32 %0, %0, %0\ns_add_u32 %1, %1, %1\n") : "+v"(v), "+s"(s) : : "scc");
    if (KIND == 3) asm volatile(REP16("v_add_u32 %0, %0, %0\nv_add_u32 %1, %1, %1\nv_add_u32 %2, %2, %2\nv_add_u32 %3, %3, %3\n")
                                : "+v"(v), "+v"(v2), "+v"(v3), "+v"(v4));
    if (KIND == 4) asm volatile(REP64("s_add_u32 %0, %0, %0\ns_add_u32 %1, %1, %1\n") : "+s"(s), "+s"(s2) : : "scc");
  }
  if ((v ^ v2 ^ v3 ^ v4 ^ s ^ s2) == 0x12345u) out[0] = v;
}

template <int KIND>
static void run(const char* what, int instr_per_iter, int n_simd, uint32_t* out, int threads = 64) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int clock_khz = 0;
  hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
  for (int per_simd : {1, 2, 4, 8}) {
    const int blocks = n_simd * per_simd, iters = 20000;
    spin<KIND><<<blocks, threads>>>(100, out);
    hipEventRecord(e0);
    spin<KIND><<<blocks, threads>>>(iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cycles = (double)ms * 1e-3 * clock_khz * 1e3;
    const double per_wave = cycles / ((double)iters * instr_per_iter);
    printf("%-34s %d wave(s) per SIMD: %.2f cycles per instruction per wave, %.3f instructions per cycle per SIMD\n", what, per_simd,
           per_wave, per_simd / per_wave);
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_simd = prop.multiProcessorCount * 4;
  printf("%s: %d CUs, %d kHz (workgroups of one wavefront; the dispatcher spreads them over the SIMDs)\n", prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate);
  uint32_t* out;
  hipMalloc(&out, 64);
  run<0>("dependent v_add_u32", 64, n_simd, out);
  run<3>("four independent v_add_u32 chains", 64, n_simd, out);
  run<1>("dependent s_add_u32", 64, n_simd, out);
  run<4>("two independent s_add_u32 chains", 128, n_simd, out);
  run<2>("v_add_u32 / s_add_u32 alternating", 128, n_simd, out);
  // round 4: does a wavefront whose upper 32 lanes are switched off (workgroup of 32 threads) issue its vector
  // instructions in half the time?  (POA rows are 15-40 lanes wide)
  run<0>("dependent v_add_u32, 32 lanes on", 64, n_simd, out, 32);
  run<3>("four independent chains, 32 lanes", 64, n_simd, out, 32);
  run<0>("dependent v_add_u32, 16 lanes on", 64, n_simd, out, 16);
  return 0;
}
