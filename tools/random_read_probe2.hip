// Developer probe (round 2): the random-read ceiling of tools/random_read_probe.hip under other kinds of device
// allocation -- uncached (does the request size drop below 128 bytes?), physically contiguous (do larger page-table
// fragments cut the TLB misses of a 64 GiB table?).
//   hipcc --offload-arch=gfx950 -O3 -o random_read_probe2 random_read_probe2.hip && ./random_read_probe2 [GiB]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256, 4) probe(const uint4* buf, uint64_t n_entries, int per_lane, int dependent, uint32_t* out) {
  uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  x = x * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (int i = 0; i < per_lane; ++i) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint4 v = buf[x % n_entries];
    acc += v.x ^ v.w;
    if (dependent) x += v.y;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 64.0;
  const uint64_t n = (uint64_t)(gib * (1ull << 30)) / 16;
  uint32_t* out;
  hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char* name; unsigned flags; } kinds[] = {{"default", hipDeviceMallocDefault}, {"uncached", hipDeviceMallocUncached},
#ifdef hipDeviceMallocContiguous
                                                          {"contiguous", hipDeviceMallocContiguous},
#endif
                                                          {"finegrained", hipDeviceMallocFinegrained}};
  for (auto& k : kinds) {
    uint4* buf = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&buf, n * 16, k.flags);
    if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", k.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    hipMemset(buf, 1, n * 16);
    for (int dependent = 0; dependent < 2; ++dependent) {
      const int blocks = 2048, per_lane = 256;
      probe<<<blocks, 256>>>(buf, n, 8, dependent, out);
      hipEventRecord(e0);
      probe<<<blocks, 256>>>(buf, n, per_lane, dependent, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double loads = (double)blocks * 256 * per_lane;
      printf("%-12s %.0f GiB, %s: %.2f G random 16-B loads/s\n", k.name, gib, dependent ? "dependent  " : "independent", loads / ms / 1e6);
    }
    hipFree(buf);
  }
  return 0;
}
