// Developer probe: how many independent random 16-byte reads per second does this GPU deliver from a buffer of
// the size of the search kernel's k-mer table?  (The search kernel's operations are such reads: one 16-B table
// entry, one 64-B BWT block, ... per lane per iteration; this is the ceiling of that access pattern.)
//   hipcc --offload-arch=gfx950 -O3 -o random_read_probe random_read_probe.hip && ./random_read_probe [GiB] [loads/lane]
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
    if (dependent) x += v.y;   // next address depends on the data: one access in flight per lane
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 64.0;
  const int per_lane = argc > 2 ? atoi(argv[2]) : 256;
  const uint64_t n = (uint64_t)(gib * (1ull << 30)) / 16;
  uint4* buf; uint32_t* out;
  if (hipMalloc(&buf, n * 16) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
  hipMemset(buf, 1, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int dependent = 0; dependent < 2; ++dependent)
    for (int blocks : {1024, 2048, 4096}) {
      probe<<<blocks, 256>>>(buf, n, 8, dependent, out);
      hipEventRecord(e0);
      probe<<<blocks, 256>>>(buf, n, per_lane, dependent, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double loads = (double)blocks * 256 * per_lane;
      printf("%.0f GiB buffer, %d blocks x 256, %s: %.2f G random 16-B loads/s (%.2f TB/s of 64-B lines)\n", gib, blocks,
             dependent ? "dependent (1 in flight per lane)" : "independent", loads / ms / 1e6, loads * 64 / ms / 1e9);
    }
  return 0;
}
