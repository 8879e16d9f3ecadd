// Developer probe: which compute units does bit b of a stream's CU mask (hipExtStreamCreateWithCUMask) stand for on
// this GPU?  One launch per mask bit; every workgroup records the XCC, shader engine and CU it ran on.
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask_probe cu_mask_probe.hip && ./cu_mask_probe
// (What for: DESIGN 5a -- the search kernel and the call-side DP kernels each fill the chip, so launched side by side
// they take turns; a partition of the CUs between their streams lets them run at the same time.)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  // (stay a little, so that the workgroups of a launch spread over everything the mask allows)
  const long long t0 = clock64();
  while (clock64() - t0 < 20000) {}
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  printf("%s: %d CUs\n", prop.gcnArchName, n_cu);
  const int words = (n_cu + 31) / 32;
  const int blocks = 2048;
  uint32_t* out;
  hipMalloc(&out, blocks * 8);
  std::vector<uint32_t> h(blocks * 2);
  for (int bit = 0; bit < n_cu; ++bit) {
    std::vector<uint32_t> mask(words, 0);
    mask[bit / 32] = 1u << (bit % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, words, mask.data()) != hipSuccess) { printf("bit %d: cannot create the stream\n", bit); continue; }
    where<<<blocks, 64, 0, st>>>(out);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    hipStreamDestroy(st);
    std::set<uint32_t> seen;
    for (int b = 0; b < blocks; ++b) {
      const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 15u;
      // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (+ se bit 3 higher on some parts)
      const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
      seen.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    printf("bit %3d:", bit);
    for (uint32_t s : seen) printf(" xcc %u se %u sh %u cu %u;", s >> 16, (s >> 8) & 255u, (s >> 4) & 15u, s & 15u);
    printf("\n");
  }
  return 0;
}
