// index_gpu.hip -- suffix sorting of the index text on the GPU (`SVDSS index`, texts below 2^31 symbols).
//
// Same result as the host builder of index_build.cpp (the suffix array of a text is unique): radix sort of the
// 63-bit keys of the first 21 symbols, then prefix doubling -- every round sorts (rank[p], rank[p+h]) pairs of all
// suffixes with hipcub's radix sort and renumbers the groups with a scan -- until every suffix has its own rank.
// A suffix that runs off the end of the text sorts before its extensions (rank -1 there, as on the host).
// Whole-array passes at HBM bandwidth: ~12 rounds x (sort + scan + scatter) of n items, versus seconds of
// 256-core host time per 100 M symbols.  Returns non-zero (and the caller falls back to the host builder) when
// there is no GPU or not enough free HBM.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace {

#define GCHK(expr)                    \
  do {                                \
    if ((expr) != hipSuccess) {       \
      (void)hipGetLastError();        \
      return 1;                       \
    }                                 \
  } while (0)

struct Bufs {
  void* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  ~Bufs() { for (void* q : p) if (q) (void)hipFree(q); }
};

__global__ void __launch_bounds__(256) key0_kernel(const uint8_t* t, int64_t n, uint64_t* keys, uint32_t* pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k = 0;
#pragma unroll
  for (int s = 0; s < 21; ++s) {
    const int64_t q = i + s;
    const uint64_t c = q < n ? t[q] : 0;     // symbols past the end count as 0 here; the doubling rounds settle them
    k |= c << (60 - 3 * s);
  }
  keys[i] = k;
  pos[i] = (uint32_t)i;
}

// flag[x] = x if the key at x starts a new group, else 0 (x = 0 always starts one); an inclusive max-scan turns
// it into "start of my group"
__global__ void __launch_bounds__(256) heads_kernel(const uint64_t* keys, int64_t n, uint32_t* head, uint32_t* is_head) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  const bool h = x == 0 || keys[x] != keys[x - 1];
  head[x] = h ? (uint32_t)x : 0u;
  is_head[x] = h ? 1u : 0u;
}

__global__ void __launch_bounds__(256) scatter_rank_kernel(const uint32_t* sa, const uint32_t* start, int64_t n, uint32_t* rank) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  rank[sa[x]] = start[x];
}

__global__ void __launch_bounds__(256) pair_keys_kernel(const uint32_t* sa, const uint32_t* rank, int64_t n, int64_t h,
                                                        uint64_t* keys) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  const int64_t p = sa[x], q = p + h;
  const uint64_t second = q < n ? (uint64_t)rank[q] + 1u : 0u;
  keys[x] = ((uint64_t)rank[p] << 32) | second;
}

struct MaxOp {
  __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

}  // namespace

// suffix array of t[0, n) into sa_out (host pointers); 0 = done on the GPU
extern "C" int svdss_sa32_gpu(const uint8_t* t, int64_t n, int32_t* sa_out) {
  if (n <= 0 || n >= (int64_t)0x7fffffff) return 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return 1; }
  size_t free_b = 0, total_b = 0;
  GCHK(hipMemGetInfo(&free_b, &total_b));
  const size_t N = (size_t)n;
  if ((N * 40 + ((size_t)1 << 30)) > free_b) return 1;
  Bufs B;
  uint8_t* d_t; uint64_t *k0, *k1; uint32_t *v0, *v1, *rank, *head, *ish;
  GCHK(hipMalloc(&B.p[0], N + 64)); d_t = (uint8_t*)B.p[0];
  GCHK(hipMalloc(&B.p[1], N * 8)); k0 = (uint64_t*)B.p[1];
  GCHK(hipMalloc(&B.p[2], N * 8)); k1 = (uint64_t*)B.p[2];
  GCHK(hipMalloc(&B.p[3], N * 4)); v0 = (uint32_t*)B.p[3];
  GCHK(hipMalloc(&B.p[4], N * 4)); v1 = (uint32_t*)B.p[4];
  GCHK(hipMalloc(&B.p[5], N * 4)); rank = (uint32_t*)B.p[5];
  GCHK(hipMalloc(&B.p[6], N * 4)); head = (uint32_t*)B.p[6];
  // is_head shares k1's storage between the sort and the next key pass? no: keep it simple, its own buffer
  Bufs B2;
  GCHK(hipMalloc(&B2.p[0], N * 4)); ish = (uint32_t*)B2.p[0];
  GCHK(hipMalloc(&B2.p[1], 16));
  unsigned long long* d_cnt = (unsigned long long*)B2.p[1];
  GCHK(hipMemcpy(d_t, t, N, hipMemcpyHostToDevice));
  size_t tb_sort = 0, tb_scan = 0, tb_sum = 0;
  GCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, k0, k1, v0, v1, (int)n, 0, 64));
  GCHK(hipcub::DeviceScan::InclusiveScan(nullptr, tb_scan, head, head, MaxOp(), (int)n));
  GCHK(hipcub::DeviceReduce::Sum(nullptr, tb_sum, ish, d_cnt, (int)n));
  size_t tb = tb_sort > tb_scan ? tb_sort : tb_scan;
  if (tb_sum > tb) tb = tb_sum;
  GCHK(hipMalloc(&B2.p[2], tb + 16));
  void* d_tmp = B2.p[2];
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(key0_kernel, dim3(nb), dim3(256), 0, 0, d_t, n, k0, v0);
  GCHK(hipGetLastError());
  uint64_t *kin = k0, *kout = k1;
  uint32_t *vin = v0, *vout = v1;
  int64_t h = 21;
  for (int round = 0; round < 64; ++round) {
    size_t tbs = tb;
    GCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tbs, kin, kout, vin, vout, (int)n, 0, 64));
    // kout / vout: sorted keys and the suffixes in that order
    hipLaunchKernelGGL(heads_kernel, dim3(nb), dim3(256), 0, 0, kout, n, head, ish);
    GCHK(hipGetLastError());
    tbs = tb;
    GCHK(hipcub::DeviceReduce::Sum(d_tmp, tbs, ish, d_cnt, (int)n));
    tbs = tb;
    GCHK(hipcub::DeviceScan::InclusiveScan(d_tmp, tbs, head, head, MaxOp(), (int)n));
    unsigned long long groups = 0;
    GCHK(hipMemcpy(&groups, d_cnt, sizeof groups, hipMemcpyDeviceToHost));
    if ((int64_t)groups == n) {   // every suffix alone in its group: vout is the suffix array
      GCHK(hipMemcpy(sa_out, vout, N * 4, hipMemcpyDeviceToHost));
      return 0;
    }
    hipLaunchKernelGGL(scatter_rank_kernel, dim3(nb), dim3(256), 0, 0, vout, head, n, rank);
    GCHK(hipGetLastError());
    // next round: the suffixes stay in their current order (vout), keys = (rank[p], rank[p+h])
    hipLaunchKernelGGL(pair_keys_kernel, dim3(nb), dim3(256), 0, 0, vout, rank, n, h, kin);
    GCHK(hipGetLastError());
    // sort (kin, vout) -> (kout, vin): swap the value buffers
    uint32_t* tv = vin; vin = vout; vout = tv;
    h *= 2;
  }
  return 1;
}
