// index_gpu.hip -- construction of the whole FM-index in HBM (`SVDSS index`, and the index every rank of a
// multi-GPU run builds for itself).  Stands where ropebwt3's `main_build` stands in the reference
// (/root/reference/main.cpp:34-37); the result is what ping_pong.cpp:245 restores.
//
// Everything happens on the device, sized for 288 GB: the text  contig $ revcomp $ ...  (n bytes), the suffix
// array (8 B per suffix while sorting), the inverse ranks (8 B), and the sort buffers of one piece.  For
// GRCh38 lengths (n = 6.18e9) that is ~125 GB at the peak.
//
// Suffix sorting (any n < 2^34):
//   phase 1  the suffixes are cut into pieces by the bucket of their first 4 symbols (a 4096-bin histogram;
//            consecutive buckets are merged while a piece stays below the piece size), every piece is
//            radix-sorted on the 63-bit key of its first 21 symbols (hipcub), and the groups of equal keys
//            get the suffix-array index of their first member as rank;
//   phase 2  Larsson-Sadakane prefix doubling on the suffixes that are still tied ONLY (near-random DNA: a
//            few percent): they stay in suffix-array order in a compact list, every round sorts
//            (group, rank[p + h]) pairs piece by piece (pieces end at group boundaries), writes the refined
//            order and ranks back, and drops the suffixes that are alone now.  A piece gathers all its keys
//            before it scatters any rank, and a group never straddles two pieces, so a group is always
//            sorted on one consistent set of ranks.
// The suffix array of a text is unique, so the index is byte-identical to the host builder's
// (index_build.cpp).  A suffix that runs off the end of the text sorts before its extensions.
//
// BWT + rank blocks (fmd_layout.h): one wavefront per 128-symbol block, two coalesced 64-entry reads of the
// suffix array, the three bit planes from wave ballots; per-block symbol counts are scanned afterwards.
//
// Returns non-zero (and the caller falls back to the host builder) when there is no GPU, not enough free
// HBM, or a degenerate input (one 4-symbol bucket or one tied group above 2^30 suffixes).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <atomic>
#include <thread>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/svdss_hip.h"
#include "fmd_layout.h"
#include "index_host.h"

namespace {

#define GCHK(expr)                                                                   \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      if (getenv("SVDSS_INDEX_VERBOSE"))                                             \
        fprintf(stderr, "[index_gpu] %s: %s\n", #expr, hipGetErrorString(e_));       \
      (void)hipGetLastError();                                                       \
      return SVDSS_GPU_NO;                                                                      \
    }                                                                                \
  } while (0)

typedef unsigned long long ull;

constexpr int SVDSS_GPU_NO = -1;   // "not possible on this device / with this text": the caller uses the host builder

// device allocations released when the builder leaves, unless handed over with take()
struct Pool {
  std::vector<void*> v;
  // an arena some of the big scratch buffers are carved from (alloc_big) instead of being allocated and freed: the memory
  // that becomes the k-mer table afterwards (svdss_index_build_gpu).  Not owned: release() of a pointer inside it does nothing.
  uint8_t* arena = nullptr;
  size_t arena_cap = 0, arena_at = 0;
  int alloc_big(void** out, size_t bytes) {
    const size_t a = (arena_at + 255) & ~(size_t)255;
    if (arena && a + bytes <= arena_cap) { *out = arena + a; arena_at = a + bytes; return 0; }
    return alloc(out, bytes);
  }
  ~Pool() { for (void* p : v) if (p) (void)hipFree(p); }
  int alloc(void** out, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { (void)hipGetLastError(); return 1; }
    v.push_back(p);
    *out = p;
    return 0;
  }
  void release(void* p) {
    if (arena && (uint8_t*)p >= arena && (uint8_t*)p < arena + arena_cap) return;   // (carved from the arena -- its first piece has the arena's own address)
    for (void*& q : v) if (q == p && p) { (void)hipFree(p); q = nullptr; }
  }
  void* take(void* p) {
    for (void*& q : v) if (q == p) q = nullptr;
    return p;
  }
};
#define PALLOC(pool, ptr, bytes) do { if ((pool).alloc((void**)&(ptr), (bytes))) return SVDSS_GPU_NO; } while (0)
#define PALLOC_BIG(pool, ptr, bytes) do { if ((pool).alloc_big((void**)&(ptr), (bytes))) return SVDSS_GPU_NO; } while (0)

constexpr int KEY_SYMS = 21;
constexpr int BUCKET_BITS = 12;              // first 4 symbols
constexpr int N_BUCKETS = 1 << BUCKET_BITS;

// ---------------------------------------------------------------- text

__global__ void __launch_bounds__(256) build_text_kernel(const uint8_t* src, const int64_t* src_off, int32_t n_contigs,
                                                         int64_t total, uint8_t* text, int* bad) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  int lo = 0, hi = n_contigs;   // last contig with src_off <= g
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (src_off[mid] <= g) lo = mid; else hi = mid;
  }
  const int64_t s = src_off[lo], len = src_off[lo + 1] - s, j = g - s;
  const int64_t o = 2 * s + 2 * (int64_t)lo;   // every earlier contig took 2 * (len + 1) symbols
  const uint8_t c = src[g];
  if (c < 1 || c > 5) { *bad = 1; return; }    // '$' cannot appear inside a record
  text[o + j] = c;
  text[o + len + 1 + (len - 1 - j)] = (uint8_t)svdss_comp(c);
}

// 8 symbols (bytes, first at the lowest address) -> 24 bits, first symbol in the top 3 bits
__device__ __forceinline__ uint64_t pack8(uint64_t w) {
  uint64_t x = __builtin_bswap64(w) & 0x0707070707070707ull;
  x = (x | (x >> 5)) & 0x003f003f003f003full;
  x = (x | (x >> 10)) & 0x00000fff00000fffull;
  x = (x | (x >> 20)) & 0xffffffull;
  return x;
}

// the 63-bit keys (21 symbols, 3 bits each, first symbol on top) of the 8 suffixes starting at i0 (a multiple of 8);
// symbols past the end of the text count as 0 (the text is followed by zero bytes)
__device__ __forceinline__ void keys8(const uint8_t* text, int64_t i0, uint64_t key[8]) {
  const uint64_t* w = (const uint64_t*)(text + i0);
  const uint64_t H = (pack8(w[0]) << 24) | pack8(w[1]);   // symbols 0..15
  const uint64_t L = (pack8(w[2]) << 24) | pack8(w[3]);   // symbols 16..31
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint64_t hmask = (1ull << (3 * (16 - k))) - 1ull;
    key[k] = ((H & hmask) << (3 * (k + 5))) | (L >> (33 - 3 * k));
  }
}

__global__ void __launch_bounds__(256) bucket_hist_kernel(const uint8_t* text, int64_t n, ull* hist) {
  __shared__ uint32_t h[N_BUCKETS];
  for (int i = threadIdx.x; i < N_BUCKETS; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i0 < n; i0 += stride) {
    uint64_t key[8];
    keys8(text, i0, key);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k < n) atomicAdd(&h[key[k] >> (63 - BUCKET_BITS)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N_BUCKETS; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[i], (ull)h[i]);
}

// (key, position) of every suffix whose bucket lies in [b_lo, b_hi), in no particular order
__global__ void __launch_bounds__(256) select_keys_kernel(const uint8_t* text, int64_t n, uint32_t b_lo, uint32_t b_hi,
                                                          uint64_t* keys, uint64_t* pos, ull* counter) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  const int64_t first = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const int64_t wave_first = first - (int64_t)lane * 8;
  for (int64_t base = wave_first; base < n; base += stride) {   // wave-uniform trip count
    const int64_t i0 = base + (int64_t)lane * 8;
    uint64_t key[8];
    uint32_t sel = 0;
    if (i0 < n) {
      keys8(text, i0, key);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t b = (uint32_t)(key[k] >> (63 - BUCKET_BITS));
        if (i0 + k < n && b >= b_lo && b < b_hi) sel |= 1u << k;
      }
    }
    const int c = __builtin_popcount(sel);
    int incl = c;   // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    const int total = __shfl(incl, 63, 64);
    ull wbase = 0;
    if (lane == 63 && total) wbase = atomicAdd(counter, (ull)total);
    wbase = __shfl(wbase, 63, 64);
    ull o = wbase + (ull)(incl - c);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((sel >> k) & 1u) { keys[o] = key[k]; pos[o] = (uint64_t)(i0 + k); ++o; }
  }
}

// ---------------------------------------------------------------- groups

// nh[j] = 1 when the key at j starts a group; hs[j] = j there, 0 elsewhere (a max-scan turns it into "my group's start")
__global__ void __launch_bounds__(256) heads_kernel(const uint64_t* keys, int64_t m, uint8_t* nh, uint32_t* hs) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const bool h = j == 0 || keys[j] != keys[j - 1];
  nh[j] = h ? 1 : 0;
  hs[j] = h ? (uint32_t)j : 0u;
}

// phase 1: piece at suffix-array indices [base, base + m)
__global__ void __launch_bounds__(256) scatter1_kernel(const uint64_t* pos, const uint8_t* nh, const uint32_t* gstart,
                                                       int64_t m, int64_t base, uint64_t* sa, uint64_t* rank,
                                                       uint32_t* unres) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint64_t p = pos[j];
  sa[base + j] = p;
  rank[p] = (uint64_t)(base + gstart[j]);
  unres[j] = (nh[j] && (j == m - 1 || nh[j + 1])) ? 0u : 1u;
}

// tied suffixes of a piece -> the compact list (suffix-array order is kept: off[] is the exclusive scan of the tied flags)
__global__ void __launch_bounds__(256) append_kernel(const uint32_t* off, int64_t m,
                                                     const uint64_t* x_src, int64_t x_base, const uint64_t* pos,
                                                     const uint8_t* nh, uint64_t* ux, uint64_t* upos, uint8_t* uhead,
                                                     int64_t u_base) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m || off[j + 1] == off[j]) return;   // off[] has m + 1 entries: tied <=> the scan advances
  const int64_t u = u_base + off[j];
  ux[u] = x_src ? x_src[j] : (uint64_t)(x_base + j);
  upos[u] = pos[j];
  uhead[u] = nh[j];
}

// phase 2 --------------------------------------------------------------

// largest u in [lo, hi] with uhead[u] != 0 (atomicMax on *out, which starts at 0 = "none": u is stored + 1)
__global__ void __launch_bounds__(256) last_head_kernel(const uint8_t* uhead, int64_t lo, int64_t hi, ull* out) {
  const int64_t u = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  ull best = 0;
  if (u <= hi && uhead[u]) best = (ull)u + 1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const ull t = __shfl_xor(best, d, 64);
    best = t > best ? t : best;
  }
  if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

__global__ void __launch_bounds__(256) ghead_kernel(const uint8_t* uhead, int64_t m, uint32_t* hs) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  hs[j] = uhead[j] ? (uint32_t)j : 0u;
}

__global__ void __launch_bounds__(256) pair_keys_kernel(const uint64_t* upos, const uint32_t* gidx, const uint64_t* rank,
                                                        int64_t m, int64_t n, int64_t h, int shift, uint64_t* keys,
                                                        uint64_t* vals) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint64_t p = upos[j];
  const int64_t q = (int64_t)p + h;
  const uint64_t second = q < n ? rank[q] + 1ull : 0ull;   // off the end: before everything
  keys[j] = ((uint64_t)gidx[j] << shift) | second;
  vals[j] = p;
}

__global__ void __launch_bounds__(256) scatter2_kernel(const uint64_t* ux, const uint64_t* pos, const uint8_t* nh,
                                                       const uint32_t* hstart, int64_t m, uint64_t* sa, uint64_t* rank,
                                                       uint32_t* unres) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint64_t p = pos[j];
  sa[ux[j]] = p;
  rank[p] = ux[hstart[j]];
  unres[j] = (nh[j] && (j == m - 1 || nh[j + 1])) ? 0u : 1u;
}

struct MaxOp {
  __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

int bits_for(uint64_t v) {   // number of bits needed to hold v
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b ? b : 1;
}

// ---------------------------------------------------------------- BWT and blocks

template <class SA>
__global__ void __launch_bounds__(256) bwt_blocks_kernel(const SA* sa, const uint8_t* text, int64_t n, int64_t nb,
                                                         svdss_u4* blocks, uint32_t* cnt /* 4 x nb */, int64_t* dollar,
                                                         ull* n_dollar) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < nb; b += waves) {
    uint64_t m0[2], m1[2], m2[2], mv[2];   // planes p0, p1, p2 and "position exists"
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int64_t x = b * SVDSS_BLOCK_SYMS + 64 * s + lane;
      int sym = -1;
      if (x < n) {
        const int64_t p = (int64_t)sa[x];
        sym = p == 0 ? 0 : text[p - 1];   // the text ends with '$': BWT[x] for p == 0 is text[n - 1] = 0
        if (sym == 0) dollar[atomicAdd(n_dollar, 1ull)] = x;
      }
      const bool acgt = sym >= 1 && sym <= 4;
      m0[s] = __ballot((acgt && ((sym - 1) & 1)) || sym == 5);
      m1[s] = __ballot(acgt && ((sym - 1) & 2));
      m2[s] = __ballot(sym == 0 || sym == 5);
      mv[s] = __ballot(sym >= 0);
    }
    if (lane < 4) {
      const int s = lane >> 1, sh = (lane & 1) * 32;
      svdss_u4 q;
      q.x = 0;
      q.y = (uint32_t)(m0[s] >> sh);
      q.z = (uint32_t)(m1[s] >> sh);
      q.w = (uint32_t)(m2[s] >> sh);
      blocks[4 * b + lane] = q;
      // symbol lane + 1 (code = lane): p0 == bit 0 of the code, p1 == bit 1, not special
      const uint64_t a0 = (lane & 1) ? m0[0] : ~m0[0], a1 = (lane & 1) ? m0[1] : ~m0[1];
      const uint64_t c0 = (lane & 2) ? m1[0] : ~m1[0], c1 = (lane & 2) ? m1[1] : ~m1[1];
      cnt[(int64_t)lane * nb + b] = (uint32_t)(__builtin_popcountll(a0 & c0 & ~m2[0] & mv[0]) +
                                               __builtin_popcountll(a1 & c1 & ~m2[1] & mv[1]));
    }
  }
}

__global__ void __launch_bounds__(256) fill_counts_kernel(svdss_u4* blocks, const uint32_t* run /* 4 x nb */, int64_t nb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // quarter index
  if (i >= 4 * nb) return;
  blocks[i].x = run[(i & 3) * nb + (i >> 2)];
}

template <class T>
__global__ void __launch_bounds__(256) narrow_kernel(const uint64_t* in, int64_t n, T* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (T)in[i];
}

inline unsigned grid_for(int64_t items) { return (unsigned)((items + 255) / 256); }

// suffix array of text[0, n) (device, followed by >= 64 zero bytes) into sa (device, uint64[n]); rank = scratch uint64[n]
int suffix_sort(const uint8_t* text, int64_t n, uint64_t* sa, uint64_t* rank, size_t free_bytes, bool verbose,
                uint8_t* arena = nullptr, size_t arena_bytes = 0) {
  Pool P;
  P.arena = arena; P.arena_cap = arena_bytes;
  // piece size: the sort buffers of a piece cost ~56 B per suffix
  int64_t M = (int64_t)1 << 29;
  while (M > ((int64_t)1 << 20) && (size_t)M * 64 > free_bytes / 2) M >>= 1;
  if (const char* e = getenv("SVDSS_SA_PIECE")) M = std::max<int64_t>(16, atoll(e));   // tests: many small pieces
  const int64_t cap = std::min<int64_t>(n, M);   // a piece may exceed M only through one big bucket / group: refused

  uint64_t *k0, *k1, *v0, *v1;
  uint32_t *hs, *unres;
  uint8_t* nh;
  ull* d_cnt;
  PALLOC_BIG(P, k0, (size_t)cap * 8); PALLOC_BIG(P, k1, (size_t)cap * 8);
  PALLOC_BIG(P, v0, (size_t)cap * 8); PALLOC_BIG(P, v1, (size_t)cap * 8);
  PALLOC(P, hs, (size_t)cap * 4); PALLOC(P, unres, (size_t)cap * 4 + 8);
  PALLOC(P, nh, (size_t)cap + 8);
  PALLOC(P, d_cnt, (N_BUCKETS + 8) * sizeof(ull));
  size_t tb_sort = 0, tb_scan = 0, tb_sum = 0;
  {
    hipcub::DoubleBuffer<uint64_t> dk(k0, k1), dv(v0, v1);
    GCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, dk, dv, (int)cap, 0, 64));
    GCHK(hipcub::DeviceScan::InclusiveScan(nullptr, tb_scan, hs, hs, MaxOp(), (int)cap));
    GCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_sum, unres, unres, (int)cap + 1));
  }
  const size_t tb = std::max(tb_sort, std::max(tb_scan, tb_sum)) + 256;
  void* d_tmp;
  PALLOC(P, d_tmp, tb);

  // the compact list of tied suffixes, double-buffered; grows on demand
  struct UList { uint64_t* x = nullptr; uint64_t* pos = nullptr; uint8_t* head = nullptr; int64_t cap = 0; };
  UList U[2];
  int64_t un = 0;   // items in U[0]
  auto grow = [&](UList& L, int64_t need, int64_t keep) -> int {
    if (need <= L.cap) return 0;
    int64_t nc = std::max<int64_t>(need, std::max<int64_t>(L.cap * 2, 1 << 16));
    nc = std::min<int64_t>(nc, n);
    if (nc < need) return SVDSS_GPU_NO;
    uint64_t *nx, *np2; uint8_t* nhd;
    PALLOC(P, nx, (size_t)nc * 8); PALLOC(P, np2, (size_t)nc * 8); PALLOC(P, nhd, (size_t)nc);
    if (keep > 0) {
      GCHK(hipMemcpy(nx, L.x, (size_t)keep * 8, hipMemcpyDeviceToDevice));
      GCHK(hipMemcpy(np2, L.pos, (size_t)keep * 8, hipMemcpyDeviceToDevice));
      GCHK(hipMemcpy(nhd, L.head, (size_t)keep, hipMemcpyDeviceToDevice));
    }
    P.release(L.x); P.release(L.pos); P.release(L.head);
    L.x = nx; L.pos = np2; L.head = nhd; L.cap = nc;
    return 0;
  };

  // sorts (k0, v0)[0, m) on bits [0, end_bit), finds the groups, leaves: sorted keys/values in *ks/*vs, nh, hs = group start
  auto sort_and_group = [&](int64_t m, int end_bit, uint64_t** ks, uint64_t** vs) -> int {
    hipcub::DoubleBuffer<uint64_t> dk(k0, k1), dv(v0, v1);
    size_t tbs = tb;
    GCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tbs, dk, dv, (int)m, 0, end_bit));
    *ks = dk.Current(); *vs = dv.Current();
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(m)), dim3(256), 0, 0, *ks, m, nh, hs);
    GCHK(hipGetLastError());
    tbs = tb;
    GCHK(hipcub::DeviceScan::InclusiveScan(d_tmp, tbs, hs, hs, MaxOp(), (int)m));
    return 0;
  };
  // exclusive scan of unres[0, m] (one extra zero slot) in place; returns the number of tied suffixes
  auto scan_unres = [&](int64_t m, int64_t* n_un) -> int {
    GCHK(hipMemsetAsync(unres + m, 0, 4, 0));
    size_t tbs = tb;
    GCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tbs, unres, unres, (int)m + 1));
    uint32_t tot = 0;
    GCHK(hipMemcpy(&tot, unres + m, 4, hipMemcpyDeviceToHost));
    *n_un = tot;
    return 0;
  };

  // ---- phase 1 ----
  std::vector<std::pair<uint32_t, uint32_t>> pieces;   // bucket ranges
  std::vector<int64_t> piece_size;
  if (n <= M) {
    pieces.emplace_back(0u, (uint32_t)N_BUCKETS);
    piece_size.push_back(n);
  } else {
    GCHK(hipMemset(d_cnt, 0, N_BUCKETS * sizeof(ull)));
    hipLaunchKernelGGL(bucket_hist_kernel, dim3(2048), dim3(256), 0, 0, text, n, d_cnt);
    GCHK(hipGetLastError());
    std::vector<ull> hist(N_BUCKETS);
    GCHK(hipMemcpy(hist.data(), d_cnt, N_BUCKETS * sizeof(ull), hipMemcpyDeviceToHost));
    uint32_t b = 0;
    while (b < (uint32_t)N_BUCKETS) {
      int64_t tot = (int64_t)hist[b];
      if (tot > cap) return SVDSS_GPU_NO;   // one bucket larger than a piece: degenerate text, host builder
      uint32_t e = b + 1;
      while (e < (uint32_t)N_BUCKETS && tot + (int64_t)hist[e] <= M) tot += (int64_t)hist[e++];
      if (tot > 0) { pieces.emplace_back(b, e); piece_size.push_back(tot); }
      b = e;
    }
  }
  int64_t base = 0;
  for (size_t pc = 0; pc < pieces.size(); ++pc) {
    const int64_t m = piece_size[pc];
    GCHK(hipMemset(d_cnt, 0, sizeof(ull)));
    hipLaunchKernelGGL(select_keys_kernel, dim3(4096), dim3(256), 0, 0, text, n, pieces[pc].first, pieces[pc].second,
                       k0, v0, d_cnt);
    GCHK(hipGetLastError());
    uint64_t *ks, *vs;
    if (sort_and_group(m, 63, &ks, &vs)) return SVDSS_GPU_NO;
    hipLaunchKernelGGL(scatter1_kernel, dim3(grid_for(m)), dim3(256), 0, 0, vs, nh, hs, m, base, sa, rank, unres);
    GCHK(hipGetLastError());
    int64_t add = 0;
    if (scan_unres(m, &add)) return SVDSS_GPU_NO;
    if (add) {
      if (grow(U[0], un + add, un)) return SVDSS_GPU_NO;
      hipLaunchKernelGGL(append_kernel, dim3(grid_for(m)), dim3(256), 0, 0, unres, m, (const uint64_t*)nullptr,
                         base, vs, nh, U[0].x, U[0].pos, U[0].head, un);
      GCHK(hipGetLastError());
      un += add;
    }
    base += m;
  }
  if (base != n) return SVDSS_GPU_NO;
  if (verbose) fprintf(stderr, "[index_gpu] phase 1: %zu piece(s), %lld of %lld suffixes still tied\n", pieces.size(),
                       (long long)un, (long long)n);

  // ---- phase 2 ----
  const int shift = bits_for((uint64_t)n + 1);
  int64_t h = KEY_SYMS;
  for (int round = 0; un > 0; ++round) {
    if (round >= 64) return SVDSS_GPU_NO;
    if (grow(U[1], un, 0)) return SVDSS_GPU_NO;
    int64_t s = 0, un_next = 0;
    while (s < un) {
      int64_t e = un;   // exclusive end of this piece: the last group head at or below s + cap, when the rest does not fit
      if (un - s > cap) {
        GCHK(hipMemset(d_cnt, 0, sizeof(ull)));
        hipLaunchKernelGGL(last_head_kernel, dim3(grid_for(cap)), dim3(256), 0, 0, U[0].head, s + 1, s + cap, d_cnt);
        GCHK(hipGetLastError());
        ull r = 0;
        GCHK(hipMemcpy(&r, d_cnt, sizeof r, hipMemcpyDeviceToHost));
        if (r == 0) return SVDSS_GPU_NO;   // one group larger than a piece
        e = (int64_t)r - 1;
      }
      const int64_t m = e - s;
      if (bits_for((uint64_t)m) + shift > 64) return SVDSS_GPU_NO;
      hipLaunchKernelGGL(ghead_kernel, dim3(grid_for(m)), dim3(256), 0, 0, U[0].head + s, m, hs);
      GCHK(hipGetLastError());
      size_t tbs = tb;
      GCHK(hipcub::DeviceScan::InclusiveScan(d_tmp, tbs, hs, hs, MaxOp(), (int)m));
      hipLaunchKernelGGL(pair_keys_kernel, dim3(grid_for(m)), dim3(256), 0, 0, U[0].pos + s, hs, rank, m, n, h, shift,
                         k0, v0);
      GCHK(hipGetLastError());
      uint64_t *ks, *vs;
      if (sort_and_group(m, std::min(64, shift + bits_for((uint64_t)m)), &ks, &vs)) return SVDSS_GPU_NO;
      hipLaunchKernelGGL(scatter2_kernel, dim3(grid_for(m)), dim3(256), 0, 0, U[0].x + s, vs, nh, hs, m, sa, rank, unres);
      GCHK(hipGetLastError());
      int64_t add = 0;
      if (scan_unres(m, &add)) return SVDSS_GPU_NO;
      if (add) {
        hipLaunchKernelGGL(append_kernel, dim3(grid_for(m)), dim3(256), 0, 0, unres, m,
                           (const uint64_t*)(U[0].x + s), (int64_t)0, vs, nh, U[1].x, U[1].pos, U[1].head, un_next);
        GCHK(hipGetLastError());
        un_next += add;
      }
      s = e;
    }
    std::swap(U[0], U[1]);
    un = un_next;
    h *= 2;
    if (verbose) fprintf(stderr, "[index_gpu] phase 2 round %d (h = %lld): %lld still tied\n", round, (long long)h,
                         (long long)un);
  }
  GCHK(hipDeviceSynchronize());
  return 0;
}

}  // namespace

// Builds the index of the given records in the HBM of `device` and leaves it resident there (d_text, d_sa,
// d_blocks, d_dollar of *ix); the host side of *ix gets n, acc, the rank blocks and the '$' list -- text and
// suffix array stay on the device until svdss_index_fetch_host() is called (save, another device).
// 0 = done; -1 = not possible here (no GPU / memory / degenerate text): use the host builder;
// SVDSS_EINVAL / SVDSS_ERANGE as the host builder reports them.
static int build_gpu_once(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs, int32_t device,
                          svdss_index* ix, bool defer_host_blocks, size_t table_bytes);

int svdss_index_build_gpu(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs, int32_t device,
                          svdss_index* ix, bool defer_host_blocks, size_t table_bytes) {
  int rc = build_gpu_once(contigs, lens, n_contigs, device, ix, defer_host_blocks, table_bytes);
  // The table's memory, taken ahead of the sort, sits on top of the suffix array, the text and the sort's other buffers:
  // where the build only just fits it is the arena that made an allocation fail.  Everything of the failed attempt has
  // been released (Pool); once more without it before the caller drops to the host builder, which takes minutes.
  if (rc == SVDSS_GPU_NO && table_bytes > 0) {
    if (getenv("SVDSS_INDEX_VERBOSE")) fprintf(stderr, "[index_gpu] no room with the table's %zu bytes taken ahead: again without\n", table_bytes);
    rc = build_gpu_once(contigs, lens, n_contigs, device, ix, defer_host_blocks, 0);
  }
  return rc;
}

static int build_gpu_once(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs, int32_t device,
                          svdss_index* ix, bool defer_host_blocks, size_t table_bytes) {
  if (!contigs || !lens || n_contigs <= 0 || !ix) return SVDSS_EINVAL;
  const bool verbose = getenv("SVDSS_INDEX_VERBOSE") != nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) { (void)hipGetLastError(); return SVDSS_GPU_NO; }
  GCHK(hipSetDevice(device));
  int64_t n = 0, total = 0;
  std::vector<int64_t> src_off((size_t)n_contigs + 1, 0);
  for (int i = 0; i < n_contigs; ++i) {
    if (lens[i] < 0) return SVDSS_EINVAL;
    total += lens[i];
    src_off[(size_t)i + 1] = total;
    n += 2 * (lens[i] + 1);
  }
  if (n >= ((int64_t)1 << 34) - 2) return SVDSS_GPU_NO;
  const bool wide = n >= (int64_t)0x7fffffff || getenv("SVDSS_FORCE_SA64") != nullptr;
  size_t free_b = 0, total_b = 0;
  GCHK(hipMemGetInfo(&free_b, &total_b));
  const size_t N = (size_t)n;
  if (N * 18 + N / 2 + ((size_t)3 << 30) > free_b) return SVDSS_GPU_NO;   // text + SA + ranks + blocks, plus the sort buffers
  const auto t_build0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {      // SVDSS_INDEX_VERBOSE: seconds since the build began (waits for the device)
    if (!verbose) return;
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[index_gpu] %-34s at +%.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build0).count());
  };
  Pool P;
  uint8_t* d_textalloc;
  const size_t text_bytes = N + 128 + 16;
  PALLOC(P, d_textalloc, text_bytes);
  GCHK(hipMemset(d_textalloc, 0, text_bytes));
  uint8_t* d_text = d_textalloc + 64;
  {
    uint8_t* d_src; int64_t* d_off; int* d_bad;
    PALLOC(P, d_src, (size_t)total + 16);
    PALLOC(P, d_off, src_off.size() * sizeof(int64_t));
    PALLOC(P, d_bad, sizeof(int));
    GCHK(hipMemset(d_bad, 0, sizeof(int)));
    GCHK(hipMemcpy(d_src, contigs, (size_t)total, hipMemcpyHostToDevice));
    GCHK(hipMemcpy(d_off, src_off.data(), src_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    if (total > 0) {
      hipLaunchKernelGGL(build_text_kernel, dim3(grid_for(total)), dim3(256), 0, 0, d_src, d_off, n_contigs, total,
                         d_text, d_bad);
      GCHK(hipGetLastError());
    }
    int bad = 0;
    GCHK(hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
    P.release(d_src); P.release(d_off); P.release(d_bad);
    if (bad) return SVDSS_EINVAL;
  }
  mark("records up, text laid down");
  uint64_t *d_sa64, *d_rank;
  PALLOC(P, d_sa64, N * 8 + 16);
  // table_bytes > 0 (the restore paths): the memory of the k-mer table is allocated HERE and lent to the sort -- the rank
  // array (8 N bytes) and the sort's four key / value buffers are carved from it -- instead of ~70 GB being allocated,
  // released after the sort and 64 GiB allocated again for the table.  The driver clears what it hands out and what it
  // gets back (30-50 GB/s, and host <-> device copies crawl meanwhile): at GRCh38 lengths that was ~140 GB of clearing
  // per restore, part of it beside the stream that follows (profiles/r05z_e2e_lib_ab.txt, section 3).
  uint8_t* d_arena = nullptr;
  if (table_bytes > 0 && P.alloc((void**)&d_arena, table_bytes) == 0) { P.arena = d_arena; P.arena_cap = table_bytes; }
  else d_arena = nullptr;
  PALLOC_BIG(P, d_rank, N * 8);
  GCHK(hipMemGetInfo(&free_b, &total_b));
  mark("suffix array + rank buffers");
  {
    const size_t used = (P.arena_at + 255) & ~(size_t)255;
    uint8_t* rest = P.arena && used < P.arena_cap ? P.arena + used : nullptr;
    if (suffix_sort(d_text, n, d_sa64, d_rank, free_b, verbose, rest, rest ? P.arena_cap - used : 0)) return SVDSS_GPU_NO;
  }
  mark("suffixes sorted");
  P.release(d_rank);
  void* d_sa = d_sa64;
  if (!wide) {
    uint32_t* d_sa32;
    PALLOC(P, d_sa32, N * 4 + 16);
    hipLaunchKernelGGL(narrow_kernel<uint32_t>, dim3(grid_for(n)), dim3(256), 0, 0, d_sa64, n, d_sa32);
    GCHK(hipGetLastError());
    GCHK(hipDeviceSynchronize());
    P.release(d_sa64);
    d_sa = d_sa32;
  }
  // BWT, rank blocks, '$' list
  const int64_t nb = n / SVDSS_BLOCK_SYMS + 1;
  svdss_u4* d_blocks; uint32_t* d_cntb; int64_t* d_dollar; ull* d_nd;
  const int64_t n_dollar = 2 * (int64_t)n_contigs;
  PALLOC(P, d_blocks, (size_t)nb * 64);
  PALLOC(P, d_cntb, (size_t)nb * 16 + 16);
  PALLOC(P, d_dollar, (size_t)(n_dollar + 1) * 8 * 2);
  PALLOC(P, d_nd, sizeof(ull));
  GCHK(hipMemset(d_nd, 0, sizeof(ull)));
  GCHK(hipMemset(d_blocks, 0, (size_t)nb * 64));
  {
    const unsigned g = (unsigned)std::min<int64_t>((nb + 3) / 4, 1 << 20);
    if (wide)
      hipLaunchKernelGGL(bwt_blocks_kernel<uint64_t>, dim3(g), dim3(256), 0, 0, (const uint64_t*)d_sa, d_text, n, nb,
                         d_blocks, d_cntb, d_dollar, d_nd);
    else
      hipLaunchKernelGGL(bwt_blocks_kernel<uint32_t>, dim3(g), dim3(256), 0, 0, (const uint32_t*)d_sa, d_text, n, nb,
                         d_blocks, d_cntb, d_dollar, d_nd);
    GCHK(hipGetLastError());
  }
  ull nd = 0;
  GCHK(hipMemcpy(&nd, d_nd, sizeof nd, hipMemcpyDeviceToHost));
  mark("BWT + rank blocks");
  if ((int64_t)nd != n_dollar) return SVDSS_GPU_NO;
  // totals per symbol (64-bit), then 32-bit exclusive scans over the blocks
  int64_t totals[4];
  {
    size_t tb1 = 0, tb2 = 0;
    ull* d_tot;
    PALLOC(P, d_tot, 4 * sizeof(ull));
    typedef hipcub::TransformInputIterator<ull, hipcub::CastOp<ull>, const uint32_t*> WideIt;   // 64-bit sums
    if (nb > ((int64_t)1 << 30)) return SVDSS_GPU_NO;
    GCHK(hipcub::DeviceReduce::Sum(nullptr, tb1, WideIt(d_cntb, hipcub::CastOp<ull>()), d_tot, (int)nb));
    GCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, d_cntb, d_cntb, (int)nb));
    void* d_tmp;
    PALLOC(P, d_tmp, std::max(tb1, tb2) + 256);
    for (int c = 0; c < 4; ++c) {
      size_t tbs = tb1;
      WideIt it(d_cntb + (size_t)c * nb, hipcub::CastOp<ull>());
      GCHK(hipcub::DeviceReduce::Sum(d_tmp, tbs, it, d_tot + c, (int)nb));
    }
    ull tot[4];
    GCHK(hipMemcpy(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost));
    for (int c = 0; c < 4; ++c) {
      totals[c] = (int64_t)tot[c];
      if (tot[c] > 0xffffffffull) return SVDSS_ERANGE;   // the 32-bit block counters of fmd_layout.h
    }
    for (int c = 0; c < 4; ++c) {
      size_t tbs = tb2;
      GCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tbs, d_cntb + (size_t)c * nb, d_cntb + (size_t)c * nb, (int)nb));
    }
    hipLaunchKernelGGL(fill_counts_kernel, dim3(grid_for(4 * nb)), dim3(256), 0, 0, d_blocks, d_cntb, nb);
    GCHK(hipGetLastError());
    // sorted '$' positions
    size_t tb3 = 0;
    GCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb3, d_dollar, d_dollar + n_dollar + 1, (int)n_dollar));
    void* d_tmp3;
    PALLOC(P, d_tmp3, tb3 + 256);
    GCHK(hipcub::DeviceRadixSort::SortKeys(d_tmp3, tb3, d_dollar, d_dollar + n_dollar + 1, (int)n_dollar));
    GCHK(hipDeviceSynchronize());
    P.release(d_tmp); P.release(d_tmp3); P.release(d_tot);
  }
  P.release(d_cntb); P.release(d_nd);
  // host side: blocks, '$' list, acc.  (defer_host_blocks: the caller brings the rank blocks down beside the k-mer table's
  // build, svdss_index_fetch_blocks -- 3.1 GB into ordinary memory are 0.7 s of a restore at GRCh38 lengths)
  try {
    if (!defer_host_blocks) ix->blocks.resize((size_t)(4 * nb));
    ix->dollar.resize((size_t)n_dollar);
  } catch (...) { return SVDSS_ENOMEM; }
  if (!defer_host_blocks) GCHK(hipMemcpy(ix->blocks.data(), d_blocks, (size_t)nb * 64, hipMemcpyDeviceToHost));
  GCHK(hipMemcpy(ix->dollar.data(), d_dollar + n_dollar + 1, (size_t)n_dollar * 8, hipMemcpyDeviceToHost));
  int64_t* d_dollar_final;
  PALLOC(P, d_dollar_final, (size_t)(n_dollar + 1) * 8);
  GCHK(hipMemcpy(d_dollar_final, d_dollar + n_dollar + 1, (size_t)n_dollar * 8, hipMemcpyDeviceToDevice));
  P.release(d_dollar);
  ix->n = n;
  ix->n_contigs = n_contigs;
  ix->sa_wide = wide;
  ix->text.clear(); ix->sa32.clear(); ix->sa64.clear();
  const int64_t n_N = n - n_dollar - totals[0] - totals[1] - totals[2] - totals[3];
  ix->acc[0] = 0;
  ix->acc[1] = n_dollar;
  for (int c = 0; c < 4; ++c) ix->acc[c + 2] = ix->acc[c + 1] + totals[c];
  ix->acc[6] = ix->acc[5] + n_N;
  ix->device = device;
  ix->d_text = P.take(d_textalloc);
  ix->d_sa = P.take(d_sa);
  ix->d_blocks = P.take(d_blocks);
  ix->d_dollar = P.take(d_dollar_final);
  ix->d_table = d_arena ? P.take(d_arena) : nullptr;   // (not a table yet: table_k = 0; build_table fills it if it is large enough)
  ix->d_table_cap = d_arena ? table_bytes : 0;
  ix->table_k = 0;
  mark("counters, blocks to the host");
  return 0;
}

// the rank blocks of an index built with defer_host_blocks -> host vector
int svdss_index_fetch_blocks(svdss_index* ix) {
  if (!ix || ix->device < 0 || !ix->d_blocks) return SVDSS_EINVAL;
  const int64_t nb = ix->n / SVDSS_BLOCK_SYMS + 1;
  if ((int64_t)ix->blocks.size() == 4 * nb) return SVDSS_OK;
  if (hipSetDevice(ix->device) != hipSuccess) { (void)hipGetLastError(); return SVDSS_EHIP; }
  try { ix->blocks.resize((size_t)(4 * nb)); } catch (...) { return SVDSS_ENOMEM; }
  // Plain blocking copies into ordinary memory, a quarter of the blocks per thread (the copy of ordinary memory is staged
  // by the calling thread: one thread 0.4 s, and the vector's zero fill before it 0.3 s -- gone with the allocator of
  // index_host.h).  Two asynchronous versions beside the k-mer table's build took the copy off the critical path
  // altogether; they were dropped when `SVDSS search` seemed to stream slower behind them (profiles/r05z_restore_ab.txt)
  // -- which turned out to be the driver clearing the sort's released buffers beside the stream (profiles/r05z_e2e_lib_ab.txt,
  // sections 3-4: fixed by lending the table's memory to the sort); this version is simple and costs 0.2 s.
  const size_t total = (size_t)nb * 64;
  const int T = 4;
  std::atomic<int> bad(0);
  uint8_t* dst = (uint8_t*)ix->blocks.data();
  const int dev = ix->device;
  const void* src = ix->d_blocks;
  auto part = [&](int t) {
    if (hipSetDevice(dev) != hipSuccess) { bad = 1; return; }
    const size_t a = total * (size_t)t / T & ~(size_t)63, b = t + 1 == T ? total : (total * (size_t)(t + 1) / T & ~(size_t)63);
    if (b > a && hipMemcpy(dst + a, (const uint8_t*)src + a, b - a, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(part, t);
  part(0);
  for (std::thread& x : th) x.join();
  if (bad.load()) { (void)hipGetLastError(); ix->blocks.clear(); return SVDSS_EHIP; }
  return SVDSS_OK;
}

// the text of a device-built index -> host vector (no-op when it is there already)
static std::mutex g_fetch_m;   // guards the host-side vectors of a shared handle while they are brought down

int svdss_index_fetch_text(svdss_index* ix) {
  if (!ix) return SVDSS_EINVAL;
  std::lock_guard<std::mutex> fetch_lk(g_fetch_m);
  if ((int64_t)ix->text.size() == ix->n) return SVDSS_OK;
  if (ix->device < 0 || !ix->d_text) return SVDSS_ENODEV;
  if (hipSetDevice(ix->device) != hipSuccess) { (void)hipGetLastError(); return SVDSS_EHIP; }
  try { ix->text.resize((size_t)ix->n); } catch (...) { return SVDSS_ENOMEM; }
  if (hipMemcpy(ix->text.data(), (const uint8_t*)ix->d_text + 64, (size_t)ix->n, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    ix->text.clear();
    return SVDSS_EHIP;
  }
  return SVDSS_OK;
}

// text and suffix array of a device-built index -> host vectors (no-op when they are there already)
int svdss_index_fetch_host(svdss_index* ix) {
  if (!ix) return SVDSS_EINVAL;
  // several threads may come here with the SAME handle (svdss_index_replicate of `--gpus N`, one thread per replica): the
  // vectors are resized and filled by one of them, the others find them complete (the size test alone would let a
  // second thread through between the resize and the end of the copy)
  std::lock_guard<std::mutex> fetch_lk(g_fetch_m);
  const bool have_sa = ix->sa_wide ? (int64_t)ix->sa64.size() == ix->n : (int64_t)ix->sa32.size() == ix->n;
  if ((int64_t)ix->text.size() == ix->n && have_sa) return SVDSS_OK;
  if (ix->device < 0 || !ix->d_text || !ix->d_sa) return SVDSS_ENODEV;
  if (hipSetDevice(ix->device) != hipSuccess) { (void)hipGetLastError(); return SVDSS_EHIP; }
  try {
    ix->text.resize((size_t)ix->n);
    if (ix->sa_wide) ix->sa64.resize((size_t)ix->n); else ix->sa32.resize((size_t)ix->n);
  } catch (...) { return SVDSS_ENOMEM; }
  if (hipMemcpy(ix->text.data(), (const uint8_t*)ix->d_text + 64, (size_t)ix->n, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(ix->sa_wide ? (void*)ix->sa64.data() : (void*)ix->sa32.data(), ix->d_sa,
                (size_t)ix->n * (ix->sa_wide ? 8 : 4), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return SVDSS_EHIP;
  }
  return SVDSS_OK;
}
