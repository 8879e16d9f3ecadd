// index_verify.hip -- a resident index checked against its own text by direct comparison (svdss_index_verify_device).
//
// The index that `SVDSS search` walks stands where the one rb3_fmi_restore returns stands in the reference
// (/root/reference/ping_pong.cpp:245): every interval size the search looks at (ping_pong.cpp:15-22,31-37) is a
// statement about the suffix array of  contig $ revcomp $ ...  This file shares nothing with the builders
// (index_build.cpp, index_gpu.hip -- key sort, prefix doubling, pieces): it only knows what a suffix array and a BWT
// ARE, and checks it row by row, so that a checker built from this index's BWT is not common-mode with the builder:
//   rows     SA[i] < n;  text[SA[i]..) < text[SA[i+1]..) as strings (a suffix that runs off the text sorts before its
//            extensions) -- strictly increasing rows of n in-range entries are a permutation, hence THE suffix array;
//            BWT[i] == text[SA[i] - 1] (cyclic: the row of suffix 0 holds the last symbol, a '$');
//   blocks   every block's four counters continue the previous block's by the symbols its bit planes hold; the
//            last block closes on acc[]; the '$' rows are exactly the sorted list `dollar`;
//   text     the text's symbol histogram is acc[].
// One lane per row (two random lines of text per row, more in repeats), grid-stride; GRCh38 lengths (6.18e9 rows):
// about a second.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/svdss_hip.h"
#include "fmd_layout.h"
#include "index_host.h"

namespace {

typedef unsigned long long ull;

struct VerifyOut {
  ull rows, bad_order, bad_bwt, bad_range, bad_block, bad_dollar, first_bad, max_lcp, hist[6];
};

template <class I>
__global__ void __launch_bounds__(256) verify_rows_kernel(SvdssDevIndex ix, int64_t stride, VerifyOut* out) {
  const I* sa = (const I*)ix.sa;
  const int64_t n = ix.n;
  ull rows = 0, bad_order = 0, bad_bwt = 0, bad_range = 0, max_lcp = 0, first_bad = ~0ull;
  const int64_t n_samples = (n + stride - 1) / stride;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_samples; s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = s * stride;
    ++rows;
    const ull a = (ull)sa[i];
    bool bad = false;
    if (a >= (ull)n) { ++bad_range; bad = true; }
    else {
      const int want = ix.text[a ? (int64_t)a - 1 : n - 1];
      if (svdss_bwt_at(ix, i) != want) { ++bad_bwt; bad = true; }
      if (i + 1 < n) {
        const ull b = (ull)sa[i + 1];
        if (b >= (ull)n) { ++bad_range; bad = true; }
        else if (a == b) { ++bad_order; bad = true; }
        else {
          // first difference of text[a..) and text[b..): 8 symbols per step while both stay inside the text
          const uint8_t* ta = ix.text + a;
          const uint8_t* tb = ix.text + b;
          const int64_t room = n - (int64_t)(a > b ? a : b);   // symbols the shorter suffix has
          int64_t k = 0;
          bool decided = false, less = false;
          while (k + 8 <= room) {
            uint64_t x, y;
            __builtin_memcpy(&x, ta + k, 8);
            __builtin_memcpy(&y, tb + k, 8);
            if (x != y) {
              const int d = __builtin_ctzll(x ^ y) >> 3;
              less = ta[k + d] < tb[k + d];
              k += d;
              decided = true;
              break;
            }
            k += 8;
          }
          if (!decided) {
            while (k < room && ta[k] == tb[k]) ++k;
            if (k < room) less = ta[k] < tb[k];
            else less = a > b;          // the shorter suffix is a prefix of the longer one: it comes first
          }
          if ((ull)k > max_lcp) max_lcp = (ull)k;
          if (!less) { ++bad_order; bad = true; }
        }
      }
    }
    if (bad && (ull)i < first_bad) first_bad = (ull)i;
  }
  if (rows) atomicAdd(&out->rows, rows);
  if (bad_order) atomicAdd(&out->bad_order, bad_order);
  if (bad_bwt) atomicAdd(&out->bad_bwt, bad_bwt);
  if (bad_range) atomicAdd(&out->bad_range, bad_range);
  if (max_lcp) atomicMax(&out->max_lcp, max_lcp);
  if (first_bad != ~0ull) atomicMin(&out->first_bad, first_bad);
}

// one thread per block of 128 rows: counters continue, '$' rows are the listed ones
__global__ void __launch_bounds__(256) verify_blocks_kernel(SvdssDevIndex ix, VerifyOut* out) {
  const int64_t nb = ix.n / SVDSS_BLOCK_SYMS + 1;
  ull bad_block = 0, bad_dollar = 0;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (int64_t)gridDim.x * blockDim.x) {
    const svdss_u4* q = ix.blocks + 4 * b;
    const int64_t valid = ix.n - b * SVDSS_BLOCK_SYMS;   // rows of this block that exist (>= 128 but for the last)
    uint32_t cnt[4] = {0, 0, 0, 0};
    int64_t n_dollar_here = 0;
    for (int j = 0; j < 4; ++j) {
      const uint32_t live = svdss_lowmask((int)(valid - 32 * j > 32 ? 32 : valid - 32 * j));
      const uint32_t p0 = q[j].y, p1 = q[j].z, p2 = q[j].w;
      cnt[0] += svdss_popc(~p0 & ~p1 & ~p2 & live);
      cnt[1] += svdss_popc(p0 & ~p1 & ~p2 & live);
      cnt[2] += svdss_popc(~p0 & p1 & ~p2 & live);
      cnt[3] += svdss_popc(p0 & p1 & ~p2 & live);
      uint32_t dl = p2 & ~p0 & live;
      if ((p2 & p1) & live) ++bad_block;              // special symbols keep p1 clear
      if ((p0 | p1 | p2) & ~live) ++bad_block;        // nothing beyond the last row
      while (dl) {                                     // every '$' row is in the list
        const int bit = __builtin_ctz(dl);
        dl &= dl - 1;
        const int64_t row = b * SVDSS_BLOCK_SYMS + 32 * j + bit;
        const int64_t r = svdss_rank_dollar(ix, row);
        if (r >= ix.n_dollar || ix.dollar[r] != row) ++bad_dollar;
        ++n_dollar_here;
      }
    }
    {   // and the list names no other row of this block
      const int64_t lo = svdss_rank_dollar(ix, b * SVDSS_BLOCK_SYMS), hi = svdss_rank_dollar(ix, (b + 1) * SVDSS_BLOCK_SYMS);
      if (hi - lo != n_dollar_here) ++bad_dollar;
    }
    const uint32_t have[4] = {q[0].x, q[1].x, q[2].x, q[3].x};
    if (b == 0) {
      for (int c = 0; c < 4; ++c) if (have[c] != 0) ++bad_block;
    }
    if (b + 1 < nb) {
      const svdss_u4* nx = q + 4;
      const uint32_t next[4] = {nx[0].x, nx[1].x, nx[2].x, nx[3].x};
      for (int c = 0; c < 4; ++c) if (next[c] != have[c] + cnt[c]) ++bad_block;
    } else {
      for (int c = 0; c < 4; ++c)
        if ((int64_t)have[c] + cnt[c] != ix.acc[c + 2] - ix.acc[c + 1]) ++bad_block;
    }
  }
  if (bad_block) atomicAdd(&out->bad_block, bad_block);
  if (bad_dollar) atomicAdd(&out->bad_dollar, bad_dollar);
}

__global__ void __launch_bounds__(256) verify_text_kernel(SvdssDevIndex ix, VerifyOut* out) {
  __shared__ ull sh[6];
  if (threadIdx.x < 6) sh[threadIdx.x] = 0;
  __syncthreads();
  ull h[6] = {0, 0, 0, 0, 0, 0};
  const int64_t n16 = ix.n / 16;
  const uint8_t* t = ix.text;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= n16; w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = w * 16, e = s + 16 < ix.n ? s + 16 : ix.n;
    for (int64_t i = s; i < e; ++i) {
      const int c = t[i];
      if (c < 6) ++h[c]; else ++h[5], atomicAdd(&out->bad_block, 1ull);   // not an nt6 symbol
    }
  }
  for (int c = 0; c < 6; ++c) if (h[c]) atomicAdd(&sh[c], h[c]);
  __syncthreads();
  if (threadIdx.x < 6 && sh[threadIdx.x]) atomicAdd(&out->hist[threadIdx.x], sh[threadIdx.x]);
}

}  // namespace

extern "C" int svdss_index_verify_device(const svdss_index_t* ixp, int64_t stride, int64_t out[8]) {
  const svdss_index* ix = ixp;
  if (!ix || !out || stride < 1) return SVDSS_EINVAL;
  if (ix->device < 0 || !ix->d_sa || !ix->d_text || !ix->d_blocks) return SVDSS_EINVAL;   // resident indexes only
  if (hipSetDevice(ix->device) != hipSuccess) return SVDSS_EHIP;
  SvdssDevIndex v;
  v.blocks = (const svdss_u4*)ix->d_blocks;
  v.dollar = (const int64_t*)ix->d_dollar;
  v.n = ix->n;
  v.n_dollar = (int32_t)ix->dollar.size();
  v.k = 0; v.bs_after = 0; v.pad_ = 0;
  memcpy(v.acc, ix->acc, sizeof v.acc);
  v.text = (const uint8_t*)ix->d_text + 64;
  v.sa = ix->d_sa;
  v.table = nullptr;
  VerifyOut* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(VerifyOut)) != hipSuccess) return SVDSS_ENOMEM;
  VerifyOut h;
  memset(&h, 0, sizeof h);
  h.first_bad = ~0ull;
  int rc = SVDSS_OK;
  if (hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) rc = SVDSS_EHIP;
  if (rc == SVDSS_OK) {
    const int grid = 256 * 16;
    if (ix->sa_wide) hipLaunchKernelGGL(verify_rows_kernel<uint64_t>, dim3(grid), dim3(256), 0, 0, v, stride, d);
    else hipLaunchKernelGGL(verify_rows_kernel<uint32_t>, dim3(grid), dim3(256), 0, 0, v, stride, d);
    hipLaunchKernelGGL(verify_blocks_kernel, dim3(grid), dim3(256), 0, 0, v, d);
    hipLaunchKernelGGL(verify_text_kernel, dim3(grid), dim3(256), 0, 0, v, d);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) rc = SVDSS_EHIP;
  }
  (void)hipFree(d);
  if (rc != SVDSS_OK) { (void)hipGetLastError(); return rc; }
  ull bad_hist = 0;
  for (int c = 0; c < 6; ++c) bad_hist += (int64_t)h.hist[c] != ix->acc[c + 1] - ix->acc[c];
  out[0] = (int64_t)h.rows;
  out[1] = (int64_t)h.bad_order;
  out[2] = (int64_t)h.bad_bwt;
  out[3] = (int64_t)h.bad_range;
  out[4] = (int64_t)(h.bad_block + bad_hist);
  out[5] = (int64_t)h.bad_dollar;
  out[6] = h.first_bad == ~0ull ? -1 : (int64_t)h.first_bad;
  out[7] = (int64_t)h.max_lcp;
  return SVDSS_OK;
}
