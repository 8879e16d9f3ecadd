// call_dp.hip -- gfx950 kernels + C-ABI for the two third-party DP seams of `SVDSS call`:
//   * ksw_extd2_sse(..., w=-1, zdrop=-1, end_bonus=-1, flag=0)  /root/reference/caller.cpp:348-349
//     (consensus -> reference window, global, dual affine gap, full matrix, CIGAR)
//   * rapidfuzz::fuzz::ratio(a, b)                              /root/reference/caller.cpp:456,458
//
// Both are integer DPs whose anti-diagonals are independent: one workgroup per
// pair sweeps the anti-diagonals, lanes stride over the cells of a diagonal,
// the previous two diagonals stay resident (LDS when they fit, HBM otherwise).
// Not HBM- and not MFMA-bound: the figure of merit is cell updates/s.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/svdss_hip.h"

extern thread_local std::string g_svdss_hip_err;

#define HIPCHK2(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      g_svdss_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);        \
      return (e_ == hipErrorOutOfMemory) ? SVDSS_ENOMEM : SVDSS_EHIP;             \
    }                                                                             \
  } while (0)

#define DP_NEG (-0x20000000)
#define DP_THREADS 256

struct GapModel { int q, e, q2, e2; };

__device__ __forceinline__ int dp_gap(int l, const GapModel& g) {
  const int a = g.q + l * g.e, b = g.q2 + l * g.e2;
  return a < b ? a : b;
}

struct AlnPair {
  int64_t q_off, t_off;   // into the concatenated query / target symbol buffers
  int64_t ws_off;         // int32 workspace: 11 arrays of (tl + 1)
  int64_t dir_off;        // (tl+ql)*tl direction bytes, diagonal-major: cell (i, j) at (i+j)*tl + i
  int64_t cig_off;        // uint32 ops in backtrack order, capacity tl + ql + 2
  int32_t ql, tl;
};

// barrier that orders LDS traffic only (the direction bytes stream out to HBM on every diagonal; waiting for them
// at each of the ~2000 barriers of a pair would cost more than the diagonal itself)
__device__ __forceinline__ void dp_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One workgroup per pair.  Cell (i, j): i indexes the target, j the query, r = i + j.
// Recurrences and tie rules are those of ksw2's extd2 (left-aligned), see the oracle
// (oracle/svdss_oracle_call.c, orc_ksw_extd2_global) which this kernel must match bit for bit.
// The rotating diagonals (H of r-1 and r-2, the four gap states of r-1: 11 arrays of tl+1) live in LDS when
// they fit (LDS = true, 44 bytes per target symbol), else in the HBM workspace.
template <bool LDS>
__global__ void __launch_bounds__(DP_THREADS) align_global_kernel(
    const AlnPair* pairs, const uint8_t* qsyms, const uint8_t* tsyms, int m, const int8_t* mat_g,
    GapModel gm, int32_t* ws, uint8_t* dirs, uint32_t* cigars, int32_t* scores, int32_t* n_cigar) {
  extern __shared__ int32_t dp_lds[];
  __shared__ int8_t mat[64];
  const AlnPair P = pairs[blockIdx.x];
  const int ql = P.ql, tl = P.tl;
  if (threadIdx.x < m * m && threadIdx.x < 64) mat[threadIdx.x] = mat_g[threadIdx.x];
  if (ql <= 0 || tl <= 0) {   // ksw2 returns before touching ez: score 0, no CIGAR
    if (threadIdx.x == 0) { scores[blockIdx.x] = 0; n_cigar[blockIdx.x] = 0; }
    return;
  }
  const int stride = tl + 1;
  int32_t* buf = LDS ? dp_lds : ws + P.ws_off;
  // the two sequences sit behind the diagonals in LDS (a global byte load per cell would put ~1 us of latency on
  // every diagonal)
  const uint8_t* q = qsyms + P.q_off;
  const uint8_t* t = tsyms + P.t_off;
  if (LDS) {
    uint8_t* sq = (uint8_t*)(dp_lds + 11 * stride);
    uint8_t* st_ = sq + ql;
    for (int x = threadIdx.x; x < ql; x += DP_THREADS) sq[x] = q[x];
    for (int x = threadIdx.x; x < tl; x += DP_THREADS) st_[x] = t[x];
    q = sq;
    t = st_;
  }
  // array k of the 11 at buf + k * stride: H x3 (rotating: needs r-1 and r-2), then E, F, E2, F2 x2 (r-1)
  uint8_t* dir = dirs + P.dir_off;
  __syncthreads();
  const int n_diag = tl + ql - 1;
  for (int r = 0; r < n_diag; ++r) {
    const int32_t* Hm1 = buf + ((r + 2) % 3) * stride;
    const int32_t* Hm2 = buf + ((r + 1) % 3) * stride;
    int32_t* Hc = buf + (r % 3) * stride;
    const int cur = r & 1, prv = cur ^ 1;
    const int32_t *Ep_ = buf + (3 + prv) * stride, *Fp_ = buf + (5 + prv) * stride;
    const int32_t *E2p_ = buf + (7 + prv) * stride, *F2p_ = buf + (9 + prv) * stride;
    int32_t *Ec = buf + (3 + cur) * stride, *Fc = buf + (5 + cur) * stride;
    int32_t *E2c = buf + (7 + cur) * stride, *F2c = buf + (9 + cur) * stride;
    const int ilo = r - (ql - 1) > 0 ? r - (ql - 1) : 0;
    const int ihi = r < tl - 1 ? r : tl - 1;
    for (int i = ilo + (int)threadIdx.x; i <= ihi; i += DP_THREADS) {
      const int j = r - i;
      int32_t hdiag, hup, hleft, Ep, E2p, Fp, F2p;
      if (i > 0 && j > 0) hdiag = Hm2[i - 1];
      else if (i == 0) hdiag = j == 0 ? 0 : -dp_gap(j, gm);
      else hdiag = -dp_gap(i, gm);
      if (i > 0) { hup = Hm1[i - 1]; Ep = Ep_[i - 1]; E2p = E2p_[i - 1]; }
      else { hup = -dp_gap(j + 1, gm); Ep = DP_NEG; E2p = DP_NEG; }
      if (j > 0) { hleft = Hm1[i]; Fp = Fp_[i]; F2p = F2p_[i]; }
      else { hleft = -dp_gap(i + 1, gm); Fp = DP_NEG; F2p = DP_NEG; }
      const int32_t Ein = (hup - gm.q > Ep ? hup - gm.q : Ep) - gm.e;
      const int32_t E2in = (hup - gm.q2 > E2p ? hup - gm.q2 : E2p) - gm.e2;
      const int32_t Fin = (hleft - gm.q > Fp ? hleft - gm.q : Fp) - gm.e;
      const int32_t F2in = (hleft - gm.q2 > F2p ? hleft - gm.q2 : F2p) - gm.e2;
      int32_t z = hdiag + mat[t[i] * m + q[j]];
      uint32_t d = 0;
      if (Ein > z) { d = 1; z = Ein; }
      if (Fin > z) { d = 2; z = Fin; }
      if (E2in > z) { d = 3; z = E2in; }
      if (F2in > z) { d = 4; z = F2in; }
      if (Ein > z - gm.q) d |= 0x08;
      if (Fin > z - gm.q) d |= 0x10;
      if (E2in > z - gm.q2) d |= 0x20;
      if (F2in > z - gm.q2) d |= 0x40;
      dir[(int64_t)r * tl + i] = (uint8_t)d;   // diagonal-major: the lanes of a diagonal write consecutive bytes
      Hc[i] = z;
      Ec[i] = Ein; E2c[i] = E2in;
      Fc[i] = Fin; F2c[i] = F2in;
    }
    if (LDS) dp_lds_barrier(); else __syncthreads();
  }
  __syncthreads();   // the direction bytes are in HBM
  if (threadIdx.x < 64) {
    // ksw_backtrack from (tl-1, ql-1) by the first wavefront; ops are left in backtrack order, the host reverses
    // them.  A register window holds the direction bytes of 64 rows x 4 columns along the current diagonal (one HBM
    // latency per ~60 steps of a mostly diagonal path instead of one per step); runs are merged in registers.
    const int lane = threadIdx.x;
    if (lane == 0) scores[blockIdx.x] = buf[((n_diag - 1) % 3) * stride + tl - 1];
    uint32_t* cg = cigars + P.cig_off;
    int n = 0, i = tl - 1, j = ql - 1, state = 0;
    uint32_t cur_op = 0xffffffffu, cur_len = 0;
    auto push = [&](uint32_t op, uint32_t len) {
      if (op == cur_op) { cur_len += len; return; }
      if (cur_op != 0xffffffffu) { if (lane == 0) cg[n] = (cur_len << 4) | cur_op; ++n; }
      cur_op = op; cur_len = len;
    };
    int i0 = -(1 << 28), j0 = 0;
    uint32_t win = 0;
    while (i >= 0 && j >= 0) {
      int k = i0 - i, d = j - (j0 - k);
      if (k < 0 || k > 63 || d < 0 || d > 3) {
        i0 = i; j0 = j - 1; k = 0; d = 1;       // columns j0-k .. j0-k+3 of row i0-k (room for two D and one I moves)
        const int ii = i0 - lane;
        const int64_t jj = (int64_t)j0 - lane;
        win = 0;
        if (ii >= 0) {
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int64_t c = jj + x;
            if (c >= 0 && c < ql) win |= (uint32_t)dir[(int64_t)(ii + c) * tl + ii] << (8 * x);
          }
        }
        asm volatile("" : "+v"(win));   // wait for the window here, not at the join below
      }
      const uint32_t tmp = ((uint32_t)__builtin_amdgcn_readlane((int)win, k) >> (8 * d)) & 0xffu;
      if (state == 0) state = tmp & 7;
      else if (!((tmp >> (state + 2)) & 1)) state = 0;
      if (state == 0) state = tmp & 7;
      if (state == 0) { push(0, 1); --i; --j; }
      else if (state == 1 || state == 3) { push(2, 1); --i; }
      else { push(1, 1); --j; }
    }
    if (i >= 0) push(2, (uint32_t)(i + 1));
    if (j >= 0) push(1, (uint32_t)(j + 1));
    if (cur_op != 0xffffffffu) { if (lane == 0) cg[n] = (cur_len << 4) | cur_op; ++n; }
    if (lane == 0) n_cigar[blockIdx.x] = n;
  }
}

struct LcsPair {
  int64_t a_off, b_off, ws_off;   // ws: 3 arrays of (la + 1) int32
  int32_t la, lb;
};

// LCS length by anti-diagonals, then rapidfuzz::fuzz::ratio's arithmetic.
// (LDS = true: the three rotating diagonals and both strings live in LDS)
template <bool LDS>
__global__ void __launch_bounds__(DP_THREADS) lcs_ratio_kernel(const LcsPair* pairs, const uint8_t* as,
                                                              const uint8_t* bs, int32_t* ws,
                                                              int64_t* lcs_out, double* ratio_out) {
  extern __shared__ int32_t dp_lds[];
  const LcsPair P = pairs[blockIdx.x];
  const int la = P.la, lb = P.lb;
  int64_t lcs = 0;
  if (la > 0 && lb > 0) {
    const uint8_t* a = as + P.a_off;
    const uint8_t* b = bs + P.b_off;
    const int stride = la + 1;
    int32_t* buf = LDS ? dp_lds : ws + P.ws_off;
    if (LDS) {
      uint8_t* sa = (uint8_t*)(dp_lds + 3 * stride);
      uint8_t* sb = sa + la;
      for (int x = threadIdx.x; x < la; x += DP_THREADS) sa[x] = a[x];
      for (int x = threadIdx.x; x < lb; x += DP_THREADS) sb[x] = b[x];
      a = sa;
      b = sb;
      __syncthreads();
    }
    const int n_diag = la + lb - 1;
    for (int r = 0; r < n_diag; ++r) {
      const int32_t* Lm1 = buf + ((r + 2) % 3) * stride;
      const int32_t* Lm2 = buf + ((r + 1) % 3) * stride;
      int32_t* Lc = buf + (r % 3) * stride;
      const int ilo = r - (lb - 1) > 0 ? r - (lb - 1) : 0;
      const int ihi = r < la - 1 ? r : la - 1;
      for (int i = ilo + (int)threadIdx.x; i <= ihi; i += DP_THREADS) {
        const int j = r - i;
        const int32_t diag = (i > 0 && j > 0) ? Lm2[i - 1] : 0;
        const int32_t up = i > 0 ? Lm1[i - 1] : 0;
        const int32_t left = j > 0 ? Lm1[i] : 0;
        Lc[i] = a[i] == b[j] ? diag + 1 : (up > left ? up : left);
      }
      if (LDS) dp_lds_barrier(); else __syncthreads();
    }
    lcs = buf[((n_diag - 1) % 3) * stride + la - 1];
  }
  if (threadIdx.x == 0) {
    const int64_t maximum = (int64_t)la + lb;
    const int64_t dist = maximum - 2 * lcs;
    const double norm_dist = maximum ? (double)dist / (double)maximum : 0.0;
    const double norm_sim = 1.0 - norm_dist;
    lcs_out[blockIdx.x] = lcs;
    ratio_out[blockIdx.x] = norm_sim * 100.0;
  }
}

// ------------------------------------------------------------------- ABI

struct svdss_aln_batch {
  int64_t n_pairs = 0;
  int64_t total_cigar = 0;
  int64_t cells = 0;
  double kernel_ms = 0.0;
  std::vector<int32_t> scores;
  std::vector<int64_t> n_cigar;
  std::vector<uint32_t> cigar;   // per pair, forward order, concatenated
};

namespace {
struct DevMem {
  void* p = nullptr;
  ~DevMem() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) {
    HIPCHK2(hipMalloc(&p, bytes ? bytes : 16));
    return SVDSS_OK;
  }
};
}  // namespace

extern "C" int svdss_align_global_batch(const uint8_t* queries, const int64_t* q_off,
                                        const uint8_t* targets, const int64_t* t_off, int64_t n_pairs,
                                        int32_t m, const int8_t* mat, int32_t gapo, int32_t gape,
                                        int32_t gapo2, int32_t gape2, int32_t device,
                                        svdss_aln_batch_t** out) {
  if (!out || n_pairs < 0 || m < 1 || m > 8 || !mat || device < 0) return SVDSS_EINVAL;
  if (n_pairs > 0 && (!queries || !q_off || !targets || !t_off)) return SVDSS_EINVAL;
  HIPCHK2(hipSetDevice(device));
  svdss_aln_batch* b = *out ? *out : new (std::nothrow) svdss_aln_batch();
  if (!b) return SVDSS_ENOMEM;
  *out = b;
  b->n_pairs = n_pairs;
  b->total_cigar = 0;
  b->cells = 0;
  b->kernel_ms = 0.0;
  b->scores.assign((size_t)n_pairs, 0);
  b->n_cigar.assign((size_t)n_pairs, 0);
  b->cigar.clear();
  if (n_pairs == 0) return SVDSS_OK;
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int64_t ql = q_off[i + 1] - q_off[i], tl = t_off[i + 1] - t_off[i];
    if (ql < 0 || tl < 0) return SVDSS_EINVAL;
    if (ql >= (1 << 28) || tl >= (1 << 28)) return SVDSS_ERANGE;
  }
  const int64_t qtot = q_off[n_pairs], ttot = t_off[n_pairs];
  DevMem d_q, d_t, d_mat;
  int rc;
  if ((rc = d_q.alloc((size_t)qtot)) || (rc = d_t.alloc((size_t)ttot)) || (rc = d_mat.alloc(64))) return rc;
  if (qtot) HIPCHK2(hipMemcpy(d_q.p, queries + q_off[0], (size_t)(qtot - q_off[0]), hipMemcpyHostToDevice));
  if (ttot) HIPCHK2(hipMemcpy(d_t.p, targets + t_off[0], (size_t)(ttot - t_off[0]), hipMemcpyHostToDevice));
  HIPCHK2(hipMemcpy(d_mat.p, mat, (size_t)(m * m), hipMemcpyHostToDevice));
  const GapModel gm{gapo, gape, gapo2, gape2};
  hipEvent_t ev0, ev1;
  HIPCHK2(hipEventCreate(&ev0));
  HIPCHK2(hipEventCreate(&ev1));
  // chunks of pairs whose direction matrices fit the workspace budget (8 GiB; HBM has 288)
  const int64_t dir_budget = (int64_t)8 << 30;
  int64_t start = 0;
  while (start < n_pairs) {
    std::vector<AlnPair> hp;
    int64_t ws = 0, dirb = 0, cig = 0, end = start, tl_max = 0, ql_max = 0;
    while (end < n_pairs) {
      const int64_t ql = q_off[end + 1] - q_off[end], tl = t_off[end + 1] - t_off[end];
      const int64_t need = (ql + tl) * tl;   // direction bytes, one row of tl per anti-diagonal
      if (end > start && dirb + need > dir_budget) break;
      AlnPair a;
      a.q_off = q_off[end] - q_off[0];
      a.t_off = t_off[end] - t_off[0];
      a.ws_off = ws;
      a.dir_off = dirb;
      a.cig_off = cig;
      a.ql = (int32_t)ql;
      a.tl = (int32_t)tl;
      hp.push_back(a);
      ws += 11 * (tl + 1);
      if (tl > tl_max) tl_max = tl;
      if (ql > ql_max) ql_max = ql;
      dirb += need;
      cig += ql + tl + 2;
      b->cells += ql * tl;
      ++end;
    }
    const int64_t np = end - start;
    DevMem d_pairs, d_ws, d_dir, d_cig, d_sc, d_nc;
    if ((rc = d_pairs.alloc(sizeof(AlnPair) * (size_t)np)) || (rc = d_ws.alloc(sizeof(int32_t) * (size_t)ws)) ||
        (rc = d_dir.alloc((size_t)dirb)) || (rc = d_cig.alloc(sizeof(uint32_t) * (size_t)cig)) ||
        (rc = d_sc.alloc(sizeof(int32_t) * (size_t)np)) || (rc = d_nc.alloc(sizeof(int32_t) * (size_t)np)))
      return rc;
    HIPCHK2(hipMemcpy(d_pairs.p, hp.data(), sizeof(AlnPair) * (size_t)np, hipMemcpyHostToDevice));
    HIPCHK2(hipEventRecord(ev0, 0));
    const size_t lds_need = sizeof(int32_t) * 11 * (size_t)(tl_max + 1) + (size_t)ql_max + (size_t)tl_max + 16;
    if (lds_need <= 150 * 1024) {
      HIPCHK2(hipFuncSetAttribute((const void*)align_global_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
      hipLaunchKernelGGL(align_global_kernel<true>, dim3((unsigned)np), dim3(DP_THREADS), lds_need, 0,
                         (const AlnPair*)d_pairs.p, (const uint8_t*)d_q.p, (const uint8_t*)d_t.p, (int)m,
                         (const int8_t*)d_mat.p, gm, (int32_t*)d_ws.p, (uint8_t*)d_dir.p, (uint32_t*)d_cig.p,
                         (int32_t*)d_sc.p, (int32_t*)d_nc.p);
    } else {
      hipLaunchKernelGGL(align_global_kernel<false>, dim3((unsigned)np), dim3(DP_THREADS), 0, 0,
                         (const AlnPair*)d_pairs.p, (const uint8_t*)d_q.p, (const uint8_t*)d_t.p, (int)m,
                         (const int8_t*)d_mat.p, gm, (int32_t*)d_ws.p, (uint8_t*)d_dir.p, (uint32_t*)d_cig.p,
                         (int32_t*)d_sc.p, (int32_t*)d_nc.p);
    }
    HIPCHK2(hipGetLastError());
    HIPCHK2(hipEventRecord(ev1, 0));
    HIPCHK2(hipDeviceSynchronize());
    float ms = 0.f;
    HIPCHK2(hipEventElapsedTime(&ms, ev0, ev1));
    b->kernel_ms += ms;
    std::vector<int32_t> nc((size_t)np);
    std::vector<uint32_t> cg((size_t)cig);
    HIPCHK2(hipMemcpy(&b->scores[(size_t)start], d_sc.p, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost));
    HIPCHK2(hipMemcpy(nc.data(), d_nc.p, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost));
    if (cig) HIPCHK2(hipMemcpy(cg.data(), d_cig.p, sizeof(uint32_t) * (size_t)cig, hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < np; ++k) {
      b->n_cigar[(size_t)(start + k)] = nc[(size_t)k];
      const uint32_t* src = cg.data() + hp[(size_t)k].cig_off;
      for (int x = nc[(size_t)k] - 1; x >= 0; --x) b->cigar.push_back(src[x]);  // reverse (ksw_backtrack tail)
    }
    start = end;
  }
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  b->total_cigar = (int64_t)b->cigar.size();
  return SVDSS_OK;
}

extern "C" int64_t svdss_aln_batch_npairs(const svdss_aln_batch_t* b) { return b ? b->n_pairs : -1; }
extern "C" int64_t svdss_aln_batch_total_cigar(const svdss_aln_batch_t* b) { return b ? b->total_cigar : -1; }
extern "C" int64_t svdss_aln_batch_cells(const svdss_aln_batch_t* b) { return b ? b->cells : -1; }
extern "C" double svdss_aln_batch_kernel_ms(const svdss_aln_batch_t* b) { return b ? b->kernel_ms : -1.0; }

extern "C" int svdss_aln_batch_fetch(const svdss_aln_batch_t* b, int32_t* scores, int64_t* n_cigar,
                                     uint32_t* cigar) {
  if (!b) return SVDSS_EINVAL;
  if (scores) memcpy(scores, b->scores.data(), sizeof(int32_t) * b->scores.size());
  if (n_cigar) memcpy(n_cigar, b->n_cigar.data(), sizeof(int64_t) * b->n_cigar.size());
  if (cigar) memcpy(cigar, b->cigar.data(), sizeof(uint32_t) * b->cigar.size());
  return SVDSS_OK;
}

extern "C" void svdss_aln_batch_free(svdss_aln_batch_t* b) { delete b; }

extern "C" int svdss_indel_ratio_batch(const uint8_t* a, const int64_t* a_off, const uint8_t* bsy,
                                       const int64_t* b_off, int64_t n_pairs, int32_t device,
                                       double* ratio_out, int64_t* lcs_out) {
  if (n_pairs < 0 || device < 0) return SVDSS_EINVAL;
  if (n_pairs == 0) return SVDSS_OK;
  if (!a || !a_off || !bsy || !b_off || !ratio_out) return SVDSS_EINVAL;
  HIPCHK2(hipSetDevice(device));
  std::vector<LcsPair> hp((size_t)n_pairs);
  int64_t ws = 0, la_max = 0, lb_max = 0;
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int64_t la = a_off[i + 1] - a_off[i], lb = b_off[i + 1] - b_off[i];
    if (la < 0 || lb < 0) return SVDSS_EINVAL;
    if (la >= (1 << 30) || lb >= (1 << 30)) return SVDSS_ERANGE;
    hp[(size_t)i].a_off = a_off[i] - a_off[0];
    hp[(size_t)i].b_off = b_off[i] - b_off[0];
    hp[(size_t)i].ws_off = ws;
    hp[(size_t)i].la = (int32_t)la;
    hp[(size_t)i].lb = (int32_t)lb;
    ws += 3 * (la + 1);
    if (la > la_max) la_max = la;
    if (lb > lb_max) lb_max = lb;
  }
  const int64_t atot = a_off[n_pairs] - a_off[0], btot = b_off[n_pairs] - b_off[0];
  DevMem d_a, d_b, d_pairs, d_ws, d_lcs, d_ratio;
  int rc;
  if ((rc = d_a.alloc((size_t)atot)) || (rc = d_b.alloc((size_t)btot)) ||
      (rc = d_pairs.alloc(sizeof(LcsPair) * (size_t)n_pairs)) || (rc = d_ws.alloc(sizeof(int32_t) * (size_t)ws)) ||
      (rc = d_lcs.alloc(sizeof(int64_t) * (size_t)n_pairs)) || (rc = d_ratio.alloc(sizeof(double) * (size_t)n_pairs)))
    return rc;
  if (atot) HIPCHK2(hipMemcpy(d_a.p, a + a_off[0], (size_t)atot, hipMemcpyHostToDevice));
  if (btot) HIPCHK2(hipMemcpy(d_b.p, bsy + b_off[0], (size_t)btot, hipMemcpyHostToDevice));
  HIPCHK2(hipMemcpy(d_pairs.p, hp.data(), sizeof(LcsPair) * (size_t)n_pairs, hipMemcpyHostToDevice));
  const size_t lds_need = sizeof(int32_t) * 3 * (size_t)(la_max + 1) + (size_t)la_max + (size_t)lb_max + 16;
  if (lds_need <= 150 * 1024) {
    HIPCHK2(hipFuncSetAttribute((const void*)lcs_ratio_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    hipLaunchKernelGGL(lcs_ratio_kernel<true>, dim3((unsigned)n_pairs), dim3(DP_THREADS), lds_need, 0,
                       (const LcsPair*)d_pairs.p, (const uint8_t*)d_a.p, (const uint8_t*)d_b.p, (int32_t*)d_ws.p,
                       (int64_t*)d_lcs.p, (double*)d_ratio.p);
  } else {
    hipLaunchKernelGGL(lcs_ratio_kernel<false>, dim3((unsigned)n_pairs), dim3(DP_THREADS), 0, 0,
                       (const LcsPair*)d_pairs.p, (const uint8_t*)d_a.p, (const uint8_t*)d_b.p, (int32_t*)d_ws.p,
                       (int64_t*)d_lcs.p, (double*)d_ratio.p);
  }
  HIPCHK2(hipGetLastError());
  HIPCHK2(hipDeviceSynchronize());
  HIPCHK2(hipMemcpy(ratio_out, d_ratio.p, sizeof(double) * (size_t)n_pairs, hipMemcpyDeviceToHost));
  if (lcs_out) HIPCHK2(hipMemcpy(lcs_out, d_lcs.p, sizeof(int64_t) * (size_t)n_pairs, hipMemcpyDeviceToHost));
  return SVDSS_OK;
}
