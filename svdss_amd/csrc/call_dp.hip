// call_dp.hip -- gfx950 kernels + C-ABI for the two third-party DP seams of `SVDSS call`:
//   * ksw_extd2_sse(..., w=-1, zdrop=-1, end_bonus=-1, flag=0)  /root/reference/caller.cpp:348-349
//     (consensus -> reference window, global, dual affine gap, full matrix, CIGAR)
//   * rapidfuzz::fuzz::ratio(a, b)                              /root/reference/caller.cpp:456,458
//
// Both are integer DPs whose anti-diagonals are independent.
//   realignment: one WAVEFRONT per pair, no LDS and no barrier.  The target is cut into stripes of 64 rows, lane l
//     owns row 64 s + l; at step tau the lane computes column tau - l, so the wave sweeps an anti-diagonal per step:
//     H/F/F2 of (i, j-1) are the lane's own registers, H/E/E2 of (i-1, j) arrive from lane l-1 with one DPP shift
//     each, H of (i-1, j-1) is last step's shifted H.  The last row of a stripe leaves H/E/E2 per column in a
//     boundary buffer the next stripe's lane 0 reads 64 columns at a time.  ~1 wave instruction per cell; the chip is
//     filled by running thousands of pairs at once (longest first).
//   ratio: one workgroup per pair sweeps the anti-diagonals, the previous two diagonals stay resident in LDS.
// Not HBM- and not MFMA-bound: the figure of merit is cell updates/s.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/svdss_hip.h"
#include "dev_arena.h"
#include "hip_check.h"

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
  int64_t bnd_off;        // int32 workspace: (waves + 1) x 3 arrays of (ql + 64): H, E, E2 of a stripe's last row
  int64_t dir_off;        // (tl+ql)*tl direction bytes, diagonal-major: cell (i, j) at (i+j)*tl + i
  int64_t cig_off;        // uint32 ops in backtrack order, capacity tl + ql + 2
  int32_t ql, tl;
  int32_t slot;           // index of the pair in the caller's order (results are written there)
  int32_t waves;          // wavefronts that share the pair's stripes (1, or the kernel's W for the longest pairs)
};

__device__ __forceinline__ int dp_shr1(int x) {   // value of the lane below (lane 0: unchanged, it is overwritten)
  return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
}

// One wavefront per pair -- or, for the longest pairs of a batch, W wavefronts (round 5): a pair is a chain of
// (tl / 64) x (ql + 63) dependent steps, so a launch lasts as long as its longest pair (26 ms of a bench step's
// realignment, all of it the 2.6 kb x 2.6 kb pairs, while the whole batch is 8 ms of work).  The stripes of a pair only
// depend on each other through the boundary row, column by column: stripe s + 1 can run three blocks of 64 columns behind
// stripe s.  Wavefront v of the pair's workgroup takes stripes v, v + W, ...; the boundary rows go through W + 1 buffers
// in HBM (workgroup-scope release / loads: producer and consumer sit on the same CU and share its L1), progress through one LDS word per buffer, (stripe + 1) << 20 | columns done -- the stripe number in
// it, so that what an earlier stripe left in the word reads as "not yet".  Nobody waits for a stripe with a higher number,
// the lowest unfinished stripe never waits: no deadlock; a wait that does not end all the same (1 << 22 polls) raises
// *abort_flag and the host runs the batch again with one wavefront per pair.
// Cell (i, j): i indexes the target, j the query.
// Recurrences and tie rules are those of ksw2's extd2 (left-aligned), see the oracle
// (oracle/svdss_oracle_call.c, orc_ksw_extd2_global) which this kernel must match bit for bit.
template <int W>
__global__ void __launch_bounds__(64 * W) align_wave_kernel(
    const AlnPair* pairs, const uint8_t* qsyms, const uint8_t* tsyms, int m, const int8_t* mat_g,
    GapModel gm, int32_t* ws, uint8_t* dirs, uint32_t* cigars, int32_t* scores, int32_t* n_cigar, int32_t* abort_flag) {
  const AlnPair P = pairs[blockIdx.x];
  const int ql = P.ql, tl = P.tl;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Wp = W > 1 ? P.waves : 1, NB = Wp + 1;
  if (wave >= Wp) return;
  if (ql <= 0 || tl <= 0) {   // ksw2 returns before touching ez: score 0, no CIGAR
    if (lane == 0) { scores[P.slot] = 0; n_cigar[P.slot] = 0; }
    return;
  }
  __shared__ volatile int32_t prog[W + 1];
  __shared__ int32_t s_score;
  const uint8_t* q = qsyms + P.q_off;
  const uint8_t* t = tsyms + P.t_off;
  uint8_t* dir = dirs + P.dir_off;
  const int bstride = ql + 64;
  int32_t* bnd = ws + P.bnd_off;
  // boundary above row 0 ("stripe -1", buffer NB - 1): H(-1, j) = -gap(j + 1), no gap state
  {
    int32_t* b0 = bnd + (int64_t)(NB - 1) * 3 * bstride;
    for (int j = threadIdx.x; j < ql; j += 64 * Wp) { b0[j] = -dp_gap(j + 1, gm); b0[bstride + j] = DP_NEG; b0[2 * bstride + j] = DP_NEG; }
    if ((int)threadIdx.x < NB) prog[threadIdx.x] = (int)threadIdx.x == NB - 1 ? ql : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
  __syncthreads();
  const int n_stripes = (tl + 63) >> 6;
  int32_t final_score = 0;
  bool aborted = false;
  for (int s = wave; s < n_stripes && !aborted; s += Wp) {
    const int in_slot = (s + NB - 1) % NB, out_slot = s % NB;
    const int32_t* bin = bnd + (int64_t)in_slot * 3 * bstride;
    int32_t* bout = bnd + (int64_t)out_slot * 3 * bstride;
    if (W > 1 && lane == 0) prog[out_slot] = (s + 1) << 20;
    const int i = (s << 6) + lane;
    const bool row_ok = i < tl;
    const int rows = tl - (s << 6) < 64 ? tl - (s << 6) : 64;
    const bool writes_bnd = lane == 63 && s + 1 < n_stripes;
    // substitution scores of this row's target symbol, one byte per query symbol (m <= 8)
    uint64_t rowbits = 0;
    {
      const int ti = row_ok ? t[i] : 0;
      for (int k = 0; k < m; ++k) rowbits |= (uint64_t)(uint8_t)mat_g[ti * m + k] << (8 * k);
    }
    int32_t Hl = -dp_gap(i + 1, gm), Fl = DP_NEG, F2l = DP_NEG;          // (i, j-1): the column left of j = 0
    int32_t Hout = 0, Eout = DP_NEG, E2out = DP_NEG;                    // what the lane above reads next step
    int32_t Hd = 0;                                                     // (i-1, j-1)
    int qcur = 0;
    const int n_steps = ql + rows - 1;
    // the next 64 columns of the query and of the boundary row, one per lane, fetched a block ahead
    int32_t nH = 0, nE = 0, nE2 = 0;
    int nq = 0;
    auto fetch = [&](int tau0) {
      const int jj = tau0 + lane;
      const bool in = jj < ql;
      nq = in ? q[jj] : 0;
      if (W > 1 && Wp > 1) {
        // columns tau0 .. tau0 + 63 of the stripe above: wait until its wavefront has them in memory
        const int need = tau0 + 64 < ql ? tau0 + 64 : ql;
        int polls = 0;
        for (;;) {
          const int32_t v = prog[in_slot];
          if ((v >> 20) == s && (v & 0xFFFFF) >= need) break;
          if (++polls > (1 << 22)) { aborted = true; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        if (aborted) { nH = 0; nE = 0; nE2 = 0; return; }
        // acquire side of the producer's workgroup-scope release: the boundary row is read after the progress word
        // whatever the CU mode (relaxed loads alone are only ordered while both waves share one L1)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        nH = in ? __hip_atomic_load(&bin[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
        nE = in ? __hip_atomic_load(&bin[bstride + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
        nE2 = in ? __hip_atomic_load(&bin[2 * bstride + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
      } else {
        nH = in ? bin[jj] : 0; nE = in ? bin[bstride + jj] : 0; nE2 = in ? bin[2 * bstride + jj] : 0;
      }
    };
    fetch(0);
    uint8_t* dcell = dir + (int64_t)(s << 6) * tl + i;   // cell (i, j) at (i + j) * tl + i: + tl per step
    for (int tau0 = 0; tau0 < n_steps && !aborted; tau0 += 64) {
      // wait for the block here, once, so that the 64 steps below never wait on memory (their direction-byte stores
      // stay in flight: vmcnt counts stores too)
      asm volatile("" : "+v"(nq), "+v"(nH), "+v"(nE), "+v"(nE2));
      const int qblk = nq;
      const int32_t bH = nH, bE = nE, bE2 = nE2;
      if (tau0 + 64 < n_steps) fetch(tau0 + 64);
      const int kmax = n_steps - tau0 < 64 ? n_steps - tau0 : 64;
      for (int k = 0; k < kmax; ++k) {
        const int tau = tau0 + k;
        const int qs = __builtin_amdgcn_readlane(qblk, k);
        const int32_t hb = __builtin_amdgcn_readlane(bH, k), eb = __builtin_amdgcn_readlane(bE, k),
                      e2b = __builtin_amdgcn_readlane(bE2, k);
        qcur = dp_shr1(qcur);
        int32_t Hu = dp_shr1(Hout), Eu = dp_shr1(Eout), E2u = dp_shr1(E2out);
        if (lane == 0) { qcur = qs; Hu = hb; Eu = eb; E2u = e2b; }
        const int j = tau - lane;
        if (row_ok && j >= 0 && j < ql) {
          const int32_t hdiag = j == 0 ? (i == 0 ? 0 : -dp_gap(i, gm)) : Hd;
          const int32_t Ein = (Hu - gm.q > Eu ? Hu - gm.q : Eu) - gm.e;
          const int32_t E2in = (Hu - gm.q2 > E2u ? Hu - gm.q2 : E2u) - gm.e2;
          const int32_t Fin = (Hl - gm.q > Fl ? Hl - gm.q : Fl) - gm.e;
          const int32_t F2in = (Hl - gm.q2 > F2l ? Hl - gm.q2 : F2l) - gm.e2;
          int32_t z = hdiag + (int32_t)(int8_t)(rowbits >> (8 * qcur));
          uint32_t d = 0;
          if (Ein > z) { d = 1; z = Ein; }
          if (Fin > z) { d = 2; z = Fin; }
          if (E2in > z) { d = 3; z = E2in; }
          if (F2in > z) { d = 4; z = F2in; }
          if (Ein > z - gm.q) d |= 0x08;
          if (Fin > z - gm.q) d |= 0x10;
          if (E2in > z - gm.q2) d |= 0x20;
          if (F2in > z - gm.q2) d |= 0x40;
          *dcell = (uint8_t)d;   // diagonal-major: the lanes of a step write consecutive bytes
          Hout = z; Eout = Ein; E2out = E2in;
          Hl = z; Fl = Fin; F2l = F2in;
          if (writes_bnd) { bout[j] = z; bout[bstride + j] = Ein; bout[2 * bstride + j] = E2in; }
          if (i == tl - 1 && j == ql - 1) final_score = z;
        }
        Hd = Hu;
        dcell += tl;
      }
      if (W > 1 && Wp > 1 && s + 1 < n_stripes) {
        // lane 63 has finished columns 0 .. tau0 + kmax - 64 of the boundary row: in memory, then announced
        // (workgroup scope: the stores have reached the L2 of this XCD, which the reader's CU -- the same one -- reads
        // through the L1 it shares with the writer.  An agent-scope fence writes the whole L2 back on this multi-die part:
        // the first version of this kernel spent more time in it than it saved)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const int done = tau0 + kmax - 63;
        if (lane == 63) prog[out_slot] = ((s + 1) << 20) | (done < 0 ? 0 : done > ql ? ql : done);
      }
    }
    if (W == 1 || Wp == 1) __syncthreads();   // (one wavefront: the boundary row is in memory before the next stripe reads it)
    else if (s + 1 < n_stripes) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (lane == 63) prog[out_slot] = ((s + 1) << 20) | ql; }
    if (s == n_stripes - 1) {   // the lane that owned (tl-1, ql-1) has the score
      const int owner = (tl - 1) & 63;
      final_score = __builtin_amdgcn_readlane(final_score, owner);
      if (lane == 0) s_score = final_score;
    }
  }
  if (aborted && lane == 0) atomicExch(abort_flag, 1);
  __syncthreads();   // the direction bytes are in HBM, the score in LDS
  if (wave != 0) return;
  final_score = s_score;
  {
    // ksw_backtrack from (tl-1, ql-1); ops are left in backtrack order, the host reverses
    // them.  A register window holds the direction bytes of 64 rows x 4 columns along the current diagonal (one HBM
    // latency per ~60 steps of a mostly diagonal path instead of one per step); runs are merged in registers.
    if (lane == 0) scores[P.slot] = final_score;
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
    if (lane == 0) n_cigar[P.slot] = n;
  }
}

// barrier that orders LDS traffic only
__device__ __forceinline__ void dp_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
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

// LCS length bit-parallel (Crochemore et al. / Hyyro: V = (V + U) | (V - U) with U = V & PM[text symbol]; the LCS is the
// number of zero bits of V), one wavefront per pair: lane w holds word w of the bit vector over the shorter string (up to
// 64 x 64 = 4,096 symbols), the carries of the multi-word addition come from two ballots -- lanes that generate a carry
// and lanes that would pass one on -- and one 64-bit scalar addition: carry-in mask = ((pass + (gen << 1)) ^ pass).  The
// match masks PM[symbol] (at most 8 distinct symbols, `sym_id` maps bytes to 0..7, 255 = absent) sit in LDS, the one of
// the next text symbol is fetched while the current step computes.  ~20 instructions per text symbol instead of one DP
// cell per lane per step: 5,092 pairs of ~1.2 kb take well under a millisecond (the anti-diagonal kernel: 11.6 ms).
__global__ void __launch_bounds__(64) lcs_bits_kernel(const LcsPair* pairs, const uint8_t* as, const uint8_t* bs,
                                                      const uint8_t* sym_id, int64_t* lcs_out, double* ratio_out) {
  __shared__ unsigned long long pm[8][64];
  const LcsPair P = pairs[blockIdx.x];
  const int lane = threadIdx.x;
  // pattern = the shorter string, text = the longer one
  const bool swap = P.la > P.lb;
  const uint8_t* pat = swap ? bs + P.b_off : as + P.a_off;
  const uint8_t* txt = swap ? as + P.a_off : bs + P.b_off;
  const int m = swap ? P.lb : P.la, n = swap ? P.la : P.lb;
  int64_t lcs = 0;
  if (m > 0) {
    unsigned long long mk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) {
      const int pos = 64 * lane + i;
      const int id = pos < m ? (int)sym_id[pat[pos]] : 255;
#pragma unroll
      for (int k = 0; k < 8; ++k) mk[k] |= (unsigned long long)(id == k) << i;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) pm[k][lane] = mk[k];
    __syncthreads();
    unsigned long long V = ~0ull;
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int tid = j0 + lane < n ? (int)sym_id[txt[j0 + lane]] : 255;
      const int cnt = n - j0 < 64 ? n - j0 : 64;
      int id = __builtin_amdgcn_readlane(tid, 0);
      unsigned long long cur = id < 8 ? pm[id][lane] : 0ull;
      for (int t = 0; t < cnt; ++t) {
        const int idn = __builtin_amdgcn_readlane(tid, t + 1 < cnt ? t + 1 : t);
        const unsigned long long nxt = idn < 8 ? pm[idn][lane] : 0ull;
        const unsigned long long U = V & cur;
        unsigned long long sum = V + U;
        const unsigned long long gen = __ballot(sum < V), pass = __ballot(sum == ~0ull);
        const unsigned long long carry = (pass + (gen << 1)) ^ pass;
        sum += (carry >> lane) & 1ull;
        V = sum | (V - U);
        cur = nxt;
      }
    }
    // zeros of V among the pattern's m bits
    const int valid = m - 64 * lane;
    const unsigned long long vm = valid >= 64 ? ~0ull : valid <= 0 ? 0ull : ((1ull << valid) - 1);
    int z = (int)__popcll(~V & vm);
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
    lcs = z;
  }
  if (lane == 0) {
    const int64_t maximum = (int64_t)P.la + P.lb;
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
  // device state kept between calls
  int device = -1;
  DevArena arena;
  hipStream_t stream = nullptr;   // the batch object's own non-blocking stream: calls on different objects overlap
  ~svdss_aln_batch() {
    if (stream) { if (device >= 0) (void)hipSetDevice(device); (void)hipStreamDestroy(stream); }
  }
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
  const int64_t qtot = q_off[n_pairs] - q_off[0], ttot = t_off[n_pairs] - t_off[0];
  if (b->device != device) {
    b->arena.drop();
    if (b->stream) { (void)hipStreamDestroy(b->stream); b->stream = nullptr; }
    b->device = device;
  }
  if (!b->stream) HIPCHK2(svdss_make_stream(&b->stream, "SVDSS_CALL_CUS"));
  const hipStream_t st = b->stream;
  const GapModel gm{gapo, gape, gapo2, gape2};
  hipEvent_t ev0, ev1;
  HIPCHK2(hipEventCreate(&ev0));
  HIPCHK2(hipEventCreate(&ev1));
  // chunks of pairs whose direction matrices fit the workspace budget (HBM has 288 GB; the index may hold half of it)
  int64_t dir_budget = (int64_t)24 << 30;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      dir_budget = std::min<int64_t>(dir_budget, (int64_t)((free_b + b->arena.cap) / 2));
    if (dir_budget < ((int64_t)1 << 30)) dir_budget = (int64_t)1 << 30;
  }
  std::vector<int32_t> h_nc((size_t)n_pairs, 0);
  std::vector<std::pair<int64_t, int64_t>> cig_at((size_t)n_pairs);   // (chunk-local offset, chunk index)
  std::vector<std::vector<uint32_t>> chunk_cigs;
  int64_t start = 0;
  // the longest pairs of the batch get ALN_W wavefronts each (align_wave_kernel): within a factor of four of the largest
  // matrix (SVDSS_ALIGN_FRAC, default eight), or every pair of a batch too small to fill the GPU with one wavefront per pair.
  // SVDSS_ALIGN_WAVES=1: never; =8: eight.  Measured on a bench step (5,093 pairs): 26.5 -> 16.5 ms.
  const int ALN_W = getenv("SVDSS_ALIGN_WAVES") && atoi(getenv("SVDSS_ALIGN_WAVES")) == 8 ? 8 : 4;
  const int64_t aln_frac = getenv("SVDSS_ALIGN_FRAC") && atoll(getenv("SVDSS_ALIGN_FRAC")) > 0 ? atoll(getenv("SVDSS_ALIGN_FRAC")) : 8;
  int64_t max_cells = 0;
  for (int64_t i = 0; i < n_pairs; ++i) max_cells = std::max(max_cells, (q_off[i + 1] - q_off[i]) * (t_off[i + 1] - t_off[i]));
  bool multi = !(getenv("SVDSS_ALIGN_WAVES") && atoi(getenv("SVDSS_ALIGN_WAVES")) <= 1);
  auto waves_of = [&](int64_t ql, int64_t tl) {
    if (!multi || tl < 8 * 64 || ql >= (1 << 20)) return 1;
    return (ql * tl * aln_frac >= max_cells || n_pairs < 2048) ? ALN_W : 1;
  };
  while (start < n_pairs) {
    std::vector<AlnPair> hp;
    int64_t ws = 0, dirb = 0, cig = 0, end = start;
    bool any_multi = false;
    while (end < n_pairs) {
      const int64_t ql = q_off[end + 1] - q_off[end], tl = t_off[end + 1] - t_off[end];
      const int64_t need = (ql + tl) * tl;   // direction bytes, one row of tl per anti-diagonal
      if (end > start && dirb + need > dir_budget) break;
      AlnPair a;
      memset(&a, 0, sizeof a);
      a.q_off = q_off[end] - q_off[0];
      a.t_off = t_off[end] - t_off[0];
      a.bnd_off = ws;
      a.dir_off = dirb;
      a.cig_off = cig;
      a.ql = (int32_t)ql;
      a.tl = (int32_t)tl;
      a.slot = (int32_t)(end - start);
      a.waves = waves_of(ql, tl);
      any_multi = any_multi || a.waves > 1;
      hp.push_back(a);
      cig_at[(size_t)end] = {cig, (int64_t)chunk_cigs.size()};
      ws += 3 * (a.waves + 1) * (ql + 64);
      dirb += need;
      cig += ql + tl + 2;
      b->cells += ql * tl;
      ++end;
    }
    const int64_t np = end - start;
    // longest pairs first: a pair is one chain of (tl / 64) x (ql + 63) dependent steps, the longest ones decide when
    // the launch ends
    std::stable_sort(hp.begin(), hp.end(), [](const AlnPair& x, const AlnPair& y) {
      return (int64_t)x.ql * x.tl > (int64_t)y.ql * y.tl;
    });
    const size_t need_bytes = DevArena::padded((size_t)qtot) + DevArena::padded((size_t)ttot) + DevArena::padded(64) +
                              DevArena::padded(sizeof(AlnPair) * (size_t)np) + DevArena::padded(sizeof(int32_t) * (size_t)ws) +
                              DevArena::padded((size_t)dirb) + DevArena::padded(sizeof(uint32_t) * (size_t)cig) +
                              2 * DevArena::padded(sizeof(int32_t) * (size_t)np) + DevArena::padded(64);
    HIPCHK2(b->arena.reserve(need_bytes));
    void* d_q = b->arena.take((size_t)qtot);
    void* d_t = b->arena.take((size_t)ttot);
    void* d_mat = b->arena.take(64);
    void* d_pairs = b->arena.take(sizeof(AlnPair) * (size_t)np);
    void* d_ws = b->arena.take(sizeof(int32_t) * (size_t)ws);
    void* d_dir = b->arena.take((size_t)dirb);
    void* d_cig = b->arena.take(sizeof(uint32_t) * (size_t)cig);
    void* d_sc = b->arena.take(sizeof(int32_t) * (size_t)np);
    void* d_nc = b->arena.take(sizeof(int32_t) * (size_t)np);
    void* d_abort = b->arena.take(64);
    if (qtot) HIPCHK2(hipMemcpyAsync(d_q, queries + q_off[0], (size_t)qtot, hipMemcpyHostToDevice, st));
    if (ttot) HIPCHK2(hipMemcpyAsync(d_t, targets + t_off[0], (size_t)ttot, hipMemcpyHostToDevice, st));
    HIPCHK2(hipMemcpyAsync(d_mat, mat, (size_t)(m * m), hipMemcpyHostToDevice, st));
    HIPCHK2(hipMemcpyAsync(d_pairs, hp.data(), sizeof(AlnPair) * (size_t)np, hipMemcpyHostToDevice, st));
    HIPCHK2(hipMemsetAsync(d_abort, 0, 64, st));
    HIPCHK2(hipEventRecord(ev0, st));
    if (any_multi && ALN_W == 8)
      hipLaunchKernelGGL(align_wave_kernel<8>, dim3((unsigned)np), dim3(64 * 8), 0, st, (const AlnPair*)d_pairs,
                         (const uint8_t*)d_q, (const uint8_t*)d_t, (int)m, (const int8_t*)d_mat, gm, (int32_t*)d_ws,
                         (uint8_t*)d_dir, (uint32_t*)d_cig, (int32_t*)d_sc, (int32_t*)d_nc, (int32_t*)d_abort);
    else if (any_multi)
      hipLaunchKernelGGL(align_wave_kernel<4>, dim3((unsigned)np), dim3(64 * 4), 0, st, (const AlnPair*)d_pairs,
                         (const uint8_t*)d_q, (const uint8_t*)d_t, (int)m, (const int8_t*)d_mat, gm, (int32_t*)d_ws,
                         (uint8_t*)d_dir, (uint32_t*)d_cig, (int32_t*)d_sc, (int32_t*)d_nc, (int32_t*)d_abort);
    else
      hipLaunchKernelGGL(align_wave_kernel<1>, dim3((unsigned)np), dim3(64), 0, st, (const AlnPair*)d_pairs,
                         (const uint8_t*)d_q, (const uint8_t*)d_t, (int)m, (const int8_t*)d_mat, gm, (int32_t*)d_ws,
                         (uint8_t*)d_dir, (uint32_t*)d_cig, (int32_t*)d_sc, (int32_t*)d_nc, (int32_t*)d_abort);
    HIPCHK2(hipGetLastError());
    HIPCHK2(hipEventRecord(ev1, st));
    if (any_multi) {
      int32_t ab = 0;
      HIPCHK2(hipMemcpyAsync(&ab, d_abort, sizeof ab, hipMemcpyDeviceToHost, st));
      HIPCHK2(hipStreamSynchronize(st));
      if (ab) {   // a wavefront gave up waiting for its neighbour (never seen): the chunk again, one wavefront per pair
        fprintf(stderr, "[svdss] realignment: a multi-wavefront pair did not finish; the batch runs again with one wavefront per pair\n");
        multi = false;
        b->cells -= [&] { int64_t c = 0; for (const AlnPair& a : hp) c += (int64_t)a.ql * a.tl; return c; }();
        for (int64_t k = start; k < end; ++k) cig_at[(size_t)k] = {0, 0};
        continue;   // (same `start`: the chunk is laid out again with waves = 1)
      }
    }
    chunk_cigs.emplace_back((size_t)cig);
    HIPCHK2(hipMemcpyAsync(&b->scores[(size_t)start], d_sc, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost, st));
    HIPCHK2(hipMemcpyAsync(&h_nc[(size_t)start], d_nc, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost, st));
    if (cig) HIPCHK2(hipMemcpyAsync(chunk_cigs.back().data(), d_cig, sizeof(uint32_t) * (size_t)cig, hipMemcpyDeviceToHost, st));
    HIPCHK2(hipStreamSynchronize(st));
    float ms = 0.f;
    HIPCHK2(hipEventElapsedTime(&ms, ev0, ev1));
    b->kernel_ms += ms;
    start = end;
  }
  {
    int64_t total = 0;
    for (int64_t k = 0; k < n_pairs; ++k) total += h_nc[(size_t)k];
    b->cigar.resize((size_t)total);
    int64_t o = 0;
    for (int64_t k = 0; k < n_pairs; ++k) {
      const int32_t nc = h_nc[(size_t)k];
      b->n_cigar[(size_t)k] = nc;
      const uint32_t* src = chunk_cigs[(size_t)cig_at[(size_t)k].second].data() + cig_at[(size_t)k].first;
      for (int x = nc - 1; x >= 0; --x) b->cigar[(size_t)o++] = src[x];  // reverse (ksw_backtrack tail)
    }
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
  // workspace and stream of the calling thread, kept between calls (no hipMalloc / hipFree -- a device-wide
  // synchronisation each -- and nothing on the default stream)
  struct RatioState {
    int device = -1;
    DevArena arena;
    hipStream_t stream = nullptr;
    ~RatioState() { if (stream) { if (device >= 0) (void)hipSetDevice(device); (void)hipStreamDestroy(stream); } }
  };
  static thread_local RatioState R;
  if (R.device != device) {
    R.arena.drop();
    if (R.stream) { (void)hipStreamDestroy(R.stream); R.stream = nullptr; }
    R.device = device;
  }
  if (!R.stream) HIPCHK2(svdss_make_stream(&R.stream, "SVDSS_CALL_CUS"));
  const hipStream_t st = R.stream;
  HIPCHK2(R.arena.reserve(DevArena::padded((size_t)atot) + DevArena::padded((size_t)btot) +
                          DevArena::padded(sizeof(LcsPair) * (size_t)n_pairs) + DevArena::padded(sizeof(int32_t) * (size_t)ws) +
                          DevArena::padded(sizeof(int64_t) * (size_t)n_pairs) + DevArena::padded(sizeof(double) * (size_t)n_pairs) +
                          DevArena::padded(256)));
  struct { void* p; } d_a{R.arena.take((size_t)atot)}, d_b{R.arena.take((size_t)btot)},
      d_pairs{R.arena.take(sizeof(LcsPair) * (size_t)n_pairs)}, d_ws{R.arena.take(sizeof(int32_t) * (size_t)ws)},
      d_lcs{R.arena.take(sizeof(int64_t) * (size_t)n_pairs)}, d_ratio{R.arena.take(sizeof(double) * (size_t)n_pairs)};
  if (atot) HIPCHK2(hipMemcpyAsync(d_a.p, a + a_off[0], (size_t)atot, hipMemcpyHostToDevice, st));
  if (btot) HIPCHK2(hipMemcpyAsync(d_b.p, bsy + b_off[0], (size_t)btot, hipMemcpyHostToDevice, st));
  HIPCHK2(hipMemcpyAsync(d_pairs.p, hp.data(), sizeof(LcsPair) * (size_t)n_pairs, hipMemcpyHostToDevice, st));
  // the bit-parallel kernel takes batches over at most 8 distinct symbols whose shorter strings fit 64 x 64 bits
  bool bits_ok = !getenv("SVDSS_RATIO_DP");
  uint8_t sym_id[256];
  if (bits_ok) {
    for (int64_t i = 0; i < n_pairs && bits_ok; ++i) bits_ok = std::min(hp[(size_t)i].la, hp[(size_t)i].lb) <= 4096;
    uint8_t any = 0;
    for (int64_t i = 0; i < atot; ++i) any |= a[a_off[0] + i];
    for (int64_t i = 0; i < btot; ++i) any |= bsy[b_off[0] + i];
    memset(sym_id, 255, sizeof sym_id);
    if (any < 8) {
      for (int k = 0; k < 8; ++k) sym_id[k] = (uint8_t)k;
    } else if (bits_ok) {
      bool seen[256] = {false};
      for (int64_t i = 0; i < atot; ++i) seen[a[a_off[0] + i]] = true;
      for (int64_t i = 0; i < btot; ++i) seen[bsy[b_off[0] + i]] = true;
      int nd = 0;
      for (int c = 0; c < 256; ++c)
        if (seen[c]) { if (nd < 8) sym_id[c] = (uint8_t)nd; ++nd; }
      if (nd > 8) bits_ok = false;
    }
  }
  if (bits_ok) {
    void* d_map = R.arena.take(256);
    HIPCHK2(hipMemcpyAsync(d_map, sym_id, 256, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lcs_bits_kernel, dim3((unsigned)n_pairs), dim3(64), 0, st, (const LcsPair*)d_pairs.p,
                       (const uint8_t*)d_a.p, (const uint8_t*)d_b.p, (const uint8_t*)d_map, (int64_t*)d_lcs.p, (double*)d_ratio.p);
    HIPCHK2(hipGetLastError());
    HIPCHK2(hipMemcpyAsync(ratio_out, d_ratio.p, sizeof(double) * (size_t)n_pairs, hipMemcpyDeviceToHost, st));
    if (lcs_out) HIPCHK2(hipMemcpyAsync(lcs_out, d_lcs.p, sizeof(int64_t) * (size_t)n_pairs, hipMemcpyDeviceToHost, st));
    HIPCHK2(hipStreamSynchronize(st));   // (sym_id is a local: the copy must have left it)
    return SVDSS_OK;
  }
  const size_t lds_need = sizeof(int32_t) * 3 * (size_t)(la_max + 1) + (size_t)la_max + (size_t)lb_max + 16;
  if (lds_need <= 150 * 1024) {
    HIPCHK2(hipFuncSetAttribute((const void*)lcs_ratio_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    hipLaunchKernelGGL(lcs_ratio_kernel<true>, dim3((unsigned)n_pairs), dim3(DP_THREADS), lds_need, st,
                       (const LcsPair*)d_pairs.p, (const uint8_t*)d_a.p, (const uint8_t*)d_b.p, (int32_t*)d_ws.p,
                       (int64_t*)d_lcs.p, (double*)d_ratio.p);
  } else {
    hipLaunchKernelGGL(lcs_ratio_kernel<false>, dim3((unsigned)n_pairs), dim3(DP_THREADS), 0, st,
                       (const LcsPair*)d_pairs.p, (const uint8_t*)d_a.p, (const uint8_t*)d_b.p, (int32_t*)d_ws.p,
                       (int64_t*)d_lcs.p, (double*)d_ratio.p);
  }
  HIPCHK2(hipGetLastError());
  HIPCHK2(hipMemcpyAsync(ratio_out, d_ratio.p, sizeof(double) * (size_t)n_pairs, hipMemcpyDeviceToHost, st));
  if (lcs_out) HIPCHK2(hipMemcpyAsync(lcs_out, d_lcs.p, sizeof(int64_t) * (size_t)n_pairs, hipMemcpyDeviceToHost, st));
  HIPCHK2(hipStreamSynchronize(st));
  return SVDSS_OK;
}
