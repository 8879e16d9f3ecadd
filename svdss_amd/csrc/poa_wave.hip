// poa_wave.hip -- POA consensus, one wavefront per sub-cluster (fast path of svdss_poa_consensus_batch).
//
// Same specification as poa.hip / oracle/svdss_oracle_poa.c, bit for bit.  POA over a dozen ~1 kb reads is
// a chain of ~10^4 short dependent steps per sub-cluster (graph rows, traceback steps); what bounds it on
// this machine is the latency of each step and how many sub-clusters a CU can keep in flight, not bandwidth.
// So the kernel keeps in LDS only what the row loop touches -- ~7 KB per sub-cluster of ~1 kb reads, twenty
// per CU:
//   * the read being aligned;
//   * a ring of the last `ring` DP rows (H, E1, E2) plus two staging rows, as wide as the band gets in
//     practice (2w + 33 columns; a wider row sends the sub-cluster to a second round with the
//     specification's 2w + 129).
// One 32-bit descriptor per graph row (base, up to two predecessor-row deltas, flags) is rebuilt in parallel
// in HBM before each read is aligned, so the row loop never chases graph pointers: it takes the descriptors
// of 64 rows at a time into a register and reads them with v_readlane.  The graph itself (int32 arrays)
// lives in HBM: it is touched by the parallel phases only.  The heaviest-bundle consensus runs as its own
// small kernel afterwards (poa_bundle_kernel), so that its per-row tables do not count against the LDS of
// the row loop.
//
//   forward    one wavefront computes a DP row per step, C consecutive band columns per lane (C = 1, 3, 5:
//              odd strides are LDS-bank-conflict free).  The horizontal gap states are prefix maxima
//              F(j) = max_{k<j}(H'(k)+k*e) - o - j*e: a serial pass over the lane's C columns and one DPP
//              max-scan over the lane totals; no barrier anywhere.  HBM receives write-once streams: a
//              32-bit *direction word* per cell that encodes every decision the traceback can take there
//              (so the traceback never compares scores), the predecessor-row deltas of each row, and --
//              only for rows flagged as the source of a long deletion edge -- a copy of H/E1/E2 that is
//              read back into a staging row when its successor comes up.
//   traceback  a register window holds the direction words of 64 rows x 4 columns along the current
//              diagonal (one HBM latency per ~30-60 steps); the walk itself is scalar.
//   update     parallel over the alignment: wave scans number the surviving path elements and the new
//              nodes; every path edge touches edge lists no other edge touches, so edges are added
//              concurrently.
//   order      any topological order gives the same DP values, traceback (predecessor slots are edge
//              order) and consensus.  Instead of Kahn's serial queue the kernel keeps a column rank per
//              node (aligned nodes share a column, a read's path is column-monotone, inserted bases open
//              new columns right after their anchor's) and rebuilds the order with scans and a counting
//              sort by column.
//   consensus  heaviest bundle over a per-row successor table staged in LDS.
// Clusters that do not fit (graph beyond its allocation, > 8 predecessors or > 2 far predecessors on a
// node, rows wider than the LDS ring rows) report status 3 and are redone by the HBM kernel of poa.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "poa_wave.h"

#define PNEG (-0x20000000)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
// the chain rows compute on score * TG + tag (see there)
#define TG 16
#define TGB 0x20000000
#define TG_M 15
#define TG_E1 7
#define TG_E2 6
#define TG_F1 5
#define TG_F2 4
#define P_O1 4
#define P_E1 2
#define P_O2 24
#define P_E2 1
#define P_MATCH 2
#define P_MISMATCH 4
#define COL_SINK 0x7FFFFFFE
#define COL_NEW 0x7FFFFFFF
#define RI_SLOW 3u

struct WsLayout {
  int64_t out_head, in_head, order, index, col, base;
  int64_t row_beg, row_end, hl, prow0, prow1, row_mpl, row_mpr;
  int64_t aln, scr, rinfo, keepf;
  int64_t e_from, e_to, e_w, e_next_out, e_next_in;
  int64_t op_node, op_q, path_use, path_aux;
  int64_t gdir, gH, gE1, gE2;
  int64_t total;
};

__host__ __device__ inline WsLayout ws_layout(int nc, int ec, int max_len, int ws) {
  WsLayout w;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += n; return at; };
  w.out_head = take(nc); w.in_head = take(nc); w.order = take(nc); w.index = take(nc); w.col = take(nc); w.base = take(nc);
  w.row_beg = take(nc); w.row_end = take(nc); w.hl = take(nc); w.prow0 = take(nc); w.prow1 = take(nc);
  w.row_mpl = take(nc); w.row_mpr = take(nc);
  w.aln = take(5 * (int64_t)nc);
  w.scr = take((int64_t)nc + 64);
  w.rinfo = take((int64_t)nc + 64); w.keepf = take((int64_t)nc + 64);
  w.e_from = take(ec); w.e_to = take(ec); w.e_w = take(ec); w.e_next_out = take(ec); w.e_next_in = take(ec);
  const int64_t opcap = (int64_t)nc + max_len + 4;
  w.op_node = take(opcap); w.op_q = take(opcap); w.path_use = take(opcap); w.path_aux = take(opcap);
  const int64_t pool = (int64_t)nc * ws;
  w.gdir = take(pool); w.gH = take(pool); w.gE1 = take(pool); w.gE2 = take(pool);
  w.total = o;
  return w;
}

// (... + TGB: "no path" is 0 and everything near it, so that a lane shift can fill with zero)
__device__ __forceinline__ int tg_up(int x, int tag) { return x <= PNEG / 2 ? 0 : (int)(((unsigned)x << 4) | (unsigned)tag) + TGB; }
__device__ __forceinline__ int tg_down(int x) { return x <= TGB / 2 ? PNEG : (x - TGB) >> 4; }
__device__ __forceinline__ int pl_score(int a, int b) { return (a >= 4 || b >= 4) ? 0 : (a == b ? P_MATCH : -P_MISMATCH); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

template <int CTRL, int RMASK>
__device__ __forceinline__ int dppi(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, RMASK, 0xf, false);
}

// inclusive scans over the 64 lanes of a wave: four row_shr steps, then row_bcast15 / row_bcast31
__device__ __forceinline__ int wave_scan_max(int x, int ident) {
  x = imax(x, dppi<0x111, 0xf>(ident, x));
  x = imax(x, dppi<0x112, 0xf>(ident, x));
  x = imax(x, dppi<0x114, 0xf>(ident, x));
  x = imax(x, dppi<0x118, 0xf>(ident, x));
  x = imax(x, dppi<0x142, 0xa>(ident, x));
  x = imax(x, dppi<0x143, 0xc>(ident, x));
  return x;
}

__device__ __forceinline__ int wave_scan_min(int x, int ident) {
  x = imin(x, dppi<0x111, 0xf>(ident, x));
  x = imin(x, dppi<0x112, 0xf>(ident, x));
  x = imin(x, dppi<0x114, 0xf>(ident, x));
  x = imin(x, dppi<0x118, 0xf>(ident, x));
  x = imin(x, dppi<0x142, 0xa>(ident, x));
  x = imin(x, dppi<0x143, 0xc>(ident, x));
  return x;
}

__device__ __forceinline__ int wave_scan_add(int x) {
  x += dppi<0x111, 0xf>(0, x);
  x += dppi<0x112, 0xf>(0, x);
  x += dppi<0x114, 0xf>(0, x);
  x += dppi<0x118, 0xf>(0, x);
  x += dppi<0x142, 0xa>(0, x);
  x += dppi<0x143, 0xc>(0, x);
  return x;
}

// value of the lane below (lane 0 receives `fill`)
__device__ __forceinline__ int wave_shr1(int x, int fill) { return dppi<0x138, 0xf>(fill, x); }
// the same with 0 for the lane that has no neighbour (bound_ctrl: no register to preload with the fill value)
__device__ __forceinline__ int wave_shr1z(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_shl1z(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true); }
// value of the lane above (lane 63 receives `fill`)
__device__ __forceinline__ int wave_shl1(int x, int fill) { return dppi<0x130, 0xf>(fill, x); }

// inclusive max-scan whose steps are single DPP instructions: a lane without a source keeps its own value (INT_MIN is
// the identity the DPP combiner folds into v_max_i32_dpp)
__device__ __forceinline__ int wave_scan_max_self(int x) { return wave_scan_max(x, -0x7fffffff - 1); }

// direction word layout
//  bits 0-3  source of H : 0-7 match through predecessor slot k, 8 E1, 9 E2, 10 F1, 11 F2
//  bits 4-7  source of H': 0-7 match through slot k, 8 E1, 9 E2
//  bits 8-11 E1: 0-7 opened from H of slot k, 8-15 extended from E1 of slot k-8;  bits 12-15 E2 likewise
//  bit 16    F1 opened from H'(v, j-1) (else extended);  bit 17 F2 likewise
//
// row descriptor layout (rowinfo[r], one per topological position)
//  16 bits: bits 0-2 base, bits 3-4 number of predecessors (0-2, or RI_SLOW: look at the graph), bits 5-9 and
//  10-14 the row deltas (< 32) of predecessor 0 / 1, bit 15: some later row reads this row after it left the ring,
//  bit 16: some later row other than the next one reads this row from the ring
#define RI_KEEP 0x8000u
#define RI_RING 0x10000u
// a chain row: one predecessor, the previous row, and no copy kept in HBM
#define RI_CHAIN_MASK ((3u << 3) | (31u << 5) | RI_KEEP)
#define RI_CHAIN_VAL ((1u << 3) | (1u << 5))
// one predecessor (any row), no copy kept in HBM
#define RI_ONE_MASK ((3u << 3) | RI_KEEP)
#define RI_ONE_VAL (1u << 3)

struct ArrI {
  int32_t* W; uint32_t o;
  __device__ __forceinline__ int32_t& operator[](int i) const { return W[o + (uint32_t)i]; }
};
struct ArrU {
  int32_t* W; uint32_t o;
  __device__ __forceinline__ uint32_t& operator[](int i) const { return ((uint32_t*)W)[o + (uint32_t)i]; }
};

__device__ unsigned long long g_poaw_prof[8];   // SVDSS_DEBUG: time in prepare, forward, traceback, update, bundle
#ifdef POA_COUNT_ROWS
__device__ unsigned long long g_poaw_rows[16];  // rows by type (developer build)
#define ROWCNT(k) (++rowcnt[k])
#else
#define ROWCNT(k)
#endif
#define PROF_T() (prof_t = wall_clock64())
#define PROF_ADD(k) do { const unsigned long long t_ = wall_clock64(); prof[k] += t_ - prof_t; prof_t = t_; } while (0)

// POA_MIN_WAVES (developer builds, `make poaocc W=5`): wavefronts per SIMD the register allocation is asked to leave room for
#ifdef POA_MIN_WAVES
#define POA_BOUNDS __launch_bounds__(64, POA_MIN_WAVES)
#else
#define POA_BOUNDS __launch_bounds__(64)
#endif
template <int C>
__global__ void POA_BOUNDS poa_wave_kernel(const PoaWaveTask* tasks, const uint8_t* seqs, const int64_t* seq_off,
                                                     int32_t* ws32, uint8_t* ws8, int32_t* cons_len, int32_t* status,
                                                     unsigned long long* cells) {
  extern __shared__ __align__(16) unsigned char smem[];
  const PoaWaveTask T = tasks[blockIdx.x];
  const int lane = threadIdx.x;
  const int nc = T.nc, ec = T.ec, WS = T.ws, wm = WS - 1, RS = T.rs, RING = T.ring, rm = RING - 1, NS = RING + 2;
  constexpr int G = 4;            // -inf guard cells on each side of a ring row
  constexpr bool CHAIN = C <= 2;  // (the wider instantiations only see the second and third rounds)
  const int RST = RS + 2 * G;     // LDS stride of a ring row
  // ---- LDS
  int32_t* rH = (int32_t*)smem;
  int32_t* rE1 = rH + NS * RST;
  int32_t* rE2 = rE1 + NS * RST;
  int32_t* rbeg = rE2 + NS * RST;
  int32_t* rend = rbeg + NS;
  int32_t* rmpl = rend + NS;
  int32_t* rmpr = rmpl + NS;
  int32_t* sh = rmpr + NS;          // 0 nodes, 1 edges, 2 columns, 3 nops; 8..15 predecessor slots of a slow row
  uint8_t* q = (uint8_t*)(sh + 16) + 16;   // q[-1] = N: column 0 has no match score
  // ---- HBM
  const WsLayout wl = ws_layout(nc, ec, T.max_len, WS);
  // every HBM array of the sub-cluster is (one base pointer, a 32-bit offset): thirty 64-bit pointers would not fit
  // the scalar registers and the row loop would keep reloading spilled ones
  int32_t* const W = ws32 + T.ws_off;
  const ArrI out_head{W, (uint32_t)wl.out_head}, in_head{W, (uint32_t)wl.in_head}, order{W, (uint32_t)wl.order},
      index{W, (uint32_t)wl.index}, col{W, (uint32_t)wl.col}, base{W, (uint32_t)wl.base}, row_beg{W, (uint32_t)wl.row_beg},
      row_end{W, (uint32_t)wl.row_end}, hl{W, (uint32_t)wl.hl}, row_mpl{W, (uint32_t)wl.row_mpl},
      row_mpr{W, (uint32_t)wl.row_mpr}, aln{W, (uint32_t)wl.aln}, scr{W, (uint32_t)wl.scr}, e_from{W, (uint32_t)wl.e_from},
      e_to{W, (uint32_t)wl.e_to}, e_w{W, (uint32_t)wl.e_w}, e_next_out{W, (uint32_t)wl.e_next_out},
      e_next_in{W, (uint32_t)wl.e_next_in}, op_node{W, (uint32_t)wl.op_node}, op_q{W, (uint32_t)wl.op_q},
      path_use{W, (uint32_t)wl.path_use}, gH{W, (uint32_t)wl.gH}, gE1{W, (uint32_t)wl.gE1}, gE2{W, (uint32_t)wl.gE2};
  const ArrU prow0{W, (uint32_t)wl.prow0}, prow1{W, (uint32_t)wl.prow1}, rinfo{W, (uint32_t)wl.rinfo},
      keepf{W, (uint32_t)wl.keepf}, path_aux{W, (uint32_t)wl.path_aux}, gdir{W, (uint32_t)wl.gdir};
  const int opcap = nc + T.max_len + 4;
  const int n = (int)T.n_seqs;
  unsigned long long my_cells = 0;
  unsigned long long prof[5] = {0, 0, 0, 0, 0}, prof_t;
#ifdef POA_COUNT_ROWS
  unsigned rowcnt[16] = {0};
#endif
#ifdef POA_FINE_PROF
  long long fp[6] = {0, 0, 0, 0, 0, 0}, ft = 0;
#define FP(k) do { const long long t_ = clock64(); fp[k] += t_ - ft; ft = t_; } while (0)
#else
#define FP(k)
#endif
  if (n <= 0) { if (lane == 0) { cons_len[blockIdx.x] = 0; status[blockIdx.x] = 0; } return; }
  if (T.prio >= 3) __builtin_amdgcn_s_setprio(3);
  else if (T.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (T.prio == 1) __builtin_amdgcn_s_setprio(1);
  // (every lane stores the same word: a branch on the lane number in front of a return would make every loop
  // around it a loop with a divergent exit, and the compiler would carry all their scalars in vector registers)
#define FAIL(code) do { status[blockIdx.x] = (code); return; } while (0)
  // ------------------------------------------------------------------ graph of the first read
  {
    const uint8_t* q0 = seqs + seq_off[T.seq_first];
    const int L0 = (int)(seq_off[T.seq_first + 1] - seq_off[T.seq_first]);
    if (L0 + 2 > nc || L0 + 1 > ec) FAIL(3 | (1 << 8));
    for (int v = lane; v < L0 + 2; v += 64) {
      const int b = v < 2 ? 4 : q0[v - 2];
      base[v] = b;
      out_head[v] = v == 1 ? -1 : (v == 0 ? 0 : v - 1);      // edge e: (e == 0 ? source : node e+1) -> ...
      in_head[v] = v == 0 ? -1 : (v == 1 ? L0 : v - 2);
      for (int x = 0; x < 5; ++x) aln[5 * v + x] = -1;
      if (v >= 2) aln[5 * v + b] = v;
      const int idx = v == 0 ? 0 : v == 1 ? L0 + 1 : v - 1;
      col[v] = v == 1 ? COL_SINK : idx;
      index[v] = idx;
      order[idx] = v;
    }
    for (int e = lane; e <= L0; e += 64) {
      e_from[e] = e == 0 ? 0 : e + 1;
      e_to[e] = e == L0 ? 1 : e + 2;
      e_w[e] = 1;
      e_next_out[e] = -1; e_next_in[e] = -1;
    }
    if (lane == 0) { sh[0] = L0 + 2; sh[1] = L0 + 1; sh[2] = L0 + 1; }
    __syncthreads();
  }
  for (int i = 1; i < n; ++i) {
    const uint8_t* qg = seqs + seq_off[T.seq_first + i];
    const int L = UNI((int)(seq_off[T.seq_first + i + 1] - seq_off[T.seq_first + i]));
    // (values read back from LDS / HBM are uniform by construction; the compiler has to be told, or the whole row
    // loop -- its counter, bands and branches -- is carried in vector registers under exec masks)
    const int N = __builtin_amdgcn_readfirstlane(sh[0]);
    if (L > T.max_len) FAIL(3 | (6 << 8));
    PROF_T();
    for (int j = lane; j < L; j += 64) q[j] = qg[j];
    if (lane == 0) q[-1] = 4;
    // ---------------------------------------------------------- row descriptors (HBM)
    for (int r = lane; r < N + 64; r += 64) {
      uint32_t ri = 0;
      if (r < N) {
        const int v = order[r];
        ri = (uint32_t)base[v] & 7u;
        int np = 0, d0 = 0, d1 = 0;
        for (int e = in_head[v]; e >= 0; e = e_next_in[e]) {
          const int d = r - index[e_from[e]];
          if (np == 0) d0 = d; else if (np == 1) d1 = d;
          ++np;
        }
        if (np > 2 || d0 > 31 || d1 > 31) ri |= RI_SLOW << 3;
        else {
          ri |= ((uint32_t)np << 3) | ((uint32_t)d0 << 5) | ((uint32_t)d1 << 10);
          // (the predecessor deltas the traceback reads: chain rows do not store them again)
          prow0[r] = (uint32_t)d0 | ((uint32_t)d1 << 8);
          prow1[r] = 0;
        }
        hl[r] = PNEG;
      }
      rinfo[r] = ri;
      keepf[r] = 0;
    }
    __syncthreads();
    // flag the rows that are read back after they left the ring, and those a row other than the next one reads
    // from the ring (a chain row leaves its values in registers unless it is flagged)
    auto flag_row = [&](int rr, int d) { if (d >= 2) atomicOr(&keepf[rr], d >= RING ? RI_KEEP : RI_RING); };
    for (int r = lane; r < N - 1; r += 64) {
      const uint32_t ri = rinfo[r];
      const uint32_t np = (ri >> 3) & 3u;
      if (np == RI_SLOW) {
        for (int e = in_head[order[r]]; e >= 0; e = e_next_in[e]) {
          const int d = r - index[e_from[e]];
          flag_row(r - d, d);
        }
      } else {
        const int d0 = (int)((ri >> 5) & 31u), d1 = (int)((ri >> 10) & 31u);
        if (np >= 1) flag_row(r - d0, d0);
        if (np >= 2) flag_row(r - d1, d1);
      }
    }
    __syncthreads();
    PROF_ADD(0);
    int nops = -1;
    // banded first; if the band loses the sink the read is aligned again with the full matrix (w = L)
    for (int attempt = 0; attempt < 2 && nops < 0; ++attempt) {
      const int w = UNI(attempt ? L : 10 + (int)(0.01 * L));
      int last_r = -1, last_mpl = 0, last_mpr = 0, last_beg = 0, last_end = 0;
      // a row that left the ring comes back from HBM into staging slot s (its stores may still be in flight)
      auto stage = [&](int ur, int s) {
        __syncthreads();
        const int pb = __builtin_amdgcn_readfirstlane(row_beg[ur]), pe = __builtin_amdgcn_readfirstlane(row_end[ur]);
        const int po = ur * WS;
        for (int x = lane; x <= pe - pb; x += 64) {
          const int o = po + ((pb + x) & wm);
          rH[s * RST + G + x] = gH[o]; rE1[s * RST + G + x] = gE1[o]; rE2[s * RST + G + x] = gE2[o];
        }
        if (lane < G) {
          const int o1 = s * RST + lane, o2 = s * RST + G + (pe - pb + 1) + lane;
          rH[o1] = PNEG; rE1[o1] = PNEG; rE2[o1] = PNEG;
          rH[o2] = PNEG; rE1[o2] = PNEG; rE2[o2] = PNEG;
        }
        if (lane == 0) { rbeg[s] = pb; rend[s] = pe; rmpl[s] = row_mpl[ur]; rmpr[s] = row_mpr[ur]; }
        __syncthreads();
      };
      uint32_t ri_blk = 0;          // descriptors of rows [blk0, blk0 + 64), one per lane
      int blk0 = -64;
      // ------------------------------------------------------------ forward (the sink is order[N-1])
      for (int r_ = 0; r_ < N - 1; ++r_) {
        // (the loop-carried scalars, told to be uniform once per row: the tail of the body is a branch on the lane
        // number, which makes the analysis treat everything that flows around the loop as divergent)
        int r = UNI(r_);
        last_r = UNI(last_r); last_mpl = UNI(last_mpl); last_mpr = UNI(last_mpr); last_beg = UNI(last_beg); last_end = UNI(last_end);
        blk0 = UNI(blk0);
        FP(5);
        if (r - blk0 >= 64) {   // (both arrays are N + 64 long)
          blk0 = r;
          ri_blk = rinfo[r + lane] | keepf[r + lane];
          // wait for the block here, once per 64 rows: a wait at the use below would be a vmcnt(0) in every row, i.e.
          // every row would wait for its own direction-word stores to reach HBM
          asm volatile("" : "+v"(ri_blk));
        }
        uint32_t ri = __builtin_amdgcn_readlane(ri_blk, r - blk0);
        // ---- chain rows: one predecessor, the previous row, nothing kept in HBM (9 rows in 10).  The previous row
        // stays in registers -- lane l holds columns beg + C*l .. beg + C*l + C-1 of its band, -inf outside -- and
        // moves with the band (a band that starts one column further reads its right neighbour's values through one
        // DPP shift); a row goes to the LDS ring only if a row other than the next one reads it (RI_RING), or when
        // the chain ends.  Same arithmetic and tie rules as the general code below for np == 1.
        if (CHAIN && 64 * C <= WS && r > 0 && (ri & RI_ONE_MASK) == RI_ONE_VAL && (int)((ri >> 5) & 31u) < RING &&
            ((ri >> 5) & 31u) >= 1u && (((ri >> 5) & 31u) != 1u || last_r == r - 1)) {
#ifdef POA_COUNT_ROWS
          const unsigned long long chain_t0 = wall_clock64();
#endif
          int pbeg = 0, pend = 0, pm_l = 0, pm_r = 0;   // band and maximum columns of the row in the registers
          int32_t pH[C], pE1[C], pE2[C];
          bool in_ring = true;
          // read symbols: qc = q[j - 1] of this row's columns, qx = q[j] (what the next row needs if its band starts
          // one column further; fetched a row ahead)
          int qc[C], qx[C];
          // the row ur of the ring becomes the row in the registers (a chain starts, or goes on from an earlier row:
          // a row whose one predecessor is not the previous row)
          auto load_pred = [&](int ur) {
            const int sl = ur & rm;
            if (ur == last_r) { pbeg = last_beg; pend = last_end; pm_l = last_mpl; pm_r = last_mpr; }
            else { pbeg = UNI(rbeg[sl]); pend = UNI(rend[sl]); pm_l = UNI(rmpl[sl]); pm_r = UNI(rmpr[sl]); }
            const int so = sl * RST + G;
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const int idx = lane * C + c;
              const bool ok = idx <= pend - pbeg;
              const int a = so + (ok ? idx : 0);
              const int32_t x0 = rH[a], x1 = rE1[a], x2 = rE2[a];
              pH[c] = ok ? tg_up(x0, TG_M) : 0; pE1[c] = ok ? tg_up(x1, TG_E1) : 0; pE2[c] = ok ? tg_up(x2, TG_E2) : 0;
            }
            const int jb = pbeg + lane * C;
#pragma unroll
            for (int c = 0; c < C; ++c) { qc[c] = q[imin(jb + c, L) - 1]; qx[c] = q[imin(jb + c, L - 1)]; }
            in_ring = true;
          };
          load_pred(r - (int)((ri >> 5) & 31u));
          auto ring_store = [&](int rr_, int b_, int e_, const int32_t* h_, const int32_t* e1_, const int32_t* e2_, int l_, int r_) {
            const int sl = rr_ & rm, sb_ = sl * RST + G;
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const int idx = lane * C + c;
              if (idx <= e_ - b_) { rH[sb_ + idx] = tg_down(h_[c]); rE1[sb_ + idx] = tg_down(e1_[c]); rE2[sb_ + idx] = tg_down(e2_[c]); }
            }
            if (lane < G) {
              const int o1 = sl * RST + lane, o2 = sb_ + (e_ - b_ + 1) + lane;
              rH[o1] = PNEG; rE1[o1] = PNEG; rE2[o1] = PNEG;
              rH[o2] = PNEG; rE1[o2] = PNEG; rE2[o2] = PNEG;
            }
            if (lane == 0) { rbeg[sl] = b_; rend[sl] = e_; rmpl[sl] = l_; rmpr[sl] = r_; }
          };
          for (;;) {
            int beg = pm_l + 1 - w; if (beg < 0) beg = 0;
            int end = pm_r + 1 + w; if (end > L) end = L;
            if (end - beg + 1 > 2 * w + 129) end = beg + 2 * w + 128;
            const int width = end - beg + 1;
            const int delta = beg - pbeg;
            if (width > 64 * C || width > RS || (unsigned)delta > 1u) break;   // (the general code's business)
            if (__builtin_expect(delta == 0, 0)) {
              // one row in fifty: the band did not move.  The previous row moves one column up the lanes (its origin
              // becomes pbeg - 1) and the code below applies unchanged
              if (pend - pbeg + 1 >= 64 * C) break;
              const int32_t tH = wave_shr1z(pH[C - 1]), t1 = wave_shr1z(pE1[C - 1]), t2 = wave_shr1z(pE2[C - 1]);
#pragma unroll
              for (int c = C - 1; c > 0; --c) { pH[c] = pH[c - 1]; pE1[c] = pE1[c - 1]; pE2[c] = pE2[c - 1]; }
              pH[0] = tH; pE1[0] = t1; pE2[0] = t2;
#pragma unroll
              for (int c = 0; c < C; ++c) qx[c] = qc[c];
            }
            my_cells += (unsigned long long)width;
            ROWCNT(1);
            const int bv = (int)(ri & 7u);
            const int s_mat = bv < 4 ? P_MATCH * TG : 0, s_mis = bv < 4 ? -P_MISMATCH * TG : 0;
            const int jb = beg + lane * C;
            // predecessor values of columns j - 1 (own registers) and j (the next register / the next lane's first)
            const int32_t nH = wave_shl1z(pH[0]), n1 = wave_shl1z(pE1[0]), n2 = wave_shl1z(pE2[0]);
            // All values of the chain loop are TGB + score * 16 + tag ("no path": 0 and what is near it, so that the
            // lane shifts fill with zero through bound_ctrl), the tag naming where the value came from by the code the
            // traceback reads, inverted (15 - code: M 15, E1 7, E2 6, F1 5, F2 4): a three-way maximum then picks the
            // value AND, among equal scores, the source the specification tries first, and the direction nibbles are
            // the inverted low bits of H and H' -- no chain of compares and selects.  (A larger score wins whatever the
            // tags: 16 > 15.)  Values that meet each other as candidates of one state carry the same tag, so their
            // order and their ties are those of the scores.
            int32_t m0[C], e1[C], e2[C], hp[C], hq[C], t1[C], t2[C], p1[C], p2[C];
            uint32_t dw[C];
            bool valid[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const int j = jb + c;
              valid[c] = j <= end;
              const int32_t hA = pH[c], hB = c + 1 < C ? pH[c + 1] : nH, xa = c + 1 < C ? pE1[c + 1] : n1, xb = c + 1 < C ? pE2[c + 1] : n2;
              qc[c] = qx[c];
              int sc = qc[c] == bv ? s_mat : s_mis;
              sc = qc[c] >= 4 ? 0 : sc;
              m0[c] = hA + sc;                                                                   // tag M
              const int32_t a1 = hB - (P_O1 + P_E1) * TG - (TG_M - TG_E1), b1 = xa - P_E1 * TG;   // tag E1
              const int32_t a2 = hB - (P_O2 + P_E2) * TG - (TG_M - TG_E2), b2 = xb - P_E2 * TG;   // tag E2
              e1[c] = imax(a1, b1); e2[c] = imax(a2, b2);
              dw[c] = (b1 > a1 ? 0x800u : 0u) | (b2 > a2 ? 0x8000u : 0u);
              hp[c] = valid[c] ? imax(m0[c], imax(e1[c], e2[c])) : 0;                         // tag = its source
              hq[c] = hp[c] | TG_M;                                                              // H' as a candidate: tag M
              t1[c] = hq[c] + j * (P_E1 * TG); t2[c] = hq[c] + j * (P_E2 * TG);
              p1[c] = c ? imax(p1[c - 1], t1[c]) : t1[c];
              p2[c] = c ? imax(p2[c - 1], t2[c]) : t2[c];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) qx[c] = q[imin(jb + c, L - 1)];
            // the row maximum rides along with the two F scans: F(j) = max_{k<j} H'(k) - o - (j - k) e < max H', so the
            // maximum of H over the row is the maximum of H' and H(j) attains it exactly where H'(j) does -- known three
            // scans earlier than from the finished H (a third DPP chain in the shadow of the other two)
            int32_t lmax = hq[0];
#pragma unroll
            for (int c = 1; c < C; ++c) lmax = imax(lmax, hq[c]);
            const int32_t s1 = wave_scan_max_self(p1[C - 1]), s2 = wave_scan_max_self(p2[C - 1]), s3 = wave_scan_max_self(lmax);
            const int32_t X1 = wave_shr1z(s1), X2 = wave_shr1z(s2);
            // (F opens at j from H'(j - 1) when that is a maximum of the prefix: t(j - 1) == x(j), the same test as
            // H'(j - 1) - o - e == F(j) without the two subtractions)
            const int32_t t1_prev = wave_shr1z(t1[C - 1]), t2_prev = wave_shr1z(t2[C - 1]);
            int32_t h[C];
            const int rowo = r * WS;
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const int j = jb + c;
              const int32_t x1 = c ? imax(X1, p1[c - 1]) : X1, x2 = c ? imax(X2, p2[c - 1]) : X2;
              const int32_t f1 = x1 - (P_O1 * TG + (TG_M - TG_F1)) - j * (P_E1 * TG), f2 = x2 - (P_O2 * TG + (TG_M - TG_F2)) - j * (P_E2 * TG);
              h[c] = imax(hp[c], imax(f1, f2));
              // the traceback's decisions.  Every lane stores: the columns past the band's end land in slots of this
              // row that nothing reads (64 * C <= WS)
              const uint32_t dH = ~(uint32_t)h[c] & 15u, dHp = ~(uint32_t)hp[c] & 15u;
              const uint32_t o1 = (c ? t1[c - 1] : t1_prev) == x1 ? 0x10000u : 0u;
              const uint32_t o2 = (c ? t2[c - 1] : t2_prev) == x2 ? 0x20000u : 0u;
              // (a 32-bit byte offset from the uniform base: one address instruction less than a 64-bit sum.  gdir is
              // the first of the row pools and a task has at most 65,000 rows of at most 4,096 words -- poa.hip --, so
              // it ends below 2^30 words of the sub-cluster's workspace)
              *(uint32_t*)((char*)W + (((gdir.o + (uint32_t)(rowo + (j & wm))) << 2) & 0xfffffffcu)) = dw[c] | dH | (dHp << 4) | o1 | o2;
            }
            if (end == L) {
#pragma unroll
              for (int c = 0; c < C; ++c) if (jb + c == L) hl[r] = tg_down(h[c]);
            }
#pragma unroll
            // (E1 / E2 of the columns past the band's end are reset like H: the end of the band can move left -- the
            // rightmost maximum jumps --, so such a column may have been inside the previous row's band and hold
            // real values, and it may be inside the next row's band again)
            for (int c = 0; c < C; ++c) { pH[c] = valid[c] ? (h[c] | TG_M) : 0; pE1[c] = valid[c] ? e1[c] : 0; pE2[c] = valid[c] ? e2[c] : 0; }
            // leftmost / rightmost column of the row maximum (the columns past the end hold -inf: they can only tie
            // with a row of unreachable cells, which the test below sends to beg / end anyway)
            const int32_t wmx = __builtin_amdgcn_readlane(s3, 63);
            int l = 1 << 20, rr = -1;
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const unsigned long long em = __ballot(hq[c] == wmx);
              if (em) { l = imin(l, C * (int)__builtin_ctzll(em) + c); rr = imax(rr, C * (63 - (int)__builtin_clzll(em)) + c); }
            }
            l += beg; rr += beg;
            if (wmx <= TGB / 2) { l = beg; rr = end; }
            last_r = r; last_mpl = UNI(l); last_mpr = UNI(rr); last_beg = beg; last_end = end;
            pbeg = beg; pend = end; pm_l = last_mpl; pm_r = last_mpr;
            in_ring = (ri & RI_RING) != 0;
            if (in_ring) ring_store(r, beg, end, pH, pE1, pE2, l, rr);
            ++r;
            if (r >= N - 1) break;
            if (r - blk0 >= 64) {
              blk0 = r;
              ri_blk = rinfo[r + lane] | keepf[r + lane];
              asm volatile("" : "+v"(ri_blk));
            }
            ri = __builtin_amdgcn_readlane(ri_blk, r - blk0);
            if ((ri & RI_ONE_MASK) != RI_ONE_VAL) break;
            const int dn = (int)((ri >> 5) & 31u);
            if (dn != 1) {
              // the one predecessor of the next row is an earlier row of the ring: the row just computed stays behind
              // (it is in the ring if any later row reads it), that one comes into the registers
              if (dn < 1 || dn >= RING) break;
              ROWCNT(9);
              load_pred(r - dn);
            }
          }
          if (!in_ring) ring_store(r - 1, pbeg, pend, pH, pE1, pE2, last_mpl, last_mpr);
#ifdef POA_COUNT_ROWS
          prof[4] += wall_clock64() - chain_t0;
          ROWCNT(2);
#endif
          if (r >= N - 1) break;
          r_ = r;
        }
        const int slot = r & rm;
        const int bv = (int)(ri & 7u);
        int np = (int)((ri >> 3) & 3u);
        const bool keep = ((ri >> 15) & 1u) != 0;
        const bool slow = np == (int)RI_SLOW;
        int ps0 = 0, ps1 = 0;
        int lo = 1 << 30, hi = -1;
        uint32_t pd0 = 0, pd1 = 0;
        if (!slow) {
          int nfar = 0;
          if (np >= 1) {
            const int d = (int)((ri >> 5) & 31u), ur = r - d;
            pd0 = (uint32_t)imin(d, 255);
            if (d < RING) ps0 = ur & rm; else { ps0 = RING + nfar++; stage(ur, ps0); }
            const int a = ur == last_r ? last_mpl : __builtin_amdgcn_readfirstlane(rmpl[ps0]);
            const int b = ur == last_r ? last_mpr : __builtin_amdgcn_readfirstlane(rmpr[ps0]);
            lo = imin(lo, a); hi = imax(hi, b);
          }
          if (np >= 2) {
            const int d = (int)((ri >> 10) & 31u), ur = r - d;
            pd0 |= (uint32_t)imin(d, 255) << 8;
            if (d < RING) ps1 = ur & rm; else { ps1 = RING + nfar++; stage(ur, ps1); }
            const int a = ur == last_r ? last_mpl : __builtin_amdgcn_readfirstlane(rmpl[ps1]);
            const int b = ur == last_r ? last_mpr : __builtin_amdgcn_readfirstlane(rmpr[ps1]);
            lo = imin(lo, a); hi = imax(hi, b);
          }
        } else {
          int nfar = 0;
          np = 0;
          for (int e = __builtin_amdgcn_readfirstlane(in_head[order[r]]); e >= 0; e = __builtin_amdgcn_readfirstlane(e_next_in[e])) {
            const int ur = __builtin_amdgcn_readfirstlane(index[e_from[e]]);
            if (np < 8) {
              const uint32_t dl = (uint32_t)imin(r - ur, 255);
              if (np < 4) pd0 |= dl << (8 * np); else pd1 |= dl << (8 * (np - 4));
              int s;
              if (r - ur < RING) s = ur & rm;
              else {
                if (nfar >= 2) FAIL(3 | (2 << 8));
                s = RING + nfar++;
                stage(ur, s);
              }
              sh[8 + np] = s;
              const int a = ur == last_r ? last_mpl : __builtin_amdgcn_readfirstlane(rmpl[s]);
              const int b = ur == last_r ? last_mpr : __builtin_amdgcn_readfirstlane(rmpr[s]);
              lo = imin(lo, a); hi = imax(hi, b);
            }
            ++np;
          }
          if (np > 8) FAIL(3 | (2 << 8));
        }
        int beg, end;
        if (r == 0) { beg = 0; end = w < L ? w : L; }
        else {
          beg = lo + 1 - w; if (beg < 0) beg = 0;
          end = hi + 1 + w; if (end > L) end = L;
          if (end - beg + 1 > 2 * w + 129) end = beg + 2 * w + 128;
        }
        const int width = end - beg + 1;
        if (width > RS || width > WS) FAIL(3 | (3 << 8));
        my_cells += (unsigned long long)width;
        if (lane == 0) {
          rbeg[slot] = beg; rend[slot] = end;
          prow0[r] = pd0; prow1[r] = pd1;
          if (end < L) hl[r] = PNEG;
          if (keep) { row_beg[r] = beg; row_end[r] = end; }
        }
        FP(0);
        const int rowo = r * WS;
        const int sb = slot * RST + G;
        if (lane < G) {   // -inf guard cells on both sides of the row: successors read them unchecked
          const int o1 = slot * RST + lane, o2 = sb + width + lane;
          rH[o1] = PNEG; rE1[o1] = PNEG; rE2[o1] = PNEG;
          rH[o2] = PNEG; rE1[o2] = PNEG; rE2[o2] = PNEG;
        }
        // fast rows: one or two predecessors whose stored band (plus guards) covers every column this row reads
        int pb0 = 0, pb1 = 0;
        bool fast = !slow && np >= 1;
        if (fast) {
          // (the predecessor is nearly always the previous row: its band is still in scalar registers)
          int pe0;
          if (np == 1 && (int)(pd0 & 255u) == 1 && last_r == r - 1) { pb0 = last_beg; pe0 = last_end; }
          else { pb0 = __builtin_amdgcn_readfirstlane(rbeg[ps0]); pe0 = __builtin_amdgcn_readfirstlane(rend[ps0]); }
          fast = beg - 1 >= pb0 - G && end <= pe0 + G;
          if (np == 2) {
            pb1 = __builtin_amdgcn_readfirstlane(rbeg[ps1]);
            fast = fast && beg - 1 >= pb1 - G && end <= __builtin_amdgcn_readfirstlane(rend[ps1]) + G;
          }
        }
        const int this_beg = beg, this_end = end;
#ifdef POA_COUNT_ROWS
        ROWCNT(7);
        if (slow) ROWCNT(4); else if (!fast) ROWCNT(3); else if (np == 2) ROWCNT(15);
        else if (C == 1 && !keep && width <= 64) ROWCNT(0); else ROWCNT(14);
        if (!slow && np == 1 && (pd0 & 255u) == 1 && !keep) { ROWCNT(5); if (width <= 64 * C) ROWCNT(8); if (last_beg == beg - 1) ROWCNT(9); if (last_beg == beg) ROWCNT(10); }
        if (!slow && np == 2 && ((pd0 & 255u) == 1 || ((pd0 >> 8) & 255u) == 1) && fast && !keep) ROWCNT(6);
        if (!slow && np == 1 && (pd0 & 255u) > 1 && fast && !keep) ROWCNT(11);
        if (keep) ROWCNT(12);
        if (width <= 64) ROWCNT(13);
#endif
        if (C == 1 && fast && np == 1 && !keep && width <= 64) {
          // ---- the common row, written out flat: one predecessor inside the ring whose band covers this one, one
          // column per lane, nothing to keep for later.  Same arithmetic and the same tie rules as the general code
          // below (which it mirrors line by line for C = 1, one chunk, km = 0), without its loops, carries and
          // per-cell validity branches.
          const int j = beg + lane;
          const bool valid = j <= end;
          const int scv = (j >= 1 && valid) ? pl_score(bv, q[j - 1]) : 0;
          const int bi0 = ps0 * RST + G + (j - 1 - pb0);
          const int32_t hvA = rH[bi0], hvB = rH[bi0 + 1], xa = rE1[bi0 + 1], xb = rE2[bi0 + 1];
          const int32_t m0 = hvA + scv;
          const int32_t a1 = hvB - P_O1 - P_E1, b1 = xa - P_E1;
          const int32_t e1v = imax(a1, b1);
          const uint32_t dE1v = a1 == e1v ? 0u : 8u;
          const int32_t a2 = hvB - P_O2 - P_E2, b2 = xb - P_E2;
          const int32_t e2v = imax(a2, b2);
          const uint32_t dE2v = a2 == e2v ? 0u : 8u;
          const int32_t hpv = valid ? imax(m0, imax(e1v, e2v)) : PNEG;
          // (the third scan: the row maximum is the maximum of H', attained where H' attains it -- see the chain rows)
          const int32_t s1 = wave_scan_max(hpv + j * P_E1, PNEG), s2 = wave_scan_max(hpv + j * P_E2, PNEG), s3 = wave_scan_max(hpv, PNEG);
          const int32_t f1 = wave_shr1(s1, PNEG) - P_O1 - j * P_E1, f2 = wave_shr1(s2, PNEG) - P_O2 - j * P_E2;
          const int32_t h = imax(hpv, imax(f1, f2));
          const int32_t hp_left = wave_shr1(hpv, PNEG);
          if (valid) {
            const uint32_t dH = m0 == h ? 0u : e1v == h ? 8u : e2v == h ? 9u : f1 == h ? 10u : 11u;
            const uint32_t dHp = m0 == hpv ? 0u : e1v == hpv ? 8u : 9u;
            const uint32_t o1 = hp_left - P_O1 - P_E1 == f1 ? 1u : 0u;
            const uint32_t o2 = hp_left - P_O2 - P_E2 == f2 ? 1u : 0u;
            gdir[rowo + (j & wm)] = dH | (dHp << 4) | (dE1v << 8) | (dE2v << 12) | (o1 << 16) | (o2 << 17);
            rH[sb + lane] = h; rE1[sb + lane] = e1v; rE2[sb + lane] = e2v;
            if (j == L) hl[r] = h;
          }
          const int32_t wmx = __builtin_amdgcn_readlane(s3, 63);
          const unsigned long long em = __ballot(valid && hpv == wmx);
          int l = beg + (int)__builtin_ctzll(em), rr = beg + 63 - (int)__builtin_clzll(em);
          if (wmx <= PNEG / 2) { l = beg; rr = end; }
          last_r = r; last_mpl = UNI(l); last_mpr = UNI(rr); last_beg = this_beg; last_end = this_end;
          if (lane == 0) { rmpl[slot] = l; rmpr[slot] = rr; }
          continue;
        }
        // leftmost / rightmost column of the row maximum, per lane (cells in increasing column order)
        int32_t lbest = -0x7fffffff - 1; int ll = -1, lr = -1;
        int32_t g1 = PNEG, g2 = PNEG, hpc = PNEG;   // carries between chunks of 64*C columns
        for (int j0 = beg; j0 <= end; j0 += 64 * C) {
          const int jb = j0 + lane * C;
          int32_t m[C], e1[C], e2[C];
          int km[C];
          uint32_t dE1[C], dE2[C];
          int sc[C];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int j = jb + c;
            sc[c] = (j >= 1 && j <= end) ? pl_score(bv, q[j - 1]) : 0;
          }
          if (fast) {
            // values below -inf/2 are "no path": they are never clamped here, only kept far below any score
            int32_t hv0[C + 1], xa0[C], xb0[C];
            const int bi0 = ps0 * RST + G + (jb - 1 - pb0);
#pragma unroll
            for (int t = 0; t <= C; ++t) hv0[t] = rH[bi0 + t];
#pragma unroll
            for (int t = 0; t < C; ++t) { xa0[t] = rE1[bi0 + 1 + t]; xb0[t] = rE2[bi0 + 1 + t]; }
            if (np == 1) {
#pragma unroll
              for (int c = 0; c < C; ++c) {
                m[c] = hv0[c] + sc[c]; km[c] = 0;
                const int32_t a1 = hv0[c + 1] - P_O1 - P_E1, b1 = xa0[c] - P_E1;
                e1[c] = imax(a1, b1); dE1[c] = a1 == e1[c] ? 0u : 8u;
                const int32_t a2 = hv0[c + 1] - P_O2 - P_E2, b2 = xb0[c] - P_E2;
                e2[c] = imax(a2, b2); dE2[c] = a2 == e2[c] ? 0u : 8u;
              }
            } else {
              int32_t hv1[C + 1], xa1[C], xb1[C];
              const int bi1 = ps1 * RST + G + (jb - 1 - pb1);
#pragma unroll
              for (int t = 0; t <= C; ++t) hv1[t] = rH[bi1 + t];
#pragma unroll
              for (int t = 0; t < C; ++t) { xa1[t] = rE1[bi1 + 1 + t]; xb1[t] = rE2[bi1 + 1 + t]; }
#pragma unroll
              for (int c = 0; c < C; ++c) {
                const int32_t m0 = hv0[c] + sc[c], m1 = hv1[c] + sc[c];
                m[c] = imax(m0, m1); km[c] = m1 > m0 ? 1 : 0;
                {
                  const int32_t a0 = hv0[c + 1] - P_O1 - P_E1, b0 = xa0[c] - P_E1, a1 = hv1[c + 1] - P_O1 - P_E1, b1 = xa1[c] - P_E1;
                  const int32_t e = imax(imax(a0, b0), imax(a1, b1));
                  e1[c] = e; dE1[c] = a0 == e ? 0u : a1 == e ? 1u : b0 == e ? 8u : 9u;
                }
                {
                  const int32_t a0 = hv0[c + 1] - P_O2 - P_E2, b0 = xb0[c] - P_E2, a1 = hv1[c + 1] - P_O2 - P_E2, b1 = xb1[c] - P_E2;
                  const int32_t e = imax(imax(a0, b0), imax(a1, b1));
                  e2[c] = e; dE2[c] = a0 == e ? 0u : a1 == e ? 1u : b0 == e ? 8u : 9u;
                }
              }
            }
          } else {
            int ko1[C], kx1[C], ko2[C], kx2[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
              m[c] = PNEG; e1[c] = PNEG; e2[c] = PNEG;
              km[c] = 15; ko1[c] = 15; kx1[c] = 15; ko2[c] = 15; kx2[c] = 15;
            }
            auto accum = [&](int k, int sl) {
              const int pb = __builtin_amdgcn_readfirstlane(rbeg[sl]), pe = __builtin_amdgcn_readfirstlane(rend[sl]);
              const int so = sl * RST + G;
              int32_t hv[C + 1], x1v[C], x2v[C];
#pragma unroll
              for (int t = 0; t <= C; ++t) {
                const int jj = jb - 1 + t;
                const bool ok = jj >= pb && jj <= pe && jj <= end;
                const int32_t val = rH[so + (ok ? jj - pb : 0)];
                hv[t] = ok ? val : PNEG;
              }
#pragma unroll
              for (int t = 0; t < C; ++t) {
                const int jj = jb + t;
                const bool ok = jj >= pb && jj <= pe && jj <= end;
                const int32_t v1 = rE1[so + (ok ? jj - pb : 0)], v2 = rE2[so + (ok ? jj - pb : 0)];
                x1v[t] = ok ? v1 : PNEG;
                x2v[t] = ok ? v2 : PNEG;
              }
#pragma unroll
              for (int c = 0; c < C; ++c) {
                const int32_t hm1 = hv[c], h = hv[c + 1];
                if (hm1 > PNEG / 2) { const int32_t x = hm1 + sc[c]; if (x > m[c]) { m[c] = x; km[c] = k; } }
                {
                  const int32_t a = h > PNEG / 2 ? h - P_O1 - P_E1 : PNEG, b = x1v[c] > PNEG / 2 ? x1v[c] - P_E1 : PNEG;
                  const int32_t cc = a > b ? a : b;
                  if (cc > PNEG / 2) {
                    if (cc > e1[c]) { e1[c] = cc; ko1[c] = 15; kx1[c] = 15; }
                    if (cc == e1[c]) { if (a == cc && ko1[c] == 15) ko1[c] = k; if (b == cc && kx1[c] == 15) kx1[c] = k; }
                  }
                }
                {
                  const int32_t a = h > PNEG / 2 ? h - P_O2 - P_E2 : PNEG, b = x2v[c] > PNEG / 2 ? x2v[c] - P_E2 : PNEG;
                  const int32_t cc = a > b ? a : b;
                  if (cc > PNEG / 2) {
                    if (cc > e2[c]) { e2[c] = cc; ko2[c] = 15; kx2[c] = 15; }
                    if (cc == e2[c]) { if (a == cc && ko2[c] == 15) ko2[c] = k; if (b == cc && kx2[c] == 15) kx2[c] = k; }
                  }
                }
              }
            };
            if (r == 0) {
#pragma unroll
              for (int c = 0; c < C; ++c) if (jb + c == 0) m[c] = 0;
            } else if (!slow) {
              if (np >= 1) accum(0, ps0);
              if (np >= 2) accum(1, ps1);
            } else {
#pragma unroll 1
              for (int k = 0; k < np; ++k) accum(k, __builtin_amdgcn_readfirstlane(sh[8 + k]));
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
              dE1[c] = ko1[c] != 15 ? (uint32_t)ko1[c] : (uint32_t)(8 + (kx1[c] & 7));
              dE2[c] = ko2[c] != 15 ? (uint32_t)ko2[c] : (uint32_t)(8 + (kx2[c] & 7));
            }
          }
          FP(1);
          // H' and the lane-serial half of the F prefix maxima
          int32_t hp[C], p1[C], p2[C];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int j = jb + c;
            int32_t x = imax(m[c], imax(e1[c], e2[c]));
            if (j > end) x = PNEG;
            hp[c] = x;
            const int32_t t1 = x + j * P_E1, t2 = x + j * P_E2;
            p1[c] = c ? imax(p1[c - 1], t1) : t1;
            p2[c] = c ? imax(p2[c - 1], t2) : t2;
          }
          const int32_t s1 = wave_scan_max(p1[C - 1], PNEG), s2 = wave_scan_max(p2[C - 1], PNEG);
          const int32_t X1 = imax(wave_shr1(s1, PNEG), g1), X2 = imax(wave_shr1(s2, PNEG), g2);
          g1 = imax(g1, __builtin_amdgcn_readlane(s1, 63));
          g2 = imax(g2, __builtin_amdgcn_readlane(s2, 63));
          FP(2);
          int32_t hp_prev = wave_shr1(hp[C - 1], PNEG);
          if (lane == 0) hp_prev = hpc;
          hpc = __builtin_amdgcn_readlane(hp[C - 1], 63);
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int j = jb + c;
            const int32_t x1 = c ? imax(X1, p1[c - 1]) : X1, x2 = c ? imax(X2, p2[c - 1]) : X2;
            const int32_t f1 = x1 - P_O1 - j * P_E1, f2 = x2 - P_O2 - j * P_E2;
            const int32_t h = imax(hp[c], imax(f1, f2));
            const int32_t hp_left = c ? hp[c - 1] : hp_prev;
            if (j <= end) {
              // the traceback's decisions, in the order the specification tries them
              const bool mk = km[c] != 15;
              const uint32_t dH = (m[c] == h && mk) ? (uint32_t)km[c] : e1[c] == h ? 8u : e2[c] == h ? 9u : f1 == h ? 10u : 11u;
              const uint32_t dHp = (m[c] == hp[c] && mk) ? (uint32_t)km[c] : e1[c] == hp[c] ? 8u : 9u;
              const uint32_t o1 = hp_left - P_O1 - P_E1 == f1 ? 1u : 0u;
              const uint32_t o2 = hp_left - P_O2 - P_E2 == f2 ? 1u : 0u;
              const int o = rowo + (j & wm);
              gdir[o] = dH | (dHp << 4) | (dE1[c] << 8) | (dE2[c] << 12) | (o1 << 16) | (o2 << 17);
              if (keep) { gH[o] = h; gE1[o] = e1[c]; gE2[o] = e2[c]; }
              rH[sb + (j - beg)] = h; rE1[sb + (j - beg)] = e1[c]; rE2[sb + (j - beg)] = e2[c];
              if (j == L) hl[r] = h;
              if (h > lbest) { lbest = h; ll = j; lr = j; }
              else if (h == lbest) lr = j;
            }
          }
        }
        FP(3);
        {
          // row maximum: leftmost / rightmost column over the lanes that own cells.  (No valid cell at all: the
          // specification leaves mpl at the first column and moves mpr to the last.)
          const bool own = ll >= 0;
          const int32_t wmx = __builtin_amdgcn_readlane(wave_scan_max(own ? lbest : -0x7fffffff - 1, -0x7fffffff - 1), 63);
          const bool eq = own && lbest == wmx;
          int l, rr;
          if (width <= 64 * C) {   // lane order = column order
            const unsigned long long em = __ballot(eq);
            l = __builtin_amdgcn_readlane(ll, (int)__builtin_ctzll(em));
            rr = __builtin_amdgcn_readlane(lr, 63 - (int)__builtin_clzll(em));
          } else {
            l = __builtin_amdgcn_readlane(wave_scan_min(eq ? ll : 0x7fffffff, 0x7fffffff), 63);
            rr = __builtin_amdgcn_readlane(wave_scan_max(eq ? lr : -1, -1), 63);
          }
          if (wmx <= PNEG / 2) { l = beg; rr = end; }
          last_r = r; last_mpl = UNI(l); last_mpr = UNI(rr); last_beg = this_beg; last_end = this_end;
          if (lane == 0) {
            rmpl[slot] = l; rmpr[slot] = rr;
            if (keep) { row_mpl[r] = l; row_mpr[r] = rr; }
          }
        }
        FP(4);
      }
      __syncthreads();   // direction words, deltas and end cells are in HBM
      PROF_ADD(1);
      // --------------------------------------------------------- traceback
      {
        int bu = -1; int32_t bsc = PNEG;
        for (int e = in_head[1]; e >= 0; e = e_next_in[e]) {
          const int ur = index[e_from[e]];
          const int32_t h = hl[ur];
          if (h > bsc) { bsc = h; bu = ur; }
        }
        nops = -1;
        if (bu >= 0 && bsc > PNEG / 2) {
          nops = 0;
          int r = __builtin_amdgcn_readfirstlane(bu), j = L, st = 0;   // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 H'
          int r0 = -(1 << 28), j0 = 0;
          uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0, P0 = 0, P1 = 0;
          while (r != 0 || j > 0) {
            r = UNI(r); j = UNI(j); st = UNI(st); nops = UNI(nops); r0 = UNI(r0); j0 = UNI(j0);
            if (nops + j + 2 > opcap) { nops = -1; break; }   // cannot happen on a valid path
            if (r == 0) {   // only inserted bases remain
              for (int t = lane; t < j; t += 64) { op_node[nops + t] = -1; op_q[nops + t] = j - 1 - t; }
              nops += j; j = 0;
              break;
            }
            int k = r0 - r, d = k - (j0 - j);
            if (k < 0 || k > 63 || d < 0 || d > 3) {
              r0 = r; j0 = j; k = 0; d = 0;
              const int rr = r0 - lane;
              if (rr >= 0) {
                const int bp = rr * WS;
                const int c = j0 - lane;
                W0 = gdir[bp + (c & wm)]; W1 = gdir[bp + ((c + 1) & wm)]; W2 = gdir[bp + ((c + 2) & wm)];
                W3 = gdir[bp + ((c + 3) & wm)];
                P0 = prow0[rr]; P1 = prow1[rr];
              }
              // wait for the window here, not at the join below (where the wait would also cover the op
              // stores of every step: vmcnt is in-order)
              asm volatile("" : "+v"(W0), "+v"(W1), "+v"(W2), "+v"(W3), "+v"(P0), "+v"(P1));
            }
            const uint32_t wsel = d == 0 ? W0 : d == 1 ? W1 : d == 2 ? W2 : W3;
            if (st == 0) {
              // a run of plain diagonal steps -- a match through predecessor slot 0, which is the row above -- all at
              // once: the cells of the window's current diagonal that are such steps form a ballot, the run is its
              // ones from lane k on, and lane i stores the i-th operation of the run (most of an alignment is this)
              const bool diag = (wsel & 15u) == 0u && (P0 & 255u) == 1u && r0 - lane >= 1;
              const unsigned long long dm = __ballot(diag) >> k;
              int run = dm == ~0ull ? 64 : (int)__builtin_ctzll(~dm);
              if (run > j) run = j;
              if (run > 0) {
                if (lane < run) { op_node[nops + lane] = r - lane; op_q[nops + lane] = j - 1 - lane; }
                nops += run; r -= run; j -= run;
                continue;
              }
            }
            const uint32_t dw = __builtin_amdgcn_readlane(wsel, k);
            const uint32_t p0 = __builtin_amdgcn_readlane(P0, k), p1 = __builtin_amdgcn_readlane(P1, k);
            int s = -1;   // predecessor slot to follow
            if (st == 0 || st == 5) {
              const uint32_t dd = st == 0 ? (dw & 15u) : ((dw >> 4) & 15u);
              if (dd < 8) {
                if (lane == 0) { op_node[nops] = r; op_q[nops] = j - 1; }
                ++nops; --j; st = 0; s = (int)dd;
              } else st = (int)dd - 7;   // 8 -> E1, 9 -> E2, 10 -> F1, 11 -> F2
            } else if (st == 1 || st == 2) {
              const uint32_t dd = st == 1 ? ((dw >> 8) & 15u) : ((dw >> 12) & 15u);
              if (lane == 0) { op_node[nops] = r; op_q[nops] = -1; }
              ++nops; s = (int)(dd & 7u);
              if (dd < 8) st = 0;
            } else {
              const uint32_t open = st == 3 ? ((dw >> 16) & 1u) : ((dw >> 17) & 1u);
              if (lane == 0) { op_node[nops] = -1; op_q[nops] = j - 1; }
              ++nops;
              if (open) st = 5;
              --j;
            }
            if (s >= 0) {
              const uint32_t dl = ((s < 4 ? p0 : p1) >> (8 * (s & 3))) & 255u;
              if (dl != 255u) r -= (int)dl;
              else {
                int e = in_head[order[r]];
                for (int t = 0; t < s; ++t) e = e_next_in[e];
                r = __builtin_amdgcn_readfirstlane(index[e_from[e]]);
              }
            }
          }
        }
      }
      __syncthreads();
      PROF_ADD(2);
    }
    if (nops < 0) FAIL(3 | (4 << 8));
    if (nops > 32000) FAIL(3 | (5 << 8));   // (path positions are packed into 15 bits of a scan key)
    // ------------------------------------------------------- graph update
    const int ncols = __builtin_amdgcn_readfirstlane(sh[2]), n_old = N;
    for (int c = lane; c < ncols; c += 64) scr[c] = 0;
    __syncthreads();
    int carry_c = 0, carry_n = 0, carry_key = 0;
    for (int p0 = 0; p0 < nops; p0 += 64) {
      const int p = p0 + lane;
      const bool valid = p < nops;
      int row = -1, j = -1;
      if (valid) { row = op_node[nops - 1 - p]; j = op_q[nops - 1 - p]; }
      const bool ali = valid && row >= 0 && j >= 0, ins = valid && row < 0;
      const int v = ali ? order[row] : -1;
      const int qb = (ali || ins) ? (int)q[j] : 0;
      int use = -1;
      bool isnew = ins;
      if (ali) {
        if (base[v] == qb) use = v;
        else { const int a = aln[5 * v + qb]; if (a >= 0) use = a; else isnew = true; }
      }
      const int inc_c = wave_scan_add((ali || ins) ? 1 : 0), inc_n = wave_scan_add(isnew ? 1 : 0);
      const int cidx = carry_c + inc_c - ((ali || ins) ? 1 : 0);
      const int nrank = carry_n + inc_n - (isnew ? 1 : 0);
      const int key = ali ? (((cidx + 1) << 16) | col[v]) : 0;     // (columns and path positions < 65536)
      const int inc_k = wave_scan_max(key, 0);
      const int ikey = imax(carry_key, inc_k);
      carry_c += __builtin_amdgcn_readlane(inc_c, 63);
      carry_n += __builtin_amdgcn_readlane(inc_n, 63);
      carry_key = imax(carry_key, __builtin_amdgcn_readlane(inc_k, 63));
      uint32_t aux = 0xFFFFFFFFu;
      if (isnew && n_old + nrank < nc) {   // (capacity is checked after the loop)
        const int nid = n_old + nrank;
        use = nid;
        base[nid] = qb;
        out_head[nid] = -1; in_head[nid] = -1;
        if (ali) {   // a new base at the column of v: joins v's aligned group
          for (int b = 0; b < 5; ++b) {
            const int sib = aln[5 * v + b];
            aln[5 * nid + b] = sib;
            if (sib >= 0) aln[5 * sib + qb] = nid;
          }
          aln[5 * nid + qb] = nid;
          col[nid] = col[v];
        } else {     // an inserted base: a new column, the t-th after its anchor's
          for (int b = 0; b < 5; ++b) aln[5 * nid + b] = b == qb ? nid : -1;
          col[nid] = COL_NEW;
          const int ac = ikey & 0xffff, t = (cidx + 1) - (ikey >> 16);
          atomicMax(&scr[ac], t);
          aux = (uint32_t)ac | ((uint32_t)t << 16);
        }
      }
      if (ali || ins) { path_use[cidx] = use; path_aux[cidx] = aux; }
    }
    const int PC = carry_c, n_new = n_old + carry_n;
    // the graph outgrew its allocation: redo this cluster with the HBM kernel
    if (n_new > nc || ncols + carry_n > nc || ncols + carry_n > 65000 || PC > 65000) FAIL(3 | (5 << 8));
    __syncthreads();
    // one edge per path step; no other lane touches the out-list of u or the in-list of v
    for (int t = lane; t <= PC; t += 64) {
      const int u = t == 0 ? 0 : path_use[t - 1], v = t == PC ? 1 : path_use[t];
      int tail = -1, e = out_head[u];
      bool found = false;
      for (; e >= 0; e = e_next_out[e]) {
        if (e_to[e] == v) { e_w[e]++; found = true; break; }
        tail = e;
      }
      if (found) continue;
      const int ne = atomicAdd(&sh[1], 1);
      if (ne >= ec) continue;   // out of edge slots: sh[1] > ec below gives the cluster up
      e_from[ne] = u; e_to[ne] = v; e_w[ne] = 1;
      e_next_out[ne] = -1; e_next_in[ne] = -1;
      if (tail < 0) out_head[u] = ne; else e_next_out[tail] = ne;
      int ie = in_head[v];
      if (ie < 0) in_head[v] = ne;
      else {
        while (e_next_in[ie] >= 0) ie = e_next_in[ie];
        e_next_in[ie] = ne;
      }
    }
    __syncthreads();
    if (sh[1] > ec) FAIL(3 | (5 << 8));
    // column ranks: every column moves right by the number of columns inserted before it
    int carry = 0;
    for (int c0 = 0; c0 < ncols; c0 += 64) {
      const int c = c0 + lane;
      const int x = c < ncols ? scr[c] : 0;
      const int inc = wave_scan_add(x);
      if (c < ncols) scr[c] = carry + inc - x;
      carry += __builtin_amdgcn_readlane(inc, 63);
    }
    __syncthreads();
    for (int v = lane; v < n_new; v += 64) {
      const int cv = col[v];
      if (cv < COL_SINK) col[v] = cv + scr[cv];
    }
    __syncthreads();
    for (int t = lane; t < PC; t += 64) {
      const uint32_t aux = path_aux[t];
      if (aux != 0xFFFFFFFFu) { const int ac = (int)(aux & 0xffffu); col[path_use[t]] = ac + scr[ac] + (int)(aux >> 16); }
    }
    const int ncols_new = ncols + carry;
    __syncthreads();
    // counting sort of the nodes by column = a topological order; the sink goes last
    for (int c = lane; c < ncols_new; c += 64) scr[c] = 0;
    __syncthreads();
    for (int v = lane; v < n_new; v += 64) if (v != 1) atomicAdd(&scr[col[v]], 1);
    __syncthreads();
    carry = 0;
    for (int c0 = 0; c0 < ncols_new; c0 += 64) {
      const int c = c0 + lane;
      const int x = c < ncols_new ? scr[c] : 0;
      const int inc = wave_scan_add(x);
      if (c < ncols_new) scr[c] = carry + inc - x;
      carry += __builtin_amdgcn_readlane(inc, 63);
    }
    __syncthreads();
    for (int v = lane; v < n_new; v += 64) if (v != 1) {
      const int pos = atomicAdd(&scr[col[v]], 1);
      order[pos] = v; index[v] = pos;
    }
    if (lane == 0) { order[n_new - 1] = 1; index[1] = n_new - 1; sh[0] = n_new; sh[2] = ncols_new; }
    __syncthreads();
    PROF_ADD(3);
  }
  // the heaviest-bundle consensus is poa_bundle_kernel's job: hand over the number of graph rows
  if (lane == 0) {
    cons_len[blockIdx.x] = sh[0];
    status[blockIdx.x] = 0;
    atomicAdd(cells, my_cells);
    for (int k = 0; k < 5; ++k) atomicAdd(&g_poaw_prof[k], prof[k]);
#ifdef POA_COUNT_ROWS
    for (int k = 0; k < 16; ++k) atomicAdd(&g_poaw_rows[k], (unsigned long long)rowcnt[k]);
#endif
#ifdef POA_FINE_PROF
    for (int k = 0; k < 6; ++k) printf("fp%d %lld\n", k, fp[k]);
#endif
  }
#undef FAIL
}

// ----------------------------------------------------------- heaviest bundle (consensus), one wavefront per sub-cluster
// in: cons_len[task] = number of graph rows N left by poa_wave_kernel (status 0); out: the consensus and its length
__global__ void __launch_bounds__(64) poa_bundle_kernel(const PoaWaveTask* tasks, int32_t* ws32, uint8_t* ws8, int32_t* cons_len,
                                                       const int32_t* status) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (status[blockIdx.x] != 0) return;
  const PoaWaveTask T = tasks[blockIdx.x];
  const int lane = threadIdx.x;
  const int nc = T.nc;
  const int N = cons_len[blockIdx.x];
  if (T.n_seqs <= 0 || N <= 0) return;   // (an empty sub-cluster already has length 0)
  const WsLayout wl = ws_layout(nc, T.ec, T.max_len, T.ws);
  int32_t* W = ws32 + T.ws_off;
  const int32_t *out_head = W + wl.out_head, *order = W + wl.order, *index = W + wl.index, *base = W + wl.base;
  const int32_t *e_to = W + wl.e_to, *e_w = W + wl.e_w, *e_next_out = W + wl.e_next_out;
  uint8_t* cons = ws8 + T.cons_off;
  // per row: up to two successors (row, weight) and the base, staged in LDS; rows with more go to the graph
  uint32_t* succ0 = (uint32_t*)smem;
  uint32_t* succ1 = succ0 + nc;
  int32_t* score = (int32_t*)(succ1 + nc);
  for (int r = lane; r < N; r += 64) {
    const int v = order[r];
    uint32_t s0 = 0xFFFFu, s1 = 0xFFFFu;   // row 0xFFFF: none; weight 0x1FFF in s1: more than two, walk the list
    int k = 0;
    for (int e = out_head[v]; e >= 0; e = e_next_out[e]) {
      const uint32_t ent = (uint32_t)index[e_to[e]] | ((uint32_t)e_w[e] << 16);
      if (k == 0) s0 = ent; else if (k == 1) s1 = ent;
      ++k;
    }
    if (k > 2) s1 = 0xFFFFu | (0x1FFFu << 16);
    succ0[r] = s0 | ((uint32_t)base[v] << 29);
    succ1[r] = s1;
  }
  __syncthreads();
  if (lane == 0) {
    for (int r = N - 1; r >= 0; --r) {
      const uint32_t s0 = succ0[r], s1 = succ1[r];
      int bst = -1, bw = -1; int32_t bsc = -1;
      if ((s1 & 0xFFFFu) == 0xFFFFu && ((s1 >> 16) & 0x1FFFu) == 0x1FFFu) {
        for (int e = out_head[order[r]]; e >= 0; e = e_next_out[e]) {
          const int x = index[e_to[e]], wgt = e_w[e];
          const int32_t sx = score[x];
          if (wgt > bw || (wgt == bw && sx > bsc)) { bw = wgt; bsc = sx; bst = x; }
        }
      } else {
        if ((s0 & 0xFFFFu) != 0xFFFFu) { bst = (int)(s0 & 0xFFFFu); bw = (int)((s0 >> 16) & 0x1FFFu); bsc = score[bst]; }
        if ((s1 & 0xFFFFu) != 0xFFFFu) {
          const int x = (int)(s1 & 0xFFFFu), wgt = (int)((s1 >> 16) & 0x1FFFu);
          const int32_t sx = score[x];
          if (wgt > bw || (wgt == bw && sx > bsc)) { bw = wgt; bsc = sx; bst = x; }
        }
      }
      succ1[r] = (uint32_t)bst;
      score[r] = bst >= 0 ? bw + bsc : 0;
    }
    int len = 0;
    const int sink_row = N - 1;
    for (int r = (int)succ1[0]; r >= 0 && r != sink_row; r = (int)succ1[r]) cons[len++] = (uint8_t)(succ0[r] >> 29);
    cons_len[blockIdx.x] = len;
  }
}

size_t poa_wave_lds_bytes(int nc, int max_len, int rs, int ring) {
  (void)nc;
  const size_t ns = (size_t)ring + 2;
  return 12 * ns * ((size_t)rs + 8) + 16 * ns + 64 + (((size_t)max_len + 15) & ~(size_t)15) + 64 + 256;
}

size_t poa_bundle_lds_bytes(int nc) { return 12 * (size_t)nc + 64; }

int64_t poa_wave_ws_ints(int nc, int ec, int max_len, int ws) { return ws_layout(nc, ec, max_len, ws).total; }

template <int C>
static hipError_t launch_c(const PoaWaveTask* d_tasks, int n_tasks, size_t lds_bytes, const uint8_t* d_seqs,
                           const int64_t* d_seq_off, int32_t* ws32, uint8_t* ws8, int32_t* d_len, int32_t* d_status,
                           unsigned long long* d_cells, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute((const void*)poa_wave_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(poa_wave_kernel<C>, dim3((unsigned)n_tasks), dim3(64), lds_bytes, stream, d_tasks, d_seqs, d_seq_off, ws32, ws8,
                     d_len, d_status, d_cells);
  return hipGetLastError();
}

hipError_t poa_wave_launch(int cols, const PoaWaveTask* d_tasks, int n_tasks, size_t lds_bytes, size_t bundle_lds_bytes,
                           const uint8_t* d_seqs, const int64_t* d_seq_off, int32_t* ws32, uint8_t* ws8, int32_t* d_len,
                           int32_t* d_status, unsigned long long* d_cells, hipStream_t stream) {
  hipError_t e;
  if (cols == 1) e = launch_c<1>(d_tasks, n_tasks, lds_bytes, d_seqs, d_seq_off, ws32, ws8, d_len, d_status, d_cells, stream);
  else if (cols == 2) e = launch_c<2>(d_tasks, n_tasks, lds_bytes, d_seqs, d_seq_off, ws32, ws8, d_len, d_status, d_cells, stream);
  else if (cols == 3) e = launch_c<3>(d_tasks, n_tasks, lds_bytes, d_seqs, d_seq_off, ws32, ws8, d_len, d_status, d_cells, stream);
  else if (cols == 5) e = launch_c<5>(d_tasks, n_tasks, lds_bytes, d_seqs, d_seq_off, ws32, ws8, d_len, d_status, d_cells, stream);
  else return hipErrorInvalidValue;
  if (e != hipSuccess) return e;
  return poa_bundle_launch(d_tasks, n_tasks, bundle_lds_bytes, ws32, ws8, d_len, (const int32_t*)d_status, stream);
}

hipError_t poa_bundle_launch(const PoaWaveTask* d_tasks, int n_tasks, size_t bundle_lds_bytes, int32_t* ws32, uint8_t* ws8, int32_t* d_len,
                             const int32_t* d_status, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute((const void*)poa_bundle_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bundle_lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(poa_bundle_kernel, dim3((unsigned)n_tasks), dim3(64), bundle_lds_bytes, stream, d_tasks, ws32, ws8, d_len, d_status);
  return hipGetLastError();
}

// SVDSS_DEBUG: in-kernel phase timers summed over the sub-clusters run since the last call
void poa_wave_debug_report() {
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_poaw_prof), sizeof h) != hipSuccess) return;
  fprintf(stderr, "[poa_wave] 100MHz ticks (sum over clusters): prepare %llu forward %llu traceback %llu update %llu bundle %llu\n",
          h[0], h[1], h[2], h[3], h[4]);
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_poaw_prof), h, sizeof h);
#ifdef POA_COUNT_ROWS
  unsigned long long rc[16];
  if (hipMemcpyFromSymbol(rc, HIP_SYMBOL(g_poaw_rows), sizeof rc) != hipSuccess) return;
  fprintf(stderr, "[poa_wave] rows: chain %llu in %llu chains (time: the 'bundle' figure above) | other rows %llu: flat %llu fast1 %llu fast2 %llu general %llu slow %llu | "
          "of them chain-eligible %llu (fits lanes %llu, band +1 %llu, +0 %llu) np2-with-prev %llu np1-far %llu keep %llu width<=64 %llu\n", rc[1], rc[2], rc[7],
          rc[0], rc[14], rc[15], rc[3], rc[4], rc[5], rc[8], rc[9], rc[10], rc[6], rc[11], rc[12], rc[13]);
  memset(rc, 0, sizeof rc);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_poaw_rows), rc, sizeof rc);
#endif
}
