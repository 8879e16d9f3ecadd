// poa_lds.hip -- LDS-resident POA consensus kernel (fast path of svdss_poa_consensus_batch).
//
// Same specification as poa.hip / oracle/svdss_oracle_poa.c, bit for bit.  What changes is where
// things live and who does the work.  POA over a handful of reads is a chain of short dependent
// steps (graph rows, traceback steps, edge updates); with the graph in HBM and one lane doing the
// serial parts, every step costs a ~1 us round trip and the kernel crawls at ~1 GCUPS.  Here one
// workgroup of four wavefronts owns one sub-cluster and keeps in LDS
//   * the whole graph with 16-bit node/edge indices (~25 B per node),
//   * the read being aligned,
//   * a ring of the last `ring` DP rows (H, E1, E2) -- the predecessors of a row are almost always
//     among them,
// while HBM only receives write-once streams: per-cell *direction words* that encode every decision
// the traceback will take (so the traceback never compares scores), the predecessor-row deltas of
// each row, and a copy of H/E1/E2 for the rare predecessor that left the ring (long deletion edges).
//
//   forward    one DP row per step, one column per thread (4 waves = 256 columns, wider rows loop);
//              the F recurrences are max-scans done with DPP inside a wave and one LDS exchange
//              across the four waves; two barriers per row.
//   traceback  wave 0.  A register window holds the direction words of 64 rows x 4 columns along the
//              current diagonal (one global latency per ~30-60 steps instead of one per step); the
//              walk itself is scalar.
//   update     parallel over the alignment: block scans number the surviving path elements and the new
//              nodes; every path edge touches edge lists no other edge touches, so edges are added
//              concurrently.
//   order      any topological order gives the same DP values, traceback (predecessor slots are edge
//              order) and consensus.  Instead of Kahn's serial queue the kernel keeps a column rank
//              per node (aligned nodes share a column, a read's path is column-monotone, inserted
//              bases open new columns right after their anchor) and rebuilds the order with a
//              counting sort by column: scans and atomics only.
// Clusters that do not fit (graph beyond the LDS budget, > 8 predecessors on a node, band fallback to
// the full matrix) report status 3 and are redone by the HBM kernel of poa.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "poa_lds.h"

#define PNEG (-0x20000000)
#define P_O1 4
#define P_E1 2
#define P_O2 24
#define P_E2 1
#define P_MATCH 2
#define P_MISMATCH 4
#define NIL 0xFFFFu
#define COL_SINK 0xFFFEu
#define COL_NEW 0xFFFFu
#define PT 256

typedef uint16_t u16;

struct LGraph {
  u16 *out_head, *in_head, *order, *index, *col;
  u16 *e_from, *e_to, *e_w, *e_next_out, *e_next_in;
  uint8_t *base, *q;
};

__device__ __forceinline__ int pl_score(int a, int b) { return (a >= 4 || b >= 4) ? 0 : (a == b ? P_MATCH : -P_MISMATCH); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

template <int CTRL, int RMASK>
__device__ __forceinline__ int dppi(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, RMASK, 0xf, false);
}

// inclusive max-scan over the 64 lanes of a wave: four row_shr steps, then row_bcast15 / row_bcast31
__device__ __forceinline__ int wave_scan_max(int x) {
  x = imax(x, dppi<0x111, 0xf>(PNEG, x));
  x = imax(x, dppi<0x112, 0xf>(PNEG, x));
  x = imax(x, dppi<0x114, 0xf>(PNEG, x));
  x = imax(x, dppi<0x118, 0xf>(PNEG, x));
  x = imax(x, dppi<0x142, 0xa>(PNEG, x));
  x = imax(x, dppi<0x143, 0xc>(PNEG, x));
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

// barrier that orders LDS traffic only: __syncthreads() would also wait for every outstanding HBM
// store (s_waitcnt vmcnt(0), ~1-2 us) -- the DP rows stream direction words out on every row
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// value of the lane below (lane 0 receives `fill`)
__device__ __forceinline__ int wave_shr1(int x, int fill) { return dppi<0x138, 0xf>(fill, x); }

// exclusive sum over the 256 threads; *total = sum of all.  bs: 8 ints of LDS, toggled between calls
__device__ __forceinline__ int block_excl_sum(int x, int* bs, int& tog, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int inc = wave_scan_add(x);
  int* b = bs + 4 * tog;
  tog ^= 1;
  if (lane == 63) b[wv] = inc;
  lds_barrier();
  const int t0 = b[0], t1 = b[1], t2 = b[2], t3 = b[3];
  *total = t0 + t1 + t2 + t3;
  return inc - x + (wv > 0 ? t0 : 0) + (wv > 1 ? t1 : 0) + (wv > 2 ? t2 : 0);
}

// inclusive max over the 256 threads (values >= 0)
__device__ __forceinline__ int block_incl_max(int x, int* bs, int& tog, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = x;
  inc = imax(inc, dppi<0x111, 0xf>(0, inc));
  inc = imax(inc, dppi<0x112, 0xf>(0, inc));
  inc = imax(inc, dppi<0x114, 0xf>(0, inc));
  inc = imax(inc, dppi<0x118, 0xf>(0, inc));
  inc = imax(inc, dppi<0x142, 0xa>(0, inc));
  inc = imax(inc, dppi<0x143, 0xc>(0, inc));
  int* b = bs + 4 * tog;
  tog ^= 1;
  if (lane == 63) b[wv] = inc;
  lds_barrier();
  const int t0 = b[0], t1 = b[1], t2 = b[2], t3 = b[3];
  *total = imax(imax(t0, t1), imax(t2, t3));
  if (wv > 0) inc = imax(inc, t0);
  if (wv > 1) inc = imax(inc, t1);
  if (wv > 2) inc = imax(inc, t2);
  return inc;
}

// one edge of the new path; no other thread touches the out-list of u or the in-list of v
__device__ __forceinline__ void lg_add_edge_par(LGraph& g, int u, int v, int* n_edges, int ec) {
  unsigned tail = NIL;
  for (unsigned e = g.out_head[u]; e != NIL; e = g.e_next_out[e]) {
    if (g.e_to[e] == v) { g.e_w[e]++; return; }
    tail = e;
  }
  const int ne = atomicAdd(n_edges, 1);
  if (ne >= ec) return;   // out of edge slots: the caller sees n_edges > ec and gives the cluster up
  g.e_from[ne] = (u16)u; g.e_to[ne] = (u16)v; g.e_w[ne] = 1;
  g.e_next_out[ne] = NIL; g.e_next_in[ne] = NIL;
  if (tail == NIL) g.out_head[u] = (u16)ne; else g.e_next_out[tail] = (u16)ne;
  unsigned e = g.in_head[v];
  if (e == NIL) g.in_head[v] = (u16)ne;
  else {
    while (g.e_next_in[e] != NIL) e = g.e_next_in[e];
    g.e_next_in[e] = (u16)ne;
  }
}

// direction word layout
//  bits 0-3  source of H : 0-7 match through predecessor slot k, 8 E1, 9 E2, 10 F1, 11 F2
//  bits 4-7  source of H': 0-7 match through slot k, 8 E1, 9 E2
//  bits 8-11 E1: 0-7 opened from H of slot k, 8-15 extended from E1 of slot k-8;  bits 12-15 E2 likewise
//  bit 16    F1 opened from H'(v, j-1) (else extended);  bit 17 F2 likewise

__device__ unsigned long long g_poa_prof[8];   // SVDSS_DEBUG: time in forward, traceback, update, bundle
#define PROF_T() (prof_t = wall_clock64())
#define PROF_ADD(k) do { const unsigned long long t_ = wall_clock64(); prof[k] += t_ - prof_t; prof_t = t_; } while (0)

__global__ void __launch_bounds__(PT) poa_lds_kernel(const PoaLdsTask* tasks, const uint8_t* seqs, const int64_t* seq_off,
                                                    int32_t* ws32, uint8_t* ws8, int32_t* cons_len, int32_t* status,
                                                    unsigned long long* cells) {
  extern __shared__ __align__(16) unsigned char smem[];
  const PoaLdsTask T = tasks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nc = T.nc, ec = T.ec, WS = T.ws, wm = WS - 1, RING = T.ring, rm = RING - 1;
  LGraph g;
  u16* p16 = (u16*)smem;
  g.out_head = p16; g.in_head = p16 + nc; g.order = p16 + 2 * nc; g.index = p16 + 3 * nc; g.col = p16 + 4 * nc;
  u16* pe = p16 + 5 * nc;
  g.e_from = pe; g.e_to = pe + ec; g.e_w = pe + 2 * ec; g.e_next_out = pe + 3 * ec; g.e_next_in = pe + 4 * ec;
  g.base = (uint8_t*)(pe + 5 * ec);
  g.q = g.base + nc;
  size_t off = (size_t)(g.q - smem) + (size_t)T.max_len;
  off = (off + 15) & ~(size_t)15;
  const int NS = RING + 2;                  // ring slots + 2 staging slots for rows read back from HBM
  int32_t* rH = (int32_t*)(smem + off); off += sizeof(int32_t) * (size_t)NS * WS;
  int32_t* rE1 = (int32_t*)(smem + off); off += sizeof(int32_t) * (size_t)NS * WS;
  int32_t* rE2 = (int32_t*)(smem + off); off += sizeof(int32_t) * (size_t)NS * WS;
  int32_t* scr = rH;                       // 3*NS*WS ints of scratch outside the forward pass
  const int scr_n = 3 * NS * WS;
  int32_t* rbeg = (int32_t*)(smem + off); off += sizeof(int32_t) * NS;     // per slot: band and the columns
  int32_t* rend = (int32_t*)(smem + off); off += sizeof(int32_t) * NS;     // of the row maximum
  int32_t* rmpl = (int32_t*)(smem + off); off += sizeof(int32_t) * NS;
  int32_t* rmpr = (int32_t*)(smem + off); off += sizeof(int32_t) * NS;
  int32_t* part = (int32_t*)(smem + off); off += sizeof(int32_t) * RING * 8;   // [slot][wave]{max, l | r<<16}
  int32_t* xch = (int32_t*)(smem + off); off += sizeof(int32_t) * 32;          // [2][4]{s1, s2, hp, -}
  int32_t* bs = (int32_t*)(smem + off); off += sizeof(int32_t) * 8;
  int32_t* sh = (int32_t*)(smem + off);   // 0 nodes, 1 edges, 2 columns, 3 nops/flag

  int32_t* row_beg = ws32 + T.row_off;
  int32_t* row_end = row_beg + nc;
  int32_t* hl = row_beg + 2 * nc;            // H(row, L) for the end cell
  uint32_t* prow = (uint32_t*)(row_beg + 3 * nc);   // 2 words per row: predecessor-row deltas, 8 bits per slot
  int32_t* row_mpl = row_beg + 5 * nc;
  int32_t* row_mpr = row_beg + 6 * nc;
  const int64_t pool = (int64_t)nc * WS;
  int32_t* gH = ws32 + T.dp_off;
  int32_t* gE1 = gH + pool;
  int32_t* gE2 = gH + 2 * pool;
  uint32_t* gdir = (uint32_t*)(gH + 3 * pool);
  int32_t* aln = ws32 + T.aln_off;
  const int opcap = nc + T.max_len + 4;
  int32_t* op_node = ws32 + T.op_off;
  int32_t* op_q = op_node + opcap;
  int32_t* path_use = op_node + 2 * opcap;
  uint32_t* path_aux = (uint32_t*)(op_node + 3 * opcap);
  uint8_t* cons = ws8 + T.cons_off;
  const int n = (int)T.n_seqs;
  unsigned long long my_cells = 0;
  unsigned long long prof[4] = {0, 0, 0, 0}, prof_t;
#ifdef POA_FINE_PROF
  long long fp[6] = {0, 0, 0, 0, 0, 0}, ft = 0;
#define FP(k) do { const long long t_ = clock64(); fp[k] += t_ - ft; ft = t_; } while (0)
#else
#define FP(k)
#endif
  int tog = 0;
  if (n <= 0) { if (tid == 0) { cons_len[blockIdx.x] = 0; status[blockIdx.x] = 0; } return; }
#define FAIL(code) do { if (tid == 0) status[blockIdx.x] = (code); return; } while (0)
  // ------------------------------------------------------------------ graph of the first read
  {
    const uint8_t* q0 = seqs + seq_off[T.seq_first];
    const int L0 = (int)(seq_off[T.seq_first + 1] - seq_off[T.seq_first]);
    if (L0 + 2 > nc || L0 + 1 > ec || L0 + 1 > scr_n) FAIL(3 | (1 << 8));
    for (int v = tid; v < L0 + 2; v += PT) {
      const int b = v < 2 ? 4 : q0[v - 2];
      g.base[v] = (uint8_t)b;
      g.out_head[v] = NIL; g.in_head[v] = NIL;
      for (int x = 0; x < 5; ++x) aln[5 * v + x] = -1;
      if (v >= 2) aln[5 * v + b] = v;
      const int idx = v == 0 ? 0 : v == 1 ? L0 + 1 : v - 1;
      g.col[v] = v == 1 ? (u16)COL_SINK : (u16)idx;
      g.index[v] = (u16)idx;
      g.order[idx] = (u16)v;
    }
    __syncthreads();
    for (int e = tid; e <= L0; e += PT) {
      const int from = e == 0 ? 0 : e + 1, to = e == L0 ? 1 : e + 2;
      g.e_from[e] = (u16)from; g.e_to[e] = (u16)to; g.e_w[e] = 1;
      g.e_next_out[e] = NIL; g.e_next_in[e] = NIL;
      g.out_head[from] = (u16)e; g.in_head[to] = (u16)e;
    }
    if (tid == 0) { sh[0] = L0 + 2; sh[1] = L0 + 1; sh[2] = L0 + 1; }
    __syncthreads();
  }
  for (int i = 1; i < n; ++i) {
    const uint8_t* qg = seqs + seq_off[T.seq_first + i];
    const int L = (int)(seq_off[T.seq_first + i + 1] - seq_off[T.seq_first + i]);
    const int N = sh[0];
    if (L > T.max_len) FAIL(3 | (6 << 8));
    PROF_T();
    for (int j = tid; j < L; j += PT) g.q[j] = qg[j];
    __syncthreads();
    int nops = -1;
    // banded first; if the band loses the sink the read is aligned again with the full matrix (w = L)
    for (int attempt = 0; attempt < 2 && nops < 0; ++attempt) {
    const int w = attempt ? L : 10 + (int)(0.01 * L);
    int last_r = -1, last_mpl = 0, last_mpr = 0;
    int par = 0;
    // ------------------------------------------------------------ forward (the sink is order[N-1])
    // the first predecessor of the next row is looked up while the current row is computed
    int v_n = __builtin_amdgcn_readfirstlane(g.order[0]);
    unsigned e_n = NIL, en_n = NIL, f_n = 0;
    int i_n = 0;
    for (int r = 0; r < N - 1; ++r) {
      FP(5);
      const int v = v_n;
      const int slot = r & rm;
      const int v_nx = g.order[r + 1];   // (consumed after the first barrier)
      int ps[8];          // LDS slot of each predecessor row (ring, or staging for rows that left it)
      int far_r[2] = {0, 0};
      int np = 0, nfar = 0;
      int lo = 1 << 30, hi = -1;
      uint32_t pd0 = 0, pd1 = 0;
      {
        unsigned e = e_n, en = en_n;
        int ur = i_n;
        while (e != NIL) {
          if (np < 8) {
            const uint32_t dl = (uint32_t)(r - ur) < 255u ? (uint32_t)(r - ur) : 255u;
            if (np < 4) pd0 |= dl << (8 * np); else pd1 |= dl << (8 * (np - 4));
            if (r - ur < RING) {
              const int s = ur & rm;
              ps[np] = s;
              const int a = ur == last_r ? last_mpl : rmpl[s], b = ur == last_r ? last_mpr : rmpr[s];
              if (a < lo) lo = a;
              if (b > hi) hi = b;
            } else {
              if (nfar < 2) far_r[nfar] = ur;
              ps[np] = RING + (nfar & 1);
              ++nfar;
            }
          }
          ++np;
          e = en;
          if (e != NIL) { ur = __builtin_amdgcn_readfirstlane(g.index[g.e_from[e]]); en = g.e_next_in[e]; }
        }
      }
      if (np > 8 || nfar > 2) FAIL(3 | (2 << 8));
      if (nfar) {
        // rows that left the ring (sources of long deletion edges) come back from HBM into the staging slots;
        // their stores were issued by other waves, so this is a full barrier
        __syncthreads();
        for (int f = 0; f < nfar; ++f) {
          const int64_t po = (int64_t)far_r[f] * WS;
          const int so = (RING + f) * WS;
          for (int x = tid; x < WS; x += PT) { rH[so + x] = gH[po + x]; rE1[so + x] = gE1[po + x]; rE2[so + x] = gE2[po + x]; }
          if (tid == 0) {
            rbeg[RING + f] = row_beg[far_r[f]]; rend[RING + f] = row_end[far_r[f]];
            rmpl[RING + f] = row_mpl[far_r[f]]; rmpr[RING + f] = row_mpr[far_r[f]];
          }
        }
        __syncthreads();
        for (int f = 0; f < nfar; ++f) {
          const int a = rmpl[RING + f], b = rmpr[RING + f];
          if (a < lo) lo = a;
          if (b > hi) hi = b;
        }

      }
      int beg, end;
      if (r == 0) { beg = 0; end = w < L ? w : L; }
      else {
        beg = lo + 1 - w; if (beg < 0) beg = 0;
        end = hi + 1 + w; if (end > L) end = L;
        if (end - beg + 1 > 2 * w + 129) end = beg + 2 * w + 128;
      }
      const int width = end - beg + 1;
      if (width > WS) FAIL(3 | (3 << 8));
      my_cells += (tid == 0) ? (unsigned long long)width : 0ull;
      if (tid == 0) {
        row_beg[r] = beg; row_end[r] = end; rbeg[slot] = beg; rend[slot] = end;
        prow[2 * r] = pd0; prow[2 * r + 1] = pd1;
        if (end < L) hl[r] = PNEG;
      }
      FP(0);
      const int bv = g.base[v];
      const int64_t rowo = (int64_t)r * WS;
      int32_t best = PNEG; int bl = beg, br = beg;     // per wave
      int32_t g1 = PNEG, g2 = PNEG, hpc = PNEG;        // carries between 256-column chunks
      for (int j0 = beg; j0 <= end; j0 += PT) {
        const int j = j0 + tid;
        const bool in = j <= end;
        int32_t m = PNEG, e1 = PNEG, e2 = PNEG;
        int km = 15, ko1 = 15, kx1 = 15, ko2 = 15, kx2 = 15;
        if (r == 0) { if (in && j == 0) m = 0; }
        else {
          const int sc = (in && j >= 1) ? pl_score(bv, g.q[j - 1]) : 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (k >= np) break;
            int32_t hm1 = PNEG, h = PNEG, x1 = PNEG, x2 = PNEG;
            {
              const int sl = ps[k];
              const int pb = rbeg[sl], pe_ = rend[sl];
              const int32_t* a0 = rH + sl * WS; const int32_t* a1 = rE1 + sl * WS; const int32_t* a2 = rE2 + sl * WS;
              if (in) {
                const int32_t t0 = a0[(j - 1) & wm], t1 = a0[j & wm], t2 = a1[j & wm], t3 = a2[j & wm];
                if (j - 1 >= pb && j - 1 <= pe_) hm1 = t0;
                if (j >= pb && j <= pe_) { h = t1; x1 = t2; x2 = t3; }
              }
            }
            if (hm1 > PNEG / 2) { const int32_t x = hm1 + sc; if (x > m) { m = x; km = k; } }
            {
              const int32_t a = h > PNEG / 2 ? h - P_O1 - P_E1 : PNEG, b = x1 > PNEG / 2 ? x1 - P_E1 : PNEG;
              const int32_t c = a > b ? a : b;
              if (c > PNEG / 2) {
                if (c > e1) { e1 = c; ko1 = 15; kx1 = 15; }
                if (c == e1) { if (a == c && ko1 == 15) ko1 = k; if (b == c && kx1 == 15) kx1 = k; }
              }
            }
            {
              const int32_t a = h > PNEG / 2 ? h - P_O2 - P_E2 : PNEG, b = x2 > PNEG / 2 ? x2 - P_E2 : PNEG;
              const int32_t c = a > b ? a : b;
              if (c > PNEG / 2) {
                if (c > e2) { e2 = c; ko2 = 15; kx2 = 15; }
                if (c == e2) { if (a == c && ko2 == 15) ko2 = k; if (b == c && kx2 == 15) kx2 = k; }
              }
            }
          }
        }
        int32_t hp = m; if (e1 > hp) hp = e1; if (e2 > hp) hp = e2;
        if (!in) hp = PNEG;
        const int32_t t1 = hp > PNEG / 2 ? hp + j * P_E1 : PNEG;
        const int32_t t2 = hp > PNEG / 2 ? hp + j * P_E2 : PNEG;
        const int32_t s1 = wave_scan_max(t1), s2 = wave_scan_max(t2);
        int32_t* xc = xch + 16 * par;
        par ^= 1;
        if (lane == 63) { xc[wv * 4] = s1; xc[wv * 4 + 1] = s2; xc[wv * 4 + 2] = hp; }
        FP(1);
        lds_barrier();
        FP(2);
        if (j0 == beg) { v_n = __builtin_amdgcn_readfirstlane(v_nx); e_n = g.in_head[v_n]; }
        int32_t x1 = wave_shr1(s1, PNEG), x2 = wave_shr1(s2, PNEG);
        int32_t hp_left = wave_shr1(hp, PNEG);
        {
          int32_t p1 = g1, p2 = g2, hpl = hpc, n1 = g1, n2 = g2;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) {
            const int4 xv = *reinterpret_cast<const int4*>(xc + ww * 4);
            const int32_t a = xv.x, b = xv.y, c = xv.z;
            if (ww < wv) { p1 = imax(p1, a); p2 = imax(p2, b); hpl = c; }
            n1 = imax(n1, a); n2 = imax(n2, b);
            if (ww == 3) hpc = c;
          }
          g1 = n1; g2 = n2;
          x1 = imax(x1, p1); x2 = imax(x2, p2);
          if (lane == 0) hp_left = hpl;
        }
        const int32_t f1 = x1 > PNEG / 2 ? x1 - P_O1 - j * P_E1 : PNEG;
        const int32_t f2 = x2 > PNEG / 2 ? x2 - P_O2 - j * P_E2 : PNEG;
        int32_t h = hp; if (f1 > h) h = f1; if (f2 > h) h = f2;
        if (in) {
          // the traceback's decisions, in the order the specification tries them
          uint32_t dH, dHp;
          if (m == h && km != 15) dH = (uint32_t)km; else if (e1 == h) dH = 8; else if (e2 == h) dH = 9; else if (f1 == h) dH = 10; else dH = 11;
          if (m == hp && km != 15) dHp = (uint32_t)km; else if (e1 == hp) dHp = 8; else dHp = 9;
          const uint32_t dE1 = ko1 != 15 ? (uint32_t)ko1 : (uint32_t)(8 + (kx1 & 7));
          const uint32_t dE2 = ko2 != 15 ? (uint32_t)ko2 : (uint32_t)(8 + (kx2 & 7));
          const uint32_t o1 = (hp_left > PNEG / 2 && hp_left - P_O1 - P_E1 == f1) ? 1u : 0u;
          const uint32_t o2 = (hp_left > PNEG / 2 && hp_left - P_O2 - P_E2 == f2) ? 1u : 0u;
          const int64_t o = rowo + (j & wm);
          gdir[o] = dH | (dHp << 4) | (dE1 << 8) | (dE2 << 12) | (o1 << 16) | (o2 << 17);
          gH[o] = h; gE1[o] = e1; gE2[o] = e2;
          rH[slot * WS + (j & wm)] = h; rE1[slot * WS + (j & wm)] = e1; rE2[slot * WS + (j & wm)] = e2;
          if (j == L) hl[r] = h;
        }
        const int32_t hm = in ? h : PNEG;
        const int32_t wmx = __builtin_amdgcn_readlane(wave_scan_max(hm), 63);
        const unsigned long long eq = __ballot(in && h == wmx);
        if (eq) {
          const int l = j0 + wv * 64 + __builtin_ctzll(eq), rr = j0 + wv * 64 + 63 - __builtin_clzll(eq);
          if (wmx > best) { best = wmx; bl = l; br = rr; }
          else if (wmx == best) br = rr;
        }
      }
      if (lane == 0) { part[slot * 8 + wv * 2] = best; part[slot * 8 + wv * 2 + 1] = bl | (br << 16); }
      f_n = 0; en_n = NIL;
      if (e_n != NIL) { f_n = g.e_from[e_n]; en_n = g.e_next_in[e_n]; }
      FP(3);
      lds_barrier();
      FP(4);
      i_n = __builtin_amdgcn_readfirstlane(g.index[f_n]);
      {
        // leftmost / rightmost column attaining the row maximum, from the four waves' partial results
        int32_t bb = PNEG; int l = beg, rr = beg;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
          const int32_t b = part[slot * 8 + ww * 2];
          const int lr = part[slot * 8 + ww * 2 + 1];
          const int pl = lr & 0xffff, prr = (lr >> 16) & 0xffff;
          if (b > bb) { bb = b; l = pl; rr = prr; }
          else if (b == bb) { if (pl < l) l = pl; if (prr > rr) rr = prr; }
        }
        last_r = r; last_mpl = l; last_mpr = rr;
        if (tid == 0) { rmpl[slot] = l; rmpr[slot] = rr; row_mpl[r] = l; row_mpr[r] = rr; }
      }
    }
    __syncthreads();   // direction words, deltas and end cells are in HBM
    PROF_ADD(0);
    // --------------------------------------------------------- traceback (wave 0)
    if (wv == 0) {
      int bu = -1; int32_t bsc = PNEG;
      for (unsigned e = g.in_head[1]; e != NIL; e = g.e_next_in[e]) {
        const int ur = g.index[g.e_from[e]];
        const int32_t h = hl[ur];
        if (h > bsc) { bsc = h; bu = ur; }
      }
      int nops = -1;
      if (bu >= 0 && bsc > PNEG / 2) {
        nops = 0;
        int r = __builtin_amdgcn_readfirstlane(bu), j = L, st = 0;   // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 H'
        int r0 = -(1 << 28), j0 = 0;
        uint32_t W0 = 0, W1 = 0, W2 = 0, W3 = 0, P0 = 0, P1 = 0;
        while (r != 0 || j > 0) {
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
              const uint32_t* bp = gdir + (int64_t)rr * WS;
              const int c = j0 - lane;
              W0 = bp[c & wm]; W1 = bp[(c + 1) & wm]; W2 = bp[(c + 2) & wm]; W3 = bp[(c + 3) & wm];
              P0 = prow[2 * rr]; P1 = prow[2 * rr + 1];
            }
            // wait for the window here, not at the join below (where the wait would also cover the op stores
            // of every step: vmcnt is in-order)
            asm volatile("" : "+v"(W0), "+v"(W1), "+v"(W2), "+v"(W3), "+v"(P0), "+v"(P1));
          }
          const uint32_t wsel = d == 0 ? W0 : d == 1 ? W1 : d == 2 ? W2 : W3;
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
              unsigned e = g.in_head[g.order[r]];
              for (int t = 0; t < s; ++t) e = g.e_next_in[e];
              r = __builtin_amdgcn_readfirstlane(g.index[g.e_from[e]]);
            }
          }
        }
      }
      if (lane == 0) sh[3] = nops;
    }
    __syncthreads();
    nops = sh[3];
    __syncthreads();
    PROF_ADD(1);
    }
    if (nops < 0) FAIL(3 | (4 << 8));
    // ------------------------------------------------------- graph update
    const int ncols = sh[2], n_old = N;
    for (int c = tid; c < ncols; c += PT) scr[c] = 0;
    __syncthreads();
    int carry_c = 0, carry_n = 0, carry_key = 0;
    for (int p0 = 0; p0 < nops; p0 += PT) {
      const int p = p0 + tid;
      const bool valid = p < nops;
      int row = -1, j = -1;
      if (valid) { row = op_node[nops - 1 - p]; j = op_q[nops - 1 - p]; }
      const bool ali = valid && row >= 0 && j >= 0, ins = valid && row < 0;
      const int v = ali ? (int)g.order[row] : -1;
      const int qb = (ali || ins) ? (int)g.q[j] : 0;
      int use = -1;
      bool isnew = ins;
      if (ali) {
        if (g.base[v] == qb) use = v;
        else { const int a = aln[5 * v + qb]; if (a >= 0) use = a; else isnew = true; }
      }
      int tot_c, tot_n, tot_k;
      const int cidx = carry_c + block_excl_sum((ali || ins) ? 1 : 0, bs, tog, &tot_c);
      const int nrank = carry_n + block_excl_sum(isnew ? 1 : 0, bs, tog, &tot_n);
      const int key = ali ? (((cidx + 1) << 16) | (int)g.col[v]) : 0;
      const int ikey = imax(carry_key, block_incl_max(key, bs, tog, &tot_k));
      carry_c += tot_c; carry_n += tot_n; carry_key = imax(carry_key, tot_k);
      uint32_t aux = 0xFFFFFFFFu;
      if (isnew && n_old + nrank < nc) {   // (capacity is checked after the loop)
        const int nid = n_old + nrank;
        use = nid;
        g.base[nid] = (uint8_t)qb;
        g.out_head[nid] = NIL; g.in_head[nid] = NIL;
        if (ali) {   // a new base at the column of v: joins v's aligned group
          for (int b = 0; b < 5; ++b) {
            const int sib = aln[5 * v + b];
            aln[5 * nid + b] = sib;
            if (sib >= 0) aln[5 * sib + qb] = nid;
          }
          aln[5 * nid + qb] = nid;
          g.col[nid] = g.col[v];
        } else {     // an inserted base: a new column, the t-th after its anchor's
          for (int b = 0; b < 5; ++b) aln[5 * nid + b] = b == qb ? nid : -1;
          g.col[nid] = (u16)COL_NEW;
          const int ac = ikey & 0xffff, t = (cidx + 1) - (ikey >> 16);
          atomicMax(&scr[ac], t);
          aux = (uint32_t)ac | ((uint32_t)t << 16);
        }
      }
      if (ali || ins) { path_use[cidx] = use; path_aux[cidx] = aux; }
    }
    const int PC = carry_c, n_new = n_old + carry_n;
    // the graph outgrew its LDS allocation: redo this cluster in HBM
    if (n_new > nc || ncols + carry_n > scr_n) FAIL(3 | (5 << 8));
    __syncthreads();
    for (int t = tid; t <= PC; t += PT) {
      const int u = t == 0 ? 0 : path_use[t - 1], v = t == PC ? 1 : path_use[t];
      lg_add_edge_par(g, u, v, &sh[1], ec);
    }
    __syncthreads();
    if (sh[1] > ec) FAIL(3 | (5 << 8));
    // column ranks: every column moves right by the number of columns inserted before it
    int carry = 0;
    for (int c0 = 0; c0 < ncols; c0 += PT) {
      const int c = c0 + tid;
      const int x = c < ncols ? scr[c] : 0;
      int tot;
      const int ex = block_excl_sum(x, bs, tog, &tot);
      if (c < ncols) scr[c] = carry + ex;
      carry += tot;
    }
    __syncthreads();
    for (int v = tid; v < n_new; v += PT) {
      const unsigned cv = g.col[v];
      if (cv < COL_SINK) g.col[v] = (u16)(cv + (unsigned)scr[cv]);
    }
    __syncthreads();
    for (int t = tid; t < PC; t += PT) {
      const uint32_t aux = path_aux[t];
      if (aux != 0xFFFFFFFFu) { const int ac = (int)(aux & 0xffffu); g.col[path_use[t]] = (u16)(ac + scr[ac] + (int)(aux >> 16)); }
    }
    const int ncols_new = ncols + carry;
    __syncthreads();
    // counting sort of the nodes by column = a topological order; the sink goes last
    for (int c = tid; c < ncols_new; c += PT) scr[c] = 0;
    __syncthreads();
    for (int v = tid; v < n_new; v += PT) if (v != 1) atomicAdd(&scr[g.col[v]], 1);
    __syncthreads();
    carry = 0;
    for (int c0 = 0; c0 < ncols_new; c0 += PT) {
      const int c = c0 + tid;
      const int x = c < ncols_new ? scr[c] : 0;
      int tot;
      const int ex = block_excl_sum(x, bs, tog, &tot);
      if (c < ncols_new) scr[c] = carry + ex;
      carry += tot;
    }
    __syncthreads();
    for (int v = tid; v < n_new; v += PT) if (v != 1) {
      const int pos = atomicAdd(&scr[g.col[v]], 1);
      g.order[pos] = (u16)v; g.index[v] = (u16)pos;
    }
    if (tid == 0) { g.order[n_new - 1] = 1; g.index[1] = (u16)(n_new - 1); sh[0] = n_new; sh[2] = ncols_new; }
    __syncthreads();
    PROF_ADD(2);
  }
  // ----------------------------------------------------------- heaviest bundle (lane 0)
  PROF_T();
  if (tid == 0) {
    const int N = sh[0];
    int32_t* score = 2 * N <= scr_n ? scr : gH;   // score[node], best[node]
    int32_t* bestn = score + N;
    for (int r = N - 1; r >= 0; --r) {
      const int v = g.order[r];
      int bst = -1, bw = -1; int32_t bsc = -1;
      for (unsigned e = g.out_head[v]; e != NIL; e = g.e_next_out[e]) {
        const int x = g.e_to[e];
        const int wgt = g.e_w[e];
        const int32_t sx = score[x];
        if (wgt > bw || (wgt == bw && sx > bsc)) { bw = wgt; bsc = sx; bst = x; }
      }
      bestn[v] = bst;
      score[v] = bst >= 0 ? bw + bsc : 0;
    }
    int len = 0;
    for (int v = bestn[0]; v >= 0 && v != 1; v = bestn[v]) cons[len++] = g.base[v];
    cons_len[blockIdx.x] = len;
    status[blockIdx.x] = 0;
    atomicAdd(cells, my_cells);
    PROF_ADD(3);
    for (int k = 0; k < 4; ++k) atomicAdd(&g_poa_prof[k], prof[k]);
#ifdef POA_FINE_PROF
    for (int k = 0; k < 6; ++k) printf("fp%d %lld\n", k, fp[k]);
#endif
  }
#undef FAIL
}

size_t poa_lds_bytes(int nc, int ec, int max_len, int ws, int ring) {
  size_t b = sizeof(u16) * (5 * (size_t)nc + 5 * (size_t)ec) + (size_t)nc + (size_t)max_len;
  b = (b + 15) & ~(size_t)15;
  const size_t ns = (size_t)ring + 2;
  b += sizeof(int32_t) * (3 * ns * (size_t)ws + 4 * ns + 8 * (size_t)ring + 32 + 8 + 8) + 64;
  return b;
}

hipError_t poa_lds_launch(const PoaLdsTask* d_tasks, int n_tasks, size_t lds_bytes, const uint8_t* d_seqs,
                          const int64_t* d_seq_off, int32_t* ws32, uint8_t* ws8, int32_t* d_len, int32_t* d_status,
                          unsigned long long* d_cells) {
  hipError_t e = hipFuncSetAttribute((const void*)poa_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(poa_lds_kernel, dim3((unsigned)n_tasks), dim3(PT), lds_bytes, 0, d_tasks, d_seqs, d_seq_off, ws32, ws8,
                     d_len, d_status, d_cells);
  e = hipGetLastError();
  if (e == hipSuccess && getenv("SVDSS_DEBUG")) {
    unsigned long long h[8];
    e = hipDeviceSynchronize();
    if (e == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_poa_prof), sizeof h) == hipSuccess)
      fprintf(stderr, "[poa_lds] n=%d lds=%zu  100MHz ticks (lane 0 of each cluster): forward %llu traceback %llu update %llu bundle %llu\n",
              n_tasks, lds_bytes, h[0], h[1], h[2], h[3]);
    memset(h, 0, sizeof h);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_poa_prof), h, sizeof h);
  }
  return e;
}
