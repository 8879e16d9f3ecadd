// poa.hip -- gfx950 kernel + C-ABI for the partial-order-alignment consensus of `SVDSS call`.
//
// Replaces Caller::run_poa's abpoa_msa + consensus (/root/reference/caller.cpp:257-308, abPOA
// v1.5.3) for a batch of sub-clusters.  abPOA itself is not available (git-fetched), so the
// algorithm is the published one (POA with adaptive band, convex gap, heaviest-bundle consensus)
// under the deterministic specification written out in oracle/svdss_oracle_poa.c, which this
// kernel must reproduce bit for bit.
//
// Mapping: one wavefront per sub-cluster.  The reads of a cluster are aligned to the growing
// graph one after the other (inherently sequential); inside one alignment the rows (graph nodes
// in topological order) depend on their predecessors, but the columns of a row -- the band of
// ~2w+1 read positions -- are independent once the horizontal-gap state F is written as a
// prefix maximum, F(j) = max_{k<j}(H'(k) + k e) - o - j e, which is a wave-level scan.  So the 64
// lanes sweep the band; graph bookkeeping (topological sort, traceback, graph update, heaviest
// bundle) runs on lane 0.  Banded DP matrices, the graph and the traceback live in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/svdss_hip.h"
#include "dev_arena.h"
#include "hip_check.h"
#include "poa_wave.h"
#include "poa_quad.h"

extern thread_local std::string g_svdss_hip_err;

#define HIPCHK3(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      g_svdss_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);        \
      return (e_ == hipErrorOutOfMemory) ? SVDSS_ENOMEM : SVDSS_EHIP;             \
    }                                                                             \
  } while (0)

#define PNEG (-0x20000000)
#define P_O1 4
#define P_E1 2
#define P_O2 24
#define P_E2 1
#define P_MATCH 2
#define P_MISMATCH 4

struct PoaTask {
  int64_t seq_first, n_seqs;   // reads of this cluster: seq_off[seq_first .. seq_first+n_seqs]
  int32_t cap_nodes, cap_edges, max_len;
  int64_t pool_cap;            // int32 cells per DP array
  // workspace offsets (elements of the respective typed pools)
  int64_t node_off;            // per-node int32 arrays (stride cap_nodes): out_head,out_tail,in_head,in_tail,order,index,deg,best,row_beg,row_end,mpl,mpr + aln[5]
  int64_t edge_off;            // per-edge int32 arrays (stride cap_edges): from,to,w,next_out,next_in
  int64_t dp_off;              // 6 arrays of pool_cap int32
  int64_t op_off;              // 2 arrays of (cap_nodes + max_len + 4) int32
  int64_t row_off64;           // per-node int64: row offset into the DP arrays; then score[cap_nodes]
  int64_t base_off;            // per-node uint8 base
  int64_t cons_off;            // output consensus (uint8), capacity cap_nodes
};

struct PoaGraph {
  int n_nodes, n_edges, cap_nodes, cap_edges;
  uint8_t* base;
  int *out_head, *out_tail, *in_head, *in_tail, *order, *index, *deg, *best, *row_beg, *row_end, *mpl, *mpr, *aln;
  int *e_from, *e_to, *e_w, *e_next_out, *e_next_in;
  int64_t *row_off, *score;
};

__device__ __forceinline__ int p_score(int a, int b) { return (a >= 4 || b >= 4) ? 0 : (a == b ? P_MATCH : -P_MISMATCH); }

__device__ int g_new_node(PoaGraph& g, int base) {
  const int v = g.n_nodes++;
  g.base[v] = (uint8_t)base;
  g.out_head[v] = g.out_tail[v] = g.in_head[v] = g.in_tail[v] = -1;
  for (int b = 0; b < 5; ++b) g.aln[5 * v + b] = -1;
  return v;
}

__device__ void g_add_edge(PoaGraph& g, int u, int v) {
  for (int e = g.out_head[u]; e >= 0; e = g.e_next_out[e])
    if (g.e_to[e] == v) { g.e_w[e]++; return; }
  const int e = g.n_edges++;
  g.e_from[e] = u; g.e_to[e] = v; g.e_w[e] = 1;
  g.e_next_out[e] = -1; g.e_next_in[e] = -1;
  if (g.out_tail[u] < 0) g.out_head[u] = e; else g.e_next_out[g.out_tail[u]] = e;
  g.out_tail[u] = e;
  if (g.in_tail[v] < 0) g.in_head[v] = e; else g.e_next_in[g.in_tail[v]] = e;
  g.in_tail[v] = e;
}

__device__ void g_toposort(PoaGraph& g) {   // lane 0
  for (int v = 0; v < g.n_nodes; ++v) g.deg[v] = 0;
  for (int e = 0; e < g.n_edges; ++e) g.deg[g.e_to[e]]++;
  int qh = 0, qt = 0;
  g.order[qt++] = 0;
  while (qh < qt) {
    const int u = g.order[qh];
    g.index[u] = qh++;
    for (int e = g.out_head[u]; e >= 0; e = g.e_next_out[e])
      if (--g.deg[g.e_to[e]] == 0) g.order[qt++] = g.e_to[e];
  }
}

struct PoaDp { int32_t *H, *Hp, *E1, *E2, *F1, *F2; };

__device__ __forceinline__ int32_t dp_at(const int32_t* arr, const PoaGraph& g, int r, int j) {
  return (j < g.row_beg[r] || j > g.row_end[r]) ? PNEG : arr[g.row_off[r] + (j - g.row_beg[r])];
}

// wave-wide inclusive prefix maximum over the 64 lanes
__device__ __forceinline__ int32_t wave_scan_max(int32_t x, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t y = __shfl_up(x, d, 64);
    if (lane >= d && y > x) x = y;
  }
  return x;
}

// Forward DP of one read against the graph (all 64 lanes).  Returns false if the DP arrays would
// overflow pool_cap.  *cells accumulates the number of DP cells.
__device__ bool poa_forward(PoaGraph& g, const PoaDp& dp, const uint8_t* q, int L, int w, int64_t pool_cap,
                            unsigned long long& cells) {
  const int lane = threadIdx.x & 63;
  const int n = g.n_nodes;
  int64_t used = 0;
  for (int r = 0; r < n; ++r) {
    const int v = g.order[r];
    if (v == 1) {
      if (lane == 0) { g.row_beg[r] = 0; g.row_end[r] = -1; g.row_off[r] = used; g.mpl[r] = 0; g.mpr[r] = 0; }
      __syncthreads();
      continue;
    }
    int beg, end;
    if (r == 0) { beg = 0; end = w < L ? w : L; }
    else {
      int lo = 1 << 30, hi = -1;
      for (int e = g.in_head[v]; e >= 0; e = g.e_next_in[e]) {
        const int ur = g.index[g.e_from[e]];
        const int a = g.mpl[ur], b = g.mpr[ur];
        if (a < lo) lo = a;
        if (b > hi) hi = b;
      }
      beg = lo + 1 - w; if (beg < 0) beg = 0;
      end = hi + 1 + w; if (end > L) end = L;
      if (end - beg + 1 > 2 * w + 129) end = beg + 2 * w + 128;
    }
    const int width = end - beg + 1;
    if (used + width > pool_cap) return false;   // uniform across lanes
    if (lane == 0) { g.row_beg[r] = beg; g.row_end[r] = end; g.row_off[r] = used; }
    const int64_t off = used;
    used += width;
    int32_t best = PNEG; int mpl = beg, mpr = beg;
    int32_t g1 = PNEG, g2 = PNEG;   // running prefix maxima of H'(k) + k e over finished chunks
    for (int j0 = beg; j0 <= end; j0 += 64) {
      const int j = j0 + lane;
      const bool in = j <= end;
      int32_t m = PNEG, e1 = PNEG, e2 = PNEG;
      if (in) {
        if (r == 0) m = (j == 0) ? 0 : PNEG;
        else {
          const int bv = g.base[v];
          for (int e = g.in_head[v]; e >= 0; e = g.e_next_in[e]) {
            const int ur = g.index[g.e_from[e]];
            if (j >= 1) {
              const int32_t h = dp_at(dp.H, g, ur, j - 1);
              if (h > PNEG / 2) { const int32_t x = h + p_score(bv, q[j - 1]); if (x > m) m = x; }
            }
            const int32_t h = dp_at(dp.H, g, ur, j);
            {
              const int32_t x = dp_at(dp.E1, g, ur, j);
              const int32_t a = h > PNEG / 2 ? h - P_O1 : PNEG, b = x > PNEG / 2 ? x : PNEG;
              int32_t c = a > b ? a : b;
              if (c > PNEG / 2) { c -= P_E1; if (c > e1) e1 = c; }
            }
            {
              const int32_t x = dp_at(dp.E2, g, ur, j);
              const int32_t a = h > PNEG / 2 ? h - P_O2 : PNEG, b = x > PNEG / 2 ? x : PNEG;
              int32_t c = a > b ? a : b;
              if (c > PNEG / 2) { c -= P_E2; if (c > e2) e2 = c; }
            }
          }
        }
      }
      int32_t hp = m; if (e1 > hp) hp = e1; if (e2 > hp) hp = e2;
      // F(j) = max_{k<j}(H'(k) + k e) - o - j e: exclusive prefix max = scan of the left neighbour
      const int32_t t1 = (in && hp > PNEG / 2) ? hp + j * P_E1 : PNEG;
      const int32_t t2 = (in && hp > PNEG / 2) ? hp + j * P_E2 : PNEG;
      const int32_t s1 = wave_scan_max(t1, lane), s2 = wave_scan_max(t2, lane);
      int32_t x1 = __shfl_up(s1, 1, 64), x2 = __shfl_up(s2, 1, 64);
      if (lane == 0) { x1 = PNEG; x2 = PNEG; }
      if (g1 > x1) x1 = g1;
      if (g2 > x2) x2 = g2;
      const int32_t f1 = x1 > PNEG / 2 ? x1 - P_O1 - j * P_E1 : PNEG;
      const int32_t f2 = x2 > PNEG / 2 ? x2 - P_O2 - j * P_E2 : PNEG;
      int32_t h = hp; if (f1 > h) h = f1; if (f2 > h) h = f2;
      if (in) {
        const int64_t o = off + (j - beg);
        dp.Hp[o] = hp; dp.E1[o] = e1; dp.E2[o] = e2; dp.F1[o] = f1; dp.F2[o] = f2; dp.H[o] = h;
      }
      // carry the chunk's maxima
      const int32_t c1 = __shfl(s1, 63, 64), c2 = __shfl(s2, 63, 64);
      if (c1 > g1) g1 = c1;
      if (c2 > g2) g2 = c2;
      // row maximum with its leftmost / rightmost column
      int32_t hm = in ? h : PNEG;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const int32_t y = __shfl_xor(hm, d, 64); if (y > hm) hm = y; }
      const unsigned long long eq = __ballot(in && h == hm);
      if (eq) {
        const int l = j0 + __builtin_ctzll(eq), rr = j0 + 63 - __builtin_clzll(eq);
        if (hm > best) { best = hm; mpl = l; mpr = rr; }
        else if (hm == best) mpr = rr;
      }
    }
    if (lane == 0) { g.mpl[r] = mpl; g.mpr[r] = mpr; }
    cells += (unsigned long long)width;
    __syncthreads();   // the row (and its geometry) is visible to every lane before its successors
  }
  return true;
}

// Traceback (lane 0): ops from the sink backwards, (node or -1, qpos or -1).  -1: sink unreachable.
__device__ int poa_traceback(const PoaGraph& g, const PoaDp& dp, const uint8_t* q, int L, int* op_node, int* op_q) {
  int bu = -1; int32_t bs = PNEG;
  for (int e = g.in_head[1]; e >= 0; e = g.e_next_in[e]) {
    const int32_t h = dp_at(dp.H, g, g.index[g.e_from[e]], L);
    if (h > bs) { bs = h; bu = g.e_from[e]; }
  }
  if (bu < 0 || bs <= PNEG / 2) return -1;
  int nops = 0, v = bu, j = L, state = 0;   // 0 H, 1 E1, 2 E2, 3 F1, 4 F2, 5 H' (no F)
  while (v != 0 || j > 0) {
    const int r = g.index[v];
    if (v == 0) { op_node[nops] = -1; op_q[nops] = j - 1; ++nops; --j; continue; }
    if (state == 0 || state == 5) {
      const int32_t h = state == 0 ? dp_at(dp.H, g, r, j) : dp_at(dp.Hp, g, r, j);
      bool moved = false;
      if (j >= 1) {
        for (int e = g.in_head[v]; e >= 0 && !moved; e = g.e_next_in[e]) {
          const int u = g.e_from[e];
          const int32_t x = dp_at(dp.H, g, g.index[u], j - 1);
          if (x > PNEG / 2 && x + p_score(g.base[v], q[j - 1]) == h) {
            op_node[nops] = v; op_q[nops] = j - 1; ++nops; v = u; --j; state = 0; moved = true;
          }
        }
      }
      if (moved) continue;
      if (dp_at(dp.E1, g, r, j) == h) { state = 1; continue; }
      if (dp_at(dp.E2, g, r, j) == h) { state = 2; continue; }
      if (state == 0 && dp_at(dp.F1, g, r, j) == h) { state = 3; continue; }
      if (state == 0 && dp_at(dp.F2, g, r, j) == h) { state = 4; continue; }
      return -1;
    } else if (state == 1 || state == 2) {
      const int32_t* E = state == 1 ? dp.E1 : dp.E2;
      const int o = state == 1 ? P_O1 : P_O2, ee = state == 1 ? P_E1 : P_E2;
      const int32_t x = dp_at(E, g, r, j);
      bool moved = false;
      for (int e = g.in_head[v]; e >= 0 && !moved; e = g.e_next_in[e]) {
        const int u = g.e_from[e];
        const int32_t h = dp_at(dp.H, g, g.index[u], j);
        if (h > PNEG / 2 && h - o - ee == x) { op_node[nops] = v; op_q[nops] = -1; ++nops; v = u; state = 0; moved = true; }
      }
      for (int e = g.in_head[v]; e >= 0 && !moved; e = g.e_next_in[e]) {
        const int u = g.e_from[e];
        const int32_t y = dp_at(E, g, g.index[u], j);
        if (y > PNEG / 2 && y - ee == x) { op_node[nops] = v; op_q[nops] = -1; ++nops; v = u; moved = true; }
      }
      if (!moved) return -1;
    } else {
      const int32_t* F = state == 3 ? dp.F1 : dp.F2;
      const int o = state == 3 ? P_O1 : P_O2, ee = state == 3 ? P_E1 : P_E2;
      const int32_t x = dp_at(F, g, r, j);
      op_node[nops] = -1; op_q[nops] = j - 1; ++nops;
      const int32_t hp = dp_at(dp.Hp, g, r, j - 1);
      if (hp > PNEG / 2 && hp - o - ee == x) state = 5;
      --j;
    }
  }
  return nops;
}

// status per cluster: 0 ok, 1 workspace too small (host retries with a larger DP pool)
__global__ void __launch_bounds__(64) poa_consensus_kernel(const PoaTask* tasks, const uint8_t* seqs, const int64_t* seq_off,
                                                          int32_t* ws32, int64_t* ws64, uint8_t* ws8,
                                                          int32_t* cons_len, int32_t* status, unsigned long long* cells) {
  const PoaTask T = tasks[blockIdx.x];
  const int lane = threadIdx.x;
  __shared__ int sh_flag;
  PoaGraph g;
  g.cap_nodes = T.cap_nodes; g.cap_edges = T.cap_edges; g.n_nodes = 0; g.n_edges = 0;
  int32_t* nb = ws32 + T.node_off;
  const int64_t cn = T.cap_nodes, ce = T.cap_edges;
  g.out_head = nb; g.out_tail = nb + cn; g.in_head = nb + 2 * cn; g.in_tail = nb + 3 * cn;
  g.order = nb + 4 * cn; g.index = nb + 5 * cn; g.deg = nb + 6 * cn; g.best = nb + 7 * cn;
  g.row_beg = nb + 8 * cn; g.row_end = nb + 9 * cn; g.mpl = nb + 10 * cn; g.mpr = nb + 11 * cn; g.aln = nb + 12 * cn;
  int32_t* eb = ws32 + T.edge_off;
  g.e_from = eb; g.e_to = eb + ce; g.e_w = eb + 2 * ce; g.e_next_out = eb + 3 * ce; g.e_next_in = eb + 4 * ce;
  g.row_off = ws64 + T.row_off64; g.score = ws64 + T.row_off64 + cn;
  g.base = ws8 + T.base_off;
  PoaDp dp;
  int32_t* db = ws32 + T.dp_off;
  dp.H = db; dp.Hp = db + T.pool_cap; dp.E1 = db + 2 * T.pool_cap; dp.E2 = db + 3 * T.pool_cap;
  dp.F1 = db + 4 * T.pool_cap; dp.F2 = db + 5 * T.pool_cap;
  int* op_node = ws32 + T.op_off;
  int* op_q = op_node + (T.cap_nodes + T.max_len + 4);
  uint8_t* cons = ws8 + T.cons_off;
  const int n = (int)T.n_seqs;
  if (n <= 0) { if (lane == 0) { cons_len[blockIdx.x] = 0; status[blockIdx.x] = 0; } return; }
  // all lanes track n_nodes / n_edges (they are needed for uniform control flow): lane 0 mutates
  // the graph in memory, then broadcasts the counters through LDS
  __shared__ int sh_nodes, sh_edges;
  unsigned long long my_cells = 0;
  if (lane == 0) {
    g_new_node(g, 4); g_new_node(g, 4);
    const uint8_t* q = seqs + seq_off[T.seq_first];
    const int L = (int)(seq_off[T.seq_first + 1] - seq_off[T.seq_first]);
    int last = 0;
    for (int j = 0; j < L; ++j) { const int v = g_new_node(g, q[j]); g.aln[5 * v + q[j]] = v; g_add_edge(g, last, v); last = v; }
    g_add_edge(g, last, 1);
    sh_nodes = g.n_nodes; sh_edges = g.n_edges;
  }
  __syncthreads();
  g.n_nodes = sh_nodes; g.n_edges = sh_edges;
  for (int i = 1; i < n; ++i) {
    const uint8_t* q = seqs + seq_off[T.seq_first + i];
    const int L = (int)(seq_off[T.seq_first + i + 1] - seq_off[T.seq_first + i]);
    if (lane == 0) g_toposort(g);
    __syncthreads();
    int w = 10 + (int)(0.01 * L);
    int nops = -1;
    for (int attempt = 0; attempt < 2; ++attempt) {
      const bool fit = poa_forward(g, dp, q, L, w, T.pool_cap, my_cells);
      if (!fit) { if (lane == 0) status[blockIdx.x] = 1; return; }
      if (lane == 0) sh_flag = poa_traceback(g, dp, q, L, op_node, op_q);
      __syncthreads();
      nops = sh_flag;
      __syncthreads();
      if (nops >= 0) break;
      w = L;   // the band lost the sink: full matrix
    }
    if (nops < 0) { if (lane == 0) status[blockIdx.x] = 2; return; }
    if (lane == 0) {
      int last = 0;
      for (int k = nops - 1; k >= 0; --k) {
        const int v = op_node[k], j = op_q[k];
        if (v >= 0 && j >= 0) {
          int use;
          if (g.base[v] == q[j]) use = v;
          else if (g.aln[5 * v + q[j]] >= 0) use = g.aln[5 * v + q[j]];
          else {
            use = g_new_node(g, q[j]);
            for (int b = 0; b < 5; ++b) {
              const int sib = g.aln[5 * v + b];
              g.aln[5 * use + b] = sib;
              if (sib >= 0) g.aln[5 * sib + q[j]] = use;
            }
            g.aln[5 * use + q[j]] = use;
          }
          g_add_edge(g, last, use); last = use;
        } else if (v < 0) {
          const int use = g_new_node(g, q[j]);
          g.aln[5 * use + q[j]] = use;
          g_add_edge(g, last, use); last = use;
        }
      }
      g_add_edge(g, last, 1);
      sh_nodes = g.n_nodes; sh_edges = g.n_edges;
    }
    __syncthreads();
    g.n_nodes = sh_nodes; g.n_edges = sh_edges;
  }
  if (lane == 0) {
    g_toposort(g);
    for (int r = g.n_nodes - 1; r >= 0; --r) {
      const int v = g.order[r];
      int bst = -1, bw = -1; int64_t bsc = -1;
      for (int e = g.out_head[v]; e >= 0; e = g.e_next_out[e]) {
        const int x = g.e_to[e];
        if (g.e_w[e] > bw || (g.e_w[e] == bw && g.score[x] > bsc)) { bw = g.e_w[e]; bsc = g.score[x]; bst = x; }
      }
      g.best[v] = bst;
      g.score[v] = bst >= 0 ? bw + bsc : 0;
    }
    int len = 0;
    for (int v = g.best[0]; v >= 0 && v != 1; v = g.best[v]) cons[len++] = g.base[v];
    cons_len[blockIdx.x] = len;
    status[blockIdx.x] = 0;
    atomicAdd(cells, my_cells);
  }
}

// ------------------------------------------------------------------- ABI

namespace {
struct DevMem3 {
  void* p = nullptr;
  ~DevMem3() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { HIPCHK3(hipMalloc(&p, bytes ? bytes : 16)); return SVDSS_OK; }
};

}  // namespace

struct svdss_poa_batch {
  int64_t n_clusters = 0;
  int64_t n_hbm = 0;   // clusters the LDS kernel handed to the HBM kernel
  int64_t n_quad_back = 0;   // clusters poa_quad.hip handed to poa_wave.hip's rounds
  int64_t cells = 0;
  double kernel_ms = 0.0;
  std::vector<int64_t> cons_len;
  std::vector<uint8_t> cons;   // concatenated, symbols 0..4
  // device state kept between calls
  int device = -1;
  DevArena in_arena, ws_arena;
  std::vector<hipStream_t> streams;
  ~svdss_poa_batch() {
    if (device >= 0) (void)hipSetDevice(device);
    for (hipStream_t st : streams) (void)hipStreamDestroy(st);
  }
};

extern "C" int svdss_poa_consensus_batch(const uint8_t* seqs, const int64_t* seq_off, const int64_t* cluster_off,
                                         int64_t n_clusters, int32_t device, svdss_poa_batch_t** out) {
  if (!out || n_clusters < 0 || device < 0) return SVDSS_EINVAL;
  if (n_clusters > 0 && (!seq_off || !cluster_off)) return SVDSS_EINVAL;
  HIPCHK3(hipSetDevice(device));
  svdss_poa_batch* b = *out ? *out : new (std::nothrow) svdss_poa_batch();
  if (!b) return SVDSS_ENOMEM;
  *out = b;
  b->n_clusters = n_clusters;
  b->cells = 0;
  b->n_hbm = 0;
  b->n_quad_back = 0;
  b->kernel_ms = 0.0;
  b->cons_len.assign((size_t)n_clusters, 0);
  b->cons.clear();
  if (n_clusters == 0) return SVDSS_OK;
  const int64_t n_seqs_total = cluster_off[n_clusters];
  const int64_t total_syms = seq_off[n_seqs_total];
  for (int64_t i = 0; i < n_seqs_total; ++i) {
    const int64_t l = seq_off[i + 1] - seq_off[i];
    if (l < 0) return SVDSS_EINVAL;
    if (l >= (1 << 24)) return SVDSS_ERANGE;
  }
  if (total_syms > 0 && !seqs) return SVDSS_EINVAL;
  int rc;
  if (b->device != device) {   // (a batch object is normally used with one device)
    for (hipStream_t st : b->streams) (void)hipStreamDestroy(st);
    b->streams.clear();
    b->in_arena.drop(); b->ws_arena.drop();
    b->device = device;
  }
  const size_t off_bytes = sizeof(int64_t) * (size_t)(n_seqs_total + 1);
  HIPCHK3(b->in_arena.reserve(DevArena::padded((size_t)total_syms) + DevArena::padded(off_bytes) + DevArena::padded(8)));
  struct { void* p; } d_seqs{b->in_arena.take((size_t)total_syms)}, d_off{b->in_arena.take(off_bytes)},
      d_cells{b->in_arena.take(8)};
  // every copy and launch of this call goes to the batch object's own non-blocking streams and every wait is a
  // wait for those streams: calls on different batch objects (threads) and a search running beside them overlap
  // (few streams: the runtime multiplexes streams onto a handful of hardware queues, and streams that share one run
  // in order)
  while (b->streams.size() < 6) {   // (the launches of a round run side by side: one stream each while they last)
    hipStream_t st;
    HIPCHK3(svdss_make_stream(&st, "SVDSS_CALL_CUS"));
    b->streams.push_back(st);
  }
  const hipStream_t s0 = b->streams[0];
  if (total_syms) HIPCHK3(hipMemcpyAsync(d_seqs.p, seqs, (size_t)total_syms, hipMemcpyHostToDevice, s0));
  HIPCHK3(hipMemcpyAsync(d_off.p, seq_off, off_bytes, hipMemcpyHostToDevice, s0));
  HIPCHK3(hipMemsetAsync(d_cells.p, 0, 8, s0));
  hipEvent_t ev0, ev1;
  HIPCHK3(hipEventCreate(&ev0));
  HIPCHK3(hipEventCreate(&ev1));
  std::vector<std::vector<uint8_t>> results((size_t)n_clusters);
  // fast path: the LDS-resident kernel in up to three rounds of growing generosity -- ring rows of 2w + 33 columns
  // and a graph of ~1.5 x the longest read (what nearly every sub-cluster needs); then the specification's widest
  // band (2w + 129) and 3 x; then full-matrix rows (a band that lost the sink).  What is still left (status 3)
  // falls through to the HBM kernel: pass 0 with a banded workspace, pass 1 with a full-size DP pool
  std::vector<int64_t> todo;
  const bool use_lds = getenv("SVDSS_POA_HBM") == nullptr;
  std::vector<int64_t> cur((size_t)n_clusters), retry;
  for (int64_t c = 0; c < n_clusters; ++c) cur[(size_t)c] = c;
  int n_cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
  }
  // bytes of workspace per wave of launches (SVDSS_POA_WS_GB, default 32: a whole genome's 21,500 sub-clusters want ~64 GB and
  // run in two waves of ~75 ms; one wave needs an allocation that large on a device other processes have just left)
  size_t ws_budget = (size_t)(getenv("SVDSS_POA_WS_GB") && atoll(getenv("SVDSS_POA_WS_GB")) > 0 ? atoll(getenv("SVDSS_POA_WS_GB")) : 32) << 30;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ws_budget = std::min(ws_budget, (free_b + b->ws_arena.cap) / 2);
    if (ws_budget < ((size_t)1 << 30)) ws_budget = (size_t)1 << 30;
  }
  const int n_rounds = 3;
  // round -1 (poa_quad.hip): several sub-clusters per wavefront -- the band as it is in practice, a graph of ~1.5 x the
  // longest read, at most 7 predecessors per node; whatever it hands back starts round 0 (SVDSS_POA_QUAD=0: skip it)
  const bool use_quad = use_lds && !(getenv("SVDSS_POA_QUAD") && atoi(getenv("SVDSS_POA_QUAD")) == 0);
  const int64_t quad_short = getenv("SVDSS_POA_QUAD_SHORT") ? atoll(getenv("SVDSS_POA_QUAD_SHORT")) : 0;
  const int64_t quad_minwork_pct = getenv("SVDSS_POA_QUAD_MINWORK") ? atoll(getenv("SVDSS_POA_QUAD_MINWORK")) : 0;
  const int64_t quad_rows16 = getenv("SVDSS_POA_QUAD_ROWS16") ? atoll(getenv("SVDSS_POA_QUAD_ROWS16")) : 0;
  const int64_t quad_rows32 = getenv("SVDSS_POA_QUAD_ROWS32") ? atoll(getenv("SVDSS_POA_QUAD_ROWS32")) : 0;
  int64_t batch_max_work = 1;
  for (int64_t c = 0; c < n_clusters; ++c) {
    int64_t maxl = 0;
    for (int64_t s = cluster_off[c]; s < cluster_off[c + 1]; ++s) maxl = std::max(maxl, seq_off[s + 1] - seq_off[s]);
    batch_max_work = std::max(batch_max_work, (cluster_off[c + 1] - cluster_off[c]) * maxl);
  }
  // A sub-cluster the first stage hands back because a row got wider than its lanes hold goes straight to the round that has
  // wider rows when round 0's rows are no wider than the first stage's were (`call` at 30x: one such sub-cluster of 21,500
  // cost a round of 39 ms that could only fail the same way)
  std::vector<uint8_t> round0_no_wider((size_t)n_clusters, 0), skip_round0((size_t)n_clusters, 0);
  for (int round = use_quad ? -1 : 0; round < n_rounds && !cur.empty(); ++round) {
    const bool quad = round < 0;
    struct Cand { int64_t c; size_t lds; int cols; int gw; PoaWaveTask t; };
    std::vector<Cand> cands;
    const size_t LDS_MAX = 160 * 1024 - 512;
    for (int64_t c : cur) {
      PoaWaveTask t;
      memset(&t, 0, sizeof t);
      t.seq_first = cluster_off[c];
      t.n_seqs = cluster_off[c + 1] - cluster_off[c];
      int64_t tot = 0, maxl = 0;
      for (int64_t s = t.seq_first; s < t.seq_first + t.n_seqs; ++s) {
        const int64_t l = seq_off[s + 1] - seq_off[s];
        tot += l;
        if (l > maxl) maxl = l;
      }
      // the graph rarely grows beyond ~1.5 x the longest read (later rounds: 3 x); a cluster that outgrows its
      // allocation is redone.  SVDSS_POA_NC scales the first estimate (percent).
      const int nc_pct = round > 0 ? 300 : getenv("SVDSS_POA_NC") ? std::max(atoi(getenv("SVDSS_POA_NC")), 100) : 150;
      int64_t nc = std::min<int64_t>(tot + 2, maxl * nc_pct / 100 + 8 * t.n_seqs + 64);
      if (nc > 65000) nc = 65000;
      const int64_t ecap = std::min<int64_t>(nc + nc / (round > 0 ? 2 : 4) + t.n_seqs + 64, 100000);
      // widest row the ring holds: the band as it is in practice (round 0), as wide as the specification lets it
      // get (round 1), the full matrix (round 2, after the band lost the sink)
      const int64_t w_band = 10 + (int64_t)(0.01 * (double)maxl);
      // (round 0: 2w + 1 columns plus slack for the spread of the predecessors' maxima -- rounded up to 64 where that
      // leaves at least 8 of slack, so that a row is one column per lane: the C = 1 instantiation)
      const int64_t w2 = 2 * w_band + 1;
      const int64_t wcap0 = w2 + 8 <= 64 ? 64 : w2 + 32;
      const int64_t wcap = std::min<int64_t>(round <= 0 ? wcap0 : round == 1 ? 2 * w_band + 129 : maxl + 1, maxl + 1);
      if (quad) {
        // group width x columns per lane >= 2w + 1 columns plus 8 of slack for the spread of the predecessors' maxima
        // (SVDSS_POA_QUAD_GW: 16 -- four sub-clusters per wavefront --, 32 or 64)
        // Group width: four short sub-clusters share a wavefront (a quarter of the wavefront slots for the latency-bound
        // traceback / graph-update phases); a long one gets the wavefront to itself -- the longest chains of a batch
        // decide when it ends, and a row of C = 2 columns per lane is the quickest there is.
        // (SVDSS_POA_QUAD_ROWS16 / _ROWS32: a sub-cluster of at most that many reads x length shares its wavefront with three /
        // one other: its chain is short enough not to become the batch's tail at the slower lock-step pace)
        const int64_t chain = t.n_seqs * maxl;
        const int gw = getenv("SVDSS_POA_QUAD_GW") ? atoi(getenv("SVDSS_POA_QUAD_GW"))
                       : (maxl <= quad_short || chain <= quad_rows16) ? 16 : chain <= quad_rows32 ? 32 : 64;
        // SVDSS_POA_QUAD_MINWORK (percent of the batch's largest reads x length): only the long chains take this stage
        if (t.n_seqs * maxl * 100 < quad_minwork_pct * batch_max_work) { retry.push_back(c); continue; }
        const int64_t need = std::min<int64_t>(w2 + 8, maxl + 1);
        const int qc = (int)std::max<int64_t>((need + gw - 1) / gw, gw == 16 ? 3 : gw == 32 ? 2 : 1);
        if ((gw != 16 && gw != 32 && gw != 64) || !poa_quad_supported(gw, qc) || t.n_seqs <= 0 || t.n_seqs > 8191 ||
            poa_bundle_lds_bytes((int)nc) > LDS_MAX || poa_quad_lds_bytes(gw, qc, (int)maxl) > LDS_MAX) {
          retry.push_back(c);
          continue;
        }
        t.nc = (int32_t)nc; t.ec = (int32_t)ecap; t.max_len = (int32_t)maxl; t.ws = gw * qc; t.rs = 0; t.ring = 0;
        round0_no_wider[(size_t)c] = wcap0 <= (int64_t)gw * qc ? 1 : 0;
        cands.push_back(Cand{c, poa_quad_lds_bytes(gw, qc, (int)maxl), qc, gw, t});
        continue;
      }
      if (round == 0 && skip_round0[(size_t)c] && n_rounds > 1) { retry.push_back(c); continue; }
      int64_t ws = 64;
      while (ws < wcap) ws <<= 1;
      const int64_t rs = (wcap + 3) & ~(int64_t)3;
      const int cols = wcap <= 64 ? 1 : wcap <= 128 ? 2 : wcap <= 192 ? 3 : 5;
      const int64_t ring = 4;
      t.nc = (int32_t)nc; t.ec = (int32_t)ecap; t.max_len = (int32_t)maxl; t.ws = (int32_t)ws; t.rs = (int32_t)rs;
      t.ring = (int32_t)ring;
      const size_t lds = poa_wave_lds_bytes(t.nc, t.max_len, t.rs, t.ring);
      if (!use_lds || lds > LDS_MAX || poa_bundle_lds_bytes(t.nc) > LDS_MAX || ws > 4096 || t.n_seqs <= 0 || t.n_seqs > 8191) {
        todo.push_back(c);
        if (t.n_seqs > 0) ++b->n_hbm;
        continue;
      }
      cands.push_back(Cand{c, lds, cols, 0, t});
    }
    // launches are grouped by instantiation and by LDS size class, so that small clusters are not charged the
    // LDS of the largest one (LDS decides how many sub-clusters a CU keeps in flight); the groups run
    // concurrently on their own streams (one sub-cluster is a chain of dependent steps: the machine is
    // filled by running many of them, whichever launch they came from)
    struct Group {
      int cols = 0, gw = 0, max_len = 0;   // gw != 0: a launch of poa_quad.hip
      int wave = 0;                        // groups of one wave of launches run together
      size_t lds = 0, bundle_lds = 0;
      std::vector<PoaWaveTask> tasks;
      std::vector<int64_t> ids;
      int64_t w32 = 0, w8 = 0;
      void *d_tasks = nullptr, *d32 = nullptr, *d8 = nullptr, *d_len = nullptr, *d_st = nullptr;
      size_t bytes() const {
        const size_t nt = tasks.size();
        return DevArena::padded(sizeof(PoaWaveTask) * nt) + DevArena::padded(sizeof(int32_t) * (size_t)w32) +
               DevArena::padded((size_t)w8) + 2 * DevArena::padded(sizeof(int32_t) * nt);
      }
    };
    std::vector<std::unique_ptr<Group>> groups;
    const int64_t group_budget32 = (int64_t)2 << 30;    // ints of workspace per launch
    std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.lds > y.lds; });
    // a sub-cluster is one chain of n_seqs x length dependent row steps: the longest chains of the batch decide
    // when it ends, so they get the issue priority (s_setprio) over the short ones that fill the CUs beside them
    if (!getenv("SVDSS_POA_NOPRIO")) {
      int64_t wmax = 1;
      for (const Cand& cd : cands) wmax = std::max<int64_t>(wmax, cd.t.n_seqs * (int64_t)cd.t.max_len);
      for (Cand& cd : cands) {
        const int64_t wk = cd.t.n_seqs * (int64_t)cd.t.max_len;
        cd.t.prio = wk * 2 > wmax ? 3 : wk * 4 > wmax ? 2 : wk * 8 > wmax ? 1 : 0;
      }
    }
    if (quad) {
      // wavefronts of sub-clusters that are alike (the groups of a wavefront walk in lock-step: it lasts as long as its
      // longest), the longest first; one launch per variant
      // A batch beyond the workspace budget (a whole genome's sub-clusters at once) runs in several waves of launches, and
      // every wave lasts at least as long as its longest chain: the sub-clusters are dealt to the waves longest first, one
      // each in turn, so that every wave has its share of long chains and of short ones to fill the machine beside them
      // (rounds 2-4 cut the sorted list into consecutive pieces: the first wave was all long chains, the last all short).
      std::vector<int> wave_of(cands.size(), 0);
      {
        size_t total = 0;
        for (const Cand& cd : cands) total += sizeof(int32_t) * (size_t)poa_wave_ws_ints(cd.t.nc, cd.t.ec, cd.t.max_len, cd.t.ws) + (size_t)cd.t.nc + 256;
        const size_t n_waves = std::max<size_t>(1, (total + ws_budget - ws_budget / 8 - 1) / (ws_budget - ws_budget / 8));
        if (n_waves > 1 && !getenv("SVDSS_POA_NO_MIX")) {
          std::vector<size_t> by_work(cands.size());
          for (size_t i = 0; i < by_work.size(); ++i) by_work[i] = i;
          std::sort(by_work.begin(), by_work.end(), [&](size_t x, size_t y) {
            const int64_t wx = cands[x].t.n_seqs * (int64_t)cands[x].t.max_len, wy = cands[y].t.n_seqs * (int64_t)cands[y].t.max_len;
            return wx != wy ? wx > wy : cands[x].c < cands[y].c;
          });
          for (size_t k = 0; k < by_work.size(); ++k) wave_of[by_work[k]] = (int)(k % n_waves);
        }
      }
      std::vector<size_t> order(cands.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](size_t xi, size_t yi) {
        const Cand &x = cands[xi], &y = cands[yi];
        if (wave_of[xi] != wave_of[yi]) return wave_of[xi] < wave_of[yi];
        if (x.gw != y.gw) return x.gw > y.gw;
        if (x.cols != y.cols) return x.cols > y.cols;
        const int64_t wx = x.t.n_seqs * (int64_t)x.t.max_len, wy = y.t.n_seqs * (int64_t)y.t.max_len;
        if (wx != wy) return wx > wy;
        return x.c < y.c;
      });
      Group* g = nullptr;
      for (size_t oi : order) {
        Cand& cd = cands[oi];
        PoaWaveTask t = cd.t;
        const int64_t need = poa_wave_ws_ints(t.nc, t.ec, t.max_len, t.ws);
        if (!g || g->wave != wave_of[oi] || g->gw != cd.gw || g->cols != cd.cols || g->w32 + need > group_budget32) {
          groups.emplace_back(new Group);
          g = groups.back().get();
          g->cols = cd.cols; g->gw = cd.gw; g->lds = cd.lds; g->wave = wave_of[oi];
        }
        g->max_len = std::max(g->max_len, (int)t.max_len);
        t.ws_off = g->w32; g->w32 += need;
        t.cons_off = g->w8; g->w8 += t.nc;
        g->bundle_lds = std::max(g->bundle_lds, poa_bundle_lds_bytes(t.nc));
        g->tasks.push_back(t);
        g->ids.push_back(cd.c);
      }
    }
    for (int ci = 0; ci < kPoaWaveNCols && !quad; ++ci) {
      Group* g = nullptr;
      size_t fill = 0;   // sub-clusters that fill the machine at the group's LDS size
      for (Cand& cd : cands) {
        if (cd.cols != kPoaWaveCols[ci]) continue;
        PoaWaveTask t = cd.t;
        const int64_t need = poa_wave_ws_ints(t.nc, t.ec, t.max_len, t.ws);
        // a new launch (with the smaller LDS of the clusters that follow) only once the current one fills all CUs
        if (!g || g->w32 + need > group_budget32 || g->tasks.size() >= fill) {
          groups.emplace_back(new Group);
          g = groups.back().get();
          g->cols = kPoaWaveCols[ci];
          g->lds = cd.lds;
          fill = (size_t)n_cus * std::min<size_t>(32, std::max<size_t>(1, LDS_MAX / cd.lds));
        }
        t.ws_off = g->w32; g->w32 += need;
        t.cons_off = g->w8; g->w8 += t.nc;
        g->bundle_lds = std::max(g->bundle_lds, poa_bundle_lds_bytes(t.nc));
        g->tasks.push_back(t);
        g->ids.push_back(cd.c);
      }
    }
    // run the groups in waves that fit the workspace budget
    size_t gpos = 0;
    while (gpos < groups.size()) {
      size_t gend = gpos, tot_bytes = 0;
      while (gend < groups.size() && (gend == gpos || (groups[gend]->wave == groups[gpos]->wave && tot_bytes + groups[gend]->bytes() <= ws_budget)))
        tot_bytes += groups[gend++]->bytes();
      {
        const auto ta = std::chrono::steady_clock::now();
        HIPCHK3(b->ws_arena.reserve(tot_bytes));
        const double as = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
        if (getenv("SVDSS_DEBUG") && as > 0.005) fprintf(stderr, "[poa] workspace of %.1f GB taken in %.3f s\n", (double)tot_bytes / 1073741824.0, as);
      }
      for (size_t gi = gpos; gi < gend; ++gi) {
        Group& g = *groups[gi];
        const size_t nt = g.tasks.size();
        g.d_tasks = b->ws_arena.take(sizeof(PoaWaveTask) * nt);
        g.d32 = b->ws_arena.take(sizeof(int32_t) * (size_t)g.w32);
        g.d8 = b->ws_arena.take((size_t)g.w8);
        g.d_len = b->ws_arena.take(sizeof(int32_t) * nt);
        g.d_st = b->ws_arena.take(sizeof(int32_t) * nt);
        HIPCHK3(hipMemcpyAsync(g.d_tasks, g.tasks.data(), sizeof(PoaWaveTask) * nt, hipMemcpyHostToDevice, s0));
        HIPCHK3(hipMemsetAsync(g.d_st, 0xff, sizeof(int32_t) * nt, s0));
      }
      HIPCHK3(hipStreamSynchronize(s0));
      const auto t0 = std::chrono::steady_clock::now();
      for (size_t gi = gpos; gi < gend; ++gi) {
        Group& g = *groups[gi];
        const hipStream_t gs = b->streams[(gi - gpos) % b->streams.size()];
        if (g.gw) {
          HIPCHK3(poa_quad_launch(g.gw, g.cols, (const PoaWaveTask*)g.d_tasks, (int)g.tasks.size(), g.max_len, (const uint8_t*)d_seqs.p,
                                  (const int64_t*)d_off.p, (int32_t*)g.d32, (int32_t*)g.d_len, (int32_t*)g.d_st,
                                  (unsigned long long*)d_cells.p, gs));
          HIPCHK3(poa_bundle_launch((const PoaWaveTask*)g.d_tasks, (int)g.tasks.size(), g.bundle_lds, (int32_t*)g.d32, (uint8_t*)g.d8,
                                    (int32_t*)g.d_len, (const int32_t*)g.d_st, gs));
        } else
        HIPCHK3(poa_wave_launch(g.cols, (const PoaWaveTask*)g.d_tasks, (int)g.tasks.size(), g.lds, g.bundle_lds,
                                (const uint8_t*)d_seqs.p, (const int64_t*)d_off.p, (int32_t*)g.d32, (uint8_t*)g.d8,
                                (int32_t*)g.d_len, (int32_t*)g.d_st, (unsigned long long*)d_cells.p, gs));
      }
      for (size_t k = 0; k < std::min(gend - gpos, b->streams.size()); ++k) HIPCHK3(hipStreamSynchronize(b->streams[k]));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      b->kernel_ms += ms;   // wall time of the concurrent launches
      int why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int64_t n_run = 0;
      for (size_t gi = gpos; gi < gend; ++gi) {
        Group& g = *groups[gi];
        const int64_t nt = (int64_t)g.tasks.size();
        n_run += nt;
        std::vector<int32_t> lens((size_t)nt), st((size_t)nt);
        std::vector<uint8_t> h8((size_t)g.w8);
        HIPCHK3(hipMemcpyAsync(lens.data(), g.d_len, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, s0));
        HIPCHK3(hipMemcpyAsync(st.data(), g.d_st, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, s0));
        if (g.w8) HIPCHK3(hipMemcpyAsync(h8.data(), g.d8, (size_t)g.w8, hipMemcpyDeviceToHost, s0));
        HIPCHK3(hipStreamSynchronize(s0));
        for (int64_t k = 0; k < nt; ++k) {
          if (st[(size_t)k] == 0) {
            const uint8_t* src = h8.data() + g.tasks[(size_t)k].cons_off;
            results[(size_t)g.ids[(size_t)k]].assign(src, src + lens[(size_t)k]);
          } else {
            const int reason = (st[(size_t)k] >> 8) & 7;
            if (quad) { retry.push_back(g.ids[(size_t)k]); if (reason == 3 && round0_no_wider[(size_t)g.ids[(size_t)k]]) skip_round0[(size_t)g.ids[(size_t)k]] = 1; }
            else if (round + 1 < n_rounds && (reason == 3 || reason == 4 || reason == 5)) retry.push_back(g.ids[(size_t)k]);
            else { todo.push_back(g.ids[(size_t)k]); ++b->n_hbm; }
            ++why[reason];
            if (quad) ++b->n_quad_back;
          }
        }
        groups[gi].reset();
      }
      if (getenv("SVDSS_DEBUG")) { if (quad) poa_quad_debug_report(); else poa_wave_debug_report(); }
      if (getenv("SVDSS_DEBUG"))
        fprintf(stderr, "[poa] %s round %d: %lld clusters in %zu launches, %.3f ms, not done: first-read %d preds %d width %d band %d capacity %d other %d\n",
                quad ? "quad" : "wave", round, (long long)n_run, gend - gpos, ms, why[1], why[2], why[3], why[4], why[5], why[0] + why[6] + why[7]);
      gpos = gend;
    }
    cur.swap(retry);
    retry.clear();
    if (quad) std::sort(cur.begin(), cur.end());
  }
  std::sort(todo.begin(), todo.end());
  for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
    std::vector<int64_t> next;
    size_t pos = 0;
    while (pos < todo.size()) {
      std::vector<PoaTask> tasks;
      std::vector<int64_t> ids;
      int64_t w32 = 0, w64 = 0, w8 = 0;
      const int64_t budget32 = (int64_t)3 << 30;   // 12 GiB of int32 workspace per launch
      while (pos < todo.size()) {
        const int64_t c = todo[pos];
        PoaTask t;
        memset(&t, 0, sizeof t);
        t.seq_first = cluster_off[c];
        t.n_seqs = cluster_off[c + 1] - cluster_off[c];
        int64_t tot = 0, maxl = 0;
        for (int64_t s = t.seq_first; s < t.seq_first + t.n_seqs; ++s) {
          const int64_t l = seq_off[s + 1] - seq_off[s];
          tot += l;
          if (l > maxl) maxl = l;
        }
        t.cap_nodes = (int32_t)(tot + 2);
        t.cap_edges = (int32_t)(tot + t.n_seqs + 2);
        t.max_len = (int32_t)maxl;
        const int64_t wband = 2 * (10 + (int64_t)(0.01 * (double)maxl)) + 129;
        t.pool_cap = pass == 0 ? (int64_t)t.cap_nodes * (wband < maxl + 1 ? wband : maxl + 1)
                               : (int64_t)t.cap_nodes * (maxl + 1);
        const int64_t need32 = 17 * (int64_t)t.cap_nodes + 5 * (int64_t)t.cap_edges + 6 * t.pool_cap +
                               2 * ((int64_t)t.cap_nodes + maxl + 4);
        if (!tasks.empty() && w32 + need32 > budget32) break;
        t.node_off = w32; w32 += 17 * (int64_t)t.cap_nodes;
        t.edge_off = w32; w32 += 5 * (int64_t)t.cap_edges;
        t.dp_off = w32; w32 += 6 * t.pool_cap;
        t.op_off = w32; w32 += 2 * ((int64_t)t.cap_nodes + maxl + 4);
        t.row_off64 = w64; w64 += 2 * (int64_t)t.cap_nodes;
        t.base_off = w8; w8 += t.cap_nodes;
        t.cons_off = w8; w8 += t.cap_nodes;
        tasks.push_back(t);
        ids.push_back(c);
        ++pos;
      }
      const int64_t nt = (int64_t)tasks.size();
      DevMem3 d_tasks, d32, d64, d8, d_len, d_st;
      if ((rc = d_tasks.alloc(sizeof(PoaTask) * (size_t)nt)) || (rc = d32.alloc(sizeof(int32_t) * (size_t)w32)) ||
          (rc = d64.alloc(sizeof(int64_t) * (size_t)w64)) || (rc = d8.alloc((size_t)w8)) ||
          (rc = d_len.alloc(sizeof(int32_t) * (size_t)nt)) || (rc = d_st.alloc(sizeof(int32_t) * (size_t)nt)))
        return rc;
      HIPCHK3(hipMemcpyAsync(d_tasks.p, tasks.data(), sizeof(PoaTask) * (size_t)nt, hipMemcpyHostToDevice, s0));
      HIPCHK3(hipMemsetAsync(d_st.p, 0xff, sizeof(int32_t) * (size_t)nt, s0));
      HIPCHK3(hipEventRecord(ev0, s0));
      hipLaunchKernelGGL(poa_consensus_kernel, dim3((unsigned)nt), dim3(64), 0, s0, (const PoaTask*)d_tasks.p,
                         (const uint8_t*)d_seqs.p, (const int64_t*)d_off.p, (int32_t*)d32.p, (int64_t*)d64.p,
                         (uint8_t*)d8.p, (int32_t*)d_len.p, (int32_t*)d_st.p, (unsigned long long*)d_cells.p);
      HIPCHK3(hipGetLastError());
      HIPCHK3(hipEventRecord(ev1, s0));
      HIPCHK3(hipStreamSynchronize(s0));
      float ms = 0.f;
      HIPCHK3(hipEventElapsedTime(&ms, ev0, ev1));
      b->kernel_ms += ms;
      std::vector<int32_t> lens((size_t)nt), st((size_t)nt);
      std::vector<uint8_t> h8((size_t)w8);
      HIPCHK3(hipMemcpyAsync(lens.data(), d_len.p, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, s0));
      HIPCHK3(hipMemcpyAsync(st.data(), d_st.p, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, s0));
      if (w8) HIPCHK3(hipMemcpyAsync(h8.data(), d8.p, (size_t)w8, hipMemcpyDeviceToHost, s0));
      HIPCHK3(hipStreamSynchronize(s0));
      for (int64_t k = 0; k < nt; ++k) {
        if (st[(size_t)k] == 0) {
          const uint8_t* src = h8.data() + tasks[(size_t)k].cons_off;
          results[(size_t)ids[(size_t)k]].assign(src, src + lens[(size_t)k]);
        } else if (st[(size_t)k] == 1 && pass == 0) {
          next.push_back(ids[(size_t)k]);
        } else {
          (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
          return SVDSS_ERANGE;   // internal inconsistency
        }
      }
    }
    todo.swap(next);
  }
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  unsigned long long cells = 0;
  HIPCHK3(hipMemcpyAsync(&cells, d_cells.p, 8, hipMemcpyDeviceToHost, s0));
  HIPCHK3(hipStreamSynchronize(s0));
  b->cells = (int64_t)cells;
  for (int64_t c = 0; c < n_clusters; ++c) {
    b->cons_len[(size_t)c] = (int64_t)results[(size_t)c].size();
    b->cons.insert(b->cons.end(), results[(size_t)c].begin(), results[(size_t)c].end());
  }
  return SVDSS_OK;
}

extern "C" int64_t svdss_poa_batch_nclusters(const svdss_poa_batch_t* b) { return b ? b->n_clusters : -1; }
extern "C" int64_t svdss_poa_batch_total(const svdss_poa_batch_t* b) { return b ? (int64_t)b->cons.size() : -1; }
extern "C" int64_t svdss_poa_batch_cells(const svdss_poa_batch_t* b) { return b ? b->cells : -1; }
extern "C" double svdss_poa_batch_kernel_ms(const svdss_poa_batch_t* b) { return b ? b->kernel_ms : -1.0; }
extern "C" int64_t svdss_poa_batch_hbm(const svdss_poa_batch_t* b) { return b ? b->n_hbm : -1; }
extern "C" int64_t svdss_poa_batch_quad_back(const svdss_poa_batch_t* b) { return b ? b->n_quad_back : -1; }
extern "C" int svdss_poa_batch_fetch(const svdss_poa_batch_t* b, int64_t* cons_len, uint8_t* cons) {
  if (!b) return SVDSS_EINVAL;
  if (cons_len) memcpy(cons_len, b->cons_len.data(), sizeof(int64_t) * b->cons_len.size());
  if (cons) memcpy(cons, b->cons.data(), b->cons.size());
  return SVDSS_OK;
}
extern "C" void svdss_poa_batch_free(svdss_poa_batch_t* b) { delete b; }
