// sfs_search.hip -- gfx950 kernels and C-ABI for the SFS-extraction hot path.
//
// Replaces PingPong::ping_pong_search + the rb3_fmd_* calls under it
// (/root/reference/ping_pong.cpp:4-49) and the per-read Assembler::assemble
// (/root/reference/assembler.cpp:34-56) for a whole batch of reads
// (PingPong::process_batch, ping_pong.cpp:176-209).
//
// Mapping: one work item (a read, or one of up to 8 segments of a read) per lane,
// persistent lanes that pull the next item from per-wavefront ticket pools when
// they finish.  Every loop iteration performs one memory operation per active
// lane -- a k-mer table entry, a BWT block (LF step), suffix-array rows, text --
// chosen by the flattened state machine of sfs_core2.h.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/svdss_hip.h"
#include "fmd_layout.h"
#include "hip_check.h"
#include "index_host.h"
#ifdef SV_COUNT_ITERS
// counting build: passes of sv_decide's loop, per wavefront (any lane inside)
extern __device__ unsigned long long g_sfs_iters[64];
#define SV_COUNT_PASS() do { if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(1))) atomicAdd(&g_sfs_iters[1], 1ULL); } while (0)
#endif
#include "sfs_core2.h"
#include "sym_window.h"
// profiling build (make searchprof -> libsvdss_hip_searchprof.so): shader-clock cycles per section of the main loop, summed over
// the wavefronts -- [0] tickets + item start, [1] sv_decide, [2] addresses + loads issued, [3..] the apply of each operation
// (which is where the wait for the loads lands), [15] passes
#ifdef SV_PROF
__device__ unsigned long long g_sfs_prof[16];
#define SV_PROF_DECL unsigned long long pr_t = __builtin_readcyclecounter(), pr_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define SV_PROF_LAP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); pr_acc[k] += t_ - pr_t; pr_t = t_; } while (0)
#define SV_PROF_OUT() do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 16; ++k_) atomicAdd(&g_sfs_prof[k_], pr_acc[k_]); } while (0)
#else
#define SV_PROF_DECL
#define SV_PROF_LAP(k)
#define SV_PROF_OUT()
#endif

// ---------------------------------------------------------------- kernels

struct SfsParams {
  SvdssDevIndex ix;
  const svdss_u4* chunks;   // reads viewed as 16-byte chunks
  int64_t max_chunk;
  const int64_t* offsets;   // n_reads + 1
  int64_t n_reads;
  uint2* rec;               // per-read record regions
  const int64_t* rec_base;  // explicit region starts (rerun pass) or nullptr
  const int64_t* rec_cap;   // explicit capacities (rerun pass) or nullptr
  int64_t* counts;          // n_reads + 1 (last stays 0)
  int64_t* n_ext;           // n_reads
  unsigned long long* next_read;
  int32_t assemble;
  // segmented search (small batches): every read is cut into n_seg segments, one lane each
  int32_t n_seg;            // 1: one lane per read; a power of two
  int32_t seg_shift;        // log2(n_seg)
  uint4* seg_rec;           // per-item raw records {qs, len, ext_at_begin, 0}
  SvSegInfo* seg_info;      // per item
  const int64_t* read_ids;  // fallback pass: the reads to search (nullptr: all)
  int64_t n_items;          // work items of this launch
  unsigned long long* n_fallback;   // stitch kernel: number of reads to redo unsegmented
  int64_t* fallback_ids;
  int32_t ticket_chunk;    // work-item tickets a wavefront takes from next_read at a time
  int32_t use_set;         // follow 2-4 occurrences in the text instead of walking the BWT (SV_OP_SET)
  int32_t use_bs;          // finish backward phases on deep intervals by binary search of the suffix array (SV_OP_BS_*)
  const int64_t* sub_ids;  // stitch / assemble kernels: the reads to process (nullptr: all n_reads)
  int64_t n_sub;
  int32_t* seg_take;       // per read and segment: [lo, hi) of the records that belong to the read's chain (-1: redo)
  uint32_t epoch;           // tag of the records written by this launch (see peek)
};

// segmented layout: item (r, j), j < seg_count(len): records at seg_region_base + j * seg_region_cap
__host__ __device__ inline int seg_count(int64_t len, int n_seg) {
  const int64_t c = len >> 8;
  return (int)(c < 1 ? 1 : (c < n_seg ? c : n_seg));
}
// segment j of cr covers read positions [j*sl, (j+1)*sl), the last one up to len; sl = len / cr
__host__ __device__ inline int32_t seg_lo_pos(int32_t sl, int j) { return j * sl; }
__host__ __device__ inline int32_t seg_start_pos(int32_t len, int32_t sl, int j, int cr) {
  return j == cr - 1 ? len - 1 : (j + 1) * sl - 1;
}
__host__ __device__ inline int64_t seg_region_base(int64_t off, int64_t r, int n_seg) {
  return (off >> 3) + (8 + 24 * (int64_t)n_seg) * r;
}
__host__ __device__ inline int64_t seg_region_cap(int64_t len, int cr) { return (len >> 3) / cr + 24; }

// default per-read record region: (len/8 + 8) records starting at off/8 + 8 r
__host__ __device__ inline int64_t rec_region_base(int64_t off, int64_t r) { return (off >> 3) + 8 * r; }
__host__ __device__ inline int64_t rec_region_cap(int64_t len) { return (len >> 3) + 8; }

// ---- v2: k-mer table + LF + unique-match TEXT mode (sfs_core2.h) -------------
struct __attribute__((packed, aligned(1))) SvU4u { uint32_t x, y, z, w; };  // 16 bytes, any alignment

__device__ __forceinline__ svdss_u4 sv_load16(const uint8_t* p) {
  const SvU4u v = *(const SvU4u*)p;
  svdss_u4 r;
  r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  return r;
}

// Longest-processing-time-first scheduling.  Lanes fetch items dynamically, so the launch ends when the last
// item that was started ends: the items that take 5-10 x the median (reads in repeats) must start first, not
// last.  Sixteen
// lanes per read sample 16 pairs of adjacent K-mers in the table; reads where at least two pairs occur more than
// once in the reference go to the front of the order, the rest to the back.  The order only decides when a
// read is searched, never what is found.
__global__ void __launch_bounds__(256) sfs_order_kernel(SfsParams p, int64_t* heavy) {
  // 16 lanes per read, one sample each
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const int64_t r = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 + grp;
  const bool live = r < p.n_reads;
  const int K = p.ix.k;
  const uint8_t* reads = (const uint8_t*)p.chunks;
  bool hit = false;
  if (live) {
    const int64_t off = p.offsets[r];
    const int64_t len = p.offsets[r + 1] - off;
    if (K >= 8 && K <= 16 && p.ix.table != nullptr && len >= 8 * K) {
      // the 2K symbols at an even spacing along the read (two 16-byte loads)
      const int64_t pos = off + ((len - 32) * (2 * sub + 1)) / 32;
      const svdss_u4 w0 = sv_load16(reads + pos), w1 = sv_load16(reads + pos + K);
      const uint32_t wa[4] = {w0.x, w0.y, w0.z, w0.w}, wb[4] = {w1.x, w1.y, w1.z, w1.w};
      uint32_t k1 = 0, k2 = 0, bad = 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const uint32_t a = ((wa[t >> 2] >> (8 * (t & 3))) & 0xffu) - 1u;
        const uint32_t c = ((wb[t >> 2] >> (8 * (t & 3))) & 0xffu) - 1u;
        if (t < K) { bad |= (a | c) & ~3u; k1 |= (a & 3u) << (2 * t); k2 |= (c & 3u) << (2 * t); }
      }
      if (!bad) hit = (p.ix.table[k1].info >> 62) >= SVDSS_TAB_MULTI && (p.ix.table[k2].info >> 62) >= SVDSS_TAB_MULTI;   // (2+ occurrences: MULTI or FEW)
    }
  }
  const unsigned long long hm = __ballot(hit);
  const int hits = __builtin_popcountll((hm >> (16 * grp)) & 0xffffULL);
  if (live && sub == 0) heavy[r] = hits >= 2 ? 1 : 0;
}

// stable partition by the flags: heavy reads first (scan = exclusive prefix sum of heavy, n_reads + 1 entries)
__global__ void __launch_bounds__(256) sfs_order_scatter_kernel(int64_t n_reads, const int64_t* heavy, const int64_t* scan,
                                                                int64_t* order) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const int64_t before = scan[r], n_heavy = scan[n_reads];
  order[heavy[r] ? before : n_heavy + (r - before)] = r;
}

typedef uint32_t sv_u32x4 __attribute__((ext_vector_type(4)));

#ifdef SV_COUNT_ITERS
__device__ unsigned long long g_sfs_iters[64];   // [0] wave iterations, [1] passes of the decision loop, [2+op] lane ops by type, [32+op] wave iterations with a lane in op
#endif

// counting build (make count -> libsvdss_hip_count.so, SVDSS_LIB selects it): lane operations of the launches since
// the last report, by type -- what `useful_bytes` in profiles/traffic.json is computed from
static void sfs_report_op_counts() {
#ifdef SV_PROF
  {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sfs_prof), sizeof h) == hipSuccess) {
      double tot = 0;
      for (int k = 0; k < 15; ++k) tot += (double)h[k];
      fprintf(stderr, "[svdss] search kernel, cycles per pass of a wavefront (%llu passes, %.0f cycles each): tickets/start %.0f decide %.0f addresses+issue %.0f | apply LF %.0f TABLE %.0f "
              "SA %.0f TEXT %.0f SET %.0f SA_SET %.0f FILL %.0f | loop edge %.0f\n", h[15], tot / (double)h[15], (double)h[0] / h[15], (double)h[1] / h[15], (double)h[2] / h[15],
              (double)h[3] / h[15], (double)h[4] / h[15], (double)h[5] / h[15], (double)h[6] / h[15], (double)h[7] / h[15], (double)h[8] / h[15], (double)h[9] / h[15], (double)h[14] / h[15]);
    }
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sfs_prof), h, sizeof h);
  }
#endif
#ifdef SV_COUNT_ITERS
  unsigned long long h[64];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sfs_iters), sizeof h) == hipSuccess) {
    fprintf(stderr, "[svdss] decision-loop passes %llu; wave-iterations with a lane in: DONE %llu LF %llu TABLE %llu SA %llu TEXT %llu FILL %llu SLOW %llu "
            "PEEK %llu SA_SET %llu SET %llu BS_SA %llu BS_TEXT %llu BS_ORD %llu\n", h[1], h[32], h[33], h[34], h[35], h[36], h[37], h[38], h[39], h[40], h[41],
            h[42], h[43], h[45]);
    fprintf(stderr, "[svdss] wave-iterations %llu; lane ops: DONE %llu LF %llu TABLE %llu SA %llu TEXT %llu FILL %llu SLOW %llu "
            "PEEK %llu SA_SET %llu SET %llu BS_SA %llu BS_TEXT %llu BS_ORD %llu\n", h[0], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11],
            h[12], h[13], h[15]);
  }
  fprintf(stderr, "[svdss] items by ops (2^k .. 2^(k+1)-1, k=4..15):");
  for (int k = 4; k < 16; ++k) fprintf(stderr, " %llu", h[16 + k]);
  fprintf(stderr, "\n");
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sfs_iters), h, sizeof h);
#endif
}

// BS: the instantiation that can finish deep backward phases by binary search (sfs_core2.h).  It is the kernel for
// references rich in repeats (svdss_index::deep_frac, estimated when the k-mer table is built); the plain one carries none
// of that code (its branches cost the headline workload ~4 % when compiled in).  Both the one-lane-per-read and the
// segmented launches have it since round 5 (round 4 left the segmented one out because its text differed from run to
// run: a hardware hazard behind the inline-asm store of emit(), see there -- not the BS code, which never ran).
template <class P, bool SEG, bool BS>
#ifndef SV_SEARCH_OCC
#define SV_SEARCH_OCC 4   // workgroups per CU the register budget is set for
#endif
__global__ void __launch_bounds__(256, SV_SEARCH_OCC) sfs_search2_kernel(SfsParams p) {
  __shared__ uint32_t ring_lds[16 * 256];   // 64 read symbols per lane: row r of lane t at [r*256 + t]
  SvRing g;
  g.base = &ring_lds[threadIdx.x];
  g.stride = 256;
  SvLane<P> st;
  int64_t r = 0, off = 0, base = 0, cap = 0, item = 0;
  int32_t nb_cur = 0;       // SEG: cursor into the left neighbour's records
  bool has_left = false;
  bool active = false;
  const bool assemble = SEG ? false : p.assemble != 0;   // segments produce raw SFS; the stitcher assembles
  const uint8_t* reads = (const uint8_t*)p.chunks;
  const uint8_t* blocks = (const uint8_t*)p.ix.blocks;

  __shared__ uint32_t stash_lds[SEG ? 3 * 256 : 1];   // SEG: a record waiting for its pair (qs, l, ext_at_begin)
  uint32_t* stash = &stash_lds[SEG ? threadIdx.x : 0];
  auto emit = [&](int32_t idx, int32_t qs, int32_t l) {
    if (idx < cap) {
      if (SEG)   // ext_at_begin: the forward phase of this SFS made pos - begin of the extensions
      {
        const uint32_t ext_at_begin = (uint32_t)(st.n_ext - (st.pos - st.begin));
        if (idx < SV_PEEK_VISIBLE) {
          // the first records of a segment are the ones its right neighbour peeks at (it synchronises on the first
          // SFS below the boundary): one 16-byte agent-scope (write-through) store, visible to lanes on other XCDs
          // (a torn or late view only lengthens the neighbour's overrun or sends the read to the exact fallback)
          // The two wait states behind the store are part of it.  A vector-memory store of more than 8 bytes reads its
          // data registers AFTER it has issued; a vector instruction that writes one of them has to keep two wait states
          // (gfx940 and later; one before) behind it.  For its own instructions the compiler's hazard recogniser does
          // that -- it cannot see inside this asm.  Round 4's "segmented kernel with the BS code compiled in gives
          // run-to-run different text" was this: in that instantiation the register allocation puts `v_mov_b32 v3, 1`
          // one instruction (an exec restore) behind the store of v[2:5], so the record's length went to memory as 1 or
          // as itself depending on how long the memory pipeline took to fetch the data -- i.e. on load, hence four
          // concurrent launches.  In the plain instantiation the next writers of v[2:5] happened to be far away.
          const sv_u32x4 v = {(uint32_t)qs, (uint32_t)l, ext_at_begin, p.epoch};
#ifndef SV_NO_STORE_NOP   // (developer build `make hazard`: the store as round 4 had it, for tools/seg_stress.py to show the difference)
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p.seg_rec + (base + idx)), "v"(v) : "memory");
#else
          asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p.seg_rec + (base + idx)), "v"(v) : "memory");
#endif
        } else if ((((base + idx) & 1) == 0)) {
          // the rest is only read after the launch (stitch / assemble).  A 16-byte record written on its own is one
          // 32-byte memory transaction (the line leaves L2 long before the lane's next record): records wait in LDS
          // for their odd neighbour and the pair is stored back to back
          stash[0] = (uint32_t)qs; stash[256] = (uint32_t)l; stash[512] = ext_at_begin;
        } else {
          if (idx > SV_PEEK_VISIBLE)
            p.seg_rec[base + idx - 1] = make_uint4(stash[0], stash[256], stash[512], p.epoch);
          p.seg_rec[base + idx] = make_uint4((uint32_t)qs, (uint32_t)l, ext_at_begin, p.epoch);
        }
      }
      else
        p.rec[base + idx] = make_uint2((uint32_t)qs, (uint32_t)l);
    }
  };

  // The next item is fetched while the current one is being finished (ticket, read id, offsets: three
  // dependent loads that ride along with the lane's memory operation of three iterations) -- fetched
  // inline they stall the whole wavefront every time one of its 64 lanes starts an item.
  // The prefetched read id / offset / length wait in LDS (the kernel has no VGPR to spare).
  __shared__ int64_t set_lds[SV_SET_MAX * 256];   // SET mode: text index deltas of up to 4 occurrences per lane
  const SvSet ts{&set_lds[threadIdx.x], 256};
  __shared__ uint32_t pf_lds[4 * 256];
  uint32_t* pfl = &pf_lds[threadIdx.x];   // rows: read id, offset lo, offset hi, length
  // BS mode: the text positions of the '$' (suffix array rows 0 .. n_dollar - 1), sorted -- which record pair a text
  // position lies in decides where its mirror image in the other strand is (sv_mirror)
  __shared__ int64_t dsort_lds[BS ? SV_BS_MAX_DOLLAR : 1];
  const int n_d = BS && p.use_bs ? p.ix.n_dollar : 0;
  if (n_d > 0) {
    __shared__ int64_t dtmp_lds[BS ? SV_BS_MAX_DOLLAR : 1];
    if ((int)threadIdx.x < n_d) dtmp_lds[threadIdx.x] = (int64_t)((const P*)p.ix.sa)[threadIdx.x];
    __syncthreads();
    if ((int)threadIdx.x < n_d) {
      const int64_t v = dtmp_lds[threadIdx.x];
      int rank = 0;
      for (int j = 0; j < n_d; ++j) rank += dtmp_lds[j] < v ? 1 : 0;
      dsort_lds[rank] = v;
    }
    __syncthreads();
  }
#ifdef SV_COUNT_ITERS
  uint32_t item_ops = 0;
#endif
  int pf = 0;                       // 0 nothing, 1 ticket, 2 + read id, 3 + offsets
  uint32_t nt = 0;                  // ticket (n_items < 2^31: checked by the host)
  auto pf_read_id = [&]() {
    const uint32_t slot = SEG ? nt >> p.seg_shift : nt;
    pfl[0] = p.read_ids ? (uint32_t)p.read_ids[slot] : slot;
  };
  auto pf_offsets = [&]() {
    const int64_t rr = (int64_t)pfl[0];
    const int64_t o0 = p.offsets[rr], o1 = p.offsets[rr + 1];
    pfl[256] = (uint32_t)o0; pfl[512] = (uint32_t)((uint64_t)o0 >> 32); pfl[768] = (uint32_t)(o1 - o0);
  };
  // Tickets come from a per-wavefront pool refilled ticket_chunk at a time: one atomic on the global counter per
  // chunk instead of one per item (2 M items serialise on that one address for ~20 ms otherwise).
  uint32_t pool_next = 0, pool_end = 0;   // wave-uniform
  SV_PROF_DECL;
  for (;;) {
    SV_PROF_LAP(14);
#ifdef SV_PROF
    pr_acc[15] += 1;
#endif
    {
      const int lane = threadIdx.x & 63;
      const bool want = pf == 0 && (!active || st.pos - st.stop_lo < 256);   // idle, or close to the end of its item
      const unsigned long long wm = __ballot(want);
      if (wm) {
        const uint32_t c = (uint32_t)__builtin_popcountll(wm);
        const uint32_t rank = (uint32_t)__builtin_popcountll(wm & ((1ULL << lane) - 1ULL));
        const uint32_t avail = pool_end - pool_next;
        if (c > avail) {
          const uint32_t take = c - avail > (uint32_t)p.ticket_chunk ? c - avail : (uint32_t)p.ticket_chunk;
          const int leader = (int)__builtin_ctzll(wm);
          uint32_t nb = 0;
          if (lane == leader) nb = (uint32_t)atomicAdd(p.next_read, (unsigned long long)take);
          nb = (uint32_t)__shfl((int)nb, leader, 64);
          if (want) nt = rank < avail ? pool_next + rank : nb + (rank - avail);
          pool_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nb + (c - avail)));   // (kept in SGPRs)
          pool_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nb + take));
        } else {
          if (want) nt = pool_next + rank;
          pool_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pool_next + c));
        }
        if (want) pf = 1;
      }
    }
    if (!active) {
      if (nt >= (uint32_t)p.n_items) break;
      if (pf < 2) pf_read_id();
      if (pf < 3) pf_offsets();
      pf = 0;
      item = (int64_t)nt;   // consecutive items = the segments of one read, left to right
      r = (int64_t)pfl[0];
      off = (int64_t)((uint64_t)pfl[256] | ((uint64_t)pfl[512] << 32));
      const int64_t len = (int64_t)pfl[768];
      if (SEG) {
        const int j = (int)(item & (p.n_seg - 1));
        const int cr = seg_count(len, p.n_seg);
        if (j >= cr) {                       // short read: fewer segments than lanes reserved for it
          SvSegInfo z; z.n_rec = 0; z.cap = 0; z.ext_total = 0; z.complete = 0;
          p.seg_info[(r << p.seg_shift) + j] = z;
          continue;
        }
        cap = seg_region_cap(len, cr);
        base = seg_region_base(off, r, p.n_seg) + j * cap;
        has_left = j > 0;
        nb_cur = 0;
        const int32_t sl = (int32_t)((uint32_t)len / (uint32_t)cr);
        sv_lane_init(st, (int32_t)len, seg_start_pos((int32_t)len, sl, j, cr), seg_lo_pos(sl, j));
      } else {
        base = p.rec_base ? p.rec_base[r] : rec_region_base(off, r);
        cap = p.rec_cap ? p.rec_cap[r] : rec_region_cap(len);
        sv_lane_init(st, (int32_t)len);
      }
      active = true;
    }
    SV_PROF_LAP(0);
    const SvOp o = sv_decide(st, p.ix, g, off, assemble, emit, SEG && has_left, p.use_set != 0, BS);
    SV_PROF_LAP(1);
#ifdef SV_COUNT_ITERS
    if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(1))) atomicAdd(&g_sfs_iters[0], 1ULL);
    atomicAdd(&g_sfs_iters[2 + o.op], 1ULL);
    for (int q = 0; q < SV_N_OPS; ++q) {
      const unsigned long long bm = __ballot(o.op == q);
      if (bm && (threadIdx.x & 63) == __builtin_ctzll(bm)) atomicAdd(&g_sfs_iters[32 + q], 1ULL);
    }
    ++item_ops;
    if (o.op == SV_OP_DONE) { atomicAdd(&g_sfs_iters[16 + (31 - __builtin_clz(item_ops | 1))], 1ULL); item_ops = 0; }
#endif
    if (SEG && o.op != SV_OP_DONE && st.n_sfs > (int32_t)cap) {
      // More SFS than the segment's region holds: the stitcher sends the whole read to the next level whatever this
      // lane does from here on (sv_stitch: n_rec > cap), so it stops now.  A read with a long novel insertion has an
      // SFS at every base of it -- the lane that owns that stretch used to walk all of it, 1,900 dependent SFS for
      // nothing, and the launch lasted as long as that lane (8.8 ms per batch of `SVDSS search` in the chain bench).
      st.mode |= SV_M_PARTIAL;
      continue;
    }
    if (o.op == SV_OP_DONE) {
      if (SEG) {
        {   // a record still waiting for its pair
          const int32_t last = (st.n_sfs < (int32_t)cap ? st.n_sfs : (int32_t)cap) - 1;
          if (last >= SV_PEEK_VISIBLE && ((base + last) & 1) == 0)
            p.seg_rec[base + last] = make_uint4(stash[0], stash[256], stash[512], p.epoch);
        }
        SvSegInfo z;
        z.n_rec = st.n_sfs; z.cap = (int32_t)cap; z.ext_total = st.n_ext;
        z.complete = (st.mode & SV_M_PARTIAL) ? 0 : 1;
        p.seg_info[(r << p.seg_shift) + (item & (p.n_seg - 1))] = z;
      } else {
        sv_flush(st, assemble, emit);
        p.counts[r] = st.n_sfs;
        p.n_ext[r] = st.n_ext;
      }
      active = false;
      continue;
    }
    if (o.op == SV_OP_TEXT_SLOW) {   // only the first 64 bytes of the whole batch
      sv_apply_text_slow(st, p.ix.text, reads, off);
      continue;
    }
    if (pf >= 1 && pf < 3) {   // the next item's ticket is here: get its read id, then its offsets, under way
      if (nt < (uint32_t)p.n_items) {
        if (pf == 1) pf_read_id(); else pf_offsets();
      }
      ++pf;
    }
    // one memory operation per lane: up to 4 x 16 B from pa and 4 x 16 B from pb
    const uint8_t* pa = blocks;
    const uint8_t* pb = blocks;
    bool wide_a = false, need_b = false;
    int64_t c0 = 0;
    if (o.op == SV_OP_LF) {
      const int64_t blo = (int64_t)st.lo >> SVDSS_BLOCK_SHIFT, bhi = (int64_t)st.hi >> SVDSS_BLOCK_SHIFT;
      pa = blocks + blo * SVDSS_BLOCK_BYTES;
      pb = blocks + bhi * SVDSS_BLOCK_BYTES;
      wide_a = true;
      need_b = bhi != blo;
    } else if (o.op == SV_OP_TABLE) {
      pa = (const uint8_t*)(p.ix.table + o.a);
    } else if (o.op == SV_OP_SA || (BS && o.op == SV_OP_BS_SA)) {
      pa = (const uint8_t*)p.ix.sa + o.a * (int64_t)sizeof(P);
    } else if (o.op == SV_OP_TEXT) {
      pa = p.ix.text + st.tdelta + st.pos - 64;
      pb = reads + off + st.pos - 64;
      wide_a = true;
      need_b = true;
    } else if (BS && o.op == SV_OP_BS_TEXT) {
      const int64_t cp = (int64_t)st.pos + p.ix.k - st.bs_m;   // read positions [cp - 64, cp) against the text of the middle row
      pa = p.ix.text + st.tdelta + cp - 64;
      pb = reads + off + cp - 64;
      wide_a = true;
      need_b = true;
    } else if (o.op == SV_OP_PEEK) {
      // the left neighbour's records sit right below this lane's region, tagged with this launch's epoch as
      // they are produced; the segments of a read are fetched back to back, so the neighbour is normally far ahead
      int32_t i0 = nb_cur;
      if (i0 > (int32_t)cap - SV_PEEK_RECS) i0 = (int32_t)cap - SV_PEEK_RECS;   // stay inside its region
      if (i0 < 0) i0 = 0;
      pa = (const uint8_t*)(p.seg_rec + (base - cap + i0));
      c0 = i0;
    } else if (o.op == SV_OP_SET || o.op == SV_OP_SA_SET || (BS && o.op == SV_OP_BS_ORD)) {
      // (their loads are issued below, next to the others)
    } else {  // SV_OP_FILL
      c0 = o.a;
      if (c0 > p.max_chunk - 3) c0 = p.max_chunk - 3;
      if (c0 < 0) c0 = 0;
      pb = reads + 16 * c0;
      need_b = true;
    }
    svdss_u4 A[4], B[4];
    if (o.op == SV_OP_PEEK) {
      // agent-scope loads: the records were stored (write-through) by lanes that may sit on another XCD
#pragma unroll
      for (int i = 0; i < SV_PEEK_RECS; ++i) {
        const unsigned long long* rp = (const unsigned long long*)pa + 2 * i;
        const unsigned long long lo = __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(rp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        A[i].x = (uint32_t)lo; A[i].y = (uint32_t)(lo >> 32); A[i].z = (uint32_t)hi; A[i].w = (uint32_t)(hi >> 32);
      }
    } else if (o.op == SV_OP_SET) {
      const int alive = (st.mode >> SV_SET_SHIFT) & ((1 << SV_SET_MAX) - 1);
#pragma unroll
      for (int i = 0; i < SV_SET_MAX; ++i) {
        A[i].x = A[i].y = A[i].z = A[i].w = 0;
        if ((alive >> i) & 1) A[i] = sv_load16(p.ix.text + ts.base[i * ts.stride] + st.pos - SV_SET_WIN);
      }
      B[0] = sv_load16(reads + off + st.pos - SV_SET_WIN);
    } else if (BS && o.op == SV_OP_BS_ORD) {
      // the two symbols a comparison stopped at (their lines were fetched a moment ago)
      const int64_t rp = (int64_t)st.pos + p.ix.k - 1 - st.bs_m;
      A[0] = sv_load16(p.ix.text + st.tdelta + rp);
      B[0] = sv_load16(reads + off + rp);
    } else if (o.op == SV_OP_SA_SET) {
      const int n_occ = (int)(st.hi - st.lo);
#pragma unroll
      for (int i = 0; i < SV_SET_MAX; ++i) {   // (entries past the interval are not used: read the first one again)
        const P v = ((const P*)p.ix.sa)[(int64_t)st.lo + (i < n_occ ? i : 0)];
        A[i].x = (uint32_t)v;
        A[i].y = (uint32_t)((uint64_t)v >> 32);
      }
    } else if (o.op != SV_OP_FILL) A[0] = sv_load16(pa);
    if (wide_a) {
      A[1] = sv_load16(pa + 16);
      A[2] = sv_load16(pa + 32);
      A[3] = sv_load16(pa + 48);
    }
    if (need_b) {
      B[0] = sv_load16(pb);
      B[1] = sv_load16(pb + 16);
      B[2] = sv_load16(pb + 32);
      B[3] = sv_load16(pb + 48);
    }
    SV_PROF_LAP(2);
    if (o.op == SV_OP_LF) {
      sv_apply_lf(st, p.ix, A, B, !need_b);
      SV_PROF_LAP(3);
    } else if (o.op == SV_OP_TABLE) {
      sv_apply_table(st, p.ix, (uint64_t)A[0].x | ((uint64_t)A[0].y << 32),
                     (uint64_t)A[0].z | ((uint64_t)A[0].w << 32), g, off, p.use_set != 0 && off >= 64, BS && n_d > 0 && off >= 64);
      SV_PROF_LAP(4);
    } else if (o.op == SV_OP_SA) {
      const int64_t tp = sizeof(P) == 4 ? (int64_t)A[0].x
                                        : (int64_t)((uint64_t)A[0].x | ((uint64_t)A[0].y << 32));
      sv_apply_sa(st, tp);
      SV_PROF_LAP(5);
    } else if (o.op == SV_OP_TEXT) {
      sv_apply_text(st, A, B);
      SV_PROF_LAP(6);
    } else if (BS && o.op == SV_OP_BS_SA) {
      const int64_t tp = sizeof(P) == 4 ? (int64_t)A[0].x
                                        : (int64_t)((uint64_t)A[0].x | ((uint64_t)A[0].y << 32));
      sv_apply_bs_sa(st, p.ix, tp, dsort_lds, n_d);
    } else if (BS && o.op == SV_OP_BS_TEXT) {
      sv_apply_bs_text(st, p.ix, A, B);
    } else if (BS && o.op == SV_OP_BS_ORD) {
      sv_apply_bs_ord(st, p.ix, (int)(B[0].x & 0xffu), (int)(A[0].x & 0xffu));
    } else if (o.op == SV_OP_SET) {
      sv_apply_set(st, ts, A, B[0]);
      SV_PROF_LAP(7);
    } else if (o.op == SV_OP_SA_SET) {
      int64_t tp[SV_SET_MAX];
#pragma unroll
      for (int i = 0; i < SV_SET_MAX; ++i) tp[i] = (int64_t)((uint64_t)A[i].x | ((uint64_t)A[i].y << 32));
      sv_apply_sa_set(st, ts, tp);
      SV_PROF_LAP(8);
    } else if (o.op == SV_OP_PEEK) {
      int32_t q[SV_PEEK_RECS];
      bool written[SV_PEEK_RECS];
      const int32_t sh = nb_cur - (int32_t)c0;   // records of the window already skipped (window clamped to the region)
#pragma unroll
      for (int i = 0; i < SV_PEEK_RECS; ++i) {
        const int k = i + sh;
        const uint32_t qq = k == 0 ? A[0].x : k == 1 ? A[1].x : k == 2 ? A[2].x : A[3].x;
        const uint32_t ee = k == 0 ? A[0].w : k == 1 ? A[1].w : k == 2 ? A[2].w : A[3].w;
        q[i] = (int32_t)qq;
        written[i] = k < SV_PEEK_RECS && ee == p.epoch;
      }
      sv_apply_peek(st, q, written, nb_cur, (int32_t)cap);
    } else {
      sv_ring_fill(g, c0, B);
      st.wrel = (int32_t)(16 * c0 - off);
      SV_PROF_LAP(9);
    }
  }
  SV_PROF_OUT();
}

// One lane per read: stitch the chains of its segments (sv_stitch), run the streaming
// assembler over the stitched chain and leave the records in the read's default region, exactly
// what the unsegmented kernel would have written.  Reads whose overrun did not reach the shared
// SFS start are queued for an unsegmented search.
__global__ void __launch_bounds__(256) sfs_stitch_kernel(SfsParams p) {
  // phase 1, one thread per read: find the shared SFS starts (a short walk over the records next to each
  // segment boundary) and leave the record range taken from every segment in seg_take
  const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ti >= (p.sub_ids ? p.n_sub : p.n_reads)) return;
  const int64_t r = p.sub_ids ? p.sub_ids[ti] : ti;
  const int64_t off = p.offsets[r];
  const int64_t len = p.offsets[r + 1] - off;
  const int cr = seg_count(len, p.n_seg);
  const int64_t icap = seg_region_cap(len, cr);
  const int64_t ibase = seg_region_base(off, r, p.n_seg);
  SvSegInfo info[16];
  int32_t seg_lo[16], tlo[16], thi[16];
  const int32_t sl = (int32_t)((uint32_t)len / (uint32_t)cr);
  for (int j = 0; j < cr; ++j) {
    info[j] = p.seg_info[r * p.n_seg + j];
    seg_lo[j] = seg_lo_pos(sl, j);
  }
  auto get = [&](int sg, int32_t i, int32_t& q, int32_t& e) {
    const uint4 v = p.seg_rec[ibase + sg * icap + i];
    q = (int32_t)v.x;
    e = (int32_t)v.z;
  };
  int64_t ext = 0;
  int32_t* take = p.seg_take + 2 * r * p.n_seg;
  if (!sv_stitch(cr, info, seg_lo, get, tlo, thi, &ext)) {
    const unsigned long long k = atomicAdd(p.n_fallback, 1ULL);
    p.fallback_ids[k] = r;
    p.counts[r] = 0;
    p.n_ext[r] = 0;
    take[0] = -1;
    return;
  }
  for (int j = 0; j < cr; ++j) { take[2 * j] = tlo[j]; take[2 * j + 1] = thi[j]; }
  p.n_ext[r] = ext;
}

__global__ void __launch_bounds__(256) sfs_assemble_kernel(SfsParams p) {
  // phase 2, one wavefront per read: run the assembler over the stitched chain and leave the records in the
  // read's default region, exactly what the unsegmented kernel would have written.
  // The chain in production order: the taken records of segment cr-1, then cr-2, ... (descending qs).
  // Streaming Assembler::assemble (sv_emit) opens a new assembled SFS at record i iff
  // qs_i + l_i <= qs_(i-1), and an assembled SFS is (qs of its last record, end of its first record):
  // both are functions of adjacent records, so 64 records are assembled per step.
  __shared__ int32_t s_cum[4][17], s_lo[4][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t wi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wi >= (p.sub_ids ? p.n_sub : p.n_reads)) return;
  const int64_t r = p.sub_ids ? p.sub_ids[wi] : wi;
  const int32_t* take = p.seg_take + 2 * r * p.n_seg;
  if (take[0] < 0) return;   // queued for an unsegmented search
  const int64_t off = p.offsets[r];
  const int64_t len = p.offsets[r + 1] - off;
  const int cr = seg_count(len, p.n_seg);
  const int64_t icap = seg_region_cap(len, cr);
  const int64_t ibase = seg_region_base(off, r, p.n_seg);
  const int64_t base = rec_region_base(off, r);
  const int64_t cap = rec_region_cap(len);
  int32_t* cum = s_cum[wv];
  int32_t* tlo = s_lo[wv];
  if (lane == 0) {
    int32_t c = 0;
    cum[0] = 0;
    for (int k = 0; k < cr; ++k) {   // k-th segment of the chain = segment cr-1-k
      const int sg = cr - 1 - k;
      tlo[k] = take[2 * sg];
      c += take[2 * sg + 1] - take[2 * sg];
      cum[k + 1] = c;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int32_t M = cum[cr];
  auto load = [&](int32_t g, int32_t& q, int32_t& l) {
    int k = 0;
    while (k + 1 < cr && g >= cum[k + 1]) ++k;
    const uint4 v = p.seg_rec[ibase + (int64_t)(cr - 1 - k) * icap + tlo[k] + (g - cum[k])];
    q = (int32_t)v.x;
    l = (int32_t)v.y;
  };
  if (!p.assemble) {
    for (int32_t g = lane; g < M; g += 64) {
      int32_t q, l;
      load(g, q, l);
      if (g < cap) p.rec[base + g] = make_uint2((uint32_t)q, (uint32_t)l);
    }
    if (lane == 0) p.counts[r] = M;
    return;
  }
  int32_t runs = 0;        // assembled SFS opened so far
  int32_t open_end = 0;    // end of the first record of the open one
  int32_t prev_q = 0;      // qs of the last record of the previous step
  int32_t q_pre = 0, l_pre = 0;   // the records of the next step are loaded while this one is assembled
  if (lane < M) load(lane, q_pre, l_pre);
  for (int32_t g0 = 0; g0 < M; g0 += 64) {
    const int32_t g = g0 + lane;
    const bool in = g < M;
    const int32_t q = q_pre, l = l_pre;
    q_pre = 0; l_pre = 0;
    if (g + 64 < M) load(g + 64, q_pre, l_pre);
    const bool has_next = g + 1 < M;
    int32_t qn = __shfl_down(q, 1, 64), ln = __shfl_down(l, 1, 64);
    {   // lane 63's successor is lane 0 of the next step
      const int32_t q0 = __shfl(q_pre, 0, 64), l0 = __shfl(l_pre, 0, 64);
      if (lane == 63) { qn = q0; ln = l0; }
    }
    int32_t qp = __shfl_up(q, 1, 64);
    if (lane == 0) qp = prev_q;
    const bool flag = in && (g == 0 || q + l <= qp);
    const bool last = in && (!has_next || qn + ln <= q);
    const unsigned long long fm = __ballot(flag);
    const unsigned long long le = lane == 63 ? ~0ULL : ((1ULL << (lane + 1)) - 1ULL);
    const unsigned long long mine = fm & le;
    const int32_t run_id = runs + __builtin_popcountll(mine) - 1;
    const int start_lane = mine ? 63 - __builtin_clzll(mine) : -1;
    const int32_t e_start = __shfl(q + l, start_lane < 0 ? 0 : start_lane, 64);
    const int32_t e_first = start_lane < 0 ? open_end : e_start;
    if (last && run_id < cap) p.rec[base + run_id] = make_uint2((uint32_t)q, (uint32_t)(e_first - q));
    if (fm) open_end = __shfl(q + l, 63 - __builtin_clzll(fm), 64);
    runs += __builtin_popcountll(fm);
    const int nin = M - g0 < 64 ? M - g0 : 64;
    prev_q = __shfl(q, nin - 1, 64);
  }
  if (lane == 0) p.counts[r] = runs;
}

// One wavefront per read: copy its records from the region to the compact
// output (reversed when assembled: the lanes produce chains in descending qs,
// Assembler::assemble returns ascending, assembler.cpp:36).
__global__ void __launch_bounds__(256) sfs_gather_kernel(SfsParams p, const int64_t* out_off,
                                                         int32_t* out_qs, int32_t* out_len,
                                                         unsigned long long* n_overflow) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < p.n_reads; r += nwaves) {
    const int64_t off = p.offsets[r];
    const int64_t len = p.offsets[r + 1] - off;
    const int64_t base = p.rec_base ? p.rec_base[r] : rec_region_base(off, r);
    const int64_t cap = p.rec_cap ? p.rec_cap[r] : rec_region_cap(len);
    const int64_t cnt = p.counts[r];
    if (cnt > cap) {
      if (lane == 0) atomicAdd(n_overflow, 1ULL);
      continue;
    }
    const int64_t o = out_off[r];
    for (int64_t i = lane; i < cnt; i += 64) {
      const uint2 v = p.rec[base + (p.assemble ? cnt - 1 - i : i)];
      out_qs[o + i] = (int32_t)v.x;
      out_len[o + i] = (int32_t)v.y;
    }
  }
}

// ------------------------------------------------------------------ batch

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct svdss_sfs_batch {
  int device = -1;
  int64_t n_reads = 0;
  int64_t total = 0;
  int64_t total_ext = 0;
  double kernel_ms = 0.0;
  double search_ms = 0.0;   // the segmented / one-lane-per-read search kernel of pass 0 alone
  DevBuf rec, counts, n_ext, out_off, out_qs, out_len, tmp, misc, reads, offsets, base2, sum;
  DevBuf seg_rec, seg_info, fallback, fallback2, seg_take, order, order_cnt, tiny;
  int64_t n_fallback = 0;   // reads of the last call that were redone unsegmented
  int32_t n_seg = 1;        // segments per read used by the last call
  int32_t used_bs = 0;      // the last call launched the BS instantiation
  uint32_t epoch = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ek0 = nullptr, ek1 = nullptr;
  // host-buffer entry points: the batch object's own non-blocking stream (copies in, kernels, copies out), so that
  // calls on different batch objects -- e.g. two threads feeding the GPU from a BAM file -- overlap
  hipStream_t own_stream = nullptr;
  DevBuf packed, byte_off, lens32;   // svdss_sfs_search_batch_bam: the 4-bit bases as they sit in the BAM records
};

static int ensure(DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return SVDSS_OK;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + (bytes >> 3) + 256;
  HIPCHK(hipMalloc(&b.p, want));
  b.cap = want;
  return SVDSS_OK;
}

static void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

extern "C" void svdss_sfs_batch_free(svdss_sfs_batch_t* b) {
  if (!b) return;
  if (b->device >= 0) (void)hipSetDevice(b->device);
  for (DevBuf* d : {&b->rec, &b->counts, &b->n_ext, &b->out_off, &b->out_qs, &b->out_len, &b->tmp,
                    &b->misc, &b->reads, &b->offsets, &b->base2, &b->sum, &b->seg_rec, &b->seg_info,
                    &b->fallback, &b->fallback2, &b->seg_take, &b->order, &b->order_cnt, &b->tiny, &b->packed, &b->byte_off, &b->lens32})
    release(*d);
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->ek0) (void)hipEventDestroy(b->ek0);
  if (b->ek1) (void)hipEventDestroy(b->ek1);
  delete b;
}

extern "C" int64_t svdss_sfs_batch_nreads(const svdss_sfs_batch_t* b) { return b ? b->n_reads : -1; }
extern "C" int64_t svdss_sfs_batch_total(const svdss_sfs_batch_t* b) { return b ? b->total : -1; }
extern "C" int64_t svdss_sfs_batch_total_ext(const svdss_sfs_batch_t* b) { return b ? b->total_ext : -1; }
extern "C" double svdss_sfs_batch_kernel_ms(const svdss_sfs_batch_t* b) { return b ? b->kernel_ms : -1.0; }
extern "C" double svdss_sfs_batch_search_kernel_ms(const svdss_sfs_batch_t* b) { return b ? b->search_ms : -1.0; }
extern "C" int32_t svdss_sfs_batch_segments(const svdss_sfs_batch_t* b) { return b ? b->n_seg : -1; }
extern "C" int64_t svdss_sfs_batch_fallbacks(const svdss_sfs_batch_t* b) { return b ? b->n_fallback : -1; }
extern "C" int32_t svdss_sfs_batch_used_bs(const svdss_sfs_batch_t* b) { return b ? b->used_bs : -1; }

static int launch_grid(int device, int* blocks_out) {
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  *blocks_out = prop.multiProcessorCount * 8;  // 8 x 256 threads = 32 waves per CU
  return SVDSS_OK;
}

extern "C" int svdss_sfs_search_batch_device(const svdss_index_t* ix, const uint8_t* d_reads,
                                             const int64_t* d_offsets, int64_t n_reads,
                                             int64_t total_syms, int32_t flags, void* stream_,
                                             svdss_sfs_batch_t** out) {
  if (!ix || !out || n_reads < 0 || total_syms < 0) return SVDSS_EINVAL;
  if (n_reads > 0 && (!d_reads || !d_offsets)) return SVDSS_EINVAL;
  if (((uintptr_t)d_reads & 15u) != 0) return SVDSS_EINVAL;
  if (ix->device < 0 || !ix->d_blocks) return SVDSS_ENODEV;
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(hipSetDevice(ix->device));
  svdss_sfs_batch* b = *out;
  if (!b) {
    b = new (std::nothrow) svdss_sfs_batch();
    if (!b) return SVDSS_ENOMEM;
    b->device = ix->device;
    *out = b;
  }
  if (b->device != ix->device) return SVDSS_EINVAL;
  if (!b->ev0) HIPCHK(hipEventCreate(&b->ev0));
  if (!b->ev1) HIPCHK(hipEventCreate(&b->ev1));
  if (!b->ek0) HIPCHK(hipEventCreate(&b->ek0));
  if (!b->ek1) HIPCHK(hipEventCreate(&b->ek1));
  b->n_reads = n_reads;
  b->total = 0;
  b->total_ext = 0;
  b->kernel_ms = 0.0;
  b->search_ms = 0.0;
  if (n_reads == 0) return SVDSS_OK;

  int rc;
  const int64_t rec_total = (total_syms >> 3) + 8 * n_reads + 16;
  if ((rc = ensure(b->rec, (size_t)rec_total * sizeof(uint2)))) return rc;
  if ((rc = ensure(b->counts, (size_t)(n_reads + 1) * sizeof(int64_t)))) return rc;
  if ((rc = ensure(b->n_ext, (size_t)n_reads * sizeof(int64_t)))) return rc;
  if ((rc = ensure(b->out_off, (size_t)(n_reads + 1) * sizeof(int64_t)))) return rc;
  if ((rc = ensure(b->misc, 64))) return rc;
  if ((rc = ensure(b->sum, 64))) return rc;

  SfsParams p;
  p.ix = svdss_device_view(ix);
  p.chunks = (const svdss_u4*)d_reads;
  p.max_chunk = total_syms > 0 ? ((total_syms + 15) >> 4) - 1 : 0;
  p.offsets = d_offsets;
  p.n_reads = n_reads;
  p.rec = (uint2*)b->rec.p;
  p.rec_base = nullptr;
  p.rec_cap = nullptr;
  p.counts = (int64_t*)b->counts.p;
  p.n_ext = (int64_t*)b->n_ext.p;
  p.next_read = (unsigned long long*)b->misc.p;
  p.assemble = (flags & SVDSS_SFS_ASSEMBLE) ? 1 : 0;
  unsigned long long* d_overflow = (unsigned long long*)b->misc.p + 1;
  p.n_seg = 1;
  p.seg_shift = 0;
  p.seg_rec = nullptr;
  p.seg_info = nullptr;
  p.read_ids = nullptr;
  p.sub_ids = nullptr;
  p.n_sub = 0;
  p.use_set = 1;
  if (const char* e = getenv("SVDSS_SET")) p.use_set = atoi(e) != 0;
  // (BS needs the suffix array, a k-mer table, and few enough records for their '$' positions to sit in LDS)
  p.use_bs = p.ix.sa != nullptr && p.ix.k > 0 && p.ix.n_dollar > 0 && p.ix.n_dollar <= SV_BS_MAX_DOLLAR;
  // The BS instantiation is chosen by the reference (round 5; rounds 3-4: opt-in).  Measured at GRCh38 lengths with 45 % of
  // the bases in 40 repeat families (profiles/r04h_search_bs.txt, ms per 1,048,576 reads, plain -> BS kernel): copies 1 %
  // apart 408 -> 363, 5 % apart 197 -> 209, 15 % apart 103 -> 108, the headline reference 67.7 either way when the plain
  // kernel is the one launched.  svdss_index::deep_frac (the share of K-mer occurrences in K-mers with SV_BS_MIN or more
  // of them, sampled while the table is built) was 0.001 / 0.206 / 0.329 / 0.371 on the four: from 0.35 on the copies are
  // young enough for the binary search to pay.  Most of a real genome's repeats are old: it keeps the plain kernel.
  // SVDSS_BS=0|1 overrides, SVDSS_BS_DEEP moves the threshold.  Either kernel gives the same SFS and extension counts.
  const double bs_deep = getenv("SVDSS_BS_DEEP") ? atof(getenv("SVDSS_BS_DEEP")) : 0.35;
  bool use_bs_kernel = p.use_bs && ix->deep_frac >= bs_deep;
  if (const char* e = getenv("SVDSS_BS")) if (*e == '0' || *e == '1') use_bs_kernel = p.use_bs && *e == '1';
  p.use_bs = use_bs_kernel;
  b->used_bs = use_bs_kernel ? 1 : 0;
  p.ticket_chunk = 8;
  if (const char* e = getenv("SVDSS_TICKETS")) p.ticket_chunk = atoi(e) > 0 ? atoi(e) : 8;
  p.n_items = n_reads;
  p.n_fallback = (unsigned long long*)b->misc.p + 2;
  p.fallback_ids = nullptr;
  p.epoch = ++b->epoch ? b->epoch : ++b->epoch;

  // (the kernel fetches a read's symbols 64 bytes at a time: a batch smaller than that is searched in a padded copy)
  if (total_syms < 64) {
    if ((rc = ensure(b->tiny, 96))) return rc;
    HIPCHK(hipMemsetAsync(b->tiny.p, 0, 96, stream));
    if (total_syms > 0) HIPCHK(hipMemcpyAsync(b->tiny.p, d_reads, (size_t)total_syms, hipMemcpyDeviceToDevice, stream));
    p.chunks = (const svdss_u4*)b->tiny.p;
    p.max_chunk = 3;
  }
  int max_blocks = 0;
  if ((rc = launch_grid(ix->device, &max_blocks))) return rc;
  // Small batches cannot fill the GPU with one lane per read (256 CUs x 16 waves x 64 lanes):
  // cut every read into n_seg segments searched by separate lanes and stitch the chains.
  int n_seg = 1;
  {
    const int64_t lanes = (int64_t)max_blocks * 256 / 2;   // 4 waves per SIMD resident
    const int64_t want = 4 * lanes / (n_reads > 0 ? n_reads : 1);   // ~4 items per resident lane: short items bound the tail
    n_seg = (int)(want < 4 ? 1 : (want > 8 ? 8 : want));   // large batches fill the GPU with one lane per read; beyond 8
                                                              // the odd unstitchable read costs more than it saves
    // On the rank blocks alone (no text, no suffix array: svdss_index_attach_blocks) a read is thousands of dependent rank steps
    // whatever its segments do, and the reads such a search gets -- smoothed ones -- have too few SFS for their segments to be
    // stitched at (27 % searched again at 30x): from half a lane-load of reads on, one lane per read (`search` at 30x: the two
    // launches 0.44 -> 0.29 s, profiles/r06av_*)
    if ((!ix->d_text || !ix->d_sa) && 2 * n_reads >= lanes) n_seg = 1;
    if (const char* e = getenv("SVDSS_SEGMENTS")) n_seg = atoi(e);
    if (n_seg < 1) n_seg = 1;
    if (n_seg > 16) n_seg = 16;
    while (n_seg & (n_seg - 1)) n_seg &= n_seg - 1;   // power of two: item -> (read, segment) by shift
    if (total_syms / (n_reads > 0 ? n_reads : 1) < 1024) n_seg = 1;
    if (n_reads * (int64_t)n_seg >= ((int64_t)1 << 31)) n_seg = 1;   // item tickets are 32-bit in the kernel
  }
  if (n_seg > 1) {
    const int64_t seg_total = (total_syms >> 3) + (8 + 24 * (int64_t)n_seg) * n_reads + 64;
    if ((rc = ensure(b->seg_rec, (size_t)seg_total * sizeof(uint4)))) return rc;
    if ((rc = ensure(b->seg_info, (size_t)(n_reads * n_seg) * sizeof(SvSegInfo)))) return rc;
    if ((rc = ensure(b->fallback, (size_t)n_reads * sizeof(int64_t)))) return rc;
    if ((rc = ensure(b->seg_take, (size_t)(2 * n_reads * n_seg) * sizeof(int32_t)))) return rc;
    p.seg_rec = (uint4*)b->seg_rec.p;
    p.seg_info = (SvSegInfo*)b->seg_info.p;
    p.fallback_ids = (int64_t*)b->fallback.p;
    p.seg_take = (int32_t*)b->seg_take.p;
  }
  b->n_seg = n_seg;
  b->n_fallback = 0;
  int cap_blocks = max_blocks;
  if (const char* e = getenv("SVDSS_BLOCKS")) cap_blocks = atoi(e) > 0 ? atoi(e) : max_blocks;
  auto blocks_for = [&](int64_t items) {
    const int64_t w = (items + 255) / 256;
    return (int)(w < 1 ? 1 : (w < cap_blocks ? w : cap_blocks));
  };
  const int64_t gw = (n_reads + 3) / 4;  // 4 waves per block, one read per wave iteration
  const int gblocks = (int)(gw < 4 * (int64_t)max_blocks ? (gw > 0 ? gw : 1) : 4 * (int64_t)max_blocks);

  size_t tmp_bytes = 0, tmp2 = 0;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, p.counts, (int64_t*)b->out_off.p,
                                          (int)(n_reads + 1), stream));
  HIPCHK(hipcub::DeviceReduce::Sum(nullptr, tmp2, p.n_ext, (int64_t*)b->sum.p, (int)n_reads, stream));
  if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
  if ((rc = ensure(b->tmp, tmp_bytes))) return rc;

  // heavy reads first (see sfs_order_kernel); SVDSS_ORDER=0 keeps the input order
  const char* ord_env = getenv("SVDSS_ORDER");
  const bool use_order = n_reads >= 1024 && p.ix.k >= 8 && !(ord_env && atoi(ord_env) == 0);
  if (use_order) {
    if ((rc = ensure(b->order, (size_t)n_reads * sizeof(int64_t)))) return rc;
    if ((rc = ensure(b->order_cnt, (size_t)(2 * n_reads + 2) * sizeof(int64_t)))) return rc;   // flags, their scan
    size_t t3 = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, t3, (int64_t*)b->order_cnt.p, (int64_t*)b->order_cnt.p + n_reads + 1,
                                            (int)(n_reads + 1), stream));
    if (t3 > b->tmp.cap && (rc = ensure(b->tmp, t3))) return rc;
  }
  for (int pass = 0; pass < 2; ++pass) {
    HIPCHK(hipMemsetAsync(b->misc.p, 0, 64, stream));
    HIPCHK(hipMemsetAsync((int64_t*)b->counts.p + n_reads, 0, sizeof(int64_t), stream));
    const bool wide = ix->sa_wide;
    const bool seg = n_seg > 1 && pass == 0;   // the exact-capacity rerun is always unsegmented
    HIPCHK(hipEventRecord(b->ev0, stream));
    if (use_order && pass == 0) {
      int64_t* heavy = (int64_t*)b->order_cnt.p;
      int64_t* hscan = heavy + n_reads + 1;
      HIPCHK(hipMemsetAsync(heavy + n_reads, 0, sizeof(int64_t), stream));
      hipLaunchKernelGGL(sfs_order_kernel, dim3((unsigned)((n_reads + 15) / 16)), dim3(256), 0, stream, p, heavy);
      HIPCHK(hipGetLastError());
      size_t t3 = b->tmp.cap;
      HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->tmp.p, t3, heavy, hscan, (int)(n_reads + 1), stream));
      hipLaunchKernelGGL(sfs_order_scatter_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, stream, n_reads, heavy,
                         hscan, (int64_t*)b->order.p);
      HIPCHK(hipGetLastError());
      p.read_ids = (const int64_t*)b->order.p;
    }
    if (seg) {
      p.n_seg = n_seg;
      p.seg_shift = __builtin_ctz((unsigned)n_seg);
      p.n_items = n_reads * n_seg;
      HIPCHK(hipEventRecord(b->ek0, stream));
      // SVDSS_EXTRA_LDS (developer knob): unused dynamic LDS per block, to study the kernel at lower occupancy
      const char* xl = getenv("SVDSS_EXTRA_LDS");
      const size_t extra_lds = xl ? (size_t)atol(xl) : 0;
      if (wide && use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, true, true>), dim3(blocks_for(p.n_items)), dim3(256), extra_lds, stream, p);
      else if (wide) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, true, false>), dim3(blocks_for(p.n_items)), dim3(256), extra_lds, stream, p);
      else if (use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, true, true>), dim3(blocks_for(p.n_items)), dim3(256), extra_lds, stream, p);
      else hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, true, false>), dim3(blocks_for(p.n_items)), dim3(256), extra_lds, stream, p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(b->ek1, stream));
      hipLaunchKernelGGL(sfs_stitch_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, stream, p);
      HIPCHK(hipGetLastError());
      hipLaunchKernelGGL(sfs_assemble_kernel, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, stream, p);
      HIPCHK(hipGetLastError());
      unsigned long long n_fb = 0;
      HIPCHK(hipMemcpyAsync(&n_fb, p.n_fallback, sizeof n_fb, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      b->n_fallback = (int64_t)n_fb;
      if (getenv("SVDSS_DEBUG")) {
        hipEvent_t e2;
        (void)hipEventCreate(&e2);
        (void)hipEventRecord(e2, stream);
        (void)hipEventSynchronize(e2);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, b->ev0, e2);
        fprintf(stderr, "[svdss] segmented search + stitch: %.3f ms, %llu of %lld reads to redo\n", t, n_fb,
                (long long)n_reads);
        sfs_report_op_counts();
        (void)hipEventDestroy(e2);
      }
      // reads whose chains could not be stitched are searched again with a quarter of the segments (their
      // boundaries fall elsewhere), at the end one lane per read
      int lvl_seg = n_seg;
      // (a handful of reads: one lane each at once -- a launch of so few lanes lasts as long as its longest read either
      // way, and an intermediate level that fails again costs that time twice; SVDSS_FALLBACK_DIRECT moves the limit)
      const unsigned long long direct = getenv("SVDSS_FALLBACK_DIRECT") ? (unsigned long long)atoll(getenv("SVDSS_FALLBACK_DIRECT")) : 64ull;
      while (n_fb > 0) {
        lvl_seg = lvl_seg / 4 < 1 || n_fb <= direct ? 1 : lvl_seg / 4;
        if ((rc = ensure(b->fallback2, (size_t)n_fb * sizeof(int64_t)))) return rc;
        HIPCHK(hipMemcpyAsync(b->fallback2.p, p.fallback_ids, (size_t)n_fb * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
        SfsParams q = p;
        q.read_ids = (const int64_t*)b->fallback2.p;
        HIPCHK(hipMemsetAsync(b->misc.p, 0, 8, stream));                    // next_read
        if (lvl_seg == 1) {
          q.n_seg = 1;
          q.n_items = (int64_t)n_fb;
          if (wide && use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, false, true>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
          else if (wide) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, false, false>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
          else if (use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, false, true>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
          else hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, false, false>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
          HIPCHK(hipGetLastError());
          break;
        }
        HIPCHK(hipMemsetAsync(p.n_fallback, 0, 8, stream));
        q.n_seg = lvl_seg;
        q.seg_shift = __builtin_ctz((unsigned)lvl_seg);
        q.n_items = (int64_t)n_fb * lvl_seg;
        q.sub_ids = q.read_ids;
        q.n_sub = (int64_t)n_fb;
        q.epoch = ++b->epoch;
        if (wide && use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, true, true>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
        else if (wide) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, true, false>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
        else if (use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, true, true>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
        else hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, true, false>), dim3(blocks_for(q.n_items)), dim3(256), 0, stream, q);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(sfs_stitch_kernel, dim3((unsigned)((n_fb + 255) / 256)), dim3(256), 0, stream, q);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(sfs_assemble_kernel, dim3((unsigned)((n_fb + 3) / 4)), dim3(256), 0, stream, q);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&n_fb, p.n_fallback, sizeof n_fb, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
      }
    } else {
      p.n_seg = 1;
      p.n_items = n_reads;
      if (pass == 0) HIPCHK(hipEventRecord(b->ek0, stream));
      const char* xl1 = getenv("SVDSS_EXTRA_LDS");   // (developer knob, as above)
      const size_t xlds = xl1 ? (size_t)atol(xl1) : 0;
      if (wide && use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, false, true>), dim3(blocks_for(n_reads)), dim3(256), xlds, stream, p);
      else if (wide) hipLaunchKernelGGL((sfs_search2_kernel<uint64_t, false, false>), dim3(blocks_for(n_reads)), dim3(256), xlds, stream, p);
      else if (use_bs_kernel) hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, false, true>), dim3(blocks_for(n_reads)), dim3(256), xlds, stream, p);
      else hipLaunchKernelGGL((sfs_search2_kernel<uint32_t, false, false>), dim3(blocks_for(n_reads)), dim3(256), xlds, stream, p);
      if (pass == 0) HIPCHK(hipEventRecord(b->ek1, stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->ev1, stream));
    size_t tb = b->tmp.cap;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->tmp.p, tb, p.counts, (int64_t*)b->out_off.p,
                                            (int)(n_reads + 1), stream));
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, (int64_t*)b->out_off.p + n_reads, sizeof(int64_t),
                          hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (p.n_seg == 1 && getenv("SVDSS_DEBUG")) sfs_report_op_counts();
    if ((rc = ensure(b->out_qs, (size_t)(total + 1) * sizeof(int32_t)))) return rc;
    if ((rc = ensure(b->out_len, (size_t)(total + 1) * sizeof(int32_t)))) return rc;
    hipLaunchKernelGGL(sfs_gather_kernel, dim3(gblocks), dim3(256), 0, stream, p,
                       (const int64_t*)b->out_off.p, (int32_t*)b->out_qs.p, (int32_t*)b->out_len.p,
                       d_overflow);
    HIPCHK(hipGetLastError());
    tb = b->tmp.cap;
    HIPCHK(hipcub::DeviceReduce::Sum(b->tmp.p, tb, p.n_ext, (int64_t*)b->sum.p, (int)n_reads, stream));
    unsigned long long n_over = 0;
    int64_t total_ext = 0;
    HIPCHK(hipMemcpyAsync(&n_over, d_overflow, sizeof n_over, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&total_ext, b->sum.p, sizeof total_ext, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->kernel_ms += ms;
    if (pass == 0) {
      float ks = 0.f;
      HIPCHK(hipEventElapsedTime(&ks, b->ek0, b->ek1));
      b->search_ms = ks;
    }
    b->total = total;
    b->total_ext = total_ext;
    if (n_over == 0) break;
    if (pass == 1) return SVDSS_ERANGE;  // cannot happen: capacities were exact
    // Some read produced more SFS than its default region holds (e.g. a read
    // of N against an N-free reference gives one SFS per base).  The counts
    // are exact, so rerun with regions sized from them.
    if ((rc = ensure(b->rec, (size_t)(total + 1) * sizeof(uint2)))) return rc;
    if ((rc = ensure(b->base2, (size_t)(2 * (n_reads + 1)) * sizeof(int64_t)))) return rc;
    int64_t* base2 = (int64_t*)b->base2.p;
    int64_t* cap2 = base2 + (n_reads + 1);
    HIPCHK(hipMemcpyAsync(base2, b->out_off.p, (size_t)(n_reads + 1) * sizeof(int64_t),
                          hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(cap2, b->counts.p, (size_t)(n_reads + 1) * sizeof(int64_t),
                          hipMemcpyDeviceToDevice, stream));
    p.rec = (uint2*)b->rec.p;
    p.rec_base = base2;
    p.rec_cap = cap2;
  }
  return SVDSS_OK;
}

static int batch_for_host_entry(const svdss_index_t* ix, svdss_sfs_batch_t** out, svdss_sfs_batch** bp) {
  HIPCHK(hipSetDevice(ix->device));
  svdss_sfs_batch* b = *out;
  if (!b) {
    b = new (std::nothrow) svdss_sfs_batch();
    if (!b) return SVDSS_ENOMEM;
    b->device = ix->device;
    *out = b;
  }
  if (!b->own_stream) HIPCHK(svdss_make_stream(&b->own_stream, "SVDSS_SEARCH_CUS"));
  *bp = b;
  return SVDSS_OK;
}

extern "C" int svdss_sfs_search_batch(const svdss_index_t* ix, const uint8_t* reads,
                                      const int64_t* offsets, int64_t n_reads, int32_t flags,
                                      svdss_sfs_batch_t** out) {
  if (!ix || !out || n_reads < 0) return SVDSS_EINVAL;
  if (n_reads > 0 && (!reads || !offsets)) return SVDSS_EINVAL;
  if (ix->device < 0 || !ix->d_blocks) return SVDSS_ENODEV;
  svdss_sfs_batch* b = nullptr;
  int rc;
  if ((rc = batch_for_host_entry(ix, out, &b))) return rc;
  if (n_reads == 0) {
    b->n_reads = 0; b->total = 0; b->total_ext = 0; b->kernel_ms = 0.0; b->search_ms = 0.0;
    return SVDSS_OK;
  }
  if (offsets[0] != 0) return SVDSS_EINVAL;
  for (int64_t i = 0; i < n_reads; ++i) {
    const int64_t l = offsets[i + 1] - offsets[i];
    if (l < 0) return SVDSS_EINVAL;
    if (l >= (int64_t)0x7fffffff) return SVDSS_ERANGE;
  }
  const int64_t total = offsets[n_reads];
  const size_t padded = (size_t)((total + 15) & ~(int64_t)15) + 16;
  if ((rc = ensure(b->reads, padded))) return rc;
  if ((rc = ensure(b->offsets, (size_t)(n_reads + 1) * sizeof(int64_t)))) return rc;
  const hipStream_t st = b->own_stream;
  HIPCHK(hipMemsetAsync((uint8_t*)b->reads.p + (padded >= 32 ? padded - 32 : 0), 0, padded >= 32 ? 32 : padded, st));   // the bytes past the last read
  if (total > 0) HIPCHK(hipMemcpyAsync(b->reads.p, reads, (size_t)total, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->offsets.p, offsets, (size_t)(n_reads + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
  return svdss_sfs_search_batch_device(ix, (const uint8_t*)b->reads.p, (const int64_t*)b->offsets.p,
                                       n_reads, total, flags, st, out);
}

// ---- BAM records in: 4-bit bases -> nt6 on the GPU (ping_pong.cpp:90-94: seq_nt16_str, then seq_nt6_table) ----

// read r = blockIdx.y (+ y0), 16 symbols per thread: 8 packed bytes in, 16 nt6 bytes out
__global__ void __launch_bounds__(256) decode_seq4_kernel(const uint8_t* packed, const int64_t* byte_off, const int64_t* sym_off,
                                                          int64_t y0, int64_t n_reads, uint8_t* out) {
  const int64_t r = y0 + blockIdx.y;
  if (r >= n_reads) return;
  const int64_t len = sym_off[r + 1] - sym_off[r];
  const int64_t j0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (j0 >= len) return;
  const uint8_t* src = packed + byte_off[r] + (j0 >> 1);
  uint8_t* dst = out + sym_off[r] + j0;
  const int n = len - j0 < 16 ? (int)(len - j0) : 16;
  // "=ACMGRSVTWYHKDBN": A=1 C=2 G=4 T=8 -> nt6 1..4, every other code (IUPAC, '=') -> 5 like seq_nt6_table
  const uint64_t lut = 0x5555555455535215ull >> 0;   // nibble v -> nt6 at bits 4v (v=1:1, 2:2, 4:3, 8:4, else 5)
  for (int k = 0; k < n; ++k) {
    const uint8_t bb = src[k >> 1];
    const int v = (k & 1) ? (bb & 15) : (bb >> 4);
    dst[k] = (uint8_t)((lut >> (4 * v)) & 15u);
  }
}

extern "C" int svdss_sfs_search_batch_bam(const svdss_index_t* ix, const uint8_t* seq4, const int64_t* byte_off,
                                          const int32_t* l_seq, int64_t n_reads, int32_t flags,
                                          svdss_sfs_batch_t** out) {
  if (!ix || !out || n_reads < 0) return SVDSS_EINVAL;
  if (n_reads > 0 && (!seq4 || !byte_off || !l_seq)) return SVDSS_EINVAL;
  if (ix->device < 0 || !ix->d_blocks) return SVDSS_ENODEV;
  svdss_sfs_batch* b = nullptr;
  int rc;
  if ((rc = batch_for_host_entry(ix, out, &b))) return rc;
  if (n_reads == 0) {
    b->n_reads = 0; b->total = 0; b->total_ext = 0; b->kernel_ms = 0.0; b->search_ms = 0.0;
    return SVDSS_OK;
  }
  std::vector<int64_t> sym_off((size_t)n_reads + 1, 0);
  int64_t max_len = 0;
  for (int64_t i = 0; i < n_reads; ++i) {
    if (l_seq[i] < 0 || byte_off[i + 1] - byte_off[i] < ((int64_t)l_seq[i] + 1) / 2) return SVDSS_EINVAL;
    sym_off[(size_t)i + 1] = sym_off[(size_t)i] + l_seq[i];
    if (l_seq[i] > max_len) max_len = l_seq[i];
  }
  const int64_t total = sym_off[(size_t)n_reads], pbytes = byte_off[n_reads] - byte_off[0];
  const size_t padded = (size_t)((total + 15) & ~(int64_t)15) + 16;
  if ((rc = ensure(b->reads, padded))) return rc;
  if ((rc = ensure(b->offsets, (size_t)(n_reads + 1) * sizeof(int64_t)))) return rc;
  if ((rc = ensure(b->byte_off, (size_t)(n_reads + 1) * sizeof(int64_t)))) return rc;
  if ((rc = ensure(b->packed, (size_t)pbytes + 16))) return rc;
  const hipStream_t st = b->own_stream;
  std::vector<int64_t> rel((size_t)n_reads + 1);
  for (int64_t i = 0; i <= n_reads; ++i) rel[(size_t)i] = byte_off[i] - byte_off[0];
  HIPCHK(hipMemsetAsync((uint8_t*)b->reads.p + (padded >= 32 ? padded - 32 : 0), 0, padded >= 32 ? 32 : padded, st));
  if (pbytes > 0) HIPCHK(hipMemcpyAsync(b->packed.p, seq4 + byte_off[0], (size_t)pbytes, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->byte_off.p, rel.data(), (size_t)(n_reads + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->offsets.p, sym_off.data(), (size_t)(n_reads + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
  if (max_len > 0) {
    const unsigned gx = (unsigned)((max_len + 256 * 16 - 1) / (256 * 16));
    for (int64_t y0 = 0; y0 < n_reads; y0 += 65535) {
      const unsigned gy = (unsigned)std::min<int64_t>(65535, n_reads - y0);
      hipLaunchKernelGGL(decode_seq4_kernel, dim3(gx, gy), dim3(256), 0, st, (const uint8_t*)b->packed.p,
                         (const int64_t*)b->byte_off.p, (const int64_t*)b->offsets.p, y0, n_reads, (uint8_t*)b->reads.p);
    }
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(st));   // rel / sym_off are locals: the copies must have left them
  return svdss_sfs_search_batch_device(ix, (const uint8_t*)b->reads.p, (const int64_t*)b->offsets.p,
                                       n_reads, total, flags, st, out);
}

extern "C" int svdss_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes < 0) return SVDSS_EINVAL;
  *out = nullptr;
  HIPCHK(hipHostMalloc(out, (size_t)(bytes ? bytes : 16), hipHostMallocDefault));
  return SVDSS_OK;
}

extern "C" void svdss_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

extern "C" int svdss_sfs_batch_device_ptrs(const svdss_sfs_batch_t* b, void** counts, void** qs,
                                           void** len, void** n_ext) {
  if (!b) return SVDSS_EINVAL;
  if (counts) *counts = b->counts.p;
  if (qs) *qs = b->out_qs.p;
  if (len) *len = b->out_len.p;
  if (n_ext) *n_ext = b->n_ext.p;
  return SVDSS_OK;
}

extern "C" int svdss_sfs_batch_fetch(const svdss_sfs_batch_t* b, int64_t* counts, int32_t* qs,
                                     int32_t* len, int64_t* n_ext) {
  if (!b) return SVDSS_EINVAL;
  if (b->n_reads == 0) return SVDSS_OK;
  HIPCHK(hipSetDevice(b->device));
  if (b->own_stream) {   // (results of a host-buffer call: stay off the default stream)
    const hipStream_t st = b->own_stream;
    if (counts) HIPCHK(hipMemcpyAsync(counts, b->counts.p, (size_t)b->n_reads * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (n_ext) HIPCHK(hipMemcpyAsync(n_ext, b->n_ext.p, (size_t)b->n_reads * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (qs && b->total > 0) HIPCHK(hipMemcpyAsync(qs, b->out_qs.p, (size_t)b->total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (len && b->total > 0) HIPCHK(hipMemcpyAsync(len, b->out_len.p, (size_t)b->total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SVDSS_OK;
  }
  if (counts)
    HIPCHK(hipMemcpy(counts, b->counts.p, (size_t)b->n_reads * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (n_ext)
    HIPCHK(hipMemcpy(n_ext, b->n_ext.p, (size_t)b->n_reads * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (qs && b->total > 0)
    HIPCHK(hipMemcpy(qs, b->out_qs.p, (size_t)b->total * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (len && b->total > 0)
    HIPCHK(hipMemcpy(len, b->out_len.p, (size_t)b->total * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SVDSS_OK;
}
