// fmd_layout.h -- HBM layout of the FM-index used by the MI355X SFS kernels.
//
// Replaces ropebwt3's rld0 run-length/delta BWT (reached through
// rb3_fmd_set_intv / rb3_fmd_extend at /root/reference/ping_pong.cpp:12,20,30,35)
// with a fixed-stride layout where one rank query == one aligned 64-byte block.
//
// The text is  contig_0 $ revcomp(contig_0) $ contig_1 $ ...  over nt6
// {$=0,A=1,C=2,G=3,T=4,N=5}.  Because that collection is closed under reverse
// complement, count(W) == count(revcomp(W)), so the reference's forward
// extension of W by c has the same interval size as a backward extension of
// revcomp(W) by comp(c).  The kernels therefore only ever do *backward* LF
// steps on a unidirectional FM-index: one symbol's rank at two positions per
// step instead of rb3_fmd_extend's six symbols at two positions.
//
// Block b (64 B, four 16-B quarters) covers BWT[128 b, 128 b + 128):
//   quarter q = { cnt, p0, p1, p2 }   (4 x uint32)
//     cnt : number of symbol (q+1) [A,C,G,T] in BWT[0, 128 b)
//     p0/p1/p2 bit i describe BWT[128 b + 32 q + i]:
//       p2 = 0 : code = (p1<<1|p0) = symbol-1   (A,C,G,T)
//       p2 = 1 : special; p0 = 1 -> N, p0 = 0 -> $   (p1 = 0)
// '$' positions in the BWT (2 per contig) are also kept as a sorted list so
// that rank($,k) / rank(N,k) (reads containing N, Appendix A#6 of SURVEY.md)
// can be answered on a slow path without per-block counters.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SVDSS_HD __host__ __device__ __forceinline__
#else
#define SVDSS_HD inline
#endif

#define SVDSS_BLOCK_SYMS 128
#define SVDSS_BLOCK_SHIFT 7
#define SVDSS_BLOCK_BYTES 64

struct __attribute__((aligned(16))) svdss_u4 { uint32_t x, y, z, w; };  // == uint4 on device

// k-mer table entry (16 B): the state of a phase of ping_pong_search after its
// first K symbols, for every ACGT K-mer W (key = sum W[i] << 2i, text order).
//   info >> 62 == SVDSS_TAB_EMPTY  : W absent; info & 0xff = d = number of symbols
//                                    (taken from the END of W) that still occur; (info >> 8) & 0xff = df =
//                                    how many leading symbols of the d + 1 that do not occur together still
//                                    do (outcome of the forward phase that follows, 0 = not recorded)
//   info >> 62 == SVDSS_TAB_UNIQUE : one occurrence; lo = SA index, info bits 0-39 = text position,
//                                    bits 40-57 = its extension symbols
//   info >> 62 == SVDSS_TAB_FEW    : 2-4 occurrences (SA rows lo .. lo+size-1); lo bits 0-35 = SA index,
//                                    info bits 59-61 = size, extension symbols of occurrences 0, 1, 2 in info bits
//                                    0-17, 18-35, 36-53, of occurrence 3 in lo bits 36-53
//   info >> 62 == SVDSS_TAB_MULTI  : lo = SA index of the interval, info & MASK = size (>= 5)
// Extension symbols of an occurrence: the SVDSS_TAB_EXT text symbols in front of it (nt6, 3 bits each, the
// nearest one in the low bits; '$' from the record start on) -- what the next SVDSS_TAB_EXT backward extensions
// (or, for the reverse-complement key of a forward phase, forward extensions) of the K-mer must match.  A chance
// match of a K-mer that contains a sequencing error dies within a symbol or two: with these the lane sees
// that in the entry it already holds instead of fetching a BWT block or a text window per symbol.
struct __attribute__((aligned(16))) SvdssTabEntry { uint64_t lo, info; };
#define SVDSS_TAB_EMPTY 0ull
#define SVDSS_TAB_UNIQUE 1ull
#define SVDSS_TAB_MULTI 2ull
#define SVDSS_TAB_FEW 3ull
#define SVDSS_TAB_MASK ((1ull << 62) - 1)
#define SVDSS_TAB_EXT 6
#define SVDSS_TAB_POS_MASK ((1ull << 40) - 1)
#define SVDSS_TAB_LO_MASK ((1ull << 36) - 1)

struct SvdssDevIndex {
  const svdss_u4* blocks;   // 4 quarters per block, (n/128 + 1) blocks
  const int64_t* dollar;    // sorted BWT positions holding '$'
  int64_t n;                // BWT length
  int32_t n_dollar;
  int32_t k;                // K of the k-mer table (0: no table)
  int32_t bs_after;         // BS takes a deep backward phase over when more than this many rank steps are still expected (sfs_core2.h)
  int32_t pad_;
  int64_t acc[7];           // acc[c] = #symbols < c
  const uint8_t* text;      // nt6 text; text[-64 .. n+64) is readable ('$' padding)
  const void* sa;           // suffix array: uint32[n] (n < 2^32) or uint64[n]
  const SvdssTabEntry* table;  // 4^k entries or nullptr
};

// acc[c] through a select chain: a dynamically indexed kernel-argument array
// would be spilled to scratch memory and cost a memory access per step.
SVDSS_HD int64_t svdss_acc(const SvdssDevIndex& ix, int c) {
  return c <= 0 ? ix.acc[0] : c == 1 ? ix.acc[1] : c == 2 ? ix.acc[2] : c == 3 ? ix.acc[3]
       : c == 4 ? ix.acc[4] : c == 5 ? ix.acc[5] : ix.acc[6];
}

SVDSS_HD int svdss_comp(int a) { return (a >= 1 && a <= 4) ? 5 - a : a; }  // ping_pong.hpp:38

SVDSS_HD int svdss_popc(uint32_t x) { return __builtin_popcount(x); }
SVDSS_HD float svdss_log2f(float x) { return __builtin_log2f(x); }

// mask with the low r bits set, r clamped to [0,32]
SVDSS_HD uint32_t svdss_lowmask(int r) {
  return r >= 32 ? 0xffffffffu : (r <= 0 ? 0u : ((1u << r) - 1u));
}

// number of '$' in BWT[0,k)
SVDSS_HD int64_t svdss_rank_dollar(const SvdssDevIndex& ix, int64_t k) {
  int lo = 0, hi = ix.n_dollar;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (ix.dollar[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// rank of symbol c in BWT[0,k) from one block already in registers.
// c in 1..4 is the fast path; c in {0,5} is the slow path.
SVDSS_HD int64_t svdss_rank_in_block(const SvdssDevIndex& ix, const svdss_u4 q[4], int c, int64_t k) {
  const int r = (int)(k & (SVDSS_BLOCK_SYMS - 1));
  if (c >= 1 && c <= 4) {
    const uint32_t code = (uint32_t)(c - 1);
    const uint32_t m0 = (code & 1u) ? 0u : 0xffffffffu;  // xor mask: match where p0 == bit0
    const uint32_t m1 = (code & 2u) ? 0u : 0xffffffffu;
    uint32_t cnt = code == 0 ? q[0].x : code == 1 ? q[1].x : code == 2 ? q[2].x : q[3].x;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t match = (q[j].y ^ m0) & (q[j].z ^ m1) & ~q[j].w;
      s += svdss_popc(match & svdss_lowmask(r - 32 * j));
    }
    return (int64_t)cnt + s;
  }
  // slow path: '$' from the sorted list, N = k - $ - (A+C+G+T)
  int64_t nd = svdss_rank_dollar(ix, k);
  if (c == 0) return nd;
  int64_t acgt = (int64_t)q[0].x + q[1].x + q[2].x + q[3].x;
#pragma unroll
  for (int j = 0; j < 4; ++j) acgt += svdss_popc(~q[j].w & svdss_lowmask(r - 32 * j));
  return k - nd - acgt;
}

// BWT symbol at row i, from the block layout
SVDSS_HD int svdss_bwt_at(const SvdssDevIndex& ix, int64_t i) {
  const svdss_u4 q = ix.blocks[4 * (i >> SVDSS_BLOCK_SHIFT) + ((i >> 5) & 3)];
  const int bit = (int)(i & 31);
  const uint32_t p0 = (q.y >> bit) & 1u, p1 = (q.z >> bit) & 1u, p2 = (q.w >> bit) & 1u;
  return p2 ? (p0 ? 5 : 0) : (int)(1u + (p1 << 1 | p0));
}

// extension symbols (see SvdssTabEntry) of the suffix at SA row `row`: BWT[row], BWT[LF(row)], ...
SVDSS_HD uint32_t svdss_ext_symbols(const SvdssDevIndex& ix, int64_t row) {
  uint32_t x = 0;
  for (int e = 0; e < SVDSS_TAB_EXT; ++e) {
    const int sym = svdss_bwt_at(ix, row);
    if (sym == 0) break;   // start of the record: '$' from here on
    x |= (uint32_t)sym << (3 * e);
    row = svdss_acc(ix, sym) + svdss_rank_in_block(ix, ix.blocks + 4 * (row >> SVDSS_BLOCK_SHIFT), sym, row);
  }
  return x;
}

// extension symbols of occurrence j of a UNIQUE / FEW entry
SVDSS_HD uint32_t svdss_tab_ext(uint64_t lo, uint64_t info, int j) {
  if ((info >> 62) == SVDSS_TAB_UNIQUE) return (uint32_t)(info >> 40) & 0x3ffffu;
  return j < 3 ? (uint32_t)(info >> (18 * j)) & 0x3ffffu : (uint32_t)(lo >> 36) & 0x3ffffu;
}
