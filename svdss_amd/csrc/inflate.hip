// inflate.hip -- BGZF blocks inflated on the GPU (svdss_bgzf_inflate).
//
// Stands where htslib's bgzf_read / inflate stand under sam_read1 (/root/reference/ping_pong.cpp:58,247-249;
// clusterer.cpp:101; smoother.cpp:262): `SVDSS search` reads ~1.5 bytes of BAM per base, all of it deflate streams of
// at most 64 KB, and inflating them is what bounds the binary end to end (a host core inflates ~0.3 GB/s of this kind of
// data; a 30x human sample is ~140 GB inflated).  Every BGZF block is an independent stream, so the GPU takes one
// wavefront per block and a few thousand blocks at a time:
//   * the most recent 4 KB of output live in an LDS ring: literals and near matches never touch HBM, the ring is written
//     out in aligned dwords 512 bytes at a time; a match that reaches further back (deflate allows 32 KB) reads the bytes
//     the block itself wrote to HBM earlier -- the ring is that small so that a CU holds 18 blocks instead of 4, and
//     the latency of those reads (and of everything else) is hidden by the other wavefronts;
//   * the compressed bytes pass through a 1 KB LDS ring, refilled 512 bytes at a time by the whole wave;
//   * Huffman tables (10-bit literal/length, 8-bit distance, 16-bit entries; longer codes are decoded canonically) are
//     built by the wave in parallel: the canonical code of a symbol is the rank of the symbol among those of its length
//     (wave ballots), and every lane fills the table entries of its own symbols;
//   * symbols are decoded in rounds of 64 bits: every lane decodes the whole symbol that would start at its bit offset
//     (literal, or length + extra bits + distance code + extra bits, or end of block) from the 64 bits that start there;
//     the scalar unit follows the chain of symbol starts with lane reads, a wave scan places the outputs, literals are
//     stored together, matches are copied in order by all 64 lanes.
// 8.7 KB of LDS per block.  No CRC check on this path (the caller checks the BGZF footers on the host).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/svdss_hip.h"
#include "inflate_dev.h"

#define UNI(x) __builtin_amdgcn_readfirstlane(x)

namespace {

// the LDS ring holds the most recent output only; a match that reaches further back than NEAR reads the bytes the block
// itself wrote to HBM earlier (they left the ring at least FLUSH + one round ago)
#ifndef INF_WIN
#define INF_WIN 4096
#endif
#ifndef INF_FRAC
#define INF_FRAC 8
#endif
constexpr int WIN = INF_WIN, WM = WIN - 1;
constexpr int FLUSH = WIN / INF_FRAC;        // the ring is written out whenever this many bytes are waiting
constexpr int ROUND_MAX = WIN / INF_FRAC;    // a round stops taking symbols once it has produced this many bytes (+ one match)
// sources within this distance of a match's output position are read from the ring: everything further back has been
// written out (FLUSH + ROUND_MAX + 258 < NEAR) and nothing closer has been overwritten (NEAR + ROUND_MAX + 258 < WIN)
constexpr int NEAR = WIN / 2 - WIN / INF_FRAC;
static_assert(FLUSH + ROUND_MAX + 258 + 64 < NEAR && NEAR + ROUND_MAX + 258 + 64 < WIN, "ring too small");
#ifndef INF_LB
#define INF_LB 10
#endif
#ifndef INF_DB
#define INF_DB 8
#endif
constexpr int LB = INF_LB, DB = INF_DB, CB = 7;   // bits of the direct tables (literal / length, distance, code lengths)
constexpr int INB = 1024, HALF = INB / 2;   // input ring, refilled a half at a time


struct Lds {
  uint32_t win[WIN / 4];
  uint16_t lit[1 << LB];
  uint16_t dst[1 << DB];
  uint32_t inb[INB / 4];
  uint8_t lens[320];
  uint16_t lsym[288];
  uint16_t dsym[32];
  uint16_t lcnt[16], dcnt[16];
};

#ifdef INF_COUNT
__device__ unsigned long long g_inf_cnt[8];   // rounds, literals, matches, one-symbol steps, deflate blocks, copy bytes, overlap copies
#define CNT(k, v) (cnt[k] += (v))
#else
#define CNT(k, v)
#endif

// status codes (per block)
enum { ST_OK = 0, ST_BTYPE = 1, ST_STORED = 2, ST_LENS = 3, ST_CODE = 4, ST_DIST = 5, ST_OUT = 6, ST_IN = 7, ST_SIZE = 8 };

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

template <int CTRL, int RMASK>
__device__ __forceinline__ int dpp_add(int x) { return x + __builtin_amdgcn_update_dpp(0, x, CTRL, RMASK, 0xf, false); }
// inclusive sum over the 64 lanes: four row_shr steps, then row_bcast15 / row_bcast31
__device__ __forceinline__ int wave_scan_add(int x) {
  x = dpp_add<0x111, 0xf>(x);
  x = dpp_add<0x112, 0xf>(x);
  x = dpp_add<0x114, 0xf>(x);
  x = dpp_add<0x118, 0xf>(x);
  x = dpp_add<0x142, 0xa>(x);
  x = dpp_add<0x143, 0xc>(x);
  return x;
}

// Canonical Huffman code of n symbols with lengths lens[0..n): table of 2^tb 16-bit entries (symbol << 4 | length, 0 =
// a longer code or none), per-length counts and the symbols sorted by (length, symbol) for the bit-by-bit decoder.
// Returns false if the lengths over-subscribe the code space.
__device__ bool build_table(const uint8_t* lens, int n, uint16_t* tab, int tb, uint16_t* cnt, uint16_t* sym, int lane,
                            int n_flagged = 0) {   // entries of the symbols below n_flagged carry bit 15 (literals)
  for (int k = lane; k < (1 << tb); k += 64) tab[k] = 0;
  int count[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) count[l] = 0;
  for (int s0 = 0; s0 < n; s0 += 64) {
    const int s = s0 + lane;
    const int ml = s < n ? (int)lens[s] : 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) count[l] += (int)__popcll(__ballot(ml == l));
  }
  int offs[16], code0[16];
  int left = 1, code = 0, o = 0;
  offs[0] = 0; code0[0] = 0;
  bool over = false;
#pragma unroll
  for (int l = 1; l < 16; ++l) {
    left = (left << 1) - count[l];
    if (left < 0) over = true;
    code = (code + (l > 1 ? count[l - 1] : 0)) << 1;
    code0[l] = code;
    offs[l] = o;
    o += count[l];
  }
  if (over) return false;
  if (lane < 16) {
    int c = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) c = lane == l ? count[l] : c;
    cnt[lane] = (uint16_t)c;
  }
  int run[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) run[l] = 0;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  for (int s0 = 0; s0 < n; s0 += 64) {
    const int s = s0 + lane;
    const int ml = s < n ? (int)lens[s] : 0;
    int rank = 0, first = 0, base = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
      const unsigned long long m = __ballot(ml == l);
      if (ml == l) { rank = run[l] + (int)__popcll(m & lt); first = code0[l]; base = offs[l]; }
      run[l] += (int)__popcll(m);
    }
    if (ml) {
      sym[base + rank] = (uint16_t)s;
      if (ml <= tb) {
        const uint32_t rev = __brev((uint32_t)(first + rank)) >> (32 - ml);
        const uint16_t e = (uint16_t)((s << 4) | ml | (s < n_flagged ? 0x8000 : 0));
        for (uint32_t k = rev; k < (1u << tb); k += (1u << ml)) tab[k] = e;
      }
    }
  }
  return true;
}

__global__ void __launch_bounds__(64) bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const svdss_bgzf_block_t* __restrict__ blks,
                                                         uint8_t* out, int32_t* __restrict__ status) {
  __shared__ Lds L;
  const int lane = threadIdx.x;
  const svdss_bgzf_block_t B = blks[blockIdx.x];
  const uint32_t isize = (uint32_t)B.isize;
  const uint64_t in_first = (uint64_t)B.coff, in_end = in_first + (uint64_t)(uint32_t)B.clen;
  uint8_t* const o8 = out + B.uoff;
  uint8_t* const winb = (uint8_t*)L.win;
  if (isize == 0) { if (lane == 0) status[blockIdx.x] = ST_OK; return; }
#ifdef INF_COUNT
  unsigned cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#define FAIL(code) do { status[blockIdx.x] = (code); return; } while (0)

  // ---- input: absolute offsets into comp; the ring holds [base, base + INB)
  uint64_t base = in_first & ~(uint64_t)(HALF - 1), in_addr = in_first;
  auto load_half = [&](uint64_t a0) {   // HALF bytes, 8 per lane
    static_assert(HALF == 64 * 8, "one uint2 per lane");
    const uint2 v = *(const uint2*)(comp + a0 + (uint64_t)lane * 8);
    *(uint2*)((uint8_t*)L.inb + ((a0 + (uint64_t)lane * 8) & (INB - 1))) = v;
  };
  load_half(base);
  load_half(base + HALF);
  uint64_t bb = 0;
  int bc = 0;
  auto step_half = [&]() {
    // (32 bytes late: the bit buffer holds up to 8 bytes that were read before in_addr, and the literal runs read the
    // ring at the position of the first unused bit)
    if (in_addr - base >= HALF + 32) { load_half(base + INB); base += HALF; }
  };
  for (; (in_addr & 3) != 0; ++in_addr) {
    const uint32_t b = UNI((uint32_t)((const uint8_t*)L.inb)[in_addr & (INB - 1)]);
    bb |= (uint64_t)b << bc;
    bc += 8;
  }
  step_half();
  auto refill = [&]() {   // at least 33 bits afterwards
    if (bc <= 32) {
      const uint32_t w = UNI(L.inb[(in_addr & (INB - 1)) >> 2]);
      bb |= (uint64_t)w << bc;
      bc += 32;
      in_addr += 4;
      step_half();
    }
  };
  auto take = [&](int n) { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; };
  auto decode = [&](const uint16_t* tab, int tb, const uint16_t* cnt, const uint16_t* sym) -> int {
    refill();
    const uint32_t e = UNI((uint32_t)tab[bb & ((1u << tb) - 1)]);
    const int len = (int)(e & 15u);
    if (len) { bb >>= len; bc -= len; return (int)((e & 0x7fffu) >> 4); }
    int code = 0, first = 0, index = 0;
    uint64_t b = bb;
    for (int l = 1; l <= 15; ++l) {
      code |= (int)(b & 1);
      b >>= 1;
      const int count = UNI((int)cnt[l]);
      if (code - count < first) { bb >>= l; bc -= l; return UNI((int)sym[index + (code - first)]); }
      index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
  };

  // ---- output: the ring holds the last 32 KB; [flushed, wpos) is not in HBM yet
  uint32_t wpos = 0, flushed = 0;
  auto flush = [&](bool final) {
    const uint32_t upto = wpos;
    // head: single bytes until the HBM address is dword-aligned
    uint32_t head = (uint32_t)((4 - ((B.uoff + flushed) & 3)) & 3);
    if (head > upto - flushed) head = upto - flushed;
    if ((uint32_t)lane < head) o8[flushed + lane] = winb[(flushed + lane) & WM];
    flushed += head;
    const uint32_t nd = (upto - flushed) >> 2;
    uint32_t* const o32 = (uint32_t*)(o8 + flushed);
    for (uint32_t i = lane; i < nd; i += 64) {
      const uint32_t r = (flushed + 4 * i) & WM;
      const uint32_t w0 = L.win[r >> 2], w1 = L.win[((r >> 2) + 1) & (WIN / 4 - 1)];
      o32[i] = __builtin_amdgcn_alignbyte(w1, w0, r & 3);
    }
    flushed += 4 * nd;
    if (final) {
      const uint32_t tail = upto - flushed;
      if ((uint32_t)lane < tail) o8[flushed + lane] = winb[(flushed + lane) & WM];
      flushed += tail;
    }
  };

  // a match of ml bytes at output position o, dd bytes back.  Sources closer than NEAR are in the ring; older ones were
  // written to HBM by an earlier flush (complete 128-byte lines by now: nothing of this block's output is read from HBM
  // within FLUSH + ROUND_MAX + 258 of its end) and are read back from there -- with the ring this small a CU holds a dozen
  // blocks, and that latency is what the other wavefronts are for.
  auto copy_match = [&](uint32_t o, uint32_t ml, uint32_t dd) {
    const uint32_t from = o - dd;
    if (dd <= (uint32_t)NEAR && dd >= ml && ml <= 64) {   // the common one: short, from the ring, not overlapping itself
      if ((uint32_t)lane < ml) winb[(o + (uint32_t)lane) & WM] = winb[(from + (uint32_t)lane) & WM];
      return;
    }
    const bool far = dd > (uint32_t)NEAR;
    if (far) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this block's own stores of long ago)
    uint8_t v[5];
    int nk = 0;
    for (uint32_t k = lane; k < ml; k += 64) {
      const uint32_t sp = from + (dd >= ml ? k : k % dd);   // (an overlapping match repeats its first dd bytes)
      v[nk++] = (far && o - sp > (uint32_t)NEAR) ? __hip_atomic_load(o8 + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : winb[sp & WM];
    }
    nk = 0;
    for (uint32_t k = lane; k < ml; k += 64) winb[(o + k) & WM] = v[nk++];
  };

  for (;;) {
    refill();
    const uint32_t bfinal = take(1), btype = take(2);
    if (btype == 0) {
      // stored: to the byte boundary, LEN / NLEN, LEN bytes straight from the input
      take(bc & 7);
      refill();
      const uint32_t len = take(16), nlen = take(16);
      if ((len ^ 0xffffu) != nlen) FAIL(ST_STORED);
      if (wpos + len > isize) FAIL(ST_OUT);
      // the whole bytes still in the bit buffer are the first of them
      const uint64_t src = in_addr - (uint64_t)(bc >> 3);
      if (src + len > in_end) FAIL(ST_IN);
      // (through the ring in pieces of at most FLUSH bytes: less than that is waiting to be written out at any time)
      for (uint32_t done = 0; done < len;) {
        const uint32_t n = len - done < (uint32_t)FLUSH ? len - done : (uint32_t)FLUSH;
        for (uint32_t i = lane; i < n; i += 64) winb[(wpos + i) & WM] = comp[src + done + i];
        wpos += n;
        done += n;
        if (wpos - flushed >= (uint32_t)FLUSH) flush(false);
      }
      // restart the reader behind the stored bytes
      in_addr = src + len;
      base = in_addr & ~(uint64_t)(HALF - 1);
      load_half(base);
      load_half(base + HALF);
      bb = 0; bc = 0;
      for (; (in_addr & 3) != 0; ++in_addr) {
        const uint32_t b = UNI((uint32_t)((const uint8_t*)L.inb)[in_addr & (INB - 1)]);
        bb |= (uint64_t)b << bc;
        bc += 8;
      }
      step_half();
    } else if (btype == 1 || btype == 2) {
      int nlen, ndist;
      if (btype == 1) {
        for (int s = lane; s < 320; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
        nlen = 288; ndist = 30;
      } else {
        nlen = (int)take(5) + 257;
        ndist = (int)take(5) + 1;
        const int ncode = (int)take(4) + 4;
        if (nlen > 286 || ndist > 30) FAIL(ST_LENS);
        if (lane < 19) L.lens[lane] = 0;
        for (int i = 0; i < ncode; ++i) {
          refill();
          const uint32_t v = take(3);
          // (the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, five bits each)
          const int ord = i < 12 ? (int)((0x022caa324e804a30ull >> (5 * i)) & 31) : (int)((0x3c2e1346cull >> (5 * (i - 12))) & 31);
          if (lane == 0) L.lens[ord] = (uint8_t)v;
        }
        if (!build_table(L.lens, 19, L.dst, CB, L.dcnt, L.dsym, lane)) FAIL(ST_LENS);
        // the code lengths of the literal/length and distance codes, run-length coded (all lanes keep the same copy)
        uint8_t* const tmp = (uint8_t*)L.lsym;   // (free until the tables are built)
        int i = 0, prev = 0;
        while (i < nlen + ndist) {
          const int s = decode(L.dst, CB, L.dcnt, L.dsym);
          if (s < 0) FAIL(ST_LENS);
          if (s < 16) { if (lane == 0) tmp[i] = (uint8_t)s; prev = s; ++i; continue; }
          refill();
          int rep, val = 0;
          if (s == 16) { if (i == 0) FAIL(ST_LENS); rep = 3 + (int)take(2); val = prev; }
          else if (s == 17) { rep = 3 + (int)take(3); prev = 0; }
          else { rep = 11 + (int)take(7); prev = 0; }
          if (i + rep > nlen + ndist) FAIL(ST_LENS);
          for (int k = lane; k < rep; k += 64) tmp[i + k] = (uint8_t)val;
          i += rep;
        }
        for (int s = lane; s < 320; s += 64) {
          const int v = s < nlen ? tmp[s] : (s >= 288 && s - 288 < ndist) ? tmp[nlen + (s - 288)] : 0;
          // (read everything before anything is overwritten: lens and tmp are different arrays)
          L.lens[s] = (uint8_t)v;
        }
        if (UNI((int)L.lens[256]) == 0) FAIL(ST_LENS);
        nlen = 288; ndist = 30;
      }
      CNT(4, 1);
      if (!build_table(L.lens, nlen, L.lit, LB, L.lcnt, L.lsym, lane, 256)) FAIL(ST_LENS);
      if (!build_table(L.lens + 288, ndist, L.dst, DB, L.dcnt, L.dsym, lane)) FAIL(ST_LENS);
      // ---- symbols, a round of up to 64 bits at a time.  Lane l fetches the 32 bits that start l bits ahead (straight
      // from the input ring) and looks them up in both tables; the scalar unit then follows the chain from offset 0 --
      // a code's length names the lane that holds what comes next: the next code, a length's extra bits, the distance
      // code, its extra bits -- with lane reads only, no memory in the chain.  Literals are stored together by the lanes
      // they start at; a match is copied by all lanes as soon as the literals before it are in the window.  A code
      // longer than the table's index bits takes the one-symbol path (bit buffer, canonical decoding).  The cursor of
      // the rounds is the bit position P alone; errors are collected and reported after the loop.
      uint32_t err = 0;
      bool eob = false;
      uint64_t P = in_addr * 8 - (uint64_t)bc;
      for (;;) {
        if ((P >> 3) - base >= HALF + 32) { load_half(base + INB); base += HALF; }
        CNT(0, 1);
        // ---- every lane decodes the symbol that would start at its bit offset, whole: a literal, or a match with its
        // extra bits and its distance (all inside the 64 bits that start there), or the end-of-block code -- and where
        // the next symbol starts
        const uint32_t pl = (uint32_t)P + (uint32_t)lane;   // (the ring is indexed modulo 2^14 bits)
        const uint32_t di = pl >> 5;
        const uint32_t w0 = L.inb[di & (INB / 4 - 1)], w1 = L.inb[(di + 1) & (INB / 4 - 1)], w2 = L.inb[(di + 2) & (INB / 4 - 1)];
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, pl & 31u), hi = __builtin_amdgcn_alignbit(w2, w1, pl & 31u);
        const uint64_t b64 = ((uint64_t)hi << 32) | lo;
        const uint32_t E = (uint32_t)L.lit[lo & ((1u << LB) - 1)];
        const uint32_t len = E & 15u, sym = (E & 0x7fffu) >> 4;
        const bool is_lit = (E & 0x8000u) != 0;
        const uint32_t ls = umin(sym - 257u, 28u);   // (lanes that hold no length code compute on a harmless value)
        const uint32_t lx = ls < 8 || ls == 28 ? 0 : (ls >> 2) - 1;
        const uint32_t lb = ls < 8 ? 3u + ls : ls == 28 ? 258u : 3u + ((4u + (ls & 3u)) << lx);
        const uint32_t mlen = lb + ((lo >> len) & ((1u << lx) - 1));
        const uint32_t doff = len + lx;
        const uint32_t D = (uint32_t)L.dst[(uint32_t)(b64 >> doff) & ((1u << DB) - 1)];
        const uint32_t dl = D & 15u, ds = umin(D >> 4, 29u);
        const uint32_t dx = ds < 4 ? 0 : (ds >> 1) - 1;
        const uint32_t db = ds < 4 ? 1u + ds : 1u + ((2u + (ds & 1u)) << dx);
        const uint32_t eoff = doff + dl;
        const uint32_t dist = db + ((uint32_t)(b64 >> eoff) & ((1u << dx) - 1));
        // kind: 0 literal, 1 match, 2 end of block, 3 = a code longer than the table's index bits or an invalid one (the
        // one-symbol path decodes it, or reports it)
        const bool is_match = !is_lit && sym >= 257u;
        uint32_t kind = is_lit ? 0u : sym == 256u ? 2u : 1u;
        if (len == 0 || (is_match && (sym > 285u || dl == 0 || (D >> 4) > 29u))) kind = 3u;
        const uint32_t nbits = kind == 1u ? eoff + dx : len;
        const uint32_t outlen = kind == 0u ? 1u : kind == 1u ? mlen : 0u;
        // ---- the chain of symbol starts from offset 0, by pointer doubling on the lanes: lane l knows where the symbol
        // after its own starts (J) and which offsets the chain from l visits (M); six rounds of "append the chain of
        // the lane I point at" close M over the 64 offsets (a symbol is at least one bit long), and the chain of the
        // round is M of lane 0.  A symbol the tables cannot decode (kind 3) is on nobody's chain and ends the chains
        // that reach it; so do the end-of-block code and a symbol that ends beyond the 64 offsets -- after being
        // visited.  (One scalar instruction per CU and cycle: followed symbol by symbol on the scalar unit, ~14
        // instructions each, the chain was what bound the kernel.)
        const uint32_t nextl = (uint32_t)lane + nbits;
        uint32_t J = (kind == 3u || kind == 2u || nextl >= 64u) ? 64u : nextl;
        uint32_t Mlo = kind == 3u ? 0u : (lane < 32 ? 1u << lane : 0u), Mhi = kind == 3u ? 0u : (lane >= 32 ? 1u << (lane - 32) : 0u);
        while (__ballot(J < 64u)) {
          const int a = (int)(J << 2);
          const uint32_t jj = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)J);
          const uint32_t ml_ = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)Mlo), mh_ = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)Mhi);
          if (J < 64u) { Mlo |= ml_; Mhi |= mh_; J = jj; }
        }
        const unsigned long long chain0 = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)Mlo) |
                                          ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)Mhi) << 32);
        // ---- where every symbol of the chain puts its output: a scan over the chain's lanes; the round takes the
        // symbols that start before ROUND_MAX bytes of it
        const bool onc0 = ((chain0 >> lane) & 1ull) != 0;
        const uint32_t x0 = onc0 ? outlen : 0u;
        const uint32_t incl = (uint32_t)wave_scan_add((int)x0);
        const bool onc = onc0 && incl - x0 < (uint32_t)ROUND_MAX;
        const unsigned long long chain = __ballot(onc);
        uint32_t off = 0, total = 0;
        bool slow = false;
        if (chain) {
          const int h = 63 - (int)__builtin_clzll(chain);          // the last symbol taken
          total = __builtin_amdgcn_readlane(incl, h);
          off = __builtin_amdgcn_readlane(nextl, h);
          if (__builtin_amdgcn_readlane(kind, h) == 2u) eob = true;
          else if (off < 64u && total < (uint32_t)ROUND_MAX) slow = __builtin_amdgcn_readlane(kind, off) == 3u;
        } else slow = true;                                          // (the symbol at offset 0 itself)
        const uint32_t opos = wpos + (incl - x0);
        if (wpos + total > isize) { err = ST_OUT; break; }
        if (__ballot(onc && kind == 1u && dist > opos)) { err = ST_DIST; break; }
        if (onc && kind == 0u) winb[opos & WM] = (uint8_t)sym;
        // ---- the matches, in order
        unsigned long long mm = __ballot(onc && kind == 1u);
        while (mm) {
          const int h = (int)__builtin_ctzll(mm);
          mm &= mm - 1;
          CNT(2, 1);
          const uint32_t o = __builtin_amdgcn_readlane(opos, h), ml = __builtin_amdgcn_readlane(mlen, h), dd = __builtin_amdgcn_readlane(dist, h);
          copy_match(o, ml, dd);
        }
        wpos += total;
        P += (uint64_t)off;
        if ((P >> 3) > in_end + 16) err = ST_IN;
        if (err || eob) break;
        if (wpos - flushed >= (uint32_t)FLUSH) flush(false);
        if (!slow && off != 0) continue;
        // ---- one symbol through the bit buffer
        CNT(3, 1);
        {
          in_addr = (P >> 5) << 2;
          const uint32_t v0 = UNI(L.inb[(in_addr & (INB - 1)) >> 2]), v1 = UNI(L.inb[((in_addr + 4) & (INB - 1)) >> 2]);
          const int sh = (int)(P & 31);
          bb = (((uint64_t)v1 << 32) | v0) >> sh;
          bc = 64 - sh;
          in_addr += 8;
          step_half();
        }
        const int s = decode(L.lit, LB, L.lcnt, L.lsym);
        if (s < 0) { err = ST_CODE; break; }
        if (s < 256) {
          if (wpos >= isize) { err = ST_OUT; break; }
          if (lane == 0) winb[wpos & WM] = (uint8_t)s;
          ++wpos;
        } else if (s == 256) {
          eob = true;
        } else {
          if (s > 285) { err = ST_CODE; break; }
          refill();
          const int ls = s - 257;
          const int lx = ls < 8 || ls == 28 ? 0 : (ls >> 2) - 1;
          const uint32_t lb = ls < 8 ? 3u + (uint32_t)ls : ls == 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3u)) << lx);
          const uint32_t len = lb + take(lx);
          const int ds = decode(L.dst, DB, L.dcnt, L.dsym);
          if (ds < 0 || ds > 29) { err = ST_DIST; break; }
          refill();
          const int dx = ds < 4 ? 0 : (ds >> 1) - 1;
          const uint32_t db = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << dx);
          const uint32_t dist = db + take(dx);
          if (dist > wpos) { err = ST_DIST; break; }
          if (wpos + len > isize) { err = ST_OUT; break; }
          copy_match(wpos, len, dist);
          wpos += len;
        }
        P = in_addr * 8 - (uint64_t)bc;
        if (eob) break;
      }
      if (err) FAIL((int)err);
      {   // the bit buffer again, behind the end-of-block code
        in_addr = (P >> 5) << 2;
        step_half();
        const uint32_t v0 = UNI(L.inb[(in_addr & (INB - 1)) >> 2]), v1 = UNI(L.inb[((in_addr + 4) & (INB - 1)) >> 2]);
        const int sh = (int)(P & 31);
        bb = (((uint64_t)v1 << 32) | v0) >> sh;
        bc = 64 - sh;
        in_addr += 8;
        step_half();
      }
      if (wpos - flushed >= (uint32_t)FLUSH) flush(false);
    } else {
      FAIL(ST_BTYPE);
    }
    if (bfinal) break;
  }
  if (wpos != isize) FAIL(ST_SIZE);
  flush(true);
  if (lane == 0) status[blockIdx.x] = ST_OK;
#ifdef INF_COUNT
  if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_inf_cnt[k], (unsigned long long)cnt[k]);
#endif
#undef FAIL
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

hipError_t svdss_inflate_enqueue(hipStream_t st, const uint8_t* d_comp, const svdss_bgzf_block_t* d_blocks, int64_t n_blocks,
                                 uint8_t* d_out, int32_t* d_status) {
  if (n_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)n_blocks), dim3(64), 0, st, d_comp, d_blocks, d_out, d_status);
  return hipGetLastError();
}

struct svdss_inflate {
  int device = -1;
  hipStream_t st = nullptr;
  DevBuf comp, blks, status;
  std::vector<int32_t> h_status;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double kernel_ms = 0.0;
};

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      if (getenv("SVDSS_DEBUG")) fprintf(stderr, "[inflate] %s: %s\n", #x, hipGetErrorString(e_));  \
      return e_ == hipErrorOutOfMemory ? SVDSS_ENOMEM : SVDSS_EHIP;                                 \
    }                                                                                               \
  } while (0)

static int ensure(DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return SVDSS_OK;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  const size_t want = bytes + (bytes >> 2) + 4096;
  HIPCHK(hipMalloc(&b.p, want));
  b.cap = want;
  return SVDSS_OK;
}

extern "C" int svdss_bgzf_inflate(svdss_inflate_t** obj, int device, const uint8_t* comp, int64_t comp_bytes,
                                  const svdss_bgzf_block_t* blocks, int64_t n_blocks, void* d_out, uint8_t* host_out,
                                  int64_t out_bytes, int64_t* bad_block) {
  if (!obj || comp_bytes < 0 || n_blocks < 0 || out_bytes < 0) return SVDSS_EINVAL;
  if (n_blocks > 0 && (!comp || !blocks || !d_out)) return SVDSS_EINVAL;
  if (bad_block) *bad_block = -1;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return SVDSS_ENODEV;
  for (int64_t i = 0; i < n_blocks; ++i) {
    const svdss_bgzf_block_t& b = blocks[i];
    if (b.coff < 0 || b.clen < 0 || b.isize < 0 || b.isize > 65536 || b.uoff < 0 || b.coff + b.clen > comp_bytes ||
        b.uoff + b.isize > out_bytes)
      return SVDSS_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  svdss_inflate* o = *obj;
  if (!o) {
    o = new (std::nothrow) svdss_inflate();
    if (!o) return SVDSS_ENOMEM;
    o->device = device;
    *obj = o;
  }
  if (o->device != device) return SVDSS_EINVAL;
  if (!o->st) HIPCHK(hipStreamCreateWithFlags(&o->st, hipStreamNonBlocking));
  if (n_blocks == 0) return SVDSS_OK;
  int rc;
  // (the kernel reads the input in aligned 1 KB pieces, up to 3 KB past a block's last byte)
  if ((rc = ensure(o->comp, (size_t)comp_bytes + 4096))) return rc;
  if ((rc = ensure(o->blks, sizeof(svdss_bgzf_block_t) * (size_t)n_blocks))) return rc;
  if ((rc = ensure(o->status, sizeof(int32_t) * (size_t)n_blocks))) return rc;
  HIPCHK(hipMemcpyAsync(o->comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, o->st));
  HIPCHK(hipMemcpyAsync(o->blks.p, blocks, sizeof(svdss_bgzf_block_t) * (size_t)n_blocks, hipMemcpyHostToDevice, o->st));
  if (!o->ev0) { HIPCHK(hipEventCreate(&o->ev0)); HIPCHK(hipEventCreate(&o->ev1)); }
  HIPCHK(hipEventRecord(o->ev0, o->st));
  hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)n_blocks), dim3(64), 0, o->st, (const uint8_t*)o->comp.p,
                     (const svdss_bgzf_block_t*)o->blks.p, (uint8_t*)d_out, (int32_t*)o->status.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(o->ev1, o->st));
  o->h_status.resize((size_t)n_blocks);
  HIPCHK(hipMemcpyAsync(o->h_status.data(), o->status.p, sizeof(int32_t) * (size_t)n_blocks, hipMemcpyDeviceToHost, o->st));
  if (host_out && out_bytes > 0) HIPCHK(hipMemcpyAsync(host_out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, o->st));
  HIPCHK(hipStreamSynchronize(o->st));
  {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, o->ev0, o->ev1) == hipSuccess) o->kernel_ms = (double)ms;
  }
#ifdef INF_COUNT
  {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_inf_cnt), sizeof h) == hipSuccess) {
      fprintf(stderr, "[inflate] per block: rounds %.0f literals %.0f matches %.0f (bytes %.0f, overlapping %.0f) one-symbol steps %.0f deflate blocks %.2f\n",
              (double)h[0] / n_blocks, (double)h[1] / n_blocks, (double)h[2] / n_blocks, (double)h[5] / n_blocks, (double)h[6] / n_blocks,
              (double)h[3] / n_blocks, (double)h[4] / n_blocks);
      memset(h, 0, sizeof h);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_inf_cnt), h, sizeof h);
    }
  }
#endif
  for (int64_t i = 0; i < n_blocks; ++i)
    if (o->h_status[(size_t)i] != 0) {
      if (bad_block) *bad_block = i;
      if (getenv("SVDSS_DEBUG")) fprintf(stderr, "[inflate] block %lld: status %d\n", (long long)i, o->h_status[(size_t)i]);
      return SVDSS_EIO;
    }
  return SVDSS_OK;
}

extern "C" double svdss_inflate_kernel_ms(const svdss_inflate_t* o) { return o ? o->kernel_ms : -1.0; }

extern "C" void svdss_inflate_free(svdss_inflate_t* o) {
  if (!o) return;
  if (o->device >= 0) (void)hipSetDevice(o->device);
  if (o->ev0) (void)hipEventDestroy(o->ev0);
  if (o->ev1) (void)hipEventDestroy(o->ev1);
  for (DevBuf* d : {&o->comp, &o->blks, &o->status})
    if (d->p) (void)hipFree(d->p);
  if (o->st) (void)hipStreamDestroy(o->st);
  delete o;
}

extern "C" int svdss_device_alloc(int device, int64_t bytes, void** out) {
  if (!out || bytes < 0) return SVDSS_EINVAL;
  *out = nullptr;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipMalloc(out, (size_t)(bytes ? bytes : 16)));
  return SVDSS_OK;
}

extern "C" void svdss_device_free(int device, void* p) {
  if (!p) return;
  (void)hipSetDevice(device);
  (void)hipFree(p);
}

extern "C" int svdss_device_memset(int device, void* d_dst, int value, int64_t bytes) {
  if (bytes < 0 || (bytes > 0 && !d_dst)) return SVDSS_EINVAL;
  HIPCHK(hipSetDevice(device));
  if (bytes) HIPCHK(hipMemset(d_dst, value, (size_t)bytes));
  return SVDSS_OK;
}

extern "C" int svdss_device_download(int device, void* dst, const void* d_src, int64_t bytes) {
  if (bytes < 0 || (bytes > 0 && (!dst || !d_src))) return SVDSS_EINVAL;
  HIPCHK(hipSetDevice(device));
  if (bytes) HIPCHK(hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
  return SVDSS_OK;
}
