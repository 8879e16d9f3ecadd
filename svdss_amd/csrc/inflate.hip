// inflate.hip -- BGZF blocks inflated on the GPU (svdss_bgzf_inflate).
//
// Stands where htslib's bgzf_read / inflate stand under sam_read1 (/root/reference/ping_pong.cpp:58,247-249;
// clusterer.cpp:101; smoother.cpp:262): `SVDSS search` reads ~1.5 bytes of BAM per base, all of it deflate streams of
// at most 64 KB, and inflating them is the largest piece of GPU work of the binary end to end (a host core inflates
// ~0.3 GB/s of this kind of data; a 30x human sample is ~140 GB inflated).  Every BGZF block is an independent stream, so
// the GPU takes one wavefront per block and a few thousand blocks at a time:
//   * the most recent 4 KB of output live in an LDS ring and leave for HBM as aligned dwords; a match that reaches
//     further back than the ring (deflate allows 32 KB) reads the bytes the block itself wrote to HBM earlier -- the ring is
//     that small so that a CU holds 10 blocks, and the latency of those reads is hidden by the other wavefronts;
//   * the compressed bytes pass through a 4 KB LDS ring, refilled 512 bytes at a time by the whole wave;
//   * Huffman tables (9-bit literal/length, 8-bit distance; 32-bit entries that hold everything a lane needs to know
//     about a code: length, kind, extra bits, base; longer codes are decoded canonically by the lane that meets one) are
//     built by the wave in parallel: the canonical code of a symbol is the rank of the symbol among those of its length
//     (wave ballots), and every lane fills the table entries of its own symbols;
//   * symbols are decoded in PASSES over 64 x 288 bits (round 4): every lane walks the symbols of its own 288 bits one
//     after the other -- 64 symbols per instruction instead of the ~8 that start within 64 bits --, from a guessed start
//     first; Huffman codes resynchronise, so the walks' exits are mostly right, every lane walks again from its
//     neighbour's exit, and the lanes whose start then equals their neighbour's exit are the true chain of symbols.  A
//     scan places their output, a last walk stores the literals and queues the matches, the matches are copied 64 side
//     by side.  (See the comment above the symbol loop; rounds of 64 bits -- rounds 2-3's engine, every lane decoding the
//     symbol that would start at its bit offset, the chain found by pointer doubling -- remain for what a pass cannot
//     take, the tail of a stream, and data that does not resynchronise.)
// 15.4 KB of LDS per block.  No CRC check here (bam_device.hip's crc32_kernel, or the caller on the host, checks the BGZF
// footers).  Measured on 16,384 blocks of 64 KB (tools/inflate_probe.py, GB/s of output; rounds 2-3's kernel in
// brackets): packed bases + random qualities zlib level 1: 82 (23.7), level 6: 88 (25); binned qualities 75 (35); skewed
// 94-value qualities 79 (27); literals only, as csrc/deflate.hip writes them: 81 (42); text 333 (335).
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

// ---- sizes (macros so that tools/r04_inflate_sweep.sh can build variants)
#ifndef INF_WIN
#define INF_WIN 4096        // the LDS ring of recent output, bytes
#endif
#ifndef INF_INB
#define INF_INB 4096        // the LDS ring of compressed input, bytes
#endif
#ifndef INF_SD
#define INF_SD 9            // dwords of input per lane and pass (odd: the lanes' reads fall into different LDS banks)
#endif
#ifndef INF_MAXM
#define INF_MAXM 512        // matches a pass may hold back until its literals are placed
#endif
#ifndef INF_WALKS
#define INF_WALKS 1         // walks per pass: then it takes the lanes that agree, and the others keep theirs for the next ...
#endif
#ifndef INF_TAKE
#define INF_TAKE 24         // ... unless fewer than this many agree: then up to ...
#endif
#ifndef INF_MAXW
#define INF_MAXW 3          // ... this many walks
#endif
#ifndef INF_SIDE_MAX
#define INF_SIDE_MAX 32     // longest match (a multiple of 4) copied side by side by the lane that holds it
#endif
#ifndef INF_GUESTS
#define INF_GUESTS 1        // the lanes a pass does not take guess, during its last walk, for the pieces of the next pass
#endif
#ifndef INF_LB
#define INF_LB 9
#endif
#ifndef INF_DB
#define INF_DB 8
#endif
constexpr int WIN = INF_WIN, WM = WIN - 1;
constexpr int FLUSH = WIN / 8;        // (rounds) the ring is written out whenever this many bytes are waiting
constexpr int ROUND_MAX = WIN / 8;    // a round stops taking symbols once it has produced this many bytes (+ one match)
// Which of a match's source bytes are read from the ring, and which from HBM, where an earlier flush put them:
// * in a round (flushes when FLUSH bytes wait, writes at most ROUND_MAX + 258): the bytes closer than NEAR to the match's
//   output position are in the ring -- the two static_asserts below;
// * in a pass (flushes before it starts, then writes at most PASS_MAX bytes): the bytes from PASS_BACK before the pass's
//   first output byte on -- PASS_BACK + PASS_MAX <= WIN keeps them in the ring until the pass is over, and everything
//   before them is in HBM (a flush leaves fewer than 4 bytes behind).
constexpr int NEAR = WIN / 2 - 96;
constexpr int PASS_BACK = 64;
constexpr int PASS_MAX = WIN - PASS_BACK - 264;
static_assert(FLUSH + ROUND_MAX + 258 + 64 < NEAR && NEAR + ROUND_MAX + 258 + 64 < WIN, "ring too small for a round");
constexpr int LB = INF_LB, DB = INF_DB, CB = 7;   // bits of the direct tables (literal / length, distance, code lengths)
constexpr int INB = INF_INB, HALF = 512;          // input ring, refilled HALF bytes at a time
constexpr int SB = 32 * INF_SD;                   // bits of input per lane and pass
constexpr int PASS_BYTES = 64 * SB / 8 + 16;      // what a pass may read beyond its first bit
static_assert(INB - HALF - 32 >= PASS_BYTES, "input ring too small for a pass");
static_assert(PASS_MAX < 65536, "match records hold 16 bits of output offset");
constexpr int MAXM = INF_MAXM;

struct Lds {
  uint32_t win[WIN / 4];
  uint32_t lit[1 << LB];
  uint32_t dst[1 << DB];       // (its first 256 bytes also hold the code-length code's table while a header is read)
  uint32_t inb[INB / 4];
  uint32_t mrec[MAXM];         // the matches a pass holds back: (length - 3) << 15 | (distance - 1) ...
  uint16_t mpos[MAXM];         // ... and where they write, relative to the pass's first output byte
  uint8_t lens[320];
  uint16_t lsym[288];
  uint16_t dsym[32];
  uint16_t lcnt[16], dcnt[16];
};

#ifdef INF_COUNT
// rounds, literals(unused), matches of rounds, one-symbol steps, deflate blocks, passes, lanes taken, walks, walk steps, matches of passes, of them one by one, run fills of rounds
__device__ unsigned long long g_inf_cnt[12];
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

// ---- table entries: everything a lane needs to know about a code, so that decoding is two lookups and a few shifts
// literal / length code:  bits 0-3 code length (0: no code this short starts with these bits), 4-5 kind (0 literal, 1 length,
//   2 end of block, 3 none / invalid), 6-8 number of extra bits, 9-17 the literal or the length's base
constexpr uint32_t LIT_NONE = 3u << 4;
struct LitEntry {
 __device__ __forceinline__ uint32_t operator()(uint32_t s, uint32_t ml) const {
  if (s < 256u) return ml | (s << 9);
  if (s == 256u) return ml | (2u << 4);
  if (s > 285u) return ml | (3u << 4);
  const uint32_t ls = s - 257u;
  const uint32_t lx = ls < 8u || ls == 28u ? 0u : (ls >> 2) - 1u;
  const uint32_t lb = ls < 8u ? 3u + ls : ls == 28u ? 258u : 3u + ((4u + (ls & 3u)) << lx);
  return ml | (1u << 4) | (lx << 6) | (lb << 9);
 }
};
// distance code: bits 0-3 code length (0: none), 4-7 number of extra bits, 8-23 base
struct DstEntry {
 __device__ __forceinline__ uint32_t operator()(uint32_t s, uint32_t ml) const {
  if (s > 29u) return 0u;
  const uint32_t dx = s < 4u ? 0u : (s >> 1) - 1u;
  const uint32_t db = s < 4u ? 1u + s : 1u + ((2u + (s & 1u)) << dx);
  return ml | (dx << 4) | (db << 8);
 }
};
struct LenEntry {
  __device__ __forceinline__ uint16_t operator()(uint32_t s, uint32_t ml) const { return (uint16_t)((s << 4) | ml); }
};

// Canonical Huffman code of n symbols with lengths lens[0..n): table of 2^tb entries (make(symbol, length); `none` where a
// longer code or none starts), per-length counts and the symbols sorted by (length, symbol) for the bit-by-bit decoder.
// Returns false if the lengths over-subscribe the code space.
template <class T, class F>
__device__ bool build_table(const uint8_t* lens, int n, T* tab, int tb, uint16_t* cnt, uint16_t* sym, int lane, T none, F make,
                            uint32_t* state = nullptr) {   // the bit-by-bit decoder's first << 16 | index after tb levels
  for (int k = lane; k < (1 << tb); k += 64) tab[k] = none;
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
  if (state) {
    int f = 0, ix = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l)
      if (l <= tb) { ix += count[l]; f = (f + count[l]) << 1; }
    *state = ((uint32_t)f << 16) | (uint32_t)ix;
  }
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
        const T e = make((uint32_t)s, (uint32_t)ml);
        for (uint32_t k = rev; k < (1u << tb); k += (1u << ml)) tab[k] = e;
      }
    }
  }
  return true;
}

// A code longer than the direct table's tb index bits, decoded canonically from level tb + 1 on by the lane that met it:
// bits = the stream from the code's first bit (at least 15 valid), state = the decoder's first << 16 | index after tb
// levels.  Symbol | length << 16, or ~0 when no code of at most 15 bits starts like this.
__device__ __forceinline__ uint32_t long_code(uint32_t bits, int tb, uint32_t state, const uint16_t* cnt, const uint16_t* sym) {
  int code = (int)(__brev(bits) >> (32 - tb)) << 1, first = (int)(state >> 16), index = (int)(state & 0xffffu);
  uint32_t b = bits >> tb;
  for (int l = tb + 1; l <= 15; ++l) {
    code |= (int)(b & 1u);
    b >>= 1;
    const int count = (int)cnt[l];
    if (code - count < first) return (uint32_t)sym[index + (code - first)] | ((uint32_t)l << 16);
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return ~0u;
}

// what a lane finds at a bit position
struct Sym {
  uint32_t kind;     // 0 literal, 1 match, 2 end of block, 3 = a code longer than the table's index bits, or an invalid one
  uint32_t nbits;    // the whole symbol: code, extra bits, distance code, extra bits
  uint32_t outlen;   // bytes it produces
  uint32_t val;      // the literal, or the match's length
  uint32_t dist;
};

// one BGZF member (block `member` of blks), by one wavefront
__device__ __forceinline__ void inflate_member(Lds& L, const uint8_t* __restrict__ comp, const svdss_bgzf_block_t* __restrict__ blks, uint8_t* out,
                               int32_t* __restrict__ status, const int64_t member) {
  const int lane = threadIdx.x;
  const svdss_bgzf_block_t B = blks[member];
  const uint32_t isize = (uint32_t)B.isize;
  const uint64_t in_first = (uint64_t)B.coff, in_end = in_first + (uint64_t)(uint32_t)B.clen;
  uint8_t* const o8 = out + B.uoff;
  uint8_t* const winb = (uint8_t*)L.win;
  if (isize == 0) { if (lane == 0) status[member] = ST_OK; return; }
#ifdef INF_COUNT
  unsigned cnt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#define FAIL(code) do { status[member] = (code); return; } while (0)

  // ---- input: absolute offsets into comp; the ring holds [base, base + INB)
  uint64_t base = in_first & ~(uint64_t)(HALF - 1), in_addr = in_first;
  static_assert(HALF == 64 * 8, "one uint2 per lane");
  auto fill_from = [&](uint64_t a0, int n_half) {   // n_half (at most 8) pieces of HALF bytes from a0 on, 8 bytes per lane each
    uint2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < n_half) v[i] = *(const uint2*)(comp + a0 + (uint64_t)(i * HALF) + (uint64_t)lane * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < n_half) *(uint2*)((uint8_t*)L.inb + ((a0 + (uint64_t)(i * HALF) + (uint64_t)lane * 8) & (INB - 1))) = v[i];
  };
  auto restart_at = [&](uint64_t a) {   // the ring anew, from the piece that holds byte a
    base = a & ~(uint64_t)(HALF - 1);
    for (int h = 0; h < INB / HALF; h += 8) fill_from(base + (uint64_t)(h * HALF), INB / HALF - h < 8 ? INB / HALF - h : 8);
  };
  // the ring follows the reader: whenever the reader is HALF + 32 bytes past the ring's first byte, the piece it left is
  // replaced by the next one not held yet (32 bytes late: the bit buffer holds up to 8 bytes that were read before
  // in_addr, and rounds and passes read the ring at the position of the first unused bit)
  auto catch_up = [&](uint64_t reader) {
    while (reader - base >= (uint64_t)(HALF + 32)) {
      const uint32_t can = (uint32_t)((reader - base - 32) / HALF);
      const int n = (int)(can < 8u ? can : 8u);
      fill_from(base + INB, n);
      base += (uint64_t)(n * HALF);
    }
  };
  restart_at(in_addr);
  uint64_t bb = 0;
  int bc = 0;
  auto align_reader = [&]() {   // single bytes until in_addr is a multiple of 4
    for (; (in_addr & 3) != 0; ++in_addr) {
      const uint32_t b = UNI((uint32_t)((const uint8_t*)L.inb)[in_addr & (INB - 1)]);
      bb |= (uint64_t)b << bc;
      bc += 8;
    }
  };
  align_reader();
  auto refill = [&]() {   // at least 33 bits afterwards
    if (bc <= 32) {
      const uint32_t w = UNI(L.inb[(in_addr & (INB - 1)) >> 2]);
      bb |= (uint64_t)w << bc;
      bc += 32;
      in_addr += 4;
      catch_up(in_addr);
    }
  };
  auto take = [&](int n) { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; };
  auto reader_to_bit = [&](uint64_t P) {   // the bit buffer, from bit position P
    in_addr = (P >> 5) << 2;
    catch_up(in_addr);
    const uint32_t v0 = UNI(L.inb[(in_addr & (INB - 1)) >> 2]), v1 = UNI(L.inb[((in_addr + 4) & (INB - 1)) >> 2]);
    const int sh = (int)(P & 31);
    bb = (((uint64_t)v1 << 32) | v0) >> sh;
    bc = 64 - sh;
    in_addr += 8;
    catch_up(in_addr);
  };
  // one code through the bit buffer: the table's entry, or -- a code longer than the table's index bits -- the entry
  // `make` gives the symbol that canonical decoding finds; ~0 if there is no such code
  auto decode = [&](const auto* tab, int tb, const uint16_t* cnt, const uint16_t* sym, auto make) -> uint32_t {
    refill();
    const uint32_t e = UNI((uint32_t)tab[bb & ((1u << tb) - 1)]);
    const int len = (int)(e & 15u);
    if (len) { bb >>= len; bc -= len; return e; }
    int code = 0, first = 0, index = 0;
    uint64_t b = bb;
    for (int l = 1; l <= 15; ++l) {
      code |= (int)(b & 1);
      b >>= 1;
      const int count = UNI((int)cnt[l]);
      if (code - count < first) { bb >>= l; bc -= l; return (uint32_t)make((uint32_t)UNI((int)sym[index + (code - first)]), (uint32_t)l); }
      index += count; first += count; first <<= 1; code <<= 1;
    }
    return ~0u;
  };

  // ---- output: the ring holds the most recent WIN bytes; [flushed, wpos) is not in HBM yet
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

  // a match of ml bytes at output position o, dd bytes back, copied by the whole wave.  Source bytes from position
  // ring_lo on are in the ring; older ones were written to HBM by an earlier flush and are read back from there -- with
  // a ring this small a CU holds several blocks, and that latency is what the other wavefronts are for.
  const uint32_t inv_lane = lane ? 65536u / (uint32_t)lane + 1u : 0u;   // (k mod d for d < 64, k < 512: k - ((k * inv_d) >> 16) * d; lane d keeps inv_d)
  auto copy_match = [&](uint32_t o, uint32_t ml, uint32_t dd, uint32_t ring_lo) {
    const uint32_t from = o - dd;
    if (from >= ring_lo && dd >= ml && ml <= 64) {   // the common one: short, from the ring, not overlapping itself
      if ((uint32_t)lane < ml) winb[(o + (uint32_t)lane) & WM] = winb[(from + (uint32_t)lane) & WM];
      return;
    }
    const bool far = from < ring_lo;
    if (far) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this block's own stores of long ago)
    uint8_t v[5];
    int nk = 0;
    const uint32_t inv = dd < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)inv_lane, (int)(dd & 63u)) : 0u;
    for (uint32_t k = lane; k < ml; k += 64) {
      // (an overlapping match repeats its first dd bytes: k mod dd, without a division for the distances runs have)
      const uint32_t sp = from + (dd >= ml ? k : dd < 64u ? k - ((k * inv) >> 16) * dd : k % dd);
      v[nk++] = sp < ring_lo ? __hip_atomic_load(o8 + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : winb[sp & WM];
    }
    nk = 0;
    for (uint32_t k = lane; k < ml; k += 64) winb[(o + k) & WM] = v[nk++];
  };
  auto near_lo = [&](uint32_t o) { return o > (uint32_t)NEAR ? o - (uint32_t)NEAR : 0u; };   // (rounds)

  uint32_t lstate = 0, dstate = 0;   // (long_code: the canonical decoders' state behind the direct tables' levels)
  // the symbol that starts at bit position pl (modulo 2^32: the ring is indexed by the low bits), whole: a literal, or
  // a length with its extra bits, its distance code and that one's extra bits -- at most 48 bits --, or the end-of-block code
  auto symbol_at = [&](uint32_t pl) -> Sym {
    const uint32_t di = pl >> 5;
    const uint32_t w0 = L.inb[di & (INB / 4 - 1)], w1 = L.inb[(di + 1) & (INB / 4 - 1)], w2 = L.inb[(di + 2) & (INB / 4 - 1)];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, pl & 31u), hi = __builtin_amdgcn_alignbit(w2, w1, pl & 31u);
    const uint64_t b64 = ((uint64_t)hi << 32) | lo;
    uint32_t E = L.lit[lo & ((1u << LB) - 1)];
    uint32_t len = E & 15u, lx = (E >> 6) & 7u, kind = (E >> 4) & 3u;
    uint32_t doff = len + lx;
    uint32_t dbits = (uint32_t)(b64 >> doff);
    uint32_t D = L.dst[dbits & ((1u << DB) - 1)];
    // (codes longer than the tables' index bits: one test for both tables, the lanes that met one finish it canonically)
    if (__ballot(len == 0u || (kind == 1u && (D & 15u) == 0u))) {
      if (len == 0u) {
        const uint32_t r = long_code(lo, LB, lstate, L.lcnt, L.lsym);
        if (r != ~0u) {
          E = LitEntry()(r & 0xffffu, r >> 16);
          len = E & 15u; lx = (E >> 6) & 7u; kind = (E >> 4) & 3u;
          doff = len + lx;
          dbits = (uint32_t)(b64 >> doff);
          D = L.dst[dbits & ((1u << DB) - 1)];
        }
      }
      if (kind == 1u && (D & 15u) == 0u) {
        const uint32_t r = long_code(dbits, DB, dstate, L.dcnt, L.dsym);
        if (r != ~0u) D = DstEntry()(r & 0xffffu, r >> 16);
      }
    }
    Sym s;
    s.kind = kind;
    s.val = ((E >> 9) & 511u) + __builtin_amdgcn_ubfe(lo, len, lx);
    const uint32_t dl = D & 15u, dx = (D >> 4) & 15u;
    s.dist = (D >> 8) + __builtin_amdgcn_ubfe(dbits, dl, dx);
    if (s.kind == 1u && dl == 0u) s.kind = 3u;
    s.nbits = s.kind == 1u ? doff + dl + dx : len;
    s.outlen = s.kind == 0u ? 1u : s.kind == 1u ? s.val : 0u;
    return s;
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
      restart_at(in_addr);
      bb = 0; bc = 0;
      align_reader();
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
        uint16_t* const ctab = (uint16_t*)L.dst;
        if (!build_table(L.lens, 19, ctab, CB, L.dcnt, L.dsym, lane, (uint16_t)0, LenEntry())) FAIL(ST_LENS);
        // the code lengths of the literal/length and distance codes, run-length coded (all lanes keep the same copy)
        uint8_t* const tmp = (uint8_t*)L.lsym;   // (free until the tables are built)
        int i = 0, prev = 0;
        while (i < nlen + ndist) {
          const uint32_t e = decode(ctab, CB, L.dcnt, L.dsym, LenEntry());
          if (e == ~0u) FAIL(ST_LENS);
          const int s = (int)(e >> 4);
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
      if (!build_table(L.lens, nlen, L.lit, LB, L.lcnt, L.lsym, lane, LIT_NONE, LitEntry(), &lstate)) FAIL(ST_LENS);
      if (!build_table(L.lens + 288, ndist, L.dst, DB, L.dcnt, L.dsym, lane, 0u, DstEntry(), &dstate)) FAIL(ST_LENS);
      lstate = (uint32_t)UNI((int)lstate); dstate = (uint32_t)UNI((int)dstate);
      // ---- symbols.  Two engines share the rings, the tables and the cursor (the bit position P, the output position):
      //
      // A PASS takes 64 * SB bits at once, SB bits per lane.  Every lane walks the symbols of its own piece one after
      // the other (symbol_at: two table lookups per symbol, 64 symbols per instruction) from a start that is a guess at
      // first -- the piece's first bit -- and notes where its walk leaves the piece.  Huffman codes resynchronise: after a
      // few symbols a walk from a wrong start falls in step with the true chain of symbols, so most exits are right
      // although most starts were wrong.  Then every lane walks again from where its neighbour left.  Lane 0's start is
      // true; by induction so are all that agree with their neighbour's exit, up to the first lane that does not (its
      // neighbour had not fallen in step by the end of its piece: one piece in twenty).  The pass takes the lanes in
      // agreement: a scan over their byte and match counts places every lane's output, a last walk stores the literals in
      // the ring and queues the matches, which are then copied 64 at a time.  The lanes it does not take move down and keep
      // their walks: in the next pass they are repaired (the first of them starts where this pass ended).  And they are idle
      // during the pass's last walk: there they guess for the pieces that ENTER with the next pass (piece 64 + i by lane
      // T + i, as far as the 3.5 KB of input ring ahead of P reach: ~30 pieces), noting the exit only -- those pieces start
      // the next pass with their second walk.  One counting walk per pass (up to three while fewer than 24 lanes agree) and
      // the storing walk; ~40 lanes taken.  A pass ends early at the end-of-block code, at a code no table entry or
      // canonical decoding explains (the round below reports it), at the ring's or the queue's capacity.
      //
      // A ROUND takes up to 64 bits: lane l decodes the symbol that would start l bits ahead, the chain of symbol starts
      // from offset 0 is found by pointer doubling on the lanes (lane l knows where the symbol after its own starts, J,
      // and which offsets the chain from l visits, M; six steps of "append the chain of the lane I point at" close M),
      // literals are stored together, matches copied in order by all lanes.  A code longer than the table's index bits
      // takes the one-symbol path (bit buffer, canonical decoding).  Rounds run between passes -- they take what ends a
      // pass -- and instead of passes where passes do not pay: near the end of the input, and for a while after a pass that
      // found few lanes in agreement.
      uint32_t err = 0;
      bool eob = false;
      uint64_t P = in_addr * 8 - (uint64_t)bc;
      uint32_t skip_pass = 0, small = 0;
      // what the lanes keep from pass to pass (within this deflate block): each lane's last walk -- where it started and
      // where it left its piece, relative to P; the bytes and matches it counted; how it ended (0 left the piece, 1 end of
      // block, 2 stopped at a code the tables cannot decode) -- and where the pieces lie: lane l's ends (l + 1) * SB - delta
      // bits behind P.  The lanes a pass does not take move down and keep theirs.
      uint32_t start = 0, x = 0, c = 0, m = 0, st = 0, delta = 0;
      bool valid = false;
      for (;;) {
        catch_up(P >> 3);
        bool round_now = true;
        if (skip_pass) --skip_pass;
        else if ((P >> 3) + (uint64_t)(PASS_BYTES / 4) < in_end) {
          CNT(5, 1);
          flush(false);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (far sources of this pass's matches: stores of this block)
          const uint32_t P32 = (uint32_t)P;
          const uint32_t ring_lo = wpos > (uint32_t)PASS_BACK ? wpos - (uint32_t)PASS_BACK : 0u;
          const uint32_t hi_bound = (uint32_t)(lane + 1) * (uint32_t)SB - delta;
          // ---- INF_WALKS walks: a lane that has not walked yet starts at its piece's first bit (a guess); one whose
          // neighbour left elsewhere than it started walks again from there; the others keep what they have
          uint32_t n_ok = 0;
          for (int it = 0;; ++it) {
            const uint32_t px = (uint32_t)__shfl_up((int)x, 1), pst = (uint32_t)__shfl_up((int)st, 1) & 3u;   // (bit 2 of st: see the last walk)
            const bool pv = __shfl_up((int)valid, 1) != 0;
            bool walk;
            uint32_t from_;
            if (lane == 0) { walk = !valid || start != 0u || (st & 4u) != 0u; from_ = 0u; }
            else if (!valid) { walk = true; from_ = (uint32_t)lane * (uint32_t)SB - delta; }
            else { walk = pv && pst == 0u && (px != start || (st & 4u) != 0u); from_ = px; }   // (a walk that did not count, from the right bit: again)
            // in agreement: lane 0 if it started at P, a lane whose neighbour left where it started (by induction all up
            // to the first that is not are the true chain of symbols)
            const bool agrees = valid && !walk && (lane == 0 || (pv && pst == 0u));
            const unsigned long long ag = __ballot(agrees);
            n_ok = ag == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~ag);
            // (the first lane not in agreement walks now -- unless the chain ended before it: end of block, a stop)
            if (n_ok == 64u || !__builtin_amdgcn_readlane((int)walk, (int)n_ok) || it >= INF_MAXW || (it >= INF_WALKS && n_ok >= (uint32_t)INF_TAKE)) break;
            CNT(7, 1);
            if (walk) { start = from_; c = 0; m = 0; st = 0; valid = true; }
            uint32_t r = walk ? start : x;
            bool going = walk && r < hi_bound;
            while (__ballot(going)) {
              CNT(8, 1);
              const Sym s = symbol_at(P32 + r);
              if (going) {
                if (s.kind == 3u) { st = 2u; going = false; }
                else {
                  r += s.nbits;
                  c += s.outlen;
                  m += s.kind & 1u;
                  if (s.kind == 2u) { st = 1u; going = false; }
                  else going = r < hi_bound;
                }
              }
            }
            x = r;
          }
          // ---- the lanes taken: in agreement, within the ring's and the queue's capacity
          const bool mine = (uint32_t)lane < n_ok;
          const uint32_t ic = (uint32_t)wave_scan_add((int)(mine ? c : 0u)), im = (uint32_t)wave_scan_add((int)(mine ? m : 0u));
          const unsigned long long fits = __ballot(mine && ic <= (uint32_t)PASS_MAX && im <= (uint32_t)MAXM);
          const uint32_t T = fits == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fits);
          if (T == 0u) skip_pass = 4;
          else {
            CNT(6, T);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)ic, (int)(T - 1u));
            const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)im, (int)(T - 1u));
            const uint32_t adv = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)(T - 1u));
            const uint32_t lst = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)(T - 1u));
            if (wpos + total > isize) { err = ST_OUT; break; }
            // ---- the last walk: literals into the ring, matches into the queue.  The lanes the pass does not take have
            // nothing to do in it: they guess for the pieces that enter with the next pass (piece 64 + i by lane T + i, as
            // far as the input ring reaches), so that those start the next pass with their second walk.
            uint32_t gstart = 0, gx = 0, gst = 0;
            bool gvalid = false;
            {
              const bool fin = (uint32_t)lane < T;
              gstart = (64u + ((uint32_t)lane - T)) * (uint32_t)SB - delta;
              const uint32_t ghi = gstart + (uint32_t)SB;
              gvalid = INF_GUESTS && !fin && lst == 0u && ghi + 128u <= (uint32_t)(INB - HALF - 32) * 8u;
              const uint32_t hi = fin ? hi_bound : ghi;
              uint32_t r = fin ? start : gstart, o = wpos + (ic - c), q = im - m;   // (lanes below T: c and m are those of the walk from start)
              bool going = fin ? r < hi_bound : gvalid;
              bool bad = false;
              while (__ballot(going)) {
                const Sym s = symbol_at(P32 + r);
                if (going) {
                  if (s.kind >= 2u) {
                    if (!fin) { gst = s.kind == 2u ? 1u : 2u; if (s.kind == 2u) r += s.nbits; }
                    going = false;
                  } else {
                    if (fin) {
                      if (s.kind == 0u) winb[o & WM] = (uint8_t)s.val;
                      else {
                        if (s.dist > o) bad = true;
                        L.mrec[q] = ((s.val - 3u) << 15) | (s.dist - 1u);
                        L.mpos[q] = (uint16_t)(o - wpos);
                        ++q;
                      }
                      o += s.outlen;
                    }
                    r += s.nbits;
                    going = r < hi;
                  }
                }
              }
              gx = r;
              if (__ballot(bad)) { err = ST_DIST; break; }
            }
            // ---- the matches, 64 at a time in output order, each lane holding one.  Everything below the first one's output
            // position is complete (literals, earlier matches).  Copied side by side by the lanes that hold them: the short
            // ones whose source ends there, or begins behind the match before them (between two matches there are only
            // literals, and those are in place; such a match may overlap its own output).  The others -- chained closely, as
            // in low-entropy data, or long -- one after the other by the whole wave.  (Tried: further side-by-side rounds for
            // those that become ready -- 8 rounds per 64 on binned qualities, each dearer than three single copies.)
            CNT(9, M);
            for (uint32_t g0 = 0; g0 < M; g0 += 64) {
              const bool have = g0 + (uint32_t)lane < M;
              const uint32_t rec = have ? L.mrec[g0 + (uint32_t)lane] : 0u;
              const uint32_t o = wpos + (have ? (uint32_t)L.mpos[g0 + (uint32_t)lane] : 0u), ml = (rec >> 15) + 3u, dd = (rec & 0x7fffu) + 1u;
              const uint32_t from = o - dd;
              const uint32_t o_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
              const uint32_t pe = (uint32_t)__shfl_up((int)(o + ml), 1);
              const bool side = have && ml <= (uint32_t)INF_SIDE_MAX && (from + ml <= o_first || lane == 0 || from >= pe);
              uint32_t sp = from;
              for (uint32_t k0 = 0; k0 < (uint32_t)INF_SIDE_MAX; k0 += 4u) {
                if (!__ballot(side && k0 < ml)) break;
                uint8_t v[4];
#pragma unroll
                for (uint32_t t = 0; t < 4u; ++t) {
                  if (side && k0 + t < ml) {
                    v[t] = sp < ring_lo ? __hip_atomic_load(o8 + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : winb[sp & WM];
                    ++sp;
                    if (sp == o) sp = from;     // (a match that overlaps its own output repeats its first dd bytes)
                  }
                }
#pragma unroll
                for (uint32_t t = 0; t < 4u; ++t) {
                  const uint32_t k = k0 + t;
                  if (side && k < ml) winb[(o + k) & WM] = v[t];
                }
              }
              // the others, in order, by the whole wave.  What a chain of such copies costs is instructions (the scalar side
              // fetches each match from the lane that holds it), so the lanes prepare two words per match: position, length
              // and "short, from the ring, not overlapping itself" in one, the source position in the other
              unsigned long long rest = __ballot(have && !side);
              // (bit 31: source and output apart; bit 30: the match overlaps its own output, distance below 64 -- a run: lane k
              // reads byte k mod distance of the source, by a multiplication with 65536 / distance + 1, which lane `distance`
              // keeps)
              const uint32_t near_short = from >= ring_lo && ml <= 64u;
              const uint32_t pa = (o & 0xffffu) | (ml << 16) | ((near_short && dd >= ml) ? 0x80000000u : 0u) | ((near_short && dd < ml && dd < 64u) ? 0x40000000u : 0u);
              while (rest) {
                const int h = (int)__builtin_ctzll(rest);
                rest &= rest - 1;
                CNT(10, 1);
                const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)pa, h), Bf = (uint32_t)__builtin_amdgcn_readlane((int)from, h);
                if (A & 0x80000000u) {
                  if ((uint32_t)lane < ((A >> 16) & 0x1ffu)) winb[((A & 0xffffu) + (uint32_t)lane) & WM] = winb[(Bf + (uint32_t)lane) & WM];
                } else if (A & 0x40000000u) {
                  const uint32_t o1 = A & 0xffffu, d1 = o1 - Bf;
                  const uint32_t inv = (uint32_t)__builtin_amdgcn_readlane((int)inv_lane, (int)d1);
                  const uint32_t k = (uint32_t)lane, r = k - ((k * inv) >> 16) * d1;
                  if (k < ((A >> 16) & 0x1ffu)) winb[(o1 + k) & WM] = winb[(Bf + r) & WM];
                } else {
                  const uint32_t o1 = A & 0xffffu;
                  copy_match(o1, (A >> 16) & 0x1ffu, o1 - Bf, ring_lo);
                }
              }
            }
            wpos += total;
            P += (uint64_t)adv;
            if ((P >> 3) > in_end + 16) { err = ST_IN; break; }
            if (lst == 1u) { eob = true; break; }
            if (wpos - flushed >= (uint32_t)FLUSH) flush(false);
            catch_up(P >> 3);
            if (T < 8u && ++small >= 2u) { skip_pass = 24; small = 0; }   // (no resynchronisation to speak of: rounds for a while)
            else if (lst == 0u && delta + adv - T * (uint32_t)SB < (uint32_t)(SB / 2)) {
              // the next pass: the lanes not taken move down by T, their positions by what this pass consumed
              if (T >= 8u) small = 0;
              const int src = ((lane + (int)T) & 63) << 2;
              start = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)start) - adv;
              x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)x) - adv;
              c = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)c);
              m = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)m);
              st = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)st);
              valid = __builtin_amdgcn_ds_bpermute(src, (int)valid) != 0 && (uint32_t)lane + T < 64u;
              // (the pieces that enter: guessed in the last walk by the lane that is now lane - (64 - T))
              const int glane = lane + 2 * (int)T - 64, gsrc = (glane & 63) << 2;
              const uint32_t g_start = (uint32_t)__builtin_amdgcn_ds_bpermute(gsrc, (int)gstart), g_x = (uint32_t)__builtin_amdgcn_ds_bpermute(gsrc, (int)gx);
              const uint32_t g_st = (uint32_t)__builtin_amdgcn_ds_bpermute(gsrc, (int)gst);
              const bool g_valid = __builtin_amdgcn_ds_bpermute(gsrc, (int)gvalid) != 0;
              if (INF_GUESTS && (uint32_t)lane + T >= 64u && glane >= (int)T && glane < 64 && g_valid) {
                // (bit 2 of st: a walk that noted its exit only -- the lane walks again before it can agree, even from the same bit)
                start = g_start - adv; x = g_x - adv; c = 0; m = 0; st = g_st | 4u; valid = true;
              }
              delta = delta + adv - T * (uint32_t)SB;
              round_now = false;
            }
          }
        }
        if (!round_now) continue;
        valid = false;
        delta = 0;
        CNT(0, 1);
        // ---- a round: every lane decodes the symbol that would start at its bit offset, and where the next one starts
        const Sym s = symbol_at((uint32_t)P + (uint32_t)lane);
        const uint32_t kind = s.kind, outlen = s.outlen;
        // the chain of symbol starts from offset 0.  A symbol the tables cannot decode (kind 3) is on nobody's chain and
        // ends the chains that reach it; so do the end-of-block code and a symbol that ends beyond the 64 offsets --
        // after being visited.
        const uint32_t nextl = (uint32_t)lane + s.nbits;
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
        // where every symbol of the chain puts its output: a scan over the chain's lanes; the round takes the symbols
        // that start before ROUND_MAX bytes of it
        const bool onc0 = ((chain0 >> lane) & 1ull) != 0;
        const uint32_t x0 = onc0 ? outlen : 0u;
        const uint32_t incl = (uint32_t)wave_scan_add((int)x0);
        bool onc = onc0 && incl - x0 < (uint32_t)ROUND_MAX;
        unsigned long long chain = __ballot(onc);
        // RUNS (round 6).  A stretch of equal bytes -- absent qualities are 15,000 x 0xff per record, HiFi qualities sit at
        // their top value -- is a chain of matches of up to 258 bytes at distance 1, four or five bits each: a 288-bit piece
        // of them decodes to more than the ring holds, so they always come here, where ROUND_MAX let a round take two or
        // three of the ~14 that its 64 bit offsets hold, each copied by the wave through the ring (profiles/r06j_*: 89
        // rounds per member of a BAM without qualities).  Now the round also takes the distance-1 matches that FOLLOW what it
        // takes, as far as they go, and the run at its end -- those and the distance-1 matches it took anyway -- is written
        // as ONE fill: everything in front of it goes to HBM, the bytes of the fill straight to HBM as 16-byte stores,
        // and the ring gets its last WIN bytes.
        const unsigned long long runm = __ballot(onc0 && kind == 1u && s.dist == 1u);
        unsigned long long fillm = 0;
        if (chain) {
          const unsigned long long after = chain0 & ~chain;            // on the chain, beyond what the round takes (all above it)
          const unsigned long long stop = after & ~runm;                // the first of them that is no run ends the extension
          const unsigned long long ext = stop ? after & ((stop & (0ull - stop)) - 1ull) : after;
          const unsigned long long taken = chain | ext;
          const unsigned long long nonrun = taken & ~runm;
          const unsigned long long suffix = nonrun ? ((63 - (int)__builtin_clzll(nonrun)) == 63 ? 0ull : taken & ~((2ull << (63 - (int)__builtin_clzll(nonrun))) - 1ull)) : taken;
          if (suffix) {
            const int g0 = (int)__builtin_ctzll(suffix), g1 = 63 - (int)__builtin_clzll(suffix);
            const uint32_t flen = (uint32_t)__builtin_amdgcn_readlane((int)incl, g1) - ((uint32_t)__builtin_amdgcn_readlane((int)incl, g0) - (uint32_t)__builtin_amdgcn_readlane((int)x0, g0));
            if (ext || flen >= 2u * 258u) {     // (a short run that the round takes anyway: the ring's way)
              fillm = suffix;
              chain = taken;
              onc = ((taken >> lane) & 1ull) != 0;
            }
          }
        }
        uint32_t off = 0, total = 0;
        bool slow = false;
        if (chain) {
          const int h = 63 - (int)__builtin_clzll(chain);          // the last symbol taken
          total = __builtin_amdgcn_readlane(incl, h);
          off = __builtin_amdgcn_readlane(nextl, h);
          if (__builtin_amdgcn_readlane(kind, h) == 2u) eob = true;
          else if (off < 64u && total < (uint32_t)ROUND_MAX) slow = __builtin_amdgcn_readlane(kind, off) == 3u;
        } else slow = true;                                          // (the symbol at offset 0 itself)
        // ... and the run goes on behind the round's 64 bit offsets: its symbols are all the SAME bits (one length code, one
        // distance code, no extra bits that differ), so the lanes compare the nb bits at Q, Q + nb, Q + 2 nb, ... with the
        // round's last symbol -- 64 symbols (16 KB of output) at once; a prefix code makes "the same bits" the same symbol
        if (fillm && !eob && chain) {
          const int h = 63 - (int)__builtin_clzll(chain);
          const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)s.nbits, h), lh = (uint32_t)__builtin_amdgcn_readlane((int)s.val, h);
          if (((fillm >> h) & 1ull) && nb <= 16u && lh > 0u) {
            auto peek32 = [&](uint32_t pl) -> uint32_t {
              const uint32_t di = pl >> 5;
              return __builtin_amdgcn_alignbit(L.inb[(di + 1) & (INB / 4 - 1)], L.inb[di & (INB / 4 - 1)], pl & 31u);
            };
            const uint32_t mask = (1u << nb) - 1u;
            const uint32_t pat = peek32((uint32_t)P + (uint32_t)h) & mask;
            const bool same = (peek32((uint32_t)P + off + (uint32_t)lane * nb) & mask) == pat;
            const unsigned long long diff = ~__ballot(same);
            uint32_t rep = diff ? (uint32_t)__builtin_ctzll(diff) : 64u;
            const uint32_t room = wpos + total <= isize ? (isize - (wpos + total)) / lh : 0u;
            if (rep > room) rep = room;
            total += rep * lh;
            off += rep * nb;
          }
        }
        const uint32_t opos = wpos + (incl - x0);
        if (wpos + total > isize) { err = ST_OUT; break; }
        if (__ballot(onc && kind == 1u && s.dist > opos)) { err = ST_DIST; break; }
        if (onc && kind == 0u) winb[opos & WM] = (uint8_t)s.val;
        // the matches, in order
        unsigned long long mm = __ballot(onc && kind == 1u) & ~fillm;
        while (mm) {
          const int h = (int)__builtin_ctzll(mm);
          mm &= mm - 1;
          CNT(2, 1);
          const uint32_t o = __builtin_amdgcn_readlane(opos, h), ml = __builtin_amdgcn_readlane(s.val, h), dd = __builtin_amdgcn_readlane(s.dist, h);
          copy_match(o, ml, dd, near_lo(o));
        }
        if (fillm) {
          // the run at the round's end: [fo, wpos + total) all equal the byte in front of it
          CNT(11, 1);
          const uint32_t fo = (uint32_t)__builtin_amdgcn_readlane((int)opos, (int)__builtin_ctzll(fillm)), fend = wpos + total, flen = fend - fo;
          const uint32_t v = winb[(fo - 1u) & WM];
          const uint32_t v4 = v * 0x01010101u;
          // everything in front of it to HBM (a flush leaves fewer than four bytes behind: those as bytes)
          const uint32_t wkeep = wpos;
          wpos = fo;
          flush(false);
          if ((uint32_t)lane < fo - flushed) o8[flushed + (uint32_t)lane] = winb[(flushed + (uint32_t)lane) & WM];
          wpos = wkeep;
          // the fill, straight to HBM: bytes up to a 16-byte boundary, 16-byte stores, bytes
          {
            uint8_t* const d = o8 + fo;
            uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)d & 15u)) & 15u);
            if (head > flen) head = flen;
            if ((uint32_t)lane < head) d[lane] = (uint8_t)v;
            const uint32_t n16 = (flen - head) >> 4;
            uint4* const d16 = (uint4*)(d + head);
            const uint4 vv = make_uint4(v4, v4, v4, v4);
            for (uint32_t i = (uint32_t)lane; i < n16; i += 64u) d16[i] = vv;
            const uint32_t done = head + (n16 << 4);
            if ((uint32_t)lane < flen - done) d[done + (uint32_t)lane] = (uint8_t)v;
          }
          // the ring: the last min(flen, WIN) bytes of the fill
          {
            const uint32_t keep = flen < (uint32_t)WIN ? flen : (uint32_t)WIN;
            const uint32_t r0 = fend - keep;
            uint32_t head = (4u - (r0 & 3u)) & 3u;
            if (head > keep) head = keep;
            if ((uint32_t)lane < head) winb[(r0 + (uint32_t)lane) & WM] = (uint8_t)v;
            const uint32_t n4 = (keep - head) >> 2;
            for (uint32_t i = (uint32_t)lane; i < n4; i += 64u) L.win[((r0 + head + 4u * i) & WM) >> 2] = v4;
            const uint32_t done = head + (n4 << 2);
            if ((uint32_t)lane < keep - done) winb[(r0 + done + (uint32_t)lane) & WM] = (uint8_t)v;
          }
          flushed = fend;
        }
        wpos += total;
        P += (uint64_t)off;
        if ((P >> 3) > in_end + 16) err = ST_IN;
        if (err || eob) break;
        if (wpos - flushed >= (uint32_t)FLUSH) flush(false);
        if (!slow && off != 0) continue;
        // ---- one symbol through the bit buffer
        CNT(3, 1);
        reader_to_bit(P);
        const uint32_t E = decode(L.lit, LB, L.lcnt, L.lsym, LitEntry());
        const uint32_t k1 = E == ~0u ? 3u : (E >> 4) & 3u;
        if (k1 == 3u) { err = ST_CODE; break; }
        if (k1 == 0u) {
          if (wpos >= isize) { err = ST_OUT; break; }
          if (lane == 0) winb[wpos & WM] = (uint8_t)(E >> 9);
          ++wpos;
        } else if (k1 == 2u) {
          eob = true;
        } else {
          refill();
          const uint32_t len = ((E >> 9) & 511u) + take((int)((E >> 6) & 7u));
          const uint32_t D = decode(L.dst, DB, L.dcnt, L.dsym, DstEntry());
          if (D == ~0u || (D & 15u) == 0u) { err = ST_DIST; break; }
          refill();
          const uint32_t dist = (D >> 8) + take((int)((D >> 4) & 15u));
          if (dist > wpos) { err = ST_DIST; break; }
          if (wpos + len > isize) { err = ST_OUT; break; }
          copy_match(wpos, len, dist, near_lo(wpos));
          wpos += len;
        }
        P = in_addr * 8 - (uint64_t)bc;
        if (eob) break;
      }
      if (err) FAIL((int)err);
      reader_to_bit(P);   // the bit buffer again, behind the end-of-block code
      if (wpos - flushed >= (uint32_t)FLUSH) flush(false);
    } else {
      FAIL(ST_BTYPE);
    }
    if (bfinal) break;
  }
  if (wpos != isize) FAIL(ST_SIZE);
  flush(true);
  if (lane == 0) status[member] = ST_OK;
#ifdef INF_COUNT
  if (lane == 0) for (int k = 0; k < 12; ++k) atomicAdd(&g_inf_cnt[k], (unsigned long long)cnt[k]);
#endif
#undef FAIL
}

// One wavefront per member when the grid is as large as the batch (the default); a smaller grid walks the members with a
// stride: SVDSS_INFLATE_PER_CU=n launches at most n wavefronts per compute unit.  What for: ten members fit on a CU and
// take all of its LDS, so no block of the search / CRC / record kernels of the neighbouring batches (other streams) can
// start beside them -- in `search` end to end the kernel trace shows inflate and search kernels one after the other
// (0.39 s + 0.13 s of a 0.58 s stream).  Measured with n = 8 (37 KB left per CU): they do overlap then (0.42 s and 0.19 s
// in a 0.59 s stream) and each runs that much longer -- the sum of the GPU's work is what bounds the stream, not their
// order (profiles/r04q_e2e_kernel_overlap.txt).  Not the default.
__global__ void __launch_bounds__(64) bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const svdss_bgzf_block_t* __restrict__ blks,
                                                         uint8_t* out, int32_t* __restrict__ status, const int64_t n_members) {
  __shared__ Lds L;
  for (int64_t m = blockIdx.x; m < n_members; m += gridDim.x) inflate_member(L, comp, blks, out, status, m);
}

static unsigned inflate_grid(int64_t n_blocks) {
  static int n_cu = 0;
  static int per_cu = -1;
  if (per_cu < 0) {
    const char* e = getenv("SVDSS_INFLATE_PER_CU");
    per_cu = e ? atoi(e) : 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  const int64_t cap = per_cu > 0 ? (int64_t)per_cu * n_cu : n_blocks;
  return (unsigned)(n_blocks < cap ? n_blocks : cap);
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

hipError_t svdss_inflate_enqueue(hipStream_t st, const uint8_t* d_comp, const svdss_bgzf_block_t* d_blocks, int64_t n_blocks,
                                 uint8_t* d_out, int32_t* d_status) {
  if (n_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(inflate_grid(n_blocks)), dim3(64), 0, st, d_comp, d_blocks, d_out, d_status, n_blocks);
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
  // (the kernel reads the input in aligned 512-byte pieces, up to 5 KB past a block's last byte)
  if ((rc = ensure(o->comp, (size_t)comp_bytes + 8192))) return rc;
  if ((rc = ensure(o->blks, sizeof(svdss_bgzf_block_t) * (size_t)n_blocks))) return rc;
  if ((rc = ensure(o->status, sizeof(int32_t) * (size_t)n_blocks))) return rc;
  HIPCHK(hipMemcpyAsync(o->comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, o->st));
  HIPCHK(hipMemcpyAsync(o->blks.p, blocks, sizeof(svdss_bgzf_block_t) * (size_t)n_blocks, hipMemcpyHostToDevice, o->st));
  if (!o->ev0) { HIPCHK(hipEventCreate(&o->ev0)); HIPCHK(hipEventCreate(&o->ev1)); }
  HIPCHK(hipEventRecord(o->ev0, o->st));
  hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(inflate_grid(n_blocks)), dim3(64), 0, o->st, (const uint8_t*)o->comp.p,
                     (const svdss_bgzf_block_t*)o->blks.p, (uint8_t*)d_out, (int32_t*)o->status.p, n_blocks);
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
    unsigned long long h[12];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_inf_cnt), sizeof h) == hipSuccess) {
      fprintf(stderr, "[inflate] per block: deflate blocks %.2f; passes %.1f (lanes taken %.1f, walks %.1f, walk steps %.0f, matches %.0f of them one by one %.0f); "
                      "rounds %.0f (matches %.0f, run fills %.1f, one-symbol steps %.0f)\n",
              (double)h[4] / n_blocks, (double)h[5] / n_blocks, (double)h[6] / n_blocks, (double)h[7] / n_blocks, (double)h[8] / n_blocks,
              (double)h[9] / n_blocks, (double)h[10] / n_blocks, (double)h[0] / n_blocks, (double)h[2] / n_blocks, (double)h[11] / n_blocks, (double)h[3] / n_blocks);
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
