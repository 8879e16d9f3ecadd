// deflate.hip -- BGZF blocks DEFLATED on the GPU (svdss_bgzf_deflate): the writing half of the BAM stream.
//
// Stands where htslib's bgzf_write / deflate stand under sam_write1 in `SVDSS smooth` (/root/reference/smoother.cpp:441-494,
// hts_set_threads on the output file): the smoothed BAM is ~1.5 bytes per base of packed bases and qualities, cut into
// independent deflate streams of at most 0xff00 bytes, and deflating it at level 6 was half of what the host cores did
// while `smooth` streams.  SURVEY 8(f)3 names the encoder ("later GPU deflate").
//
// One wavefront per BGZF block.  The encoder is a level-1-class one made for this data: packed bases and quality
// bytes hold almost no repeats inside a 32 KB window, what compresses them is their symbol statistics -- so a block is
// coded with literals under dynamic Huffman codes rebuilt four times per block (the statistics of bases, qualities,
// names and tags differ; zlib re-derives its trees every ~16 K symbols for the same reason) and with ONE kind of match:
// RUNS (4, 8, ... 256 equal bytes, at distance 1; second session of round 5).  Quality strings are where a BAM has them -- absent
// qualities are 15,000 x 0xff per read, HiFi qualities sit at their top value for long stretches -- and as literals a
// run is a bit per byte at best: the smoothed BAM of the chain bench was 1.8 x its input, and this repo's own inflater
// (csrc/inflate.hip) reads a stream of one-bit codes at a quarter of its rate (a 288-bit piece then decodes to more
// bytes than the output ring holds, it falls back to rounds).  A run costs a length code, its extra bits and one bit
// for the distance; a lane finds the runs of its own slice alone (dwords whose bytes equal the byte in front of them,
// even across the slice's start), so the three passes over a slice (count, size, pack) see the same tokens without
// storing them.
// No general LZ77: BGZF members are independent 64 KB streams, and bases and qualities do not repeat inside one.
//   * the quarter block is staged in LDS (coalesced loads), every lane counts the bytes of its 1/64 of it (LDS atomics);
//   * the 286 symbols (literals, end of block, length codes) are ranked by (count, symbol) -- every lane ranks its symbols against
//     all others with broadcast LDS reads --, one lane runs the two-queue Huffman merge over the sorted leaves, turns
//     node depths into counts per length, folds lengths above 15 back the way zlib's gen_bitlen does, and hands the
//     lengths out longest-first to the rarest symbols; canonical codes, bit-reversed for deflate's LSB-first stream;
//   * the header is the plainest valid one: HLIT = 286, two distance codes of one bit each (only distance 1 is ever
//     used; two codes make the distance code complete, which is what zlib itself always writes and every inflater
//     accepts), the code-length alphabet with 4 bits for each of 0..15 (a complete code), every length sent as
//     itself: 1,226 bits per quarter block;
//   * the bit offset of every lane's first symbol is a wave scan over the lanes' code lengths; a lane packs its symbols
//     into 32-bit words and stores them -- the first and the last word, which it shares with its neighbours, with an
//     atomic OR on the zeroed output --; a quarter block that Huffman coding would not shrink is stored (BTYPE 00).
// The CRC32 / ISIZE footer is the caller's (the host has the bytes in its hands anyway).  Output: any inflater reads it
// (tests: zlib, Python's gzip on the whole file, and csrc/inflate.hip); it is NOT the byte stream zlib or libdeflate
// would have written for the same input.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/svdss_hip.h"
#include "hip_check.h"
#include "deflate_dev.h"

#define UNI(x) __builtin_amdgcn_readfirstlane(x)

namespace {

constexpr int SUBS = 4;                    // deflate blocks per BGZF block
constexpr int MAX_IN = 0xff00;             // BGZF's block size (htslib BGZF_BLOCK_SIZE)
constexpr int MAX_SUB = (MAX_IN + SUBS - 1) / SUBS;
constexpr int NSYM = 286;                  // literals, end of block, the 29 length codes
constexpr int NDIST = 2;                   // distance codes 0 (distance 1: the runs) and 1 (never used), one bit each
constexpr int HDR_BITS = 3 + 5 + 5 + 4 + 19 * 3 + (NSYM + NDIST) * 4;   // 1,226
constexpr int MAX_RUN = 256;               // runs are found a dword at a time: 4, 8, ... 256 bytes (deflate allows 258)

struct Lds {
  uint32_t in[MAX_SUB / 4 + 8];
  uint32_t freq[NSYM + 6];
  uint16_t order[NSYM + 6];    // symbols with a non-zero count, rarest first
  union {                      // (the codes are written when the merge's weights are dead: seven blocks per CU stay)
    uint32_t weight[2 * NSYM];
    uint32_t enc[NSYM + 6];    // reversed code | length << 16
  };
  uint16_t parent[2 * NSYM];
  uint8_t depth[2 * NSYM];
  uint8_t len[NSYM + 6];
  int32_t n_used;
};

// length 3..258 -> its length symbol (257..285), the number of extra bits and their value (RFC 1951 3.2.5)
__device__ __forceinline__ int len_symbol(int len, int& ebits, int& eval) {
  ebits = 0; eval = 0;
  const int l = len - 3;                      // (lengths here are 4 .. 256: symbol 285, length 258, is never needed)
  if (l < 8) return 257 + l;
  const int e = (31 - __builtin_clz((unsigned)l)) - 2;
  ebits = e;
  eval = l & ((1 << e) - 1);
  return 261 + 4 * e + ((l >> e) & 3);
}

// The tokens of the slice [c0, c1) of the staged bytes (c0 a multiple of 4), found a DWORD at a time: a dword whose four
// bytes all equal the byte in front of it continues a run -- runs are 4, 8, ... 256 bytes long, match(length) --, any
// other dword is four literals, lit(dword, 4); the last bytes of the quarter block that do not fill a dword are literals
// (lit(dword, 1..3)).  prev = the byte in front of the slice (-1: none -- the first byte of the member).  A function of
// the slice's bytes and prev alone: every pass over the slice sees the same tokens, and a lane finds its own without its
// neighbours (a run goes on across slices: the next lane's first dword equals this lane's last byte).  What a run loses
// at its ends (up to three bytes each, coded as literals) does not matter for runs worth coding.  Two byte-wise versions
// came first: a state machine over the bytes costs several divergent branches per byte, 2.3 x the literal-only kernel's
// time on data without a single run (profiles/r05w_deflate_runs_ab.txt); per dword the kernel is where it was.
template <class FL, class FM>
__device__ __forceinline__ void for_tokens(const uint32_t* in, int c0, int c1, int prev, FL&& lit, FM&& match) {
  if (c0 >= c1) return;
  const int nw = (c1 - c0) >> 2;
  int w = c0 >> 2, rl = 0;
  bool have_prev = prev >= 0;
  uint32_t sp = have_prev ? (uint32_t)prev * 0x01010101u : 0u;   // the byte in front, four times
  uint32_t nxt = in[w];
  for (int j = 0; j < nw; ++j) {
    const uint32_t cur = nxt;
    nxt = in[++w];                              // (in[] has words to spare behind the quarter block)
    const bool same = have_prev && cur == sp;
    if (same && rl < MAX_RUN) { rl += 4; continue; }
    if (rl) { match(rl); rl = 0; }
    if (same) { rl = 4; continue; }             // (a run longer than MAX_RUN goes on as the next match)
    lit(cur, 4);
    sp = (cur >> 24) * 0x01010101u;
    have_prev = true;
  }
  if (rl) match(rl);
  if ((c1 - c0) & 3) lit(nxt, (c1 - c0) & 3);
}

template <int CTRL, int RMASK>
__device__ __forceinline__ int dpp_add(int x) { return x + __builtin_amdgcn_update_dpp(0, x, CTRL, RMASK, 0xf, false); }
__device__ __forceinline__ int wave_scan_add(int x) {   // inclusive
  x = dpp_add<0x111, 0xf>(x);
  x = dpp_add<0x112, 0xf>(x);
  x = dpp_add<0x114, 0xf>(x);
  x = dpp_add<0x118, 0xf>(x);
  x = dpp_add<0x142, 0xa>(x);
  x = dpp_add<0x143, 0xc>(x);
  return x;
}

// `nbits` bits of v (LSB first) at bit position pos of the zeroed word array w; other lanes may own the other bits of
// the words touched
__device__ __forceinline__ void put_bits_atomic(uint32_t* w, int64_t pos, uint32_t v, int nbits) {
  if (nbits <= 0) return;
  const int sh = (int)(pos & 31);
  const uint64_t x = (uint64_t)v << sh;
  atomicOr(&w[pos >> 5], (uint32_t)x);
  if (sh + nbits > 32) atomicOr(&w[(pos >> 5) + 1], (uint32_t)(x >> 32));
}

// Huffman code lengths (1..15) of the symbols with a non-zero count; one lane, everything in LDS.
__device__ void build_lengths(Lds& S) {
  const int n = S.n_used;
  for (int s = 0; s < NSYM; ++s) S.len[s] = 0;
  if (n == 1) { S.len[S.order[0]] = 1; return; }        // (cannot happen: the end-of-block symbol is always counted)
  // two-queue merge: leaves 0..n-1 in ascending count, internal nodes n.. in creation order (their weights ascend too)
  for (int i = 0; i < n; ++i) S.weight[i] = S.freq[S.order[i]];
  int li = 0, ii = n, next = n;
  for (int k = 0; k < n - 1; ++k) {
    int a, b;
    if (li < n && (ii >= next || S.weight[li] <= S.weight[ii])) a = li++; else a = ii++;
    if (li < n && (ii >= next || S.weight[li] <= S.weight[ii])) b = li++; else b = ii++;
    S.weight[next] = S.weight[a] + S.weight[b];
    S.parent[a] = (uint16_t)next;
    S.parent[b] = (uint16_t)next;
    ++next;
  }
  int bl[40];
  for (int b = 0; b < 40; ++b) bl[b] = 0;
  S.depth[next - 1] = 0;
  int overflow = 0;
  for (int i = next - 2; i >= 0; --i) {
    int d = S.depth[S.parent[i]] + 1;
    if (d > 39) d = 39;
    S.depth[i] = (uint8_t)d;
    if (i < n) {
      if (d > 15) { ++overflow; d = 15; }
      ++bl[d];
    }
  }
  // lengths above 15 folded back (zlib trees.c gen_bitlen): take a leaf from the deepest level that has one above the
  // limit, make it an internal node with the overflowing leaf as its sibling
  while (overflow > 0) {
    int bits = 14;
    while (bl[bits] == 0) --bits;
    --bl[bits];
    bl[bits + 1] += 2;
    --bl[15];
    overflow -= 2;
  }
  // longest codes to the rarest symbols
  int i = 0;
  for (int bits = 15; bits >= 1; --bits)
    for (int c = bl[bits]; c > 0; --c) S.len[S.order[i++]] = (uint8_t)bits;
  // canonical codes (RFC 1951 3.2.2), stored reversed: deflate packs Huffman codes most significant bit first
  uint32_t next_code[16];
  uint32_t code = 0;
  bl[0] = 0;
  for (int bits = 1; bits <= 15; ++bits) {
    code = (code + (uint32_t)bl[bits - 1]) << 1;
    next_code[bits] = code;
  }
  for (int s = 0; s < NSYM; ++s) {
    const int l = S.len[s];
    if (!l) { S.enc[s] = 0; continue; }
    const uint32_t c = next_code[l]++;
    S.enc[s] = (__builtin_bitreverse32(c) >> (32 - l)) | ((uint32_t)l << 16);
  }
}

__global__ void __launch_bounds__(64) bgzf_deflate_kernel(const uint8_t* in, int64_t in_bytes, int32_t block_bytes,
                                                         uint8_t* out, int64_t out_stride, int32_t* out_len) {
  __shared__ Lds S;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t start = b * (int64_t)block_bytes;
  const int n = (int)(in_bytes - start < block_bytes ? in_bytes - start : block_bytes);
  uint8_t* const ob = out + b * out_stride;
  uint32_t* const ow = (uint32_t*)ob;          // (out and out_stride are multiples of 4; the region is zeroed)
  int64_t bitpos = 18 * 8;                       // the deflate stream starts behind the 18-byte BGZF header
  const int base = (n + SUBS - 1) / SUBS;
  int prev_last = -1;                            // the byte in front of the quarter block (runs go on across quarters)
  for (int s0 = 0; s0 < n; s0 += base) {
    const int m = UNI(n - s0 < base ? n - s0 : base);
    const bool final_sub = s0 + m >= n;
    const uint8_t* src = in + start + s0;
    // ---- stage + count
    for (int w = lane; w < (m + 3) / 4; w += 64) {
      uint32_t v;
      __builtin_memcpy(&v, src + 4 * w, 4);      // (the input buffer has 16 bytes of slack behind its end)
      S.in[w] = v;
    }
    for (int t = lane; t < NSYM + 6; t += 64) S.freq[t] = 0;
    __syncthreads();
    const uint8_t* sb = (const uint8_t*)S.in;
    const int cl = ((m + 63) / 64 + 3) & ~3;     // bytes per lane (whole dwords: for_tokens reads its slice by dwords)
    const int c0 = lane * cl, c1 = c0 + cl < m ? c0 + cl : m;
    const int prev = c0 > 0 ? (c0 <= m ? (int)sb[c0 - 1] : -1) : prev_last;
    for_tokens(S.in, c0, c1, prev,
               [&](uint32_t v, int nb) {
                 for (int k = 0; k < nb; ++k) atomicAdd(&S.freq[(v >> (8 * k)) & 0xffu], 1u);
               },
               [&](int len) {
                 int eb, ev;
                 atomicAdd(&S.freq[len_symbol(len, eb, ev)], 1u);
               });
    if (lane == 0) S.freq[256] = 1;
    __syncthreads();
    // ---- symbols ranked by (count, symbol), zero counts left out
    {
      int used = 0;
      for (int t0 = 0; t0 < NSYM; t0 += 64) {
        const int t = t0 + lane;
        const uint32_t f = t < NSYM ? S.freq[t] : 0;
        int rank = 0;
        if (f) {
          for (int u = 0; u < NSYM; ++u) {
            const uint32_t g = S.freq[u];
            rank += (g != 0 && (g < f || (g == f && u < t))) ? 1 : 0;
          }
          S.order[rank] = (uint16_t)t;
        }
        used += (int)__popcll(__ballot(f != 0));
      }
      if (lane == 0) S.n_used = used;
    }
    __syncthreads();
    if (lane == 0) build_lengths(S);
    __syncthreads();
    // ---- sizes
    int my_bits = 0;
    for_tokens(S.in, c0, c1, prev,
               [&](uint32_t v, int nb) {
                 for (int k = 0; k < nb; ++k) my_bits += (int)(S.enc[(v >> (8 * k)) & 0xffu] >> 16);
               },
               [&](int len) {
                 int eb, ev;
                 my_bits += (int)(S.enc[len_symbol(len, eb, ev)] >> 16) + eb + 1;   // (+ the one-bit distance code)
               });
    const int incl = wave_scan_add(my_bits);
    const int body_bits = __builtin_amdgcn_readlane(incl, 63);
    const int eob = (int)S.enc[256];
    const int total_bits = HDR_BITS + body_bits + (eob >> 16);
    if (total_bits > 8 * (m + 5)) {
      // ---- stored: 3 header bits, pad to a byte, LEN, NLEN, the bytes
      if (lane == 0) put_bits_atomic(ow, bitpos, final_sub ? 1u : 0u, 3);
      const int64_t byte0 = (bitpos + 3 + 7) >> 3;
      if (lane == 0) {
        ob[byte0] = (uint8_t)(m & 0xff); ob[byte0 + 1] = (uint8_t)(m >> 8);
        ob[byte0 + 2] = (uint8_t)(~m & 0xff); ob[byte0 + 3] = (uint8_t)((~m >> 8) & 0xff);
      }
      for (int i = lane; i < m; i += 64) ob[byte0 + 4 + i] = sb[i];
      bitpos = (byte0 + 4 + m) * 8;
    } else {
      // ---- header (lane 0; the words are shared with the neighbouring quarter blocks' bits)
      if (lane == 0) {
        int64_t p = bitpos;
        put_bits_atomic(ow, p, (final_sub ? 1u : 0u) | (2u << 1), 3); p += 3;
        put_bits_atomic(ow, p, (uint32_t)(NSYM - 257), 5); p += 5;  // HLIT: 286 codes
        put_bits_atomic(ow, p, (uint32_t)(NDIST - 1), 5); p += 5;   // HDIST: 2 codes
        put_bits_atomic(ow, p, 15u, 4); p += 4;                     // HCLEN: 19 code-length codes
        // order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15: the three run-length codes unused, 4 bits for 0..15
        for (int k = 0; k < 19; ++k) { put_bits_atomic(ow, p, k < 3 ? 0u : 4u, 3); p += 3; }
        // every length sent as itself: code-length symbol v has the 4-bit code v, packed most significant bit first
        for (int t = 0; t < NSYM + NDIST; ++t) {
          const uint32_t v = t < NSYM ? S.len[t] : 1u;               // (the two distance codes: one bit each)
          put_bits_atomic(ow, p, __builtin_bitreverse32(v) >> 28, 4);
          p += 4;
        }
      }
      // ---- body: every lane packs its symbols; the first and the last word are shared with the neighbours
      int64_t p = bitpos + HDR_BITS + (incl - my_bits);
      if (c1 > c0 || lane == 63) {
        uint64_t acc = 0;
        int have = (int)(p & 31);                 // bits of the current word below this lane's first bit: not ours
        int64_t w = p >> 5;
        bool first = true;
        auto flush = [&](bool last) {
          // a full word (or, with last, the partial one): the first / last word of the lane by atomic OR
          const uint32_t v = (uint32_t)acc;
          if (first || last) atomicOr(&ow[w], v); else ow[w] = v;
          first = false;
          acc >>= 32; have -= 32; ++w;
        };
        for_tokens(S.in, c0, c1, prev,
                   [&](uint32_t v, int nb) {
                     for (int k = 0; k < nb; ++k) {
                       const uint32_t e = S.enc[(v >> (8 * k)) & 0xffu];
                       acc |= (uint64_t)(e & 0xffffu) << have;
                       have += (int)(e >> 16);
                       if (have >= 32) flush(false);
                     }
                   },
                   [&](int len) {
                     int eb, ev;
                     const uint32_t e = S.enc[len_symbol(len, eb, ev)];
                     acc |= (uint64_t)(e & 0xffffu) << have;
                     have += (int)(e >> 16);
                     if (have >= 32) flush(false);
                     acc |= (uint64_t)(uint32_t)ev << have;   // the length's extra bits, then distance code 0 (one bit, 0)
                     have += eb + 1;
                     if (have >= 32) flush(false);
                   });
        if (lane == 63) {                         // end of block, behind the last lane's symbols (lane 63 may hold none)
          acc |= (uint64_t)((uint32_t)eob & 0xffffu) << have;
          have += eob >> 16;
          if (have >= 32) flush(false);
        }
        if (have > 0) flush(true);
      }
      bitpos += total_bits;
    }
    prev_last = (int)sb[m - 1];
    __syncthreads();
  }
  if (lane == 0) {
    const int clen = n > 0 ? (int)((bitpos - 18 * 8 + 7) >> 3) : 0;
    // BGZF header (SAM spec 4.1): gzip member with the BC extra field holding the block size - 1
    const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    for (int i = 0; i < 16; ++i) ob[i] = hdr[i];
    const int bsize = clen + 25;
    ob[16] = (uint8_t)(bsize & 0xff); ob[17] = (uint8_t)(bsize >> 8);
    out_len[b] = clen + 26;                       // + the 8 bytes of CRC32 / ISIZE the caller fills in
  }
}

// exclusive prefix sum of the members' lengths (one wavefront; a launch has a few thousand blocks at most)
__global__ void __launch_bounds__(64) deflate_offsets_kernel(const int32_t* len, int64_t nb, int64_t* off) {
  const int lane = threadIdx.x;
  int64_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += 64) {
    const int64_t b = b0 + lane;
    const int x = b < nb ? len[b] : 0;
    const int incl = wave_scan_add(x);
    if (b < nb) off[b] = carry + incl - x;
    carry += __builtin_amdgcn_readlane(incl, 63);
  }
  if (lane == 0) off[nb] = carry;
}

// the members back to back: block b's bytes from its stride slot to off[b] (dword loads, byte-granular destination)
__global__ void __launch_bounds__(256) deflate_compact_kernel(const uint8_t* strided, int64_t stride, const int32_t* len,
                                                             const int64_t* off, uint8_t* dense) {
  const int64_t b = blockIdx.x;
  const uint8_t* src = strided + b * stride;
  uint8_t* dst = dense + off[b];
  const int n = len[b];
  // head bytes until dst is dword-aligned, then dwords (src is dword-aligned at every multiple of 4: shift by the head)
  const int head = (int)((4 - ((uintptr_t)dst & 3)) & 3) < n ? (int)((4 - ((uintptr_t)dst & 3)) & 3) : n;
  if ((int)threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
  const int nd = (n - head) >> 2;
  const uint32_t* s32 = (const uint32_t*)src;
  uint32_t* d32 = (uint32_t*)(dst + head);
  for (int i = threadIdx.x; i < nd; i += 256) {
    const int byte = head + 4 * i;
    const uint32_t w0 = s32[byte >> 2], w1 = s32[(byte >> 2) + 1];
    d32[i] = __builtin_amdgcn_alignbyte(w1, w0, byte & 3);
  }
  const int done = head + 4 * nd;
  if ((int)threadIdx.x < n - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}

}  // namespace

int64_t svdss_deflate_stride(int32_t block_bytes) { return ((int64_t)block_bytes + 26 + 6 * SUBS + 4 + 63) & ~(int64_t)63; }

hipError_t svdss_deflate_enqueue(hipStream_t st, const uint8_t* d_in, int64_t in_bytes, int32_t block_bytes, uint8_t* d_out,
                                 int64_t stride, int32_t* d_len) {
  if (in_bytes <= 0 || block_bytes <= 0 || block_bytes > MAX_IN || stride < svdss_deflate_stride(block_bytes)) return hipErrorInvalidValue;
  const int64_t nb = (in_bytes + block_bytes - 1) / block_bytes;
  hipLaunchKernelGGL(bgzf_deflate_kernel, dim3((unsigned)nb), dim3(64), 0, st, d_in, in_bytes, block_bytes, d_out, stride, d_len);
  return hipGetLastError();
}

hipError_t svdss_deflate_compact_enqueue(hipStream_t st, const uint8_t* d_strided, int64_t stride, const int32_t* d_len, int64_t nb,
                                         int64_t* d_off, uint8_t* d_dense) {
  hipLaunchKernelGGL(deflate_offsets_kernel, dim3(1), dim3(64), 0, st, d_len, nb, d_off);
  hipLaunchKernelGGL(deflate_compact_kernel, dim3((unsigned)nb), dim3(256), 0, st, d_strided, stride, d_len, (const int64_t*)d_off, d_dense);
  return hipGetLastError();
}

struct svdss_deflate {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  uint8_t* d_in = nullptr; size_t in_cap = 0;
  uint8_t* d_out = nullptr; size_t out_cap = 0;
  int32_t* d_len = nullptr; size_t len_cap = 0;
  uint8_t* d_dense = nullptr; size_t dense_cap = 0;   // compact mode: the members back to back
  int64_t* d_off = nullptr; size_t off_cap = 0;
  double kernel_ms = 0;
};

extern "C" void svdss_deflate_free(svdss_deflate_t* o) {
  if (!o) return;
  if (o->device >= 0) (void)hipSetDevice(o->device);
  if (o->d_in) (void)hipFree(o->d_in);
  if (o->d_out) (void)hipFree(o->d_out);
  if (o->d_len) (void)hipFree(o->d_len);
  if (o->d_dense) (void)hipFree(o->d_dense);
  if (o->d_off) (void)hipFree(o->d_off);
  if (o->e0) (void)hipEventDestroy(o->e0);
  if (o->e1) (void)hipEventDestroy(o->e1);
  if (o->stream) (void)hipStreamDestroy(o->stream);
  delete o;
}

extern "C" double svdss_deflate_kernel_ms(const svdss_deflate_t* o) { return o ? o->kernel_ms : -1.0; }

extern "C" int svdss_bgzf_deflate(svdss_deflate_t** obj, int32_t device, const uint8_t* in, int64_t in_bytes,
                                  int32_t block_bytes, uint8_t* out, int64_t out_stride, int32_t* out_len) {
  if (!obj || !in || !out || !out_len || in_bytes <= 0 || block_bytes <= 0 || block_bytes > MAX_IN) return SVDSS_EINVAL;
  // out_stride 0: the members back to back in `out` (room for in_bytes + 64 per block), out_len gives their lengths
  const bool dense = out_stride == 0;
  if (dense) out_stride = ((int64_t)block_bytes + 26 + 6 * SUBS + 4 + 63) & ~(int64_t)63;
  // a stored quarter block costs 5 bytes (+ 1 of padding), header and footer 26
  if (out_stride < block_bytes + 26 + 6 * SUBS + 4 || (out_stride & 3)) return SVDSS_EINVAL;
  HIPCHK(hipSetDevice(device));
  svdss_deflate* o = *obj;
  if (!o) {
    o = new (std::nothrow) svdss_deflate();
    if (!o) return SVDSS_ENOMEM;
    o->device = device;
    *obj = o;
    HIPCHK(hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&o->e0));
    HIPCHK(hipEventCreate(&o->e1));
  } else if (o->device != device) return SVDSS_EINVAL;
  const int64_t nb = (in_bytes + block_bytes - 1) / block_bytes;
  const size_t in_need = (size_t)in_bytes + 64, out_need = (size_t)(nb * out_stride), len_need = (size_t)nb * 4;
  if (o->in_cap < in_need) {
    if (o->d_in) (void)hipFree(o->d_in);
    o->d_in = nullptr; o->in_cap = 0;
    HIPCHK(hipMalloc((void**)&o->d_in, in_need + in_need / 4));
    o->in_cap = in_need + in_need / 4;
  }
  if (o->out_cap < out_need) {
    if (o->d_out) (void)hipFree(o->d_out);
    o->d_out = nullptr; o->out_cap = 0;
    HIPCHK(hipMalloc((void**)&o->d_out, out_need + out_need / 4));
    o->out_cap = out_need + out_need / 4;
  }
  if (o->len_cap < len_need) {
    if (o->d_len) (void)hipFree(o->d_len);
    o->d_len = nullptr; o->len_cap = 0;
    HIPCHK(hipMalloc((void**)&o->d_len, len_need * 2 + 64));
    o->len_cap = len_need * 2 + 64;
  }
  HIPCHK(hipMemcpyAsync(o->d_in, in, (size_t)in_bytes, hipMemcpyHostToDevice, o->stream));
  HIPCHK(hipMemsetAsync(o->d_in + in_bytes, 0, 64, o->stream));
  HIPCHK(hipMemsetAsync(o->d_out, 0, out_need, o->stream));
  HIPCHK(hipEventRecord(o->e0, o->stream));
  hipLaunchKernelGGL(bgzf_deflate_kernel, dim3((unsigned)nb), dim3(64), 0, o->stream, o->d_in, in_bytes, block_bytes, o->d_out,
                     out_stride, o->d_len);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(o->e1, o->stream));
  if (dense) {
    if (o->dense_cap < out_need) {
      if (o->d_dense) (void)hipFree(o->d_dense);
      o->d_dense = nullptr; o->dense_cap = 0;
      HIPCHK(hipMalloc((void**)&o->d_dense, out_need + out_need / 4));
      o->dense_cap = out_need + out_need / 4;
    }
    if (o->off_cap < (size_t)(nb + 1) * 8) {
      if (o->d_off) (void)hipFree(o->d_off);
      o->d_off = nullptr; o->off_cap = 0;
      HIPCHK(hipMalloc((void**)&o->d_off, (size_t)(nb + 1) * 16 + 64));
      o->off_cap = (size_t)(nb + 1) * 16 + 64;
    }
    hipLaunchKernelGGL(deflate_offsets_kernel, dim3(1), dim3(64), 0, o->stream, o->d_len, nb, o->d_off);
    hipLaunchKernelGGL(deflate_compact_kernel, dim3((unsigned)nb), dim3(256), 0, o->stream, o->d_out, out_stride, o->d_len, o->d_off,
                       o->d_dense);
    HIPCHK(hipGetLastError());
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(out_len, o->d_len, len_need, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipMemcpyAsync(&total, o->d_off + nb, 8, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipMemcpyAsync(out, o->d_dense, (size_t)total, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
  } else {
    HIPCHK(hipMemcpyAsync(out, o->d_out, out_need, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipMemcpyAsync(out_len, o->d_len, len_need, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
  }
  float ms = 0;
  if (hipEventElapsedTime(&ms, o->e0, o->e1) == hipSuccess) o->kernel_ms = ms;
  return SVDSS_OK;
}
