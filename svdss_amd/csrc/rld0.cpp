// rld0.cpp -- ropebwt3's on-disk index format (`.fmd`, magic "RLD\3"): export and import.
//
// `SVDSS index` in the reference is ropebwt3's `build -d`, which dumps the BWT of all records and their reverse
// complements as an rld0 structure (/root/reference/main.cpp:34-37, run_svdss:142), and `SVDSS search` restores it
// with rb3_fmi_restore (ping_pong.cpp:245).  ropebwt3 (@0ea3919e) is not in the tree; this file restates the
// published rld0 format (rld0.c / rld0.h of ropebwt2 / ropebwt3 / fermi, version 3)  [UPSTREAM-UNVERIFIED]:
//
//   header   "RLD\3" | u32 asize << 16 | sbits | u64 k = data length in 64-bit words | u64 n_frames
//            | u64 mcnt[asize] (occurrences of every symbol)
//   data     k words, a sequence of small blocks of ssize = 2^sbits words.  A block starts with the symbol counts of
//            the PREVIOUS block -- asize + 1 values (total first), 16-bit each if the total is below 0x4000 (type 0,
//            2 words for asize = 6), 32-bit below 2^30 (type 1, 4 words), else 64-bit (type 2, 7 words); the type sits in the top two bits of the first
//            word -- followed by the runs of the BWT: Elias-delta code of the run length, then the symbol in
//            abits = ilog2(asize) + 1 bits, packed most-significant-bit first.  A code never crosses a block
//            boundary (the block is zero-padded); data is kept in 2^23-word pieces, the last block of a piece ends
//            one word early.  After the last run comes one more block header.
//   frames   n_frames x (asize + 1) words: rank checkpoints (block offset, symbol counts before it), one per
//            2^ibits BWT positions.
//
// Export writes the BWT of this library's index (the suffix-sorted text  contig $ revcomp $ ...: a valid BWT of the
// same string collection, sentinels ordered by what follows them).  Import decodes the runs, recovers the strings of
// the collection by walking LF from every sentinel row, checks that they come in reverse-complement pairs, and
// rebuilds this library's index from one string of every pair: the occurrence predicate `size != 0` of
// ping_pong.cpp:4-49 only depends on the set of strings.
#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/svdss_hip.h"
#include "index_host.h"
#include "rld0.h"

namespace {

constexpr int RLD_LBITS = 23;
constexpr int64_t RLD_LSIZE = (int64_t)1 << RLD_LBITS;
constexpr int RLD_IBITS_PLUS = 4;
constexpr int ASIZE = 6, ASIZE1 = 7, ABITS = 3, SBITS = 3, SSIZE = 1 << SBITS;
constexpr int OFFSET0[3] = {(ASIZE1 * 16 + 63) / 64, (ASIZE1 * 32 + 63) / 64, ASIZE1};   // header words of a type 0 / 1 / 2 block

inline int ilog2_64(uint64_t v) { return 63 - __builtin_clzll(v); }

// Elias delta code of x >= 1: value and width (rld_delta_enc1)
inline uint64_t delta_enc(uint64_t x, int* width) {
  const int y = ilog2_64(x);
  const int z = ilog2_64((uint64_t)y + 1);
  *width = (z << 1) + 1 + y;
  return (x ^ ((uint64_t)1 << y)) | ((uint64_t)(y + 1) << y);
}

// last usable word of the small block that starts at word `shead`
inline int64_t block_tail(int64_t shead) {
  return shead + SSIZE - (((shead + SSIZE) & (RLD_LSIZE - 1)) == 0 ? 2 : 1);
}

struct Encoder {
  std::vector<uint64_t> z;       // the data words
  uint64_t cnt[ASIZE1] = {0};    // running totals: [0] all symbols, [c + 1] symbol c
  uint64_t mcnt[ASIZE1] = {0};   // the same at the start of the current block
  int64_t shead = 0, p = 0, stail = 0;
  int r = 64;                    // free bits in word p
  int64_t run_l = 0;
  int run_c = -1;

  Encoder() {
    z.assign((size_t)SSIZE, 0);
    // the first block: a type-0 header of zeros
    p = shead + OFFSET0[0];
    stail = block_tail(shead);
  }
  void need(int64_t words) { if ((int64_t)z.size() < words) z.resize((size_t)words, 0); }
  void next_block() {
    shead += SSIZE;
    need(shead + SSIZE);
    int type;
    if (cnt[0] - mcnt[0] < 0x4000) {
      uint16_t h[ASIZE1];
      for (int i = 0; i < ASIZE1; ++i) h[i] = (uint16_t)(cnt[i] - mcnt[i]);
      memcpy(&z[(size_t)shead], h, sizeof h);
      type = 0;
    } else if (cnt[0] - mcnt[0] < 0x40000000) {
      uint32_t h[ASIZE1];
      for (int i = 0; i < ASIZE1; ++i) h[i] = (uint32_t)(cnt[i] - mcnt[i]);
      memcpy(&z[(size_t)shead], h, sizeof h);
      type = 1;
    } else {   // a run of 2^30 symbols or more: 64-bit counts (readers mask type-1 counts to 30 bits)
      for (int i = 0; i < ASIZE1; ++i) z[(size_t)shead + (size_t)i] = cnt[i] - mcnt[i];
      type = 2;
    }
    z[(size_t)shead] |= (uint64_t)type << 62;
    p = shead + OFFSET0[type];
    stail = block_tail(shead);
    r = 64;
    for (int i = 0; i < ASIZE1; ++i) mcnt[i] = cnt[i];
    if (p > stail) next_block();   // (a type-2 header in the short last block of a piece leaves no data word)
  }
  // Elias-delta codes of the run lengths below 1024 (nearly every run of a DNA BWT), value and width
  struct DeltaLut {
    uint32_t val[1024];
    uint8_t width[1024];
    DeltaLut() {
      val[0] = 0; width[0] = 0;
      for (int x = 1; x < 1024; ++x) { int w; val[x] = (uint32_t)delta_enc((uint64_t)x, &w); width[x] = (uint8_t)w; }
    }
  };
  void enc1(int64_t l, int c) {   // rld_enc1
    static const DeltaLut lut;
    int w;
    uint64_t x;
    if (l < 1024) { x = ((uint64_t)lut.val[l] << ABITS) | (uint64_t)c; w = lut.width[l] + ABITS; }
    else { x = (delta_enc((uint64_t)l, &w) << ABITS) | (uint64_t)c; w += ABITS; }
    if (w > r && p == stail) next_block();
    uint64_t* zp = &z[(size_t)p];
    if (w > r) {
      w -= r;
      zp[0] |= x >> w;
      ++p;
      r = 64 - w;
      zp[1] = x << r;
    } else {
      r -= w;
      zp[0] |= x << r;
    }
    cnt[0] += (uint64_t)l;
    cnt[c + 1] += (uint64_t)l;
  }
  void push(int64_t l, int c) {   // rld_enc: adjacent runs of one symbol merge
    if (l == 0) return;
    if (run_c != c) {
      if (run_l) enc1(run_l, run_c);
      run_l = l; run_c = c;
    } else run_l += l;
  }
  int64_t finish() {              // rld_enc_finish; returns the data length in words
    if (run_l) enc1(run_l, run_c);
    run_l = 0;
    next_block();
    return p;
  }
};

bool write_all(FILE* f, const void* p, size_t bytes) { return bytes == 0 || fwrite(p, 1, bytes, f) == bytes; }

}  // namespace

bool rld0_is_fmd(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char m[4] = {0, 0, 0, 0};
  const bool ok = fread(m, 1, 4, f) == 4 && memcmp(m, "RLD\3", 4) == 0;
  fclose(f);
  return ok;
}

// All maximal runs of bwt[0, n) through the encoder.  The encoder's cursor lives in locals here (the stores into the
// data words may alias its members, and the compiler would reload them after every one); the boundaries of 64 symbols
// at a time come from a vectorisable compare, so that the loop has no data-dependent branch per SYMBOL.
static void encode_runs(Encoder& e, const uint8_t* bwt, int64_t n) {
  if (n <= 0) return;
  static const Encoder::DeltaLut lut;
  int64_t p = e.p, stail = e.stail;
  int r = e.r;
  uint64_t cnt[ASIZE1];
  for (int i = 0; i < ASIZE1; ++i) cnt[i] = e.cnt[i];
  uint64_t* z = e.z.data();
  auto emit = [&](int64_t l, int c) {
    int w;
    uint64_t x;
    if (l < 1024) { x = ((uint64_t)lut.val[l] << ABITS) | (uint64_t)c; w = lut.width[l] + ABITS; }
    else { x = (delta_enc((uint64_t)l, &w) << ABITS) | (uint64_t)c; w += ABITS; }
    if (w > r && p == stail) {
      e.p = p; e.r = r;
      for (int i = 0; i < ASIZE1; ++i) e.cnt[i] = cnt[i];
      e.next_block();
      p = e.p; r = e.r; stail = e.stail; z = e.z.data();
    }
    if (w > r) {
      w -= r;
      z[p] |= x >> w;
      ++p;
      r = 64 - w;
      z[p] = x << r;
    } else {
      r -= w;
      z[p] |= x << r;
    }
    cnt[0] += (uint64_t)l;
    cnt[c + 1] += (uint64_t)l;
  };
  uint8_t c = bwt[0];
  int64_t start = 0;
  int64_t i = 1;
  for (; i + 64 <= n; i += 64) {
    uint64_t m = 0;
    for (int k = 0; k < 64; ++k)   // (vectorises: bit k = symbol i + k differs from the one before it)
      m |= (uint64_t)(bwt[i + k] != bwt[i + k - 1]) << k;
    while (m) {
      const int k = __builtin_ctzll(m);
      m &= m - 1;
      emit(i + k - start, c);
      c = bwt[i + k];
      start = i + k;
    }
  }
  for (; i < n; ++i) {
    const uint8_t d = bwt[i];
    if (d == c) continue;
    emit(i - start, c);
    c = d;
    start = i;
  }
  emit(n - start, c);
  e.p = p; e.r = r;
  for (int k = 0; k < ASIZE1; ++k) e.cnt[k] = cnt[k];
}

int rld0_write(const char* path, const uint8_t* bwt, int64_t n) {
  {
    // every symbol must be below ASIZE: a 6 or 7 would be written as a symbol the format does not have
    bool ok = true;
#pragma omp parallel for reduction(&& : ok) schedule(static)
    for (int64_t i = 0; i < n; ++i) ok = ok && bwt[i] < ASIZE;
    if (!ok) return SVDSS_EINVAL;
  }
  Encoder e;
  // (the data grows by a block whenever one fills up: reserve what a DNA BWT needs, ~5 bits per symbol)
  try { e.z.reserve((size_t)(n / 12 + 1024)); } catch (...) {}
  encode_runs(e, bwt, n);
  const int64_t k = e.finish();
  // rld_rank_index: one frame per 2^ibits positions
  const uint64_t total = e.cnt[0];
  const uint64_t n_blks = (uint64_t)k / SSIZE + 1;
  const int64_t last = (k >> SBITS) << SBITS;
  const int ibits = ilog2_64(std::max<uint64_t>(1, total / n_blks)) + RLD_IBITS_PLUS;
  const uint64_t n_frames = ((total + ((uint64_t)1 << ibits) - 1) >> ibits) + 1;
  std::vector<uint64_t> frame((size_t)(n_frames * ASIZE1), 0);
  {
    uint64_t cnt[ASIZE] = {0};
    uint64_t fk = 1;
    for (int64_t i = SSIZE; i <= last; i += SSIZE) {
      const uint64_t w0 = e.z[(size_t)i];
      if ((w0 >> 62) == 2) {
        for (int j = 1; j <= ASIZE; ++j) cnt[j - 1] += e.z[(size_t)i + (size_t)j];
      } else if (w0 >> 62) {
        uint32_t h[ASIZE1];
        memcpy(h, &e.z[(size_t)i], sizeof h);
        for (int j = 1; j <= ASIZE; ++j) cnt[j - 1] += h[j] & 0x3fffffffu;
      } else {
        uint16_t h[ASIZE1];
        memcpy(h, &e.z[(size_t)i], sizeof h);
        for (int j = 1; j <= ASIZE; ++j) cnt[j - 1] += h[j];
      }
      uint64_t sum = 0;
      for (int j = 0; j < ASIZE; ++j) sum += cnt[j];
      while (sum >= (fk << ibits)) ++fk;
      if (fk < n_frames) {
        const uint64_t x = fk * ASIZE1;
        frame[(size_t)x] = (uint64_t)i;
        for (int j = 0; j < ASIZE; ++j) frame[(size_t)(x + j + 1)] = cnt[j];
      }
    }
    for (uint64_t f = 1; f < n_frames; ++f) {   // frames no block starts in: the previous one
      const uint64_t x = f * ASIZE1;
      if (frame[(size_t)x] == 0)
        for (int j = 0; j <= ASIZE; ++j) frame[(size_t)(x + j)] = frame[(size_t)(x - ASIZE1 + j)];
    }
  }
  FILE* f = fopen(path, "wb");
  if (!f) return SVDSS_EIO;
  const uint32_t a = (uint32_t)ASIZE << 16 | (uint32_t)SBITS;
  const uint64_t k64 = (uint64_t)k;
  bool ok = write_all(f, "RLD\3", 4) && write_all(f, &a, 4) && write_all(f, &k64, 8) && write_all(f, &n_frames, 8) &&
            write_all(f, e.cnt + 1, 8 * ASIZE) && write_all(f, e.z.data(), 8 * (size_t)k) &&
            write_all(f, frame.data(), 8 * frame.size());
  ok = (fclose(f) == 0) && ok;
  return ok ? SVDSS_OK : SVDSS_EIO;
}

int rld0_header_counts(const char* path, uint64_t mcnt_out[6]) {
  FILE* f = fopen(path, "rb");
  if (!f) return SVDSS_EIO;
  char magic[4];
  uint32_t a = 0;
  uint64_t k = 0, n_frames = 0;
  const bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "RLD\3", 4) == 0 && fread(&a, 4, 1, f) == 1 &&
                  fread(&k, 8, 1, f) == 1 && fread(&n_frames, 8, 1, f) == 1 && (a >> 16) == (uint32_t)ASIZE &&
                  fread(mcnt_out, 8, ASIZE, f) == (size_t)ASIZE;
  fclose(f);
  return ok ? SVDSS_OK : SVDSS_EIO;
}

int rld0_read(const char* path, std::vector<uint8_t>& bwt) {
  FILE* f = fopen(path, "rb");
  if (!f) return SVDSS_EIO;
  char magic[4];
  uint32_t a = 0;
  uint64_t k = 0, n_frames = 0, mcnt[16];
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "RLD\3", 4) != 0 || fread(&a, 4, 1, f) != 1 ||
      fread(&k, 8, 1, f) != 1 || fread(&n_frames, 8, 1, f) != 1) { fclose(f); return SVDSS_EIO; }
  const int asize = (int)(a >> 16), sbits = (int)(a & 0xffff);
  if (asize != ASIZE || sbits < 2 || sbits > 16) { fclose(f); return SVDSS_EIO; }   // nt6 indexes only
  if (fread(mcnt, 8, (size_t)asize, f) != (size_t)asize) { fclose(f); return SVDSS_EIO; }
  // (a block is read up to its tail whatever k says: room for the whole last block, zeroed)
  const uint64_t ssz = (uint64_t)1 << sbits;
  {   // the data length the header states must be in the file (it is not trusted with an allocation otherwise)
    const off_t here = ftello(f);
    if (here < 0 || fseeko(f, 0, SEEK_END) != 0) { fclose(f); return SVDSS_EIO; }
    const off_t end = ftello(f);
    if (end < here || fseeko(f, here, SEEK_SET) != 0 || k > (uint64_t)(end - here) / 8) { fclose(f); return SVDSS_EIO; }
  }
  std::vector<uint64_t> z;
  try { z.resize((size_t)((k + ssz - 1) / ssz * ssz + ssz + 2), 0); } catch (...) { fclose(f); return SVDSS_ENOMEM; }
  if (k && fread(z.data(), 8, (size_t)k, f) != (size_t)k) { fclose(f); return SVDSS_EIO; }
  fclose(f);
  uint64_t total = 0;
  for (int c = 0; c < asize; ++c) {
    if (mcnt[c] > ((uint64_t)1 << 48) || total + mcnt[c] > ((uint64_t)1 << 48)) return SVDSS_EIO;
    total += mcnt[c];
  }
  const int ssize = 1 << sbits;
  const int off0[3] = {(ASIZE1 * 16 + 63) / 64, (ASIZE1 * 32 + 63) / 64, ASIZE1};
  // The runs are walked twice: first only counted -- the symbol counts of the header must be the decoded ones before
  // a byte is allocated for them (a damaged header would otherwise ask for terabytes) --, then written.
  auto walk = [&](auto&& emit) -> int {   // emit(symbol, run length, position); returns SVDSS_OK or SVDSS_EIO
    uint64_t out = 0;
    for (int64_t shead = 0; shead < (int64_t)k && out < total; shead += ssize) {
      const int type = (int)(z[(size_t)shead] >> 62);
      if (type > 2) return SVDSS_EIO;
      int64_t p = shead + off0[type];
      const int64_t stail = shead + ssize - (((shead + ssize) & (RLD_LSIZE - 1)) == 0 ? 2 : 1);
      int r = 64;
      while (p <= stail && out < total) {
        // the next 64 bits of the block, zero beyond its last word
        uint64_t x = r == 64 ? z[(size_t)p] : (z[(size_t)p] << (64 - r)) | (p != stail ? z[(size_t)p + 1] >> r : 0);
        if ((x >> 58) == 0) break;   // no code starts with six zeros: the rest of the block is padding
        const int zc = __builtin_clzll(x);
        const int y = (int)((x << zc) >> (64 - (zc + 1))) - 1;   // the gamma part holds y + 1 in zc + 1 bits
        int w = 2 * zc + 1;
        if (y < 0 || w + y + ABITS > 64) return SVDSS_EIO;
        const uint64_t l = y ? ((x << w) >> (64 - y)) | ((uint64_t)1 << y) : 1;
        w += y;
        const int c = (int)((x << w) >> (64 - ABITS));
        w += ABITS;
        if (c >= asize) break;
        if (l > total - out) return SVDSS_EIO;
        emit(c, l, out);
        out += l;
        if (r > w) r -= w;
        else { ++p; r = 64 + r - w; }
      }
    }
    return out == total ? SVDSS_OK : SVDSS_EIO;
  };
  {
    uint64_t seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int rc = walk([&](int c, uint64_t l, uint64_t) { seen[c & 7] += l; });
    if (rc != SVDSS_OK) return rc;
    for (int c = 0; c < asize; ++c)
      if (seen[c] != mcnt[c]) return SVDSS_EIO;
  }
  try { bwt.assign((size_t)total, 0); } catch (...) { return SVDSS_ENOMEM; }
  return walk([&](int c, uint64_t l, uint64_t at) { memset(&bwt[(size_t)at], c, (size_t)l); });
}

// rank blocks + acc + '$' rows of a BWT (the layout of fmd_layout.h); text and suffix array stay empty
int svdss_blocks_from_bwt(const uint8_t* bwt, int64_t n, int threads, svdss_index* ix) {
  if (threads < 1) threads = 1;
  ix->n = n;
  const int64_t nb = n / SVDSS_BLOCK_SYMS + 1;
  try { ix->blocks.assign((size_t)(4 * nb), svdss_u4{0, 0, 0, 0}); } catch (...) { return SVDSS_ENOMEM; }
  std::vector<int64_t> bc((size_t)(nb * 6), 0);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t s = b * SVDSS_BLOCK_SYMS, e = std::min(n, s + SVDSS_BLOCK_SYMS);
    int64_t* c = &bc[(size_t)(b * 6)];
    svdss_u4* q = &ix->blocks[(size_t)(4 * b)];
    for (int64_t i = s; i < e; ++i) {
      const uint8_t sym = bwt[(size_t)i];
      c[sym]++;
      const int j = (int)((i - s) >> 5), bit = (int)((i - s) & 31);
      if (sym >= 1 && sym <= 4) {
        const uint32_t code = sym - 1u;
        q[j].y |= (code & 1u) << bit;
        q[j].z |= ((code >> 1) & 1u) << bit;
      } else {
        q[j].w |= 1u << bit;
        if (sym == 5) q[j].y |= 1u << bit;
      }
    }
  }
  int64_t run[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t b = 0; b < nb; ++b) {
    for (int c = 1; c <= 4; ++c) {
      if (run[c] > (int64_t)0xffffffffLL) return SVDSS_ERANGE;
      ix->blocks[(size_t)(4 * b + (c - 1))].x = (uint32_t)run[c];
    }
    for (int c = 0; c < 6; ++c) run[c] += bc[(size_t)(b * 6 + c)];
  }
  ix->acc[0] = 0;
  for (int c = 0; c < 6; ++c) ix->acc[c + 1] = ix->acc[c] + run[c];
  ix->dollar.clear();
  for (int64_t i = 0; i < n; ++i)
    if (bwt[(size_t)i] == 0) ix->dollar.push_back(i);
  return SVDSS_OK;
}

// The strings of the collection a BWT stands for: row r < m (m = number of sentinels) is the suffix "$_r"; walking LF
// from it spells string r backwards until the sentinel in front of it comes up.
int rld0_strings_of_bwt(const uint8_t* bwt, int64_t n, int threads, std::vector<std::vector<uint8_t>>& out) {
  svdss_index tmp;
  const int rc = svdss_blocks_from_bwt(bwt, n, threads, &tmp);
  if (rc != SVDSS_OK) return rc;
  const int64_t m = tmp.acc[1];
  if (m <= 0 || m > n) return SVDSS_EIO;
  SvdssDevIndex v;
  v.blocks = tmp.blocks.data();
  v.dollar = tmp.dollar.data();
  v.n = n;
  v.n_dollar = (int32_t)tmp.dollar.size();
  v.k = 0; v.bs_after = 0; v.pad_ = 0; v.text = nullptr; v.sa = nullptr; v.table = nullptr;
  memcpy(v.acc, tmp.acc, sizeof v.acc);
  out.assign((size_t)m, std::vector<uint8_t>());
  int bad = 0;
#pragma omp parallel for num_threads(threads < 1 ? 1 : threads) schedule(dynamic, 1)
  for (int64_t r = 0; r < m; ++r) {
    std::vector<uint8_t>& s = out[(size_t)r];
    int64_t i = r;
    for (int64_t steps = 0; steps <= n; ++steps) {
      const int c = bwt[(size_t)i];
      if (c == 0) break;
      s.push_back((uint8_t)c);
      i = v.acc[c] + svdss_rank_in_block(v, v.blocks + 4 * (i >> SVDSS_BLOCK_SHIFT), c, i);
      if (steps == n) {
#pragma omp atomic write
        bad = 1;
      }
    }
    std::reverse(s.begin(), s.end());
  }
  return bad ? SVDSS_EIO : SVDSS_OK;
}

// one string of every reverse-complement pair; SVDSS_EIO if the collection is not closed under reverse complement
int rld0_pick_strands(std::vector<std::vector<uint8_t>>& strings, std::vector<int64_t>& picked) {
  const int64_t m = (int64_t)strings.size();
  std::vector<int64_t> order((size_t)m);
  for (int64_t i = 0; i < m; ++i) order[(size_t)i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return strings[(size_t)a] < strings[(size_t)b]; });
  std::vector<char> used((size_t)m, 0);
  picked.clear();
  for (int64_t i = 0; i < m; ++i) {   // in the collection's own order
    if (used[(size_t)i]) continue;
    const std::vector<uint8_t>& s = strings[(size_t)i];
    std::vector<uint8_t> rc(s.rbegin(), s.rend());
    for (uint8_t& c : rc) c = (uint8_t)svdss_comp(c);
    used[(size_t)i] = 1;
    // an unused string equal to the reverse complement
    auto lo = std::lower_bound(order.begin(), order.end(), rc, [&](int64_t a, const std::vector<uint8_t>& key) {
      return strings[(size_t)a] < key;
    });
    bool found = false;
    for (; lo != order.end() && strings[(size_t)*lo] == rc; ++lo)
      if (!used[(size_t)*lo]) { used[(size_t)*lo] = 1; found = true; break; }
    if (!found) return SVDSS_EIO;
    picked.push_back(i);
  }
  return SVDSS_OK;
}
