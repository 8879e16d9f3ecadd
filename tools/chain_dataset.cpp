// tools/chain_dataset.cpp -- BENCH TOOLING, not product: the synthetic data set of the end-to-end chain at the metric's
// scale (BASELINE.json config 4: whole-genome 30x HiFi, 15 kb reads, 0.5 % errors sub:ins:del = 2:1.5:1.5),
// written as the files run_svdss starts from: ref.fa, a coordinate-sorted reads.bam with truth alignments, its .bai, and
// the implanted SVs (truth.tsv).  The C++ successor of tools/e2e_call_wg.py::write_dataset (a Python loop per read: 22 s
// per 1.03 M substitution-only reads; a 30x set is 6.18 M reads with ~45 indels each in the CIGAR).
//
//   chain_dataset <workdir> <n_reads> <n_svs> [scale=1] [err=0.005] [threads=16] [codec=zlib1|zlib6|libdeflate1|libdeflate6] [ref=0]
//       ref: 0 = generate the reference and write <workdir>/ref.fa; 1 = generate it, leave an existing ref.fa alone (same seeds,
//       same file); 2 = READ <workdir>/ref.fa (any FASTA; the contigs and their order are the file's)
//
// Per contig c (GRCh38 primary lengths x scale): iid ACGT reference (seeded), n_svs * len / total SVs -- one per stretch,
// INS / DEL alternating, length U[50, 2000], every other pair heterozygous --, two haplotypes (all SVs / homozygous ones
// only), half of the contig's reads from each; a read = 15,000 bases of its haplotype from a uniform start, forward
// strand, with errors at rate err: 40 % substitutions, 30 % one-base insertions, 30 % one-base deletions; CIGAR = the truth
// (M / I / D of the errors and of the SVs; an insertion at a read's end is a soft clip, as an aligner would leave it);
// MAPQ 60, qualities absent (0xff), no tags (err = 0: XF:C:0, a "smoothed" set).  Records in BGZF members of at most
// 65,280 bytes; every 2,048 records begin a member (the chunks are built in parallel).
// Deterministic for a given command line whatever the thread count.
#include <dlfcn.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static const int64_t GRCH38_PRIMARY[24] = {248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                                           138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                                           83257441,  80373285,  58617616,  64444167,  46709983,  50818468,  156040895, 57227415};
static const int L = 15000;

struct Rng {   // xoshiro256**
  uint64_t s[4];
  static uint64_t sm(uint64_t& x) { uint64_t z = (x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
  explicit Rng(uint64_t seed) { for (auto& v : s) v = sm(seed); }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() { const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  int64_t below(int64_t n) { return (int64_t)(((unsigned __int128)next() * (unsigned __int128)n) >> 64); }
};
static uint64_t mix(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  uint64_t x = a * 0x9e3779b97f4a7c15ull ^ (b + 0x7f4a7c15ull) * 0xbf58476d1ce4e5b9ull ^ (c + 0x1ce4e5b9ull) * 0x94d049bb133111ebull ^ (d + 12345) * 0xd6e8feb86659fd93ull;
  return Rng::sm(x);
}

// ---------------------------------------------------------------- deflate back ends
struct Codec {
  bool use_ld = false; int level = 1;
  void* (*ld_alloc)(int) = nullptr; size_t (*ld_comp)(void*, const void*, size_t, void*, size_t) = nullptr;
  void (*ld_free)(void*) = nullptr;
};
static Codec g_codec;
struct Deflater {
  void* ld = nullptr; z_stream zs; bool z_init = false;
  Deflater() {
    if (g_codec.use_ld) ld = g_codec.ld_alloc(g_codec.level);
    else { memset(&zs, 0, sizeof zs); deflateInit2(&zs, g_codec.level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY); z_init = true; }
  }
  ~Deflater() { if (ld) g_codec.ld_free(ld); if (z_init) deflateEnd(&zs); }
  size_t run(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    if (ld) return g_codec.ld_comp(ld, in, n, out, cap);
    deflateReset(&zs);
    zs.next_in = (Bytef*)in; zs.avail_in = (uInt)n; zs.next_out = out; zs.avail_out = (uInt)cap;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return 0;
    return cap - zs.avail_out;
  }
};
// one BGZF member of data[0..n) appended to out; returns its size
static size_t bgzf_member(Deflater& d, const uint8_t* data, size_t n, std::vector<uint8_t>& out) {
  const size_t at = out.size();
  out.resize(at + 18 + 65536 + 1024 + 8);
  uint8_t* p = out.data() + at;
  size_t c = d.run(data, n, p + 18, 65536 + 1024);
  if (c == 0 || c + 26 > 65536) {   // stored block (never for this data; kept for safety)
    p[18] = 1; const uint16_t len = (uint16_t)n, nlen = (uint16_t)~len; memcpy(p + 19, &len, 2); memcpy(p + 21, &nlen, 2); memcpy(p + 23, data, n); c = n + 5;
  }
  static const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
  memcpy(p, hdr, 16);
  const uint16_t bs = (uint16_t)(c + 25); memcpy(p + 16, &bs, 2);
  const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), data, (uInt)n), isz = (uint32_t)n;
  memcpy(p + 18 + c, &crc, 4); memcpy(p + 22 + c, &isz, 4);
  out.resize(at + c + 26);
  return c + 26;
}

static int reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (int)(4681 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (int)(585 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (int)(73 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (int)(9 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (int)(1 + (beg >> 26));
  return 0;
}

struct Seg { int64_t h0, h1; char kind; int64_t rs; };   // haplotype [h0, h1) = 'M': reference from rs; 'I': inserted bases
struct Sv { int tid; int64_t pos; bool ins; int len; bool het; };
struct Hap { std::vector<uint8_t> code; std::vector<Seg> segs; };   // code: nt 1..4 per base
struct Desc { int64_t pos; int64_t a; int32_t r; int8_t hap; };
struct ChunkOut { std::vector<uint8_t> bytes; std::vector<uint32_t> msize, misize; std::vector<int64_t> pos, end; std::vector<uint32_t> vm, vo; uint32_t last_m = 0, last_o = 0; uint64_t ops = 0; bool done = false; };

static const uint8_t CODE16[6] = {0, 1, 2, 4, 8, 15};

// reference position of the first aligned base of a read that starts at haplotype offset a (-1: none within L bases)
static int64_t first_pos(const Hap& h, int64_t a) {
  size_t j = (size_t)(std::upper_bound(h.segs.begin(), h.segs.end(), a, [](int64_t v, const Seg& s) { return v < s.h0; }) - h.segs.begin()) - 1;
  for (; j < h.segs.size() && h.segs[j].h0 < a + L; ++j)
    if (h.segs[j].kind == 'M') { const int64_t lo = std::max(a, h.segs[j].h0); if (lo < h.segs[j].h1) return h.segs[j].rs + (lo - h.segs[j].h0); }
  return -1;
}

// one BAM record (block_size included) appended to rec; returns the reference end
static int64_t build_record(const Hap& h, const Desc& d, int tid, double err, uint64_t seed, std::vector<uint8_t>& rec, std::vector<uint32_t>& cig, uint8_t* bases) {
  Rng rng(mix(seed, (uint64_t)tid, (uint64_t)d.hap, (uint64_t)d.r));
  cig.clear();
  int64_t last_r = -1, pos = -1;
  auto push = [&](int op, int64_t n) {
    if (n <= 0) return;
    if (!cig.empty() && (int)(cig.back() & 15) == op) cig.back() += (uint32_t)(n << 4);
    else cig.push_back((uint32_t)(n << 4) | (uint32_t)op);
  };
  size_t j = (size_t)(std::upper_bound(h.segs.begin(), h.segs.end(), d.a, [](int64_t v, const Seg& s) { return v < s.h0; }) - h.segs.begin()) - 1;
  const uint8_t* src = h.code.data();
  int out = 0;
  int64_t i = d.a;
  // ops of the haplotype bases [x, y) that go into the read
  auto take = [&](int64_t x, int64_t y) {
    while (x < y) {
      while (h.segs[j].h1 <= x) ++j;
      const Seg& s = h.segs[j];
      const int64_t n = std::min(y, s.h1) - x;
      if (s.kind == 'M') {
        const int64_t r0 = s.rs + (x - s.h0);
        if (pos < 0) pos = r0;
        else if (r0 > last_r) push(2, r0 - last_r);
        push(0, n);
        last_r = r0 + n;
      } else push(pos < 0 ? 4 : 1, n);
      x += n;
    }
  };
  auto gap = [&]() -> int64_t {   // bases until the next error: geometric
    if (err <= 0) return (int64_t)1 << 40;
    const double u = rng.unit();
    return (int64_t)(std::log1p(-u) / std::log1p(-err));
  };
  int64_t next_err = i + 2 + gap();
  while (out < L) {
    const int64_t run_end = std::min(next_err, i + (L - out));
    if (run_end > i) { memcpy(bases + out, src + i, (size_t)(run_end - i)); take(i, run_end); out += (int)(run_end - i); i = run_end; }
    if (out >= L - 2) {   // (no errors on the last two bases: the record ends on an aligned base)
      if (out < L) { next_err = i + L; continue; }
      break;
    }
    const double u = rng.unit();
    while (h.segs[j].h1 <= i) ++j;
    const bool in_m = h.segs[j].kind == 'M';
    if (u < 0.4) {   // substitution
      bases[out] = (uint8_t)(((src[i] - 1 + 1 + (int)rng.below(3)) & 3) + 1);
      take(i, i + 1); ++out; ++i;
    } else if (u < 0.7) {   // one inserted base behind this one
      bases[out] = src[i]; take(i, i + 1); ++out; ++i;
      bases[out++] = (uint8_t)(1 + rng.below(4));
      push(pos < 0 ? 4 : 1, 1);
    } else if (in_m && pos < 0) {   // (no deletion in front of the first aligned base: the record's position is the descriptor's)
      bases[out] = src[i]; take(i, i + 1); ++out; ++i;
    } else {   // this base is missing from the read
      if (in_m) {
        const Seg& s = h.segs[j];
        const int64_t r0 = s.rs + (i - s.h0);
        if (r0 > last_r) push(2, r0 - last_r);   // (an SV deletion right in front)
        push(2, 1);
        last_r = r0 + 1;
      }
      ++i;
    }
    next_err = i + 1 + gap();
  }
  while (!cig.empty() && (cig.back() & 15) == 2) { last_r -= cig.back() >> 4; cig.pop_back(); }
  if (!cig.empty() && (cig.back() & 15) == 1) cig.back() = (cig.back() & ~15u) | 4;
  if (pos != d.pos) { fprintf(stderr, "position mismatch\n"); abort(); }
  char name[32];
  const int l_name = snprintf(name, sizeof name, "r%02d_%d_%07d", tid, (int)d.hap, d.r) + 1;
  const int32_t n_cig = (int32_t)cig.size();
  const int tag_n = err <= 0 ? 4 : 0;
  const int32_t block = 32 + l_name + 4 * n_cig + (L + 1) / 2 + L + tag_n;
  const size_t at = rec.size();
  rec.resize(at + 4 + (size_t)block);
  uint8_t* p = rec.data() + at;
  const int32_t i32[9] = {block, tid, (int32_t)pos, 0, 0, L, -1, -1, 0};
  memcpy(p, &i32[0], 4); memcpy(p + 4, &i32[1], 4); memcpy(p + 8, &i32[2], 4);
  p[12] = (uint8_t)l_name; p[13] = 60;
  const uint16_t bin = (uint16_t)reg2bin(pos, last_r), nc = (uint16_t)n_cig, flag = 0;
  memcpy(p + 14, &bin, 2); memcpy(p + 16, &nc, 2); memcpy(p + 18, &flag, 2);
  memcpy(p + 20, &i32[5], 4); memcpy(p + 24, &i32[6], 4); memcpy(p + 28, &i32[7], 4); memcpy(p + 32, &i32[8], 4);
  uint8_t* q = p + 36;
  memcpy(q, name, (size_t)l_name); q += l_name;
  memcpy(q, cig.data(), 4 * (size_t)n_cig); q += 4 * n_cig;
  for (int k = 0; k < L; k += 2) *q++ = (uint8_t)((CODE16[bases[k]] << 4) | (k + 1 < L ? CODE16[bases[k + 1]] : 0));
  memset(q, 0xff, L); q += L;
  if (tag_n) memcpy(q, "XFC\0", 4);
  return last_r;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: chain_dataset <workdir> <n_reads> <n_svs> [scale] [err] [threads] [codec] [keep_ref]\n"); return 2; }
  const std::string work = argv[1];
  const int64_t n_reads = atoll(argv[2]), n_svs = atoll(argv[3]);
  const double scale = argc > 4 ? atof(argv[4]) : 1.0, err = argc > 5 ? atof(argv[5]) : 0.005;
  const int threads = std::max(1, argc > 6 ? atoi(argv[6]) : 16);
  const std::string codec = argc > 7 ? argv[7] : "zlib1";
  const int ref_mode = argc > 8 ? atoi(argv[8]) : 0;
  const bool keep_ref = ref_mode != 0;
  g_codec.level = codec.back() - '0';
  if (codec.rfind("libdeflate", 0) == 0) {
    void* h = dlopen("libdeflate.so.0", RTLD_NOW);
    if (!h) h = dlopen("libdeflate.so", RTLD_NOW);
    if (!h) { fprintf(stderr, "no libdeflate\n"); return 1; }
    g_codec.ld_alloc = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
    g_codec.ld_comp = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
    g_codec.ld_free = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
    if (!g_codec.ld_alloc || !g_codec.ld_comp || !g_codec.ld_free) { fprintf(stderr, "libdeflate lacks a symbol\n"); return 1; }
    g_codec.use_ld = true;
  }
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
  const std::string fa_path = work + "/ref.fa", bam_path = work + "/reads.bam";
  std::vector<int64_t> lens;
  std::vector<std::vector<uint8_t>> given;       // ref = 2: the contigs of the FASTA, nt 1..5
  std::vector<std::string> names;
  int64_t total = 0;
  if (ref_mode == 2) {
    FILE* f = fopen(fa_path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", fa_path.c_str()); return 1; }
    uint8_t map[256]; memset(map, 5, sizeof map);
    map[(int)'A'] = map[(int)'a'] = 1; map[(int)'C'] = map[(int)'c'] = 2; map[(int)'G'] = map[(int)'g'] = 3; map[(int)'T'] = map[(int)'t'] = 4;
    std::vector<char> buf(1 << 24);
    bool in_header = false;
    std::string hdr;
    for (size_t n; (n = fread(buf.data(), 1, buf.size(), f)) > 0;)
      for (size_t k = 0; k < n; ++k) {
        const char c = buf[k];
        if (in_header) { if (c == '\n') { in_header = false; names.push_back(hdr.substr(0, hdr.find_first_of(" \t"))); given.emplace_back(); } else hdr += c; }
        else if (c == '>') { in_header = true; hdr.clear(); }
        else if (c != '\n' && c != '\r' && !given.empty()) given.back().push_back(map[(uint8_t)c]);
      }
    fclose(f);
    for (auto& g : given) { lens.push_back((int64_t)g.size()); total += (int64_t)g.size(); }
    if (lens.empty()) { fprintf(stderr, "no contigs in %s\n", fa_path.c_str()); return 1; }
  } else {
    for (int i = 0; i < 24; ++i) { lens.push_back(std::max<int64_t>(200000, (int64_t)(GRCH38_PRIMARY[i] * scale))); total += lens.back(); names.push_back("c" + std::to_string(i + 1)); }
  }
  const int nctg = (int)lens.size();
  const bool write_fa = !(keep_ref && access(fa_path.c_str(), R_OK) == 0);
  FILE* fa = write_fa ? fopen(fa_path.c_str(), "wb") : nullptr;
  FILE* bam = fopen(bam_path.c_str(), "wb");
  if ((write_fa && !fa) || !bam) { fprintf(stderr, "cannot write into %s\n", work.c_str()); return 1; }
  setvbuf(bam, nullptr, _IOFBF, 8 << 20);
  // BAM header
  {
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int i = 0; i < nctg; ++i) text += "@SQ\tSN:" + names[i] + "\tLN:" + std::to_string(lens[i]) + "\n";
    std::vector<uint8_t> hdr = {'B', 'A', 'M', 1};
    auto put32 = [&](int32_t v) { const uint8_t* b = (const uint8_t*)&v; hdr.insert(hdr.end(), b, b + 4); };
    put32((int32_t)text.size()); hdr.insert(hdr.end(), text.begin(), text.end()); put32(nctg);
    for (int i = 0; i < nctg; ++i) { const std::string nm = names[i]; put32((int32_t)nm.size() + 1); hdr.insert(hdr.end(), nm.begin(), nm.end()); hdr.push_back(0); put32((int32_t)lens[i]); }
    Deflater d; std::vector<uint8_t> m; bgzf_member(d, hdr.data(), hdr.size(), m);
    fwrite(m.data(), 1, m.size(), bam);
  }
  int64_t file_off = ftell(bam);
  std::vector<uint8_t> bai = {'B', 'A', 'I', 1};
  auto bai32 = [&](int32_t v) { const uint8_t* b = (const uint8_t*)&v; bai.insert(bai.end(), b, b + 4); };
  auto bai64 = [&](uint64_t v) { const uint8_t* b = (const uint8_t*)&v; bai.insert(bai.end(), b, b + 8); };
  bai32(nctg);
  std::vector<Sv> svs;
  int64_t sv_left = n_svs, rd_left = n_reads, n_written = 0, n_cigar_ops = 0;
  static const char LUT[6] = {'N', 'A', 'C', 'G', 'T', 'N'};
  for (int tid = 0; tid < nctg; ++tid) {
    const int64_t ref_len = lens[tid];
    const int64_t ns = tid == nctg - 1 ? sv_left : llround((double)n_svs * ref_len / total), nr = tid == nctg - 1 ? rd_left : llround((double)n_reads * ref_len / total);
    sv_left -= ns; rd_left -= nr;
    const uint64_t seed = 1000 + (uint64_t)tid;
    // ---- reference (blocks of 1 M bases, seeded per block)
    std::vector<uint8_t> ref;
    if (ref_mode == 2) ref.swap(given[(size_t)tid]);
    else {
      ref.resize((size_t)ref_len);
      std::atomic<int64_t> nextb{0};
      const int64_t nb = (ref_len + (1 << 20) - 1) >> 20;
      std::vector<std::thread> th;
      for (int t = 0; t < threads; ++t) th.emplace_back([&]() {
        for (int64_t b; (b = nextb.fetch_add(1)) < nb;) {
          Rng r(mix(seed, 0xabcdef, (uint64_t)b, 7));
          const int64_t lo = b << 20, hi = std::min(ref_len, lo + (1 << 20));
          for (int64_t k = lo; k < hi;) { uint64_t v = r.next(); for (int q = 0; q < 32 && k < hi; ++q, ++k, v >>= 2) ref[(size_t)k] = (uint8_t)(1 + (v & 3)); }
        }
      });
      for (auto& t : th) t.join();
    }
    if (fa) {
      fprintf(fa, ">%s\n", names[(size_t)tid].c_str());
      std::vector<char> line(1 << 22);
      for (int64_t k = 0; k < ref_len; k += (int64_t)line.size()) { const size_t n = (size_t)std::min<int64_t>((int64_t)line.size(), ref_len - k); for (size_t q = 0; q < n; ++q) line[q] = LUT[ref[(size_t)k + q]]; fwrite(line.data(), 1, n, fa); }
      fputc('\n', fa);
    }
    // ---- SVs and the two haplotypes
    Rng rs(mix(seed, 0x5151, 1, 2));
    struct SvSeq { int64_t pos; bool ins; int len; bool het; std::vector<uint8_t> seq; };
    std::vector<SvSeq> cs;
    if (ns > 0) {
      const int64_t step = ref_len / ns;
      for (int64_t k = 0; k < ns; ++k) {
        SvSeq s; s.len = 50 + (int)rs.below(1951); s.pos = k * step + step / 4 + rs.below(std::max<int64_t>(1, step / 4)); s.ins = k % 2 == 0; s.het = (k / 2) % 2 == 0;
        if (s.ins) { s.seq.resize((size_t)s.len); for (auto& b : s.seq) b = (uint8_t)(1 + rs.below(4)); }
        if (s.pos + s.len + 1000 >= ref_len) continue;
        svs.push_back({tid, s.pos, s.ins, s.len, s.het});
        cs.push_back(std::move(s));
      }
    }
    Hap haps[2];
    for (int only_hom = 0; only_hom < 2; ++only_hom) {
      Hap& h = haps[only_hom];
      h.code.reserve((size_t)ref_len + 4096 * cs.size() / 2);
      int64_t last = 0, hh = 0;
      for (const auto& s : cs) {
        if (only_hom && s.het) continue;
        if (s.pos > last) { h.segs.push_back({hh, hh + s.pos - last, 'M', last}); h.code.insert(h.code.end(), ref.begin() + last, ref.begin() + s.pos); hh += s.pos - last; last = s.pos; }
        if (s.ins) { h.segs.push_back({hh, hh + s.len, 'I', s.pos}); h.code.insert(h.code.end(), s.seq.begin(), s.seq.end()); hh += s.len; }
        else last = s.pos + s.len;
      }
      h.segs.push_back({hh, hh + ref_len - last, 'M', last});
      h.code.insert(h.code.end(), ref.begin() + last, ref.end());
    }
    std::vector<uint8_t>().swap(ref);
    // ---- read descriptors, sorted by position
    std::vector<Desc> ds;
    ds.reserve((size_t)nr);
    for (int hi = 0; hi < 2; ++hi) {
      const int64_t n = nr / 2 + (hi == 0 ? nr % 2 : 0), room = (int64_t)haps[hi].code.size() - L - L / 10;
      Rng rr(mix(seed, 0x7777, (uint64_t)hi, 3));
      for (int64_t r = 0; r < n; ++r) {
        const int64_t a = rr.below(room);
        const int64_t p = first_pos(haps[hi], a);
        if (p >= 0) ds.push_back({p, a, (int32_t)r, (int8_t)hi});
      }
    }
    std::sort(ds.begin(), ds.end(), [](const Desc& x, const Desc& y) { return x.pos != y.pos ? x.pos < y.pos : x.hap != y.hap ? x.hap < y.hap : x.r < y.r; });
    // ---- records -> BGZF members, chunks of 2,048 records in parallel, written in order
    const int64_t CH = 2048, n_chunks = ((int64_t)ds.size() + CH - 1) / CH;
    std::vector<ChunkOut> outs((size_t)n_chunks);
    std::mutex mu; std::condition_variable cv;
    std::atomic<int64_t> next_chunk{0};
    int64_t written_chunks = 0;
    const int64_t WINDOW = 4 * threads;     // chunks in flight ahead of the writer
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([&]() {
      Deflater defl;
      std::vector<uint8_t> rec, bases((size_t)L + 8);
      std::vector<uint32_t> cig;
      for (;;) {
        const int64_t c = next_chunk.fetch_add(1);
        if (c >= n_chunks) return;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return c < written_chunks + WINDOW; }); }
        ChunkOut& o = outs[(size_t)c];
        rec.clear();
        std::vector<size_t> starts;
        const int64_t lo = c * CH, hi = std::min<int64_t>((int64_t)ds.size(), lo + CH);
        for (int64_t k = lo; k < hi; ++k) {
          starts.push_back(rec.size());
          const int64_t e = build_record(haps[ds[(size_t)k].hap], ds[(size_t)k], tid, err, seed, rec, cig, bases.data());
          o.pos.push_back(ds[(size_t)k].pos); o.end.push_back(e);
          o.ops += cig.size();
        }
        for (size_t s : starts) { o.vm.push_back((uint32_t)(s / 65280)); o.vo.push_back((uint32_t)(s % 65280)); }
        o.last_m = (uint32_t)(rec.size() / 65280); o.last_o = (uint32_t)(rec.size() % 65280);
        for (size_t at = 0; at < rec.size(); at += 65280) {
          const size_t n = std::min<size_t>(65280, rec.size() - at);
          o.msize.push_back((uint32_t)bgzf_member(defl, rec.data() + at, n, o.bytes)); o.misize.push_back((uint32_t)n);
        }
        { std::lock_guard<std::mutex> lk(mu); o.done = true; }
        cv.notify_all();
      }
    });
    // writer + BAI of this contig
    std::map<int, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> lin;
    for (int64_t c = 0; c < n_chunks; ++c) {
      ChunkOut& o = outs[(size_t)c];
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return o.done; }); }
      n_cigar_ops += (int64_t)o.ops;
      fwrite(o.bytes.data(), 1, o.bytes.size(), bam);
      std::vector<int64_t> mstart(o.msize.size() + 1, file_off);
      for (size_t m = 0; m < o.msize.size(); ++m) mstart[m + 1] = mstart[m] + o.msize[m];
      const size_t n = o.vm.size();
      auto voff = [&](size_t k) -> uint64_t {   // record k's start (k = n: the chunk's end); bgzf_tell's form at a member's end
        const uint32_t m = k < n ? o.vm[k] : o.last_m, off = k < n ? o.vo[k] : o.last_o;
        if (off == 0 && m > 0) return ((uint64_t)mstart[m - 1] << 16) | o.misize[m - 1];
        return ((uint64_t)mstart[m] << 16) | off;
      };
      for (size_t k = 0; k < n; ++k) {
        const uint64_t v0 = voff(k), v1 = voff(k + 1);
        auto& ch = bins[reg2bin(o.pos[k], o.end[k])];
        if (!ch.empty() && ch.back().second == v0) ch.back().second = v1; else ch.emplace_back(v0, v1);
        for (int64_t w = o.pos[k] >> 14; w <= (o.end[k] - 1) >> 14; ++w) { if ((int64_t)lin.size() <= w) lin.resize((size_t)w + 1, 0); if (!lin[(size_t)w]) lin[(size_t)w] = v0; }
      }
      file_off = mstart.back();
      n_written += (int64_t)n;
      std::vector<uint8_t>().swap(o.bytes);
      { std::lock_guard<std::mutex> lk(mu); written_chunks = c + 1; }
      cv.notify_all();
    }
    for (auto& t : th) t.join();
    bai32((int32_t)bins.size());
    for (auto& b : bins) { bai32(b.first); bai32((int32_t)b.second.size()); for (auto& c : b.second) { bai64(c.first); bai64(c.second); } }
    bai32((int32_t)lin.size());
    uint64_t lastv = 0;
    for (auto v : lin) { if (v) lastv = v; bai64(lastv); }
    fprintf(stderr, "[chain_dataset] %s: %lld reads at +%.1f s\n", names[(size_t)tid].c_str(), (long long)ds.size(), since());
  }
  { Deflater d; std::vector<uint8_t> m; bgzf_member(d, (const uint8_t*)"", 0, m); fwrite(m.data(), 1, m.size(), bam); }
  if (fclose(bam) != 0 || (fa && fclose(fa) != 0)) { fprintf(stderr, "write error\n"); return 1; }
  { FILE* f = fopen((bam_path + ".bai").c_str(), "wb"); if (!f || fwrite(bai.data(), 1, bai.size(), f) != bai.size() || fclose(f) != 0) { fprintf(stderr, "cannot write the .bai\n"); return 1; } }
  { FILE* f = fopen((work + "/truth.tsv").c_str(), "w"); if (!f) return 1; for (auto& s : svs) fprintf(f, "%d\t%lld\t%s\t%d\t%d\n", s.tid, (long long)s.pos, s.ins ? "INS" : "DEL", s.len, s.het ? 1 : 0); fclose(f); }
  printf("{\"reads\": %lld, \"svs\": %zu, \"reference_bp\": %lld, \"cigar_ops_per_read\": %.2f, \"codec\": \"%s\", \"generate_s\": %.2f}\n", (long long)n_written, svs.size(), (long long)total,
         n_written ? (double)n_cigar_ops / n_written : 0.0, codec.c_str(), since());
  return 0;
}
