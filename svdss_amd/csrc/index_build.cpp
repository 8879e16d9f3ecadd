// index_build.cpp -- host-side construction of the HBM FM-index layout.
//
// Stands where `SVDSS index` = ropebwt3 `main_build` stands in the reference
// (/root/reference/main.cpp:15-17,34-37): it turns the FASTA records into the
// index that `SVDSS search` restores (ping_pong.cpp:245).  The on-disk format
// is this repo's own (fmd_layout.h); rld0 import/export is SURVEY 8(f) item 1.
//
// This is the builder for machines without a GPU (and the fallback of index_gpu.hip, which builds the same
// index in HBM).  Suffix sorting: one parallel sort on 63-bit keys (first 21 symbols, 3 bits
// each) followed by prefix doubling restricted to the still-tied groups
// (Larsson-Sadakane style, Jacobi updates so groups can be refined in
// parallel).  On near-random DNA almost every suffix is a singleton after the
// key sort, so the doubling rounds touch only repeats.
#include <omp.h>
#include <parallel/algorithm>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <utility>
#include <vector>

#include "../../include/svdss_hip.h"
#include "index_host.h"

namespace {

constexpr int KEY_SYMS = 21;

template <class I>
struct KeyPos {
  uint64_t k;
  I p;
};

template <class I>
void suffix_array(const uint8_t* t, int64_t n, std::vector<I>& sa, int threads) {
  sa.resize((size_t)n);
  if (n == 0) return;
  std::vector<I> rank((size_t)n);
  std::vector<std::pair<int64_t, int64_t>> groups;  // [start,end) with size > 1
  {
    std::vector<KeyPos<I>> kp((size_t)n);
    const int64_t chunk = 1 << 20;
    const int64_t nchunks = (n + chunk - 1) / chunk;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int64_t c = 0; c < nchunks; ++c) {
      const int64_t s = c * chunk, e = std::min(n, s + chunk);
      uint64_t k = 0;
      for (int64_t i = std::min(n, e + KEY_SYMS) - 1; i >= s; --i) {
        k = (k >> 3) | ((uint64_t)t[i] << 60);
        if (i < e) { kp[(size_t)i].k = k; kp[(size_t)i].p = (I)i; }
      }
    }
    // the recurrence above starts mid-text for all chunks but the last: bits of
    // symbols beyond i+20 are shifted out, so keys are exact.
    omp_set_num_threads(threads);
    __gnu_parallel::sort(kp.begin(), kp.end(),
                         [](const KeyPos<I>& a, const KeyPos<I>& b) { return a.k < b.k; });
    int64_t i = 0;
    while (i < n) {
      int64_t j = i + 1;
      while (j < n && kp[(size_t)j].k == kp[(size_t)i].k) ++j;
      if (j - i > 1) groups.emplace_back(i, j);
      i = j;
    }
    // ranks = start index of the group (serial boundary scan above, parallel fill here)
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t x = 0; x < n; ++x) {
      sa[(size_t)x] = kp[(size_t)x].p;
      rank[(size_t)kp[(size_t)x].p] = (I)x;  // singleton default
    }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256)
    for (int64_t g = 0; g < (int64_t)groups.size(); ++g)
      for (int64_t x = groups[(size_t)g].first; x < groups[(size_t)g].second; ++x)
        rank[(size_t)sa[(size_t)x]] = (I)groups[(size_t)g].first;
  }

  std::vector<I> nr((size_t)n);
  int64_t h = KEY_SYMS;
  while (!groups.empty()) {
    const int64_t ng = (int64_t)groups.size();
    std::vector<std::vector<std::pair<int64_t, int64_t>>> next((size_t)threads);
#pragma omp parallel num_threads(threads)
    {
      std::vector<std::pair<int64_t, I>> buf;  // (rank of suffix p+h, p)
      auto& mine = next[(size_t)omp_get_thread_num()];
#pragma omp for schedule(dynamic, 64)
      for (int64_t g = 0; g < ng; ++g) {
        const int64_t s = groups[(size_t)g].first, e = groups[(size_t)g].second;
        buf.resize((size_t)(e - s));
        for (int64_t x = s; x < e; ++x) {
          const int64_t p = (int64_t)sa[(size_t)x];
          const int64_t q = p + h;
          buf[(size_t)(x - s)] = {q < n ? (int64_t)rank[(size_t)q] : -1, (I)p};
        }
        std::sort(buf.begin(), buf.end());
        int64_t sub = s;
        for (int64_t x = s; x < e; ++x) {
          if (x > s && buf[(size_t)(x - s)].first != buf[(size_t)(x - s - 1)].first) {
            if (x - sub > 1) mine.emplace_back(sub, x);
            sub = x;
          }
          sa[(size_t)x] = buf[(size_t)(x - s)].second;
          nr[(size_t)x] = (I)sub;
        }
        if (e - sub > 1) mine.emplace_back(sub, e);
      }
#pragma omp barrier
      // Jacobi update: ranks change only after every group was sorted with the old ranks
#pragma omp for schedule(dynamic, 64)
      for (int64_t g = 0; g < ng; ++g)
        for (int64_t x = groups[(size_t)g].first; x < groups[(size_t)g].second; ++x)
          rank[(size_t)sa[(size_t)x]] = nr[(size_t)x];
    }
    groups.clear();
    for (auto& v : next) groups.insert(groups.end(), v.begin(), v.end());
    h *= 2;
  }
}

}  // namespace

int svdss_index_build_host(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs,
                           int32_t threads, svdss_index* ix) {
  if (!contigs || !lens || n_contigs <= 0 || !ix) return SVDSS_EINVAL;
  if (threads < 1) threads = 1;
  int64_t n = 0;
  for (int i = 0; i < n_contigs; ++i) {
    if (lens[i] < 0) return SVDSS_EINVAL;
    n += 2 * (lens[i] + 1);
  }
  std::vector<uint8_t> t;
  try { t.resize((size_t)n); } catch (...) { return SVDSS_ENOMEM; }
  {
    int64_t o = 0, src = 0;
    for (int i = 0; i < n_contigs; ++i) {
      const int64_t l = lens[i];
      for (int64_t j = 0; j < l; ++j) {
        const uint8_t c = contigs[src + j];
        if (c < 1 || c > 5) return SVDSS_EINVAL;  // '$' cannot appear inside a record
        t[(size_t)(o + j)] = c;
      }
      o += l;
      t[(size_t)o++] = 0;
      for (int64_t j = l - 1; j >= 0; --j) t[(size_t)o++] = (uint8_t)svdss_comp(contigs[src + j]);
      t[(size_t)o++] = 0;
      src += l;
    }
  }
  std::vector<uint8_t> bwt;
  try {
    bwt.resize((size_t)n);
    const bool force64 = getenv("SVDSS_FORCE_SA64") != nullptr;  // test hook for the 64-bit path
    ix->sa_wide = !(n < (int64_t)0x7fffffff && !force64);
    if (!ix->sa_wide) {
      std::vector<int32_t> sa;
      suffix_array<int32_t>(t.data(), n, sa, threads);
      ix->sa32.resize((size_t)n);
      ix->sa64.clear();
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        const int64_t p = sa[(size_t)i];
        ix->sa32[(size_t)i] = (uint32_t)p;
        bwt[(size_t)i] = t[(size_t)(p == 0 ? n - 1 : p - 1)];
      }
    } else {
      std::vector<int64_t> sa;
      suffix_array<int64_t>(t.data(), n, sa, threads);
      ix->sa64.resize((size_t)n);
      ix->sa32.clear();
#pragma omp parallel for num_threads(threads) schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        const int64_t p = sa[(size_t)i];
        ix->sa64[(size_t)i] = (uint64_t)p;
        bwt[(size_t)i] = t[(size_t)(p == 0 ? n - 1 : p - 1)];
      }
    }
  } catch (...) { return SVDSS_ENOMEM; }
  ix->text.swap(t);

  ix->n = n;
  ix->n_contigs = n_contigs;
  const int64_t nb = n / SVDSS_BLOCK_SYMS + 1;
  try { ix->blocks.assign((size_t)(4 * nb), svdss_u4{0, 0, 0, 0}); } catch (...) { return SVDSS_ENOMEM; }
  // per-block symbol counts, then exclusive prefix over blocks
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

void svdss_index_decode_bwt(const svdss_index* ix, uint8_t* bwt) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < ix->n; ++i) {
    const svdss_u4& q = ix->blocks[(size_t)(4 * (i >> SVDSS_BLOCK_SHIFT) + ((i >> 5) & 3))];
    const int bit = (int)(i & 31);
    const uint32_t p0 = (q.y >> bit) & 1u, p1 = (q.z >> bit) & 1u, p2 = (q.w >> bit) & 1u;
    bwt[i] = p2 ? (p0 ? 5 : 0) : (uint8_t)(1u + (p1 << 1 | p0));
  }
}

namespace {
struct FileHeader {
  char magic[8];  // "SVDSSFM2"
  int64_t n;
  int64_t acc[7];
  int64_t n_blocks;
  int64_t n_dollar;
  int32_t n_contigs;
  int32_t block_syms;
  int32_t sa_wide;   // 1: 64-bit suffix array entries
  int32_t reserved;
};
}  // namespace

int svdss_index_save_host(const svdss_index* ix, const char* path) {
  FILE* f = fopen(path, "wb");
  if (!f) return SVDSS_EIO;
  FileHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "SVDSSFM2", 8);
  h.n = ix->n;
  memcpy(h.acc, ix->acc, sizeof h.acc);
  h.n_blocks = (int64_t)ix->blocks.size() / 4;
  h.n_dollar = (int64_t)ix->dollar.size();
  h.n_contigs = ix->n_contigs;
  h.block_syms = SVDSS_BLOCK_SYMS;
  h.sa_wide = ix->sa_wide ? 1 : 0;
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  ok = ok && fwrite(ix->blocks.data(), sizeof(svdss_u4), ix->blocks.size(), f) == ix->blocks.size();
  ok = ok && fwrite(ix->dollar.data(), sizeof(int64_t), ix->dollar.size(), f) == ix->dollar.size();
  ok = ok && fwrite(ix->text.data(), 1, ix->text.size(), f) == ix->text.size();
  if (ix->sa_wide)
    ok = ok && fwrite(ix->sa64.data(), sizeof(uint64_t), ix->sa64.size(), f) == ix->sa64.size();
  else
    ok = ok && fwrite(ix->sa32.data(), sizeof(uint32_t), ix->sa32.size(), f) == ix->sa32.size();
  ok = (fclose(f) == 0) && ok;
  return ok ? SVDSS_OK : SVDSS_EIO;
}

namespace {
struct RecHeader {
  char magic[8];  // "SVDSSRC1"
  int64_t n;      // BWT length of the index of these records: sum of 2 * (len + 1)
  int64_t acc[7];
  int64_t total;  // sum of the record lengths
  int32_t n_contigs;
  int32_t reserved;
};
}  // namespace

static bool file_holds(FILE* f, uint64_t bytes);

int svdss_index_save_records_host(const svdss_index* ix, const char* path) {
  RecHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "SVDSSRC1", 8);
  h.n = ix->n;
  memcpy(h.acc, ix->acc, sizeof h.acc);
  h.n_contigs = ix->n_contigs;
  std::vector<int64_t> lens;
  const uint8_t* rec = nullptr;
  std::vector<uint8_t> tmp;
  if (!ix->rec_lens.empty()) {
    lens = ix->rec_lens;
    rec = ix->records.data();
  } else {
    // the forward strands out of the text  r0 $ revcomp(r0) $ r1 $ ...
    if ((int64_t)ix->text.size() != ix->n) return SVDSS_EINVAL;
    try { tmp.reserve((size_t)(ix->n / 2)); } catch (...) { return SVDSS_ENOMEM; }
    int64_t pos = 0;
    for (int32_t i = 0; i < ix->n_contigs; ++i) {
      const void* z = memchr(ix->text.data() + pos, 0, (size_t)(ix->n - pos));
      if (!z) return SVDSS_EINVAL;
      const int64_t len = (const uint8_t*)z - (ix->text.data() + pos);
      tmp.insert(tmp.end(), ix->text.begin() + pos, ix->text.begin() + pos + len);
      lens.push_back(len);
      pos += 2 * (len + 1);
    }
    if (pos != ix->n) return SVDSS_EINVAL;
    rec = tmp.data();
  }
  for (int64_t l : lens) h.total += l;
  if ((int32_t)lens.size() != ix->n_contigs) return SVDSS_EINVAL;
  FILE* f = fopen(path, "wb");
  if (!f) return SVDSS_EIO;
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  ok = ok && fwrite(lens.data(), sizeof(int64_t), lens.size(), f) == lens.size();
  ok = ok && (h.total == 0 || fwrite(rec, 1, (size_t)h.total, f) == (size_t)h.total);
  ok = (fclose(f) == 0) && ok;
  return ok ? SVDSS_OK : SVDSS_EIO;
}

// Round 6: behind the records of a file svdss_index_save_records_host wrote, the rank blocks and the '$' rows -- the index
// as a rank structure alone ("SVDSSBK1": n_blocks, n_dollar, blocks, rows).  A `search` that has few reads to search (a
// smoothed BAM: the putative filter leaves 1 - 7 % of them) makes THAT resident in a second instead of sorting six
// billion suffixes for a text, a suffix array and a k-mer table it would hardly use (svdss_index_attach_blocks).  Older
// readers stop at the records; a file without the section restores as before.
int svdss_index_append_blocks_host(const svdss_index* ix, const char* path) {
  if (ix->blocks.empty() || (int64_t)ix->blocks.size() != 4 * (ix->n / SVDSS_BLOCK_SYMS + 1)) return SVDSS_EINVAL;
  FILE* f = fopen(path, "ab");
  if (!f) return SVDSS_EIO;
  const int64_t nb = ix->n / SVDSS_BLOCK_SYMS + 1, nd = (int64_t)ix->dollar.size();
  bool ok = fwrite("SVDSSBK1", 1, 8, f) == 8 && fwrite(&nb, 8, 1, f) == 1 && fwrite(&nd, 8, 1, f) == 1;
  ok = ok && fwrite(ix->blocks.data(), sizeof(svdss_u4), ix->blocks.size(), f) == ix->blocks.size();
  ok = ok && (nd == 0 || fwrite(ix->dollar.data(), 8, (size_t)nd, f) == (size_t)nd);
  ok = (fclose(f) == 0) && ok;
  return ok ? SVDSS_OK : SVDSS_EIO;
}

// the "SVDSSBK1" section of a records file into ix->blocks / ix->dollar (ix->n, acc, n_contigs as the records' header has
// them); SVDSS_EINVAL: the file has no such section
int svdss_index_load_blocks_host(const char* path, svdss_index* ix) {
  FILE* f = fopen(path, "rb");
  if (!f) return SVDSS_EIO;
  RecHeader h;
  if (fread(&h, sizeof h, 1, f) != 1) { fclose(f); return SVDSS_EIO; }
  if (memcmp(h.magic, "SVDSSRC1", 8) != 0) { fclose(f); return SVDSS_EINVAL; }       // (another layout: no such section)
  if (h.n < 0 || h.total < 0 || h.n_contigs <= 0 || h.total > ((int64_t)1 << 46) || h.n != 2 * (h.total + h.n_contigs)) { fclose(f); return SVDSS_EIO; }
  const off_t sec = (off_t)sizeof h + (off_t)h.n_contigs * 8 + (off_t)h.total;
  char magic[8];
  int64_t nb = 0, nd = 0;
  if (fseeko(f, sec, SEEK_SET) != 0 || fread(magic, 1, 8, f) != 8 || memcmp(magic, "SVDSSBK1", 8) != 0 || fread(&nb, 8, 1, f) != 1 || fread(&nd, 8, 1, f) != 1) {
    fclose(f);
    return SVDSS_EINVAL;
  }
  if (nb != h.n / SVDSS_BLOCK_SYMS + 1 || nd < 0 || nd > h.n || !file_holds(f, (uint64_t)nb * 64 + (uint64_t)nd * 8)) { fclose(f); return SVDSS_EIO; }
  try {
    ix->blocks.resize((size_t)(4 * nb));
    ix->dollar.resize((size_t)nd);
  } catch (...) { fclose(f); return SVDSS_ENOMEM; }
  const off_t at = ftello(f);
  const int fd = fileno(f);
  const int64_t total = nb * 64, piece = (int64_t)32 << 20, n_pieces = (total + piece - 1) / piece;
  bool good = at >= 0;
  uint8_t* dst = (uint8_t*)ix->blocks.data();
#pragma omp parallel for reduction(&& : good) schedule(dynamic, 1) num_threads(std::min(omp_get_max_threads(), 32))
  for (int64_t k = 0; k < n_pieces; ++k) {
    const int64_t a = k * piece, b = std::min(total, a + piece);
    int64_t got = a;
    while (got < b) {
      const ssize_t r = pread(fd, dst + got, (size_t)(b - got), at + (off_t)got);
      if (r <= 0) break;
      got += r;
    }
    good = good && got == b;
  }
  if (good && nd > 0) good = pread(fd, ix->dollar.data(), (size_t)nd * 8, at + (off_t)total) == (ssize_t)(nd * 8);
  fclose(f);
  if (!good) { ix->blocks.clear(); ix->dollar.clear(); return SVDSS_EIO; }
  for (int64_t i = 0; i < nd; ++i) if (ix->dollar[(size_t)i] < 0 || ix->dollar[(size_t)i] >= h.n || (i > 0 && ix->dollar[(size_t)i] <= ix->dollar[(size_t)i - 1])) {
    ix->blocks.clear(); ix->dollar.clear(); return SVDSS_EIO;
  }
  ix->n = h.n;
  memcpy(ix->acc, h.acc, sizeof h.acc);
  ix->n_contigs = h.n_contigs;
  ix->sa_wide = h.n >= (int64_t)0x7fffffff;
  return SVDSS_OK;
}

// true if the file has at least `bytes` more bytes: a header's sizes are believed only as far as the file goes
static bool file_holds(FILE* f, uint64_t bytes) {
  const off_t here = ftello(f);
  if (here < 0 || fseeko(f, 0, SEEK_END) != 0) return false;
  const off_t end = ftello(f);
  return fseeko(f, here, SEEK_SET) == 0 && end >= here && (uint64_t)(end - here) >= bytes;
}

int svdss_index_load_records_host(const char* path, svdss_index* ix) {
  FILE* f = fopen(path, "rb");
  if (!f) return SVDSS_EIO;
  RecHeader h;
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "SVDSSRC1", 8) != 0 || h.n < 0 || h.total < 0 ||
      h.n_contigs <= 0 || h.total > ((int64_t)1 << 46) || h.n != 2 * (h.total + h.n_contigs) ||
      !file_holds(f, (uint64_t)h.n_contigs * 8 + (uint64_t)h.total)) {
    fclose(f);
    return SVDSS_EIO;
  }
  try {
    ix->rec_lens.resize((size_t)h.n_contigs);
    ix->records.resize((size_t)h.total);
  } catch (...) { fclose(f); return SVDSS_ENOMEM; }
  bool ok = fread(ix->rec_lens.data(), sizeof(int64_t), ix->rec_lens.size(), f) == ix->rec_lens.size();
  const off_t rec_at = ftello(f);
  int64_t sum = 0;
  for (int64_t l : ix->rec_lens) { if (l < 0 || l > h.total) ok = false; else sum += l; }
  // The records (GRCh38: 3.1 GB) are read in pieces by all threads -- each piece read, its pages touched and its symbols
  // checked by one thread (one fread into a zero-filled vector: 0.9 s of every `SVDSS search`; now ~0.2 s).  The
  // records are nt6 symbols 1..5: anything else would index the builder's tables out of range.
  if (ok && sum == h.total && rec_at >= 0) {
    const int fd = fileno(f);
    const int64_t piece = (int64_t)32 << 20, n_pieces = (h.total + piece - 1) / piece;
    bool good = true;
    uint8_t* dst = ix->records.data();
    // (at most 32 threads: the page cache does not give more, and a box may show far more cores than its quota grants)
#pragma omp parallel for reduction(&& : good) schedule(dynamic, 1) num_threads(getenv("SVDSS_INDEX_SERIAL_READ") ? 1 : std::min(omp_get_max_threads(), 32))
    for (int64_t k = 0; k < n_pieces; ++k) {
      const int64_t a = k * piece, b = std::min(h.total, a + piece);
      int64_t got = a;
      while (got < b) {
        const ssize_t r = pread(fd, dst + got, (size_t)(b - got), rec_at + (off_t)got);
        if (r <= 0) break;
        got += r;
      }
      unsigned bad = got == b ? 0u : 1u;
      if (!bad)
        for (int64_t i = a; i < b; ++i) bad |= (unsigned)((uint8_t)(dst[i] - 1) > 4);   // (no early exit: vectorises)
      good = good && bad == 0;
    }
    ok = good;
  } else ok = false;
  fclose(f);
  if (!ok || sum != h.total) { ix->rec_lens.clear(); ix->records.clear(); return SVDSS_EIO; }
  ix->n = h.n;
  memcpy(ix->acc, h.acc, sizeof h.acc);
  ix->n_contigs = h.n_contigs;
  return SVDSS_OK;
}

int svdss_index_load_host(const char* path, svdss_index* ix) {
  FILE* f = fopen(path, "rb");
  if (!f) return SVDSS_EIO;
  FileHeader h;
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "SVDSSFM2", 8) != 0 ||
      h.block_syms != SVDSS_BLOCK_SYMS || h.n < 0 || h.n > ((int64_t)1 << 46) || h.n_blocks != h.n / SVDSS_BLOCK_SYMS + 1 ||
      h.n_dollar < 0 || h.n_dollar > h.n ||
      !file_holds(f, (uint64_t)h.n_blocks * 4 * sizeof(svdss_u4) + (uint64_t)h.n_dollar * 8 + (uint64_t)h.n * (h.sa_wide ? 9 : 5))) {
    fclose(f);
    return SVDSS_EIO;
  }
  ix->n = h.n;
  memcpy(ix->acc, h.acc, sizeof h.acc);
  ix->n_contigs = h.n_contigs;
  ix->sa_wide = h.sa_wide != 0;
  try {
    ix->blocks.resize((size_t)(4 * h.n_blocks));
    ix->dollar.resize((size_t)h.n_dollar);
    ix->text.resize((size_t)h.n);
    if (h.sa_wide) ix->sa64.resize((size_t)h.n); else ix->sa32.resize((size_t)h.n);
  } catch (...) { fclose(f); return SVDSS_ENOMEM; }
  bool ok = fread(ix->blocks.data(), sizeof(svdss_u4), ix->blocks.size(), f) == ix->blocks.size();
  ok = ok && fread(ix->dollar.data(), sizeof(int64_t), ix->dollar.size(), f) == ix->dollar.size();
  ok = ok && fread(ix->text.data(), 1, ix->text.size(), f) == ix->text.size();
  if (ix->sa_wide)
    ok = ok && fread(ix->sa64.data(), sizeof(uint64_t), ix->sa64.size(), f) == ix->sa64.size();
  else
    ok = ok && fread(ix->sa32.data(), sizeof(uint32_t), ix->sa32.size(), f) == ix->sa32.size();
  fclose(f);
  if (ok) {
    // what the kernels index with must be in range (a damaged file is refused here, not found out on the GPU): text
    // symbols, suffix-array entries, '$' rows, the symbol totals
    const int64_t n = h.n;
    bool in_range = ix->acc[0] == 0 && ix->acc[6] == n;
    for (int c = 0; c < 6 && in_range; ++c) in_range = ix->acc[c] <= ix->acc[c + 1];
    for (size_t i = 0; i < ix->dollar.size() && in_range; ++i)
      in_range = ix->dollar[i] >= 0 && ix->dollar[i] < n && (i == 0 || ix->dollar[i] > ix->dollar[i - 1]);
    bool cells = true;
#pragma omp parallel for reduction(&& : cells) schedule(static)
    for (int64_t i = 0; i < n; ++i)
      cells = cells && ix->text[(size_t)i] <= 5 && (ix->sa_wide ? ix->sa64[(size_t)i] < (uint64_t)n : (int64_t)ix->sa32[(size_t)i] < n);
    ok = in_range && cells;
  }
  return ok ? SVDSS_OK : SVDSS_EIO;
}
