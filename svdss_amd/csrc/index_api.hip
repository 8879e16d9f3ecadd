// index_api.hip -- the index side of the C-ABI (include/svdss_hip.h): build / save / load / import, residency
// (to_device, replicate), the k-mer table build, host-side accessors -- and the error strings and nt6 encoding every
// entry point shares.  Stands where ropebwt3's `main_build` and rb3_fmi_restore stand in the reference
// (/root/reference/main.cpp:34-37, ping_pong.cpp:245); the search kernels are in sfs_search.hip.
#include <hip/hip_runtime.h>

#include <omp.h>
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svdss_hip.h"
#include "fmd_layout.h"
#include "hip_check.h"
#include "index_host.h"
#include "rld0.h"
#include "sfs_core2.h"

// ------------------------------------------------------------------ errors

thread_local std::string g_svdss_hip_err;   // shared with call_dp.hip


extern "C" const char* svdss_strerror(int code) {
  switch (code) {
    case SVDSS_OK: return "ok";
    case SVDSS_EINVAL: return "invalid argument";
    case SVDSS_ENOMEM: return "out of memory";
    case SVDSS_EIO: return "i/o error or bad index file";
    case SVDSS_EHIP: return "HIP runtime error (is a GPU present?)";
    case SVDSS_ENODEV: return "index is not resident on a device";
    case SVDSS_ERANGE: return "input exceeds a layout limit";
    default: return "unknown error";
  }
}

extern "C" const char* svdss_last_hip_error(void) { return g_svdss_hip_err.c_str(); }

// --------------------------------------------------------------------- a1

extern "C" int svdss_nt6_encode(const char* seq, int64_t n, uint8_t* out) {
  if ((!seq || !out) && n > 0) return SVDSS_EINVAL;
  // seq_nt6_table, ping_pong.hpp:46-52: A/a=1 C/c=2 G/g=3 T/t=4, NUL=0, rest 5
  for (int64_t i = 0; i < n; ++i) {
    const unsigned char ch = (unsigned char)seq[i];
    uint8_t v = 5;
    switch (ch) {
      case 0: v = 0; break;
      case 'A': case 'a': v = 1; break;
      case 'C': case 'c': v = 2; break;
      case 'G': case 'g': v = 3; break;
      case 'T': case 't': v = 4; break;
      default: break;
    }
    out[i] = v;
  }
  return SVDSS_OK;
}

// ------------------------------------------------------------------ index

static void free_device_side(svdss_index* ix);

// the index of the given records into *ix: on the GPU (index_gpu.hip); the host builder -- same index -- takes over
// for SVDSS_INDEX_CPU=1 and for the texts the GPU builder refuses, not for a missing GPU (that is an error).  A GPU-built index stays resident where it was built (without a
// k-mer table): rank blocks, '$' rows and acc are on the host already, text and suffix array (5-9 n bytes) come down
// only if somebody asks for them (svdss_index_save, another device) -- `SVDSS index` never does.
static int build_into(svdss_index* ix, const uint8_t* contigs, const int64_t* lens, int32_t n_contigs, int32_t threads) {
  int rc = -1;
  if (!getenv("SVDSS_INDEX_CPU")) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
      // no GPU: an error, not a quiet run of the host builder (that one is for the texts the GPU builder refuses
      // -- below -- and for SVDSS_INDEX_CPU=1)
      (void)hipGetLastError();
      g_svdss_hip_err = "no GPU found: the index is built in HBM (SVDSS_INDEX_CPU=1 runs the host builder)";
      return SVDSS_EHIP;
    }
    rc = svdss_index_build_gpu(contigs, lens, n_contigs, dev, ix);
    if (rc == SVDSS_OK) return rc;
    if (rc > 0) return rc;   // bad input: the host builder would say the same
    free_device_side(ix);
  }
  return svdss_index_build_host(contigs, lens, n_contigs, threads, ix);
}

extern "C" int svdss_index_build(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs,
                                 int32_t threads, svdss_index_t** out) {
  if (!out) return SVDSS_EINVAL;
  svdss_index* ix = new (std::nothrow) svdss_index();
  if (!ix) return SVDSS_ENOMEM;
  const int rc = build_into(ix, contigs, lens, n_contigs, threads);
  if (rc != SVDSS_OK) { delete ix; return rc; }
  *out = ix;
  return SVDSS_OK;
}

// An index restored from a records file holds the records only; the layout is built when somebody needs it on the
// host (BWT, count, save).  Making it resident (svdss_index_to_device) builds it in HBM instead and never comes here.
static int materialize(const svdss_index* cix) {
  svdss_index* ix = const_cast<svdss_index*>(cix);
  // (several threads may hold the same handle -- the replicas of `--gpus N` are made concurrently: one of them builds)
  static std::mutex m;
  std::lock_guard<std::mutex> lk(m);
  if (!svdss_index_is_lazy(ix)) return SVDSS_OK;
  int threads = 1;
#ifdef _OPENMP
  threads = omp_get_max_threads();
#endif
  return build_into(ix, ix->records.data(), ix->rec_lens.data(), (int32_t)ix->rec_lens.size(), threads);
}

static int build_table(svdss_index* ix);
static size_t table_bytes_for(int64_t n);

extern "C" int svdss_index_build_device(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs,
                                        int32_t threads, int32_t device, svdss_index_t** out) {
  if (!out || device < 0) return SVDSS_EINVAL;
  svdss_index* ix = new (std::nothrow) svdss_index();
  if (!ix) return SVDSS_ENOMEM;
  // (the table's memory lent to the suffix sort, as the restore paths do: index_gpu.hip)
  size_t tb = 0;
  if (contigs && lens && n_contigs > 0) {
    int64_t n = 0;
    for (int32_t i = 0; i < n_contigs; ++i) n += 2 * (lens[i] + 1);
    // (the order is chosen from the free memory of the device the index goes to, not of the calling thread's current one)
    if (!getenv("SVDSS_INDEX_CPU")) (void)hipSetDevice(device);
    tb = table_bytes_for(n);
  }
  int rc = getenv("SVDSS_INDEX_CPU") ? -1 : svdss_index_build_gpu(contigs, lens, n_contigs, device, ix, false, tb);
  if (rc == SVDSS_OK) {
    rc = build_table(ix);
  } else if (rc < 0) {   // no GPU / no room / degenerate text: host builder, then upload
    free_device_side(ix);
    rc = svdss_index_build_host(contigs, lens, n_contigs, threads, ix);
    if (rc == SVDSS_OK) rc = svdss_index_to_device(ix, device);
  }
  if (rc != SVDSS_OK) { free_device_side(ix); delete ix; return rc; }
  *out = ix;
  return SVDSS_OK;
}

extern "C" int svdss_index_save(const svdss_index_t* ix, const char* path) {
  if (!ix || !path) return SVDSS_EINVAL;
  int rc = materialize(ix);
  if (rc != SVDSS_OK) return rc;
  // an index built in HBM keeps text and suffix array there until somebody needs them on the host
  rc = svdss_index_fetch_host(const_cast<svdss_index*>(ix));
  if (rc != SVDSS_OK) return rc;
  return svdss_index_save_host(ix, path);
}

extern "C" int svdss_index_save_records(const svdss_index_t* ix, const char* path) {
  if (!ix || !path) return SVDSS_EINVAL;
  if (ix->rec_lens.empty()) {   // the records come out of the text (n bytes; the suffix array stays where it is)
    const int rc = svdss_index_fetch_text(const_cast<svdss_index*>(ix));
    if (rc != SVDSS_OK) return rc;
  }
  return svdss_index_save_records_host(ix, path);
}

// The index as a rank structure ALONE (round 6): the rank blocks and '$' rows that `SVDSS index` leaves behind the records
// of its sidecar become this handle's host side, the records are dropped, and svdss_index_to_device then uploads 3 GB
// instead of sorting six billion suffixes -- no text, no suffix array, no k-mer table: every extension is one rank step
// (the reference's own cost, ~1 M reads/s at GRCh38 lengths against 8 - 24 M with the table), which is what a `search`
// with few reads to search wants (svdss_main.cpp).  path: the index as svdss_index_load takes it (an .fmd: its sidecar).
// SVDSS_EINVAL: no such section there (an older sidecar, a full-layout file, an imported .fmd): restore as before.
extern "C" int svdss_index_append_blocks(const svdss_index_t* ix, const char* path) {
  if (!ix || !path) return SVDSS_EINVAL;
  { const int rc = materialize(ix); if (rc != SVDSS_OK) return rc; }      // (an index that holds records only: built first)
  return svdss_index_append_blocks_host(ix, path);
}

extern "C" int svdss_index_attach_blocks(svdss_index_t* ix, const char* path) {
  if (!ix || !path) return SVDSS_EINVAL;
  if (ix->device >= 0) return SVDSS_EINVAL;          // (already resident)
  std::string file = path;
  if (rld0_is_fmd(path)) {
    file += ".svdss";
    struct stat a, c;
    if (getenv("SVDSS_INDEX_NO_CACHE") || stat(path, &a) != 0 || stat(file.c_str(), &c) != 0 || c.st_mtime < a.st_mtime) return SVDSS_EINVAL;
  }
  svdss_index tmp;
  const int rc = svdss_index_load_blocks_host(file.c_str(), &tmp);
  if (rc != SVDSS_OK) return rc;
  if (ix->n != 0 && (tmp.n != ix->n || memcmp(tmp.acc, ix->acc, sizeof tmp.acc) != 0)) return SVDSS_EINVAL;   // (not this index's)
  ix->n = tmp.n;
  memcpy(ix->acc, tmp.acc, sizeof ix->acc);
  ix->n_contigs = tmp.n_contigs;
  ix->sa_wide = tmp.sa_wide;
  ix->blocks.swap(tmp.blocks);
  ix->dollar.swap(tmp.dollar);
  decltype(ix->records)().swap(ix->records);
  ix->rec_lens.clear();
  ix->text.clear(); ix->sa32.clear(); ix->sa64.clear();
  return SVDSS_OK;
}

// the same as a handle of its own (svdss_index_load + svdss_index_attach_blocks without reading the records first)
extern "C" int svdss_index_load_blocks(const char* path, svdss_index_t** out) {
  if (!path || !out) return SVDSS_EINVAL;
  svdss_index* ix = new (std::nothrow) svdss_index();
  if (!ix) return SVDSS_ENOMEM;
  const int rc = svdss_index_attach_blocks(ix, path);
  if (rc != SVDSS_OK) { delete ix; return rc; }
  *out = ix;
  return SVDSS_OK;
}

extern "C" int svdss_index_save_fmd(const svdss_index_t* ix, const char* path) {
  if (!ix || !path) return SVDSS_EINVAL;
  { const int rc = materialize(ix); if (rc != SVDSS_OK) return rc; }
  std::vector<uint8_t> bwt;
  try { bwt.resize((size_t)ix->n); } catch (...) { return SVDSS_ENOMEM; }
  svdss_index_decode_bwt(ix, bwt.data());
  return rld0_write(path, bwt.data(), ix->n);
}

extern "C" int svdss_fmd_read_bwt(const char* path, uint8_t* bwt_out, int64_t cap, int64_t* n_out) {
  if (!path || !n_out) return SVDSS_EINVAL;
  std::vector<uint8_t> bwt;
  const int rc = rld0_read(path, bwt);
  if (rc != SVDSS_OK) return rc;
  *n_out = (int64_t)bwt.size();
  if (bwt_out) {
    if (cap < (int64_t)bwt.size()) return SVDSS_ERANGE;
    memcpy(bwt_out, bwt.data(), bwt.size());
  }
  return SVDSS_OK;
}

// an rld0 file (ropebwt3 / upstream `SVDSS index`): decode the BWT, recover the strings, rebuild this layout
static int import_fmd(const char* path, svdss_index_t** out) {
  std::vector<uint8_t> bwt;
  int rc = rld0_read(path, bwt);
  if (rc != SVDSS_OK) return rc;
  int threads = 1;
#ifdef _OPENMP
  threads = omp_get_max_threads();
#endif
  std::vector<std::vector<uint8_t>> strings;
  rc = rld0_strings_of_bwt(bwt.data(), (int64_t)bwt.size(), threads, strings);
  if (rc != SVDSS_OK) return rc;
  std::vector<uint8_t>().swap(bwt);
  std::vector<int64_t> picked;
  rc = rld0_pick_strands(strings, picked);
  if (rc != SVDSS_OK) {
    g_svdss_hip_err = "the .fmd is not an index of records AND their reverse complements (ropebwt3 build without -d?)";
    return rc;
  }
  std::vector<int64_t> lens;
  int64_t total = 0;
  for (int64_t i : picked) { lens.push_back((int64_t)strings[(size_t)i].size()); total += lens.back(); }
  std::vector<uint8_t> cat;
  try { cat.reserve((size_t)total); } catch (...) { return SVDSS_ENOMEM; }
  for (int64_t i : picked) {
    cat.insert(cat.end(), strings[(size_t)i].begin(), strings[(size_t)i].end());
    std::vector<uint8_t>().swap(strings[(size_t)i]);
  }
  strings.clear();
  return svdss_index_build(cat.data(), lens.data(), (int32_t)lens.size(), threads, out);
}

extern "C" int svdss_index_load(const char* path, svdss_index_t** out) {
  if (!path || !out) return SVDSS_EINVAL;
  std::string file = path;
  bool is_cache = false;
  uint64_t mcnt[6] = {0, 0, 0, 0, 0, 0};
  if (rld0_is_fmd(path)) {
    // `SVDSS index` leaves this library's own layout beside the .fmd it writes: restoring that is a plain read.  The
    // cache must belong to THIS .fmd: not older, and with the symbol counts of the .fmd's header (an .fmd replaced
    // under a preserved mtime, or an upstream-built one for another reference, is imported instead).
    const std::string cache = file + ".svdss";
    struct stat a, c;
    if (!getenv("SVDSS_INDEX_NO_CACHE") && stat(path, &a) == 0 && stat(cache.c_str(), &c) == 0 &&
        c.st_mtime >= a.st_mtime && rld0_header_counts(path, mcnt) == SVDSS_OK) {
      file = cache;
      is_cache = true;
    } else
      return import_fmd(path, out);
  } else {
    // Neither rld0 nor this library's own layout: most likely ropebwt3's other format (.fmr, mrope -- what `ropebwt3
    // build` writes without -d; rb3_fmi_restore of ping_pong.cpp:245 falls back to it when the rld0 restore fails).
    // It is not read here; say so instead of a bare I/O error.  (Its magic is not restated from memory: the file is
    // recognised by what it is not.)
    FILE* f = fopen(path, "rb");
    char m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool readable = f && fread(m, 1, 8, f) == 8;
    if (f) fclose(f);
    if (readable && memcmp(m, "SVDSSFM2", 8) != 0 && memcmp(m, "SVDSSRC1", 8) != 0) {
      g_svdss_hip_err = "not an rld0 .fmd (magic RLD\\3) and not this library's index layout; if it is an .fmr (mrope) "
                        "index, convert it with `ropebwt3 build -i in.fmr -do out.fmd` or rebuild with `SVDSS index -d`";
      return SVDSS_EINVAL;
    }
  }
  svdss_index* ix = new (std::nothrow) svdss_index();
  if (!ix) return SVDSS_ENOMEM;
  // this library's files: the records file `SVDSS index` leaves beside the .fmd ("SVDSSRC1": the index is rebuilt from
  // it, in HBM when it is made resident) or the full layout svdss_index_save writes ("SVDSSFM2")
  char magic[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (FILE* f = fopen(file.c_str(), "rb")) { (void)!fread(magic, 1, 8, f); fclose(f); }
  int rc = memcmp(magic, "SVDSSRC1", 8) == 0 ? svdss_index_load_records_host(file.c_str(), ix)
                                             : svdss_index_load_host(file.c_str(), ix);
  if (rc == SVDSS_OK && is_cache) {
    bool same = true;
    for (int c = 0; c < 6; ++c) same = same && (uint64_t)(ix->acc[c + 1] - ix->acc[c]) == mcnt[c];
    if (!same) rc = SVDSS_EIO;
  }
  if (rc != SVDSS_OK) {
    delete ix;
    if (is_cache) return import_fmd(path, out);   // stale, truncated or foreign cache: the .fmd itself is the index
    return rc;
  }
  *out = ix;
  return SVDSS_OK;
}

static void free_device_side(svdss_index* ix) {
  if (ix->device < 0) return;
  (void)hipSetDevice(ix->device);
  for (void** p : {&ix->d_blocks, &ix->d_dollar, &ix->d_text, &ix->d_sa, &ix->d_table}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  ix->table_k = 0;
  ix->device = -1;
}

extern "C" void svdss_index_free(svdss_index_t* ix) {
  if (!ix) return;
  free_device_side(ix);
  delete ix;
}

extern "C" int64_t svdss_index_size(const svdss_index_t* ix) { return ix ? ix->n : -1; }

extern "C" int svdss_index_acc(const svdss_index_t* ix, int64_t acc[7]) {
  if (!ix || !acc) return SVDSS_EINVAL;
  memcpy(acc, ix->acc, sizeof ix->acc);
  return SVDSS_OK;
}

extern "C" int svdss_index_bwt(const svdss_index_t* ix, uint8_t* bwt_out) {
  if (!ix || !bwt_out) return SVDSS_EINVAL;
  { const int rc = materialize(ix); if (rc != SVDSS_OK) return rc; }
  svdss_index_decode_bwt(ix, bwt_out);
  return SVDSS_OK;
}

static std::atomic<int> g_kmer_limit{0};
extern "C" void svdss_index_kmer_limit(int32_t k) { g_kmer_limit.store(k < 0 ? 0 : k > 16 ? 16 : k); }

static int auto_kmer(int64_t n) {
  // K = floor(log4 n) + 3, at most 16: nearly every K-mer of the text is then unique and
  // nearly every other K-mer absent, so ONE lookup resolves a phase start (unique -> TEXT
  // mode, absent -> fail depth).  The table costs 4^K * 16 B (K=16: 64 GiB of the 288 GB
  // of HBM); it is shrunk until it fits in a third of the free device memory.
  int k = 1;
  while (k < 16 && ((int64_t)1 << (2 * k)) <= n) ++k;   // floor(log4 n) + 1
  k = k + 2 > 16 ? 16 : k + 2;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
    while (k > 4 && ((size_t)16 << (2 * k)) > free_b / 3) --k;
  if (const char* e = getenv("SVDSS_KMER")) k = atoi(e);
  if (k < 0) k = 0;
  if (k > 16) k = 16;
  return k;
}

extern "C" int64_t svdss_index_device_bytes(const svdss_index_t* ix) {
  if (!ix) return -1;
  const int k = ix->table_k;   // known once the index is resident
  const int64_t sa_bytes = ix->sa_wide ? 8 * ix->n : 4 * ix->n;
  return (int64_t)(ix->blocks.size() * sizeof(svdss_u4) + ix->dollar.size() * sizeof(int64_t)) +
         ix->n + 144 + sa_bytes + (k > 0 ? ((int64_t)16 << (2 * k)) : 0);
}

extern "C" int32_t svdss_index_kmer(const svdss_index_t* ix) { return ix ? ix->table_k : -1; }
extern "C" double svdss_index_deep_frac(const svdss_index_t* ix) { return ix ? ix->deep_frac : -1.0; }

SvdssDevIndex svdss_device_view(const svdss_index* ix) {
  SvdssDevIndex v;
  v.blocks = (const svdss_u4*)ix->d_blocks;
  v.dollar = (const int64_t*)ix->d_dollar;
  v.n = ix->n;
  v.n_dollar = (int32_t)ix->dollar.size();
  v.k = ix->table_k;
  v.bs_after = getenv("SVDSS_BS_AFTER") ? atoi(getenv("SVDSS_BS_AFTER")) : SV_BS_AFTER_DEFAULT;
  v.pad_ = 0;
  memcpy(v.acc, ix->acc, sizeof v.acc);
  v.text = ix->d_text ? (const uint8_t*)ix->d_text + 64 : nullptr;
  v.sa = ix->d_sa;
  v.table = (const SvdssTabEntry*)ix->d_table;
  return v;
}

// deep[0] / deep[1]: occurrences of the sampled K-mers (every 256th key) in all / of those with SV_BS_MIN or more of them
template <class P>
__global__ void __launch_bounds__(256) build_table_kernel(SvdssDevIndex ix, SvdssTabEntry* tab, int K, int forward, unsigned long long* deep) {
  const uint64_t nkeys = (uint64_t)1 << (2 * K);
  for (uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; key < nkeys;
       key += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t lo, info;
    sv_table_entry<P>(ix, (uint32_t)key, K, lo, info);
    if ((key & 255u) == 77u && (info >> 62) != SVDSS_TAB_EMPTY) {
      const uint64_t type = info >> 62;
      const unsigned long long occ = type == SVDSS_TAB_MULTI ? (unsigned long long)(info & SVDSS_TAB_MASK)
                                    : type == SVDSS_TAB_UNIQUE ? 1ull : (unsigned long long)((info >> 59) & 7u);
      atomicAdd(&deep[0], occ);
      if (type == SVDSS_TAB_MULTI && occ >= (unsigned long long)SV_BS_MIN) atomicAdd(&deep[1], occ);
    }
    if (!forward && (info >> 62) == SVDSS_TAB_EMPTY) info &= ~0xff00ull;   // SVDSS_TABLE_FORWARD=0 (A/B measurements)
    tab[key].lo = lo;
    tab[key].info = info;
  }
}

// bytes of the k-mer table an index of n symbols gets (0: none) -- the restore paths allocate it ahead of the suffix sort
static size_t table_bytes_for(int64_t n) {
  if (getenv("SVDSS_INDEX_NO_ARENA")) return 0;   // (developer knob: the sort's buffers and the table allocated one after the other, as until round 5)
  const int k = auto_kmer(n);
  return k > 0 ? (size_t)16 << (2 * k) : 0;
}

// the 4^K k-mer table of an index whose blocks, text and suffix array are resident
static int build_table(svdss_index* ix) {
  // (SVDSS_LF_ONLY=1, a measurement knob: the index as a rank structure alone -- no text, no suffix array, no table; every
  // extension is one rank step, the reference's own cost model)
  if (getenv("SVDSS_LF_ONLY")) {
    if (ix->d_text) { (void)hipFree(ix->d_text); ix->d_text = nullptr; }
    if (ix->d_sa) { (void)hipFree(ix->d_sa); ix->d_sa = nullptr; }
    if (ix->d_table) { (void)hipFree(ix->d_table); ix->d_table = nullptr; ix->d_table_cap = 0; }
    ix->table_k = 0;
  }
  const bool wide = ix->sa_wide;
  // The rank structure alone (no text, no suffix array: svdss_index_attach_blocks) still gets a SMALL k-mer table: its
  // entries need rank steps only (sv_table_entry: every K-mer that occurs becomes an interval, every absent one its fail
  // depth), it is built in a few hundredths of a second (K = 13: 67 M keys x 13 steps) and takes the first K rank steps of
  // every phase -- of ~18 at GRCh38 lengths -- off the search.  SVDSS_LF_KMER (0: none).
  const bool rank_only = !ix->d_text || !ix->d_sa;
  int k = auto_kmer(ix->n);
  if (rank_only) {
    int lk = getenv("SVDSS_LF_KMER") ? atoi(getenv("SVDSS_LF_KMER")) : 13;
    if (lk > 14) lk = 14;
    while (lk > 0 && ((int64_t)1 << (2 * lk)) > ix->n) --lk;       // (not more keys than symbols)
    k = lk < 0 ? 0 : lk;
    if (ix->d_table) { (void)hipFree(ix->d_table); ix->d_table = nullptr; ix->d_table_cap = 0; }
  }
  // Memory allocated ahead for this table (and lent to the suffix sort meanwhile, index_gpu.hip) was sized for the order
  // chosen THEN, with the device still empty: that order stands (asked again now, with the table's own bytes counted as
  // used, auto_kmer answers one less -- a 16 GiB table in a 64 GiB allocation and a search kernel 2.5 x slower).
  if (ix->d_table && ix->table_k == 0 && ix->d_table_cap >= 64) {
    int kc = 0;
    while (kc < 16 && ((size_t)16 << (2 * (kc + 1))) <= ix->d_table_cap) ++kc;
    k = kc;
  }
  // (a caller that has learnt meanwhile how little there is to search: svdss_index_kmer_limit)
  if (const int lim = g_kmer_limit.load(); lim > 0 && k > lim && !getenv("SVDSS_KMER") && !rank_only) k = lim;
  const bool ahead = ix->d_table && ix->table_k == 0 && k > 0 && ix->d_table_cap >= ((size_t)16 << (2 * k));
  if (ix->d_table && !ahead) { (void)hipFree(ix->d_table); ix->d_table = nullptr; }
  ix->table_k = 0;
  if (!ahead) ix->d_table_cap = 0;
  if (k > 0 && ((ix->d_text && ix->d_sa) || rank_only)) {
    const size_t tbytes = (size_t)16 << (2 * k);
    const bool verbose = getenv("SVDSS_INDEX_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (!ahead) HIPCHK(hipMalloc(&ix->d_table, tbytes));
    if (verbose) fprintf(stderr, "[index] k-mer table of order %d: %.1f GiB allocated in %.3f s\n", k, (double)tbytes / (1 << 30),
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    SvdssDevIndex v = svdss_device_view(ix);
    const uint64_t nkeys = (uint64_t)1 << (2 * k);
    const int blocks = (int)((nkeys + 255) / 256 < 65536 ? (nkeys + 255) / 256 : 65536);
    const char* fw = getenv("SVDSS_TABLE_FORWARD");
    const int forward = (fw && atoi(fw) == 0) ? 0 : 1;   // 0: entries without the forward-phase outcome
    unsigned long long* d_deep = nullptr;
    HIPCHK(hipMalloc((void**)&d_deep, 16));
    HIPCHK(hipMemset(d_deep, 0, 16));
    if (wide)
      hipLaunchKernelGGL(build_table_kernel<uint64_t>, dim3(blocks), dim3(256), 0, 0, v,
                         (SvdssTabEntry*)ix->d_table, k, forward, d_deep);
    else
      hipLaunchKernelGGL(build_table_kernel<uint32_t>, dim3(blocks), dim3(256), 0, 0, v,
                         (SvdssTabEntry*)ix->d_table, k, forward, d_deep);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    {
      unsigned long long h[2] = {0, 0};
      (void)hipMemcpy(h, d_deep, sizeof h, hipMemcpyDeviceToHost);
      (void)hipFree(d_deep);
      ix->deep_frac = h[0] ? (double)h[1] / (double)h[0] : 0.0;
      if (verbose) fprintf(stderr, "[index] %.1f %% of the sampled K-mer occurrences belong to K-mers with %d or more\n", 100.0 * ix->deep_frac, SV_BS_MIN);
    }
    if (verbose) fprintf(stderr, "[index] k-mer table filled at +%.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    ix->table_k = k;
  }
  return SVDSS_OK;
}

// the host copy of the rank blocks of an index just built with defer_host_blocks (four threads: index_gpu.hip), then
// its k-mer table
static int table_and_blocks(svdss_index* ix) {
  const int rc = svdss_index_fetch_blocks(ix);
  return rc != SVDSS_OK ? rc : build_table(ix);
}

extern "C" int svdss_index_to_device(svdss_index_t* ix, int32_t device) {
  if (!ix || device < 0) return SVDSS_EINVAL;
  HIPCHK(hipSetDevice(device));
  if (ix->device == device && ix->d_blocks && ix->d_text && ix->d_sa)   // built there (svdss_index_build_device)
    return ix->d_table && ix->table_k > 0 ? SVDSS_OK : build_table(ix);
  if (svdss_index_is_lazy(ix)) {
    // restored from a records file: the index is built where it is going to live (GRCh38 lengths: seconds; the host
    // builder + upload only when the device cannot)
    int rc = getenv("SVDSS_INDEX_CPU") ? -1 : svdss_index_build_gpu(ix->records.data(), ix->rec_lens.data(),
                                                                    (int32_t)ix->rec_lens.size(), device, ix, true, table_bytes_for(ix->n));
    if (rc == SVDSS_OK) return table_and_blocks(ix);
    if (rc > 0) return rc;
    free_device_side(ix);
    rc = materialize(ix);
    if (rc != SVDSS_OK) return rc;
    HIPCHK(hipSetDevice(device));
  }
  if (ix->device >= 0 && ix->d_text && ix->d_sa) {   // resident elsewhere: the host copy travels
    const int rc = svdss_index_fetch_host(ix);
    if (rc != SVDSS_OK) return rc;
    HIPCHK(hipSetDevice(device));
  }
  free_device_side(ix);
  ix->device = device;  // so that a failure below still frees what was allocated
  const size_t bb = ix->blocks.size() * sizeof(svdss_u4);
  const size_t db = (ix->dollar.size() + 1) * sizeof(int64_t);
  HIPCHK(hipMalloc(&ix->d_blocks, bb));
  HIPCHK(hipMalloc(&ix->d_dollar, db));
  HIPCHK(hipMemcpy(ix->d_blocks, ix->blocks.data(), bb, hipMemcpyHostToDevice));
  if (!ix->dollar.empty())
    HIPCHK(hipMemcpy(ix->d_dollar, ix->dollar.data(), ix->dollar.size() * sizeof(int64_t),
                     hipMemcpyHostToDevice));
  // text with 64 bytes of '$' padding on both sides (the TEXT windows may start before /
  // end after the text); suffix array with 16 bytes of slack for the 16-byte entry loads
  const bool wide = ix->sa_wide;
  if ((int64_t)ix->text.size() == ix->n && (int64_t)(wide ? ix->sa64.size() : ix->sa32.size()) == ix->n) {
    const size_t tb = (size_t)ix->n + 128 + 16;
    HIPCHK(hipMalloc(&ix->d_text, tb));
    HIPCHK(hipMemset(ix->d_text, 0, tb));
    HIPCHK(hipMemcpy((uint8_t*)ix->d_text + 64, ix->text.data(), (size_t)ix->n, hipMemcpyHostToDevice));
    const size_t sb = (size_t)ix->n * (wide ? 8 : 4);
    HIPCHK(hipMalloc(&ix->d_sa, sb + 16));
    HIPCHK(hipMemcpy(ix->d_sa, wide ? (const void*)ix->sa64.data() : (const void*)ix->sa32.data(), sb,
                     hipMemcpyHostToDevice));
  }
  return build_table(ix);
}

extern "C" int svdss_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

extern "C" int svdss_search_stream_create(int32_t device, void** stream) {
  if (!stream) return SVDSS_EINVAL;
  HIPCHK(hipSetDevice(device));
  hipStream_t st = nullptr;
  HIPCHK(svdss_make_stream(&st, "SVDSS_SEARCH_CUS"));
  *stream = (void*)st;
  return SVDSS_OK;
}

extern "C" int svdss_stream_destroy(void* stream) {
  if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream));
  return SVDSS_OK;
}

// another replica of a resident (or host-side) index in the HBM of `device`: SURVEY 8(e), index replicated per GPU.
// The new handle owns its device buffers and a copy of the small host-side parts; the source stays as it is.
extern "C" int svdss_index_replicate(const svdss_index_t* src, int32_t device, svdss_index_t** out) {
  if (!src || !out || device < 0) return SVDSS_EINVAL;
  if (!src->rec_lens.empty() && !getenv("SVDSS_INDEX_CPU")) {
    // the source came from a records file: the replica is built in the HBM of its own device (no 4-8 n bytes of suffix
    // array through the host)
    // a fresh thread's current device is 0, which may already hold an index and its table: the replica's order (and the
    // arena lent to its sort) must come from the free memory of ITS device
    HIPCHK(hipSetDevice(device));
    svdss_index* ix = new (std::nothrow) svdss_index();
    if (!ix) return SVDSS_ENOMEM;
    const size_t tb = table_bytes_for(src->n);
    int rc = svdss_index_build_gpu(src->records.data(), src->rec_lens.data(), (int32_t)src->rec_lens.size(), device, ix, true, tb);
    if (rc == SVDSS_OK) rc = table_and_blocks(ix);
    if (rc == SVDSS_OK) { *out = ix; return SVDSS_OK; }
    free_device_side(ix);
    delete ix;
    if (rc > 0) return rc;
    { const int mrc = materialize(src); if (mrc != SVDSS_OK) return mrc; }
  }
  // text and suffix array travel through the host copy of the source
  int rc = svdss_index_fetch_host(const_cast<svdss_index*>(src));
  if (rc != SVDSS_OK && rc != SVDSS_ENODEV) return rc;
  svdss_index* ix = new (std::nothrow) svdss_index();
  if (!ix) return SVDSS_ENOMEM;
  ix->n = src->n;
  memcpy(ix->acc, src->acc, sizeof ix->acc);
  ix->n_contigs = src->n_contigs;
  ix->sa_wide = src->sa_wide;
  try { ix->blocks = src->blocks; ix->dollar = src->dollar; } catch (...) { delete ix; return SVDSS_ENOMEM; }
  HIPCHK(hipSetDevice(device));
  ix->device = device;
  const size_t bb = ix->blocks.size() * sizeof(svdss_u4), db = (ix->dollar.size() + 1) * sizeof(int64_t);
  const bool have = (int64_t)src->text.size() == src->n &&
                    (int64_t)(src->sa_wide ? src->sa64.size() : src->sa32.size()) == src->n;
  auto fail = [&](int code) { free_device_side(ix); delete ix; return code; };
  if (hipMalloc(&ix->d_blocks, bb) != hipSuccess || hipMalloc(&ix->d_dollar, db) != hipSuccess) return fail(SVDSS_ENOMEM);
  if (hipMemcpy(ix->d_blocks, ix->blocks.data(), bb, hipMemcpyHostToDevice) != hipSuccess) return fail(SVDSS_EHIP);
  if (!ix->dollar.empty() &&
      hipMemcpy(ix->d_dollar, ix->dollar.data(), ix->dollar.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
    return fail(SVDSS_EHIP);
  if (have) {
    const size_t tb = (size_t)ix->n + 128 + 16, sb = (size_t)ix->n * (ix->sa_wide ? 8 : 4);
    if (hipMalloc(&ix->d_text, tb) != hipSuccess || hipMalloc(&ix->d_sa, sb + 16) != hipSuccess) return fail(SVDSS_ENOMEM);
    if (hipMemset(ix->d_text, 0, tb) != hipSuccess ||
        hipMemcpy((uint8_t*)ix->d_text + 64, src->text.data(), (size_t)ix->n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ix->d_sa, src->sa_wide ? (const void*)src->sa64.data() : (const void*)src->sa32.data(), sb,
                  hipMemcpyHostToDevice) != hipSuccess)
      return fail(SVDSS_EHIP);
  }
  rc = build_table(ix);
  if (rc != SVDSS_OK) return fail(rc);
  *out = ix;
  return SVDSS_OK;
}

static SvdssDevIndex host_view(const svdss_index* ix) {
  SvdssDevIndex v;
  v.blocks = ix->blocks.data();
  v.dollar = ix->dollar.data();
  v.n = ix->n;
  v.n_dollar = (int32_t)ix->dollar.size();
  v.k = 0; v.bs_after = 0; v.pad_ = 0; v.text = nullptr; v.sa = nullptr; v.table = nullptr;
  memcpy(v.acc, ix->acc, sizeof v.acc);
  return v;
}

extern "C" int64_t svdss_index_count(const svdss_index_t* ix, const uint8_t* pat, int64_t len) {
  if (!ix || !pat || len <= 0) return -1;
  if (materialize(ix) != SVDSS_OK) return -1;
  const SvdssDevIndex v = host_view(ix);
  int c = pat[len - 1];
  if (c > 5) return -1;
  int64_t lo = v.acc[c], hi = v.acc[c + 1];
  for (int64_t i = len - 2; i >= 0 && hi > lo; --i) {
    c = pat[i];
    if (c > 5) return -1;
    const int64_t a = v.acc[c];
    lo = a + svdss_rank_in_block(v, v.blocks + 4 * (lo >> SVDSS_BLOCK_SHIFT), c, lo);
    hi = a + svdss_rank_in_block(v, v.blocks + 4 * (hi >> SVDSS_BLOCK_SHIFT), c, hi);
  }
  return hi - lo;
}
