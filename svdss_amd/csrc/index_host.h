// index_host.h -- host-side index object behind the opaque svdss_index_t.
#pragma once
#include <stdint.h>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "fmd_layout.h"

// std::allocator whose construct() leaves trivially constructible elements alone: resize() of a vector that is about to
// be filled from a file does not write the zeros first (3.1 GB of records: 0.3 s on one core, and every page touched by
// that core instead of by the threads that read into it)
template <class T>
struct SvdssNoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = SvdssNoInitAlloc<U>; };
  SvdssNoInitAlloc() = default;
  template <class U> SvdssNoInitAlloc(const SvdssNoInitAlloc<U>&) {}
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};

struct svdss_index {
  int64_t n = 0;              // BWT length = sum over contigs of 2*(len+1)
  int64_t acc[7] = {0};
  int32_t n_contigs = 0;
  std::vector<svdss_u4, SvdssNoInitAlloc<svdss_u4>> blocks;   // 4 * (n/128 + 1) quarters (resize() leaves the new elements alone)
  std::vector<int64_t> dollar;    // sorted BWT positions of '$'
  std::vector<uint8_t> text;      // nt6 text (contig $ revcomp $ ...), n symbols
  std::vector<uint32_t> sa32;     // suffix array when n < 2^32 ...
  std::vector<uint64_t> sa64;     // ... else 64-bit
  bool sa_wide = false;           // suffix array entries are 64-bit (n >= 2^31 - 1)
  // The records the index stands for (nt6, concatenated) -- all an index restored from a records file holds until it
  // is made resident or a host-side accessor needs the layout (svdss_index_materialize): rebuilding GRCh38 lengths in
  // HBM takes seconds, reading the 58 GB of text + suffix array from a file takes longer on any disk.
  std::vector<uint8_t, SvdssNoInitAlloc<uint8_t>> records;
  std::vector<int64_t> rec_lens;
  // device residency (filled by svdss_index_to_device)
  int device = -1;
  void* d_blocks = nullptr;
  void* d_dollar = nullptr;
  void* d_text = nullptr;         // allocation start; text begins 64 bytes in
  void* d_sa = nullptr;
  void* d_table = nullptr;
  size_t d_table_cap = 0;         // bytes behind d_table when it was allocated ahead of the table's build (index_gpu.hip)
  int32_t table_k = 0;
  double deep_frac = 0.0;   // share of the K-mer occurrences that belong to K-mers with 8 or more of them (sampled while the table is built)
};

// Builds text (contig $ revcomp $ ...), suffix array, BWT and the block layout.
// Returns 0 or a SVDSS_E* code.
int svdss_index_build_host(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs,
                           int32_t threads, svdss_index* out);
int svdss_index_save_host(const svdss_index* ix, const char* path);
// the records file ("SVDSSRC1": n, acc, record lengths, nt6 records): what `SVDSS index` leaves beside the .fmd
int svdss_index_save_records_host(const svdss_index* ix, const char* path);
int svdss_index_load_records_host(const char* path, svdss_index* ix);
// the rank blocks and '$' rows a records file may carry behind the records (round 6): the index as a rank structure alone
int svdss_index_append_blocks_host(const svdss_index* ix, const char* path);
int svdss_index_load_blocks_host(const char* path, svdss_index* ix);
// true when *ix holds records only (restored from a records file, nothing built yet)
inline bool svdss_index_is_lazy(const svdss_index* ix) { return ix->blocks.empty() && !ix->rec_lens.empty(); }
int svdss_index_load_host(const char* path, svdss_index* ix);
void svdss_index_decode_bwt(const svdss_index* ix, uint8_t* bwt);
// index_gpu.hip: the whole index built in the HBM of `device` and left resident there (text and suffix array are
// not copied to the host: svdss_index_fetch_host does that on demand).  0 = done, -1 = not possible here (use the
// host builder), SVDSS_EINVAL / SVDSS_ERANGE as svdss_index_build_host reports them.
int svdss_index_build_gpu(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs, int32_t device,
                          svdss_index* out, bool defer_host_blocks = false, size_t table_bytes = 0);
// defer_host_blocks: the host copy of the rank blocks is left out; svdss_index_fetch_blocks brings it down (the restore
// paths run it on a thread beside the k-mer table's build)
int svdss_index_fetch_blocks(svdss_index* ix);
int svdss_index_fetch_host(svdss_index* ix);
int svdss_index_fetch_text(svdss_index* ix);   // the text alone (n bytes, not the 4-8 n of the suffix array)
// the kernels' view of a resident index (index_api.hip)
SvdssDevIndex svdss_device_view(const svdss_index* ix);
