// place.hip -- SFS placement on the GPU (K7 of SURVEY 8(f)2).
//
// Replaces Clusterer::extend_alignment (/root/reference/clusterer.cpp:159-346) together with the helpers it calls --
// get_aligned_pairs (bam.cpp:92-134) and get_unique_kmers (clusterer.cpp:351-405) -- for a batch of alignments: every
// SFS (qs, l) of a read is turned into a reference interval through the read's CIGAR, extended on both sides to the
// nearest unique clean 7-mer within 100 aligned pairs, and the extended SFS of a read that overlap are merged.
//
// Mapping: one lane per alignment (the SFS of a read are handled in order: the reference carries `last_pos` from one
// SFS to the next, clusterer.cpp:172-192, and the merge at :314-336 is sequential too).  The list of aligned pairs --
// O(read length) pairs of 8 bytes in the reference -- is never materialised: pair i is computed from a prefix table of
// the CIGAR operations (index, query and reference position at which each operation starts), so a read costs
// O(n_cigar) memory and every lookup a binary search over the operations.  The two flanks of 100 pairs are expanded
// into lane-private arrays; 7-mers are compared as 56-bit integers read from the reference sequence resident in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svdss_hip.h"
#include "dev_arena.h"
#include "ref_dev.h"

extern thread_local std::string g_svdss_hip_err;

#define HIPCHK5(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      g_svdss_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);        \
      return (e_ == hipErrorOutOfMemory) ? SVDSS_ENOMEM : SVDSS_EHIP;             \
    }                                                                             \
  } while (0)

namespace {

constexpr int KSIZE = 7, FLANK = 100;   // config.hpp:89-90 (not settable from the command line)

struct OpTab {            // prefix table of one alignment's CIGAR: operation k covers pairs [idx[k], idx[k+1])
  const int32_t* idx;     // n_ops + 1
  const int32_t* q0;      // query position at the start of the operation
  const int32_t* r0;      // reference position at the start of the operation
  const uint8_t* kind;    // 0 = M/=/X (q and r advance), 1 = I/S (q only), 2 = D/N (r only)
  int n_ops;
};

// operation that holds pair i (i in [0, idx[n_ops]))
__device__ __forceinline__ int op_of(const OpTab& t, int i) {
  int lo = 0, hi = t.n_ops - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.idx[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ void pair_at(const OpTab& t, int k, int i, int& q, int& r) {
  const int d = i - t.idx[k];
  const int kind = t.kind[k];
  q = kind == 2 ? -1 : t.q0[k] + d;
  r = kind == 1 ? -1 : t.r0[k] + d;
}

// the 7-mer of the reference at r as an integer (the comparison of cseq.substr(r, 7) strings, clusterer.cpp:374,399)
__device__ __forceinline__ uint64_t kmer_at(const uint8_t* cseq, int64_t clen, int r) {
  uint64_t v = 0;
#pragma unroll
  for (int x = 0; x < KSIZE; ++x) {
    const int64_t p = (int64_t)r + x;
    v = (v << 8) | (p < clen ? cseq[p] : 0);   // (substr clamps at the end of the string)
  }
  return v;
}

// get_unique_kmers (clusterer.cpp:351-405) on the n pairs (wq, wr); returns the pair in (oq, or_)
__device__ void unique_kmer(const int* wq, const int* wr, int n, bool from_end, const uint8_t* cseq, int64_t clen, int& oq,
                            int& or_) {
  oq = -1; or_ = -1;
  if (n < KSIZE) return;
  // clean[i]: the k pairs from i on are all placed.  The reference's first loop visits exactly the clean positions
  // (its skip jumps past the unplaced pair, :365-371), so a k-mer's count is its number of clean occurrences.
  const int m = n - KSIZE + 1;
  int i = 0;
  while (i < m) {
    const int offset = from_end ? n - KSIZE - i : i;
    bool skip = false;
    for (int j = offset; j < offset + KSIZE; ++j)
      if (wq[j] == -1 || wr[j] == -1) { skip = true; i += (j - offset); break; }
    if (skip) { ++i; continue; }
    oq = wq[offset]; or_ = wr[offset];                 // last_kmer is assigned before the uniqueness test (:398)
    const uint64_t key = kmer_at(cseq, clen, wr[offset]);
    int count = 0;
    for (int a = 0; a < m; ++a) {
      bool clean = true;
      for (int j = a; j < a + KSIZE; ++j)
        if (wq[j] == -1 || wr[j] == -1) { clean = false; break; }
      if (clean && kmer_at(cseq, clen, wr[a]) == key) ++count;
    }
    if (count == 1) break;
    ++i;
  }
}

struct PlaceArgs {
  const uint8_t* ref;          // all chromosomes back to back
  const int64_t* ref_off;      // n_chrom + 1
  int32_t n_chrom;
  const int32_t* tid;          // per alignment
  const int64_t* op_off;       // per alignment: its operations in the tables below (n_aln + 1)
  const int32_t* op_idx;       // (one extra entry per alignment: op_off counts n_ops + 1 slots)
  const int32_t* op_q0;
  const int32_t* op_r0;
  const uint8_t* op_kind;
  const int32_t* sfs_qs;
  const int32_t* sfs_len;
  const int64_t* sfs_off;      // n_aln + 1
  int64_t n_aln;
  int32_t* out_count;          // per alignment
  int32_t* out;                // 5 per input SFS slot: rs, re, qs, qe, index of the first SFS merged into it
  unsigned long long* stats;   // unplaced, s_unplaced, e_unplaced, unknown
};

__global__ void __launch_bounds__(64) place_sfs_kernel(PlaceArgs A) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A.n_aln) return;
  const int64_t s0 = A.sfs_off[a], s1 = A.sfs_off[a + 1];
  int32_t* out = A.out + 5 * s0;
  int n_out = 0;
  const int tid = A.tid[a];
  if (tid < 0 || tid >= A.n_chrom) { A.out_count[a] = 0; return; }
  const uint8_t* cseq = A.ref + A.ref_off[tid];
  const int64_t clen = A.ref_off[tid + 1] - A.ref_off[tid];
  OpTab t;
  const int64_t o0 = A.op_off[a];
  t.n_ops = (int)(A.op_off[a + 1] - o0) - 1;
  t.idx = A.op_idx + o0; t.q0 = A.op_q0 + o0; t.r0 = A.op_r0 + o0; t.kind = A.op_kind + o0;
  const int n_pairs = t.n_ops > 0 ? t.idx[t.n_ops] : 0;
  unsigned long long st_unplaced = 0, st_s = 0, st_e = 0, st_unknown = 0;
  int last_pos = 0;
  int wq[FLANK], wr[FLANK];
  for (int64_t si = s0; si < s1; ++si) {
    const int s = A.sfs_qs[si], e = s + A.sfs_len[si] - 1;
    // the scan of clusterer.cpp:184-203 from last_pos on: the last placed pair with q < s, the first with q > e
    int aln_start = -1, aln_end = -1, refs = -1, refe = -1;
    if (n_pairs > 0 && last_pos < n_pairs) {
      for (int k = op_of(t, last_pos); k < t.n_ops; ++k) {
        if (t.kind[k] != 0) continue;
        const int i_lo = t.idx[k] > last_pos ? t.idx[k] : last_pos, i_hi = t.idx[k + 1] - 1;   // pairs of this op in range
        if (i_lo > i_hi) continue;
        const int q_lo = t.q0[k] + (i_lo - t.idx[k]), q_hi = t.q0[k] + (i_hi - t.idx[k]);
        if (q_lo < s) {                                  // placed pairs with q < s: the last one so far
          const int qq = q_hi < s - 1 ? q_hi : s - 1;
          aln_start = t.idx[k] + (qq - t.q0[k]);
          refs = t.r0[k] + (qq - t.q0[k]);
        }
        if (q_hi > e) {                                  // the first placed pair with q > e ends the scan
          const int qq = q_lo > e + 1 ? q_lo : e + 1;
          aln_end = t.idx[k] + (qq - t.q0[k]);
          refe = t.r0[k] + (qq - t.q0[k]);
          break;
        }
      }
    }
    if (aln_start >= 0) last_pos = aln_start;
    if (refs == -1 && refe == -1) { ++st_unplaced; continue; }
    if (refs == -1) { ++st_s; continue; }
    if (refe == -1) { ++st_e; continue; }
    // loc.front() / loc.back() (clusterer.cpp:228-244): the pairs at aln_start and aln_end
    int fq, fr, bq, br;
    pair_at(t, op_of(t, aln_start), aln_start, fq, fr);
    pair_at(t, op_of(t, aln_end), aln_end, bq, br);
    // the flank pairs in front of the region, oldest first (:248-258), and behind it (:260-269)
    int npre = aln_start < FLANK ? aln_start : FLANK;
    {
      int k = npre ? op_of(t, aln_start - npre) : 0;
      for (int x = 0; x < npre; ++x) {
        const int i = aln_start - npre + x;
        while (t.idx[k + 1] <= i) ++k;
        pair_at(t, k, i, wq[x], wr[x]);
      }
    }
    int pq, pr;
    unique_kmer(wq, wr, npre, true, cseq, clen, pq, pr);
    int npost = n_pairs - 1 - aln_end < FLANK ? n_pairs - 1 - aln_end : FLANK;
    {
      int k = npost ? op_of(t, aln_end + 1) : 0;
      for (int x = 0; x < npost; ++x) {
        const int i = aln_end + 1 + x;
        while (t.idx[k + 1] <= i) ++k;
        pair_at(t, k, i, wq[x], wr[x]);
      }
    }
    int tq, tr;
    unique_kmer(wq, wr, npost, false, cseq, clen, tq, tr);
    if (pq == -1 || pr == -1) { pq = fq; pr = fr; }
    if (tq == -1 || tr == -1) { tq = bq; tr = br; }
    if (pq == -1 || pr == -1 || tq == -1 || tr == -1) { ++st_unknown; continue; }
    if ((unsigned)pr > (unsigned)(tr + KSIZE)) continue;             // warning only (:301-303)
    const int rs = pr, re = tr + KSIZE, qs = pq, qe = tq + KSIZE;
    // merge with the extended SFS of this read that it overlaps (:314-336: first match, on rs only)
    int j;
    for (j = 0; j < n_out; ++j) {
      const int mrs = out[5 * j], mre = out[5 * j + 1];
      if ((rs <= mrs && mrs <= re) || (mrs <= rs && rs <= mre)) break;
    }
    if (j < n_out) {
      out[5 * j] = out[5 * j] < rs ? out[5 * j] : rs;
      out[5 * j + 1] = out[5 * j + 1] > re ? out[5 * j + 1] : re;
      out[5 * j + 2] = out[5 * j + 2] < qs ? out[5 * j + 2] : qs;
      out[5 * j + 3] = out[5 * j + 3] > qe ? out[5 * j + 3] : qe;
    } else {
      out[5 * n_out] = rs; out[5 * n_out + 1] = re; out[5 * n_out + 2] = qs; out[5 * n_out + 3] = qe;
      out[5 * n_out + 4] = (int32_t)(si - s0);
      ++n_out;
    }
  }
  A.out_count[a] = n_out;
  if (st_unplaced) atomicAdd(&A.stats[0], st_unplaced);
  if (st_s) atomicAdd(&A.stats[1], st_s);
  if (st_e) atomicAdd(&A.stats[2], st_e);
  if (st_unknown) atomicAdd(&A.stats[3], st_unknown);
}

}  // namespace

struct svdss_ref {
  int device = -1;
  void* d_ref = nullptr;
  void* d_off = nullptr;
  int32_t n_chrom = 0;
  DevArena arena;                // per-call buffers, reused
  hipStream_t stream = nullptr;
};

SvdssRefView svdss_ref_view(const svdss_ref_t* ref) {
  SvdssRefView v;
  if (ref) { v.device = ref->device; v.d_seq = (const uint8_t*)ref->d_ref; v.d_off = (const int64_t*)ref->d_off; v.n_chrom = ref->n_chrom; }
  return v;
}

extern "C" int svdss_ref_upload(const uint8_t* seqs, const int64_t* off, int32_t n_chrom, int32_t device, svdss_ref_t** out) {
  if (!out || n_chrom < 0 || device < 0 || (n_chrom > 0 && (!seqs || !off))) return SVDSS_EINVAL;
  HIPCHK5(hipSetDevice(device));
  svdss_ref* r = new (std::nothrow) svdss_ref();
  if (!r) return SVDSS_ENOMEM;
  r->device = device;
  r->n_chrom = n_chrom;
  const int64_t total = n_chrom ? off[n_chrom] : 0;
  auto fail = [&](int code) { svdss_ref_free(r); return code; };
  if (hipMalloc(&r->d_ref, (size_t)total + 16) != hipSuccess || hipMalloc(&r->d_off, sizeof(int64_t) * (size_t)(n_chrom + 1)) != hipSuccess)
    return fail(SVDSS_ENOMEM);
  if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) return fail(SVDSS_EHIP);
  if ((total && hipMemcpy(r->d_ref, seqs, (size_t)total, hipMemcpyHostToDevice) != hipSuccess) ||
      hipMemcpy(r->d_off, off, sizeof(int64_t) * (size_t)(n_chrom + 1), hipMemcpyHostToDevice) != hipSuccess)
    return fail(SVDSS_EHIP);
  *out = r;
  return SVDSS_OK;
}

extern "C" int svdss_ref_upload_parts(const uint8_t* const* seqs, const int64_t* lens, int32_t n_chrom, int32_t device, svdss_ref_t** out) {
  if (!out || n_chrom < 0 || device < 0 || (n_chrom > 0 && (!seqs || !lens))) return SVDSS_EINVAL;
  std::vector<int64_t> off((size_t)n_chrom + 1, 0);
  for (int32_t i = 0; i < n_chrom; ++i) {
    if (lens[i] < 0 || (lens[i] > 0 && !seqs[i])) return SVDSS_EINVAL;
    off[(size_t)i + 1] = off[(size_t)i] + lens[i];
  }
  HIPCHK5(hipSetDevice(device));
  svdss_ref* r = new (std::nothrow) svdss_ref();
  if (!r) return SVDSS_ENOMEM;
  r->device = device;
  r->n_chrom = n_chrom;
  const int64_t total = off[(size_t)n_chrom];
  auto fail = [&](int code) { svdss_ref_free(r); return code; };
  if (hipMalloc(&r->d_ref, (size_t)total + 16) != hipSuccess || hipMalloc(&r->d_off, sizeof(int64_t) * (size_t)(n_chrom + 1)) != hipSuccess)
    return fail(SVDSS_ENOMEM);
  if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) return fail(SVDSS_EHIP);
  // The chromosomes are ordinary (pageable) host memory: such a copy goes through the runtime's staging buffers at the
  // speed of one core's memcpy (GRCh38: 3.1 GB in ~0.45 s, on the critical path of `smooth` and of pass 1 of `call`).
  // Pieces of 64 MB dealt to a few threads, a stream each, overlap staging and DMA (SVDSS_REF_UPLOAD_THREADS, default 4).
  {
    struct Piece { const uint8_t* src; int64_t dst, n; };
    std::vector<Piece> pieces;
    const int64_t step = (int64_t)64 << 20;
    for (int32_t i = 0; i < n_chrom; ++i)
      for (int64_t a = 0; a < lens[i]; a += step) pieces.push_back(Piece{seqs[i] + a, off[(size_t)i] + a, std::min(step, lens[i] - a)});
    int T = getenv("SVDSS_REF_UPLOAD_THREADS") ? atoi(getenv("SVDSS_REF_UPLOAD_THREADS")) : 4;
    T = std::max(1, std::min(T, (int)std::min<size_t>(pieces.size(), 16)));
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&](hipStream_t st) {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= pieces.size() || bad.load()) break;
        const Piece& pc = pieces[k];
        if (hipMemcpyAsync((uint8_t*)r->d_ref + pc.dst, pc.src, (size_t)pc.n, hipMemcpyHostToDevice, st) != hipSuccess) { bad = 1; break; }
      }
      if (hipStreamSynchronize(st) != hipSuccess) bad = 1;
    };
    std::vector<std::thread> th;
    std::vector<hipStream_t> sts;
    for (int t = 1; t < T; ++t) {
      hipStream_t st = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
      sts.push_back(st);
      th.emplace_back([&, st, device] { if (hipSetDevice(device) != hipSuccess) { bad = 1; return; } work(st); });
    }
    work(r->stream);
    for (std::thread& x : th) x.join();
    for (hipStream_t st : sts) (void)hipStreamDestroy(st);
    if (bad.load()) return fail(SVDSS_EHIP);
  }
  if (hipMemcpyAsync(r->d_off, off.data(), sizeof(int64_t) * (size_t)(n_chrom + 1), hipMemcpyHostToDevice, r->stream) != hipSuccess ||
      hipStreamSynchronize(r->stream) != hipSuccess)
    return fail(SVDSS_EHIP);
  *out = r;
  return SVDSS_OK;
}

extern "C" void svdss_ref_free(svdss_ref_t* r) {
  if (!r) return;
  if (r->device >= 0) (void)hipSetDevice(r->device);
  if (r->d_ref) (void)hipFree(r->d_ref);
  if (r->d_off) (void)hipFree(r->d_off);
  if (r->stream) (void)hipStreamDestroy(r->stream);
  delete r;
}

extern "C" int svdss_place_sfs_batch(svdss_ref_t* ref, const int32_t* tid, const int32_t* pos, const uint32_t* cigar,
                                     const int64_t* cigar_off, const int32_t* sfs_qs, const int32_t* sfs_len,
                                     const int64_t* sfs_off, int64_t n_aln, int32_t* out_count, int32_t* out,
                                     int64_t stats[4]) {
  if (!ref || n_aln < 0 || !out_count || (n_aln > 0 && (!tid || !pos || !cigar_off || !sfs_off))) return SVDSS_EINVAL;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (n_aln == 0) return SVDSS_OK;
  HIPCHK5(hipSetDevice(ref->device));
  // the prefix tables of the CIGARs (host: O(total operations))
  const int64_t n_sfs = sfs_off[n_aln];
  if (n_sfs > 0 && (!sfs_qs || !sfs_len || !out)) return SVDSS_EINVAL;
  std::vector<int64_t> op_off((size_t)n_aln + 1, 0);
  for (int64_t a = 0; a < n_aln; ++a) op_off[(size_t)a + 1] = op_off[(size_t)a] + (cigar_off[a + 1] - cigar_off[a]) + 1;
  const int64_t n_slots = op_off[(size_t)n_aln];
  std::vector<int32_t> idx((size_t)n_slots), q0((size_t)n_slots), r0((size_t)n_slots);
  std::vector<uint8_t> kind((size_t)n_slots);
  std::vector<int64_t> op_off2((size_t)n_aln + 1, 0);   // operations that make pairs (H, P dropped), + 1 terminator each
  {
    int64_t w = 0;
    for (int64_t a = 0; a < n_aln; ++a) {
      op_off2[(size_t)a] = w;
      int32_t i = 0, q = 0, r = pos[a];
      for (int64_t c = cigar_off[a]; c < cigar_off[a + 1]; ++c) {
        const int32_t l = (int32_t)(cigar[c] >> 4), op = (int32_t)(cigar[c] & 0xf);
        int k;
        if (op == 0 || op == 7 || op == 8) k = 0;          // bam.cpp:100-106
        else if (op == 1 || op == 4) k = 1;                // :107-113
        else if (op == 2 || op == 3) k = 2;                // :114-120
        else continue;
        if (l == 0) continue;
        idx[(size_t)w] = i; q0[(size_t)w] = q; r0[(size_t)w] = r; kind[(size_t)w] = (uint8_t)k;
        ++w;
        i += l;
        if (k != 2) q += l;
        if (k != 1) r += l;
      }
      idx[(size_t)w] = i; q0[(size_t)w] = q; r0[(size_t)w] = r; kind[(size_t)w] = 3;   // terminator: idx[n_ops] = n_pairs
      ++w;
    }
    op_off2[(size_t)n_aln] = w;
  }
  const int64_t nw = op_off2[(size_t)n_aln];
  DevArena& ar = ref->arena;
  const size_t need = DevArena::padded(4 * (size_t)n_aln) + 2 * DevArena::padded(8 * (size_t)(n_aln + 1)) +
                      3 * DevArena::padded(4 * (size_t)nw) + DevArena::padded((size_t)nw) + 2 * DevArena::padded(4 * (size_t)n_sfs) +
                      DevArena::padded(4 * (size_t)n_aln) + DevArena::padded(20 * (size_t)n_sfs) + DevArena::padded(32);
  HIPCHK5(ar.reserve(need));
  PlaceArgs A;
  const hipStream_t st = ref->stream;
  auto up = [&](const void* src, size_t bytes) -> void* {
    void* d = ar.take(bytes ? bytes : 16);
    if (bytes) (void)hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st);
    return d;
  };
  A.ref = (const uint8_t*)ref->d_ref;
  A.ref_off = (const int64_t*)ref->d_off;
  A.n_chrom = ref->n_chrom;
  A.tid = (const int32_t*)up(tid, 4 * (size_t)n_aln);
  A.op_off = (const int64_t*)up(op_off2.data(), 8 * (size_t)(n_aln + 1));
  A.op_idx = (const int32_t*)up(idx.data(), 4 * (size_t)nw);
  A.op_q0 = (const int32_t*)up(q0.data(), 4 * (size_t)nw);
  A.op_r0 = (const int32_t*)up(r0.data(), 4 * (size_t)nw);
  A.op_kind = (const uint8_t*)up(kind.data(), (size_t)nw);
  A.sfs_qs = (const int32_t*)up(sfs_qs, 4 * (size_t)n_sfs);
  A.sfs_len = (const int32_t*)up(sfs_len, 4 * (size_t)n_sfs);
  A.sfs_off = (const int64_t*)up(sfs_off, 8 * (size_t)(n_aln + 1));
  A.n_aln = n_aln;
  A.out_count = (int32_t*)ar.take(4 * (size_t)n_aln);
  A.out = (int32_t*)ar.take(20 * (size_t)(n_sfs ? n_sfs : 1));
  A.stats = (unsigned long long*)ar.take(32);
  HIPCHK5(hipMemsetAsync(A.stats, 0, 32, st));
  hipLaunchKernelGGL(place_sfs_kernel, dim3((unsigned)((n_aln + 63) / 64)), dim3(64), 0, st, A);
  HIPCHK5(hipGetLastError());
  HIPCHK5(hipMemcpyAsync(out_count, A.out_count, 4 * (size_t)n_aln, hipMemcpyDeviceToHost, st));
  if (n_sfs) HIPCHK5(hipMemcpyAsync(out, A.out, 20 * (size_t)n_sfs, hipMemcpyDeviceToHost, st));
  unsigned long long hs[4] = {0, 0, 0, 0};
  HIPCHK5(hipMemcpyAsync(hs, A.stats, 32, hipMemcpyDeviceToHost, st));
  HIPCHK5(hipStreamSynchronize(st));
  if (stats) for (int k = 0; k < 4; ++k) stats[k] = (int64_t)hs[k];
  return SVDSS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// K6: `SVDSS smooth` -- Smoother::smooth_read (/root/reference/smoother.cpp:84-232) for a batch of alignments.
// One wavefront per alignment: the CIGAR operations are taken in order (their output offsets depend on the ones
// before), the bases of an operation are copied / compared 64 at a time: matches are replaced by the reference and
// counted as match / mismatch, insertions and deletions of at most 20 bases are removed (the deletion filled from
// the reference), longer ones and soft clips are kept.  Lane 0 keeps the new CIGAR (adjacent M runs merge).  The new
// bases are written as characters into a scratch row and packed to BAM's 4-bit form by the same wavefront.

namespace {

constexpr int SM_MIN_INDEL = 20;   // config.hpp:95

struct SmoothArgs {
  const uint8_t* ref;
  const int64_t* ref_off;
  const int32_t* tid;
  const int32_t* pos;
  const uint32_t* cigar;
  const int64_t* cigar_off;
  const uint8_t* seq4;        // packed bases of all records
  const int64_t* seq4_off;    // byte offset of every record's bases
  const uint8_t* qual;
  const int64_t* qual_off;    // byte offset of every record's qualities
  const int32_t* l_seq;
  const int64_t* cap_off;     // output capacity offsets in bases (even), n + 1
  int64_t n;
  uint8_t* scratch;           // cap_off[n] characters
  uint8_t* out_seq4;          // cap_off[n] / 2 bytes
  uint8_t* out_qual;          // cap_off[n] bytes
  uint32_t* out_cigar;        // at cigar_off
  int32_t* out_ncig;
  int32_t* out_len;
  unsigned long long* out_nm; // matches, mismatches (2 per record)
  uint8_t* out_ignore;
};

__device__ __forceinline__ char sm_base(const uint8_t* s4, int64_t i) {
  const uint8_t b = s4[i >> 1];
  return "=ACMGRSVTWYHKDBN"[(i & 1) ? (b & 15) : (b >> 4)];
}

__device__ __forceinline__ uint8_t sm_code(char c) {   // the inverse table write_record uses (upper / lower case)
  switch (c) {
    case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3;
    case 'G': case 'g': return 4; case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7;
    case 'T': case 't': return 8; case 'W': case 'w': return 9; case 'Y': case 'y': return 10; case 'H': case 'h': return 11;
    case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14; default: return 15;
  }
}

__global__ void __launch_bounds__(64) smooth_kernel(SmoothArgs A) {
  const int64_t a = blockIdx.x;
  if (a >= A.n) return;
  const int lane = threadIdx.x;
  const uint8_t* cseq = A.ref + A.ref_off[A.tid[a]];
  const uint8_t* s4 = A.seq4 + A.seq4_off[a];
  const uint8_t* q = A.qual + A.qual_off[a];
  const int64_t lq = A.l_seq[a];
  uint8_t* ns = A.scratch + A.cap_off[a];
  uint8_t* nq = A.out_qual + A.cap_off[a];
  uint32_t* nc = A.out_cigar + A.cigar_off[a];
  int64_t ref_off = A.pos[a], q_off = 0, no = 0;
  unsigned long long nm = 0, nx = 0;
  uint32_t m_diff = 0;
  int n_nc = 0;
  bool ignore = true;
  uint32_t last_word = 0;       // nc[n_nc - 1], kept in a register by every lane (uniform)
  for (int64_t c = A.cigar_off[a]; c < A.cigar_off[a + 1]; ++c) {
    const uint32_t w = A.cigar[c], l = w >> 4, op = w & 0xf;
    if (op == 0 || op == 7 || op == 8) {
      for (uint32_t j = lane; j < l; j += 64) {
        const char rc = (char)cseq[ref_off + j];
        ns[no + j] = (uint8_t)rc;
        nq[no + j] = q_off + j < lq ? q[q_off + j] : 255;
        if (rc == sm_base(s4, q_off + j)) ++nm; else ++nx;
      }
      no += l; ref_off += l; q_off += l;
      if (n_nc && (last_word & 0xf) == 0) last_word += (l + m_diff) << 4;
      else { last_word = ((l + m_diff) << 4) | 0; ++n_nc; }
      if (lane == 0) nc[n_nc - 1] = last_word;
      m_diff = 0;
    } else if (op == 1) {
      if ((int)l > SM_MIN_INDEL) {
        ignore = false;
        for (uint32_t j = lane; j < l; j += 64) {
          ns[no + j] = (uint8_t)sm_base(s4, q_off + j);
          nq[no + j] = q_off + j < lq ? q[q_off + j] : 255;
        }
        no += l;
        last_word = w; ++n_nc;
        if (lane == 0) nc[n_nc - 1] = w;
      }
      q_off += l;
    } else if (op == 2) {
      if ((int)l <= SM_MIN_INDEL) {
        for (uint32_t j = lane; j < l; j += 64) {
          ns[no + j] = cseq[ref_off + j];
          nq[no + j] = q_off + j < lq ? q[q_off + j] : 255;     // qualities of the NEXT read bases (smoother.cpp:164-166)
        }
        no += l;
        m_diff += l;
      } else {
        ignore = false;
        last_word = w; ++n_nc;
        if (lane == 0) nc[n_nc - 1] = w;
      }
      ref_off += l;
    } else if (op == 4) {
      ignore = false;
      for (uint32_t j = lane; j < l; j += 64) {
        ns[no + j] = (uint8_t)sm_base(s4, q_off + j);
        nq[no + j] = q_off + j < lq ? q[q_off + j] : 255;
      }
      no += l; q_off += l;
      last_word = w; ++n_nc;
      if (lane == 0) nc[n_nc - 1] = w;
    } else break;
  }
  // match / mismatch totals over the wave
  for (int d = 32; d >= 1; d >>= 1) {
    nm += __shfl_xor(nm, d, 64);
    nx += __shfl_xor(nx, d, 64);
  }
  __syncthreads();   // the scratch row is complete
  uint8_t* o4 = A.out_seq4 + (A.cap_off[a] >> 1);
  for (int64_t b = lane; b < (no + 1) / 2; b += 64) {
    const uint8_t hi = sm_code((char)ns[2 * b]), lo = 2 * b + 1 < no ? sm_code((char)ns[2 * b + 1]) : 0;
    o4[b] = (uint8_t)((hi << 4) | lo);
  }
  if (lane == 0) {
    A.out_ncig[a] = n_nc;
    A.out_len[a] = (int32_t)no;
    A.out_nm[2 * a] = nm; A.out_nm[2 * a + 1] = nx;
    A.out_ignore[a] = ignore ? 1 : 0;
  }
}

}  // namespace

extern "C" int svdss_smooth_batch(svdss_ref_t* ref, const int32_t* tid, const int32_t* pos, const uint32_t* cigar,
                                  const int64_t* cigar_off, const uint8_t* seq4, const int64_t* seq4_off, const uint8_t* qual,
                                  const int64_t* qual_off, const int32_t* l_seq, const int64_t* cap_off, int64_t n,
                                  uint8_t* out_seq4, uint8_t* out_qual, uint32_t* out_cigar, int32_t* out_ncig,
                                  int32_t* out_len, int64_t* out_match_mismatch, uint8_t* out_ignore) {
  if (!ref || n < 0) return SVDSS_EINVAL;
  if (n == 0) return SVDSS_OK;
  if (!tid || !pos || !cigar || !cigar_off || !seq4 || !seq4_off || !qual || !qual_off || !l_seq || !cap_off || !out_seq4 ||
      !out_qual || !out_cigar || !out_ncig || !out_len || !out_match_mismatch || !out_ignore)
    return SVDSS_EINVAL;
  HIPCHK5(hipSetDevice(ref->device));
  for (int64_t i = 0; i < n; ++i)
    if (tid[i] < 0 || tid[i] >= ref->n_chrom || (cap_off[i] & 1) || cap_off[i + 1] < cap_off[i]) return SVDSS_EINVAL;
  const int64_t n_cig = cigar_off[n], cap = cap_off[n];
  int64_t s4_bytes = 0, q_bytes = 0;
  for (int64_t i = 0; i < n; ++i) {
    s4_bytes = std::max<int64_t>(s4_bytes, seq4_off[i] + ((int64_t)l_seq[i] + 1) / 2);
    q_bytes = std::max<int64_t>(q_bytes, qual_off[i] + l_seq[i]);
  }
  DevArena& ar = ref->arena;
  HIPCHK5(ar.reserve(3 * DevArena::padded(4 * (size_t)n) + 4 * DevArena::padded(8 * (size_t)(n + 1)) + 2 * DevArena::padded(4 * (size_t)n_cig) +
                     DevArena::padded((size_t)s4_bytes) + DevArena::padded((size_t)q_bytes) + 2 * DevArena::padded((size_t)cap) +
                     DevArena::padded((size_t)cap / 2 + 8) + 2 * DevArena::padded(4 * (size_t)n) + DevArena::padded(16 * (size_t)n) +
                     DevArena::padded((size_t)n)));
  const hipStream_t st = ref->stream;
  auto up = [&](const void* src, size_t bytes) -> void* {
    void* d = ar.take(bytes ? bytes : 16);
    if (bytes) (void)hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st);
    return d;
  };
  SmoothArgs A;
  A.ref = (const uint8_t*)ref->d_ref;
  A.ref_off = (const int64_t*)ref->d_off;
  A.tid = (const int32_t*)up(tid, 4 * (size_t)n);
  A.pos = (const int32_t*)up(pos, 4 * (size_t)n);
  A.cigar = (const uint32_t*)up(cigar, 4 * (size_t)n_cig);
  A.cigar_off = (const int64_t*)up(cigar_off, 8 * (size_t)(n + 1));
  A.seq4 = (const uint8_t*)up(seq4, (size_t)s4_bytes);
  A.seq4_off = (const int64_t*)up(seq4_off, 8 * (size_t)n);
  A.qual = (const uint8_t*)up(qual, (size_t)q_bytes);
  A.qual_off = (const int64_t*)up(qual_off, 8 * (size_t)n);
  A.l_seq = (const int32_t*)up(l_seq, 4 * (size_t)n);
  A.cap_off = (const int64_t*)up(cap_off, 8 * (size_t)(n + 1));
  A.n = n;
  A.scratch = (uint8_t*)ar.take((size_t)cap);
  A.out_seq4 = (uint8_t*)ar.take((size_t)cap / 2 + 8);
  A.out_qual = (uint8_t*)ar.take((size_t)cap);
  A.out_cigar = (uint32_t*)ar.take(4 * (size_t)(n_cig ? n_cig : 1));
  A.out_ncig = (int32_t*)ar.take(4 * (size_t)n);
  A.out_len = (int32_t*)ar.take(4 * (size_t)n);
  A.out_nm = (unsigned long long*)ar.take(16 * (size_t)n);
  A.out_ignore = (uint8_t*)ar.take((size_t)n);
  hipLaunchKernelGGL(smooth_kernel, dim3((unsigned)n), dim3(64), 0, st, A);
  HIPCHK5(hipGetLastError());
  HIPCHK5(hipMemcpyAsync(out_seq4, A.out_seq4, (size_t)cap / 2, hipMemcpyDeviceToHost, st));
  HIPCHK5(hipMemcpyAsync(out_qual, A.out_qual, (size_t)cap, hipMemcpyDeviceToHost, st));
  if (n_cig) HIPCHK5(hipMemcpyAsync(out_cigar, A.out_cigar, 4 * (size_t)n_cig, hipMemcpyDeviceToHost, st));
  HIPCHK5(hipMemcpyAsync(out_ncig, A.out_ncig, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHK5(hipMemcpyAsync(out_len, A.out_len, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHK5(hipMemcpyAsync(out_match_mismatch, A.out_nm, 16 * (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHK5(hipMemcpyAsync(out_ignore, A.out_ignore, (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHK5(hipStreamSynchronize(st));
  return SVDSS_OK;
}
