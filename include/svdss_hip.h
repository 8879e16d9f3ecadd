/*
 * svdss_hip.h -- C ABI of the MI355X-native SVDSS hot path (libsvdss_hip.so).
 *
 * Drop-in boundary B2 of SURVEY.md section 8(b): the reference (Parsoa/SVDSS
 * v2.1.1) has no plugin layer; the seam is where its stage drivers call into
 * per-item kernels and third-party C functions.  Each entry point below cites
 * the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions: plain-old-data in and out, caller-owned buffers, int status
 * return (0 = ok; the host turns non-zero into a stderr message + exit(1)
 * like ping_pong.cpp:62-63), no exceptions across the ABI, no torch types.
 * Symbols use ropebwt3's nt6 alphabet: $=0 A=1 C=2 G=3 T=4 N/other=5.
 */
#ifndef SVDSS_HIP_H
#define SVDSS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVDSS_OK 0
#define SVDSS_EINVAL 1    /* bad argument */
#define SVDSS_ENOMEM 2    /* host or device allocation failed */
#define SVDSS_EIO 3       /* file could not be read/written or has a bad format */
#define SVDSS_EHIP 4      /* a HIP runtime call failed (no GPU, launch error, ...) */
#define SVDSS_ENODEV 5    /* index is not resident on a device */
#define SVDSS_ERANGE 6    /* input exceeds a layout limit (per-symbol count >= 2^32, read >= 2^31) */

const char* svdss_strerror(int code);
/* Last HIP error string seen by this library on the calling thread ("" if none). */
const char* svdss_last_hip_error(void);

/* ---- a1: base -> nt6 ----------------------------------------------------
 * Replaces seq_nt6_table (ping_pong.hpp:46-52) as applied at
 * ping_pong.cpp:90-94 (BAM path) and rb3_char2nt6 at ping_pong.cpp:158. */
int svdss_nt6_encode(const char* seq, int64_t n, uint8_t* out);

/* ---- a5: FM-index -------------------------------------------------------
 * svdss_index_build replaces `ropebwt3 build` re-exported as `SVDSS index`
 * (main.cpp:15-17,34-37); svdss_index_load replaces rb3_fmi_restore
 * (ping_pong.cpp:245).  contigs = concatenated nt6 symbols of all records,
 * lens[i] symbols each.  The index covers every record and its reverse
 * complement, each '$'-terminated.  The index is built in HBM; a machine without a GPU gets SVDSS_EHIP ("no GPU
 * found"), not a quiet run of the host builder -- that one serves the texts the GPU builder refuses and
 * SVDSS_INDEX_CPU=1. */
typedef struct svdss_index svdss_index_t;

int svdss_index_build(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs,
                      int32_t threads, svdss_index_t** out);
/* The same index built directly in the HBM of `device` and left resident there, k-mer table included
 * (csrc/index_gpu.hip: text, suffix sorting, BWT and rank blocks on the GPU; GRCh38 lengths in tens of seconds
 * instead of minutes of host cores).  Text and suffix array stay on the device -- svdss_index_save fetches them
 * when asked.  Falls back to svdss_index_build + svdss_index_to_device when the device lacks the memory.
 * This is what every rank of a multi-GPU run calls for its own replica (index replicated). */
int svdss_index_build_device(const uint8_t* contigs, const int64_t* lens, int32_t n_contigs,
                             int32_t threads, int32_t device, svdss_index_t** out);
int svdss_index_save(const svdss_index_t* ix, const char* path);
/* The records file (magic "SVDSSRC1": BWT length, acc[], record lengths, the nt6 records): what `SVDSS index` leaves
 * beside the .fmd as `<fmd>.svdss`.  svdss_index_load takes it and returns an index that holds the records only; the
 * layout is built when the index is made resident (svdss_index_to_device / svdss_index_replicate: in the HBM of that
 * device, seconds for GRCh38 lengths -- faster than any disk delivers the 58 GB of text + suffix array
 * svdss_index_save writes) or when a host-side accessor needs it.  Stands where the .fmd stands between main_build and
 * rb3_fmi_restore (main.cpp:34-37, ping_pong.cpp:245). */
int svdss_index_save_records(const svdss_index_t* ix, const char* path);
/* The index as ropebwt3 dumps it: the rld0 run-length BWT (`.fmd`, magic "RLD\3"), the file `ropebwt3 build -d` /
 * upstream `SVDSS index` writes and rb3_fmi_restore reads (main.cpp:34-37, ping_pong.cpp:245; run_svdss:136-147 reuses
 * an existing $FA.fmd).  Format restated from the published rld0 sources in csrc/rld0.cpp [UPSTREAM-UNVERIFIED]. */
int svdss_index_save_fmd(const svdss_index_t* ix, const char* path);
/* The BWT an rld0 `.fmd` holds, as nt6 bytes: *n_out symbols (bwt_out may be NULL to ask for the size only;
 * SVDSS_ERANGE if cap is too small). */
int svdss_fmd_read_bwt(const char* path, uint8_t* bwt_out, int64_t cap, int64_t* n_out);
/* Restores an index: this library's own files (svdss_index_save, svdss_index_save_records), or an rld0 `.fmd` -- then
 * `<path>.svdss` (the file `SVDSS index` leaves beside the .fmd) is read if it is there, not older, and carries the
 * symbol counts of the .fmd's header; otherwise the BWT is
 * decoded, the records are recovered from it (they must come with their reverse complements, as ropebwt3 build -d
 * inserts them) and the index is rebuilt on the GPU (no GPU: SVDSS_EHIP, unless SVDSS_INDEX_CPU=1 asks for the host
 * builder). */
int svdss_index_load(const char* path, svdss_index_t** out);
void svdss_index_free(svdss_index_t* ix);
/* number of BWT symbols, = sum_i 2*(lens[i]+1) */
int64_t svdss_index_size(const svdss_index_t* ix);
/* acc[c] = number of symbols < c (rb3_fmi_t::acc, used by rb3_fmd_set_intv) */
int svdss_index_acc(const svdss_index_t* ix, int64_t acc[7]);
/* decode the BWT as nt6 bytes (svdss_index_size() of them) */
int svdss_index_bwt(const svdss_index_t* ix, uint8_t* bwt_out);
/* bytes the index occupies in HBM once resident */
int64_t svdss_index_device_bytes(const svdss_index_t* ix);
/* order K of the k-mer table built by svdss_index_to_device (0 before / without it) */
int32_t svdss_index_kmer(const svdss_index_t* ix);
/* The index as a rank structure alone.  `SVDSS index` leaves the rank blocks and '$' rows behind the records of its
 * sidecar; svdss_index_attach_blocks (before svdss_index_to_device, on a handle svdss_index_load gave for the same path)
 * makes them the handle's host side and drops the records: svdss_index_to_device then uploads ~n/2 bytes instead of
 * rebuilding text, suffix array and k-mer table (rb3_fmi_restore, ping_pong.cpp:245, restores exactly this much: the
 * reference searches with rank steps alone).  The search is then one rank step per rb3_fmd_extend -- about a million reads
 * per second at GRCh38 lengths instead of 8 - 24 M, results identical -- which pays when few reads are searched.
 * SVDSS_EINVAL: the file has no such section (restore as before). */
int svdss_index_attach_blocks(svdss_index_t* ix, const char* path);
int svdss_index_load_blocks(const char* path, svdss_index_t** out);       /* the same as a handle of its own: the records are never read */
int svdss_index_append_blocks(const svdss_index_t* ix, const char* path);   /* behind the records svdss_index_save_records wrote at `path` */
/* An upper limit for the order of tables built FROM NOW ON (process-wide; 0 = none; an explicit SVDSS_KMER wins).  The
 * table's build time quarters per step down, the search kernel slows down by about a factor of two per step: a process that
 * restores the index for one input may learn, while the suffix array is still being sorted, that it has few reads to search
 * (`SVDSS search` on a smoothed BAM: the putative filter of ping_pong.cpp:202-203 skips most of them) -- the limit is read
 * when the table's build begins.  Results never depend on K. */
void svdss_index_kmer_limit(int32_t k);
/* share of the k-mer occurrences (sampled while the table is built) that belong to k-mers with 8 or more of them: a
 * reference rich in young repeat families has >= 0.35, and the search then finishes backward phases on deep intervals by
 * binary search of the suffix array (the BS instantiation of the kernel; SVDSS_BS=0|1 overrides, SVDSS_BS_DEEP moves the
 * threshold).  Results never depend on it. */
double svdss_index_deep_frac(const svdss_index_t* ix);
/* copy the index into the HBM of `device` (replicated per GPU): BWT blocks,
 * text, suffix array, and the 4^K k-mer table (K = floor(log4 n)+1, env SVDSS_KMER overrides) */
int svdss_index_to_device(svdss_index_t* ix, int32_t device);

/* A resident index checked against its own text by direct comparison, independently of how it was built
 * (csrc/index_verify.hip): every `stride`-th suffix-array row i has SA[i] in range, text[SA[i]..) < text[SA[i+1]..)
 * as strings, BWT[i] == text[SA[i]-1]; every rank block's counters continue the previous block's; the '$' rows are
 * the listed ones; the text's symbol histogram is acc[].  stride 1 = the whole index (GRCh38 lengths: ~1 s).
 * out[0] rows checked, [1] order violations, [2] BWT mismatches, [3] entries out of range, [4] block / histogram
 * violations, [5] '$'-list violations, [6] first bad row or -1, [7] longest common prefix met.  The index this
 * stands for is the one rb3_fmi_restore returns (ping_pong.cpp:245). */
int svdss_index_verify_device(const svdss_index_t* ix, int64_t stride, int64_t out[8]);

/* GPUs this process sees (0: none). */
int svdss_device_count(void);
/* A stream of the kind the library makes for its own searches, for callers that pass their stream to
 * svdss_sfs_search_batch_device (bench.py): non-blocking, and restricted to the compute units SVDSS_SEARCH_CUS names
 * ("first,count": bits of hipExtStreamCreateWithCUMask's mask; csrc/hip_check.h) when that is set.  The call-side
 * batch objects do the same with SVDSS_CALL_CUS: kernels that each fill the chip run side by side on disjoint CUs
 * instead of taking turns (no counterpart in the reference: its threads share cores the same way through OpenMP). */
int svdss_search_stream_create(int32_t device, void** stream);
int svdss_stream_destroy(void* stream);
/* One more replica of the index, in the HBM of `device` (the index is replicated per GPU, reads and
 * sub-clusters shard): a new handle with its own device buffers; *src is not modified (its text and suffix array are
 * fetched to the host first if it was built on a device). */
int svdss_index_replicate(const svdss_index_t* src, int32_t device, svdss_index_t** out);

/* Size of the interval of a pattern (occurrences in contigs + revcomps) via
 * the host copy of the index; rb3_fmd_set_intv + repeated rb3_fmd_extend(...,1)
 * as in ping_pong.cpp:12-22.  For tests/diagnostics; not a hot path. */
int64_t svdss_index_count(const svdss_index_t* ix, const uint8_t* pattern_nt6, int64_t len);

/* ---- a2+a3+a4 (+a8): batched ping-pong SFS search -----------------------
 * Replaces PingPong::ping_pong_search (ping_pong.hpp:84-85, ping_pong.cpp:4-49)
 * called per read from PingPong::process_batch (ping_pong.cpp:176-209), i.e.
 * the rb3_fmd_set_intv / rb3_fmd_extend loops, for a whole batch at once.
 *
 * reads    : concatenated nt6 symbols; read i = reads[offsets[i] .. offsets[i+1])
 *            (no terminator byte needed; the P[l]=0 of ping_pong.cpp:94 is implied)
 * flags    : SVDSS_SFS_ASSEMBLE applies Assembler::assemble (assembler.cpp:34-56)
 *            per read as ping_pong.cpp:219-222 does unless --noassemble.
 * Results per read i: counts[i] records (qs,len); without ASSEMBLE in the
 * order ping_pong.cpp:41 pushes them (descending qs), with ASSEMBLE ascending
 * qs (assembler.cpp:36).  n_ext[i] = number of rb3_fmd_extend calls the
 * reference would have made for that read (ping_pong.cpp:20,35).
 * overlap is fixed at -1 (config.hpp:87; the option is never declared,
 * config.cpp:30-55,74). */
#define SVDSS_SFS_ASSEMBLE 1

typedef struct svdss_sfs_batch svdss_sfs_batch_t;

/* Host-buffer entry point: uploads, searches on the index's device, leaves
 * results in *out (created if *out == NULL, otherwise its buffers are reused). */
int svdss_sfs_search_batch(const svdss_index_t* ix, const uint8_t* reads, const int64_t* offsets,
                           int64_t n_reads, int32_t flags, svdss_sfs_batch_t** out);
/* BAM-record entry point: the reads as they sit in BAM records -- 4-bit packed bases ("=ACMGRSVTWYHKDBN", two per
 * byte, high nibble first), read i at seq4[byte_off[i]] with l_seq[i] bases -- are uploaded packed (half the bytes of
 * the nt6 form) and expanded to nt6 on the GPU, which stands for the per-base loop of ping_pong.cpp:90-94
 * (seq_nt16_str, then seq_nt6_table: A/C/G/T -> 1..4, every other code -> 5).  Everything else as
 * svdss_sfs_search_batch.  Copies go through the batch object's own stream: calls on different batch objects overlap. */
int svdss_sfs_search_batch_bam(const svdss_index_t* ix, const uint8_t* seq4, const int64_t* byte_off,
                               const int32_t* l_seq, int64_t n_reads, int32_t flags, svdss_sfs_batch_t** out);
/* Page-locked host memory for the buffers handed to the host-buffer entry points (copies from it run at PCIe speed
 * and asynchronously; pageable memory works too, slower). */
int svdss_host_alloc(int64_t bytes, void** out);
void svdss_host_free(void* p);
/* ---- BGZF blocks inflated on the GPU (csrc/inflate.hip).  Stands where htslib's bgzf_read / inflate stand under
 * sam_read1 (/root/reference/ping_pong.cpp:58,247-249): every BGZF block is an independent deflate stream of at most
 * 64 KB, one wavefront inflates one block.  `comp` holds the compressed bytes (host memory, page-locked for speed);
 * block i is the raw deflate stream comp[coff, coff + clen) and inflates to exactly isize bytes at d_out + uoff (d_out: a
 * device buffer of out_bytes bytes, e.g. from svdss_device_alloc).  If host_out is not NULL the inflated bytes are also
 * copied there.  Returns when everything is done (work runs on the object's own stream: calls on different objects
 * overlap).  A block that does not inflate to isize bytes gives SVDSS_EIO and its index in *bad_block; the CRC32 of the
 * BGZF footer is NOT checked on this path. */
typedef struct svdss_inflate svdss_inflate_t;
typedef struct svdss_bgzf_block {
  int64_t coff;    /* first byte of the deflate stream in comp */
  int32_t clen;    /* its length */
  int32_t isize;   /* inflated size (BGZF footer), 0..65536 */
  int64_t uoff;    /* where the block inflates to, relative to d_out */
} svdss_bgzf_block_t;
int svdss_bgzf_inflate(svdss_inflate_t** obj, int device, const uint8_t* comp, int64_t comp_bytes,
                       const svdss_bgzf_block_t* blocks, int64_t n_blocks, void* d_out, uint8_t* host_out,
                       int64_t out_bytes, int64_t* bad_block);
double svdss_inflate_kernel_ms(const svdss_inflate_t* obj);   /* the inflate kernel of the last call, HIP events */
void svdss_inflate_free(svdss_inflate_t* obj);
/* ---- BGZF blocks DEFLATED on the GPU (csrc/deflate.hip).  Stands where htslib's bgzf_write / deflate stand under
 * sam_write1 in `SVDSS smooth` (/root/reference/smoother.cpp:441-494).  `in` (host memory) is cut into blocks of
 * block_bytes (<= 0xff00; the last one may be short); block i becomes a BGZF member at out + i * out_stride (host
 * memory; out_stride >= block_bytes + 64, a multiple of 4): 18-byte header with BSIZE, a deflate stream of dynamic-
 * Huffman coded literals and runs (4, 8, ... 256 equal bytes as one match at distance 1; no other matches: a
 * level-1-class encoder for packed bases and qualities; incompressible quarters are stored), and 8 bytes LEFT FOR THE CALLER to fill with CRC32 and ISIZE; out_len[i] = the member's length with
 * those 8 bytes.  out_stride 0: the members are written back to back instead (out needs in_bytes + 64 per block).
 * Any inflater reads the result; it is not the byte stream zlib would write.  Returns when done (the
 * object's own stream: calls on different objects overlap). */
typedef struct svdss_deflate svdss_deflate_t;
int svdss_bgzf_deflate(svdss_deflate_t** obj, int32_t device, const uint8_t* in, int64_t in_bytes, int32_t block_bytes,
                       uint8_t* out, int64_t out_stride, int32_t* out_len);
double svdss_deflate_kernel_ms(const svdss_deflate_t* obj);
void svdss_deflate_free(svdss_deflate_t* obj);
/* plain device memory for the callers of the entry points that take device pointers */
int svdss_device_alloc(int device, int64_t bytes, void** out);
void svdss_device_free(int device, void* p);
int svdss_device_memset(int device, void* d_dst, int value, int64_t bytes);
int svdss_device_download(int device, void* dst, const void* d_src, int64_t bytes);
/* Device-buffer entry point: d_reads/d_offsets already resident in HBM on the
 * index's device; work is enqueued on `stream` (a hipStream_t, NULL = default
 * stream) and is complete when this returns. */
int svdss_sfs_search_batch_device(const svdss_index_t* ix, const uint8_t* d_reads,
                                  const int64_t* d_offsets, int64_t n_reads, int64_t total_syms,
                                  int32_t flags, void* stream, svdss_sfs_batch_t** out);
int64_t svdss_sfs_batch_nreads(const svdss_sfs_batch_t* b);
int64_t svdss_sfs_batch_total(const svdss_sfs_batch_t* b);       /* sum of counts */
int64_t svdss_sfs_batch_total_ext(const svdss_sfs_batch_t* b);   /* sum of n_ext */
/* Small batches are searched with several lanes per read (segments stitched exactly, see
 * DESIGN.md); these report the segment count the last call used (1 = one lane per read) and
 * how many reads had to be redone unsegmented because their chains could not be stitched. */
int32_t svdss_sfs_batch_segments(const svdss_sfs_batch_t* b);
int64_t svdss_sfs_batch_fallbacks(const svdss_sfs_batch_t* b);
int32_t svdss_sfs_batch_used_bs(const svdss_sfs_batch_t* b);   /* 1: the last call launched the BS instantiation */
/* duration of the search kernel(s) of the last call, from HIP events on its stream */
double svdss_sfs_batch_kernel_ms(const svdss_sfs_batch_t* b);
/* HIP-event time of the search kernel alone (first pass; without ordering, stitching, assembling, gather) */
double svdss_sfs_batch_search_kernel_ms(const svdss_sfs_batch_t* b);
/* copy results to host; any pointer may be NULL to skip it.
 * counts,n_ext: n_reads entries; qs,len: svdss_sfs_batch_total() entries. */
int svdss_sfs_batch_fetch(const svdss_sfs_batch_t* b, int64_t* counts, int32_t* qs, int32_t* len,
                          int64_t* n_ext);
/* device addresses of the result arrays (valid until the next search on this batch
 * object or svdss_sfs_batch_free): counts int64[n_reads], qs/len int32[total],
 * n_ext int64[n_reads].  Lets a caller hand results to RCCL without a host hop. */
int svdss_sfs_batch_device_ptrs(const svdss_sfs_batch_t* b, void** counts, void** qs, void** len,
                                void** n_ext);
void svdss_sfs_batch_free(svdss_sfs_batch_t* b);

/* ---- a6 + a7 on the device: BAM records in, SFS out (csrc/bam_device.hip) ------------------------------------
 * Replaces PingPong::load_batch_bam (ping_pong.cpp:53-128: sam_read1 = BGZF inflate + the block_size chain of the
 * records, the flag / l_qseq / tid filters :66-79, 4-bit -> nt6 :90-94) and the head of PingPong::process_batch (:196-203:
 * bam_aux_get("XF"/"HP"), the putative filter) for a run of consecutive BGZF blocks at once, followed by the search of
 * svdss_sfs_search_batch_device.  Only compressed bytes go up; names, tags and SFS come down.
 *
 * A svdss_bam_stream_t is one BAM file being read: it holds the bytes of the record that straddles two consecutive
 * batches and gives the batches their turn in file order (everything else of a batch -- inflate, the record chain of
 * its segments, unpack, search -- overlaps with the other batches, each on its own svdss_bam_batch_t / stream / thread).
 * n_ref = reference sequences in the BAM header (tid / mtid outside -1 .. n_ref - 1 never start a record).
 *
 * svdss_bam_batch_run, batch number seq = 0, 1, 2, ... of the file (every number exactly once, from any thread; is_last on
 * the final one): n_chunks pieces of the file in page-locked or ordinary host memory; piece c holds n_blocks[c] BGZF
 * blocks, block i = the raw deflate stream comp[c][coff, coff + clen) that inflates to isize bytes with CRC32 crc[c][i]
 * (uoff is ignored: the blocks of a batch inflate back to back in the order given).  skip = inflated bytes in front of the
 * first record (the BAM header; batch 0 only).  flags: SVDSS_SFS_ASSEMBLE, SVDSS_BAM_PUTATIVE (reads whose XF tag is not 0
 * keep their slot but are not searched, ping_pong.cpp:202-203).
 * Errors: SVDSS_EIO with svdss_bam_batch_error() = "BGZF inflate failed" | "BGZF block CRC mismatch" | "truncated record" |
 * "corrupt record" | "core.tid < 0. Why are we here? Please check" (the reference's fatal message, :76-79); once a batch
 * failed every later batch of the stream returns its error. */
typedef struct svdss_bam_stream svdss_bam_stream_t;
typedef struct svdss_bam_batch svdss_bam_batch_t;
#define SVDSS_BAM_PUTATIVE 0x100
int svdss_bam_stream_create(int32_t n_ref, svdss_bam_stream_t** out);
void svdss_bam_stream_free(svdss_bam_stream_t* s);
const char* svdss_bam_stream_error(const svdss_bam_stream_t* s);
/* A stream over a REGION of the file (`SVDSS search --gpus N` gives every GPU its own range of BGZF blocks -- north_star:
 * "BAM regions partition across the GPUs"; the per-shard loop of ping_pong.cpp:53-128).  Call before batch 0.
 * open_start: the region begins inside a record nobody has located yet: batch 0 starts the chain at the first record its
 *   segments guess (plausible fields, a plausible record behind it) and keeps the bytes in front of it (svdss_bam_stream_head).
 *   The guess is proved by the caller: the previous region's tail + this head must be a chain of whole records -- run
 *   them as a one-batch stream of their own; if they are not, run the region again with open_start = 0 and carry = the
 *   previous region's tail (the incomplete record in front of batch 0).
 * open_end: the last batch may end inside a record; svdss_bam_stream_tail are its bytes (no "truncated record"). */
int svdss_bam_stream_region(svdss_bam_stream_t* s, int32_t open_start, int32_t open_end, const uint8_t* carry, int64_t n_carry);
int64_t svdss_bam_stream_head(const svdss_bam_stream_t* s, const uint8_t** bytes);   /* after batch 0 */
int64_t svdss_bam_stream_tail(const svdss_bam_stream_t* s, const uint8_t** bytes);   /* after the last batch */
/* segments whose guessed first record the chain did not arrive at (walked again, exactly), of *n_segments in all */
int64_t svdss_bam_stream_rewalked(const svdss_bam_stream_t* s, int64_t* n_segments);
int svdss_bam_batch_run(svdss_bam_stream_t* s, int64_t seq, int32_t is_last, int64_t skip, const svdss_index_t* ix,
                        int32_t n_chunks, const uint8_t* const* comp, const int64_t* comp_bytes,
                        const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* n_blocks,
                        int32_t flags, svdss_bam_batch_t** out);
/* The two halves of svdss_bam_batch_run, and a PARK for the reads of batches whose front half runs while no index is
 * resident yet: PingPong::run loads the index (ping_pong.cpp:245, seconds for a human genome) and only then starts reading
 * the BAM (:329-363); here the BAM front end (inflate, CRC, record chain, filters, 4-bit -> nt6) runs beside the index
 * restore, the unpacked reads wait in HBM, and are searched in large launches -- one lane per read -- when the index is there.
 *   svdss_bam_batch_front   everything of svdss_bam_batch_run up to the search.  park = NULL, or no room in it, or closed:
 *                           the reads stay in the batch object and svdss_bam_batch_search(b, ix) finishes the batch.
 *                           Else they go into the park (svdss_bam_batch_parked: group >= 0, index of the batch's first
 *                           searched read in the group) and the batch object is free for the next batch; names / tags /
 *                           slots of the batch are on the host either way (svdss_bam_batch_result: counts / qs / len empty).
 *                           group = -2: the batch has no read to search.
 *   svdss_bam_park_create   read_bytes of nt6 symbols + max_reads offsets in the HBM of `device` -- allocate it BEFORE the
 *                           index restore starts so that nothing is handed back to the driver mid-stream.
 *   svdss_bam_park_close    no more reservations (the index is resident); the open group is closed.
 *   svdss_bam_park_search   group g (closed, all its batches unpacked: waits for that) as ONE svdss_sfs_search_batch_device
 *                           launch; results with svdss_sfs_batch_fetch, reads in the order of their reservations.
 * A group closes at SVDSS_PARK_GROUP_READS reads (262,144) or SVDSS_PARK_GROUP_MB of symbols (4,096). */
typedef struct svdss_bam_park svdss_bam_park_t;
int svdss_bam_park_create(int32_t device, int64_t read_bytes, int64_t max_reads, svdss_bam_park_t** out);
void svdss_bam_park_free(svdss_bam_park_t* p);
int svdss_bam_park_close(svdss_bam_park_t* p);
int64_t svdss_bam_park_groups(svdss_bam_park_t* p);
int svdss_bam_park_group(svdss_bam_park_t* p, int64_t g, int64_t* n_batches, int64_t* n_reads, int64_t* n_syms);
int32_t svdss_bam_park_group_ready(svdss_bam_park_t* p, int64_t g);   /* 1: closed and unpacked -- may be searched while later groups still fill */
int svdss_bam_park_search(svdss_bam_park_t* p, int64_t g, const svdss_index_t* ix, int32_t flags, svdss_sfs_batch_t** sfs);
int svdss_bam_batch_front(svdss_bam_stream_t* s, int64_t seq, int32_t is_last, int64_t skip, int32_t device, svdss_bam_park_t* park,
                          int32_t n_chunks, const uint8_t* const* comp, const int64_t* comp_bytes,
                          const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* n_blocks,
                          int32_t flags, svdss_bam_batch_t** out);
int svdss_bam_batch_parked(const svdss_bam_batch_t* b, int64_t* group, int64_t* first, int64_t* n_reads);
int svdss_bam_batch_search(svdss_bam_batch_t* b, const svdss_index_t* ix);
/* What the last run of a batch object left on the host (valid until its next run / svdss_bam_batch_free).  A "slot" is a
 * record that passed the filters of ping_pong.cpp:66-75, in file order -- what load_batch_bam deals to the threads. */
typedef struct svdss_bam_result {
  int64_t n_records;        /* records of the batch (passed or not) */
  int64_t n_slots;          /* passed the flag and length filters */
  int64_t n_searched;       /* of them searched (all, unless SVDSS_BAM_PUTATIVE) */
  int64_t n_short;          /* dropped for l_qseq < 100 (the reference warns once per read, :70-75) */
  int64_t total_sfs;
  const int32_t* name_off;  /* n_slots + 1: slot i is called names[name_off[i] .. name_off[i + 1]) */
  const char* names;
  const int32_t* hp;        /* n_slots: HP tag, 0 if absent */
  const int32_t* sidx;      /* n_slots: index among the searched reads, -1 = not searched */
  const int64_t* counts;    /* n_searched: SFS per searched read */
  const int32_t* qs;        /* total_sfs, read after read */
  const int32_t* len;
  double inflate_kernel_ms;
  double stage_ms[8];       /* host clock between the waits of the run: 0 upload + inflate + CRC + segment walk, 1 waiting for
                               the turn, 2 the turn (carry, link), 3 fields / filters / scans, 4 unpack, 5 search, 6 results down */
} svdss_bam_result_t;
int svdss_bam_batch_result(const svdss_bam_batch_t* b, svdss_bam_result_t* out);
/* The same front end (inflate, CRC, record chain, the turn) for `SVDSS call`, which needs whole records of FEW reads:
 * Clusterer::load_batch keeps the primary alignments with mapq >= min_mapq whose read has SFS (clusterer.cpp:108-145),
 * fill_clusters those that overlap a cluster (sam_itr_querys per cluster, :495-545).  A filter names what is kept: records
 * without flags 4 / 256 / 2048, with mapq >= min_mapq, and -- when names and / or regions are given -- whose read name is
 * in `names` (name i = names[name_off[i] .. name_off[i + 1])) or whose alignment [pos, bam_endpos) overlaps one of the
 * regions [reg_beg, reg_end) of its reference (sorted by (tid, beg)); names = NULL: no name test, names given with
 * n_names = 0: the empty set (no record passes by name).  Names are compared by a 64-bit hash: a kept record
 * may rarely be one nobody asked for (the caller looks at the name anyway), a wanted one is never dropped.
 * svdss_bam_select_run takes a batch like svdss_bam_batch_run; the kept records (block_size field first, as in the file)
 * come back in file order at 4-aligned offsets of one page-locked buffer. */
typedef struct svdss_bam_filter svdss_bam_filter_t;
int svdss_bam_filter_create(int32_t device, int32_t min_mapq, int32_t n_ref, const char* names, const int64_t* name_off,
                            int64_t n_names, const int32_t* reg_tid, const int32_t* reg_beg, const int32_t* reg_end,
                            int64_t n_regions, svdss_bam_filter_t** out);
void svdss_bam_filter_free(svdss_bam_filter_t* f);
int svdss_bam_select_run(svdss_bam_stream_t* s, int64_t seq, int32_t is_last, int64_t skip, const svdss_bam_filter_t* f,
                         int32_t n_chunks, const uint8_t* const* comp, const int64_t* comp_bytes,
                         const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* n_blocks,
                         svdss_bam_batch_t** out);
typedef struct svdss_bam_selection {
  int64_t n_records;        /* records of the batch */
  int64_t n_selected;       /* kept */
  int64_t n_bytes;
  const int64_t* rec_off;   /* n_selected + 1: kept record k begins at bytes + rec_off[k] (its 4-byte block_size, then the record) */
  const uint8_t* bytes;
  double inflate_kernel_ms;
  double stage_ms[8];
  int32_t slim;             /* 1: the records are slim ones (svdss_bam_store_select) */
} svdss_bam_selection_t;
int svdss_bam_batch_selection(const svdss_bam_batch_t* b, svdss_bam_selection_t* out);
/* ONE pass over the file for `SVDSS call`.  Clusterer::run reads the BAM twice: align_and_extend places the SFS of the reads
 * that have any (clusterer.cpp:56-156), fill_clusters fetches, per cluster, the alignments that overlap it (sam_itr_querys,
 * :477-610).  What the second pass looks at of a record is its core, name, CIGAR, bases and HP tag; a svdss_bam_store_t keeps
 * exactly that ("slim" record: block_size | core | name | CIGAR | packed bases | HP as one int32 tag when the record had an
 * integer one -- no qualities, no other tags; ~1/3 of the record) of EVERY record that passes the flag / mapq filters, in
 * HBM, while the first pass runs (svdss_bam_select_store_run = svdss_bam_select_run + the store); the second pass is then a
 * kernel over resident records (svdss_bam_store_select, per stored batch: the slim records that overlap a region of the
 * filter come down, in file order).  With a store the records of the first pass come down slim as well (slim = 1).  A store
 * that would exceed max_bytes stays incomplete (svdss_bam_store_batches): the caller reads the file again, through its
 * index or the device path, as before.  initial_bytes: allocated at once (what the caller expects the store to need: an
 * allocation in the middle of the stream stalls it); the rest in arenas of SVDSS_STORE_ARENA_MB (2,048). */
typedef struct svdss_bam_store svdss_bam_store_t;
int svdss_bam_store_create(int32_t device, int64_t max_bytes, int64_t initial_bytes, svdss_bam_store_t** out);
void svdss_bam_store_free(svdss_bam_store_t* t);
int64_t svdss_bam_store_batches(svdss_bam_store_t* t, int32_t* complete, int64_t* n_records, int64_t* n_bytes);
int svdss_bam_store_reset(svdss_bam_store_t* t);   /* forget what is stored, keep the memory (a file region that runs again) */
int svdss_bam_select_store_run(svdss_bam_stream_t* s, int64_t seq, int32_t is_last, int64_t skip, const svdss_bam_filter_t* f,
                               svdss_bam_store_t* store, int32_t n_chunks, const uint8_t* const* comp, const int64_t* comp_bytes,
                               const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* n_blocks,
                               svdss_bam_batch_t** out);
int svdss_bam_store_select(svdss_bam_store_t* t, int64_t seq, const svdss_bam_filter_t* f, svdss_bam_batch_t** out);
/* `SVDSS smooth` on the same front end (csrc/bam_smooth.inc): BGZF blocks in, BGZF blocks out; the inflated records never
 * leave the device.  Stands where smoother.cpp:349-571 stand (loader :498-571 with the filters of :509-537, smooth_read
 * :84-232, rebuild_bam_entry :50-82, the writer :441-494).  A svdss_bam_smooth_t names the reference (svdss_ref_upload:
 * upper-case ASCII), tid_map[t] = chromosome of BAM reference t in it or -1 (records on it are dropped, like every record
 * that is unmapped, secondary, supplementary, below min_mapq or shorter than 2 bases).
 * svdss_bam_smooth_measure: a batch like svdss_bam_batch_run; the kept records' matches / mismatches over their M
 * operations and whether their CIGAR fits read and contig come down (compute_maxaccuracy, smoother.cpp:259-346).
 * svdss_bam_smooth_run: every kept record rewritten (XF 0: smoothed -- equal to the reference except at indels longer than
 * 20 and clips --, 1: mismatch rate above max_mismatch_rate, 2: nothing interesting, 3: CIGAR does not fit; 1-3 keep bases
 * and CIGAR), in file order, as the bytes of a BAM stream cut into BGZF blocks of 0xff00 bytes wherever the batches end:
 * batches take a second turn in file order at which the bytes behind a batch's last full block go to the next one
 * (svdss_bam_stream_set_output_prefix: what precedes batch 0 -- the BAM header of the output).  The blocks of a batch
 * (deflated by csrc/deflate.hip, CRC32 / ISIZE filled in) are in svdss_bam_smoothed_t::bgzf -- in host_out when it is
 * given and large enough (page-locked memory of the caller's, svdss_host_alloc: the writer then needs no copy), else in
 * memory of the batch object; the batch with is_last also holds the stream's short last block; the 28-byte EOF marker is
 * the caller's. */
typedef struct svdss_ref svdss_ref_t;
typedef struct svdss_bam_smooth svdss_bam_smooth_t;
int svdss_bam_smooth_create(const svdss_ref_t* ref, const int32_t* tid_map, int32_t n_ref, int32_t min_mapq,
                            svdss_bam_smooth_t** out);
void svdss_bam_smooth_free(svdss_bam_smooth_t* s);
int svdss_bam_stream_set_output_prefix(svdss_bam_stream_t* s, const uint8_t* bytes, int64_t n);
int svdss_bam_smooth_measure(svdss_bam_stream_t* s, int64_t seq, int32_t is_last, int64_t skip, const svdss_bam_smooth_t* sm,
                             int32_t n_chunks, const uint8_t* const* comp, const int64_t* comp_bytes,
                             const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* n_blocks,
                             svdss_bam_batch_t** out);
int svdss_bam_smooth_run(svdss_bam_stream_t* s, int64_t seq, int32_t is_last, int64_t skip, const svdss_bam_smooth_t* sm,
                         double max_mismatch_rate, uint8_t* host_out, int64_t host_cap, int32_t n_chunks, const uint8_t* const* comp, const int64_t* comp_bytes,
                         const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* n_blocks,
                         svdss_bam_batch_t** out);
typedef struct svdss_bam_smoothed {
  int64_t n_records;              /* records of the batch */
  int64_t n_kept;                 /* of them kept */
  const int64_t* match_mismatch;  /* measure: 2 per kept record */
  const uint8_t* fits;            /* measure: 1 per kept record */
  int64_t out_bytes;              /* run: bytes of the batch's records in the output stream */
  const uint8_t* bgzf;            /* run: the batch's BGZF members, back to back (host memory of the batch object) */
  int64_t bgzf_bytes;
  int64_t n_xf[4];                /* run: records by XF value */
  double inflate_kernel_ms;
  double stage_ms[8];             /* 0-2 as svdss_bam_result_t, 3 filters + CIGAR walk, 4 sizes + records, 5 the output turn, 6 deflate + down */
} svdss_bam_smoothed_t;
int svdss_bam_batch_smoothed(const svdss_bam_batch_t* b, svdss_bam_smoothed_t* out);
const char* svdss_bam_batch_error(const svdss_bam_batch_t* b);
void svdss_bam_batch_free(svdss_bam_batch_t* b);

/* ---- a10: SFS placement ------------------------------------------------------
 * Replaces Clusterer::extend_alignment (clusterer.cpp:159-346) with get_aligned_pairs (bam.cpp:92-134) and
 * get_unique_kmers (clusterer.cpp:351-405) for a batch of alignments: every SFS (qs, len) of a read is mapped to the
 * reference through the read's CIGAR, extended on both sides to the nearest unique clean 7-mer within 100 aligned
 * pairs (ksize 7, flank 100: config.hpp:89-90), and overlapping extended SFS of one read are merged (:314-336).
 * svdss_ref_upload: the chromosome sequences as `call` holds them (upper-case ASCII, chromosomes.cpp:9-27), back to
 * back with off[n_chrom + 1], resident on `device`.
 * Per alignment i: tid[i] (index into the uploaded chromosomes; outside -> nothing placed), pos[i], its BAM CIGAR words
 * cigar[cigar_off[i] .. cigar_off[i+1]) (len << 4 | op), and its SFS sfs_qs/sfs_len[sfs_off[i] .. sfs_off[i+1]) in the
 * order the .sfs file lists them (the reference carries `last_pos` from one to the next, :172-192).
 * Out: out_count[i] merged extended SFS, written at out[5 * (sfs_off[i] + j)]: rs, re, qs, qe (sfs.hpp:52-62 as filled at
 * clusterer.cpp:305-308) and the index (within the read's list) of the first SFS merged into it, whose qname / htag the
 * entry keeps.  stats: SFS counted unplaced, s_unplaced, e_unplaced, unknown (clusterer.cpp:207-226,296). */
typedef struct svdss_ref svdss_ref_t;
int svdss_ref_upload(const uint8_t* seqs, const int64_t* off, int32_t n_chrom, int32_t device, svdss_ref_t** out);
/* the same from one host buffer per chromosome (as load_chromosomes leaves them, chromosomes.cpp:20: a map of strings): no
 * 3 GB copy on the host to put them back to back first */
int svdss_ref_upload_parts(const uint8_t* const* seqs, const int64_t* lens, int32_t n_chrom, int32_t device, svdss_ref_t** out);
void svdss_ref_free(svdss_ref_t* ref);
int svdss_place_sfs_batch(svdss_ref_t* ref, const int32_t* tid, const int32_t* pos, const uint32_t* cigar,
                          const int64_t* cigar_off, const int32_t* sfs_qs, const int32_t* sfs_len, const int64_t* sfs_off,
                          int64_t n_aln, int32_t* out_count, int32_t* out, int64_t stats[4]);

/* ---- smooth: Smoother::smooth_read (smoother.cpp:84-232) for a batch of alignments --------------------------------
 * Per alignment i (all of them eligible and consistent: the caller filters as smoother.cpp:509-537 does and checks that
 * the CIGAR stays inside the chromosome and adds up to l_seq): tid / pos / CIGAR words as for svdss_place_sfs_batch, the
 * packed 4-bit bases at seq4[seq4_off[i]], the qualities at qual[qual_off[i]], l_seq[i].  The kernel walks the CIGAR:
 * M/=/X copy the reference and count matches / mismatches against the read; I and D of at most 20 bases (config.hpp:95)
 * are dropped / filled from the reference, longer ones and soft clips are kept; adjacent M runs merge.  Out, per
 * alignment: the new bases 4-bit packed at out_seq4[cap_off[i] / 2], the new qualities at out_qual[cap_off[i]]
 * (cap_off: caller-chosen even capacities in bases, >= query + reference length of the alignment), the new CIGAR words at
 * out_cigar[cigar_off[i]] with out_ncig[i] of them, out_len[i] bases, out_match_mismatch[2i], [2i+1] (the XF = 1 test is
 * mismatch / match > accuracy, smoother.cpp:213) and out_ignore[i] (nothing interesting: XF = 2, :215). */
int svdss_smooth_batch(svdss_ref_t* ref, const int32_t* tid, const int32_t* pos, const uint32_t* cigar,
                       const int64_t* cigar_off, const uint8_t* seq4, const int64_t* seq4_off, const uint8_t* qual,
                       const int64_t* qual_off, const int32_t* l_seq, const int64_t* cap_off, int64_t n,
                       uint8_t* out_seq4, uint8_t* out_qual, uint32_t* out_cigar, int32_t* out_ncig, int32_t* out_len,
                       int64_t* out_match_mismatch, uint8_t* out_ignore);

/* ---- a15: global dual-affine realignment of consensus to reference --------
 * Replaces ksw_extd2_sse(km=0, qlen, query, tlen, target, m, mat, q, e, q2, e2,
 * w=-1, zdrop=-1, end_bonus=-1, flag=0, &ez) as called at caller.cpp:348-349
 * (ksw2 @HEAD), for a batch of (query = POA consensus, target = reference
 * window) pairs.  Symbols are 0..m-1 (caller.hpp:25-37 _char26_table: ACGT ->
 * 0-3, other -> 4); mat is the m x m substitution matrix (caller.cpp:336-337).
 * Per pair: score = ez.score (VCF `AS`, caller.cpp:351) and the CIGAR as ksw2
 * packs it, len<<4 | op with op 0=M 1=I 2=D (caller.cpp:353-355), gaps
 * left-aligned, from a backtrack at (tlen-1, qlen-1).  Empty query or target:
 * score 0, no CIGAR (ksw2 returns before touching ez). */
typedef struct svdss_aln_batch svdss_aln_batch_t;

int svdss_align_global_batch(const uint8_t* queries, const int64_t* q_off, const uint8_t* targets,
                             const int64_t* t_off, int64_t n_pairs, int32_t m, const int8_t* mat,
                             int32_t gapo, int32_t gape, int32_t gapo2, int32_t gape2, int32_t device,
                             svdss_aln_batch_t** out);
int64_t svdss_aln_batch_npairs(const svdss_aln_batch_t* b);
int64_t svdss_aln_batch_total_cigar(const svdss_aln_batch_t* b);   /* sum of n_cigar */
int64_t svdss_aln_batch_cells(const svdss_aln_batch_t* b);         /* sum of tlen*qlen */
double svdss_aln_batch_kernel_ms(const svdss_aln_batch_t* b);
/* scores int32[n_pairs], n_cigar int64[n_pairs], cigar uint32[total_cigar] (pairs concatenated) */
int svdss_aln_batch_fetch(const svdss_aln_batch_t* b, int32_t* scores, int64_t* n_cigar, uint32_t* cigar);
void svdss_aln_batch_free(svdss_aln_batch_t* b);

/* ---- a14: partial-order-alignment consensus --------------------------------
 * Replaces Caller::run_poa's abpoa_msa + consensus read-out (caller.cpp:257-308; abPOA
 * v1.5.3: global, convex gap 4/2/24/1, match 2, mismatch 4, no seeding, input order, one
 * heaviest-bundle consensus) for a batch of sub-clusters.  abPOA's source is not available;
 * the algorithm is specified in oracle/svdss_oracle_poa.c (tolerance vs abPOA: DESIGN.md).
 * seqs: symbols 0..4 (caller.hpp:25-37) of all reads concatenated, seq_off[n_seqs+1];
 * cluster c owns reads cluster_off[c] .. cluster_off[c+1]-1, aligned in that order.
 * Result per cluster: consensus symbols 0..4 ("ACGTN"[b], caller.cpp:297); empty for a
 * cluster without reads. */
typedef struct svdss_poa_batch svdss_poa_batch_t;

int svdss_poa_consensus_batch(const uint8_t* seqs, const int64_t* seq_off, const int64_t* cluster_off,
                              int64_t n_clusters, int32_t device, svdss_poa_batch_t** out);
int64_t svdss_poa_batch_nclusters(const svdss_poa_batch_t* b);
int64_t svdss_poa_batch_total(const svdss_poa_batch_t* b);     /* sum of consensus lengths */
int64_t svdss_poa_batch_cells(const svdss_poa_batch_t* b);     /* DP cells computed */
double svdss_poa_batch_kernel_ms(const svdss_poa_batch_t* b);
/* sub-clusters the LDS-resident kernel handed to the HBM kernel (graph too large for LDS, a node with
 * more than 8 predecessors, or the full-matrix fallback of the band) */
int64_t svdss_poa_batch_hbm(const svdss_poa_batch_t* b);
/* sub-clusters the first stage (several sub-clusters per wavefront, csrc/poa_quad.hip) handed on to the
 * one-wavefront-per-sub-cluster rounds: a row wider than its lanes, a node with more than 7 predecessors, a band that
 * lost the sink, a graph beyond its first allocation */
int64_t svdss_poa_batch_quad_back(const svdss_poa_batch_t* b);
/* developer check of the cross-lane primitives of csrc/poa_quad_core.h (DPP shifts / scans / rotations, permutes,
 * ballots) on the device: in = 64 values, out = 3 x 64 x 12 values (group widths 16, 32, 64); tests/test_poa_quad_gpu.py
 * holds them against the definitions the CPU wave emulator of the tests uses */
int svdss_poa_quad_selftest(const int32_t* in, int32_t* out, int32_t device);
int svdss_poa_batch_fetch(const svdss_poa_batch_t* b, int64_t* cons_len, uint8_t* cons);
void svdss_poa_batch_free(svdss_poa_batch_t* b);

/* ---- a17: rapidfuzz::fuzz::ratio(a, b) (rapidfuzz-cpp v1.10.4) -----------
 * as called at caller.cpp:456,458 on the REF/ALT alleles of adjacent SVs:
 * 100 * (1 - (|a|+|b| - 2 LCS(a,b)) / (|a|+|b|)), 100 for two empty strings.
 * a/b: concatenated byte strings with offsets[n_pairs+1].  lcs_out may be NULL. */
int svdss_indel_ratio_batch(const uint8_t* a, const int64_t* a_off, const uint8_t* b, const int64_t* b_off,
                            int64_t n_pairs, int32_t device, double* ratio_out, int64_t* lcs_out);

#ifdef __cplusplus
}
#endif
#endif /* SVDSS_HIP_H */
