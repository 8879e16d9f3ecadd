// svdss_main.cpp -- `SVDSS` host CLI on top of libsvdss_hip.so.
//
// Keeps the process boundary B1 of SURVEY.md 8(b):
//   SVDSS index  -d ref.fa -o ref.fa.fmd [-t T]        (/root/reference/main.cpp:34-37, run_svdss:142)
//   SVDSS smooth --reference R --bam B [--threads T] [--min-mapq N] [--accp F]      (main.cpp:69-77; smooth_host.cpp)
//   SVDSS search --index F --bam B | --fastx Q [--threads T] [--bsize N] [--noputative]
//                [--noassemble] [--omax N] [--verbose]  (config.cpp:30-55, main.cpp:62-68)
//   SVDSS call   --reference R --bam B --sfs S [...]    (main.cpp:55-61; call_host.cpp)
//   SVDSS --version                                     (main.cpp:45-47)
// SFS text goes to stdout exactly as PingPong::output_batch prints it
// (ping_pong.cpp:213-236), logs to stderr, fatal conditions exit(1).
// Additions of this program: --gpus N (search, call, smooth), --io-threads N, --verbose stage timings.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <sys/stat.h>
#include <ctime>
#include <deque>
#include <functional>
#include <malloc.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utime.h>
#include <vector>

#include <unistd.h>

#include "../../include/svdss_hip.h"
#include "bam_reader.h"
#include "bgzf_scanner.h"
#include "bam_device_select.h"
#include "gpu_inflate_hook.h"
#include "cli_options.h"
#include "call_host.h"
#include "fastx_reader.h"

static const char* VERSION = "v2.1.1";  // main.cpp:19

static const char* MAIN_USAGE =
    "Usage: SVDSS <index|smooth|search|call> --help\n"
    "  index   build the FM-index of a reference (FASTA, gz ok):  SVDSS index -d ref.fa -o ref.fa.fmd [-t T]\n"
    "  search  extract sample-specific strings: SVDSS search --index ref.fa.fmd --bam reads.bam > specifics.txt\n"
    "  call    call SVs from the specific strings:    SVDSS call --reference ref.fa --bam reads.bam --sfs specifics.txt\n"
    "  smooth  smooth a BAM (reads equal the reference except at long indels): SVDSS smooth --reference ref.fa --bam in.bam > out.bam\n";

static const char* SMOOTH_USAGE =
    "Usage: SVDSS smooth --reference <FASTA> --bam <BAM> > smoothed.bam\n"
    "      --min-mapq <int>   minimum mapping quality (default: 20)\n"
    "      --accp <float>     accuracy percentile (default: 0.98)\n";

static const char* CALL_USAGE =
    "Usage: SVDSS call --reference <FASTA> --bam <BAM> --sfs <SFS>\n"
    "      --threads <int>             kept for output-order compatibility (default: 4)\n"
    "      --min-cluster-weight <int>  minimum number of supporting superstrings for a call (default: 2)\n"
    "      --min-sv-length <int>       minimum length of reported SVs (default: 25, values < 25 ignored)\n"
    "      --min-mapq <int>            minimum mapping quality (default: 20)\n"
    "      --poa <FILE>                store POA consensus alignments in .sam format to this file\n"
    "      --clusters <FILE>           store clusters to this file\n"
    "      -l <float>                  minimum length ratio for sub-clusters and chain merging (default: 0.97)\n"
    "      --noht                      ignore the HP tag\n"
    "      --clipped                   also call imprecise SVs from soft-clipped alignments (EXPERIMENTAL)\n";

static const char* SEARCH_USAGE =
    "Usage: SVDSS search --index <FMD> --bam <BAM> | --fastx <FASTA/FASTQ>\n"
    "      --threads <int>   kept for output-order compatibility (default: 4)\n"
    "      --bsize <int>     batch size (default: 10000)\n"
    "      --noputative      search all reads, not only XF == 0\n"
    "      --noassemble      do not merge overlapping specific strings\n";

static void logmsg(const char* lvl, const std::string& m) {
  time_t t = time(nullptr);
  char ts[32];
  strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&t));
  fprintf(stderr, "[%s] [stderr] [%s] %s\n", ts, lvl, m.c_str());
}

[[noreturn]] static void die(const std::string& m) {
  logmsg("critical", m);
  exit(EXIT_FAILURE);
}

static void check(int rc, const char* what) {
  if (rc != SVDSS_OK) die(std::string(what) + ": " + svdss_strerror(rc) + " " + svdss_last_hip_error());
}

// seq_nt16_str of htslib, then seq_nt6_table (ping_pong.cpp:90-94)
static const char NT16[] = "=ACMGRSVTWYHKDBN";

static Options parse(int argc, char** argv) {
  Options o;
  std::string err;
  if (!parse_options(argc, argv, 2, o, err)) die(err);
  if (o.gpus_all) o.gpus = svdss_device_count();
  return o;
}

// ---------------------------------------------------------------- index

static int main_index(int argc, char** argv) {
  // ropebwt3 `build` flags as run_svdss:142 passes them: -t T -d <fasta> -o <out>
  std::string fasta, out;
  int threads = 4;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
    else if (!strncmp(argv[i], "-t", 2) && argv[i][2]) threads = atoi(argv[i] + 2);
    else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
    else if (!strcmp(argv[i], "-d") || !strcmp(argv[i], "-b")) continue;  // output-format switches of ropebwt3
    else if (argv[i][0] == '-' && argv[i][1] == 'd' && argv[i][2]) continue;
    else if (argv[i][0] != '-') fasta = argv[i];
  }
  if (fasta.empty() || out.empty()) {
    fprintf(stderr, "Usage: SVDSS index [-t threads] -d <reference.fa[.gz]> -o <reference.fmd>\n");
    return EXIT_FAILURE;
  }
  // The index is built in HBM.  Without a GPU the command fails instead of quietly taking the host builder (which
  // stays in the library for the texts the GPU builder refuses, and for SVDSS_INDEX_CPU=1: developer runs on a
  // machine without a GPU).
  if (svdss_device_count() <= 0 && !getenv("SVDSS_INDEX_CPU"))
    die("no GPU found: SVDSS index builds the index in HBM (SVDSS_INDEX_CPU=1 runs the host builder instead)");
  const bool dbg = getenv("SVDSS_DEBUG") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (dbg) fprintf(stderr, "[index] %-28s at +%.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  };
  std::vector<uint8_t> cat;
  std::vector<int64_t> lens;
  {
    // a plain FASTA is mapped and read by several threads (fastx_reader.h, as `call` and `smooth` do), the records encoded
    // side by side into one buffer sized once; gzip / CRLF / FASTQ-like files go through the line reader as before
    std::vector<std::string> nm, sq;
    if (!getenv("SVDSS_FASTA_SERIAL") && load_fasta_mapped(fasta, std::max(1, std::min(threads < 8 ? 8 : threads, 16)), false, nm, sq)) {
      std::vector<size_t> at(sq.size() + 1, 0);
      for (size_t i = 0; i < sq.size(); ++i) { at[i + 1] = at[i] + sq[i].size(); lens.push_back((int64_t)sq[i].size()); }
      cat.resize(at.back());
      std::vector<std::thread> th;
      std::atomic<size_t> next(0);
      std::atomic<int> bad(0);
      for (int t = 0; t < (int)std::min<size_t>(sq.size(), 8); ++t)
        th.emplace_back([&] {
          for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= sq.size()) return;
            if (svdss_nt6_encode(sq[i].data(), (int64_t)sq[i].size(), cat.data() + at[i]) != SVDSS_OK) bad = 1;
            std::string().swap(sq[i]);
          }
        });
      for (std::thread& x : th) x.join();
      if (bad.load()) die("svdss_nt6_encode failed");
    } else {
      FastxReader fx(fasta);
      if (!fx.ok()) die("cannot open " + fasta);
      std::string name, seq;
      while (fx.next(name, seq)) {
        const size_t o = cat.size();
        cat.resize(o + seq.size());
        check(svdss_nt6_encode(seq.data(), (int64_t)seq.size(), cat.data() + o), "svdss_nt6_encode");
        lens.push_back((int64_t)seq.size());
      }
    }
  }
  if (lens.empty()) die("no sequence in " + fasta);
  mark("FASTA read + nt6");
  logmsg("info", "Indexing " + std::to_string(lens.size()) + " record(s), " + std::to_string(cat.size()) + " bases..");
  svdss_index_t* ix = nullptr;
  check(svdss_index_build(cat.data(), lens.data(), (int32_t)lens.size(), threads, &ix), "svdss_index_build");
  mark("index built");
  // the file ropebwt3 build -d writes (rld0), so that the index serves upstream SVDSS as well -- and beside it the
  // records themselves (nt6), from which `search` rebuilds the index in HBM in less time than the text + suffix array
  // (19 bytes per base) take to read from any disk.  SVDSS_INDEX_FULL=1: the full layout instead (a plain read).
  // (the records sidecar is written beside the rld0 encoding, by a second thread: the two read different parts of the
  // index -- the BWT, the records -- and the encoder is one serial pass over six billion symbols at human scale)
  int rc_side = SVDSS_OK;
  std::thread side;
  const bool records_side = !getenv("SVDSS_INDEX_NO_CACHE") && !getenv("SVDSS_INDEX_FULL");
  // (an older sidecar must not outlive a failed rewrite of its .fmd: it goes first, and a failure takes the .tmp with it)
  if (!getenv("SVDSS_INDEX_NO_CACHE")) (void)unlink((out + ".svdss").c_str());
  // (round 6: behind the records the rank blocks -- the index as a rank structure alone, what a `search` with few reads to
  // search makes resident instead of rebuilding everything: svdss_index_attach_blocks; SVDSS_INDEX_NO_BLOCKS=1: records only)
  if (records_side) side = std::thread([&] {
    rc_side = svdss_index_save_records(ix, (out + ".svdss.tmp").c_str());
    if (rc_side == SVDSS_OK && !getenv("SVDSS_INDEX_NO_BLOCKS")) rc_side = svdss_index_append_blocks(ix, (out + ".svdss.tmp").c_str());
  });
  const int rc_fmd = svdss_index_save_fmd(ix, out.c_str());
  if (side.joinable()) side.join();
  if (rc_fmd != SVDSS_OK || rc_side != SVDSS_OK) (void)unlink((out + ".svdss.tmp").c_str());
  check(rc_fmd, "svdss_index_save_fmd");
  mark("rld0 .fmd written");
  if (records_side) {
    check(rc_side, "svdss_index_save_records");
    // (renamed into place after the .fmd is complete: a sidecar is trusted only if it is not older than its .fmd)
    if (rename((out + ".svdss.tmp").c_str(), (out + ".svdss").c_str()) != 0 || utime((out + ".svdss").c_str(), nullptr) != 0)
      die("cannot write " + out + ".svdss");
  } else if (!getenv("SVDSS_INDEX_NO_CACHE")) {
    check(svdss_index_save(ix, (out + ".svdss").c_str()), "svdss_index_save");
  }
  mark("sidecar written");
  svdss_index_free(ix);
  return 0;
}

// --------------------------------------------------------------- search
//
// Three stages run concurrently, connected by bounded queues: (1) BGZF inflate + record parsing + nt6
// encoding into GPU-ready batches, (2) the GPU search of one batch, (3) formatting and writing the text of
// the previous batch.  The reference interleaves the same work inside one OpenMP loop
// (ping_pong.cpp:329-376: thread 0 loads and prints while the others search).

struct Read {
  std::string name;
  int hp = 0;
  int64_t len = 0;
  int64_t first = 0, count = 0;  // into the result arrays (-1: not searched)
};

// page-locked staging buffers (svdss_host_alloc), recycled between batches
struct PinnedPool {
  std::mutex m;
  std::vector<std::pair<uint8_t*, size_t>> free_;
  uint8_t* get(size_t bytes, size_t& cap) {
    {
      std::lock_guard<std::mutex> lk(m);
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].second >= bytes) {
          uint8_t* p = free_[i].first;
          cap = free_[i].second;
          free_.erase(free_.begin() + (long)i);
          return p;
        }
    }
    void* p = nullptr;
    cap = bytes + bytes / 8 + 4096;
    check(svdss_host_alloc((int64_t)cap, &p), "svdss_host_alloc");
    return (uint8_t*)p;
  }
  void put(uint8_t* p, size_t cap) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(m);
    free_.emplace_back(p, cap);
  }
  ~PinnedPool() { for (auto& f : free_) svdss_host_free(f.first); }
};

struct SearchBatch {
  uint64_t seq = 0;              // position in the input: batches are written in this order
  std::vector<Read> reads;
  std::vector<uint8_t> gbuf;     // nt6 bases of the searched reads, back to back (FASTX mode)
  // BAM mode: the 4-bit bases exactly as the records hold them, in page-locked memory; the GPU expands them
  uint8_t* seq4 = nullptr;
  size_t seq4_cap = 0;
  std::vector<int64_t> boff;     // byte offset of every searched read in seq4 (+ end)
  std::vector<int32_t> lseq;
  // the record views of the batch and the inflated chunks they point into, until the searching thread has copied the
  // packed bases out of them
  std::vector<BamReader::RawView> recs;
  std::vector<std::shared_ptr<BamReader::Bytes>> keep;
  std::vector<int64_t> goff;
  std::vector<size_t> gidx;      // searched read -> index into reads
  std::vector<int32_t> qs, ln;   // results
  std::vector<int64_t> counts;
  std::string text;              // the batch's lines, formatted by the thread that searched it
  uint64_t n_lines = 0;
};

// decimal text of v at w, returns the end
inline char* put_int(char* w, int64_t v) {
  if (v < 0) { *w++ = '-'; v = -v; }
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *w++ = tmp[--n];
  return w;
}

template <class T>
class BoundedQueue {
 public:
  explicit BoundedQueue(size_t cap) : cap_(cap) {}
  void push(std::unique_ptr<T> v) {
    std::unique_lock<std::mutex> lk(m_);
    not_full_.wait(lk, [&] { return q_.size() < cap_; });
    q_.push_back(std::move(v));
    not_empty_.notify_one();
  }
  // nullptr = the producer closed the queue and it is drained
  std::unique_ptr<T> pop() {
    std::unique_lock<std::mutex> lk(m_);
    not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return nullptr;
    std::unique_ptr<T> v = std::move(q_.front());
    q_.pop_front();
    not_full_.notify_one();
    return v;
  }
  void close() {
    std::lock_guard<std::mutex> lk(m_);
    closed_ = true;
    not_empty_.notify_all();
  }

 private:
  size_t cap_;
  std::deque<std::unique_ptr<T>> q_;
  std::mutex m_;
  std::condition_variable not_full_, not_empty_;
  bool closed_ = false;
};

static time_t g_t0 = 0;   // process start (the final log line)

// the text of one batch.  output_batch order: reference batches of bsize reads -> thread t takes reads n with
// n % T == t (ping_pong.cpp:59,101-104) -> std::map<qname, vector<SFS>> order (:217)
static void format_batch(const Options& o, SearchBatch& b) {
  const std::vector<Read>& reads = b.reads;
  std::string& out = b.text;
  out.reserve(b.qs.size() * 24 + 1024);
  char num[64];
  for (size_t b0 = 0; b0 < reads.size(); b0 += (size_t)o.bsize) {
    const size_t b1 = std::min(reads.size(), b0 + (size_t)o.bsize);
    for (int t = 0; t < o.threads; ++t) {
      std::map<std::string, std::vector<size_t>> by_name;
      for (size_t n = b0 + (size_t)t; n < b1; n += (size_t)o.threads)
        if (reads[n].count >= 0) by_name[reads[n].name].push_back(n);
      for (const auto& kv : by_name) {
        bool first = true;
        for (size_t n : kv.second) {
          const Read& r = reads[n];
          for (int64_t k = 0; k < r.count; ++k) {
            if (first) out += r.name; else out += '*';
            char* w = num;                      // "\t<qs>\t<len>\t<hp>\t\n" without printf (11 M lines per GB of reads)
            *w++ = '\t'; w = put_int(w, b.qs[(size_t)(r.first + k)]);
            *w++ = '\t'; w = put_int(w, b.ln[(size_t)(r.first + k)]);
            *w++ = '\t'; w = put_int(w, r.hp);
            *w++ = '\t'; *w++ = '\n';
            out.append(num, (size_t)(w - num));
            first = false;
            ++b.n_lines;
          }
        }
      }
    }
  }
}

// ---- `search --bam` with the records handled where they are inflated (csrc/bam_device.hip): the host reads the file,
// finds the BGZF members, hands runs of them to the GPUs and gets names, tags and SFS back.  Stages, PER REGION of the
// file: scanner (loader threads) -> batcher -> feeding threads (svdss_bam_batch_run, one batch object each); then, once
// for the file: assembler (device batches end where a BGZF member ends; the text is defined on batches of --bsize reads,
// ping_pong.cpp:213-236: the reads are dealt again into units of whole reference batches) -> formatting threads ->
// writer.  The same bytes as the host path.
//
// --gpus N (north_star: "BAM regions partition across the GPUs"; the per-shard loop of ping_pong.cpp:53-128): the file is
// cut at BGZF members into N regions of about equal size, every GPU reads, inflates, walks and searches its own region
// with its own scanner, batcher and feeders -- nothing is shared on the way in.  A region that does not begin the file
// begins inside a record: its first batch starts the record chain at the first record its segments guess
// (svdss_bam_stream_region) and sets the bytes in front aside; when the region before it has ended, what that one left
// over and those bytes go through the device as a batch of their own (the SEAM: normally one record): if they are a
// chain of whole records the guess is proved, if not -- or if the region failed in any way -- the region runs again from
// the known carry.  The reads of a region are dealt into units when everything before it has been (the unit a read
// belongs to depends on the reads in front of it), so the later regions' results wait in memory (~0.6 KB per read).
struct DevJob { uint64_t seq = 0; bool last = false; std::vector<std::unique_ptr<CompChunk>> chunks; };
struct DevOut { std::vector<Read> reads; std::vector<int32_t> qs, ln; int64_t n_short = 0; std::vector<int32_t> sidx; };
struct BamRegion;
// `SVDSS search` with the BAM front end started BEFORE the index is resident (include/svdss_hip.h, svdss_bam_park_*): while
// `ix` is null the feeders run the front half of their batches and park the unpacked reads in HBM; when the index is there the
// parked groups are searched one large launch each, and the feeders go on with whole batches.
struct EarlySearch {
  svdss_bam_park_t* park = nullptr;
  std::mutex m;
  std::condition_variable cv;
  svdss_index_t* ix = nullptr;      // set once, with `ready` -- or before it, with `ix_avail`
  bool ready = false;
  bool ix_avail = false;            // the index is resident but held back from the feeders (the rank blocks alone): the drain
                                    // thread may search the groups that have closed while the later ones still fill
  struct Pending { BamRegion* R; uint64_t seq; std::unique_ptr<DevOut> out; int64_t first, n; };
  std::map<int64_t, std::vector<Pending>> by_group;     // under m
  // what the front end has seen so far (the order of the k-mer table is chosen from it: svdss_index_kmer_limit)
  std::atomic<int64_t> records{0}, searched{0}, comp_bytes{0}, index_n{0};
  // (under m) every feeding thread has ended / a batch did not fit into the park: an index that is made resident early --
  // the rank structure alone -- is held back until one of the two, so that the parked reads go in large launches
  bool front_done = false, park_full = false;
  int64_t file_bytes = 0;
  int kmer_limit = 0;               // the last limit given (under m)
};
struct BamRegion {
  size_t begin = 0, end = 0;                 // file range (member starts)
  std::unique_ptr<BgzfScanner> sc;
  std::vector<svdss_index_t*> gpus;          // the replicas whose feeders take this region's batches
  svdss_bam_stream_t* stream = nullptr;
  int64_t skip = 0;                          // inflated bytes of BAM header in front (the file's first region)
  bool open_start = false, open_end = false;
  std::unique_ptr<BoundedQueue<DevJob>> jobs;
  std::vector<std::thread> threads;          // batcher + feeders
  // under dev_m:
  std::map<uint64_t, std::unique_ptr<DevOut>> done;
  bool finished = false;                     // its threads have ended
  bool head_known = false;                   // batch 0 had its turn (svdss_bam_stream_head is final)
  std::string error;                         // open_start only: why the run failed (the region runs again)
};

// (plan_bam_regions: bam_device_select.h -- where the regions of a file begin, shared with `SVDSS call --gpus N`)

static void search_bam_device(const Options& o, const std::vector<svdss_index_t*>& replicas, std::vector<BamRegion>& regions,
                              const BgzfScanner::Hooks& hooks, size_t slab, int loaders, size_t pool_chunks, int32_t n_ref,
                              const std::function<std::string()>& since, EarlySearch* early = nullptr) {
  const int64_t super = std::max<int64_t>(o.bsize, 32768 / o.bsize * (int64_t)o.bsize);
  const int64_t target = (getenv("SVDSS_BAM_BATCH_MB") && atoll(getenv("SVDSS_BAM_BATCH_MB")) > 0 ? atoll(getenv("SVDSS_BAM_BATCH_MB")) : 192) << 20;
  const int per_gpu = getenv("SVDSS_SEARCH_FEEDERS") ? std::max(1, atoi(getenv("SVDSS_SEARCH_FEEDERS"))) : 6;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  std::mutex t_m;
  double t_gpu = 0, t_inflate_ms = 0, t_build = 0, t_format = 0, t_write = 0, t_assemble = 0;
  double t_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double t_wait_file = 0, t_wait_gpu = 0;       // (the batchers')
  uint64_t n_seen = 0, n_batches = 0, total_sfs = 0;
  int64_t n_seg_all = 0, n_rewalk_all = 0, n_seams = 0, n_reruns = 0;
  const int32_t flags = (o.assemble ? SVDSS_SFS_ASSEMBLE : 0) | (o.putative ? SVDSS_BAM_PUTATIVE : 0);

  // device batches in file order, region after region
  std::mutex dev_m;
  std::condition_variable dev_cv;
  size_t cursor = 0;            // the region the assembler is taking batches from (under dev_m)

  // what a batch object holds after its run -> reads with their SFS
  auto collect = [&](svdss_bam_batch_t* batch, double gpu_s, std::chrono::steady_clock::time_point t1, bool searched = true) {
    svdss_bam_result_t r;
    check(svdss_bam_batch_result(batch, &r), "svdss_bam_batch_result");
    std::unique_ptr<DevOut> out(new DevOut);
    out->n_short = r.n_short;
    out->reads.resize((size_t)r.n_slots);
    if (!searched) {
      // the front half only: names and tags; counts and SFS follow when the batch's group has been searched (fill_parked)
      out->sidx.assign(r.sidx, r.sidx + r.n_slots);
      for (int64_t i = 0; i < r.n_slots; ++i) {
        Read& rd = out->reads[(size_t)i];
        rd.name.assign(r.names + r.name_off[i], (size_t)(r.name_off[i + 1] - r.name_off[i]));
        rd.hp = r.hp[i];
        rd.count = r.sidx[i] < 0 ? -1 : 0; rd.first = 0;
      }
    } else {
    out->qs.assign(r.qs, r.qs + r.total_sfs);
    out->ln.assign(r.len, r.len + r.total_sfs);
    {
      // (searched reads are numbered in slot order, so their SFS follow each other in slot order too)
      int64_t acc = 0;
      for (int64_t i = 0; i < r.n_slots; ++i) {
        Read& rd = out->reads[(size_t)i];
        rd.name.assign(r.names + r.name_off[i], (size_t)(r.name_off[i + 1] - r.name_off[i]));
        rd.hp = r.hp[i];
        if (r.sidx[i] < 0) { rd.count = -1; rd.first = acc; }
        else { rd.first = acc; rd.count = r.counts[r.sidx[i]]; acc += rd.count; }
      }
    }
    }
    std::lock_guard<std::mutex> lk(t_m);
    t_gpu += gpu_s; t_build += secs(t1, now()); t_inflate_ms += r.inflate_kernel_ms;
    n_seen += (uint64_t)r.n_records; ++n_batches;
    for (int k = 0; k < 8; ++k) t_stage[k] += r.stage_ms[k] * 1e-3;
    return out;
  };

  auto batcher = [&](BamRegion& R) {
    BgzfScanner& sc = *R.sc;
    std::unique_ptr<DevJob> cur(new DevJob);
    int64_t acc = 0;
    uint64_t seq = 0;
    bool any_last = false;
    double w_file = 0, w_gpu = 0;
    for (;;) {
      const auto w0 = now();
      std::unique_ptr<CompChunk> c = sc.next();
      w_file += secs(w0, now());           // (the loaders behind: the file is what bounds the run)
      if (!c) break;
      acc += c->inflated;
      const bool last = c->last;
      cur->chunks.push_back(std::move(c));
      if (acc >= target || last) {
        cur->seq = seq++;
        cur->last = last;
        any_last = any_last || last;
        const auto w1 = now();
        R.jobs->push(std::move(cur));
        w_gpu += secs(w1, now());          // (the feeding threads behind: the GPU side is)
        cur.reset(new DevJob);
        acc = 0;
      }
    }
    if (!sc.error().empty()) {
      if (!R.open_start) die("error reading " + o.bam + ": " + sc.error());
      std::lock_guard<std::mutex> lk(dev_m);
      if (R.error.empty()) R.error = sc.error();
    } else if (!any_last) {            // (an empty region; cannot happen otherwise: the scanner marks the final slab)
      cur->seq = seq++;
      cur->last = true;
      R.jobs->push(std::move(cur));
    }
    R.jobs->close();
    std::lock_guard<std::mutex> lk(t_m);
    t_wait_file += w_file; t_wait_gpu += w_gpu;
  };

  auto feeder = [&](BamRegion& R, size_t g, svdss_index_t* ix) {
    svdss_bam_batch_t* batch = nullptr;
    std::vector<const uint8_t*> comp;
    std::vector<int64_t> comp_bytes, n_blocks;
    std::vector<const svdss_bgzf_block_t*> blocks;
    std::vector<const uint32_t*> crcs;
    while (std::unique_ptr<DevJob> job = R.jobs->pop()) {
      const auto t0 = now();
      comp.clear(); comp_bytes.clear(); n_blocks.clear(); blocks.clear(); crcs.clear();
      for (const std::unique_ptr<CompChunk>& c : job->chunks) {
        comp.push_back(c->data); comp_bytes.push_back((int64_t)c->n_bytes); n_blocks.push_back((int64_t)c->blocks.size());
        blocks.push_back(c->blocks.data()); crcs.push_back(c->crc.data());
      }
      // (early: while the index is being restored the front half of the batch runs and its reads are parked)
      bool front_only = false;
      if (early) {
        std::lock_guard<std::mutex> lk(early->m);
        if (early->ready) ix = early->ix; else front_only = true;
      }
      int64_t job_comp = 0;
      for (const std::unique_ptr<CompChunk>& c : job->chunks) job_comp += (int64_t)c->n_bytes;
      int rc = front_only
                   ? svdss_bam_batch_front(R.stream, (int64_t)job->seq, job->last ? 1 : 0, job->seq == 0 ? R.skip : 0, 0, early->park, (int32_t)comp.size(),
                                           comp.data(), comp_bytes.data(), blocks.data(), crcs.data(), n_blocks.data(), flags, &batch)
                   : svdss_bam_batch_run(R.stream, (int64_t)job->seq, job->last ? 1 : 0, job->seq == 0 ? R.skip : 0, ix, (int32_t)comp.size(),
                                         comp.data(), comp_bytes.data(), blocks.data(), crcs.data(), n_blocks.data(), flags, &batch);
      for (std::unique_ptr<CompChunk>& c : job->chunks) R.sc->recycle(std::move(c));
      if (rc == SVDSS_OK && front_only) {
        int64_t grp = -1, first = 0, n_srch = 0;
        check(svdss_bam_batch_parked(batch, &grp, &first, &n_srch), "svdss_bam_batch_parked");
        {
          // how much there will be to search, from what has been seen: the order of the k-mer table (its build begins when the
          // suffix array is sorted; the limit is read then)
          svdss_bam_result_t r0;
          check(svdss_bam_batch_result(batch, &r0), "svdss_bam_batch_result");
          const int64_t recs = (early->records += r0.n_records), srch = (early->searched += r0.n_searched), cb = (early->comp_bytes += job_comp);
          const int64_t n_ix = early->index_n.load();
          if (n_ix >= ((int64_t)1 << 31) && recs >= 50000 && cb > 0 && !getenv("SVDSS_KMER") && !getenv("SVDSS_NO_KMER_LIMIT")) {
            const double est = (double)srch / (double)recs * ((double)recs * (double)early->file_bytes / (double)cb);
            // build: 1.6 s at K = 16, a quarter of that per step down; kernel: 16 M reads/s at K = 16, half of that per step down
            // (profiles/r05i_restore_by_table_order.txt); its seconds count double, as in main_search
            auto cost = [&](int k) { return 1.6 * std::pow(4.0, k - 16) + 2 * est / 16e6 * std::pow(2.2, 16 - k); };
            int best = 16;
            for (int k = 15; k >= 12; --k) if (cost(k) < cost(best)) best = k;
            if (cost(best) > 0.8 * cost(16)) best = 16;     // (a clear gain or none)
            std::lock_guard<std::mutex> lk(early->m);
            if (best != early->kmer_limit) { early->kmer_limit = best; svdss_index_kmer_limit(best == 16 ? 0 : best); }
          }
        }
        if (grp >= 0) {
          const auto t1 = now();
          std::unique_ptr<DevOut> out = collect(batch, secs(t0, t1), t1, false);
          { std::lock_guard<std::mutex> lk(early->m); early->by_group[grp].push_back(EarlySearch::Pending{&R, job->seq, std::move(out), first, n_srch}); }
          early->cv.notify_all();
          continue;
        }
        if (grp == -1) {
          // no room in the park (or it has just been closed): this batch waits here for the index
          { std::unique_lock<std::mutex> lk(early->m); early->park_full = true; early->cv.notify_all(); early->cv.wait(lk, [&] { return early->ready; }); ix = early->ix; }
          rc = svdss_bam_batch_search(batch, ix);
        }   // (grp == -2: nothing to search in this batch, its results are complete)
      }
      if (rc != SVDSS_OK) {
        std::string msg = batch ? svdss_bam_batch_error(batch) : "";
        if (msg.empty()) msg = svdss_bam_stream_error(R.stream);
        if (R.open_start) {
          // (a guess that was not a record can end in any of the messages below: the region runs again from the known
          // carry and says then what a reader of the whole file would have said)
          { std::lock_guard<std::mutex> lk(dev_m); if (R.error.empty()) R.error = msg.empty() ? svdss_strerror(rc) : msg; R.head_known = true; }
          dev_cv.notify_all();
          continue;   // (the stream has failed: the batches left return at once; the queue drains)
        }
        if (msg.find("core.tid") != std::string::npos) die(msg);                       // ping_pong.cpp:76-79
        if (rc == SVDSS_EIO) die("error reading " + o.bam + ": " + msg);
        die(std::string("svdss_bam_batch_run: ") + svdss_strerror(rc) + " " + msg + " " + svdss_last_hip_error());
      }
      const auto t1 = now();
      std::unique_ptr<DevOut> out = collect(batch, secs(t0, t1), t1);
      {
        std::unique_lock<std::mutex> lk(dev_m);
        const uint64_t sq = job->seq;
        // (only the region the assembler is at holds its feeders back; the others' results wait for their turn)
        dev_cv.wait(lk, [&] { return cursor != g || R.done.size() < 8 || R.done.begin()->first > sq; });
        R.done[sq] = std::move(out);
        R.head_known = true;
      }
      dev_cv.notify_all();
    }
    svdss_bam_batch_free(batch);
  };

  auto launch = [&](BamRegion& R, size_t g, const uint8_t* carry, int64_t n_carry) {
    if (!R.sc) R.sc.reset(new BgzfScanner(o.bam, hooks, slab, loaders, pool_chunks, R.begin, R.end));
    if (!R.sc->ok()) die("cannot open " + o.bam);
    check(svdss_bam_stream_create(n_ref, &R.stream), "svdss_bam_stream_create");
    check(svdss_bam_stream_region(R.stream, R.open_start ? 1 : 0, R.open_end ? 1 : 0, carry, n_carry), "svdss_bam_stream_region");
    R.jobs.reset(new BoundedQueue<DevJob>(2));
    R.threads.emplace_back(batcher, std::ref(R));
    for (svdss_index_t* ix : R.gpus)
      for (int k = 0; k < per_gpu; ++k) R.threads.emplace_back(feeder, std::ref(R), g, ix);
  };
  auto join_region = [&](BamRegion& R) {
    for (std::thread& th : R.threads) th.join();
    R.threads.clear();
    int64_t n_seg = 0;
    const int64_t rew = svdss_bam_stream_rewalked(R.stream, &n_seg);
    { std::lock_guard<std::mutex> lk(t_m); n_seg_all += n_seg; n_rewalk_all += rew; }
    if (early) return;      // (its parked batches are still to come: finished when they have been searched and handed over)
    { std::lock_guard<std::mutex> lk(dev_m); R.finished = true; }
    dev_cv.notify_all();
  };

  // the seam in front of region g: `tail` (what the region before left) + `head` (what region g set aside) as a stream of
  // its own -- stored deflate blocks, the same entry point.  false: not a chain of whole records (the guess was wrong).
  auto run_seam = [&](const uint8_t* tail, int64_t n_tail, const uint8_t* head, int64_t n_head, svdss_index_t* ix, std::unique_ptr<DevOut>& out) {
    std::vector<uint8_t> bytes((size_t)(n_tail + n_head));
    if (n_tail) memcpy(bytes.data(), tail, (size_t)n_tail);
    if (n_head) memcpy(bytes.data() + n_tail, head, (size_t)n_head);
    std::vector<uint8_t> comp;
    std::vector<svdss_bgzf_block_t> blk;
    std::vector<uint32_t> crc;
    for (size_t off = 0; off < bytes.size(); off += 0xff00) {
      const size_t len = std::min<size_t>(0xff00, bytes.size() - off);
      while (comp.size() & 15) comp.push_back(0);
      svdss_bgzf_block_t b;
      b.coff = (int64_t)comp.size(); b.clen = (int32_t)(5 + len); b.isize = (int32_t)len; b.uoff = 0;
      comp.push_back(1);   // BFINAL, stored
      comp.push_back((uint8_t)(len & 0xff)); comp.push_back((uint8_t)(len >> 8));
      comp.push_back((uint8_t)(~len & 0xff)); comp.push_back((uint8_t)((~len >> 8) & 0xff));
      comp.insert(comp.end(), bytes.begin() + (long)off, bytes.begin() + (long)(off + len));
      blk.push_back(b);
      crc.push_back((uint32_t)crc32(crc32(0L, Z_NULL, 0), bytes.data() + off, (uInt)len));
    }
    comp.resize(comp.size() + 64);
    svdss_bam_stream_t* st = nullptr;
    check(svdss_bam_stream_create(n_ref, &st), "svdss_bam_stream_create");
    svdss_bam_batch_t* batch = nullptr;
    const uint8_t* cp = comp.data();
    const int64_t cb = (int64_t)comp.size(), nb = (int64_t)blk.size();
    const svdss_bgzf_block_t* bp = blk.data();
    const uint32_t* rp = crc.data();
    const auto t0 = now();
    const int rc = svdss_bam_batch_run(st, 0, 1, 0, ix, 1, &cp, &cb, &bp, &rp, &nb, flags, &batch);
    bool ok = rc == SVDSS_OK;
    if (ok) { const auto t1 = now(); out = collect(batch, secs(t0, t1), t1); }
    else {
      const std::string msg = batch ? svdss_bam_batch_error(batch) : "";
      if (rc != SVDSS_EIO) die(std::string("svdss_bam_batch_run (seam): ") + svdss_strerror(rc) + " " + msg + " " + svdss_last_hip_error());
    }
    svdss_bam_batch_free(batch);
    svdss_bam_stream_free(st);
    return ok;
  };

  // units of whole reference batches, formatted by a few threads, written in order
  BoundedQueue<SearchBatch> units(4);
  std::mutex done_m;
  std::condition_variable done_cv;
  std::map<uint64_t, std::unique_ptr<SearchBatch>> done;
  bool format_finished = false;
  std::mutex pool_m;
  std::vector<std::unique_ptr<SearchBatch>> batch_pool;
  auto new_unit = [&]() {
    std::unique_ptr<SearchBatch> b;
    {
      std::lock_guard<std::mutex> lk(pool_m);
      if (!batch_pool.empty()) { b = std::move(batch_pool.back()); batch_pool.pop_back(); }
    }
    if (!b) b.reset(new SearchBatch);
    b->reads.clear(); b->qs.clear(); b->ln.clear(); b->text.clear(); b->n_lines = 0; b->seq = 0;
    return b;
  };
  std::thread assembler([&] {
    uint64_t unit_seq = 0;
    std::unique_ptr<SearchBatch> unit = new_unit();
    auto deal = [&](DevOut& d) {
      const auto ta = now();
      // (said when the batch is dealt, not when it was read: a region that runs twice says it once)
      for (int64_t k = 0; k < d.n_short; ++k) logmsg("warning", "Alignment filtered due to l_qseq. Why are we here? Please check");   // :70-75
      for (Read& r : d.reads) {
        const int64_t first = r.first;
        r.first = (int64_t)unit->qs.size();
        if (r.count > 0) {
          unit->qs.insert(unit->qs.end(), d.qs.begin() + first, d.qs.begin() + first + r.count);
          unit->ln.insert(unit->ln.end(), d.ln.begin() + first, d.ln.begin() + first + r.count);
        }
        unit->reads.push_back(std::move(r));
        if ((int64_t)unit->reads.size() == super) {
          unit->seq = unit_seq++;
          units.push(std::move(unit));
          unit = new_unit();
        }
      }
      t_assemble += secs(ta, now());
    };
    for (size_t g = 0; g < regions.size(); ++g) {
      BamRegion& R = regions[g];
      std::thread rerun_joiner;
      if (g > 0) {
        // the region in front has ended and is dealt: its tail is final.  This region's head: after its first batch.
        BamRegion& P = regions[g - 1];
        const uint8_t *tail = nullptr, *head = nullptr;
        const int64_t n_tail = svdss_bam_stream_tail(P.stream, &tail);
        bool good;
        {
          std::unique_lock<std::mutex> lk(dev_m);
          dev_cv.wait(lk, [&] { return R.head_known || R.finished; });
          good = R.error.empty();
        }
        std::unique_ptr<DevOut> seam;
        if (good) {
          const int64_t n_head = svdss_bam_stream_head(R.stream, &head);
          if (n_tail + n_head > 0) { good = run_seam(tail, n_tail, head, n_head, R.gpus[0], seam); ++n_seams; }
        }
        if (!good) {
          // not proved (or the region failed): once more, from the record the region before ended in
          if (o.verbose) logmsg("debug", "region " + std::to_string(g) + " runs again from the end of region " + std::to_string(g - 1) +
                                             (R.error.empty() ? std::string(" (its first record was not where the chain arrives)") : " (" + R.error + ")"));
          {
            std::unique_lock<std::mutex> lk(dev_m);
            dev_cv.wait(lk, [&] { return R.finished; });   // (its first run: what is left of it ends at once, the stream has failed)
            R.done.clear(); R.error.clear(); R.finished = false; R.head_known = false; R.open_start = false;
          }
          svdss_bam_stream_free(R.stream);
          R.stream = nullptr;
          R.sc.reset();
          ++n_reruns;
          launch(R, g, tail, n_tail);
          rerun_joiner = std::thread([&join_region, &R] { join_region(R); });
        } else if (seam) deal(*seam);
        svdss_bam_stream_free(P.stream);
        P.stream = nullptr;
      }
      { std::lock_guard<std::mutex> lk(dev_m); cursor = g; }
      dev_cv.notify_all();
      for (uint64_t want = 0;; ++want) {
        std::unique_ptr<DevOut> d;
        {
          std::unique_lock<std::mutex> lk(dev_m);
          dev_cv.wait(lk, [&] { return R.done.count(want) || !R.error.empty() || (R.finished && R.done.empty()); });
          if (!R.error.empty()) die(R.error.find("core.tid") != std::string::npos ? R.error : "error reading " + o.bam + ": " + R.error);   // (after the proof: the file's own fault)
          auto it = R.done.find(want);
          if (it == R.done.end()) break;
          d = std::move(it->second);
          R.done.erase(it);
        }
        dev_cv.notify_all();
        deal(*d);
      }
      if (rerun_joiner.joinable()) rerun_joiner.join();
    }
    if (!unit->reads.empty()) { unit->seq = unit_seq++; units.push(std::move(unit)); }
    units.close();
  });
  auto formatter = [&] {
    while (std::unique_ptr<SearchBatch> u = units.pop()) {
      const auto tf = now();
      format_batch(o, *u);
      { std::lock_guard<std::mutex> lk(t_m); t_format += secs(tf, now()); }
      {
        std::unique_lock<std::mutex> lk(done_m);
        const uint64_t sq = u->seq;
        done_cv.wait(lk, [&] { return done.size() < 8 || done.begin()->first > sq; });
        done[sq] = std::move(u);
      }
      done_cv.notify_all();
    }
  };
  std::thread writer([&] {
    uint64_t want = 0;
    for (;;) {
      std::unique_ptr<SearchBatch> bt;
      {
        std::unique_lock<std::mutex> lk(done_m);
        done_cv.wait(lk, [&] { return done.count(want) || (format_finished && done.empty()); });
        auto it = done.find(want);
        if (it == done.end()) break;
        bt = std::move(it->second);
        done.erase(it);
        ++want;
      }
      done_cv.notify_all();
      const auto tw0 = now();
      fwrite(bt->text.data(), 1, bt->text.size(), stdout);
      total_sfs += bt->n_lines;
      t_write += secs(tw0, now());
      std::lock_guard<std::mutex> lk(pool_m);
      if (batch_pool.size() < 16) batch_pool.push_back(std::move(bt));
    }
    fflush(stdout);
  });
  {
    // (formatting the text costs about one core-second per million reads: five threads per GPU, as many as the cores allow)
    const int n_fmt = getenv("SVDSS_FORMAT_THREADS") ? std::max(1, atoi(getenv("SVDSS_FORMAT_THREADS")))
                                                     : (int)std::max<size_t>(5, std::min<size_t>(5 * replicas.size(), effective_cpus()));
    std::vector<std::thread> fmt;
    for (int k = 0; k < n_fmt; ++k) fmt.emplace_back(formatter);
    for (size_t g = 0; g < regions.size(); ++g) launch(regions[g], g, nullptr, 0);
    // early: once the index is resident, the parked groups -- ONE launch each, one lane per read -- and their batches' results
    std::thread drain;
    if (early) drain = std::thread([&] {
      svdss_index_t* ix = nullptr;
      { std::unique_lock<std::mutex> lk(early->m); early->cv.wait(lk, [&] { return early->ready || early->ix_avail; }); ix = early->ix; }
      svdss_sfs_batch_t* sfs = nullptr;
      std::vector<int64_t> counts, prefix;
      std::vector<int32_t> qs, ln;
      int64_t n_parked = 0, n_parked_batches = 0, n_groups = 0, n_early_groups = 0;
      double t_search = 0;
      bool closed = false;
      for (int64_t g = 0;; ++g) {
        // the next group: one that has closed while the index is held back from the feeders, or -- once the feeders have the
        // index and the park is closed -- whatever is left
        for (;;) {
          if (!closed) {
            bool rdy;
            { std::lock_guard<std::mutex> lk(early->m); rdy = early->ready; }
            if (rdy) { check(svdss_bam_park_close(early->park), "svdss_bam_park_close"); closed = true; n_groups = svdss_bam_park_groups(early->park); }
          }
          if (closed || svdss_bam_park_group_ready(early->park, g)) break;
          std::unique_lock<std::mutex> lk(early->m);
          early->cv.wait_for(lk, std::chrono::milliseconds(2));
        }
        if (closed && g >= n_groups) break;
        if (!closed) ++n_early_groups;
        int64_t nb = 0, nr = 0, ns = 0;
        check(svdss_bam_park_group(early->park, g, &nb, &nr, &ns), "svdss_bam_park_group");
        const auto t0 = now();
        check(svdss_bam_park_search(early->park, g, ix, flags, &sfs), "svdss_bam_park_search");
        const int64_t total = svdss_sfs_batch_total(sfs);
        counts.resize((size_t)nr); qs.resize((size_t)total); ln.resize((size_t)total);
        check(svdss_sfs_batch_fetch(sfs, counts.data(), qs.data(), ln.data(), nullptr), "svdss_sfs_batch_fetch");
        t_search += secs(t0, now());
        prefix.assign((size_t)nr + 1, 0);
        for (int64_t i = 0; i < nr; ++i) prefix[(size_t)i + 1] = prefix[(size_t)i] + counts[(size_t)i];
        std::vector<EarlySearch::Pending> pend;
        {
          std::unique_lock<std::mutex> lk(early->m);
          early->cv.wait(lk, [&] { return (int64_t)early->by_group[g].size() == nb; });
          pend.swap(early->by_group[g]);
        }
        for (EarlySearch::Pending& P : pend) {
          DevOut& d = *P.out;
          int64_t acc = 0;
          for (size_t i = 0; i < d.reads.size(); ++i) {
            Read& rd = d.reads[i];
            rd.first = acc;
            if (d.sidx[i] < 0) { rd.count = -1; continue; }
            const size_t k = (size_t)(P.first + d.sidx[i]);
            rd.count = counts[k];
            d.qs.insert(d.qs.end(), qs.begin() + prefix[k], qs.begin() + prefix[k + 1]);
            d.ln.insert(d.ln.end(), ln.begin() + prefix[k], ln.begin() + prefix[k + 1]);
            acc += rd.count;
          }
          d.sidx.clear();
          { std::lock_guard<std::mutex> lk(dev_m); P.R->done[P.seq] = std::move(P.out); P.R->head_known = true; }
          dev_cv.notify_all();
        }
        n_parked += nr; n_parked_batches += nb;
      }
      if (sfs) svdss_sfs_batch_free(sfs);
      { std::lock_guard<std::mutex> lk(t_m); t_stage[5] += t_search; }
      if (o.verbose)
        logmsg("debug", "front end beside the index restore: " + std::to_string(n_parked_batches) + " batches (" + std::to_string(early->records.load()) +
                            " records) had been read when the index was resident; their " + std::to_string(n_parked) + " reads searched in " +
                            std::to_string(n_groups) + " launch(es), " + std::to_string(t_search) + " s" +
                            (n_early_groups ? " (" + std::to_string(n_early_groups) + " of them while the file was still being read)" : "") + ", done at +" + since() + " s");
    });
    std::vector<std::thread> joiners;
    for (BamRegion& R : regions) joiners.emplace_back([&join_region, &R] { join_region(R); });
    for (std::thread& th : joiners) th.join();
    if (early) { { std::lock_guard<std::mutex> lk(early->m); early->front_done = true; } early->cv.notify_all(); }
    if (drain.joinable()) {
      drain.join();
      { std::lock_guard<std::mutex> lk(dev_m); for (BamRegion& R : regions) R.finished = true; }
      dev_cv.notify_all();
    }
    assembler.join();
    for (std::thread& th : fmt) th.join();
    { std::lock_guard<std::mutex> lk(done_m); format_finished = true; }
    done_cv.notify_all();
    writer.join();
  }
  if (o.verbose) {
    logmsg("debug", std::to_string(n_seen) + " records read, " + std::to_string(total_sfs) + " SFS written at +" + since() + " s");
    if (regions.size() > 1)
      logmsg("debug", std::to_string(regions.size()) + " regions of the file, one per GPU: " + std::to_string(n_seams) + " seam(s) run, " +
                          std::to_string(n_reruns) + " region(s) run again");
    logmsg("debug", "device path: " + std::to_string(n_batches) + " batches, " + std::to_string(n_seg_all) + " segments (" + std::to_string(n_rewalk_all) +
                        " walked again); busy seconds: GPU batches " + std::to_string(t_gpu) + " (inflate kernels " + std::to_string(t_inflate_ms * 1e-3) +
                        "), result unpacking " + std::to_string(t_build) + ", re-dealing " + std::to_string(t_assemble) + ", format " + std::to_string(t_format) +
                        ", write " + std::to_string(t_write));
    char buf[480];
    snprintf(buf, sizeof buf, "device batches, seconds summed: upload+inflate+crc+walk %.3f, waiting for the turn %.3f, turn (carry, link) %.3f, "
             "fields+scans %.3f, unpack %.3f, search %.3f, results down %.3f; the batchers waited %.3f s for the file's loaders and %.3f s for the feeding threads",
             t_stage[0], t_stage[1], t_stage[2], t_stage[3], t_stage[4], t_stage[5], t_stage[6], t_wait_file, t_wait_gpu);
    logmsg("debug", buf);
  }
  for (BamRegion& R : regions) if (R.stream) { svdss_bam_stream_free(R.stream); R.stream = nullptr; }
}

int main_search(const Options& o) {
  logmsg("info", "Restoring index..");
  svdss_index_t* ix = nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&] { return std::to_string(std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count()); };
  const bool bam_mode = !o.bam.empty();
  BamReader* bam = nullptr;
  std::thread bam_prewarm;
  // BAM records handled on the GPU (csrc/bam_device.hip; the default when there is one): only compressed bytes go up.
  // SVDSS_BAM_DEVICE=0: the host path below (BamReader: chunks inflated on the GPU or by the host pool, records sliced
  // on the host, packed bases uploaded) -- the tested fallback, and what a reader of stdin-like inputs needs.
  const bool dev_bam = bam_mode && svdss_device_count() > 0 && !(getenv("SVDSS_BAM_DEVICE") && atoi(getenv("SVDSS_BAM_DEVICE")) == 0);
  std::vector<BamRegion> bam_regions;
  BgzfScanner::Hooks bam_hooks;
  size_t bam_slab = 0, bam_pool_chunks = 0;
  int bam_loaders = 0;
  int32_t bam_n_ref = 0;
  int64_t bam_skip = 0;
  if (dev_bam) {
    std::string herr;
    if (!bam_header_probe(o.bam, bam_n_ref, bam_skip, herr, nullptr)) die("cannot read " + o.bam + ": " + herr);
    BgzfScanner::Hooks hooks;
    hooks.host_alloc = svdss_host_alloc;
    hooks.host_free = svdss_host_free;
    const size_t slab = (getenv("SVDSS_BAM_SLAB_KB") && atoll(getenv("SVDSS_BAM_SLAB_KB")) >= 64 ? (size_t)atoll(getenv("SVDSS_BAM_SLAB_KB")) << 10 : (size_t)16 << 20);
    const size_t target = (size_t)(getenv("SVDSS_BAM_BATCH_MB") && atoll(getenv("SVDSS_BAM_BATCH_MB")) > 0 ? atoll(getenv("SVDSS_BAM_BATCH_MB")) : 192) << 20;
    const int n_dev0 = std::max(1, svdss_device_count());
    const int n_g = std::max(1, getenv("SVDSS_GPUS_OVERSUBSCRIBE") ? o.gpus : std::min(o.gpus, n_dev0));
    const int per_gpu = getenv("SVDSS_SEARCH_FEEDERS") ? std::max(1, atoi(getenv("SVDSS_SEARCH_FEEDERS"))) : 6;
    // slabs alive at once: those the loaders read ahead + those of the batches being fed, queued and cut
    const size_t per_batch = target / slab + 2;
    const int loaders = getenv("SVDSS_BAM_LOADERS") ? std::max(1, atoi(getenv("SVDSS_BAM_LOADERS"))) : 8;
    // the file's regions, one per GPU (one region for a small file, or SVDSS_REGION_SHARDS=0: every GPU's feeders take its batches)
    const std::vector<size_t> cuts = plan_bam_regions(o.bam, n_g, bam_skip);
    bam_regions.resize(cuts.size() - 1);
    bam_hooks = hooks; bam_slab = slab;
    bam_loaders = bam_regions.size() > 1 ? std::max(2, std::min(loaders, (int)effective_cpus() / (int)bam_regions.size())) : loaders;
    const size_t feeders_per_region = bam_regions.size() > 1 ? (size_t)per_gpu : (size_t)(n_g * per_gpu);
    bam_pool_chunks = (size_t)bam_loaders + (feeders_per_region + 3) * per_batch;
    for (size_t g = 0; g < bam_regions.size(); ++g) {
      BamRegion& R = bam_regions[g];
      R.begin = cuts[g]; R.end = cuts[g + 1];
      R.skip = g == 0 ? bam_skip : 0;
      R.open_start = g > 0;
      R.open_end = g + 1 < bam_regions.size();
      R.sc.reset(new BgzfScanner(o.bam, hooks, slab, bam_loaders, bam_pool_chunks, R.begin, R.end));
      if (!R.sc->ok()) die("cannot open " + o.bam);
    }
    if (!getenv("SVDSS_NO_PREWARM"))
      bam_prewarm = std::thread([&bam_regions] {
        std::vector<std::thread> th;
        for (BamRegion& R : bam_regions) th.emplace_back([&R] { R.sc->prewarm(); });
        for (std::thread& t : th) t.join();
      });
  } else if (bam_mode) {
    // the reader's page-locked chunk buffers are allocated while the index is restored (BamReader::prewarm)
    bam = new BamReader(o.bam, o.io_threads);
    // (BGZF blocks inflated on the GPU, csrc/inflate.hip, on every GPU of --gpus in turn; SVDSS_GPU_INFLATE)
    const int n_dev0 = std::max(1, svdss_device_count());
    svdss_enable_gpu_inflate(*bam, 0, std::min(std::max(1, getenv("SVDSS_GPUS_OVERSUBSCRIBE") ? o.gpus : std::min(o.gpus, n_dev0)), n_dev0));
    if (bam->ok() && !getenv("SVDSS_NO_PREWARM")) bam_prewarm = std::thread([bam] { bam->prewarm(); });
  }
  // One GPU, one region: the BAM front end starts NOW, beside the index restore (EarlySearch; SVDSS_SEARCH_EARLY=0: the index
  // first, as PingPong::run does, ping_pong.cpp:245,329).  The park's first arena is allocated before the restore begins.
  std::unique_ptr<EarlySearch> early;
  std::thread early_stream;
  // (It pays when the restore takes seconds: an index of a chr20-length reference is resident in 0.4 s, and sharing the GPU
  // with the front end meanwhile only delays it -- 1.87 against 1.55 s per 1.03 M reads, profiles/r06t_*.  The sidecar holds
  // a byte per BWT symbol (records + rank blocks): from 800 MB on -- ~0.8 G symbols, a restore of ~0.7 s -- the front end starts first;
  // SVDSS_SEARCH_EARLY=1 forces it, SVDSS_EARLY_MIN_MB moves the threshold.)
  bool early_pays = getenv("SVDSS_SEARCH_EARLY") && atoi(getenv("SVDSS_SEARCH_EARLY")) != 0;
  if (!early_pays) {
    struct stat sti;
    const int64_t min_mb = getenv("SVDSS_EARLY_MIN_MB") ? atoll(getenv("SVDSS_EARLY_MIN_MB")) : 800;   // (records + rank blocks: a byte per BWT symbol)
    if (stat((o.index + ".svdss").c_str(), &sti) == 0 || stat(o.index.c_str(), &sti) == 0) early_pays = (int64_t)sti.st_size >= (min_mb << 20);
  }
  if (dev_bam && early_pays && bam_regions.size() == 1 && std::max(1, getenv("SVDSS_GPUS_OVERSUBSCRIBE") ? o.gpus : std::min(o.gpus, std::max(1, svdss_device_count()))) == 1 &&
      !(getenv("SVDSS_SEARCH_EARLY") && atoi(getenv("SVDSS_SEARCH_EARLY")) == 0)) {
    if (o.bsize <= 0) die("batch size smaller than the number of threads");
    early.reset(new EarlySearch);
    struct stat stb;
    early->file_bytes = stat(o.bam.c_str(), &stb) == 0 ? (int64_t)stb.st_size : 0;
    // (SVDSS_PARK_GB: what may be parked at most, in arenas allocated as they are needed; SVDSS_PARK_MB: the same in MB, for tests)
    const int64_t park_b = getenv("SVDSS_PARK_MB") && atoll(getenv("SVDSS_PARK_MB")) > 0 ? atoll(getenv("SVDSS_PARK_MB")) << 20
                           : (getenv("SVDSS_PARK_GB") && atoll(getenv("SVDSS_PARK_GB")) > 0 ? atoll(getenv("SVDSS_PARK_GB")) : 32) << 30;
    bam_regions[0].gpus = {nullptr};
    // (on a thread of its own from the first moment: this one goes straight to the index file)
    early_stream = std::thread([&, park_b] {
      check(svdss_bam_park_create(0, park_b, park_b / 512 + 4096, &early->park), "svdss_bam_park_create");
      if (bam_prewarm.joinable()) bam_prewarm.join();
      const std::vector<svdss_index_t*> none(1, nullptr);
      search_bam_device(o, none, bam_regions, bam_hooks, bam_slab, bam_loaders, bam_pool_chunks, bam_n_ref, since, early.get());
    });
  }
  // (Tried: the rank blocks of the sidecar read beside the records, on a thread of their own, so that they are in memory when
  // the choice falls.  Two 3 GB reads and the front end's start share the process's cores: the front end's estimate came
  // 0.4 s later and the blocks no sooner -- 5x `search` 2.4 -> 2.7 s.  They are read when they are wanted.)
  check(svdss_index_load(o.index.c_str(), &ix), "svdss_index_load");
  if (early) early->index_n.store(svdss_index_size(ix));
  if (o.verbose) logmsg("debug", "index file read at +" + since() + " s");
  const bool user_kmer = getenv("SVDSS_KMER") != nullptr;      // (the block below may set the variable itself)
  if (!getenv("SVDSS_KMER")) {
    // The order K of the k-mer table trades its build time (4^K entries: 1.6 s at K = 16, a quarter of that per step
    // down) against the search kernel's speed (about a third slower per step down).  The library's own choice (K = 16
    // from 64 Mb on) is the one for a resident index that searches batch after batch; a process that restores the
    // index for ONE input knows roughly how many reads are coming (a BAM is ~1 byte per base, a FASTQ ~2) and takes the
    // K that minimises build + search.  Results never depend on K (tests/test_sfs_gpu.py, tests/test_scale_gpu.py).
    struct stat st;
    const std::string& in = bam_mode ? o.bam : o.fastx;
    // (references above 2^31 symbols keep the library's K: nothing below 16 was measured there)
    if (stat(in.c_str(), &st) == 0 && st.st_size > 0 && svdss_index_size(ix) < ((int64_t)1 << 31)) {
      const double est_reads = (double)st.st_size / (bam_mode ? 15000.0 : 30000.0);
      const int64_t n = svdss_index_size(ix);
      int k_auto = 1;
      while (k_auto < 16 && ((int64_t)1 << (2 * k_auto)) <= n) ++k_auto;
      k_auto = std::min(16, k_auto + 2);
      int best = k_auto;
      double best_cost = 1e300;
      for (int k = k_auto; k >= std::max(8, k_auto - 5); --k) {
        // (the kernel's seconds count double: they are GPU time the BGZF inflate of the stream wants too)
        const double build = 1.6 * std::pow(4.0, k - 16), kernel = est_reads / 15e6 * std::pow(1.35, 16 - k);
        if (build + 2 * kernel < best_cost) { best_cost = build + 2 * kernel; best = k; }
      }
      if (best != k_auto) {
        setenv("SVDSS_KMER", std::to_string(best).c_str(), 0);
        if (o.verbose) logmsg("debug", "k-mer table of order " + std::to_string(best) + " for ~" + std::to_string((long long)est_reads) + " reads");
      }
    }
  }
  // Few reads to search (the front end has seen enough to say: `search` on a smoothed BAM skips what `smooth` tagged XF != 0)
  // and the sidecar carries the rank blocks: the index as a rank structure ALONE -- 3 GB uploaded instead of six billion
  // suffixes sorted for a text, a suffix array and a k-mer table; ~1 M reads/s instead of 8 - 24 M, results identical
  // (svdss_index_attach_blocks).  SVDSS_SEARCH_LF=0|1 forces the choice, SVDSS_SEARCH_LF_MAX moves the threshold (reads).
  bool lf_only = false;
  if (early && !user_kmer && !(getenv("SVDSS_SEARCH_LF") && atoi(getenv("SVDSS_SEARCH_LF")) == 0)) {
    const bool forced = getenv("SVDSS_SEARCH_LF") && atoi(getenv("SVDSS_SEARCH_LF")) != 0;
    const auto w0 = std::chrono::steady_clock::now();
    for (;;) {
      bool done;
      { std::lock_guard<std::mutex> lk(early->m); done = early->front_done; }
      if (forced || done || early->records.load() >= 50000 || std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 1.5) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    const int64_t recs = early->records.load(), srch = early->searched.load(), cb = early->comp_bytes.load();
    const std::string t_est = since();
    double est = -1;
    if (recs > 0 && cb > 0) est = (double)srch / (double)recs * ((double)recs * (double)early->file_bytes / (double)cb);
    // (what the rank structure alone saves is the rest of the restore -- ~4.5 s at GRCh38 lengths, in proportion for a
    // smaller reference --, what it costs is the search at ~1 M reads/s instead of 8 - 24 M: worth it below ~2 M reads
    // per 6.2e9 BWT symbols; profiles/r06q_*)
    const double lf_max = getenv("SVDSS_SEARCH_LF_MAX") ? atof(getenv("SVDSS_SEARCH_LF_MAX")) : 2e6 * (double)svdss_index_size(ix) / 6.18e9;
    if (forced || (est >= 0 && est <= lf_max)) {
      const int rc = svdss_index_attach_blocks(ix, o.index.c_str());
      if (rc == SVDSS_OK) {
        lf_only = true;
        if (o.verbose) logmsg("debug", "~" + std::to_string((long long)std::max(0.0, est)) + " reads to search (known at +" + t_est + " s): the index as a rank structure alone (blocks read at +" + since() + " s)");
      } else if (rc != SVDSS_EINVAL) check(rc, "svdss_index_attach_blocks");
    }
  }
  check(svdss_index_to_device(ix, 0), "svdss_index_to_device");
  if (o.verbose) logmsg("debug", "index and k-mer table on the device at +" + since() + " s" +
                                     (lf_only ? " (rank blocks alone: few reads to search)"
                                      : early && svdss_index_kmer(ix) < 16 ? " (table of order " + std::to_string(svdss_index_kmer(ix)) + ": few reads to search)" : ""));
  if (early && lf_only) {
    // (resident long before the file has been read: held back until the front end is through -- or the park is full -- so
    // that what is parked goes in large launches, one lane per read, instead of a small segmented launch per batch)
    std::unique_lock<std::mutex> lk(early->m);
    early->ix = ix; early->ix_avail = true;      // (the drain thread has it at once and searches the groups as they close)
    early->cv.notify_all();
    early->cv.wait(lk, [&] { return early->front_done || early->park_full; });
  }
  if (early) {
    logmsg("info", "Extracting SFS strings on the GPU (output order as with " + std::to_string(o.threads) + " threads)..");
    // (SVDSS_EARLY_HOLD_MS, for the tests: the index is held back that long, as if its restore had taken seconds)
    if (const char* e = getenv("SVDSS_EARLY_HOLD_MS")) if (atoi(e) > 0) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(e)));
    { std::lock_guard<std::mutex> lk(early->m); early->ix = ix; early->ready = true; }
    early->cv.notify_all();
    early_stream.join();
    if (!getenv("SVDSS_CLEAN_EXIT")) {
      logmsg("info", "All done! Runtime: " + std::to_string((long)(time(nullptr) - g_t0)) + " seconds");
      fflush(stdout);
      fflush(stderr);
      _exit(0);
    }
    svdss_bam_park_free(early->park);
    bam_regions.clear();
    svdss_index_free(ix);
    logmsg("info", "All done! Runtime: " + std::to_string((long)(time(nullptr) - g_t0)) + " seconds");
    return 0;
  }
  // --gpus N: one replica of the index per GPU; the batches of reads go to whichever GPU is free, the
  // text is written in input order whatever GPU searched a batch -- the same bytes as with one GPU
  // (SVDSS_GPUS_OVERSUBSCRIBE: more replicas than GPUs, replica d on GPU d % count -- exercises the path on a one-GPU box)
  const int n_dev = std::max(1, svdss_device_count());
  const int n_gpus = std::max(1, getenv("SVDSS_GPUS_OVERSUBSCRIBE") ? o.gpus : std::min(o.gpus, n_dev));
  std::vector<svdss_index_t*> replicas((size_t)n_gpus, ix);
  {
    // every replica is built in the HBM of its own GPU from the records (or copied there), all of them at once
    std::vector<std::thread> th;
    std::vector<int> rcs((size_t)n_gpus, SVDSS_OK);
    for (int d = 1; d < n_gpus; ++d)
      th.emplace_back([&, d] { rcs[(size_t)d] = svdss_index_replicate(ix, d % n_dev, &replicas[(size_t)d]); });
    for (std::thread& t : th) t.join();
    for (int d = 1; d < n_gpus; ++d) check(rcs[(size_t)d], "svdss_index_replicate");
  }
  if (n_gpus > 1) logmsg("info", "Index replicated on " + std::to_string(n_gpus) + " GPUs");
  FastxReader* fx = nullptr;
  if (dev_bam) {
    if (bam_prewarm.joinable()) bam_prewarm.join();
    if (o.bsize <= 0) die("batch size smaller than the number of threads");
    logmsg("info", "Extracting SFS strings on the GPU (output order as with " + std::to_string(o.threads) + " threads)..");
    // region g on GPU g (one region: every GPU's feeders take its batches)
    for (size_t g = 0; g < bam_regions.size(); ++g) {
      if (bam_regions.size() == 1) bam_regions[g].gpus = replicas;
      else bam_regions[g].gpus = {replicas[g % replicas.size()]};
    }
    if (o.verbose && bam_regions.size() > 1) {
      std::string m = "file regions (bytes):";
      for (const BamRegion& R : bam_regions) m += " " + std::to_string(R.end - R.begin);
      logmsg("debug", m);
    }
    search_bam_device(o, replicas, bam_regions, bam_hooks, bam_slab, bam_loaders, bam_pool_chunks, bam_n_ref, since);
    if (!getenv("SVDSS_CLEAN_EXIT")) {
      logmsg("info", "All done! Runtime: " + std::to_string((long)(time(nullptr) - g_t0)) + " seconds");
      fflush(stdout);
      fflush(stderr);
      _exit(0);
    }
    bam_regions.clear();
    for (svdss_index_t* r : replicas) svdss_index_free(r);
    return 0;
  }
  if (bam_mode) {
    if (bam_prewarm.joinable()) bam_prewarm.join();
    if (!bam->ok() || !bam->read_header()) die("cannot read " + o.bam + ": " + bam->error());
  } else {
    logmsg("warning", "FASTX mode is not optimized (higher running times and larger SFSs set).");
    fx = new FastxReader(o.fastx);
    if (!fx->ok()) die("cannot open " + o.fastx);
  }
  if (o.bsize <= 0) die("batch size smaller than the number of threads");
  logmsg("info", "Extracting SFS strings on the GPU (output order as with " + std::to_string(o.threads) + " threads)..");
  // One GPU launch covers many reference-sized batches; the text is still emitted batch by
  // batch, thread slice by thread slice, read names in std::map order (ping_pong.cpp:215-217).
  // (32 k reads keep the GPU efficient and let parsing, search and output of successive batches overlap)
  const int64_t super = std::max<int64_t>(o.bsize, 32768 / o.bsize * (int64_t)o.bsize);
  BoundedQueue<SearchBatch> parsed(4);   // (the GPU threads are created below, after the replicas)
  // batch objects go round: their vectors and text buffers keep their capacity (tens of MB each; a fresh allocation of
  // that size is an mmap, a page fault per 4 KB and a munmap that stalls every other thread of the process)
  std::mutex pool_m;
  std::vector<std::unique_ptr<SearchBatch>> batch_pool;
  auto new_batch = [&]() {
    std::unique_ptr<SearchBatch> b;
    {
      std::lock_guard<std::mutex> lk(pool_m);
      if (!batch_pool.empty()) { b = std::move(batch_pool.back()); batch_pool.pop_back(); }
    }
    if (!b) b.reset(new SearchBatch);
    b->reads.clear(); b->gbuf.clear(); b->boff.clear(); b->lseq.clear(); b->recs.clear(); b->keep.clear();
    b->goff.clear(); b->gidx.clear(); b->qs.clear(); b->ln.clear(); b->text.clear(); b->counts.clear();
    b->n_lines = 0; b->seq = 0;
    return b;
  };
  auto recycle_batch = [&](std::unique_ptr<SearchBatch> b) {
    std::lock_guard<std::mutex> lk(pool_m);
    if (batch_pool.size() < 32) batch_pool.push_back(std::move(b));
  };
  PinnedPool pinned;
  // searched batches wait here for their turn: two GPU threads finish them out of order
  std::mutex done_m;
  std::condition_variable done_cv;
  std::map<uint64_t, std::unique_ptr<SearchBatch>> done;
  bool gpu_finished = false;
  uint64_t next_seq = 0;
  uint64_t n_seen = 0, total_sfs = 0;
  double t_slice = 0, t_decode = 0, t_gpu = 0, t_write = 0, t_format = 0;   // busy seconds of the stages (--verbose)
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };

  // (threads of the small per-batch loops -- tag decoding, the copy of the packed bases; --io-threads sizes the inflate pool)
  const int n_workers = (int)std::max(1u, std::min(16u, effective_cpus()));
  // items [0, n) over the worker threads, contiguous slices
  auto parallel_for = [&](size_t n, const std::function<void(size_t, size_t)>& body) {
    const size_t nt = std::min<size_t>((size_t)n_workers, std::max<size_t>(1, n / 64));
    if (nt <= 1) { body(0, n); return; }
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(body, n * t / nt, n * (t + 1) / nt);
    body(0, n / nt);
    for (std::thread& th : pool) th.join();
  };

  std::thread producer([&] {
    bool eof = false;
    std::vector<BamReader::RawView> recs;
    std::vector<std::shared_ptr<BamReader::Bytes>> keep_chunks;
    while (!eof) {
      std::unique_ptr<SearchBatch> bt = new_batch();
      bt->goff.assign(1, 0);
      if (bam_mode) {
        // locate the records of one batch in the inflated chunks (sequential, no copies: the chunks are kept
        // alive until the batch is decoded), then decode them in parallel
        recs.clear();
        keep_chunks.clear();
        uint64_t seen_chunk = ~0ull;
        const auto ts0 = now();
        while ((int64_t)recs.size() < super) {
          BamReader::RawView rr;
          const int rc = bam->next_view(rr);
          if (rc == 0) { eof = true; break; }
          if (rc < 0) die("error reading " + o.bam + ": " + bam->error());
          if (bam->chunk_id() != seen_chunk) { seen_chunk = bam->chunk_id(); keep_chunks.push_back(bam->chunk()); }
          ++n_seen;
          bool keep = !(rr.flag & (4 | 2048 | 256));                     // ping_pong.cpp:66-69
          if (keep && rr.l_seq < 100) {                                  // :70-75
            logmsg("warning", "Alignment filtered due to l_qseq. Why are we here? Please check");
            keep = false;
          }
          if (keep && rr.tid < 0) die("core.tid < 0. Why are we here? Please check");  // :76-79
          if (!keep) continue;
          recs.push_back(std::move(rr));
        }
        const auto ts1 = now();
        t_slice += secs(ts0, ts1);
        const size_t n = recs.size();
        bt->reads.resize(n);
        parallel_for(n, [&](size_t lo, size_t hi) {
          for (size_t i = lo; i < hi; ++i) {
            const BamReader::RawView& rr = recs[i];
            Read& r = bt->reads[i];
            r.name.assign((const char*)rr.name(), rr.l_name ? rr.l_name - 1 : 0);
            int64_t xf = 0, hp = 0;
            BamReader::aux_int(rr.aux(), rr.l_aux, "XF", xf);   // :196-201, missing => 0
            BamReader::aux_int(rr.aux(), rr.l_aux, "HP", hp);
            r.hp = (int)hp;
            if (o.putative && xf != 0) { r.count = -1; r.len = 0; }                 // :202-203
            else r.len = rr.l_seq;
          }
        });
        for (size_t i = 0; i < n; ++i) {
          if (bt->reads[i].count < 0) continue;
          bt->gidx.push_back(i);
          bt->goff.push_back(bt->goff.back() + bt->reads[i].len);
        }
        bt->boff.assign(1, 0);
        bt->lseq.resize(bt->gidx.size());
        for (size_t k = 0; k < bt->gidx.size(); ++k) {
          const int32_t l = recs[bt->gidx[k]].l_seq;
          bt->lseq[k] = l;
          bt->boff.push_back(bt->boff.back() + ((int64_t)l + 1) / 2);
        }
        // the packed bases of the batch, back to back in page-locked memory; the inflated chunks go back to the reader
        // at once (they are page-locked too when the GPU inflates: few should be in flight)
        if (!bt->gidx.empty()) {
          bt->seq4 = pinned.get((size_t)bt->boff.back() + 16, bt->seq4_cap);
          parallel_for(bt->gidx.size(), [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k)
              memcpy(bt->seq4 + bt->boff[k], recs[bt->gidx[k]].seq4(), (size_t)(bt->boff[k + 1] - bt->boff[k]));
          });
        }
        recs.clear();
        keep_chunks.clear();
        t_decode += secs(ts1, now());
      } else {
        while ((int64_t)bt->reads.size() < super) {
          Read r;
          std::string seq;
          if (!fx->next(r.name, seq)) { eof = true; break; }
          ++n_seen;
          r.len = (int64_t)seq.size();
          const size_t at = bt->gbuf.size();
          bt->gbuf.resize(at + seq.size());
          svdss_nt6_encode(seq.data(), (int64_t)seq.size(), bt->gbuf.data() + at);
          bt->goff.push_back((int64_t)bt->gbuf.size());
          bt->gidx.push_back(bt->reads.size());
          bt->reads.push_back(std::move(r));
        }
      }
      if (!bt->reads.empty()) { bt->seq = next_seq++; parsed.push(std::move(bt)); }
    }
    parsed.close();
  });

  std::thread writer([&] {
    uint64_t want = 0;
    for (;;) {
      std::unique_ptr<SearchBatch> bt;
      {
        std::unique_lock<std::mutex> lk(done_m);
        done_cv.wait(lk, [&] { return done.count(want) || (gpu_finished && done.empty()); });
        auto it = done.find(want);
        if (it == done.end()) break;
        bt = std::move(it->second);
        done.erase(it);
        ++want;
      }
      done_cv.notify_all();
      const auto tw0 = now();
      fwrite(bt->text.data(), 1, bt->text.size(), stdout);
      total_sfs += bt->n_lines;
      t_write += secs(tw0, now());
      recycle_batch(std::move(bt));
    }
    fflush(stdout);
  });

  // two threads per GPU feed it, each with its own batch object (own stream): the upload of one batch overlaps the
  // search of the other; the thread that searched a batch also formats its text, the writer only writes
  std::mutex t_m;
  auto gpu_worker = [&](svdss_index_t* ix) {
    svdss_sfs_batch_t* res = nullptr;
    while (std::unique_ptr<SearchBatch> bt = parsed.pop()) {
      const auto tg0 = now();
      if (!bt->gidx.empty()) {
        std::vector<int64_t>& counts = bt->counts;
        counts.assign(bt->gidx.size(), 0);
        if (bam_mode)
          check(svdss_sfs_search_batch_bam(ix, bt->seq4, bt->boff.data(), bt->lseq.data(), (int64_t)bt->gidx.size(),
                                           o.assemble ? SVDSS_SFS_ASSEMBLE : 0, &res), "svdss_sfs_search_batch_bam");
        else
          check(svdss_sfs_search_batch(ix, bt->gbuf.data(), bt->goff.data(), (int64_t)bt->gidx.size(),
                                       o.assemble ? SVDSS_SFS_ASSEMBLE : 0, &res), "svdss_sfs_search_batch");
        bt->qs.resize((size_t)svdss_sfs_batch_total(res));
        bt->ln.resize(bt->qs.size());
        check(svdss_sfs_batch_fetch(res, counts.data(), bt->qs.data(), bt->ln.data(), nullptr), "svdss_sfs_batch_fetch");
        int64_t acc = 0;
        for (size_t k = 0; k < bt->gidx.size(); ++k) {
          bt->reads[bt->gidx[k]].first = acc;
          bt->reads[bt->gidx[k]].count = counts[k];
          acc += counts[k];
        }
      }
      bt->gbuf.clear();
      pinned.put(bt->seq4, bt->seq4_cap);
      bt->seq4 = nullptr;
      const auto tg1 = now();
      format_batch(o, *bt);
      { std::lock_guard<std::mutex> lk(t_m); t_gpu += secs(tg0, tg1); t_format += secs(tg1, now()); }
      {
        // (bounded: a finished batch waits until the writer is at most 3 batches behind)
        std::unique_lock<std::mutex> lk(done_m);
        const uint64_t sq = bt->seq;
        done_cv.wait(lk, [&] { return done.size() < 8 || done.begin()->first > sq; });
        done[sq] = std::move(bt);
      }
      done_cv.notify_all();
    }
    svdss_sfs_batch_free(res);
  };
  {
    std::vector<std::thread> gpu_threads;
    // (several feeding threads per GPU, each with its own batch object and stream: upload, search, download and the
    // text formatting of different batches overlap; formatting alone needs four to five threads at a million reads/s)
    const int per_gpu = getenv("SVDSS_SEARCH_FEEDERS") ? std::max(1, atoi(getenv("SVDSS_SEARCH_FEEDERS"))) : 6;
    for (int d = 0; d < n_gpus; ++d)
      for (int k = 0; k < per_gpu; ++k)
        if (d || k) gpu_threads.emplace_back(gpu_worker, replicas[(size_t)d]);
    gpu_worker(replicas[0]);
    for (std::thread& th : gpu_threads) th.join();
  }
  { std::lock_guard<std::mutex> lk(done_m); gpu_finished = true; }
  done_cv.notify_all();
  producer.join();
  writer.join();
  if (o.verbose) {
    logmsg("debug", std::to_string(n_seen) + " records read, " + std::to_string(total_sfs) + " SFS written at +" + since() + " s");
    logmsg("debug", "stage busy seconds: inflate+slice " + std::to_string(t_slice) + ", decode " + std::to_string(t_decode) +
                        ", GPU search + copies " + std::to_string(t_gpu) + ", format " + std::to_string(t_format) + ", write " + std::to_string(t_write));
  }
  // Everything is written.  Giving back gigabytes of page-locked buffers and the index on the device one by one takes
  // half a second that the operating system spends anyway when the process ends: end it here (SVDSS_CLEAN_EXIT=1 keeps
  // the orderly teardown, for leak checkers).
  if (!getenv("SVDSS_CLEAN_EXIT")) {
    if (bam) bam->report();
    logmsg("info", "All done! Runtime: " + std::to_string((long)(time(nullptr) - g_t0)) + " seconds");
    fflush(stdout);
    fflush(stderr);
    _exit(0);
  }
  for (svdss_index_t* r : replicas) svdss_index_free(r);
  delete bam;
  delete fx;
  return 0;
}

int main(int argc, char** argv) {
  const time_t t0 = time(nullptr);
  g_t0 = t0;
  // large blocks stay in the allocator instead of going back to the kernel with every free (with a hundred threads an
  // munmap is a stall for all of them)
  // (the chunk loaders, the feeding threads and the call-side batches each work on a stream of their own; the runtime
  // multiplexes streams onto 4 hardware queues by default, and streams that share one run in order)
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  mallopt(M_MMAP_THRESHOLD, 32 << 20);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 64 << 20);
  if (argc == 1) {
    fputs(MAIN_USAGE, stderr);
    return EXIT_FAILURE;
  }
  if (!strcmp(argv[1], "index")) {
    logmsg("info", "FM-index construction (stands for 'ropebwt3 build')");
    const int rc = main_index(argc, argv);
    if (rc) return rc;
  } else {
    for (int i = 1; i < argc; ++i)
      if (!strcmp(argv[i], "--version")) { printf("SVDSS, %s\n", VERSION); return EXIT_SUCCESS; }
    const Options o = parse(argc, argv);
    if (o.help) {   // Configuration::print_help(argv[1]), config.cpp:12-24: the mode's own usage text
      fputs(!strcmp(argv[1], "search") ? SEARCH_USAGE : !strcmp(argv[1], "call") ? CALL_USAGE :
            !strcmp(argv[1], "smooth") ? SMOOTH_USAGE : MAIN_USAGE, stderr);
      return EXIT_SUCCESS;
    }
    if (!strcmp(argv[1], "search")) {
      if (o.index.empty() || (o.fastx.empty() && o.bam.empty())) { fputs(SEARCH_USAGE, stderr); return EXIT_FAILURE; }
      main_search(o);
    } else if (!strcmp(argv[1], "call")) {
      if (o.reference.empty() || o.bam.empty() || o.sfs.empty()) { fputs(CALL_USAGE, stderr); return EXIT_FAILURE; }  // main.cpp:56-59
      CallOptions c;
      c.reference = o.reference; c.bam = o.bam; c.sfs = o.sfs; c.threads = o.threads; c.gpus = o.gpus;
      c.min_cluster_weight = o.min_cluster_weight; c.min_sv_length = o.min_sv_length; c.min_mapq = o.min_mapq;
      c.useht = o.useht; c.min_ratio = o.min_ratio; c.poa = o.poa; c.clusters = o.clusters; c.verbose = o.verbose;
      c.clipped = o.clipped;
      main_call(c);
    } else if (!strcmp(argv[1], "smooth")) {
      if (o.reference.empty() || o.bam.empty()) { fputs(SMOOTH_USAGE, stderr); return EXIT_FAILURE; }   // main.cpp:73-76
      CallOptions c;
      c.reference = o.reference; c.bam = o.bam; c.threads = o.threads; c.min_mapq = o.min_mapq; c.accp = o.accp; c.gpus = o.gpus;
      main_smooth(c);
    } else {
      fputs(MAIN_USAGE, stderr);
      return EXIT_FAILURE;
    }
  }
  logmsg("info", "All done! Runtime: " + std::to_string((long)(time(nullptr) - t0)) + " seconds");
  return 0;
}
