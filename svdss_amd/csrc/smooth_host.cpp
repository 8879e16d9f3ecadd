// smooth_host.cpp -- `SVDSS smooth` (/root/reference/smoother.cpp): every primary, mapq-ok
// alignment is rewritten to equal the reference except at long (> 20 bp) indels and soft clips
// and tagged XF (0 smoothed & interesting, 1 too many mismatches, 2 nothing interesting); all
// other records are dropped; output order == input order; BAM on stdout.  A CIGAR walk with
// copies from the reference -- the reference has no alignment DP here.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <chrono>
#include <string>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/svdss_hip.h"
#include "bam_reader.h"
#include "bam_device_select.h"
#include "gpu_deflate_hook.h"
#include "gpu_inflate_hook.h"
#include "bam_writer.h"
#include "call_host.h"
#include "fastx_reader.h"

namespace {
[[noreturn]] void die(const std::string& m) { fprintf(stderr, "[smooth] [critical] %s\n", m.c_str()); exit(EXIT_FAILURE); }

const int MIN_INDEL = 20;   // config.hpp:95

bool is_m(uint32_t op) { return op == 0 || op == 7 || op == 8; }

// the walks of this file index ref[pos ..] and the read by the CIGAR: true if the alignment stays inside its contig and
// its CIGAR adds up to the read (what is not is passed through with XF = 3 and left out of the accuracy percentile)
bool cigar_fits(const BamRecord& r, size_t seq_len, size_t ref_len) {
  size_t rl = 0, ql = 0;
  for (uint32_t c : r.cigar) {
    const uint32_t l = c >> 4, op = c & 0xf;
    if (is_m(op)) { rl += l; ql += l; }
    else if (op == 1 || op == 4) ql += l;
    else if (op == 2) rl += l;
    else break;
  }
  return r.pos >= 0 && (size_t)r.pos + rl <= ref_len && ql == seq_len;
}

void mismatch_counts(const BamRecord& r, const std::string& seq, const std::string& ref, double& nm, double& nx) {
  size_t ref_off = (size_t)r.pos, q_off = 0;
  nm = nx = 0;
  for (uint32_t c : r.cigar) {
    const uint32_t l = c >> 4, op = c & 0xf;
    if (is_m(op)) {
      for (uint32_t j = 0; j < l; ++j) (ref[ref_off + j] == seq[q_off + j]) ? ++nm : ++nx;
      ref_off += l; q_off += l;
    } else if (op == 1 || op == 4) q_off += l;
    else if (op == 2) ref_off += l;
    else break;
  }
}

// bam_aux_update_int(aln, "XF", v): overwrite an existing integer XF, else append XF:C
void set_xf(std::vector<uint8_t>& aux, int v) {
  size_t p = 0;
  while (p + 3 <= aux.size()) {
    const char t0 = (char)aux[p], t1 = (char)aux[p + 1], ty = (char)aux[p + 2];
    size_t sz = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': sz = 1; break;
      case 's': case 'S': sz = 2; break;
      case 'i': case 'I': case 'f': sz = 4; break;
      case 'Z': case 'H': { size_t z = p + 3; while (z < aux.size() && aux[z]) ++z; sz = z - (p + 3) + 1; break; }
      case 'B': {
        if (p + 8 > aux.size()) return;
        const char st = (char)aux[p + 3];
        int32_t cnt; memcpy(&cnt, &aux[p + 4], 4);
        sz = 5 + (size_t)cnt * ((st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4);
        break;
      }
      default: return;
    }
    if (t0 == 'X' && t1 == 'F' && strchr("cCsSiI", ty)) {
      memset(&aux[p + 3], 0, sz);
      aux[p + 3] = (uint8_t)v;
      return;
    }
    p += 3 + sz;
  }
  aux.push_back('X'); aux.push_back('F'); aux.push_back('C'); aux.push_back((uint8_t)v);
}

// grow-only buffer for what goes to / comes from the GPU in every batch: page-locked when the runtime gives it (the
// copies then run at PCIe speed), never zero-filled
struct PinBuf {
  uint8_t* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  ~PinBuf() { release(); }
  void release() {
    if (!p) return;
    if (pinned) svdss_host_free(p); else free(p);
    p = nullptr; cap = 0;
  }
  uint8_t* ensure(size_t n) {
    if (n <= cap) return p;
    release();
    const size_t want = n + n / 4 + 4096;
    void* q = nullptr;
    if (svdss_host_alloc((int64_t)want, &q) == SVDSS_OK && q) { p = (uint8_t*)q; pinned = true; }
    else { p = (uint8_t*)malloc(want); pinned = false; if (!p) { fprintf(stderr, "[smooth] [critical] out of memory\n"); exit(EXIT_FAILURE); } }
    cap = want;
    return p;
  }
};

struct ByteSink {
  std::vector<uint8_t> v;
  void write(const void* p, size_t n) { v.insert(v.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
};

// one record with the bases already in BAM's 4-bit form
void write_record_packed(ByteSink& w, const BamRecord& r, const uint32_t* cigar, size_t n_cigar, const uint8_t* packed,
                         int32_t l_seq, const uint8_t* qual, const std::vector<uint8_t>& aux) {
  // BAM keeps l_read_name in 8 bits and n_cigar_op in 16 (longer CIGARs live in a CG tag, which this writer does not
  // produce): refuse instead of writing a record that no longer parses
  if (r.qname.size() + 1 > 255) die("read name longer than 254 characters: " + r.qname);
  if (n_cigar > 65535) die("more than 65535 CIGAR operations (CG tag records are not supported): " + r.qname);
  const uint8_t l_name = (uint8_t)(r.qname.size() + 1);
  const uint16_t n_cig = (uint16_t)n_cigar;
  const size_t pbytes = ((size_t)l_seq + 1) / 2;
  const int32_t block = 32 + l_name + 4 * n_cig + (int32_t)pbytes + l_seq + (int32_t)aux.size();
  uint8_t core[36];
  memcpy(core, &block, 4);
  memcpy(core + 4, &r.tid, 4);
  memcpy(core + 8, &r.pos, 4);
  core[12] = l_name; core[13] = r.mapq;
  memcpy(core + 14, &r.bin, 2);
  memcpy(core + 16, &n_cig, 2);
  memcpy(core + 18, &r.flag, 2);
  memcpy(core + 20, &l_seq, 4);
  memcpy(core + 24, &r.mtid, 4);
  memcpy(core + 28, &r.mpos, 4);
  memcpy(core + 32, &r.isize, 4);
  w.v.reserve(w.v.size() + 4 + (size_t)block);   // (one allocation per record instead of a doubling series)
  w.write(core, 36);
  w.write(r.qname.c_str(), l_name);
  if (n_cig) w.write(cigar, 4u * n_cig);
  w.write(packed, pbytes);
  w.write(qual, (size_t)l_seq);
  w.write(aux.data(), aux.size());
}

void write_record(ByteSink& w, const BamRecord& r, const std::vector<uint32_t>& cigar, const std::string& seq,
                  const std::vector<uint8_t>& qual, const std::vector<uint8_t>& aux) {
  const int32_t l_seq = (int32_t)seq.size();
  std::vector<uint8_t> packed(((size_t)l_seq + 1) / 2, 0);
  static const struct Lut {   // (initialised once, before any worker thread runs: function-local static)
    uint8_t t[256];
    Lut() {
      memset(t, 15, sizeof t);
      const char* nt16 = "=ACMGRSVTWYHKDBN";
      for (int i = 0; i < 16; ++i) { t[(uint8_t)nt16[i]] = (uint8_t)i; t[(uint8_t)tolower(nt16[i])] = (uint8_t)i; }
    }
  } lut_obj;
  const uint8_t* lut = lut_obj.t;
  for (int32_t i = 0; i < l_seq; ++i) packed[(size_t)i >> 1] |= (uint8_t)(lut[(uint8_t)seq[(size_t)i]] << ((~i & 1) << 2));
  write_record_packed(w, r, cigar.data(), cigar.size(), packed.data(), l_seq, qual.data(), aux);
}
}  // namespace

int main_smooth(const CallOptions& o) {
  std::unordered_map<std::string, std::string> chrom;
  const auto t_fasta0 = std::chrono::steady_clock::now();
  // the HIP runtime and the device's context come up (a few tenths of a second) while the FASTA is read
  std::thread gpu_warm([] {
    void* q = nullptr;
    if (svdss_device_count() > 0 && svdss_host_alloc(1 << 20, &q) == SVDSS_OK && q) svdss_host_free(q);
  });
  // ... and so is the BAM's header (0.3 s through the host reader: its workers, its first chunks), for the device path below
  struct HeaderPre { std::string text, err; std::vector<std::string> names; std::vector<int32_t> lens; int32_t n_ref = 0; int64_t skip = 0; } hp;
  std::thread header_pre([&hp, &o] {
    BamReader hb(o.bam);
    if (!hb.ok() || !hb.read_header()) { hp.err = "cannot read " + o.bam + ": " + hb.error(); return; }
    hp.text = hb.header_text(); hp.names = hb.ref_names(); hp.lens = hb.ref_lens();
    std::string perr;
    if (!bam_header_probe(o.bam, hp.n_ref, hp.skip, perr, nullptr)) { hp.err = "cannot read " + o.bam + ": " + perr; return; }
    if ((size_t)hp.n_ref != hp.names.size()) hp.err = "cannot read " + o.bam + ": inconsistent header";
  });
  {
    // load_chromosomes (chromosomes.cpp:9-27).  A plain FASTA with '\n' line ends is mapped and read by several threads
    // (fastx_reader.h, as `SVDSS call` does: GRCh38 in ~0.3 s instead of ~2 s, which was half of a smooth run of a million
    // reads); anything else -- gzip, CRLF, FASTQ-like headers -- line by line.
    std::vector<std::string> nm, sq;
    if (!getenv("SVDSS_FASTA_SERIAL") && load_fasta_mapped(o.reference, std::max(1, std::min((int)o.threads, 16)), true, nm, sq)) {
      for (size_t i = 0; i < nm.size(); ++i) chrom[nm[i]] = std::move(sq[i]);   // (a name that occurs twice: the later record wins, as below)
    } else {
      FastxReader fx(o.reference);
      if (!fx.ok()) die("cannot open " + o.reference);
      std::string name, seq;
      while (fx.next(name, seq)) {
        for (char& c : seq) c = (char)(c - ((c >= 'a' && c <= 'z') ? 32 : 0));   // toupper (ASCII; vectorises)
        chrom[name].swap(seq);
      }
    }
  }
  gpu_warm.join();
  header_pre.join();
  const double fasta_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fasta0).count();
  auto eligible = [&](const BamRecord& r, const std::vector<std::string>& names) {
    if (r.flag & (4 | 2048 | 256)) return false;
    if ((unsigned)r.mapq < o.min_mapq || r.l_seq < 2) return false;
    if (r.tid < 0) die("core.tid < 0. Why are we here? Please check");
    return r.tid < (int)names.size() && chrom.count(names[(size_t)r.tid]) > 0;
  };
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  const bool dbg = getenv("SVDSS_DEBUG") != nullptr;
  double t_read = 0, t_proc = 0, t_write = 0;
  // ---- the device path (csrc/bam_smooth.inc): compressed blocks up, compressed blocks down; the records are filtered,
  // measured, smoothed, rebuilt and deflated in HBM.  SVDSS_BAM_DEVICE=0 (or SVDSS_SMOOTH_HOST=1): the host pipeline below,
  // which writes the same bytes.
  // (SVDSS_GPU_DEFLATE=0 asks for the host's deflate: that is the host pipeline's writer)
  if (!getenv("SVDSS_SMOOTH_HOST") && svdss_device_count() > 0 && !(getenv("SVDSS_BAM_DEVICE") && atoi(getenv("SVDSS_BAM_DEVICE")) == 0) &&
      !(getenv("SVDSS_GPU_DEFLATE") && atoi(getenv("SVDSS_GPU_DEFLATE")) == 0)) {
    if (!hp.err.empty()) die(hp.err);
    const std::string& header_text = hp.text;
    const std::vector<std::string>& names = hp.names;
    const std::vector<int32_t>& lens = hp.lens;
    const int32_t n_ref = hp.n_ref;
    const int64_t skip = hp.skip;
    const int64_t target = (getenv("SVDSS_BAM_BATCH_MB") && atoll(getenv("SVDSS_BAM_BATCH_MB")) > 0 ? atoll(getenv("SVDSS_BAM_BATCH_MB")) : 64) << 20;
    const int per_gpu = getenv("SVDSS_SEARCH_FEEDERS") ? std::max(1, atoi(getenv("SVDSS_SEARCH_FEEDERS"))) : 6;
    const std::vector<svdss_bam_filter_t*> one(1, nullptr);
    const std::vector<int> dev0(1, 0);
    // --gpus N (round 6): the file's regions, one per GPU (ShardedBamSelect, bam_device_select.h: the machinery of `SVDSS call
    // --gpus N`) -- every region has its own loaders, feeding threads and record stream, and every GPU its copy of the
    // chromosomes; SVDSS_GPUS_OVERSUBSCRIBE puts the N regions on the GPUs there are.  The records and their order are those
    // of one GPU's run; the BGZF members are not cut at the same bytes: a region ends with a short member where one GPU's
    // stream would have gone on filling it, and the record a seam completes is a member of its own (`gzip -dc` of the two
    // files is the same).
    const int n_phys = std::max(1, svdss_device_count());
    const int n_g = std::max(1, getenv("SVDSS_GPUS_OVERSUBSCRIBE") ? o.gpus : std::min(o.gpus, n_phys));
    const std::vector<size_t> cuts = n_g > 1 ? plan_bam_regions(o.bam, n_g, skip) : std::vector<size_t>{0, 0};
    const size_t n_regions = cuts.size() - 1;
    const size_t n_sm = std::min<size_t>(n_regions, (size_t)n_phys);
    std::vector<svdss_bam_smooth_t*> sms(n_sm, nullptr);
    std::vector<svdss_ref_t*> drefs(n_sm, nullptr);
    double al_accuracy = 0.0;
    // The reader of the smoothing pass starts FIRST: its loaders page-lock their slabs and read ahead while the reference
    // goes up and the accuracy pass runs; its feeding threads wait at this gate for the threshold.
    struct Gate { std::mutex m; std::condition_variable cv; bool open = false; } gate;
    // the output's header goes in front of the first batch's records
    svdss_bam_stream_t* stream = nullptr;
    if (svdss_bam_stream_create(n_ref, &stream) != SVDSS_OK) die("out of memory");
    {
      struct Sink { std::vector<uint8_t> v; void write(const void* p, size_t n) { v.insert(v.end(), (const uint8_t*)p, (const uint8_t*)p + n); } } hs;
      hs.write("BAM\1", 4);
      const int32_t lt = (int32_t)header_text.size(), nr = (int32_t)names.size();
      hs.write(&lt, 4);
      hs.write(header_text.data(), header_text.size());
      hs.write(&nr, 4);
      for (size_t i = 0; i < names.size(); ++i) {
        const int32_t ln = (int32_t)names[i].size() + 1;
        hs.write(&ln, 4);
        hs.write(names[i].c_str(), (size_t)ln);
        hs.write(&lens[i], 4);
      }
      if (svdss_bam_stream_set_output_prefix(stream, hs.v.data(), (int64_t)hs.v.size()) != SVDSS_OK) die("out of memory");
    }
    // The BGZF members of a batch come down into a page-locked buffer of this pool and are written from it: no copy on
    // the way.  When stdout is a regular file the batches are written side by side (pwrite at the offsets the ordered
    // hand-over gives them: one thread copying into the page cache is slower than the GPU side); a pipe gets them in order.
    struct Pool {
      std::mutex m; std::condition_variable cv;
      std::vector<uint8_t*> buf; std::vector<int> free_;
      size_t cap = 0;
      int max_n = 0, n_made = 0;
      bool broken = false;
      // a free buffer; while fewer than max_n exist a new one is made instead of waiting (by the feeding thread that needs
      // it: page-locking a tenth of a gigabyte takes tens of milliseconds, and the feeders start one after the other)
      int take() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          if (!free_.empty()) { const int k = free_.back(); free_.pop_back(); return k; }
          if (broken) return -1;
          if (n_made < max_n) {
            const int k = n_made++;
            lk.unlock();
            void* q = nullptr;
            const bool ok = svdss_host_alloc((int64_t)cap, &q) == SVDSS_OK && q;
            lk.lock();
            if (!ok) { broken = true; cv.notify_all(); return -1; }
            buf[(size_t)k] = (uint8_t*)q;
            return k;
          }
          cv.wait(lk);
        }
      }
      void give(int k) { { std::lock_guard<std::mutex> lk(m); free_.push_back(k); } cv.notify_all(); }
    } pool;
    pool.cap = (size_t)target + (size_t)target / 8 + ((size_t)32 << 20);   // (literals-only members: at most ~1.001 x the records, which grow by 4 bytes each)
    pool.max_n = per_gpu + 6;
    pool.buf.assign((size_t)pool.max_n, nullptr);
    static thread_local int tl_slot = -1;
    // (the pool is the first region's: a later region's members wait in plain memory until the regions in front are written,
    // and would hold every buffer of the pool while the first region's feeders wait for one)
    auto run_for = [&](size_t g, bool use_pool) {
      return DeviceBamSelect::RunFn([&, g, use_pool](svdss_bam_stream_t* st, int64_t seq, int32_t last, int64_t sk, size_t, int32_t nc, const uint8_t* const* comp,
                                                     const int64_t* cb, const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* nb,
                                                     svdss_bam_batch_t** batch) {
        { std::unique_lock<std::mutex> lk(gate.m); gate.cv.wait(lk, [&] { return gate.open; }); }
        tl_slot = use_pool ? pool.take() : -1;
        const int rc = svdss_bam_smooth_run(st, seq, last, sk, sms[g % sms.size()], al_accuracy, tl_slot >= 0 ? pool.buf[(size_t)tl_slot] : nullptr,
                                            tl_slot >= 0 ? (int64_t)pool.cap : 0, nc, comp, cb, blocks, crc, nb, batch);
        if (rc != SVDSS_OK && tl_slot >= 0) { pool.give(tl_slot); tl_slot = -1; }
        return rc;
      });
    };
    DeviceBamSelect::CollectFn collect = [&](const svdss_bam_batch_t* b, SelectedBatch& out) {
      svdss_bam_smoothed_t r;
      (void)svdss_bam_batch_smoothed(b, &r);
      out.n_records = (uint64_t)r.n_records; out.n_kept = (uint64_t)r.n_kept;
      for (int k = 0; k < 4; ++k) out.n_xf[k] = (uint64_t)r.n_xf[k];
      if (tl_slot >= 0 && r.bgzf == pool.buf[(size_t)tl_slot]) { out.ext = r.bgzf; out.ext_n = (size_t)r.bgzf_bytes; out.ext_slot = tl_slot; }
      else {
        out.bytes.assign(r.bgzf, r.bgzf + r.bgzf_bytes);
        if (tl_slot >= 0) pool.give(tl_slot);
      }
      tl_slot = -1;
      for (int k = 0; k < 8; ++k) out.stage_s[k] = r.stage_ms[k] * 1e-3;
      out.inflate_kernel_s = r.inflate_kernel_ms * 1e-3;
    };
    if (dbg) fprintf(stderr, "[smooth] BAM header read, output prefix set at +%.3f s\n", since());
    std::unique_ptr<DeviceBamSelect> rd;
    std::unique_ptr<ShardedBamSelect> rds;
    if (n_regions == 1) rd.reset(new DeviceBamSelect(o.bam, one, dev0, n_ref, skip, per_gpu, target, run_for(0, true), collect, stream));
    else {
      ShardedBamSelect::Hooks hk;
      hk.run = [&](size_t g, bool seam) { return run_for(g, g == 0 && !seam); };
      hk.collect = [&](size_t, bool) { return collect; };
      // (the output's header is in front of the first region's stream; the first region never runs again)
      hk.stream = [&](size_t g) { svdss_bam_stream_t* st = g == 0 ? stream : nullptr; if (g == 0) stream = nullptr; return st; };
      hk.device = [&](size_t g) { return (int)(g % (size_t)n_phys); };
      rds.reset(new ShardedBamSelect(o.bam, hk, n_ref, skip, per_gpu, target, cuts));
    }
    if (dbg) fprintf(stderr, "[smooth] reader of the smoothing pass started at +%.3f s\n", since());
    // the chromosomes the BAM names, in its order, one device buffer (svdss_ref_upload_parts: no concatenation on the host)
    std::vector<int32_t> tid_map(names.size(), -1);
    std::vector<const uint8_t*> parts;
    std::vector<int64_t> plen;
    for (size_t t = 0; t < names.size(); ++t) {
      auto it = chrom.find(names[t]);
      if (it == chrom.end()) continue;
      tid_map[t] = (int32_t)parts.size();
      parts.push_back((const uint8_t*)it->second.data());
      plen.push_back((int64_t)it->second.size());
    }
    {
      std::vector<int> rcs(n_sm, SVDSS_OK);
      std::vector<std::thread> up;
      auto upload = [&](size_t d) {
        rcs[d] = svdss_ref_upload_parts(parts.data(), plen.data(), (int32_t)parts.size(), (int)d, &drefs[d]);
        if (rcs[d] == SVDSS_OK) rcs[d] = svdss_bam_smooth_create(drefs[d], tid_map.data(), (int32_t)tid_map.size(), (int32_t)o.min_mapq, &sms[d]);
      };
      for (size_t d = 1; d < n_sm; ++d) up.emplace_back(upload, d);
      upload(0);
      for (std::thread& t : up) t.join();
      for (size_t d = 0; d < n_sm; ++d)
        if (rcs[d] != SVDSS_OK) die(std::string("chromosomes to GPU ") + std::to_string(d) + ": " + svdss_strerror(rcs[d]) + " " + svdss_last_hip_error());
    }
    if (dbg) fprintf(stderr, "[smooth] chromosomes uploaded at +%.3f s (%zu GPU(s), %zu region(s))\n", since(), n_sm, n_regions);
    if (dbg) fprintf(stderr, "[smooth] reference read in %.3f s, on the device at +%.3f s\n", fasta_s, since());
    // compute_maxaccuracy (smoother.cpp:259-346): the mismatch rates of the first 10,000 records that fit, their percentile
    {
      std::vector<double> acc;
      DeviceBamSelect::RunFn mrun = [&](svdss_bam_stream_t* st, int64_t seq, int32_t last, int64_t sk, size_t, int32_t nc, const uint8_t* const* comp,
                                       const int64_t* cb, const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* nb,
                                       svdss_bam_batch_t** batch) {
        return svdss_bam_smooth_measure(st, seq, last, sk, sms[0], nc, comp, cb, blocks, crc, nb, batch);
      };
      DeviceBamSelect::CollectFn mcollect = [](const svdss_bam_batch_t* b, SelectedBatch& out) {
        svdss_bam_smoothed_t r;
        (void)svdss_bam_batch_smoothed(b, &r);
        out.n_records = (uint64_t)r.n_records; out.n_kept = (uint64_t)r.n_kept;
        out.match_mismatch.assign(r.match_mismatch, r.match_mismatch + 2 * r.n_kept);
        out.fits.assign(r.fits, r.fits + r.n_kept);
      };
      // (10,000 records are a few tens of megabytes: small batches, two feeders, and the reader is dropped as soon as it has them)
      DeviceBamSelect pre(o.bam, one, dev0, n_ref, skip, 2, std::min<int64_t>(target, (int64_t)48 << 20), mrun, mcollect);
      while (acc.size() < 10000) {
        std::unique_ptr<SelectedBatch> b = pre.next();
        if (!b) { if (!pre.error().empty()) die("error reading " + o.bam + ": " + pre.error()); break; }
        for (size_t k = 0; k < b->fits.size() && acc.size() < 10000; ++k) {
          if (!b->fits[k]) continue;
          acc.push_back((double)b->match_mismatch[2 * k + 1] / (double)b->match_mismatch[2 * k]);
        }
      }
      if (!acc.empty()) {
        std::sort(acc.begin(), acc.end());
        const double id = (double)(acc.size() - 1) * (double)o.accp;   // percentile(), smoother.cpp:246-255
        const double lo = floor(id), hi = ceil(id), h = id - lo;
        al_accuracy = (1.0 - h) * acc[(size_t)lo] + h * acc[(size_t)hi];
      }
    }
    if (dbg) fprintf(stderr, "[smooth] accuracy threshold %.6g at +%.3f s\n", al_accuracy, since());
    { std::lock_guard<std::mutex> lk(gate.m); gate.open = true; }
    gate.cv.notify_all();
    uint64_t n_rec = 0, n_kept = 0, n_xf[4] = {0, 0, 0, 0}, out_bytes = 0;
    std::atomic<bool> write_ok{true};   // (set by several writer threads)
    {
      // side-by-side writers for a regular file
      fflush(stdout);
      const off_t pos0 = lseek(STDOUT_FILENO, 0, SEEK_CUR);
      struct stat sb;
      // (pwrite ignores its offset on an O_APPEND descriptor -- `SVDSS smooth ... >> out.bam` -- and the batches would land in
      // completion order: such a stdout takes the ordered path)
      const int fl = fcntl(STDOUT_FILENO, F_GETFL);
      const bool seekable = pos0 >= 0 && fstat(STDOUT_FILENO, &sb) == 0 && S_ISREG(sb.st_mode) && fl >= 0 && !(fl & O_APPEND) &&
                            !getenv("SVDSS_SMOOTH_SERIAL_WRITE");
      struct WJob { std::unique_ptr<SelectedBatch> b; off_t at; };
      std::mutex wm; std::condition_variable wcv;
      std::deque<WJob> wq;
      bool wclosed = false;
      std::vector<std::thread> writers;
      auto put = [&](const uint8_t* p, size_t n, off_t at) {
        while (n) {
          const ssize_t w = pwrite(STDOUT_FILENO, p, n, at);
          if (w <= 0) { write_ok = false; return; }
          p += w; n -= (size_t)w; at += w;
        }
      };
      if (seekable)
        for (int t = 0; t < (getenv("SVDSS_SMOOTH_WRITERS") ? std::max(1, atoi(getenv("SVDSS_SMOOTH_WRITERS"))) : 4); ++t)
          writers.emplace_back([&] {
            for (;;) {
              WJob j;
              {
                std::unique_lock<std::mutex> lk(wm);
                wcv.wait(lk, [&] { return !wq.empty() || wclosed; });
                if (wq.empty()) return;
                j = std::move(wq.front());
                wq.pop_front();
              }
              wcv.notify_all();
              if (j.b->ext) { put(j.b->ext, j.b->ext_n, j.at); pool.give(j.b->ext_slot); }
              else put(j.b->bytes.data(), j.b->bytes.size(), j.at);
            }
          });
      if (dbg) fprintf(stderr, "[smooth] streaming from +%.3f s\n", since());
      double st_s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, inf_s = 0;
      off_t at = pos0 < 0 ? 0 : pos0;
      while (std::unique_ptr<SelectedBatch> b = rd ? rd->next() : rds->next()) {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t nb = b->ext ? b->ext_n : b->bytes.size();
        n_rec += b->n_records; n_kept += b->n_kept; out_bytes += nb;
        for (int k = 0; k < 4; ++k) n_xf[k] += b->n_xf[k];
        for (int k = 0; k < 8; ++k) st_s[k] += b->stage_s[k];
        inf_s += b->inflate_kernel_s;
        if (seekable) {
          {
            std::unique_lock<std::mutex> lk(wm);
            wcv.wait(lk, [&] { return wq.size() < 4; });
            wq.push_back(WJob{std::move(b), at});
          }
          wcv.notify_all();
        } else {
          const uint8_t* p = b->ext ? b->ext : b->bytes.data();
          if (nb && fwrite(p, 1, nb, stdout) != nb) write_ok = false;
          if (b->ext) pool.give(b->ext_slot);
        }
        at += (off_t)nb;
        t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      { std::lock_guard<std::mutex> lk(wm); wclosed = true; }
      wcv.notify_all();
      for (std::thread& t : writers) t.join();
      if (seekable && lseek(STDOUT_FILENO, at, SEEK_SET) < 0) write_ok = false;   // (the EOF marker goes behind the last batch)
      const std::string rerr = rd ? rd->error() : rds->error();
      if (rds && dbg)
        fprintf(stderr, "[smooth] %zu regions on %zu GPU(s): %lld seam(s) proved, %lld region(s) run again\n", rds->n_regions(), n_sm, (long long)rds->seams_run(),
                (long long)rds->regions_run_again());
      rd.reset();
      rds.reset();
      for (uint8_t* q : pool.buf) if (q) svdss_host_free(q);
      if (!rerr.empty()) die("error reading " + o.bam + ": " + rerr);
      if (dbg)
        fprintf(stderr, "[smooth] device path: %llu records, %llu kept (XF 0/1/2/3: %llu %llu %llu %llu), %llu BGZF bytes; feeder seconds: front %.3f "
                "turn wait %.3f turn %.3f walk %.3f rebuild %.3f output turn %.3f deflate + down %.3f (inflate kernels %.3f); writing %.3f\n",
                (unsigned long long)n_rec, (unsigned long long)n_kept, (unsigned long long)n_xf[0], (unsigned long long)n_xf[1],
                (unsigned long long)n_xf[2], (unsigned long long)n_xf[3], (unsigned long long)out_bytes, st_s[0], st_s[1], st_s[2], st_s[3],
                st_s[4], st_s[5], st_s[6], inf_s, t_write);
    }
    static const uint8_t eof_marker[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fwrite(eof_marker, 1, 28, stdout) != 28 || fflush(stdout) != 0) write_ok = false;
    for (svdss_bam_smooth_t* q : sms) svdss_bam_smooth_free(q);
    for (svdss_ref_t* q : drefs) svdss_ref_free(q);
    if (!write_ok) die("error writing the BAM to stdout");
    if (dbg) fprintf(stderr, "[smooth] done at +%.3f s\n", since());
    if (!getenv("SVDSS_CLEAN_EXIT")) {
      fprintf(stderr, "[smooth] [info] All done!\n");
      fflush(stderr);
      _exit(0);
    }
    return 0;
  }
  // compute_maxaccuracy (smoother.cpp:259-346)
  double al_accuracy;
  {
    BamReader bam(o.bam);
    bam.set_ahead(2);   // (10,000 records are a few chunks)
    svdss_enable_gpu_inflate(bam);
    if (!bam.ok() || !bam.read_header()) die("cannot read " + o.bam + ": " + bam.error());
    std::vector<double> acc;
    BamRecord r;
    while (acc.size() < 10000 && bam.next(r) > 0) {
      if (!eligible(r, bam.ref_names())) continue;
      double nm, nx;
      const std::string& ref = chrom[bam.ref_names()[(size_t)r.tid]];
      // (a record the smoothing pass will refuse -- XF = 3 -- has no defined mismatch rate: the reference walks off its
      // buffers on it, smoother.cpp:259-346)
      if (!cigar_fits(r, (size_t)r.l_seq, ref.size())) continue;
      mismatch_counts(r, r.seq_string(), ref, nm, nx);
      acc.push_back(nx / nm);
    }
    if (acc.empty()) al_accuracy = 0.0;
    else {
      std::sort(acc.begin(), acc.end());
      const double id = (double)(acc.size() - 1) * (double)o.accp;   // percentile(), smoother.cpp:246-255
      const double lo = floor(id), hi = ceil(id), h = id - lo;
      al_accuracy = (1.0 - h) * acc[(size_t)lo] + h * acc[(size_t)hi];
    }
  }
  if (dbg) fprintf(stderr, "[smooth] accuracy threshold at +%.3f s\n", since());
  BamReader bam(o.bam);
  svdss_enable_gpu_inflate(bam);
  if (!bam.ok() || !bam.read_header()) die("cannot read " + o.bam + ": " + bam.error());
  const int T = std::max(1, o.threads);
  BgzfWriter w(stdout, T);
  svdss_enable_gpu_deflate(w);   // (csrc/deflate.hip; SVDSS_GPU_DEFLATE=0: libdeflate / zlib on the host)
  bam_write_header(w, bam.header_text(), bam.ref_names(), bam.ref_lens());
  // smooth_read (smoother.cpp:84-232) of one record into its serialised BAM bytes
  auto smooth_one = [&](const BamRecord& r, ByteSink& sink) {
    const std::string& ref = chrom.at(bam.ref_names()[(size_t)r.tid]);
    const std::string seq = r.seq_string();
    {
      // the walk below indexes ref[pos ..] and seq[..] by the CIGAR: an alignment that overhangs the contig end or
      // whose CIGAR does not add up to l_seq is passed through unchanged with XF = 3, the reference's tag for a record
      // it could not rebuild consistently (smoother.cpp:219-228)
      if (!cigar_fits(r, seq.size(), ref.size())) {
        std::vector<uint8_t> aux3 = r.aux;
        set_xf(aux3, 3);
        write_record(sink, r, r.cigar, seq, r.qual, aux3);
        return;
      }
    }
    std::string nseq;
    std::vector<uint8_t> nqual;
    std::vector<uint32_t> ncig;
    double nm = 0, nx = 0;
    size_t ref_off = (size_t)r.pos, q_off = 0;
    uint32_t m_diff = 0;
    bool ignore = true;
    auto qcopy = [&](size_t start, size_t len) {   // may run past the read end like the reference's memcpy
      for (size_t i = 0; i < len; ++i) nqual.push_back(start + i < r.qual.size() ? r.qual[start + i] : 255);
    };
    for (uint32_t c : r.cigar) {
      const uint32_t l = c >> 4, op = c & 0xf;
      if (is_m(op)) {
        nseq.append(ref, ref_off, l);
        qcopy(q_off, l);
        for (uint32_t j = 0; j < l; ++j) (ref[ref_off + j] == seq[q_off + j]) ? ++nm : ++nx;
        ref_off += l; q_off += l;
        if (!ncig.empty() && (ncig.back() & 0xf) == 0) ncig.back() += (l + m_diff) << 4;
        else ncig.push_back(((l + m_diff) << 4) | 0);
        m_diff = 0;
      } else if (op == 1) {
        if ((int)l > MIN_INDEL) { ignore = false; nseq.append(seq, q_off, l); qcopy(q_off, l); ncig.push_back(c); }
        q_off += l;
      } else if (op == 2) {
        if ((int)l <= MIN_INDEL) { nseq.append(ref, ref_off, l); qcopy(q_off, l); m_diff += l; }
        else { ignore = false; ncig.push_back(c); }
        ref_off += l;
      } else if (op == 4) {
        ignore = false;
        nseq.append(seq, q_off, l);
        qcopy(q_off, l);
        q_off += l;
        ncig.push_back(c);
      } else break;
    }
    std::vector<uint8_t> aux = r.aux;
    if (nx / nm > al_accuracy) { set_xf(aux, 1); write_record(sink, r, r.cigar, seq, r.qual, aux); }
    else if (ignore) { set_xf(aux, 2); write_record(sink, r, r.cigar, seq, r.qual, aux); }
    else { set_xf(aux, 0); write_record(sink, r, ncig, nseq, nqual, aux); }
  };
  // the walk runs on the GPU (SVDSS_SMOOTH_HOST=1: the host code above, a developer switch); the chromosomes go up once, in
  // BAM header order
  svdss_ref_t* dref = nullptr;
  std::vector<int32_t> tid_map(bam.ref_names().size(), -1);
  // (no GPU and no SVDSS_SMOOTH_HOST=1: the command fails rather than quietly running the host walk)
  if (!getenv("SVDSS_SMOOTH_HOST") && svdss_device_count() <= 0)
    die("no GPU found: SVDSS smooth walks the alignments on the GPU (SVDSS_SMOOTH_HOST=1 runs the host code instead)");
  if (!getenv("SVDSS_SMOOTH_HOST")) {
    std::string all;
    std::vector<int64_t> off(1, 0);
    for (size_t t = 0; t < bam.ref_names().size(); ++t) {
      auto it = chrom.find(bam.ref_names()[t]);
      if (it == chrom.end()) continue;
      tid_map[t] = (int32_t)off.size() - 1;
      all += it->second;
      off.push_back((int64_t)all.size());
    }
    if (svdss_ref_upload((const uint8_t*)all.data(), off.data(), (int32_t)off.size() - 1, 0, &dref) != SVDSS_OK)
      die(std::string("svdss_ref_upload: ") + svdss_last_hip_error());
  }
  // batches of eligible records: read in order, smoothed by T workers, written in order (the reference's
  // batch loop, smoother.cpp:441-537)
  // Three stages run side by side, joined by short queues: a thread reads and filters the next batch (BGZF inflate on
  // the GPU or the host workers), this thread smooths the current one (GPU kernel + T finishing threads, or T host
  // workers), a thread deflates and writes the previous one (T workers).  Output order = input order.
  const size_t batch_size = 4096;
  struct Item { std::vector<BamRecord> batch; std::vector<ByteSink> outs; };
  struct ItemQueue {
    std::mutex m; std::condition_variable cv; std::deque<std::unique_ptr<Item>> q; bool closed = false; size_t cap = 2;
    void push(std::unique_ptr<Item> it) {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return q.size() < cap; });
      q.push_back(std::move(it));
      cv.notify_all();
    }
    std::unique_ptr<Item> pop() {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return !q.empty() || closed; });
      if (q.empty()) return nullptr;
      std::unique_ptr<Item> it = std::move(q.front());
      q.pop_front();
      cv.notify_all();
      return it;
    }
    void close() { std::lock_guard<std::mutex> lk(m); closed = true; cv.notify_all(); }
  } q_read, q_write;
  int rc = 1;
  std::thread reader([&] {
    int r_rc = 1;
    while (r_rc > 0) {
      const auto t0 = std::chrono::steady_clock::now();
      std::unique_ptr<Item> it(new Item);
      while (it->batch.size() < batch_size) {
        BamRecord r;
        r_rc = bam.next(r);
        if (r_rc <= 0) break;
        if (!eligible(r, bam.ref_names())) continue;   // dropped from the output (smoother.cpp:509-537)
        it->batch.push_back(std::move(r));
      }
      t_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (!it->batch.empty()) q_read.push(std::move(it));
    }
    rc = r_rc;
    q_read.close();
  });
  bool write_ok = true;
  std::thread writer([&] {
    while (std::unique_ptr<Item> it = q_write.pop()) {
      const auto t0 = std::chrono::steady_clock::now();
      for (const ByteSink& sk : it->outs) w.write(sk.v.data(), sk.v.size());
      t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    write_ok = w.finish();
  });
  PinBuf pb_s4, pb_q, pb_cig, pb_o4, pb_oq, pb_ocig;   // the batch's packed bases / qualities / CIGARs, in and out
  auto process = [&](std::vector<BamRecord>& batch, std::vector<ByteSink>& outs) {
    outs.assign(batch.size(), ByteSink());
    if (dref && !batch.empty()) {
      // the CIGAR walk of the whole batch on the GPU (csrc/place.hip, smooth_kernel: one wavefront per record); the host
      // keeps what is per record and tiny: the consistency check, the XF decision, the record header
      std::vector<size_t> idx;                      // batch index of the records that go to the GPU
      std::vector<int32_t> tid, pos, lq;
      std::vector<int64_t> cig_off(1, 0), s4_off, q_off, cap_off(1, 0);
      // first the sizes (cheap, in order), then the bytes (T threads into page-locked buffers kept from batch to batch)
      size_t s4_bytes = 0, q_bytes = 0;
      for (size_t i = 0; i < batch.size(); ++i) {
        const BamRecord& r = batch[i];
        const std::string& ref = chrom.at(bam.ref_names()[(size_t)r.tid]);
        size_t rl = 0, qlen = 0;
        for (uint32_t c : r.cigar) {
          const uint32_t l = c >> 4, op = c & 0xf;
          if (is_m(op)) { rl += l; qlen += l; }
          else if (op == 1 || op == 4) qlen += l;
          else if (op == 2) rl += l;
          else break;
        }
        if (r.pos < 0 || (size_t)r.pos + rl > ref.size() || qlen != (size_t)r.l_seq || r.qual.size() != (size_t)r.l_seq) {
          smooth_one(r, outs[i]);                   // inconsistent record: the host path tags it XF = 3
          continue;
        }
        idx.push_back(i);
        tid.push_back(tid_map[(size_t)r.tid]);
        pos.push_back(r.pos);
        lq.push_back(r.l_seq);
        cig_off.push_back(cig_off.back() + (int64_t)r.cigar.size());
        s4_off.push_back((int64_t)s4_bytes);
        s4_bytes += r.seq4.size();
        q_off.push_back((int64_t)q_bytes);
        q_bytes += r.qual.size();
        cap_off.push_back(cap_off.back() + (int64_t)((qlen + rl + 1) & ~(size_t)1));
      }
      const size_t n = idx.size();
      if (n) {
        const size_t n_cig_in = (size_t)cig_off.back();
        uint8_t* s4 = pb_s4.ensure(s4_bytes + 16);
        uint8_t* ql = pb_q.ensure(q_bytes + 16);
        uint32_t* cig = (uint32_t*)pb_cig.ensure(4 * n_cig_in + 16);
        uint8_t* o4 = pb_o4.ensure((size_t)cap_off.back() / 2 + 8);
        uint8_t* oq = pb_oq.ensure((size_t)cap_off.back() + 8);
        uint32_t* ocig = (uint32_t*)pb_ocig.ensure(4 * (n_cig_in + 1) + 16);
        std::vector<uint8_t> oign(n);
        std::vector<int32_t> oncig(n), olen(n);
        std::vector<int64_t> onm(2 * n);
        {
          auto pack = [&](size_t t, size_t nt) {
            for (size_t k = n * t / nt; k < n * (t + 1) / nt; ++k) {
              const BamRecord& r = batch[idx[k]];
              memcpy(s4 + s4_off[k], r.seq4.data(), r.seq4.size());
              memcpy(ql + q_off[k], r.qual.data(), r.qual.size());
              memcpy(cig + cig_off[k], r.cigar.data(), 4 * r.cigar.size());
            }
          };
          const size_t nt = std::min<size_t>((size_t)T, std::max<size_t>(1, n / 64));
          if (nt <= 1) pack(0, 1);
          else {
            std::vector<std::thread> pool;
            for (size_t t = 1; t < nt; ++t) pool.emplace_back(pack, t, nt);
            pack(0, nt);
            for (std::thread& th : pool) th.join();
          }
        }
        if (svdss_smooth_batch(dref, tid.data(), pos.data(), cig, cig_off.data(), s4, s4_off.data(), ql,
                               q_off.data(), lq.data(), cap_off.data(), (int64_t)n, o4, oq, ocig,
                               oncig.data(), olen.data(), onm.data(), oign.data()) != SVDSS_OK)
          die(std::string("svdss_smooth_batch: ") + svdss_last_hip_error());
        auto finish = [&](size_t t, size_t nt) {
          for (size_t k = t; k < n; k += nt) {
            const BamRecord& r = batch[idx[k]];
            std::vector<uint8_t> aux = r.aux;
            const double nm = (double)onm[2 * k], nx = (double)onm[2 * k + 1];
            if (nx / nm > al_accuracy) { set_xf(aux, 1); write_record(outs[idx[k]], r, r.cigar, r.seq_string(), r.qual, aux); }
            else if (oign[k]) { set_xf(aux, 2); write_record(outs[idx[k]], r, r.cigar, r.seq_string(), r.qual, aux); }
            else {
              set_xf(aux, 0);
              write_record_packed(outs[idx[k]], r, ocig + cig_off[k], (size_t)oncig[k], o4 + cap_off[k] / 2,
                                  olen[k], oq + cap_off[k], aux);
            }
          }
        };
        const size_t nt = std::min<size_t>((size_t)T, n);
        if (nt <= 1) finish(0, 1);
        else {
          std::vector<std::thread> pool;
          for (size_t t = 1; t < nt; ++t) pool.emplace_back(finish, t, nt);
          finish(0, nt);
          for (std::thread& th : pool) th.join();
        }
      }
      return;
    }
    auto work = [&](size_t t, size_t nt) { for (size_t i = t; i < batch.size(); i += nt) smooth_one(batch[i], outs[i]); };
    const size_t nt = std::min<size_t>((size_t)T, batch.size());
    if (nt <= 1) work(0, 1);
    else {
      std::vector<std::thread> pool;
      for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t, nt);
      work(0, nt);
      for (std::thread& th : pool) th.join();
    }
  };
  if (dbg) fprintf(stderr, "[smooth] reference on the device at +%.3f s\n", since());
  while (std::unique_ptr<Item> it = q_read.pop()) {
    const auto t0 = std::chrono::steady_clock::now();
    process(it->batch, it->outs);
    t_proc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<BamRecord>().swap(it->batch);
    q_write.push(std::move(it));
  }
  q_write.close();
  reader.join();
  writer.join();
  if (dbg) fprintf(stderr, "[smooth] done at +%.3f s; stage busy seconds: read + filter %.3f, smooth %.3f, deflate + write %.3f\n", since(), t_read, t_proc, t_write);
  svdss_ref_free(dref);
  if (rc < 0) die("error reading " + o.bam + ": " + bam.error());
  if (!write_ok) die("error writing the BAM to stdout");
  if (!getenv("SVDSS_CLEAN_EXIT")) {   // (see main_search: the teardown of page-locked buffers is left to the OS)
    bam.report();
    fprintf(stderr, "[smooth] [info] All done!\n");
    fflush(stdout);
    fflush(stderr);
    _exit(0);
  }
  return 0;
}
