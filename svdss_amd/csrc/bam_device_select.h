// bam_device_select.h -- host side of the device path for callers that need whole records of a few reads (`SVDSS call`,
// csrc/bam_device.hip svdss_bam_select_run): the BAM header probe, and a reader that hands out, in file order, the records
// a svdss_bam_filter_t keeps.  Scanner (loader threads) -> batcher -> feeding threads (one batch object each) -> ordered
// output; the caller sees plain record bytes.
#pragma once
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svdss_hip.h"
#include "bam_reader.h"
#include "bgzf_scanner.h"

// The BAM header read on the host (the first BGZF members, zlib / libdeflate): number of reference sequences and the
// length of the header in the inflated stream -- where the first record begins (sam_hdr_read, ping_pong.cpp:248).
inline bool bam_header_probe(const std::string& path, int32_t& n_ref, int64_t& skip, std::string& err, std::vector<std::string>* ref_names) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open file"; return false; }
  std::vector<uint8_t> comp, buf;
  size_t pos = 0;
  bool eof = false;
  BgzfInflater inf;
  auto more = [&]() -> bool {        // one more member inflated onto buf
    for (;;) {
      if (pos + 18 <= comp.size()) {
        const uint8_t* h = comp.data() + pos;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "not a BAM file"; return false; }
        uint16_t xlen;
        memcpy(&xlen, h + 10, 2);
        int bsize = -1;
        if (pos + 12 + xlen <= comp.size()) {
          for (size_t o = 0; o + 4 <= xlen;) {
            const uint8_t* x = h + 12 + o;
            uint16_t slen;
            memcpy(&slen, x + 2, 2);
            if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) { uint16_t v; memcpy(&v, x + 4, 2); bsize = v; break; }
            o += 4u + slen;
          }
          if (bsize < 0 || (size_t)bsize + 1 < 12u + xlen + 8u) { err = "BGZF block without BC field"; return false; }
          if (pos + (size_t)bsize + 1 <= comp.size()) {
            const size_t clen = (size_t)bsize + 1 - 12 - xlen - 8;
            uint32_t crc, isize;
            memcpy(&crc, h + 12 + xlen + clen, 4);
            memcpy(&isize, h + 12 + xlen + clen + 4, 4);
            if (isize > 65536u) { err = "bad BGZF block"; return false; }
            const size_t at = buf.size();
            buf.resize(at + isize);
            if (isize) if (const char* e = inf.run(h + 12 + xlen, clen, buf.data() + at, isize, crc)) { err = e; return false; }
            pos += (size_t)bsize + 1;
            return true;
          }
        }
      }
      if (eof) { err = "truncated header"; return false; }
      const size_t at = comp.size();
      comp.resize(at + ((size_t)256 << 10));
      const size_t got = fread(comp.data() + at, 1, (size_t)256 << 10, f);
      comp.resize(at + got);
      if (got == 0) eof = true;
    }
  };
  auto need = [&](size_t n) -> bool { while (buf.size() < n) if (!more()) return false; return true; };
  bool ok = false;
  do {
    if (!need(12)) break;
    if (memcmp(buf.data(), "BAM\1", 4) != 0) { err = "not a BAM file"; break; }
    int32_t l_text;
    memcpy(&l_text, buf.data() + 4, 4);
    if (l_text < 0) { err = "corrupt header"; break; }
    if (!need(12 + (size_t)l_text)) break;
    memcpy(&n_ref, buf.data() + 8 + l_text, 4);
    if (n_ref < 0) { err = "corrupt header"; break; }
    size_t o = 12 + (size_t)l_text;
    bool bad = false;
    for (int32_t i = 0; i < n_ref && !bad; ++i) {
      if (!need(o + 4)) { bad = true; break; }
      int32_t l_name;
      memcpy(&l_name, buf.data() + o, 4);
      if (l_name < 0) { err = "corrupt header"; bad = true; break; }
      if (!need(o + 4 + (size_t)l_name + 4)) { bad = true; break; }
      if (ref_names) {
        std::string nm((const char*)buf.data() + o + 4, (size_t)l_name);
        if (!nm.empty() && nm.back() == '\0') nm.pop_back();
        ref_names->push_back(nm);
      }
      o += 4 + (size_t)l_name + 4;
    }
    if (bad) break;
    skip = (int64_t)o;
    ok = true;
  } while (false);
  fclose(f);
  return ok;
}


// the kept records of one device batch, in file order: record k = bytes[off[k] + 4 ..), block_size at bytes[off[k]]
struct SelectedBatch {
  std::vector<uint8_t> bytes;   // (smoothing: the batch's BGZF members)
  std::vector<int64_t> off;
  uint64_t n_records = 0;       // records of the batch, kept or not
  bool slim = false;            // the kept records are slim ones (svdss_bam_selection_t::slim)
  // smoothing (svdss_bam_smooth_run / _measure): kept records, by XF value; matches / mismatches and "CIGAR fits" per kept record
  uint64_t n_kept = 0, n_xf[4] = {0, 0, 0, 0};
  std::vector<int64_t> match_mismatch;
  std::vector<uint8_t> fits;
  const uint8_t* ext = nullptr;   // smoothing: the BGZF members in a page-locked buffer of the caller's pool (ext_n bytes) ...
  size_t ext_n = 0;
  int ext_slot = -1;              // ... and which one, for its return
  double stage_s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, inflate_kernel_s = 0;
};

// a record's view (BamReader::RawView: the zero-copy form the host readers hand out) over bytes that hold it
inline bool view_of_record(const uint8_t* rec, size_t avail, BamReader::RawView& v, bool slim = false) {
  if (avail < 36) return false;
  int32_t block_size;
  memcpy(&block_size, rec, 4);
  if (block_size < 32 || (size_t)block_size + 4 > avail) return false;
  const uint8_t* core = rec + 4;
  v.own.reset();
  v.p = core;
  uint16_t n_cigar;
  memcpy(&v.tid, core, 4);
  memcpy(&v.pos, core + 4, 4);
  v.l_name = core[8];
  v.mapq = core[9];
  memcpy(&n_cigar, core + 12, 2);
  v.n_cigar = n_cigar;
  memcpy(&v.flag, core + 14, 2);
  memcpy(&v.l_seq, core + 16, 4);
  if (v.l_seq < 0) return false;
  const size_t head = 32 + (size_t)v.l_name + 4u * v.n_cigar + ((size_t)v.l_seq + 1) / 2 + (slim ? 0 : (size_t)v.l_seq);
  if (head > (size_t)block_size) return false;
  v.noqual = slim;
  v.l_aux = (uint32_t)((size_t)block_size - head);
  return true;
}

struct BamSelectRegion { size_t begin = 0, end = 0; bool open_start = false, open_end = false; std::vector<uint8_t> carry; int loaders = 8; size_t pending = 64; };

class DeviceBamSelect {
 public:
  // what a feeding thread does with a batch (default: svdss_bam_select_run with the device's filter) and how its result
  // becomes a SelectedBatch -- `SVDSS smooth` runs svdss_bam_smooth_run / _measure through the same scanner, batcher,
  // feeders and ordered hand-over
  typedef std::function<int(svdss_bam_stream_t*, int64_t seq, int32_t is_last, int64_t skip, size_t dev, int32_t n_chunks, const uint8_t* const* comp,
                            const int64_t* comp_bytes, const svdss_bgzf_block_t* const* blocks, const uint32_t* const* crc, const int64_t* n_blocks,
                            svdss_bam_batch_t** batch)> RunFn;
  typedef std::function<void(const svdss_bam_batch_t*, SelectedBatch&)> CollectFn;
  // filters[d]: the filter on device d (one per GPU used); feeders: feeding threads per GPU
  // A region of the file (ShardedBamSelect below): [begin, end) at member starts (end = 0: the file's end); open_start: the
  // region begins inside a record nobody has located (svdss_bam_stream_region: the chain starts at a guess, to be proved at
  // the seam); open_end: it may end inside one; carry: the incomplete record in front of it when the region runs from a known
  // start; pending: batches that may wait for the caller (a later region's wait until the regions in front are done)
  typedef BamSelectRegion Region;
  DeviceBamSelect(const std::string& path, const std::vector<svdss_bam_filter_t*>& filters, const std::vector<int>& devices, int32_t n_ref,
                  int64_t skip, int feeders, int64_t batch_bytes, RunFn run = RunFn(), CollectFn collect = CollectFn(),
                  svdss_bam_stream_t* prepared_stream = nullptr, const Region& region = Region())
      : filters_(filters), devices_(devices), skip_(skip), target_(batch_bytes), run_(run), collect_(collect), stream_(prepared_stream),
        max_pending_(region.pending) {
    BgzfScanner::Hooks hooks;
    hooks.host_alloc = svdss_host_alloc;
    hooks.host_free = svdss_host_free;
    const size_t slab = (getenv("SVDSS_BAM_SLAB_KB") && atoll(getenv("SVDSS_BAM_SLAB_KB")) >= 64 ? (size_t)atoll(getenv("SVDSS_BAM_SLAB_KB")) << 10 : (size_t)16 << 20);
    const size_t per_batch = (size_t)target_ / slab + 2;
    feeders = std::max(1, feeders);
    sc_.reset(new BgzfScanner(path, hooks, slab, region.loaders, (size_t)region.loaders + ((size_t)filters.size() * (size_t)feeders + 3) * per_batch, region.begin,
                              region.end));
    if (!sc_->ok()) { err_ = "cannot open file"; finished_ = true; return; }
    if (!stream_ && svdss_bam_stream_create(n_ref, &stream_) != SVDSS_OK) { err_ = "out of memory"; finished_ = true; return; }
    if (region.open_start || region.open_end || !region.carry.empty())
      if (svdss_bam_stream_region(stream_, region.open_start ? 1 : 0, region.open_end ? 1 : 0, region.carry.data(), (int64_t)region.carry.size()) != SVDSS_OK) {
        err_ = "out of memory"; finished_ = true; return;
      }
    n_feeders_ = filters_.size() * (size_t)feeders;
    batcher_ = std::thread([this] { batch_loop(); });
    for (size_t d = 0; d < filters_.size(); ++d)
      for (int k = 0; k < feeders; ++k) feeders_.emplace_back([this, d] { feed_loop(d); });
  }
  ~DeviceBamSelect() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    if (batcher_.joinable()) batcher_.join();
    for (std::thread& t : feeders_) t.join();
    if (stream_) svdss_bam_stream_free(stream_);
  }
  DeviceBamSelect(const DeviceBamSelect&) = delete;
  DeviceBamSelect& operator=(const DeviceBamSelect&) = delete;

  // the next batch in file order; nullptr at the end of the file or on an error (error() says which)
  std::unique_ptr<SelectedBatch> next() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return done_.count(want_) || !err_.empty() || (finished_ && done_.empty()); });
    if (!err_.empty()) return nullptr;
    auto it = done_.find(want_);
    if (it == done_.end()) return nullptr;
    std::unique_ptr<SelectedBatch> b = std::move(it->second);
    done_.erase(it);
    ++want_;
    lk.unlock();
    cv_.notify_all();
    return b;
  }
  const std::string& error() const { return err_; }
  // blocks until batch 0 has had its turn (svdss_bam_stream_head is final), the file has ended or the run has failed
  void wait_first() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return want_ > 0 || done_.count(0) || !err_.empty() || finished_; });
  }
  // blocks until every feeding thread has ended (the stream's tail is final)
  void wait_finished() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return finished_; });
  }
  svdss_bam_stream_t* stream() const { return stream_; }
  // seconds the batcher waited for the file's loaders / for a feeding thread to take a batch (valid once the file has ended)
  double waited_for_file() const { return wait_file_s_; }
  double waited_for_feeders() const { return wait_feed_s_; }
  int64_t segments_walked_again(int64_t* n_segments) const { return stream_ ? svdss_bam_stream_rewalked(stream_, n_segments) : 0; }

 private:
  struct Job { uint64_t seq = 0; bool last = false; std::vector<std::unique_ptr<CompChunk>> chunks; };
  void fail(const std::string& e) {
    { std::lock_guard<std::mutex> lk(m_); if (err_.empty()) err_ = e; }
    cv_.notify_all();
  }
  void batch_loop() {
    std::unique_ptr<Job> cur(new Job);
    int64_t acc = 0;
    uint64_t seq = 0;
    bool any_last = false;
    auto push = [&](std::unique_ptr<Job> j) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return jobs_.size() < 2 || stop_ || !err_.empty(); });
      if (stop_ || !err_.empty()) return false;
      jobs_.push_back(std::move(j));
      lk.unlock();
      cv_.notify_all();
      return true;
    };
    for (;;) {
      const auto w0 = std::chrono::steady_clock::now();
      std::unique_ptr<CompChunk> c = sc_->next();
      wait_file_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      if (!c) break;
      acc += c->inflated;
      const bool last = c->last;
      cur->chunks.push_back(std::move(c));
      if (acc >= target_ || last) {
        cur->seq = seq++;
        cur->last = last;
        any_last = any_last || last;
        const auto w1 = std::chrono::steady_clock::now();
        if (!push(std::move(cur))) return;
        wait_feed_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - w1).count();
        cur.reset(new Job);
        acc = 0;
      }
    }
    if (!sc_->error().empty()) { fail(sc_->error()); return; }
    if (!any_last) { cur->seq = seq++; cur->last = true; if (!push(std::move(cur))) return; }
    { std::lock_guard<std::mutex> lk(m_); jobs_closed_ = true; }
    cv_.notify_all();
  }
  void feed_loop(size_t d) {
    svdss_bam_batch_t* batch = nullptr;
    std::vector<const uint8_t*> comp;
    std::vector<int64_t> comp_bytes, n_blocks;
    std::vector<const svdss_bgzf_block_t*> blocks;
    std::vector<const uint32_t*> crcs;
    for (;;) {
      std::unique_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !jobs_.empty() || jobs_closed_ || stop_ || !err_.empty(); });
        if (stop_ || !err_.empty() || jobs_.empty()) break;
        job = std::move(jobs_.front());
        jobs_.erase(jobs_.begin());
      }
      cv_.notify_all();
      comp.clear(); comp_bytes.clear(); n_blocks.clear(); blocks.clear(); crcs.clear();
      for (const std::unique_ptr<CompChunk>& c : job->chunks) {
        comp.push_back(c->data); comp_bytes.push_back((int64_t)c->n_bytes); n_blocks.push_back((int64_t)c->blocks.size());
        blocks.push_back(c->blocks.data()); crcs.push_back(c->crc.data());
      }
      const int rc = run_ ? run_(stream_, (int64_t)job->seq, job->last ? 1 : 0, job->seq == 0 ? skip_ : 0, d, (int32_t)comp.size(), comp.data(),
                                 comp_bytes.data(), blocks.data(), crcs.data(), n_blocks.data(), &batch)
                          : svdss_bam_select_run(stream_, (int64_t)job->seq, job->last ? 1 : 0, job->seq == 0 ? skip_ : 0, filters_[d], (int32_t)comp.size(),
                                                 comp.data(), comp_bytes.data(), blocks.data(), crcs.data(), n_blocks.data(), &batch);
      for (std::unique_ptr<CompChunk>& c : job->chunks) sc_->recycle(std::move(c));
      if (rc != SVDSS_OK) {
        std::string msg = batch ? svdss_bam_batch_error(batch) : "";
        if (msg.empty()) msg = svdss_bam_stream_error(stream_);
        if (msg.empty()) msg = std::string(svdss_strerror(rc)) + " " + svdss_last_hip_error();
        fail(msg);
        break;
      }
      std::unique_ptr<SelectedBatch> out(new SelectedBatch);
      if (collect_) collect_(batch, *out);
      else {
        svdss_bam_selection_t r;
        (void)svdss_bam_batch_selection(batch, &r);
        out->n_records = (uint64_t)r.n_records;
        out->slim = r.slim != 0;
        out->off.assign(r.rec_off, r.rec_off + r.n_selected + 1);
        out->bytes.assign(r.bytes, r.bytes + r.n_bytes);
        for (int k = 0; k < 8; ++k) out->stage_s[k] = r.stage_ms[k] * 1e-3;
        out->inflate_kernel_s = r.inflate_kernel_ms * 1e-3;
      }
      {
        std::unique_lock<std::mutex> lk(m_);
        const uint64_t sq = job->seq;
        cv_.wait(lk, [&] { return stop_ || done_.size() < max_pending_ || done_.begin()->first > sq; });   // (kept records are few: let the GPU run ahead)
        done_[sq] = std::move(out);
      }
      cv_.notify_all();
    }
    if (batch) svdss_bam_batch_free(batch);
    {
      std::lock_guard<std::mutex> lk(m_);
      if (++feeders_done_ == n_feeders_) finished_ = true;
    }
    cv_.notify_all();
  }

  std::vector<svdss_bam_filter_t*> filters_;
  std::vector<int> devices_;
  int64_t skip_ = 0, target_ = 0;
  double wait_file_s_ = 0, wait_feed_s_ = 0;
  RunFn run_;
  CollectFn collect_;
  std::unique_ptr<BgzfScanner> sc_;
  svdss_bam_stream_t* stream_ = nullptr;
  size_t max_pending_ = 64;
  std::thread batcher_;
  std::vector<std::thread> feeders_;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<std::unique_ptr<Job>> jobs_;
  bool jobs_closed_ = false, stop_ = false, finished_ = false;
  size_t feeders_done_ = 0, n_feeders_ = 0;
  std::map<uint64_t, std::unique_ptr<SelectedBatch>> done_;
  uint64_t want_ = 0;
  std::string err_;
};


// where `n` regions of a file begin (member starts; [0] = 0, back() = file size): fewer than n for a small file
// (SVDSS_REGION_MIN_KB, default 64 MB per region; SVDSS_REGION_SHARDS=0: one region)
inline std::vector<size_t> plan_bam_regions(const std::string& path, int n, int64_t header_inflated) {
  struct stat st;
  std::vector<size_t> cuts{0};
  if (stat(path.c_str(), &st) != 0 || st.st_size <= 0) return {0, 0};
  const size_t fsize = (size_t)st.st_size;
  const size_t min_bytes = getenv("SVDSS_REGION_MIN_KB") && atoll(getenv("SVDSS_REGION_MIN_KB")) > 0 ? (size_t)atoll(getenv("SVDSS_REGION_MIN_KB")) << 10
                                                                                                         : (size_t)64 << 20;
  // (the first region holds the whole BAM header)
  const size_t first_min = (size_t)header_inflated + ((size_t)header_inflated >> 6) + ((size_t)128 << 10);
  if (getenv("SVDSS_REGION_SHARDS") && atoi(getenv("SVDSS_REGION_SHARDS")) == 0) n = 1;
  n = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, fsize / min_bytes));
  for (int g = 1; g < n; ++g) {
    const size_t approx = std::max(first_min, (size_t)((unsigned __int128)fsize * (unsigned)g / (unsigned)n));
    if (approx >= fsize) break;
    const size_t c = BgzfScanner::member_start_near(path, approx);
    if (c > cuts.back() && c < fsize) cuts.push_back(c);
  }
  cuts.push_back(fsize);
  return cuts;
}

// `SVDSS call --gpus N` (round 6; SURVEY 8(e): the BAM's regions partition across the GPUs): the file is cut at BGZF members
// into one region per GPU, and every region has its own scanner (loader threads), batcher, feeding threads, record stream,
// filter and -- when the caller keeps one -- record store: nothing is shared on the way in, as in `SVDSS search --gpus N`
// (svdss_main.cpp).  A region that does not begin the file begins inside a record: its chain starts at a guess
// (svdss_bam_stream_region) that is PROVED when the region in front has been handed out -- its leftover + the bytes this
// region set aside must be a chain of whole records (the seam: a batch of its own through the same entry point); if they
// are not, or the region failed in any way, it runs again from the known carry.  The caller sees the batches of the file in
// file order, as from one DeviceBamSelect.
class ShardedBamSelect {
 public:
  struct Shard { svdss_bam_filter_t* filter = nullptr; int device = 0; svdss_bam_store_t* store = nullptr; svdss_bam_store_t* seam_store = nullptr; };
  // What a region's batches go through (the select / store entry point below; `SVDSS smooth --gpus N`: svdss_bam_smooth_run).
  // run(g, seam) / collect(g, seam): for the feeding threads of region g, or (seam = true) for the one batch of the seam in
  // front of it, run on the caller's thread with is_last = 1; stream(g): a prepared record stream for region g's run (nullptr
  // or no hook: a plain one) -- asked again if the region runs again; again(g): the region runs again (forget what its first
  // run left); seam_kept(g): the seam's batch stays somewhere the caller looks for it (region_has_seam)
  struct Hooks {
    std::function<DeviceBamSelect::RunFn(size_t g, bool seam)> run;
    std::function<DeviceBamSelect::CollectFn(size_t g, bool seam)> collect;
    std::function<svdss_bam_stream_t*(size_t g)> stream;
    std::function<void(size_t g)> again;
    std::function<bool(size_t g)> seam_kept;
    std::function<int(size_t g)> device;
  };
  ShardedBamSelect(const std::string& path, const std::vector<Shard>& shards, int32_t n_ref, int64_t skip, int feeders, int64_t batch_bytes,
                   const std::vector<size_t>& cuts)
      : path_(path), n_ref_(n_ref), skip_(skip), feeders_(feeders), batch_bytes_(batch_bytes) {
    hooks_.run = [shards](size_t g, bool seam) {
      const Shard S = shards[g % shards.size()];
      svdss_bam_store_t* store = seam ? S.seam_store : S.store;
      return DeviceBamSelect::RunFn([S, store, seam](svdss_bam_stream_t* s, int64_t seq, int32_t is_last, int64_t skip, size_t, int32_t n_chunks,
                                                     const uint8_t* const* comp, const int64_t* comp_bytes, const svdss_bgzf_block_t* const* blocks,
                                                     const uint32_t* const* crc, const int64_t* n_blocks, svdss_bam_batch_t** batch) {
        if (seam && store) svdss_bam_store_reset(store);
        return svdss_bam_select_store_run(s, seq, is_last, skip, S.filter, store, n_chunks, comp, comp_bytes, blocks, crc, n_blocks, batch);
      });
    };
    hooks_.again = [shards](size_t g) { if (shards[g % shards.size()].store) svdss_bam_store_reset(shards[g % shards.size()].store); };
    hooks_.seam_kept = [shards](size_t g) { return shards[g % shards.size()].seam_store != nullptr; };
    hooks_.device = [shards](size_t g) { return shards[g % shards.size()].device; };
    start(cuts);
  }
  ShardedBamSelect(const std::string& path, const Hooks& hooks, int32_t n_ref, int64_t skip, int feeders, int64_t batch_bytes, const std::vector<size_t>& cuts)
      : path_(path), hooks_(hooks), n_ref_(n_ref), skip_(skip), feeders_(feeders), batch_bytes_(batch_bytes) {
    start(cuts);
  }
  size_t n_regions() const { return regions_.size(); }
  int64_t seams_run() const { return n_seams_; }
  int64_t regions_run_again() const { return n_reruns_; }
  // the store keys of the file's batches in file order: (shard, seam?) per region: the seam's store holds key 0
  bool region_has_seam(size_t g) const { return regions_[g].seam_stored; }
  int64_t region_batches(size_t g) const { return regions_[g].n_batches; }

  std::unique_ptr<SelectedBatch> next() {
    for (;;) {
      if (cur_ >= regions_.size()) return nullptr;
      Reg& R = regions_[cur_];
      if (!R.entered) {
        R.entered = true;
        if (cur_ > 0 && !enter(cur_)) return nullptr;
        if (R.seam) { std::unique_ptr<SelectedBatch> b = std::move(R.seam); return b; }
      }
      std::unique_ptr<SelectedBatch> b = R.sel->next();
      if (b) { ++R.n_batches; return b; }
      if (!R.sel->error().empty()) { err_ = R.sel->error(); return nullptr; }
      ++cur_;
    }
  }
  const std::string& error() const { return err_; }
  double waited_for_file() const { double s = 0; for (const Reg& R : regions_) if (R.sel) s += R.sel->waited_for_file(); return s; }
  double waited_for_feeders() const { double s = 0; for (const Reg& R : regions_) if (R.sel) s += R.sel->waited_for_feeders(); return s; }

 private:
  struct Reg {
    size_t begin = 0, end = 0;
    std::unique_ptr<DeviceBamSelect> sel;
    std::unique_ptr<SelectedBatch> seam;
    bool entered = false, seam_stored = false;
    int64_t n_batches = 0;
  };
  void start(const std::vector<size_t>& cuts) {
    const size_t n = cuts.size() - 1;
    regions_.resize(n);
    for (size_t g = 0; g < n; ++g) {
      regions_[g].begin = cuts[g]; regions_[g].end = g + 1 < n ? cuts[g + 1] : 0;
      launch(g, g > 0, std::vector<uint8_t>());
    }
  }
  void launch(size_t g, bool open_start, const std::vector<uint8_t>& carry) {
    DeviceBamSelect::Region rg;
    rg.begin = regions_[g].begin; rg.end = regions_[g].end;
    rg.open_start = open_start; rg.open_end = g + 1 < regions_.size();
    rg.carry = carry;
    rg.loaders = regions_.size() > 1 ? std::max(2, 8 / (int)std::min<size_t>(regions_.size(), 4)) : 8;
    rg.pending = g == 0 ? 64 : (size_t)1 << 30;       // (a later region's results wait in memory until the regions in front are handed out)
    const std::vector<svdss_bam_filter_t*> one(1, nullptr);
    const std::vector<int> dev(1, hooks_.device ? hooks_.device(g) : 0);
    regions_[g].sel.reset(new DeviceBamSelect(path_, one, dev, n_ref_, g == 0 ? skip_ : 0, feeders_, batch_bytes_, hooks_.run(g, false),
                                              hooks_.collect ? hooks_.collect(g, false) : DeviceBamSelect::CollectFn(),
                                              hooks_.stream ? hooks_.stream(g) : nullptr, rg));
  }
  // the seam in front of region g, proved; false = the run has failed (err_)
  bool enter(size_t g) {
    Reg& P = regions_[g - 1];
    Reg& R = regions_[g];
    P.sel->wait_finished();
    const uint8_t *tail = nullptr, *head = nullptr;
    const int64_t n_tail = svdss_bam_stream_tail(P.sel->stream(), &tail);
    const std::vector<uint8_t> carry(tail, tail + (n_tail > 0 ? n_tail : 0));
    R.sel->wait_first();
    bool good = R.sel->error().empty();
    if (good) {
      const int64_t n_head = svdss_bam_stream_head(R.sel->stream(), &head);
      if ((int64_t)carry.size() + n_head > 0) { good = run_seam(g, carry, head, n_head); ++n_seams_; }
    }
    if (!good) {
      if (!err_.empty()) return false;
      // not proved (or the region failed): once more, from the record the region in front ended in
      R.sel->wait_finished();
      R.sel.reset();
      R.seam.reset();
      if (hooks_.again) hooks_.again(g);
      ++n_reruns_;
      launch(g, false, carry);
    }
    P.sel.reset();       // (its stream's tail has been copied)
    return true;
  }
  bool run_seam(size_t g, const std::vector<uint8_t>& tail, const uint8_t* head, int64_t n_head) {
    std::vector<uint8_t> bytes(tail);
    if (n_head > 0) bytes.insert(bytes.end(), head, head + n_head);
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
      crc.push_back(bgzf_crc32(bytes.data() + off, len));
    }
    comp.resize(comp.size() + 64);
    svdss_bam_stream_t* st = nullptr;
    if (svdss_bam_stream_create(n_ref_, &st) != SVDSS_OK) { err_ = "out of memory"; return false; }
    svdss_bam_batch_t* batch = nullptr;
    const uint8_t* cp = comp.data();
    const int64_t cb = (int64_t)comp.size(), nb = (int64_t)blk.size();
    const svdss_bgzf_block_t* bp = blk.data();
    const uint32_t* rp = crc.data();
    const int rc = hooks_.run(g, true)(st, 0, 1, 0, 0, 1, &cp, &cb, &bp, &rp, &nb, &batch);
    bool ok = rc == SVDSS_OK;
    if (ok) {
      std::unique_ptr<SelectedBatch> out(new SelectedBatch);
      if (hooks_.collect && hooks_.collect(g, true)) hooks_.collect(g, true)(batch, *out);
      else {
        svdss_bam_selection_t r;
        (void)svdss_bam_batch_selection(batch, &r);
        out->n_records = (uint64_t)r.n_records;
        out->slim = r.slim != 0;
        out->off.assign(r.rec_off, r.rec_off + r.n_selected + 1);
        out->bytes.assign(r.bytes, r.bytes + r.n_bytes);
      }
      regions_[g].seam = std::move(out);
      regions_[g].seam_stored = hooks_.seam_kept && hooks_.seam_kept(g);
    } else if (rc != SVDSS_EIO) {
      err_ = std::string("seam: ") + svdss_strerror(rc) + " " + (batch ? svdss_bam_batch_error(batch) : "") + " " + svdss_last_hip_error();
    }
    if (batch) svdss_bam_batch_free(batch);
    svdss_bam_stream_free(st);
    return ok;
  }
  static uint32_t bgzf_crc32(const uint8_t* p, size_t n) {
    static const std::vector<uint32_t> tab = [] { std::vector<uint32_t> t(256); for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); t[i] = c; } return t; }();
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) c = tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
  }

  std::string path_;
  Hooks hooks_;
  int32_t n_ref_ = 0;
  int64_t skip_ = 0;
  int feeders_ = 3;
  int64_t batch_bytes_ = 0;
  std::vector<Reg> regions_;
  size_t cur_ = 0;
  int64_t n_seams_ = 0, n_reruns_ = 0;
  std::string err_;
};
