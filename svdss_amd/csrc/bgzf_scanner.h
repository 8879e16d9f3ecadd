// bgzf_scanner.h -- the compressed side of a BAM file for the device path of `SVDSS search` (csrc/bam_device.hip): the
// file is read in slabs by a few loader threads (pread into page-locked buffers, in parallel), the BGZF members of every
// slab are located in file order (header, BC subfield, footer: the walk htslib's bgzf_read_block does under sam_read1,
// /root/reference/ping_pong.cpp:58,247-249), and the slabs come out in order with their block tables.  Nothing is
// inflated here.
#pragma once
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svdss_hip.h"

struct CompChunk {
  uint8_t* data = nullptr;       // the slab (+ overlap): page-locked when the hooks gave such memory
  size_t cap = 0;
  bool pinned = false;
  size_t n_bytes = 0;            // bytes read
  std::vector<svdss_bgzf_block_t> blocks;   // coff relative to data; uoff unused
  std::vector<uint32_t> crc;
  int64_t inflated = 0;          // sum of the blocks' isize
  bool last = false;             // the file ends with this chunk
};

class BgzfScanner {
 public:
  struct Hooks {
    int (*host_alloc)(int64_t, void**) = nullptr;
    void (*host_free)(void*) = nullptr;
  };
  static constexpr size_t kOverlap = (size_t)128 << 10;   // a member that starts inside a slab ends within this

  // [begin, end): the members of a region of the file (both at member starts, see member_start_near; end = 0: the file's)
  BgzfScanner(const std::string& path, Hooks hooks, size_t slab, int loaders, size_t pool_chunks, size_t begin = 0, size_t end = 0)
      : hooks_(hooks), slab_(slab), pool_chunks_(pool_chunks < 2 ? 2 : pool_chunks) {
    f_ = fopen(path.c_str(), "rb");
    struct stat st;
    if (f_ && fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode)) fsize_ = (size_t)st.st_size;
    else if (f_) { fclose(f_); f_ = nullptr; }
    size_ = end && end < fsize_ ? end : fsize_;
    begin_ = begin < size_ ? begin : size_;
    next_off_ = begin_;
    n_tickets_ = size_ > begin_ ? (size_ - begin_ + slab_ - 1) / slab_ : 0;
    if (f_) for (int i = 0; i < (loaders < 1 ? 1 : loaders); ++i) th_.emplace_back([this] { loader(); });
  }
  ~BgzfScanner() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
    for (auto& kv : ready_) release(*kv.second);
    for (std::unique_ptr<CompChunk>& c : free_) release(*c);
    if (f_) fclose(f_);
  }
  BgzfScanner(const BgzfScanner&) = delete;
  BgzfScanner& operator=(const BgzfScanner&) = delete;
  bool ok() const { return f_ != nullptr; }
  size_t file_size() const { return fsize_; }
  size_t range_bytes() const { return size_ - begin_; }

  // One BGZF member header at h (avail bytes visible): its length in the file, where its deflate stream is and how long.
  // 0 = not a member header, -1 = more bytes needed.
  static int parse_member(const uint8_t* h, size_t avail, size_t& total, size_t& data_off, size_t& data_len) {
    if (avail < 18) return -1;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return 0;
    uint16_t xlen;
    memcpy(&xlen, h + 10, 2);
    if ((size_t)12 + xlen > avail) return -1;
    int bsize = -1;
    for (size_t o = 0; o + 4 <= xlen;) {
      const uint8_t* x = h + 12 + o;
      uint16_t slen;
      memcpy(&slen, x + 2, 2);
      if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) { uint16_t v; memcpy(&v, x + 4, 2); bsize = v; break; }
      o += 4u + slen;
    }
    if (bsize < 0 || (size_t)bsize + 1 < 12u + xlen + 8u) return 0;
    total = (size_t)bsize + 1;
    data_off = (size_t)12 + xlen;
    data_len = total - 12 - xlen - 8;
    return 1;
  }
  // The first member that starts at or behind `approx`: a header whose chain of BSIZE fields leads through kChain more
  // headers (or exactly to the end of the file).  Compressed bytes can imitate one header, hardly a chain; a region cut
  // at an imitation fails its first inflate and the caller falls back to one region.  file size = none found.
  static size_t member_start_near(const std::string& path, size_t approx) {
    constexpr int kChain = 6;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    struct stat st;
    size_t fsize = 0;
    if (fstat(fileno(f), &st) == 0) fsize = (size_t)st.st_size;
    size_t found = fsize;
    if (approx < fsize) {
      const size_t want = std::min(fsize - approx, (size_t)(kChain + 2) << 16);
      std::vector<uint8_t> w(want);
      size_t got = 0;
      while (got < want) {
        const ssize_t k = pread(fileno(f), w.data() + got, want - got, (off_t)(approx + got));
        if (k <= 0) break;
        got += (size_t)k;
      }
      for (size_t p = 0; p + 18 <= got && p < ((size_t)1 << 16) + 18; ++p) {
        if (w[p] != 31 || w[p + 1] != 139) continue;
        size_t q = p;
        int n = 0;
        bool good = true;
        while (n <= kChain) {
          if (approx + q == fsize) break;                 // the chain ends with the file
          size_t tot, doff, dlen;
          const int r = parse_member(w.data() + q, got - q, tot, doff, dlen);
          if (r < 0) { good = approx + got < fsize && n >= 2; break; }   // (ran out of the window: long enough a chain)
          if (r == 0) { good = false; break; }
          q += tot;
          ++n;
          if (q > got) { good = approx + q <= fsize && n >= 2; break; }
        }
        if (good) { found = approx + p; break; }
      }
    }
    fclose(f);
    return found;
  }
  const std::string& error() const { return err_; }

  // the pool's buffers, allocated ahead of the first read (a page-locked allocation of a slab takes tens of
  // milliseconds: a caller with something else to do first -- `search` restores its index -- runs this beside it)
  void prewarm() {
    size_t tickets;
    { std::lock_guard<std::mutex> lk(m_); tickets = n_tickets_; }   // (the loaders may be shortening it: the file's last slab)
    const size_t want = std::min(std::min(pool_chunks_, tickets), (size_t)96);
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
      th.emplace_back([this, t, want] {
        for (size_t i = (size_t)t; i < want; i += 8) {
          std::unique_ptr<CompChunk> c(new CompChunk);
          if (!alloc(*c)) return;
          std::lock_guard<std::mutex> lk(m_);
          free_.push_back(std::move(c));
        }
      });
    for (std::thread& x : th) x.join();
  }

  // the next slab in file order; nullptr after the last one or on an error (error() says which)
  std::unique_ptr<CompChunk> next() {
    std::unique_lock<std::mutex> lk(m_);
    if (done_) return nullptr;
    cv_.wait(lk, [&] { return ready_.count(next_out_) || !err_.empty() || next_out_ >= n_tickets_; });
    auto it = ready_.find(next_out_);
    if (it == ready_.end()) { done_ = true; return nullptr; }
    std::unique_ptr<CompChunk> c = std::move(it->second);
    ready_.erase(it);
    ++next_out_;
    if (c->last) done_ = true;
    return c;
  }
  // a slab the caller is done with: its buffer serves a later slab
  void recycle(std::unique_ptr<CompChunk> c) {
    {
      std::lock_guard<std::mutex> lk(m_);
      c->blocks.clear(); c->crc.clear(); c->inflated = 0; c->n_bytes = 0; c->last = false;
      free_.push_back(std::move(c));
      ++recycled_;
    }
    cv_.notify_all();
  }

 private:
  bool alloc(CompChunk& c) {
    const size_t bytes = slab_ + kOverlap + 4096;
    void* q = nullptr;
    if (hooks_.host_alloc && hooks_.host_alloc((int64_t)bytes, &q) == 0 && q) { c.data = (uint8_t*)q; c.pinned = true; }
    else { c.data = (uint8_t*)malloc(bytes); c.pinned = false; }
    c.cap = c.data ? bytes : 0;
    return c.data != nullptr;
  }
  void release(CompChunk& c) {
    if (!c.data) return;
    if (c.pinned && hooks_.host_free) hooks_.host_free(c.data); else free(c.data);
    c.data = nullptr;
  }
  void fail(const std::string& e) {
    { std::lock_guard<std::mutex> lk(m_); if (err_.empty()) err_ = e; }
    cv_.notify_all();
  }

  void loader() {
    for (;;) {
      size_t t;
      std::unique_ptr<CompChunk> c;
      {
        std::unique_lock<std::mutex> lk(m_);
        t = next_ticket_;
        if (stop_ || !err_.empty() || t >= n_tickets_) return;
        ++next_ticket_;
        // buffers are handed out in ticket order: slab t may only be read once at most pool_chunks slabs are out
        cv_.wait(lk, [&] { return stop_ || !err_.empty() || t < recycled_ + pool_chunks_; });
        if (stop_ || !err_.empty()) return;
        if (!free_.empty()) { c = std::move(free_.back()); free_.pop_back(); }
      }
      if (!c) { c.reset(new CompChunk); if (!alloc(*c)) { fail("out of memory while loading a BAM chunk"); return; } }
      const size_t base = begin_ + t * slab_;
      const size_t want = std::min(slab_ + kOverlap, fsize_ - base);
      size_t got = 0;
      while (got < want) {
        const ssize_t k = pread(fileno(f_), c->data + got, want - got, (off_t)(base + got));
        if (k <= 0) break;
        got += (size_t)k;
      }
      c->n_bytes = got;
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return stop_ || !err_.empty() || locate_turn_ == t; });
      if (stop_ || !err_.empty()) return;
      std::string e;
      locate(*c, base, e);     // (short: a few hundred 18-byte headers; under the lock, it is the ordered step)
      ++locate_turn_;
      if (!e.empty()) { if (err_.empty()) err_ = e; lk.unlock(); cv_.notify_all(); return; }
      if (next_off_ >= size_ || t + 1 == n_tickets_) {
        if (next_off_ < size_) { if (err_.empty()) err_ = "truncated BGZF block"; lk.unlock(); cv_.notify_all(); return; }
        c->last = true;
        n_tickets_ = t + 1;
      }
      ready_[t] = std::move(c);
      lk.unlock();
      cv_.notify_all();
    }
  }

  // the members that start in [next_off_, base + slab) -- next_off_ is where the previous slab's last member ended
  void locate(CompChunk& c, size_t base, std::string& err) {
    if (next_off_ >= size_) return;
    if (next_off_ < base || next_off_ > base + slab_ + kOverlap) { err = "BGZF block chain lost"; return; }
    const uint8_t* src = c.data;
    const size_t avail = c.n_bytes, scan_end = std::min(std::min(slab_, c.n_bytes), size_ - base);   // (a region ends at size_)
    size_t pos = next_off_ - base;
    if (pos >= scan_end && base + c.n_bytes < size_ && c.n_bytes < slab_ + kOverlap) { err = "short read"; return; }
    while (pos < scan_end) {
      if (pos + 18 > avail) { if (base + avail >= fsize_) err = "truncated BGZF block"; break; }
      const uint8_t* h = src + pos;
      if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "bad BGZF block"; return; }
      uint16_t xlen;
      memcpy(&xlen, h + 10, 2);
      if (pos + 12 + xlen > avail) { if (base + avail >= fsize_) err = "truncated BGZF block"; break; }
      int bsize = -1;
      for (size_t o = 0; o + 4 <= xlen;) {
        const uint8_t* x = h + 12 + o;
        uint16_t slen;
        memcpy(&slen, x + 2, 2);
        if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) { uint16_t v; memcpy(&v, x + 4, 2); bsize = v; break; }
        o += 4u + slen;
      }
      if (bsize < 0 || (size_t)bsize + 1 < 12u + xlen + 8u) { err = "BGZF block without BC field"; return; }
      if (pos + (size_t)bsize + 1 > avail) { if (base + avail >= fsize_) err = "truncated BGZF block"; break; }
      if (base + pos + (size_t)bsize + 1 > size_ && size_ < fsize_) { err = "BGZF block chain lost"; return; }   // (a region's end is a member's start)
      const size_t cdata = (size_t)bsize + 1 - 12 - xlen - 8;
      svdss_bgzf_block_t b;
      b.coff = (int64_t)(pos + 12 + xlen); b.clen = (int32_t)cdata; b.uoff = 0;
      uint32_t crc, isize;
      memcpy(&crc, src + b.coff + cdata, 4);
      memcpy(&isize, src + b.coff + cdata + 4, 4);
      if (isize > 65536u) { err = "bad BGZF block"; return; }
      b.isize = (int32_t)isize;
      c.blocks.push_back(b);
      c.crc.push_back(crc);
      c.inflated += isize;
      pos += (size_t)bsize + 1;
    }
    next_off_ = base + pos;
  }

  Hooks hooks_;
  FILE* f_ = nullptr;
  size_t size_ = 0, fsize_ = 0, begin_ = 0, slab_, pool_chunks_;   // the scanner's range is [begin_, size_) of fsize_ bytes
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  size_t n_tickets_ = 0, next_ticket_ = 0, locate_turn_ = 0, next_out_ = 0, recycled_ = 0;
  size_t next_off_ = 0;     // file offset of the next member to locate
  std::map<size_t, std::unique_ptr<CompChunk>> ready_;
  std::vector<std::unique_ptr<CompChunk>> free_;
  bool stop_ = false, done_ = false;
  std::string err_;
};
