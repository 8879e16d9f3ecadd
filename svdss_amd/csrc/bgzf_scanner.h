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

  BgzfScanner(const std::string& path, Hooks hooks, size_t slab, int loaders, size_t pool_chunks)
      : hooks_(hooks), slab_(slab), pool_chunks_(pool_chunks < 2 ? 2 : pool_chunks) {
    f_ = fopen(path.c_str(), "rb");
    struct stat st;
    if (f_ && fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode)) size_ = (size_t)st.st_size;
    else if (f_) { fclose(f_); f_ = nullptr; }
    n_tickets_ = size_ ? (size_ + slab_ - 1) / slab_ : 0;
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
  size_t file_size() const { return size_; }
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
      const size_t base = t * slab_;
      const size_t want = std::min(slab_ + kOverlap, size_ - base);
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
    const size_t avail = c.n_bytes, scan_end = std::min(slab_, c.n_bytes);
    size_t pos = next_off_ - base;
    if (pos >= scan_end && base + c.n_bytes < size_ && c.n_bytes < slab_ + kOverlap) { err = "short read"; return; }
    while (pos < scan_end) {
      if (pos + 18 > avail) { if (base + avail >= size_) err = "truncated BGZF block"; break; }
      const uint8_t* h = src + pos;
      if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "bad BGZF block"; return; }
      uint16_t xlen;
      memcpy(&xlen, h + 10, 2);
      if (pos + 12 + xlen > avail) { if (base + avail >= size_) err = "truncated BGZF block"; break; }
      int bsize = -1;
      for (size_t o = 0; o + 4 <= xlen;) {
        const uint8_t* x = h + 12 + o;
        uint16_t slen;
        memcpy(&slen, x + 2, 2);
        if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) { uint16_t v; memcpy(&v, x + 4, 2); bsize = v; break; }
        o += 4u + slen;
      }
      if (bsize < 0 || (size_t)bsize + 1 < 12u + xlen + 8u) { err = "BGZF block without BC field"; return; }
      if (pos + (size_t)bsize + 1 > avail) { if (base + avail >= size_) err = "truncated BGZF block"; break; }
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
  size_t size_ = 0, slab_, pool_chunks_;
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
