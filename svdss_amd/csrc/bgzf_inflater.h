// bgzf_inflater.h -- the host-side inflate of BGZF blocks for the CLI's readers (csrc/bam_reader.h, and the harnesses
// under tests/): one block through libdeflate or zlib (BgzfInflater), a pool of worker threads that inflates the blocks of
// a chunk in parallel (InflatePool), and the number of processors this process may really use (effective_cpus).
// Stands where htslib's bgzf_read_block / bgzf thread pool stand under sam_read1 (/root/reference/ping_pong.cpp:58,
// 247-249); split from bam_reader.h in round 4.
#pragma once
#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

// the processors this process may actually use: the hardware threads, capped by the cgroup's CPU quota (a container
// with 256 visible threads and a quota of 16 runs 16 inflate workers at full speed and 128 of them at an eighth)
inline unsigned effective_cpus() {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long long period = 0;
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long long quota = atoll(q);
      if (quota > 0) hw = (unsigned)std::max<long long>(1, std::min<long long>(hw, (quota + period - 1) / period));
    }
    fclose(f);
  }
  return hw;
}

// raw-deflate decoder of one BGZF block: libdeflate when the shared library is on the machine (no header needed: its C
// API is four functions; 2-3 x zlib's speed, and inflate is what bounds `search` end to end), zlib otherwise
struct BgzfInflater {
  typedef void* (*alloc_fn)(void);
  typedef int (*decomp_fn)(void*, const void*, size_t, void*, size_t, size_t*);
  typedef void (*free_fn)(void*);
  typedef uint32_t (*crc_fn)(uint32_t, const void*, size_t);
  struct Lib {
    alloc_fn alloc = nullptr; decomp_fn decomp = nullptr; free_fn free_ = nullptr; crc_fn crc = nullptr;
    Lib() {
      if (getenv("SVDSS_NO_LIBDEFLATE")) return;
      void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
      if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
      if (!h) return;
      alloc = (alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
      decomp = (decomp_fn)dlsym(h, "libdeflate_deflate_decompress");
      free_ = (free_fn)dlsym(h, "libdeflate_free_decompressor");
      crc = (crc_fn)dlsym(h, "libdeflate_crc32");
      if (!alloc || !decomp || !free_ || !crc) alloc = nullptr;
    }
  };
  static const Lib& lib() { static Lib l; return l; }
  void* d = nullptr;
  BgzfInflater() { if (lib().alloc) d = lib().alloc(); }
  ~BgzfInflater() { if (d) lib().free_(d); }
  BgzfInflater(const BgzfInflater&) = delete;
  BgzfInflater& operator=(const BgzfInflater&) = delete;
  // nullptr = ok, else what went wrong
  const char* run(const uint8_t* in, size_t clen, uint8_t* out, uint32_t isize, uint32_t crc) {
    if (d) {
      size_t got = 0;
      if (lib().decomp(d, in, clen, out, isize, &got) != 0 || got != isize) return "BGZF inflate failed";
      if (lib().crc(0, out, isize) != crc) return "BGZF block CRC mismatch";
      return nullptr;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return "zlib init failed";
    zs.next_in = const_cast<uint8_t*>(in);
    zs.avail_in = (uInt)clen;
    zs.next_out = out;
    zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) return "BGZF inflate failed";
    if ((uint32_t)crc32(0L, out, isize) != crc) return "BGZF block CRC mismatch";
    return nullptr;
  }
};

// persistent worker threads for the inflate of BGZF blocks (the chunks in flight share them; a thread per chunk per
// block range used to be spawned and joined for every 32 MB of input)
class InflatePool {
 public:
  explicit InflatePool(int n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
  }
  ~InflatePool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
  }
  // fn(k, inflater) for k in [0, n): spread over the workers and the caller; returns when all are done
  void run(size_t n, const std::function<void(size_t, BgzfInflater&)>& fn) {
    if (n == 0) return;
    auto job = std::make_shared<Job>();
    job->n = n;
    job->fn = &fn;
    {
      std::lock_guard<std::mutex> lk(m_);
      jobs_.push_back(job);
    }
    cv_.notify_all();
    BgzfInflater mine;
    work_on(*job, mine);
    std::unique_lock<std::mutex> lk(job->dm);
    job->dcv.wait(lk, [&] { return job->done == job->n; });
  }

 private:
  struct Job {
    size_t n = 0;
    std::atomic<size_t> next{0};
    size_t done = 0;
    const std::function<void(size_t, BgzfInflater&)>* fn = nullptr;
    std::mutex dm;
    std::condition_variable dcv;
  };
  void work_on(Job& j, BgzfInflater& inf) {
    size_t did = 0;
    for (;;) {
      const size_t k = j.next.fetch_add(1);
      if (k >= j.n) break;
      (*j.fn)(k, inf);
      ++did;
    }
    if (did) {
      std::lock_guard<std::mutex> lk(j.dm);
      j.done += did;
      if (j.done == j.n) j.dcv.notify_all();
    }
  }
  void loop() {
    BgzfInflater inf;
    for (;;) {
      std::shared_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] {
          while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->n) jobs_.pop_front();
          return stop_ || !jobs_.empty();
        });
        if (stop_) return;
        j = jobs_.front();
      }
      work_on(*j, inf);
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Job>> jobs_;
  bool stop_ = false;
};
