// bam_writer.h -- BGZF/BAM writer (blocks deflated in parallel; libdeflate when the shared library is on the machine,
// as htslib does when built with it, zlib otherwise) for `SVDSS smooth`, which prints a BAM to stdout
// (/root/reference/smoother.cpp:441, sam_write1).
#pragma once
#include <dlfcn.h>
#include <zlib.h>

#include <cstdlib>

#include <cstdint>
#include <memory>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

// Blocks are independent deflate streams: the writer collects kBatch blocks of input and compresses them with
// `threads` workers, then writes them in order (htslib's bgzf_mt plays this role for the reference).  The
// output bytes do not depend on the number of threads.
class BgzfWriter {
 public:
  explicit BgzfWriter(FILE* f, int threads = 1) : f_(f), threads_(threads < 1 ? 1 : threads) { buf_.reserve(BLOCK * batch_); }
  ~BgzfWriter() {
    if (gpu_.obj && gpu_.free_) gpu_.free_(gpu_.obj);
    if (pin_ && gpu_.host_free) gpu_.host_free(pin_);
    if (pout_ && gpu_.host_free) gpu_.host_free(pout_);
  }
  BgzfWriter(const BgzfWriter&) = delete;
  BgzfWriter& operator=(const BgzfWriter&) = delete;
  // Blocks deflated on the GPU (csrc/deflate.hip through gpu_deflate_hook.h; the writer itself does not know HIP): a
  // batch of blocks goes up, comes back as BGZF members without their footers, and the workers add CRC32 / ISIZE.
  struct GpuDeflateApi {
    int (*deflate)(void** obj, int device, const uint8_t* in, int64_t in_bytes, int32_t block_bytes, uint8_t* out,
                   int64_t out_stride, int32_t* out_len) = nullptr;
    void (*free_)(void* obj) = nullptr;
    int (*host_alloc)(int64_t bytes, void** out) = nullptr;   // page-locked memory (the copies to and from the GPU)
    void (*host_free)(void* p) = nullptr;
    void* obj = nullptr;
    int device = 0;
  };
  void enable_gpu_deflate(const GpuDeflateApi& api) {
    gpu_ = api;
    batch_ = 1024;            // (a launch wants a few blocks per CU)
    // the collecting buffer itself is page-locked, and so is the one the members come back into (back to back): no
    // staging copies on either side of the GPU
    void* a = nullptr; void* b = nullptr;
    if (gpu_.host_alloc && gpu_.host_alloc((int64_t)(BLOCK * batch_ + 64), &a) == 0 && a &&
        gpu_.host_alloc((int64_t)((BLOCK + 128) * batch_), &b) == 0 && b) {
      pin_ = (uint8_t*)a; pin_cap_ = BLOCK * batch_;
      pout_ = (uint8_t*)b;
    } else {
      if (a && gpu_.host_free) gpu_.host_free(a);
      buf_.reserve(BLOCK * batch_);
    }
  }
  void write(const void* p, size_t n) {
    const uint8_t* s = (const uint8_t*)p;
    if (pin_) {
      while (n) {
        const size_t take = std::min(n, pin_cap_ - pin_n_);
        memcpy(pin_ + pin_n_, s, take);
        pin_n_ += take; s += take; n -= take;
        if (pin_n_ == pin_cap_) { emit(pin_, pin_n_, batch_); pin_n_ = 0; }
      }
      return;
    }
    buf_.insert(buf_.end(), s, s + n);
    if (buf_.size() >= BLOCK * batch_) flush_full_blocks();
  }
  bool finish() {   // flush + the 28-byte EOF marker block
    if (pin_) {
      if (pin_n_) emit(pin_, pin_n_, (pin_n_ + BLOCK - 1) / BLOCK);
      pin_n_ = 0;
    } else {
      flush_full_blocks();
      if (!buf_.empty()) { emit(buf_.data(), buf_.size(), 1); buf_.clear(); }
    }
    emit(nullptr, 0, 1);
    return fflush(f_) == 0 && ok_;
  }

 private:
  static constexpr size_t BLOCK = 0xff00;
  static constexpr size_t OUT = 0x10000 + 64;
  size_t batch_ = 256;
  GpuDeflateApi gpu_;
  bool gpu_warned_ = false;
  uint8_t* pin_ = nullptr;    // GPU path: the collecting buffer (page-locked), pin_n_ of pin_cap_ bytes filled
  size_t pin_n_ = 0, pin_cap_ = 0;
  uint8_t* pout_ = nullptr;   // ... and the members of a batch as they come back

  // libdeflate's compressor through dlopen (no header needed): level 6 at two to three times zlib's speed.  The bytes
  // differ from zlib's (both are valid deflate streams); whichever is used, the output does not depend on the threads.
  struct Deflater {
    typedef void* (*alloc_fn)(int);
    typedef size_t (*comp_fn)(void*, const void*, size_t, void*, size_t);
    typedef void (*free_fn)(void*);
    struct Lib {
      alloc_fn alloc = nullptr; comp_fn comp = nullptr; free_fn free_ = nullptr;
      Lib() {
        if (getenv("SVDSS_NO_LIBDEFLATE")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (alloc_fn)dlsym(h, "libdeflate_alloc_compressor");
        comp = (comp_fn)dlsym(h, "libdeflate_deflate_compress");
        free_ = (free_fn)dlsym(h, "libdeflate_free_compressor");
        if (!alloc || !comp || !free_) alloc = nullptr;
      }
    };
    static const Lib& lib() { static Lib l; return l; }
    void* c = nullptr;
    // (htslib's default for BAM is level 6; SVDSS_BAM_LEVEL = 1..9 for a pipe into a sorter, where speed matters more)
    static int level() {
      static const int l = [] { const char* e = getenv("SVDSS_BAM_LEVEL"); const int v = e ? atoi(e) : 6; return v < 1 ? 1 : v > 9 ? 9 : v; }();
      return l;
    }
    Deflater() { if (lib().alloc) c = lib().alloc(level()); }
    ~Deflater() { if (c) lib().free_(c); }
    Deflater(const Deflater&) = delete;
    Deflater& operator=(const Deflater&) = delete;
  };

  static size_t deflate_block(Deflater& d, const uint8_t* in, size_t n, uint8_t* out) {
    size_t clen = 0;
    if (d.c && n) clen = Deflater::lib().comp(d.c, in, n, out + 18, OUT - 18 - 8);
    if (clen == 0) {   // zlib (also: the empty block, and a block libdeflate could not fit)
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      deflateInit2(&zs, Deflater::level(), Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
      zs.next_in = const_cast<uint8_t*>(in);
      zs.avail_in = (uInt)n;
      zs.next_out = out + 18;
      zs.avail_out = (uInt)(OUT - 18 - 8);
      deflate(&zs, Z_FINISH);
      clen = zs.total_out;
      deflateEnd(&zs);
    }
    const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
    memcpy(out, hdr, 12);
    out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
    const uint16_t bsize = (uint16_t)(clen + 25);
    memcpy(out + 16, &bsize, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n);
    const uint32_t isize = (uint32_t)n;
    memcpy(out + 18 + clen, &crc, 4);
    memcpy(out + 18 + clen + 4, &isize, 4);
    return clen + 26;
  }

  // the batch through the GPU encoder, written out; false: not done (no GPU path, or it failed: the host compresses it)
  bool emit_gpu(const uint8_t* data, size_t bytes, size_t nblocks) {
    if (!gpu_.deflate || !data || bytes == 0) return false;
    std::vector<int32_t> l32(nblocks);
    std::vector<uint8_t> own;                      // (no page-locked buffer: an ordinary one)
    uint8_t* out = pout_;
    if (!out) { own.resize((BLOCK + 128) * nblocks); out = own.data(); }
    // out_stride 0: the members come back to back
    const int rc = gpu_.deflate(&gpu_.obj, gpu_.device, data, (int64_t)bytes, (int32_t)BLOCK, out, 0, l32.data());
    if (rc != 0) {
      if (!gpu_warned_) fprintf(stderr, "[bam_writer] GPU deflate call failed (code %d): the host deflates\n", rc);
      gpu_warned_ = true;
      return false;
    }
    std::vector<size_t> off(nblocks + 1, 0);
    for (size_t i = 0; i < nblocks; ++i) off[i + 1] = off[i] + (size_t)l32[i];
    // the footers: CRC32 of the block's bytes and their number (RFC 1952), by the workers
    const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)threads_, nblocks));
    auto work = [&](size_t t) {
      for (size_t i = t; i < nblocks; i += nt) {
        const size_t at = i * BLOCK, n = std::min(BLOCK, bytes - at);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data + at, (uInt)n), isize = (uint32_t)n;
        memcpy(out + off[i + 1] - 8, &crc, 4);
        memcpy(out + off[i + 1] - 4, &isize, 4);
      }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread& th : pool) th.join();
    if (fwrite(out, 1, off[nblocks], f_) != off[nblocks]) ok_ = false;
    return true;
  }

  // compresses `nblocks` consecutive blocks of `data` (the last one may be short) and writes them in order
  void emit(const uint8_t* data, size_t bytes, size_t nblocks) {
    if (emit_gpu(data, bytes, nblocks)) return;
    std::vector<uint8_t> out(nblocks * OUT);
    std::vector<size_t> len(nblocks);
    const size_t nt = std::min<size_t>((size_t)threads_, nblocks);
    while (deflaters_.size() < std::max<size_t>(nt, 1)) deflaters_.emplace_back(new Deflater());
    auto work = [&](size_t t, size_t nt) {
      for (size_t i = t; i < nblocks; i += nt) {
        const size_t off = i * BLOCK;
        const size_t n = bytes > off ? std::min(BLOCK, bytes - off) : 0;
        len[i] = deflate_block(*deflaters_[t], data ? data + off : nullptr, n, out.data() + i * OUT);
      }
    };
    if (nt <= 1) work(0, 1);
    else {
      std::vector<std::thread> pool;
      for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t, nt);
      work(0, nt);
      for (std::thread& th : pool) th.join();
    }
    for (size_t i = 0; i < nblocks; ++i)
      if (fwrite(out.data() + i * OUT, 1, len[i], f_) != len[i]) ok_ = false;
  }

  void flush_full_blocks() {
    const size_t nfull = buf_.size() / BLOCK;
    if (!nfull) return;
    emit(buf_.data(), nfull * BLOCK, nfull);
    buf_.erase(buf_.begin(), buf_.begin() + (ptrdiff_t)(nfull * BLOCK));
  }

  FILE* f_;
  int threads_;
  std::vector<uint8_t> buf_;
  std::vector<std::unique_ptr<Deflater>> deflaters_;   // one per worker of emit()
  bool ok_ = true;
};

inline void bam_write_header(BgzfWriter& w, const std::string& text, const std::vector<std::string>& names,
                             const std::vector<int32_t>& lens) {
  w.write("BAM\1", 4);
  const int32_t lt = (int32_t)text.size(), nr = (int32_t)names.size();
  w.write(&lt, 4);
  w.write(text.data(), text.size());
  w.write(&nr, 4);
  for (size_t i = 0; i < names.size(); ++i) {
    const int32_t ln = (int32_t)names[i].size() + 1;
    w.write(&ln, 4);
    w.write(names[i].c_str(), (size_t)ln);
    w.write(&lens[i], 4);
  }
}
