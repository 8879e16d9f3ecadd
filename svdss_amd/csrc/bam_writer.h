// bam_writer.h -- minimal BGZF/BAM writer (zlib only) for `SVDSS smooth`, which prints a BAM to
// stdout (/root/reference/smoother.cpp:441, sam_write1).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

class BgzfWriter {
 public:
  explicit BgzfWriter(FILE* f) : f_(f) { buf_.reserve(BLOCK); }
  void write(const void* p, size_t n) {
    const uint8_t* s = (const uint8_t*)p;
    while (n) {
      const size_t take = std::min(n, BLOCK - buf_.size());
      buf_.insert(buf_.end(), s, s + take);
      s += take;
      n -= take;
      if (buf_.size() == BLOCK) flush_block();
    }
  }
  bool finish() {   // flush + the 28-byte EOF marker block
    if (!buf_.empty()) flush_block();
    flush_block();
    return fflush(f_) == 0 && ok_;
  }

 private:
  static constexpr size_t BLOCK = 0xff00;
  void flush_block() {
    uint8_t out[0x10000 + 64];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = buf_.data();
    zs.avail_in = (uInt)buf_.size();
    zs.next_out = out + 18;
    zs.avail_out = sizeof out - 18 - 8;
    deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
    memcpy(out, hdr, 12);
    out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
    const uint16_t bsize = (uint16_t)(clen + 25);
    memcpy(out + 16, &bsize, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf_.data(), (uInt)buf_.size());
    const uint32_t isize = (uint32_t)buf_.size();
    memcpy(out + 18 + clen, &crc, 4);
    memcpy(out + 18 + clen + 4, &isize, 4);
    if (fwrite(out, 1, clen + 26, f_) != clen + 26) ok_ = false;
    buf_.clear();
  }
  FILE* f_;
  std::vector<uint8_t> buf_;
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
