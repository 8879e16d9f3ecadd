// bai_index.h -- BAI index and region-restricted reading of a BAM file.
// Stands where htslib's sam_index_load / sam_itr_querys / sam_itr_next stand in /root/reference/clusterer.cpp:495-527
// (fill_clusters queries "chrom:start-end" per cluster).  Here the regions of all clusters are turned into one sorted,
// merged list of file chunks (virtual offsets) and read once, in file order: the same records in the same relative
// order as a sequential scan that ignores the records outside the regions.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "bam_reader.h"

struct BaiIndex {
  struct Ref {
    std::vector<std::pair<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins;   // sorted by bin number
    std::vector<uint64_t> linear;                                                          // 16 kb windows
  };
  std::vector<Ref> refs;

  bool load(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = read_all(f);
    fclose(f);
    if (!ok) refs.clear();
    return ok;
  }

  // chunks that may hold records overlapping [beg, end) on reference tid (0-based, half open), appended to out
  void query(int tid, int64_t beg, int64_t end, std::vector<std::pair<uint64_t, uint64_t>>& out) const {
    if (tid < 0 || tid >= (int)refs.size() || end <= beg) return;
    if (beg < 0) beg = 0;
    const Ref& r = refs[(size_t)tid];
    uint64_t min_off = 0;
    if (!r.linear.empty()) {
      const size_t w = (size_t)(beg >> 14);
      min_off = w < r.linear.size() ? r.linear[w] : r.linear.back();
    }
    const int64_t e = end - 1;
    auto take = [&](uint32_t bin) {
      auto it = std::lower_bound(r.bins.begin(), r.bins.end(), bin, [](const auto& a, uint32_t b) { return a.first < b; });
      if (it == r.bins.end() || it->first != bin) return;
      for (const auto& c : it->second)
        if (c.second > min_off) out.emplace_back(std::max(c.first, min_off), c.second);
    };
    take(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (e >> 26); ++k) take((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (e >> 23); ++k) take((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (e >> 20); ++k) take((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (e >> 17); ++k) take((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (e >> 14); ++k) take((uint32_t)k);
  }

  // sorted, overlapping / touching chunks merged
  static void merge(std::vector<std::pair<uint64_t, uint64_t>>& v) {
    std::sort(v.begin(), v.end());
    size_t n = 0;
    for (size_t i = 0; i < v.size(); ++i) {
      if (n && v[i].first <= v[n - 1].second) v[n - 1].second = std::max(v[n - 1].second, v[i].second);
      else v[n++] = v[i];
    }
    v.resize(n);
  }

 private:
  bool read_all(FILE* f) {
    char magic[4];
    int32_t n_ref;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "BAI\1", 4) != 0 || fread(&n_ref, 4, 1, f) != 1 || n_ref < 0) return false;
    // a count in the file is believed only as far as the file has bytes for what it counts: a damaged index is
    // refused, not answered with an allocation of gigabytes
    uint64_t left = 0;
    {
      const off_t here = ftello(f);
      if (here < 0 || fseeko(f, 0, SEEK_END) != 0) return false;
      const off_t end = ftello(f);
      if (end < here || fseeko(f, here, SEEK_SET) != 0) return false;
      left = (uint64_t)(end - here);
    }
    auto fits = [&](uint64_t count, uint64_t bytes_each) { return count <= left / bytes_each; };
    if (!fits((uint64_t)n_ref, 8)) return false;
    refs.resize((size_t)n_ref);
    for (Ref& r : refs) {
      int32_t n_bin;
      if (fread(&n_bin, 4, 1, f) != 1 || n_bin < 0 || !fits((uint64_t)n_bin, 8)) return false;
      for (int32_t b = 0; b < n_bin; ++b) {
        uint32_t bin;
        int32_t n_chunk;
        if (fread(&bin, 4, 1, f) != 1 || fread(&n_chunk, 4, 1, f) != 1 || n_chunk < 0 || !fits((uint64_t)n_chunk, 16)) return false;
        std::vector<std::pair<uint64_t, uint64_t>> ch((size_t)n_chunk);
        for (auto& c : ch)
          if (fread(&c.first, 8, 1, f) != 1 || fread(&c.second, 8, 1, f) != 1) return false;
        if (bin != 37450) r.bins.emplace_back(bin, std::move(ch));   // (37450: htslib's metadata pseudo-bin)
      }
      std::sort(r.bins.begin(), r.bins.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      int32_t n_intv;
      if (fread(&n_intv, 4, 1, f) != 1 || n_intv < 0 || !fits((uint64_t)n_intv, 8)) return false;
      r.linear.resize((size_t)n_intv);
      if (n_intv && fread(r.linear.data(), 8, (size_t)n_intv, f) != (size_t)n_intv) return false;
    }
    return true;
  }
};

// Reads the records of the chunks [first, second) (virtual offsets; sorted, disjoint) of a BAM file and hands each to
// fn as a RawView, in file order.  One thread: the chunks around a few thousand clusters are a small part of the file.
// Returns an empty string, or what went wrong.
inline std::string bam_scan_chunks(const std::string& path, const std::vector<std::pair<uint64_t, uint64_t>>& chunks,
                                   const std::function<void(const BamReader::RawView&)>& fn) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "cannot open " + path;
  BgzfInflater inf;
  std::string err;
  std::vector<uint8_t> cbuf(1 << 16), ubuf;
  for (const auto& ch : chunks) {
    // inflate block after block from the chunk's first block; `ubuf` holds the inflated bytes from that block's start,
    // block_at[k] = (compressed offset, offset in ubuf) of the k-th block
    ubuf.clear();
    std::vector<std::pair<uint64_t, size_t>> block_at;
    uint64_t cpos = ch.first >> 16;
    size_t upos = (size_t)(ch.first & 0xffff);
    bool file_end = false;
    auto more = [&]() -> bool {   // appends one block; false at the end of the file or on error
      uint8_t h[18];
      if (fseeko(f, (off_t)cpos, SEEK_SET) != 0) { err = "seek failed"; return false; }
      const size_t got = fread(h, 1, 18, f);
      if (got == 0) { file_end = true; return false; }
      if (got != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "bad BGZF block"; return false; }
      uint16_t xlen, bsize;
      memcpy(&xlen, h + 10, 2);
      if (xlen != 6 || h[12] != 'B' || h[13] != 'C') { err = "BGZF block without BC field"; return false; }
      memcpy(&bsize, h + 16, 2);
      if ((size_t)bsize + 1 < 18u + 8u) { err = "bad BGZF block"; return false; }
      const size_t clen = (size_t)bsize + 1 - 18 - 8;
      if (cbuf.size() < clen + 8) cbuf.resize(clen + 8);
      if (fread(cbuf.data(), 1, clen + 8, f) != clen + 8) { err = "truncated BGZF block"; return false; }
      uint32_t crc, isize;
      memcpy(&crc, cbuf.data() + clen, 4);
      memcpy(&isize, cbuf.data() + clen + 4, 4);
      if (isize > 65536u) { err = "bad BGZF block"; return false; }   // (BGZF caps a block's inflated size)
      block_at.emplace_back(cpos, ubuf.size());
      const size_t at = ubuf.size();
      ubuf.resize(at + isize);
      if (isize) {
        if (const char* e = inf.run(cbuf.data(), clen, ubuf.data() + at, isize, crc)) { err = e; return false; }
      }
      cpos += (uint64_t)bsize + 1;
      return true;
    };
    // virtual offset of inflated position u, in its larger form: a position at the very end of the inflated data is the
    // start of the block that follows (htslib also writes "end of the previous block" for it, which compares smaller)
    auto voff_of = [&](size_t u) -> uint64_t {
      if (u >= ubuf.size()) return cpos << 16;
      size_t k = block_at.size() - 1;
      while (k > 0 && block_at[k].second > u) --k;
      return (block_at[k].first << 16) | (uint64_t)(u - block_at[k].second);
    };
    if (!more()) { if (!err.empty()) break; continue; }
    for (;;) {
      while (ubuf.size() < upos + 4 && more()) {}
      if (!err.empty()) break;
      if (ubuf.size() < upos + 4) break;   // end of the file
      if (voff_of(upos) >= ch.second) break;   // the chunk's end
      int32_t block_size;
      memcpy(&block_size, ubuf.data() + upos, 4);
      if (block_size < 32) { err = "corrupt record"; break; }
      while (ubuf.size() < upos + 4 + (size_t)block_size && more()) {}
      if (!err.empty()) break;
      if (ubuf.size() < upos + 4 + (size_t)block_size) { err = "truncated record"; break; }
      BamReader::RawView v;
      v.p = ubuf.data() + upos + 4;
      const uint8_t* core = v.p;
      uint16_t n_cigar;
      memcpy(&v.tid, core, 4);
      memcpy(&v.pos, core + 4, 4);
      v.l_name = core[8];
      v.mapq = core[9];
      memcpy(&n_cigar, core + 12, 2);
      v.n_cigar = n_cigar;
      memcpy(&v.flag, core + 14, 2);
      memcpy(&v.l_seq, core + 16, 4);
      const size_t head = 32 + (size_t)v.l_name + 4u * v.n_cigar + (v.l_seq < 0 ? (size_t)0 : ((size_t)v.l_seq + 1) / 2 + (size_t)v.l_seq);
      if (v.l_seq < 0 || head > (size_t)block_size) { err = "corrupt record"; break; }
      v.l_aux = (uint32_t)((size_t)block_size - head);
      fn(v);
      upos += 4 + (size_t)block_size;
      // a merged chunk can span whole chromosomes: drop the blocks the walk has left behind once they add up to 4 MB
      // (the block that holds upos stays, so voff_of keeps answering for every position still reachable)
      if (upos >= ((size_t)4 << 20)) {
        size_t k = 0;
        while (k + 1 < block_at.size() && block_at[k + 1].second <= upos) ++k;
        const size_t cut = std::min(block_at[k].second, upos);
        if (k > 0 && cut > 0) {
          ubuf.erase(ubuf.begin(), ubuf.begin() + (long)cut);
          block_at.erase(block_at.begin(), block_at.begin() + (long)k);
          for (auto& b : block_at) b.second -= cut;
          upos -= cut;
        }
      }
    }
    if (!err.empty()) break;
    (void)file_end;
  }
  fclose(f);
  return err;
}
