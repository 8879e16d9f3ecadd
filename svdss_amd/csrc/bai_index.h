// bai_index.h -- BAI / CSI index and region-restricted reading of a BAM file.
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
    std::vector<uint64_t> linear;                                                          // BAI: 16 kb windows
    std::vector<std::pair<uint32_t, uint64_t>> loff;                                       // CSI: per bin, sorted like bins
  };
  std::vector<Ref> refs;
  // the binning scheme: BAI is fixed at 14 / 5; a CSI index names its own (SAM specification 5.3, CSIv1: for references
  // beyond 2^29 bases, or a finer / coarser smallest bin)
  int min_shift = 14, depth = 5;
  bool csi = false;

  // .bai as written (BAI\1 ...) or .csi (BGZF-compressed, CSI\1 ...): what htslib's sam_index_load accepts for a BAM
  bool load(const std::string& path) {
    refs.clear();
    std::vector<uint8_t> raw;
    {
      FILE* f = fopen(path.c_str(), "rb");
      if (!f) return false;
      uint8_t buf[1 << 16];
      size_t k;
      while ((k = fread(buf, 1, sizeof buf, f)) > 0) {
        if (raw.size() + k > ((size_t)4 << 30)) { fclose(f); return false; }   // (an index, not a data file)
        raw.insert(raw.end(), buf, buf + k);
      }
      fclose(f);
    }
    bool ok = false;
    if (raw.size() >= 4 && memcmp(raw.data(), "BAI\1", 4) == 0) ok = parse(raw.data(), raw.size(), false);
    else if (raw.size() >= 18 && raw[0] == 31 && raw[1] == 139) {
      std::vector<uint8_t> plain;
      if (inflate_all(raw, plain) && plain.size() >= 4 && memcmp(plain.data(), "CSI\1", 4) == 0) ok = parse(plain.data(), plain.size(), true);
    }
    if (!ok) refs.clear();
    return ok;
  }

  // chunks that may hold records overlapping [beg, end) on reference tid (0-based, half open), appended to out
  void query(int tid, int64_t beg, int64_t end, std::vector<std::pair<uint64_t, uint64_t>>& out) const {
    if (tid < 0 || tid >= (int)refs.size() || end <= beg) return;
    if (beg < 0) beg = 0;
    const int64_t span = (int64_t)1 << (min_shift + 3 * depth);   // positions the scheme can bin
    if (beg >= span) return;
    const Ref& r = refs[(size_t)tid];
    const int64_t e = std::min(end, span) - 1;
    uint64_t min_off = 0;
    if (!csi) {
      if (!r.linear.empty()) {
        const size_t w = (size_t)(beg >> 14);
        min_off = w < r.linear.size() ? r.linear[w] : r.linear.back();
      }
    } else {
      // the smallest bin that holds beg and is in the index says where the records that reach it begin; without one
      // its parents do (their value is the smaller)
      uint32_t bin = (uint32_t)(first_bin(depth) + (beg >> min_shift));
      for (;;) {
        auto it = std::lower_bound(r.loff.begin(), r.loff.end(), bin, [](const auto& a, uint32_t b) { return a.first < b; });
        if (it != r.loff.end() && it->first == bin) { min_off = it->second; break; }
        if (bin == 0) break;
        bin = (bin - 1) >> 3;
      }
    }
    // per level, the bins PRESENT in [first_bin(l) + lo, first_bin(l) + hi] -- not every possible bin number (a .csi with
    // a small min_shift makes that ~1e8 lookups for a chromosome-wide query: ADVICE r5)
    for (int l = 0; l <= depth; ++l) {
      const int sh = min_shift + 3 * (depth - l);
      const int64_t lo = first_bin(l) + (beg >> sh), hi = first_bin(l) + (e >> sh);
      auto it = std::lower_bound(r.bins.begin(), r.bins.end(), lo, [](const auto& a, int64_t b) { return (int64_t)a.first < b; });
      for (; it != r.bins.end() && (int64_t)it->first <= hi; ++it)
        for (const auto& c : it->second)
          if (c.second > min_off) out.emplace_back(std::max(c.first, min_off), c.second);
    }
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
  static int64_t first_bin(int level) { return (((int64_t)1 << (3 * level)) - 1) / 7; }

  static bool inflate_all(const std::vector<uint8_t>& raw, std::vector<uint8_t>& out) {
    BgzfInflater inf;
    size_t pos = 0;
    while (pos < raw.size()) {
      if (pos + 18 > raw.size()) return false;
      const uint8_t* h = raw.data() + pos;
      uint16_t xlen, bsize;
      memcpy(&xlen, h + 10, 2);
      if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4) || xlen != 6 || h[12] != 'B' || h[13] != 'C') return false;
      memcpy(&bsize, h + 16, 2);
      if ((size_t)bsize + 1 < 18u + 8u || pos + (size_t)bsize + 1 > raw.size()) return false;
      const size_t clen = (size_t)bsize + 1 - 18 - 8;
      uint32_t crc, isize;
      memcpy(&crc, h + 18 + clen, 4);
      memcpy(&isize, h + 18 + clen + 4, 4);
      if (isize > 65536u || out.size() + isize > ((size_t)4 << 30)) return false;
      const size_t at = out.size();
      out.resize(at + isize);
      if (isize && inf.run(h + 18, clen, out.data() + at, isize, crc)) return false;
      pos += (size_t)bsize + 1;
    }
    return true;
  }

  // a count in the file is believed only as far as the file has bytes for what it counts: a damaged index is
  // refused, not answered with an allocation of gigabytes
  struct Cur {
    const uint8_t* p;
    size_t left;
    bool get(void* dst, size_t n) {
      if (n > left) return false;
      memcpy(dst, p, n);
      p += n; left -= n;
      return true;
    }
    bool skip(size_t n) { if (n > left) return false; p += n; left -= n; return true; }
    bool fits(uint64_t count, uint64_t bytes_each) const { return count <= left / bytes_each; }
  };

  bool parse(const uint8_t* data, size_t size, bool is_csi) {
    Cur c{data + 4, size - 4};
    csi = is_csi;
    min_shift = 14; depth = 5;
    if (is_csi) {
      int32_t ms, dp, l_aux;
      if (!c.get(&ms, 4) || !c.get(&dp, 4) || !c.get(&l_aux, 4) || l_aux < 0 || !c.skip((size_t)l_aux)) return false;
      if (ms < 1 || ms > 40 || dp < 0 || dp > 10 || ms + 3 * dp > 62) return false;   // (bin numbers stay below 2^31)
      min_shift = ms; depth = dp;
    }
    int32_t n_ref;
    if (!c.get(&n_ref, 4) || n_ref < 0 || !c.fits((uint64_t)n_ref, is_csi ? 4 : 8)) return false;
    const uint32_t meta_bin = (uint32_t)(first_bin(depth + 1) + 1);   // htslib's metadata pseudo-bin (37450 for BAI)
    refs.resize((size_t)n_ref);
    for (Ref& r : refs) {
      int32_t n_bin;
      if (!c.get(&n_bin, 4) || n_bin < 0 || !c.fits((uint64_t)n_bin, is_csi ? 16 : 8)) return false;
      for (int32_t b = 0; b < n_bin; ++b) {
        uint32_t bin;
        uint64_t lo = 0;
        int32_t n_chunk;
        if (!c.get(&bin, 4) || (is_csi && !c.get(&lo, 8)) || !c.get(&n_chunk, 4) || n_chunk < 0 || !c.fits((uint64_t)n_chunk, 16)) return false;
        std::vector<std::pair<uint64_t, uint64_t>> ch((size_t)n_chunk);
        for (auto& k : ch)
          if (!c.get(&k.first, 8) || !c.get(&k.second, 8)) return false;
        if (bin == meta_bin) continue;
        r.bins.emplace_back(bin, std::move(ch));
        if (is_csi) r.loff.emplace_back(bin, lo);
      }
      std::sort(r.bins.begin(), r.bins.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      std::sort(r.loff.begin(), r.loff.end());
      if (!is_csi) {
        int32_t n_intv;
        if (!c.get(&n_intv, 4) || n_intv < 0 || !c.fits((uint64_t)n_intv, 8)) return false;
        r.linear.resize((size_t)n_intv);
        if (n_intv && !c.get(r.linear.data(), 8 * (size_t)n_intv)) return false;
      }
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
