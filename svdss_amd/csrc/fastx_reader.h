// fastx_reader.h -- FASTA/FASTQ (optionally gzip) record reader for the host CLI.
// Plays the role of the vendored klib kseq reader the reference uses
// (/root/reference/fastq.hpp:17-35, chromosomes.cpp:9-27): name = first word of
// the header line, multi-line sequences concatenated, '+' section of FASTQ skipped.
//
// The file is read in 4 MB pieces and lines are found with memchr: a reference FASTA is 50 million lines of 60 bases,
// and a gzgets + two string appends per line cost more than everything `SVDSS call` does on the GPU.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <string>
#include <vector>

class FastxReader {
 public:
  explicit FastxReader(const std::string& path) : f_(gzopen(path.c_str(), "rb")), buf_((size_t)4 << 20) {
    if (f_) gzbuffer(f_, 1 << 20);
  }
  ~FastxReader() { if (f_) gzclose(f_); }
  FastxReader(const FastxReader&) = delete;
  FastxReader& operator=(const FastxReader&) = delete;
  bool ok() const { return f_ != nullptr; }

  // next record; returns false at end of file
  bool next(std::string& name, std::string& seq) {
    name.clear();
    seq.clear();
    if (!f_) return false;
    // the header line: skip what is not one (blank lines, stray text)
    for (;;) {
      const int c = peek();
      if (c < 0) return false;
      if (c == '>' || c == '@') break;
      skip_line();
    }
    const bool fastq = peek() == '@';
    line_.clear();
    append_line(line_);
    size_t e = 1;
    while (e < line_.size() && line_[e] != ' ' && line_[e] != '\t') ++e;
    name.assign(line_, 1, e - 1);
    // sequence lines up to the next header (FASTA: '>' or '@'; FASTQ: the '+' line)
    for (;;) {
      const int c = peek();
      if (c < 0) return true;
      if (c == '>' || (!fastq && c == '@') || (fastq && c == '+')) break;
      append_line(seq);
    }
    if (fastq && peek() == '+') {
      // quality: as many characters as the sequence (may span lines)
      skip_line();
      size_t got = 0;
      while (got < seq.size() && peek() >= 0) got += skip_line();
    }
    return true;
  }

 private:
  // first character of the next line (an empty line gives '\n'), -1 at the end of the file
  int peek() {
    if (pos_ == end_ && !fill()) return -1;
    return (unsigned char)buf_[pos_];
  }
  bool fill() {
    if (eof_) return false;
    pos_ = 0;
    const int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
    end_ = n > 0 ? (size_t)n : 0;
    if (end_ == 0) eof_ = true;
    return end_ > 0;
  }
  // the rest of the current line, without its end-of-line characters, appended to dst; returns the number of
  // characters appended.  A line may span any number of buffer fills (a chromosome on one line).
  size_t append_line(std::string& dst) {
    const size_t start = dst.size();
    for (;;) {
      if (pos_ == end_ && !fill()) break;
      const char* p = buf_.data() + pos_;
      const char* nl = (const char*)memchr(p, '\n', end_ - pos_);
      const size_t len = nl ? (size_t)(nl - p) : end_ - pos_;
      dst.append(p, len);
      pos_ += len + (nl ? 1 : 0);
      if (nl) break;
    }
    while (dst.size() > start && (dst.back() == '\r' || dst.back() == '\n')) dst.pop_back();
    return dst.size() - start;
  }
  size_t skip_line() {   // (its length matters for FASTQ quality lines)
    scratch_.clear();
    return append_line(scratch_);
  }
  gzFile f_;
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
  std::string line_, scratch_;
};

// A plain (not compressed) FASTA file with '\n' line ends, read by `threads` threads from a mapping: the records found by
// a parallel scan for lines that begin with '>', every record's lines joined (and upper-cased when `upper`: what
// load_chromosomes does, /root/reference/chromosomes.cpp:19) by threads that each take a stretch of the file.  The result
// is what FastxReader::next gives record by record; anything this reader does not mean to handle -- a gzip header, a
// carriage return anywhere, a line that begins with '@' (FASTQ, or FASTA with such a header), no record at all -- makes it
// return false with the outputs untouched, and the caller reads the file with FastxReader.  (GRCh38: 3.1 GB in ~0.3 s
// instead of 1.9 s, which pass 1 of `SVDSS call` spent waiting.)
inline bool load_fasta_mapped(const std::string& path, int threads, bool upper, std::vector<std::string>& names, std::vector<std::string>& seqs) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) { close(fd); return false; }
  const size_t n = (size_t)st.st_size;
  void* mp = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (mp == MAP_FAILED) return false;
  const char* d = (const char*)mp;
  bool ok = !((unsigned char)d[0] == 0x1f && (unsigned char)d[1] == 0x8b);
  const int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
  // ---- lines that begin with '>' (and what rules the fast path out), by stretches of the file
  std::vector<std::vector<size_t>> found((size_t)T);
  std::vector<char> bad((size_t)T, 0);
  if (ok) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
        if (memchr(d + lo, '\r', hi - lo)) { bad[(size_t)t] = 1; return; }
        // a line starts at 0 and behind every '\n'; this thread owns the line starts in [lo, hi)
        size_t p = lo;
        if (t == 0) { if (d[0] == '>') found[(size_t)t].push_back(0); else if (d[0] == '@') bad[(size_t)t] = 1; }
        while (p < hi) {
          const char* nl = (const char*)memchr(d + p, '\n', hi - p);
          if (!nl) break;
          p = (size_t)(nl - d) + 1;
          if (p < n) {
            if (d[p] == '>') found[(size_t)t].push_back(p);
            else if (d[p] == '@') { bad[(size_t)t] = 1; return; }
          }
        }
      });
    for (std::thread& x : th) x.join();
    for (char b : bad) if (b) ok = false;
  }
  std::vector<size_t> heads;
  if (ok) for (const std::vector<size_t>& f : found) heads.insert(heads.end(), f.begin(), f.end());
  if (!ok || heads.empty()) { munmap(mp, n); return false; }
  // ---- the records: name = the header line up to the first blank; sequence = the lines up to the next header
  const size_t R = heads.size();
  std::vector<std::string> nm(R), sq(R);
  struct Piece { size_t rec, lo, hi, out; };
  std::vector<Piece> pieces;
  std::vector<size_t> body_lo(R), body_hi(R);
  for (size_t r = 0; r < R; ++r) {
    const size_t h = heads[r], end = r + 1 < R ? heads[r + 1] : n;
    const char* nl = (const char*)memchr(d + h, '\n', end - h);
    const size_t eol = nl ? (size_t)(nl - d) : end;
    size_t e = h + 1;
    while (e < eol && d[e] != ' ' && d[e] != '\t') ++e;
    nm[r].assign(d + h + 1, e - (h + 1));
    body_lo[r] = nl ? eol + 1 : end;
    body_hi[r] = end;
    for (size_t p = body_lo[r]; p < end; p += (size_t)8 << 20) pieces.push_back(Piece{r, p, std::min(end, p + ((size_t)8 << 20)), 0});
  }
  // characters a piece contributes = its bytes that are not '\n'
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)T) {
          Piece& pc = pieces[i];
          size_t nls = 0;
          for (const char* q = d + pc.lo; q < d + pc.hi;) {
            const char* f = (const char*)memchr(q, '\n', (size_t)(d + pc.hi - q));
            if (!f) break;
            ++nls;
            q = f + 1;
          }
          pc.out = (pc.hi - pc.lo) - nls;
        }
      });
    for (std::thread& x : th) x.join();
  }
  {
    std::vector<size_t> total(R, 0);
    for (Piece& pc : pieces) { const size_t o = total[pc.rec]; total[pc.rec] += pc.out; pc.out = o; }
    // (resize zero-fills and takes the page faults of 3.1 GB: by one thread that was half of the whole load; the records
    // are dealt to the threads, largest first)
    std::vector<size_t> order(R);
    for (size_t r = 0; r < R; ++r) order[r] = r;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return total[a] > total[b]; });
    std::atomic<size_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&] { for (size_t k = next.fetch_add(1); k < R; k = next.fetch_add(1)) sq[order[k]].resize(total[order[k]]); });
    for (std::thread& x : th) x.join();
  }
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)T) {
          const Piece& pc = pieces[i];
          char* w = &sq[pc.rec][0] + pc.out;
          for (const char* q = d + pc.lo; q < d + pc.hi;) {
            const char* f = (const char*)memchr(q, '\n', (size_t)(d + pc.hi - q));
            const size_t len = f ? (size_t)(f - q) : (size_t)(d + pc.hi - q);
            if (upper) for (size_t k = 0; k < len; ++k) { const char ch = q[k]; w[k] = (char)(ch - ((ch >= 'a' && ch <= 'z') ? 32 : 0)); }
            else memcpy(w, q, len);
            w += len;
            q += len + 1;
          }
        }
      });
    for (std::thread& x : th) x.join();
  }
  munmap(mp, n);
  names.swap(nm);
  seqs.swap(sq);
  return true;
}
