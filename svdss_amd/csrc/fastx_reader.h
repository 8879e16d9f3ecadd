// fastx_reader.h -- FASTA/FASTQ (optionally gzip) record reader for the host CLI.
// Plays the role of the vendored klib kseq reader the reference uses
// (/root/reference/fastq.hpp:17-35, chromosomes.cpp:9-27): name = first word of
// the header line, multi-line sequences concatenated, '+' section of FASTQ skipped.
//
// The file is read in 4 MB pieces and lines are found with memchr: a reference FASTA is 50 million lines of 60 bases,
// and a gzgets + two string appends per line cost more than everything `SVDSS call` does on the GPU.
#pragma once
#include <zlib.h>

#include <cstring>
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
