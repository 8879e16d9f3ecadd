// fastx_reader.h -- FASTA/FASTQ (optionally gzip) record reader for the host CLI.
// Plays the role of the vendored klib kseq reader the reference uses
// (/root/reference/fastq.hpp:17-35, chromosomes.cpp:9-27): name = first word of
// the header line, multi-line sequences concatenated, '+' section of FASTQ skipped.
#pragma once
#include <zlib.h>

#include <string>

class FastxReader {
 public:
  explicit FastxReader(const std::string& path) : f_(gzopen(path.c_str(), "rb")) {
    if (f_) gzbuffer(f_, 1 << 20);
  }
  ~FastxReader() { if (f_) gzclose(f_); }
  bool ok() const { return f_ != nullptr; }

  // next record; returns false at end of file
  bool next(std::string& name, std::string& seq) {
    name.clear();
    seq.clear();
    if (!f_) return false;
    if (!have_line_ && !getline()) return false;
    while (line_.empty() || (line_[0] != '>' && line_[0] != '@')) {
      if (!getline()) return false;
    }
    const bool fastq = line_[0] == '@';
    size_t e = 1;
    while (e < line_.size() && line_[e] != ' ' && line_[e] != '\t') ++e;
    name.assign(line_, 1, e - 1);
    have_line_ = false;
    while (getline()) {
      if (!line_.empty() && (line_[0] == '>' || (!fastq && line_[0] == '@') || (fastq && line_[0] == '+'))) break;
      seq += line_;
      have_line_ = false;
    }
    if (fastq && have_line_ && !line_.empty() && line_[0] == '+') {
      // quality: as many characters as the sequence (may span lines)
      size_t got = 0;
      have_line_ = false;
      while (got < seq.size() && getline()) {
        got += line_.size();
        have_line_ = false;
      }
    }
    return true;
  }

 private:
  bool getline() {
    line_.clear();
    char buf[1 << 16];
    bool any = false;
    while (gzgets(f_, buf, sizeof buf)) {
      any = true;
      line_ += buf;
      if (!line_.empty() && line_.back() == '\n') break;
    }
    if (!any) { have_line_ = false; return false; }
    while (!line_.empty() && (line_.back() == '\n' || line_.back() == '\r')) line_.pop_back();
    have_line_ = true;
    return true;
  }
  gzFile f_;
  std::string line_;
  bool have_line_ = false;
};
