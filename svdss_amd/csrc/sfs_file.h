// sfs_file.h -- the `.sfs` text `SVDSS search` writes and `SVDSS call` reads (parse_sfsfile, /root/reference/sfs.cpp:5-30):
// lines "<read name or *>\t<start>\t<length>\t<haplotype tag>...", a '*' continuing the read of the line before.
//
// A 30x human sample is ~650 million such lines (65 per read); read with fgets + sscanf + a map lookup per line that is
// minutes on one core -- more than everything else `call` does.  Here the file is mapped, cut at line ends into one
// piece per thread, parsed by the threads into flat arrays, and the per-read lists are put into the map in file order
// afterwards (one lookup per READ).  Same result as the line-by-line reader (sfs_parse_lines, kept for files with
// lines the piecewise reader does not take on: 8,191 characters or more, which fgets would split) -- including what it
// does with malformed lines (skipped unless four fields parse), repeated read names (the later list replaces the
// earlier one) and '*' lines before any name (they belong to the read named "").
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

struct RawSFS { int qs, l, htag; };
using SfsMap = std::unordered_map<std::string, std::vector<RawSFS>>;

// the line-by-line reader (the reference's loop: `>> name >> qs >> l >> htag` per line)
inline bool sfs_parse_lines(const char* path, SfsMap& out) {
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char nm[4096]; int qs, l, ht; std::string cur;
  char line[8192];
  while (fgets(line, sizeof line, f)) {
    if (sscanf(line, "%4095s %d %d %d", nm, &qs, &l, &ht) != 4) continue;
    if (strcmp(nm, "*") != 0) { cur = nm; out[cur] = std::vector<RawSFS>(); }
    out[cur].push_back(RawSFS{qs, l, ht});
  }
  fclose(f);
  return true;
}

namespace sfs_file_detail {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// %d of sscanf: optional blanks, optional sign, digits; false if no digit is there
inline bool scan_int(const char*& p, const char* e, int& v, bool& overflow) {
  while (p < e && is_space(*p)) ++p;
  bool neg = false;
  if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
  if (p >= e || *p < '0' || *p > '9') return false;
  unsigned long long x = 0;
  while (p < e && *p >= '0' && *p <= '9') {
    x = x * 10u + (unsigned)(*p - '0');
    if (x > 0x80000000ull) overflow = true;   // (beyond int: what sscanf's %d does with it is the line-by-line reader's to say)
    ++p;
  }
  v = neg ? (int)(0u - (unsigned)x) : (int)x;
  if (x > (neg ? 0x80000000ull : 0x7fffffffull)) overflow = true;
  return true;
}
inline bool scan_int(const char*& p, const char* e, int& v) { bool o = false; return scan_int(p, e, v, o); }

struct Piece {
  std::vector<RawSFS> recs;                 // every record of the piece, in file order
  struct Group { const char* name; uint32_t name_len; size_t first, count; };
  std::vector<Group> groups;                // the named lines and how many records follow each
  size_t lead = 0;                          // records before the first named line ('*' lines of the piece before)
  bool long_line = false;
};

inline void parse_piece(const char* b, const char* e, Piece& P) {
  const char* p = b;
  while (p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    const char* le = nl ? nl : e;
    if (le - p >= 8191) { P.long_line = true; return; }
    const char* q = p;
    while (q < le && is_space(*q)) ++q;
    const char* n0 = q;
    while (q < le && !is_space(*q)) ++q;
    const size_t nlen = (size_t)(q - n0);
    int qs, l, ht;
    bool ovf = false;
    if (nlen > 4095) { P.long_line = true; return; }    // (sscanf's %4095s would cut the name there: the line-by-line reader's case)
    if (nlen >= 1 && scan_int(q, le, qs, ovf) && scan_int(q, le, l, ovf) && scan_int(q, le, ht, ovf)) {
      if (ovf) { P.long_line = true; return; }
      if (!(nlen == 1 && *n0 == '*')) P.groups.push_back(Piece::Group{n0, (uint32_t)nlen, P.recs.size(), 0});
      if (P.groups.empty()) ++P.lead; else ++P.groups.back().count;
      P.recs.push_back(RawSFS{qs, l, ht});
    }
    p = nl ? nl + 1 : e;
  }
}

}  // namespace sfs_file_detail

// the whole file into `out` with `threads` threads; false if it cannot be opened
inline bool sfs_parse_file(const char* path, int threads, SfsMap& out) {
  using namespace sfs_file_detail;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return sfs_parse_lines(path, out); }
  const size_t size = (size_t)st.st_size;
  if (size == 0) { close(fd); return true; }
  void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return sfs_parse_lines(path, out);
  const char* base = (const char*)m;
  if (memchr(base, '\0', size)) { munmap(m, size); return sfs_parse_lines(path, out); }   // (fgets / sscanf stop at a NUL)
  const size_t T = (size_t)(threads < 1 ? 1 : threads);
  const size_t n_pieces = std::max<size_t>(1, std::min<size_t>(T, size / (1 << 20) + 1));
  std::vector<const char*> cut(n_pieces + 1, base + size);
  cut[0] = base;
  for (size_t k = 1; k < n_pieces; ++k) {
    const char* p = base + size / n_pieces * k;
    const char* nl = (const char*)memchr(p, '\n', (size_t)(base + size - p));
    cut[k] = nl ? nl + 1 : base + size;
    if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
  }
  std::vector<Piece> pieces(n_pieces);
  {
    std::vector<std::thread> pool;
    for (size_t k = 1; k < n_pieces; ++k) pool.emplace_back([&, k] { parse_piece(cut[k], cut[k + 1], pieces[k]); });
    parse_piece(cut[0], cut[1], pieces[0]);
    for (std::thread& t : pool) t.join();
  }
  bool long_line = false;
  size_t n_groups = 0;
  for (const Piece& P : pieces) { long_line = long_line || P.long_line; n_groups += P.groups.size(); }
  if (long_line) { munmap(m, size); out.clear(); return sfs_parse_lines(path, out); }
  out.reserve(out.size() + n_groups + 1);
  std::vector<RawSFS>* cur = nullptr;       // the list of the read the last named line named
  for (const Piece& P : pieces) {
    if (P.lead) {
      if (!cur) cur = &out[std::string()];
      cur->insert(cur->end(), P.recs.begin(), P.recs.begin() + (long)P.lead);
    }
    for (const Piece::Group& g : P.groups) {
      cur = &out[std::string(g.name, g.name_len)];
      cur->assign(P.recs.begin() + (long)g.first, P.recs.begin() + (long)(g.first + g.count));
    }
  }
  munmap(m, size);
  return true;
}
