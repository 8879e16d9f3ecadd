// cli_options.h -- the command line of the SVDSS binary (Configuration, /root/reference/config.hpp:57-118 and
// config.cpp:26-107: the same option names, defaults and post-processing; cxxopts accepts `--opt value` and
// `--opt=value`).  In a header of its own so that tests/test_ref_pins.py can hold it against the reference's own
// config.cpp (oracle/_ref).  Additions of this program: --gpus N|all, --io-threads N.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

struct Options {
  std::string index, bam, fastx, reference, sfs, poa, clusters, append;
  unsigned min_sv_length = 25, min_mapq = 20, min_cluster_weight = 2;   // config.hpp:92-96 (uint there too)
  float accp = 0.98f, min_ratio = 0.97f;
  bool useht = true;
  int threads = 4, bsize = 10000, omax = 100000;  // config.hpp:68-69,88
  int io_threads = 0;                              // BGZF inflate workers (0: up to 16)
  int gpus = 1;                                    // --gpus N: index replicated, batches / sub-clusters shard
  bool gpus_all = false;                           // --gpus all (the caller asks the library how many there are)
  bool putative = true, assemble = true, verbose = false, version = false, help = false, clipped = false, binary = false;
};

namespace cli_detail {
// cxxopts' messages quote names like this (its LQUOTE / RQUOTE outside Windows)
inline std::string quoted(const std::string& s) { return "\xe2\x80\x98" + s + "\xe2\x80\x99"; }
// an integer option value as cxxopts takes it: "(-)?(0x)?([0-9a-zA-Z]+)|((0x)?0)", digits of the base, inside int
inline bool to_int(const std::string& v, int& out) {
  size_t i = 0;
  bool neg = false;
  if (i < v.size() && v[i] == '-') { neg = true; ++i; }
  int base = 10;
  if (i + 1 < v.size() && v[i] == '0' && v[i + 1] == 'x') { base = 16; i += 2; }
  if (i >= v.size()) return false;
  unsigned long long x = 0;
  for (; i < v.size(); ++i) {
    const char c = v[i];
    int d;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (base == 16 && c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else if (base == 16 && c >= 'A' && c <= 'F') d = c - 'A' + 10;
    else return false;
    x = x * (unsigned)base + (unsigned)d;
    if (x > 0x80000000ull) return false;
  }
  if (neg ? x > 0x80000000ull : x > 0x7fffffffull) return false;
  out = neg ? (int)(0 - (long long)x) : (int)x;
  return true;
}
// a float option value: what `stream >> float` takes from the front of the text (anything behind it is ignored)
inline bool to_float(const std::string& v, float& out) {
  size_t i = 0;
  while (i < v.size() && (v[i] == ' ' || v[i] == '\t' || v[i] == '\n')) ++i;   // (operator>> skips leading blanks)
  // the longest decimal prefix [+-]digits[.digits][e[+-]digits]: no hexadecimal, no "inf" / "nan" for a stream
  size_t j = i;
  if (j < v.size() && (v[j] == '+' || v[j] == '-')) ++j;
  size_t digits = 0;
  while (j < v.size() && v[j] >= '0' && v[j] <= '9') { ++j; ++digits; }
  if (j < v.size() && v[j] == '.') {
    ++j;
    while (j < v.size() && v[j] >= '0' && v[j] <= '9') { ++j; ++digits; }
  }
  if (digits == 0) return false;
  if (j < v.size() && (v[j] == 'e' || v[j] == 'E')) {
    size_t k = j + 1;
    if (k < v.size() && (v[k] == '+' || v[k] == '-')) ++k;
    if (k < v.size() && v[k] >= '0' && v[k] <= '9') {
      while (k < v.size() && v[k] >= '0' && v[k] <= '9') ++k;
      j = k;
    }
  }
  out = strtof(v.substr(i, j - i).c_str(), nullptr);
  return true;
}
// a boolean option's explicit value (--flag=true): cxxopts' "(t|T)(rue)?|1" and "(f|F)(alse)?|0"
inline bool to_bool(const std::string& v, bool& out) {
  if (v == "t" || v == "T" || v == "true" || v == "True" || v == "1") { out = true; return true; }
  if (v == "f" || v == "F" || v == "false" || v == "False" || v == "0") { out = false; return true; }
  return false;
}
}  // namespace cli_detail

// argv[first ..] into o; false and a message (the text cxxopts puts into its exception) if the line cannot be parsed.
// The rules are those of the reference's parser (cxxopts as vendored beside config.cpp; tests/test_ref_pins.py holds
// this function against it field by field): `--name value`, `--name=value`, groups of one-letter options (`-h`, `-l 0.5`),
// flags with an implicit "true" that never take the next argument, a lone "--" ending the options, arguments that do not
// start with '-' left alone, anything else that starts with '-' an error; numbers are checked, the last occurrence wins.
inline bool parse_options(int argc, char** argv, int first, Options& o, std::string& err) {
  using namespace cli_detail;
  enum Kind { STR, INT, FLT, FLAG };
  struct Spec { const char* name; Kind kind; };
  static const Spec specs[] = {
      {"bam", STR}, {"sfs", STR}, {"poa", STR}, {"clusters", STR}, {"index", STR}, {"fastx", STR}, {"reference", STR},
      {"append", STR}, {"threads", INT}, {"bsize", INT}, {"omax", INT}, {"min-sv-length", INT}, {"min-mapq", INT},
      {"min-cluster-weight", INT}, {"accp", FLT}, {"clipped", FLAG}, {"noht", FLAG}, {"noassemble", FLAG},
      {"noputative", FLAG}, {"binary", FLAG}, {"version", FLAG}, {"help", FLAG}, {"h", FLAG}, {"l", FLT}, {"verbose", FLAG},
      {"gpus", STR}, {"io-threads", INT}};   // (the last two: this program's own)
  auto find = [&](const std::string& name) -> const Spec* {
    for (const Spec& sp : specs)
      if (name == sp.name) return &sp;
    return nullptr;
  };
  auto failed = [&](const std::string& text) { err = "Argument " + quoted(text) + " failed to parse"; return false; };
  auto apply = [&](const Spec& sp, const std::string& v) -> bool {
    const std::string n = sp.name;
    int x = 0;
    float f = 0;
    bool b = true;
    if (sp.kind == INT && !to_int(v, x)) return failed(v);
    if (sp.kind == FLT && !to_float(v, f)) return failed(v);
    if (sp.kind == FLAG && !to_bool(v, b)) return failed(v);
    if (n == "bam") o.bam = v; else if (n == "sfs") o.sfs = v; else if (n == "poa") o.poa = v;
    else if (n == "clusters") o.clusters = v; else if (n == "index") o.index = v; else if (n == "fastx") o.fastx = v;
    else if (n == "reference") o.reference = v; else if (n == "append") o.append = v;
    else if (n == "threads") o.threads = x; else if (n == "bsize") o.bsize = x; else if (n == "omax") o.omax = x;
    else if (n == "min-sv-length") o.min_sv_length = (unsigned)std::max(25, x);     // config.cpp:87
    else if (n == "min-mapq") o.min_mapq = (unsigned)x; else if (n == "min-cluster-weight") o.min_cluster_weight = (unsigned)x;
    else if (n == "accp") o.accp = f; else if (n == "l") o.min_ratio = f;
    else if (n == "clipped") o.clipped = b;                                // config.cpp:46 (EXPERIMENTAL)
    else if (n == "noht") o.useht = !b; else if (n == "noassemble") o.assemble = !b; else if (n == "noputative") o.putative = !b;
    else if (n == "binary") o.binary = b; else if (n == "version") o.version = b; else if (n == "help" || n == "h") o.help = b;
    else if (n == "verbose") o.verbose = b; else if (n == "io-threads") o.io_threads = x;
    else if (n == "gpus") { if (v == "all") o.gpus_all = true; else if (!to_int(v, o.gpus)) return failed(v); }
    return true;
  };
  // the option's value when none is attached: the implicit "true" of a flag, else the next argument (taken whatever it
  // looks like), else an error
  auto with_next = [&](const Spec& sp, int& i) -> bool {
    if (sp.kind == FLAG) return apply(sp, "true");
    if (i + 1 >= argc) { err = "Option " + quoted(sp.name) + " is missing an argument"; return false; }
    return apply(sp, argv[++i]);
  };
  auto alnum = [](char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
  for (int i = first; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--") break;                                   // the rest are positional arguments
    if (a.size() < 2 || a[0] != '-') continue;              // a positional argument (or a lone '-')
    bool well_formed = false;
    if (a[1] == '-') {
      // --name or --name=value; a name is two or more of [alnum-_], the first alphanumeric
      const size_t eq = a.find('=');
      const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
      well_formed = name.size() >= 2 && alnum(name[0]);
      for (size_t k = 1; k < name.size() && well_formed; ++k) well_formed = alnum(name[k]) || name[k] == '-' || name[k] == '_';
      if (well_formed) {
        const Spec* sp = find(name);
        if (!sp || name.size() < 2) { err = "Option " + quoted(name) + " does not exist"; return false; }
        if (eq != std::string::npos ? !apply(*sp, a.substr(eq + 1)) : !with_next(*sp, i)) return false;
      }
    } else {
      // a group of one-letter options: all but the last must be flags, the last may take the next argument
      well_formed = true;
      for (size_t k = 1; k < a.size() && well_formed; ++k) well_formed = alnum(a[k]);
      if (well_formed) {
        for (size_t k = 1; k < a.size(); ++k) {
          const Spec* sp = find(std::string(1, a[k]));
          if (!sp) { err = "Option " + quoted(std::string(1, a[k])) + " does not exist"; return false; }
          if (k + 1 == a.size()) { if (!with_next(*sp, i)) return false; }
          else if (sp->kind == FLAG) { if (!apply(*sp, "true")) return false; }
          else { err = "Option " + quoted(sp->name) + " requires an argument"; return false; }
        }
      }
    }
    if (!well_formed) { err = "Argument " + quoted(a) + " starts with a - but has incorrect syntax"; return false; }
  }
  if (o.threads < 1) o.threads = 1;                // (the reference divides by it: config.cpp:106)
  o.bsize = (o.bsize / o.threads) * o.threads;     // config.cpp:106
  return true;
}
