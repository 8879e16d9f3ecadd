// ref_shim.cpp -- C entry points around two files of the reference that compile from their own sources
// (/root/reference/sv.cpp: the SV record and its VCF row; /root/reference/config.cpp with its vendored cxxopts.hpp: the
// command line), for tests/test_ref_pins.py.  TEST INFRASTRUCTURE (like everything under oracle/): built by
// `make -C oracle ref` into oracle/_ref/ where /root/reference is present, never linked into the product.  Nothing of
// the reference is copied here: this file only calls what those sources define.
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "config.hpp"
#include "sv.hpp"

static int put(const std::string& s, char* out, int cap) {
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// the VCF row the reference prints for an SV made by its constructor (sv.cpp:7-27, 53-80); `reads`: names joined by
// '\n' (add_reads, sv.cpp:29-33; none: not called).  COV0 / COV1 / COV2 / GQ, which nothing in the reference sets before
// printing, are set to 0 through its own setters (set_cov, set_gt) so that the row is defined.
extern "C" int ref_sv_row(const char* type, const char* chrom, unsigned s, const char* refall, const char* altall, unsigned w,
                          unsigned cov, int ngaps, int score, int imprecise, unsigned l, const char* cigar, const char* reads,
                          char* out, int cap) {
  SV sv(type, chrom, s, refall, altall, w, cov, ngaps, score, imprecise != 0, l, cigar);
  if (reads && *reads) {
    std::vector<std::string> names;
    std::stringstream ss(reads);
    std::string n;
    while (std::getline(ss, n, '\n')) names.push_back(n);
    sv.add_reads(names);
  }
  sv.set_cov((int)cov, 0, 0, 0);
  sv.set_gt(sv.gt, 0);
  std::ostringstream os;
  os << sv;
  return put(os.str(), out, cap);
}

// a < b by the reference's operator< (sv.hpp:48-56)
extern "C" int ref_sv_less(const char* chrom_a, unsigned s_a, const char* chrom_b, unsigned s_b) {
  SV a("INS", chrom_a, s_a, "A", "AC", 1, 1, 0, 0), b("INS", chrom_b, s_b, "A", "AC", 1, 1, 0, 0);
  return a < b ? 1 : 0;
}

// Configuration::parse on argv (config.cpp:59-107) in a child process -- the class is a singleton whose fields keep the
// values of earlier calls --, its fields as "name=value" lines, or "error=<what cxxopts threw>"
extern "C" int ref_config_parse(int argc, char** argv, char* out, int cap) {
  int fd[2];
  if (pipe(fd) != 0) return -1;
  const pid_t pid = fork();
  if (pid < 0) return -1;
  if (pid == 0) {
    close(fd[0]);
    std::ostringstream os;
    try {
      Configuration* c = Configuration::getInstance();
      c->parse(argc, argv);
      os << "index=" << c->index << "\nbam=" << c->bam << "\nfastx=" << c->fastq << "\nreference=" << c->reference << "\nsfs=" << c->sfs
         << "\npoa=" << c->poa << "\nclusters=" << c->clusters << "\nappend=" << c->append << "\nthreads=" << c->threads
         << "\nbsize=" << c->batch_size << "\nomax=" << c->max_output << "\nmin_sv_length=" << c->min_sv_length
         << "\nmin_mapq=" << c->min_mapq << "\nmin_cluster_weight=" << c->min_cluster_weight << "\naccp=" << c->accp
         << "\nmin_ratio=" << c->min_ratio << "\nuseht=" << c->useht << "\nputative=" << c->putative << "\nassemble=" << c->assemble
         << "\nverbose=" << c->verbose << "\nversion=" << c->version << "\nhelp=" << c->help << "\nclipped=" << c->clipped
         << "\nbinary=" << c->binary << "\n";
    } catch (const std::exception& e) {
      os.str("");
      os << "error=" << e.what() << "\n";
    }
    const std::string s = os.str();
    size_t off = 0;
    while (off < s.size()) {
      const ssize_t n = write(fd[1], s.data() + off, s.size() - off);
      if (n <= 0) break;
      off += (size_t)n;
    }
    _exit(0);
  }
  close(fd[1]);
  std::string s;
  char buf[4096];
  ssize_t n;
  while ((n = read(fd[0], buf, sizeof buf)) > 0) s.append(buf, (size_t)n);
  close(fd[0]);
  int status = 0;
  waitpid(pid, &status, 0);
  if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) s = "crash=" + std::to_string(WIFSIGNALED(status) ? WTERMSIG(status) : -1) + "\n";
  return put(s, out, cap);
}
