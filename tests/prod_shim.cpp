// prod_shim.cpp -- C entry points around the product's headers csrc/sv_record.h and csrc/cli_options.h, the counterpart
// of oracle/ref_shim.cpp for tests/test_ref_pins.py (test infrastructure).
#include <cstring>
#include <sstream>
#include <string>

#include "../svdss_amd/csrc/cli_options.h"
#include "../svdss_amd/csrc/sv_record.h"

static int put(const std::string& s, char* out, int cap) {
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

extern "C" int prod_sv_row(const char* type, const char* chrom, unsigned s, const char* refall, const char* altall, unsigned w,
                           unsigned cov, int ngaps, int score, int imprecise, unsigned l, const char* cigar, const char* reads,
                           char* out, int cap) {
  SV v = make_sv(type, chrom, (int)s, refall, altall, w, (int)cov, ngaps, score, (int)l, cigar);
  v.imprecise = imprecise != 0;
  std::string names;                      // (call_host.cpp joins the names with ',' as add_reads does)
  if (reads && *reads) {
    names = reads;
    for (char& c : names) if (c == '\n') c = ',';
  }
  v.reads = names;
  return put(v.line(), out, cap);
}

extern "C" int prod_sv_less(const char* chrom_a, unsigned s_a, const char* chrom_b, unsigned s_b) {
  SV a = make_sv("INS", chrom_a, (int)s_a, "A", "AC", 1, 1, 0, 0, 0, "."), b = make_sv("INS", chrom_b, (int)s_b, "A", "AC", 1, 1, 0, 0, 0, ".");
  return a < b ? 1 : 0;
}

extern "C" int prod_config_parse(int argc, char** argv, char* out, int cap) {
  Options o;
  std::string err;
  std::ostringstream os;
  if (!parse_options(argc, argv, 1, o, err)) os << "error=" << err << "\n";
  else
    os << "index=" << o.index << "\nbam=" << o.bam << "\nfastx=" << o.fastx << "\nreference=" << o.reference << "\nsfs=" << o.sfs
       << "\npoa=" << o.poa << "\nclusters=" << o.clusters << "\nappend=" << o.append << "\nthreads=" << o.threads
       << "\nbsize=" << o.bsize << "\nomax=" << o.omax << "\nmin_sv_length=" << o.min_sv_length << "\nmin_mapq=" << o.min_mapq
       << "\nmin_cluster_weight=" << o.min_cluster_weight << "\naccp=" << o.accp << "\nmin_ratio=" << o.min_ratio
       << "\nuseht=" << o.useht << "\nputative=" << o.putative << "\nassemble=" << o.assemble << "\nverbose=" << o.verbose
       << "\nversion=" << o.version << "\nhelp=" << o.help << "\nclipped=" << o.clipped << "\nbinary=" << o.binary << "\n";
  return put(os.str(), out, cap);
}
