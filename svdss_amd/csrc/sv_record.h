// sv_record.h -- one called SV and its VCF row (SV, /root/reference/sv.hpp:12-61, sv.cpp:5-80).  In a header of its own
// so that tests/test_ref_pins.py can hold it against the reference's own sv.cpp (oracle/_ref, built where /root/reference
// is present): constructor arithmetic (END, the record id), row text, ordering.
#pragma once
#include <cstdlib>
#include <string>

struct SV {   // sv.hpp / sv.cpp
  std::string type, chrom, idx, refall, altall, gt = "./.", cigar, reads, rvec;
  int s = 0, e = 0, cov = 0, cov0 = 0, cov1 = 0, cov2 = 0, l = 0, ngaps = 0, score = 0, gtq = 0;
  unsigned w = 0;
  bool imprecise = false;
  bool operator<(const SV& c) const { return chrom < c.chrom ? true : (chrom > c.chrom ? false : s < c.s); }
  std::string line() const {   // sv.cpp:53-80
    std::string o = chrom + "\t" + std::to_string(s) + "\t" + idx + "\t" + refall + "\t" + altall + "\t.\tPASS\t";
    o += "VARTYPE=SV;SVTYPE=" + type + ";SVLEN=" + std::to_string(type == "DEL" ? -l : l) + ";END=" + std::to_string(e);
    o += ";WEIGHT=" + std::to_string(w) + ";COV=" + std::to_string(cov) + ";COV0=" + std::to_string(cov0);
    o += ";COV1=" + std::to_string(cov1) + ";COV2=" + std::to_string(cov2) + ";AS=" + std::to_string(score);
    o += ";NV=" + std::to_string(ngaps) + ";CIGAR=" + cigar + ";RVEC=" + rvec + ";READS=" + reads;
    o += imprecise ? ";IMPRECISE\t" : "\t";
    o += "GT:GQ\t" + gt + ":" + std::to_string(gtq);
    return o;
  }
};

inline SV make_sv(const std::string& type, const std::string& chrom, int s, const std::string& refall,
           const std::string& altall, unsigned w, int cov, int ngaps, int score, int l, const std::string& cigar) {
  SV v;
  v.type = type; v.chrom = chrom; v.s = s; v.refall = refall; v.altall = altall;
  v.e = s + (int)refall.size() - 1;
  v.w = w; v.l = l; v.cov = cov; v.ngaps = ngaps; v.score = score; v.cigar = cigar;
  v.idx = type + "_" + chrom + ":" + std::to_string(s) + "-" + std::to_string(v.e) + "_" + std::to_string(std::abs(l));
  return v;
}
