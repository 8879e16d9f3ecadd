// call_host.cpp -- host side of `SVDSS call` (/root/reference/caller.cpp, clusterer.cpp, sv.cpp).
//
// Integer/interval bookkeeping restated from the reference;
// the three DP seams go to the GPU through the C-ABI in three batched calls:
//   svdss_poa_consensus_batch (abPOA, caller.cpp:291), svdss_align_global_batch (ksw_extd2_sse,
//   caller.cpp:348), svdss_indel_ratio_batch (rapidfuzz::fuzz::ratio, caller.cpp:456,458).
// Thread-count dependent orderings of the reference are replayed with T = --threads.
// BAM access: the reference streams the BAM once (placement) and then issues one BAI region query
// per cluster; here the second phase is a second sequential pass that hands every alignment to
// the clusters it overlaps -- same alignments per cluster, same (file) order, no index needed.
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/svdss_hip.h"
#include "bai_index.h"
#include "bam_reader.h"
#include "bam_device_select.h"
#include "gpu_inflate_hook.h"
#include "sfs_file.h"
#include "sv_record.h"
#include "call_host.h"
#include "fastx_reader.h"

namespace {

void logmsg(const char* lvl, const std::string& m) { fprintf(stderr, "[call] [%s] %s\n", lvl, m.c_str()); }
[[noreturn]] void die(const std::string& m) { logmsg("critical", m); exit(EXIT_FAILURE); }
void check(int rc, const char* what) {
  if (rc != SVDSS_OK) die(std::string(what) + ": " + svdss_strerror(rc) + " " + svdss_last_hip_error());
}


struct ESFS {   // SFS after placement (sfs.hpp:52-62)
  std::string chrom, qname;
  int rs, re, qs, qe, htag;
  bool operator<(const ESFS& c) const { return chrom == c.chrom ? rs < c.rs : chrom < c.chrom; }  // sfs.hpp:66-73
};

struct SubRead { std::string name, seq; int htag; };

struct Cluster {   // clusterer.hpp:38-139
  std::string chrom;
  int s = 0, e = 0, cov = 0, cov0 = 0, cov1 = 0, cov2 = 0;
  std::vector<ESFS> sfss;
  std::vector<std::pair<int, int>> reads;
  std::vector<SubRead> subreads;
  Cluster() {}
  Cluster(const std::string& c, int s_, int e_, int cov_, int c0, int c1, int c2)
      : chrom(c), s(s_), e(e_), cov(cov_), cov0(c0), cov1(c1), cov2(c2) {}
  size_t size() const { return subreads.size(); }
  int get_len() const {
    unsigned l = 0, n = 0;
    for (const auto& sr : subreads) { ++n; l += (unsigned)sr.seq.size(); }
    return (int)(l / n);
  }
  Cluster cleared_copy() const { return Cluster(chrom, s, e, cov, cov0, cov1, cov2); }  // copy ctor drops `reads`
};

typedef std::vector<std::pair<int, int>> Pairs;

// bam.cpp:92-134
Pairs get_aligned_pairs(const BamRecord& a) {
  Pairs res;
  int ref_pos = a.pos, read_pos = 0;
  for (uint32_t c : a.cigar) {
    const int l = (int)(c >> 4), op = (int)(c & 0xf);
    if (op == 0 || op == 7 || op == 8) {
      for (int i = 0; i < l; ++i) res.emplace_back(read_pos + i, ref_pos + i);
      read_pos += l; ref_pos += l;
    } else if (op == 1 || op == 4) {
      for (int i = 0; i < l; ++i) res.emplace_back(read_pos + i, -1);
      read_pos += l;
    } else if (op == 2 || op == 3) {
      for (int i = 0; i < l; ++i) res.emplace_back(-1, ref_pos + i);
      ref_pos += l;
    }
  }
  return res;
}

// clusterer.cpp:351-405
std::pair<int, int> get_unique_kmers(const Pairs& al, size_t k, bool from_end, const std::string& cseq) {
  if (al.size() < k) return {-1, -1};
  std::map<std::string, int> kmers;
  size_t i = 0;
  while (i < al.size() - k + 1) {
    bool skip = false;
    for (size_t j = i; j < i + k; ++j)
      if (al[j].first == -1 || al[j].second == -1) { skip = true; i = j + 1; break; }
    if (skip) continue;
    ++kmers[cseq.substr((size_t)al[i].second, k)];
    ++i;
  }
  std::pair<int, int> last(-1, -1);
  i = 0;
  while (i < al.size() - k + 1) {
    size_t offset = from_end ? al.size() - k - i : i;
    bool skip = false;
    for (size_t j = offset; j < offset + k; ++j)
      if (al[j].first == -1 || al[j].second == -1) { skip = true; i += (j - offset); break; }
    if (skip) { ++i; continue; }
    last = al[offset];
    if (kmers[cseq.substr((size_t)al[offset].second, k)] == 1) break;
    ++i;
  }
  return last;
}

struct Ctx {
  CallOptions o;
  std::vector<std::string> chrom_names;
  std::unordered_map<std::string, std::string> chrom_seqs;
  std::unordered_map<std::string, std::vector<RawSFS>> sfs;
  // statistics (bumped from the worker threads of pass 1)
  std::atomic<long> unplaced{0}, s_unplaced{0}, e_unplaced{0}, unknown{0}, unextended{0}, small{0}, small2{0};
};

// a soft clip next to SFS that could not be placed on one side (clipper.hpp:21-42)
struct Clip {
  std::string name, chrom;
  unsigned p = 0, l = 0;
  bool starting = false;   // leading clip (the read continues to the right of p)
  unsigned w = 0;
};

// clusterer.cpp:159-346; clips: --clipped bookkeeping (:207-226, :338-345), nullptr without the option
void extend_alignment(Ctx& C, const BamRecord& aln, const std::string& chrom, std::vector<ESFS>& out,
                      std::vector<Clip>* clips = nullptr) {
  auto cs = C.chrom_seqs.find(chrom);
  if (cs == C.chrom_seqs.end()) return;
  const std::string& cseq = cs->second;
  const Pairs al = get_aligned_pairs(aln);
  const int k = 7, flank = 100;   // config.hpp:89-90, not settable
  size_t last_pos = 0;
  std::vector<ESFS> local;
  unsigned lclip_p = 0, lclip_l = 0, rclip_p = 0, rclip_l = 0;
  for (const RawSFS& sfs : C.sfs.at(aln.qname)) {
    const int s = sfs.qs, e = sfs.qs + sfs.l - 1;
    int aln_start = -1, aln_end = -1, refs = -1, refe = -1;
    for (size_t i = last_pos; i < al.size(); ++i) {
      const int q = al[i].first, r = al[i].second;
      if (q == -1 || r == -1) continue;
      else if (q < s) { last_pos = i; refs = r; aln_start = (int)i; }
      else if (q > e) { refe = r; aln_end = (int)i; break; }
    }
    if (refs == -1 && refe == -1) { ++C.unplaced; continue; }
    else if (refs == -1) {
      // the SFS starts in front of the first placed base: with --clipped a leading soft clip is remembered instead of
      // counting the SFS (every such SFS of the alignment sets the same pair)
      const uint32_t c0 = aln.cigar.empty() ? 0u : aln.cigar.front();
      if (clips && (c0 & 0xf) == 4) { lclip_p = (unsigned)aln.pos; lclip_l = c0 >> 4; }
      else ++C.s_unplaced;
      continue;
    } else if (refe == -1) {
      const uint32_t c1 = aln.cigar.empty() ? 0u : aln.cigar.back();
      if (clips && (c1 & 0xf) == 4) { rclip_p = (unsigned)aln.endpos(); rclip_l = c1 >> 4; }
      else ++C.e_unplaced;
      continue;
    }
    Pairs loc;
    int last_r = refs - 1;
    for (int i = aln_start; i <= aln_end; ++i) {
      const int q = al[(size_t)i].first, r = al[(size_t)i].second;
      if (r == -1) { if (refs <= last_r && last_r <= refe) loc.emplace_back(q, r); }
      else { last_r = r; if (refs <= r && r <= refe) loc.emplace_back(q, r); }
      if (q != -1 && r != -1 && r >= refe) break;
    }
    Pairs pre, post;
    for (int i = aln_start - 1; i >= 0; --i) { pre.push_back(al[(size_t)i]); if ((int)pre.size() == flank) break; }
    std::reverse(pre.begin(), pre.end());
    for (size_t i = (size_t)aln_end + 1; i < al.size(); ++i) { post.push_back(al[i]); if ((int)post.size() == flank) break; }
    std::pair<int, int> prek = get_unique_kmers(pre, (size_t)k, true, cseq);
    std::pair<int, int> postk = get_unique_kmers(post, (size_t)k, false, cseq);
    if (prek.first == -1 || prek.second == -1) prek = loc.front();
    if (postk.first == -1 || postk.second == -1) postk = loc.back();
    if (prek.first == -1 || prek.second == -1 || postk.first == -1 || postk.second == -1) { ++C.unknown; continue; }
    if ((unsigned)prek.second > (unsigned)(postk.second + k)) continue;   // warning only (:301-303)
    local.push_back(ESFS{chrom, aln.qname, prek.second, postk.second + k, prek.first, postk.first + k, sfs.htag});
  }
  std::vector<ESFS> merged;   // :314-336
  for (const ESFS& x : local) {
    size_t j;
    for (j = 0; j < merged.size(); ++j)
      if ((x.rs <= merged[j].rs && merged[j].rs <= x.re) || (merged[j].rs <= x.rs && x.rs <= merged[j].re)) break;
    if (j < merged.size()) {
      merged[j].rs = std::min(merged[j].rs, x.rs); merged[j].re = std::max(merged[j].re, x.re);
      merged[j].qs = std::min(merged[j].qs, x.qs); merged[j].qe = std::max(merged[j].qe, x.qe);
    } else merged.push_back(x);
  }
  out.insert(out.end(), merged.begin(), merged.end());
  if (clips) {   // clusterer.cpp:338-345
    if (lclip_l > 0) clips->push_back(Clip{aln.qname, chrom, lclip_p, lclip_l, true, 0});
    if (rclip_l > 0) clips->push_back(Clip{aln.qname, chrom, rclip_p, rclip_l, false, 0});
  }
}

float len_ratio(float cl, float sl) { return std::min(cl, sl) / std::max(cl, sl); }

// caller.cpp:78-97
std::vector<Cluster> split_by_len(const Cluster& c, float min_ratio) {
  std::vector<Cluster> subs;
  for (const SubRead& sr : c.subreads) {
    size_t i;
    for (i = 0; i < subs.size(); ++i)
      if (len_ratio((float)subs[i].get_len(), (float)(unsigned)sr.seq.size()) >= min_ratio) break;
    if (i == subs.size()) subs.push_back(c.cleared_copy());
    subs[i].subreads.push_back(sr);
  }
  return subs;
}

int largest(const std::vector<Cluster>& v) {
  unsigned vmax = 0; int imax = -1;
  for (size_t i = 0; i < v.size(); ++i) if (v[i].size() > vmax) { vmax = (unsigned)v[i].size(); imax = (int)i; }
  return imax;
}

// caller.cpp:100-255 (int-typed best_ratio included)
std::vector<Cluster> split_cluster(const Cluster& cluster, bool useht, float min_ratio) {
  Cluster c0 = cluster.cleared_copy(), c1 = cluster.cleared_copy(), c2 = cluster.cleared_copy();
  for (const SubRead& sr : cluster.subreads) {
    if (useht && sr.htag == 1) c1.subreads.push_back(sr);
    else if (useht && sr.htag == 2) c2.subreads.push_back(sr);
    else c0.subreads.push_back(sr);
  }
  c0.cov1 = c0.cov2 = -1; c1.cov0 = c1.cov2 = -1; c2.cov0 = c2.cov1 = -1;
  std::vector<Cluster> out;
  if (c1.size() == 0 && c2.size() == 0) {
    std::vector<Cluster> subs = split_by_len(c0, min_ratio);
    int i1 = -1, i2 = -1; unsigned v1 = 0, v2 = 0;
    for (unsigned i = 0; i < subs.size(); ++i) {
      if (subs[i].size() > v1) { v2 = v1; i2 = i1; v1 = (unsigned)subs[i].size(); i1 = (int)i; }
      else if (subs[i].size() > v2) { v2 = (unsigned)subs[i].size(); i2 = (int)i; }
    }
    if (i1 != -1) out.push_back(subs[(size_t)i1]);
    if (i2 != -1) out.push_back(subs[(size_t)i2]);
    return out;
  }
  const int both = (c1.size() > 0 ? 1 : 0) + (c2.size() > 0 ? 2 : 0);
  std::vector<Cluster> s1 = split_by_len(c1, min_ratio), s2 = split_by_len(c2, min_ratio);
  Cluster nc(cluster.chrom, cluster.s, cluster.e, cluster.cov, cluster.cov0, -1, -1);
  for (const SubRead& sr : c0.subreads) {
    const float sl = (float)(unsigned)sr.seq.size();
    int best_1 = -1, best_ratio_1 = -1, best_2 = -1, best_ratio_2 = -1;
    for (unsigned i = 0; i < s1.size(); ++i) {
      const float r = len_ratio((float)s1[i].get_len(), sl);
      if (r >= min_ratio && r > best_ratio_1) { best_1 = (int)i; best_ratio_1 = (int)r; }
    }
    for (unsigned i = 0; i < s2.size(); ++i) {
      const float r = len_ratio((float)s2[i].get_len(), sl);
      if (r >= min_ratio && r > best_ratio_2) { best_2 = (int)i; best_ratio_2 = (int)r; }
    }
    if (both == 1) {
      if (best_1 == -1) nc.subreads.push_back(sr);
      else { s1[(size_t)best_1].subreads.push_back(sr); ++s1[(size_t)best_1].cov1; --nc.cov0; }
    } else if (both == 2) {
      if (best_2 == -1) nc.subreads.push_back(sr);
      else { s2[(size_t)best_2].subreads.push_back(sr); ++s2[(size_t)best_2].cov2; --nc.cov0; }
    } else {
      if (best_1 != -1 && best_ratio_1 > best_ratio_2) { s1[(size_t)best_1].subreads.push_back(sr); ++s1[(size_t)best_1].cov1; --nc.cov0; }
      else if (best_2 != -1 && best_ratio_2 > best_ratio_1) { s2[(size_t)best_2].subreads.push_back(sr); ++s2[(size_t)best_2].cov2; --nc.cov0; }
    }
  }
  int i = largest(s1);
  if (i != -1) out.push_back(s1[(size_t)i]);
  i = largest(s2);
  if (i != -1) out.push_back(s2[(size_t)i]);
  if (both != 3 && nc.size() > 0) {
    std::vector<Cluster> ns = split_by_len(nc, min_ratio);
    i = largest(ns);
    if (i != -1) {
      if (both == 1) ns[(size_t)i].cov1 = -1; else ns[(size_t)i].cov2 = -1;
      out.push_back(ns[(size_t)i]);
    }
  }
  return out;
}

// ---- imprecise SVs from soft clips (--clipped; Clipper, clipper.cpp) -------------------------------------------
//
// What the reference computes, stage by stage, with its quirks kept where they are defined behaviour:
//   * a side's clips (leading / trailing) -> first clip per read name (:5-15) -> one breakpoint per (chromosome,
//     position) with the longest clip and the number of reads (:17-52), listed chromosome slot by slot (i % 4, the
//     slots concatenated back to front) and inside a chromosome in the iteration order of a
//     std::unordered_map<uint, ...> filled in list order -- the same container here, hence the same order as a
//     reference built against libstdc++;
//   * breakpoints seen by fewer than two reads are dropped (:54-63), so are those within the +-1,000 bp window of
//     a called SV of ANY chromosome (the interval tree has no chromosome, caller.cpp:39-41, clipper.cpp:93-102);
//   * greedy grouping in list order (:66-91): a breakpoint within 1,000 bp of existing centres adds its reads to
//     ALL of them, otherwise it becomes a centre; `centre - 1000` is unsigned arithmetic, so a centre below 1,000
//     never absorbs anything (and is overwritten by a breakpoint at its own position);
//   * insertions (:170-194): every leading centre looks up a trailing centre with the search of :104-124 and calls
//     <INS> when the two are less than 1,000 bp apart; deletions (:196-214): every trailing centre looks up a leading
//     one 2,000 to 50,000 bp to its right, at least five reads; positions of different chromosomes are compared as
//     plain numbers;
//   * rows leave per OpenMP slot (i % threads; insertions before deletions), the slots back to front (caller.cpp:45-50).
// Where the reference is undefined this code is defined: the search of :104-124 walks off the
// array when the query lies left of every centre of the other side (`m - 1` on an unsigned 0: it reads two billion
// entries past the list) -- here it returns what its comment says it looks for, the first centre right of the query; the
// record's COV0 / COV1 / COV2 / GQ are printed from fields nothing ever set (sv.cpp:5-27) -- 0 here; a reference base
// beyond the chromosome's end (positions of two chromosomes mixed) is 'N'.
namespace clipper {

std::vector<Clip> breakpoints(const std::vector<Clip>& side, const std::vector<std::string>& chromosomes,
                              const std::vector<std::pair<int, int>>& called) {
  std::vector<const Clip*> first;
  {
    std::set<std::string> seen;
    for (const Clip& c : side)
      if (seen.insert(c.name).second) first.push_back(&c);
  }
  std::unordered_map<std::string, std::unordered_map<unsigned, std::vector<const Clip*>>> at;
  for (const Clip* c : first) at[c->chrom][c->p].push_back(c);
  std::vector<Clip> slot[4];
  for (size_t i = 0; i < chromosomes.size(); ++i) {
    const auto chrom = at.find(chromosomes[i]);
    if (chrom == at.end()) continue;
    for (const auto& bp : chrom->second) {
      Clip b;
      b.chrom = chromosomes[i];
      b.p = bp.first;
      for (const Clip* c : bp.second) b.l = std::max(b.l, c->l);
      b.starting = bp.second.front()->starting;
      b.w = (unsigned)bp.second.size();
      slot[i % 4].push_back(b);
    }
  }
  std::map<unsigned, Clip> centres;
  for (int k = 3; k >= 0; --k)
    for (const Clip& b : slot[k]) {
      if (b.w < 2) continue;
      bool near_call = false;
      for (const auto& iv : called)
        if (iv.first <= (int)b.p + 1 && (int)b.p <= iv.second) { near_call = true; break; }
      if (near_call) continue;
      bool absorbed = false;
      for (auto& kv : centres)
        if (kv.first - 1000u <= b.p && b.p <= kv.first + 1000u) {
          absorbed = true;
          kv.second.l = std::max(kv.second.l, b.l);
          kv.second.w += b.w;
        }
      if (!absorbed) centres[b.p] = b;
    }
  std::vector<Clip> out;
  for (const auto& kv : centres) out.push_back(kv.second);
  return out;   // ascending position (std::map), what the sort of :151,158 leaves
}

// clipper.cpp:104-124 on a list sorted by position: the entry after one at the query's position (that entry itself
// when it is the last), else the first entry to the right of the query whose left neighbour lies left of it; -1
// when there is none; a query left of every entry gets entry 0 (the reference's recursion leaves the array there)
int partner(const std::vector<Clip>& v, unsigned q) {
  if (v.empty()) return -1;
  size_t lo = 0, hi = v.size() - 1;
  for (;;) {
    if (lo > hi || lo >= v.size()) return -1;
    const size_t m = (lo + hi) / 2;
    if (v[m].p == q) return (int)(m + 1 < v.size() ? m + 1 : m);
    if (v[m].p > q) {
      if (m > 0 && v[m - 1].p < q) return (int)m;
      if (m == 0) return 0;               // (the reference goes on with end = UINT_MAX)
      hi = m - 1;
    } else lo = m + 1;
  }
}

}  // namespace clipper

// Clipper::call + the caller's collection (clipper.cpp:126-215, caller.cpp:36-53): rows in output order
std::vector<SV> call_clipped(const std::vector<Clip>& clips, const std::vector<std::string>& chromosomes,
                             const std::unordered_map<std::string, std::string>& chrom_seqs, int T,
                             const std::vector<std::pair<int, int>>& called) {
  std::vector<Clip> lead, trail;
  for (const Clip& c : clips) (c.starting ? lead : trail).push_back(c);
  const std::vector<Clip> rc = clipper::breakpoints(trail, chromosomes, called);
  const std::vector<Clip> lc = clipper::breakpoints(lead, chromosomes, called);
  std::vector<std::vector<SV>> per_slot((size_t)T);
  if (!lc.empty() && !rc.empty()) {
    auto base_at = [&](const std::string& chrom, unsigned p) {
      const auto it = chrom_seqs.find(chrom);
      return std::string(1, it != chrom_seqs.end() && p < it->second.size() ? it->second[p] : 'N');
    };
    for (size_t i = 0; i < lc.size(); ++i) {
      const Clip& l = lc[i];
      const int k = clipper::partner(rc, l.p);
      if (k < 0) continue;
      const Clip& r = rc[(size_t)k];
      if (r.w == 0) continue;
      if (std::abs((int)r.p - (int)l.p) < 1000) {
        const unsigned s = l.w > r.w ? l.p : r.p;
        SV v = make_sv("INS", l.chrom, (int)s, base_at(l.chrom, s), "<INS>", std::max(l.w, r.w), 0, 0, 0,
                       (int)std::max(l.l, r.l), ".");
        v.imprecise = true;
        per_slot[i % (size_t)T].push_back(v);
      }
    }
    for (size_t i = 0; i < rc.size(); ++i) {
      const Clip& r = rc[i];
      const int k = clipper::partner(lc, r.p);
      if (k < 0) continue;
      const Clip& l = lc[(size_t)k];
      if (l.w == 0) continue;
      const unsigned gap = l.p - r.p;     // (unsigned: a leading centre left of the trailing one is out of range)
      if (gap >= 2000 && gap <= 50000 && std::max(l.w, r.w) >= 5) {
        SV v = make_sv("DEL", r.chrom, (int)r.p, base_at(r.chrom, r.p), "<DEL>", std::max(l.w, r.w), 0, 0, 0,
                       (int)(gap + 1), ".");
        v.imprecise = true;
        per_slot[i % (size_t)T].push_back(v);
      }
    }
  }
  std::vector<SV> rows;
  for (size_t t = (size_t)T; t-- > 0;) rows.insert(rows.end(), per_slot[t].begin(), per_slot[t].end());
  return rows;
}

const char* VCF_INFO[][4] = {
    {"VARTYPE", "A", "String", "Variant class"}, {"SVTYPE", "1", "String", "Variant type"},
    {"SVLEN", "1", "Integer", "Difference in length between REF and ALT alleles"},
    {"END", "1", "Integer", "End position of the variant described in this record"},
    {"WEIGHT", "1", "Integer", "Number of alignments supporting this record"},
    {"COV", "1", "Integer", "Total number of alignments covering this locus"},
    {"COV0", "1", "Integer", "Total number of alignments covering this locus (no HP)"},
    {"COV1", "1", "Integer", "Total number of alignments covering this locus (HP=1)"},
    {"COV2", "1", "Integer", "Total number of alignments covering this locus (HP=2)"},
    {"AS", "1", "Integer", "Alignment score"}, {"NV", "1", "Integer", "Number of variations on same consensus"},
    {"IMPRECISE", "0", "Flag", "Imprecise structural variation"}, {"CIGAR", "A", "String", "CIGAR of consensus"},
    {"READS", ".", "String", "Reads identifiers supporting the call"},
    {"RVEC", ".", "String", "Reads vector used by genotyper"}};

uint8_t enc26(char c) {   // caller.hpp:25-37
  switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
               case 'T': case 't': return 3; default: return 4; }
}

}  // namespace

namespace {

// `SVDSS call`, stage by stage in the order of Caller::run / Clusterer::run (caller.cpp:12-57, clusterer.cpp:12-54); the
// members are what one stage hands to the next.
struct CallRun {
  const CallOptions& o;
  Ctx C;
  const int T;
  std::chrono::steady_clock::time_point t_last = std::chrono::steady_clock::now();
  std::vector<std::string> ref_names;    // BAM header order
  std::vector<ESFS> extended;            // pass 1: every placed SFS
  std::vector<Clip> clips;               // --clipped
  // The inflated records of pass 1 are kept for pass 2 when they fit in memory (a second inflate of the whole file
  // otherwise): SVDSS_CALL_CACHE_GB, default 40 % of MemAvailable, at most 64 GiB.
  std::thread fasta_loader;              // load_chromosomes, beside the SFS file and the start of pass 1
  // ONE pass over the BAM (round 6): what pass 2 needs of every record stays in HBM while pass 1 runs (svdss_bam_store_t)
  // (--gpus N: the file's regions, one per GPU, each with a store of its own in that GPU's HBM -- and a small one for the
  // records of the seam in front of it; ShardedBamSelect, bam_device_select.h)
  std::vector<svdss_bam_store_t*> stores, seam_stores;
  std::vector<size_t> bam_cuts;            // where the regions begin (plan_bam_regions); one region: {0, file size}
  std::vector<int64_t> region_batches;     // after pass 1: batches per region, and whether a seam batch precedes it
  std::vector<char> region_seam;
  int32_t n_ref_hdr = 0;
  std::thread store_alloc;
  double store_alloc_s = 0;
  void reference_ready() { if (fasta_loader.joinable()) fasta_loader.join(); }
  bool dev_pass = false;                 // the BAM is read through the device path (csrc/bam_device.hip): no record cache
  int64_t bam_skip = 0;                  // the BAM header's length in the inflated stream
  uint64_t n_records_seen = 0;
  static int64_t bam_batch_bytes() {
    const char* e = getenv("SVDSS_BAM_BATCH_MB");
    return (e && atoll(e) > 0 ? atoll(e) : 256) << 20;
  }
  // host threads that scan the file chunks an index names for pass 2 (SVDSS_CALL_PASS2_THREADS; 1: the single scan)
  size_t pass2_threads(size_t n_chunks) const {
    const char* e = getenv("SVDSS_CALL_PASS2_THREADS");
    const size_t want = e && atoi(e) > 0 ? (size_t)atoi(e) : std::min<size_t>((size_t)effective_cpus(), 32);
    return std::max<size_t>(1, std::min(want, n_chunks / 4 + 1));
  }
  static int bam_feeders() {             // feeding threads (device batches in flight) per GPU of the two BAM passes
    const char* e = getenv("SVDSS_CALL_FEEDERS");
    return e && atoi(e) > 0 ? atoi(e) : 3;
  }
  std::vector<BamReader::RawView> cache_views;
  std::vector<std::shared_ptr<BamReader::Bytes>> cache_chunks;
  size_t cache_bytes = 0, cache_limit = 0;
  bool cache_ok = true;
  std::thread cache_release;
  std::vector<Cluster> clusters;
  int n_dev = 1, G = 1;                  // GPUs present / shards of the DP batches (--gpus)
  struct Sub { size_t parent; Cluster cl; };
  std::vector<Sub> subs;                 // sub-clusters after split_cluster, in the reference's order
  std::vector<std::string> consensus;    // one per sub-cluster
  std::vector<SV> svs;
  std::vector<std::vector<std::string>> sam_rows;   // per reference thread, --poa only

  explicit CallRun(const CallOptions& opt) : o(opt), T(std::max(1, opt.threads)) { C.o = opt; }

  void stage(const char* what) {   // --verbose: seconds since the previous stage mark
    if (!o.verbose) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[call] [time] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  }

  // load_chromosomes + parse_sfsfile (chromosomes.cpp:9-27, sfs.cpp:5-30)
  void load_inputs() {
    // ---- load_chromosomes (chromosomes.cpp:9-27): upper-cased, FASTA order.  On a thread of its own: the SFS file is
    // parsed and pass 1 starts reading the BAM beside it; the first use of the chromosomes waits (reference_ready)
    {
      if (FILE* f = fopen(o.reference.c_str(), "rb")) fclose(f); else die("cannot open " + o.reference);
      fasta_loader = std::thread([this] {
        {   // a plain FASTA with '\n' line ends: mapped and read by several threads (fastx_reader.h)
          std::vector<std::string> nm, sq;
          if (!getenv("SVDSS_FASTA_SERIAL") && load_fasta_mapped(o.reference, std::max(1, std::min(T, 16)), true, nm, sq)) {
            for (size_t i = 0; i < nm.size(); ++i) {
              C.chrom_names.push_back(nm[i]);
              C.chrom_seqs[nm[i]] = std::move(sq[i]);   // (a name that occurs twice: the later record wins, as below)
            }
            return;
          }
        }
        FastxReader fx(o.reference);
        if (!fx.ok()) die("cannot open " + o.reference);
        std::string name, seq;
        while (fx.next(name, seq)) {
          for (char& ch : seq) ch = (char)(ch - ((ch >= 'a' && ch <= 'z') ? 32 : 0));   // toupper of chromosomes.cpp:19 (ASCII; vectorises)
          C.chrom_names.push_back(name);
          C.chrom_seqs[name] = seq;
        }
      });
    }
    // the record store(s) of the ONE pass over the BAM (round 6, align_and_extend / fill_clusters below): the memory is taken
    // NOW, on a thread of its own, beside the FASTA and the SFS file -- tens of GB that the driver clears before it hands them out
    {
      const int n_phys = std::max(1, svdss_device_count());
      const int n_dev = std::max(1, getenv("SVDSS_GPUS_OVERSUBSCRIBE") ? o.gpus : std::min(o.gpus, n_phys));
      const bool dev_bam = svdss_device_count() > 0 && !(getenv("SVDSS_BAM_DEVICE") && atoi(getenv("SVDSS_BAM_DEVICE")) == 0);
      if (dev_bam) {
        std::string herr;
        if (!bam_header_probe(o.bam, n_ref_hdr, bam_skip, herr, &ref_names)) die("cannot read " + o.bam + ": " + herr);
        bam_cuts = plan_bam_regions(o.bam, n_dev, bam_skip);
      }
      if (dev_bam && !(getenv("SVDSS_CALL_STORE") && atoi(getenv("SVDSS_CALL_STORE")) == 0)) {
        // (up to SVDSS_CALL_STORE_GB per GPU, default 160: a 30x human sample is ~50 GB; more than fits: the file is read again,
        // as before.  Expected size: the bases of the file, two per byte, + names and CIGARs -- at most ~2.5 x a well-compressed BAM)
        const int64_t gb = getenv("SVDSS_CALL_STORE_GB") && atoll(getenv("SVDSS_CALL_STORE_GB")) > 0 ? atoll(getenv("SVDSS_CALL_STORE_GB")) : 160;
        const int64_t cap = getenv("SVDSS_CALL_STORE_MB") && atoll(getenv("SVDSS_CALL_STORE_MB")) > 0 ? atoll(getenv("SVDSS_CALL_STORE_MB")) << 20 : gb << 30;
        const size_t n_reg = bam_cuts.size() - 1;
        stores.assign(n_reg, nullptr);
        seam_stores.assign(n_reg, nullptr);
        store_alloc = std::thread([this, cap, n_reg, n_phys] {
          const auto t0 = std::chrono::steady_clock::now();
          for (size_t g = 0; g < n_reg; ++g) {
            const int64_t rsz = (int64_t)(bam_cuts[g + 1] - bam_cuts[g]);
            const int64_t initial = getenv("SVDSS_CALL_STORE_INITIAL_MB") ? atoll(getenv("SVDSS_CALL_STORE_INITIAL_MB")) << 20 : std::min(cap, rsz * 5 / 2 + ((int64_t)256 << 20));
            if (svdss_bam_store_create((int32_t)(g % (size_t)n_phys), cap, initial, &stores[g]) != SVDSS_OK) stores[g] = nullptr;
            if (g > 0 && svdss_bam_store_create((int32_t)(g % (size_t)n_phys), (int64_t)64 << 20, 0, &seam_stores[g]) != SVDSS_OK) seam_stores[g] = nullptr;
          }
          store_alloc_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
      }
    }
    // ---- parse_sfsfile (sfs.cpp:5-30)
    // (csrc/sfs_file.h: the file mapped and parsed by T threads; a file that cannot be opened leaves the map empty, as
    // the reference's ifstream does)
    (void)sfs_parse_file(o.sfs.c_str(), T, C.sfs);
    stage("reference + sfs file");
  }

  // Clusterer::align_and_extend (clusterer.cpp:56-156): pass 1 over the BAM, placement of every SFS
  void align_and_extend() {
    logmsg("info", "Placing SFSs on reference genome");
    // ---- align_and_extend (clusterer.cpp:56-156): pass 1 over the BAM
    // The inflated records of pass 1 are kept for pass 2 when they fit in memory (a second inflate of the whole file
    // otherwise): SVDSS_CALL_CACHE_GB, default 40 % of MemAvailable, at most 64 GiB.
    {
      double gb = 0;
      if (const char* e = getenv("SVDSS_CALL_CACHE_GB")) gb = atof(e);
      else {
        if (FILE* f = fopen("/proc/meminfo", "r")) {
          char line[256];
          while (fgets(line, sizeof line, f)) {
            long long kb;
            if (sscanf(line, "MemAvailable: %lld kB", &kb) == 1) { gb = 0.4 * (double)kb / (1024.0 * 1024.0); break; }
          }
          fclose(f);
        }
        if (gb > 64) gb = 64;
      }
      cache_limit = gb > 0 ? (size_t)(gb * 1024.0 * 1024.0 * 1024.0) : 0;
      if (cache_limit == 0) cache_ok = false;
    }
    {
      // Records handled on the GPU (csrc/bam_device.hip, svdss_bam_select_run; the default when there is a GPU and the
      // input is a regular file): only the primary, mapq-ok alignments of reads that HAVE SFS come back to the host --
      // the records clusterer.cpp:108-145 keeps -- instead of every inflated byte.  No record cache then: pass 2 goes
      // through the BAI index, or reads the file again through the same path with the cluster regions as the filter.
      // SVDSS_BAM_DEVICE=0: the host reader (chunks inflated on the GPU or the host, records sliced here).
      const bool dev_bam = svdss_device_count() > 0 && !(getenv("SVDSS_BAM_DEVICE") && atoi(getenv("SVDSS_BAM_DEVICE")) == 0);
      std::unique_ptr<BamReader> bam_p;
      std::unique_ptr<ShardedBamSelect> sel;
      std::vector<svdss_bam_filter_t*> filters;
      if (dev_bam) {
        if (bam_cuts.empty()) {      // (load_inputs probes the header; a caller that skipped it)
          std::string herr;
          if (!bam_header_probe(o.bam, n_ref_hdr, bam_skip, herr, &ref_names)) die("cannot read " + o.bam + ": " + herr);
          bam_cuts = plan_bam_regions(o.bam, 1, bam_skip);
        }
        std::string names;
        std::vector<int64_t> name_off(1, 0);
        for (const auto& kv : C.sfs) { names += kv.first; name_off.push_back((int64_t)names.size()); }
        // --gpus N: the file's regions, one per GPU (SVDSS_GPUS_OVERSUBSCRIBE puts the N shards on the GPUs there are -- the code
        // path of N devices on a one-GPU box): every region has its own scanner, batcher, feeding threads, record stream and
        // filter; the slim form of every record that passes the flag / mapq filters stays in that GPU's HBM for pass 2
        // (load_inputs took the memory).  SVDSS_CALL_STORE=0: two passes over the file.
        const int n_phys = std::max(1, svdss_device_count());
        const size_t n_reg = bam_cuts.size() - 1;
        if (store_alloc.joinable()) store_alloc.join();
        bool all_stores = !stores.empty();
        for (svdss_bam_store_t* st : stores) all_stores = all_stores && st != nullptr;
        for (size_t g = 1; g < seam_stores.size(); ++g) all_stores = all_stores && seam_stores[g] != nullptr;
        if (!all_stores) {
          for (svdss_bam_store_t* st : stores) svdss_bam_store_free(st);
          for (svdss_bam_store_t* st : seam_stores) svdss_bam_store_free(st);
          stores.clear(); seam_stores.clear();
        }
        std::vector<ShardedBamSelect::Shard> shards;
        for (size_t g = 0; g < n_reg; ++g) {
          svdss_bam_filter_t* f = nullptr;
          check(svdss_bam_filter_create((int32_t)(g % (size_t)n_phys), (int32_t)std::min<unsigned>(o.min_mapq, 256u), n_ref_hdr, names.data(), name_off.data(),
                                        (int64_t)name_off.size() - 1, nullptr, nullptr, nullptr, 0, &f), "svdss_bam_filter_create");
          filters.push_back(f);
          ShardedBamSelect::Shard sh;
          sh.filter = f; sh.device = (int)(g % (size_t)n_phys);
          sh.store = stores.empty() ? nullptr : stores[g];
          sh.seam_store = seam_stores.empty() ? nullptr : seam_stores[g];
          shards.push_back(sh);
        }
        sel.reset(new ShardedBamSelect(o.bam, shards, n_ref_hdr, bam_skip, bam_feeders(), bam_batch_bytes(), bam_cuts));
        cache_ok = false;
        dev_pass = true;
      } else {
        bam_p.reset(new BamReader(o.bam));
        // (--gpus N: the chunks of pass 1 are inflated on all N GPUs in turn, as `search` does)
        svdss_enable_gpu_inflate(*bam_p, 0, std::max(1, std::min(o.gpus, svdss_device_count())));
        if (!bam_p->ok() || !bam_p->read_header()) die("cannot read " + o.bam + ": " + bam_p->error());
        ref_names = bam_p->ref_names();
      }
      // the next record pass 1 looks at: 1 = record, 0 = end of file (errors end the run)
      double pass1_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pass1_wait_s = 0, pass1_join_s = 0;
      uint64_t pass1_batches = 0;
      const auto pass1_t0 = std::chrono::steady_clock::now();
      std::unique_ptr<SelectedBatch> sel_cur;
      size_t sel_k = 0;
      const int bsize = std::max(T, (10000 / T) * T);   // config.hpp:69, config.cpp:106
      std::vector<std::vector<ESFS>> per_thread((size_t)T);
      std::vector<std::vector<Clip>> per_thread_clips((size_t)T);
      // the chromosomes in BAM header order on the GPU, for the placement kernel (SVDSS_PLACE_HOST=1: host code instead;
      // --clipped: host code as well -- the kernel reports how many SFS stay unplaced, not next to which soft clip).
      // Waiting for the FASTA's thread and the upload happen where the first batch is placed (the worker below, one at a
      // time): until then this thread goes on taking the records the device path selects -- a pass over a million reads
      // keeps a dozen thousand of them, and its feeders used to stand still behind a full queue while the reference
      // was read (0.7 s of a 1.2 s pass at GRCh38 lengths).
      svdss_ref_t* dref = nullptr;
      std::vector<int32_t> tid_map(ref_names.size(), -1);
      bool ref_set_up = false;
      auto set_up_reference = [this, &dref, &tid_map, &ref_set_up]() {
        if (ref_set_up) return;
        ref_set_up = true;
        reference_ready();
        if (!getenv("SVDSS_PLACE_HOST") && !o.clipped) {
          std::vector<const uint8_t*> parts;
          std::vector<int64_t> lens;
          for (size_t t = 0; t < ref_names.size(); ++t) {
            auto it = C.chrom_seqs.find(ref_names[t]);
            if (it == C.chrom_seqs.end()) continue;
            tid_map[t] = (int32_t)parts.size();
            parts.push_back((const uint8_t*)it->second.data());
            lens.push_back((int64_t)it->second.size());
          }
          check(svdss_ref_upload_parts(parts.data(), lens.data(), (int32_t)parts.size(), 0, &dref), "svdss_ref_upload_parts");
        }
      };
      // two batches: the next one is read (inflate + slicing, this thread) while the T slices of the previous one run
      std::vector<BamRecord> batches[2];
      std::thread worker;
      int cur = 0;
      std::string qname;
      bool eof = false;
      uint64_t seen_chunk = ~0ull;
      std::shared_ptr<BamReader::Bytes> cur_chunk;
      while (!eof) {
        std::vector<BamRecord>& batch = batches[cur];
        batch.clear();
        while ((int)batch.size() < bsize) {
          // records are located in the inflated chunks and only decoded if the read has SFS at all
          BamReader::RawView rr;
          if (sel) {
            while (!sel_cur || sel_k + 1 >= sel_cur->off.size()) {
              const auto tw0 = std::chrono::steady_clock::now();
              sel_cur = sel->next();
              pass1_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
              sel_k = 0;
              if (!sel_cur) break;
              n_records_seen += sel_cur->n_records;
              for (int k = 0; k < 8; ++k) pass1_stage[k] += sel_cur->stage_s[k];
              pass1_stage[7] += sel_cur->inflate_kernel_s; ++pass1_batches;
            }
            if (!sel_cur) {
              if (!sel->error().empty()) { if (worker.joinable()) worker.join(); die("error reading " + o.bam + ": " + sel->error()); }
              eof = true;
              break;
            }
            const size_t at = (size_t)sel_cur->off[sel_k], end = (size_t)sel_cur->off[sel_k + 1];
            ++sel_k;
            if (!view_of_record(sel_cur->bytes.data() + at, end - at, rr, sel_cur->slim)) { if (worker.joinable()) worker.join(); die("error reading " + o.bam + ": corrupt record"); }
          } else {
          BamReader& bam = *bam_p;
          const int rc = bam.next_view(rr);
          if (rc == 0) { eof = true; break; }
          if (rc < 0) { if (worker.joinable()) worker.join(); die("error reading " + o.bam + ": " + bam.error()); }
          if (bam.chunk_id() != seen_chunk) {
            seen_chunk = bam.chunk_id();
            cur_chunk = bam.chunk();
            if (cache_ok) {
              cache_bytes += cur_chunk->size();
              if (cache_bytes > cache_limit) {   // too big to keep: pass 2 reads the file again
                cache_ok = false;
                cache_views.clear(); cache_views.shrink_to_fit();
                cache_chunks.clear(); cache_chunks.shrink_to_fit();
              } else cache_chunks.push_back(cur_chunk);
            }
          }
          }
          if (rr.flag & (4 | 2048 | 256)) continue;   // clusterer.cpp:118-122
          if ((unsigned)rr.mapq < o.min_mapq) continue;
          if (cache_ok) cache_views.push_back(rr);    // every record pass 2 looks at (same filters, clusterer.cpp:535-540)
          qname.assign((const char*)rr.name(), rr.l_name ? rr.l_name - 1 : 0);
          if (C.sfs.find(qname) == C.sfs.end()) continue;
          BamRecord r;
          BamReader::materialize(rr, r, false);   // (placement reads position, CIGAR and name: not the bases)
          batch.push_back(std::move(r));
        }
        {
          const auto tj0 = std::chrono::steady_clock::now();
          if (worker.joinable()) worker.join();   // batches are extended in order (per-thread output order, clusterer.cpp:129-141)
          pass1_join_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tj0).count();
        }
        // the T slices of the reference's OpenMP loop (clusterer.cpp:129-141), one worker each: the same records in
        // the same order per slice
        worker = std::thread([this, &per_thread, &per_thread_clips, &batch, &dref, &tid_map, &set_up_reference]() {
          set_up_reference();
          if (dref) {
            // placement on the GPU (csrc/place.hip: one lane per alignment); the results go to the T per-thread lists in
            // the order the reference's slices would have produced them (record n belongs to slice n % T)
            std::vector<int32_t> tid, pos, sq, sl, cnt;
            std::vector<uint32_t> cig;
            std::vector<int64_t> cig_off(1, 0), sfs_off(1, 0);
            std::vector<const std::vector<RawSFS>*> lists;
            for (const BamRecord& r : batch) {
              // (a negative position on a record that claims to be mapped: not placed -- the walk indexes the chromosome with it)
              const bool known = r.tid >= 0 && r.tid < (int)ref_names.size() && tid_map[(size_t)r.tid] >= 0 && r.pos >= 0;
              tid.push_back(known ? tid_map[(size_t)r.tid] : -1);
              pos.push_back(r.pos);
              cig.insert(cig.end(), r.cigar.begin(), r.cigar.end());
              cig_off.push_back((int64_t)cig.size());
              const std::vector<RawSFS>& v = C.sfs.at(r.qname);
              lists.push_back(&v);
              if (known) for (const RawSFS& x : v) { sq.push_back(x.qs); sl.push_back(x.l); }
              sfs_off.push_back((int64_t)sq.size());
            }
            cnt.resize(batch.size());
            std::vector<int32_t> out(5 * std::max<size_t>(1, sq.size()));
            int64_t st[4];
            check(svdss_place_sfs_batch(dref, tid.data(), pos.data(), cig.data(), cig_off.data(), sq.data(), sl.data(),
                                        sfs_off.data(), (int64_t)batch.size(), cnt.data(), out.data(), st),
                  "svdss_place_sfs_batch");
            C.unplaced += st[0]; C.s_unplaced += st[1]; C.e_unplaced += st[2]; C.unknown += st[3];
            for (int t = 0; t < T; ++t)
              for (size_t n = (size_t)t; n < batch.size(); n += (size_t)T) {
                const BamRecord& r = batch[n];
                if (tid[n] < 0) continue;
                const int32_t* o = out.data() + 5 * sfs_off[n];
                for (int32_t j = 0; j < cnt[n]; ++j, o += 5)
                  per_thread[(size_t)t].push_back(ESFS{ref_names[(size_t)r.tid], r.qname, o[0], o[1], o[2], o[3],
                                                       (*lists[n])[(size_t)o[4]].htag});
              }
            return;
          }
          auto slice = [&](int t) {
            for (size_t n = (size_t)t; n < batch.size(); n += (size_t)T) {
              const BamRecord& r = batch[n];
              if (r.tid < 0 || r.tid >= (int)ref_names.size() || r.pos < 0) continue;
              extend_alignment(C, r, ref_names[(size_t)r.tid], per_thread[(size_t)t],
                               C.o.clipped ? &per_thread_clips[(size_t)t] : nullptr);
            }
          };
          if (T == 1 || batch.size() < 64) { for (int t = 0; t < T; ++t) slice(t); }
          else {
            std::vector<std::thread> pool;
            for (int t = 1; t < T; ++t) pool.emplace_back(slice, t);
            slice(0);
            for (std::thread& th : pool) th.join();
          }
        });
        cur ^= 1;
      }
      if (worker.joinable()) worker.join();
      set_up_reference();   // (an input without a single batch: the later stages still want the chromosomes)
      if (sel) {
        region_batches.clear(); region_seam.clear();
        for (size_t g = 0; g < sel->n_regions(); ++g) { region_batches.push_back(sel->region_batches(g)); region_seam.push_back(sel->region_has_seam(g) ? 1 : 0); }
        if (sel->n_regions() > 1)
          logmsg("debug", std::to_string(sel->n_regions()) + " regions of the file, one per GPU: " + std::to_string(sel->seams_run()) + " seam(s) run, " +
                              std::to_string(sel->regions_run_again()) + " region(s) run again");
      }
      if (sel && pass1_batches) {
        char buf[480];
        snprintf(buf, sizeof buf, "pass 1 on the device: %llu batches, %llu records; feeder seconds summed: upload+inflate+crc+walk %.3f, turn wait %.3f, turn %.3f, "
                 "select+scans%s %.3f, records down %.3f (inflate kernels %.3f)", (unsigned long long)pass1_batches, (unsigned long long)n_records_seen, pass1_stage[0],
                 pass1_stage[1], pass1_stage[2], !stores.empty() ? "+store" : "", pass1_stage[3], pass1_stage[6], pass1_stage[7]);
        logmsg("debug", buf);
        snprintf(buf, sizeof buf, "pass 1, this thread: %.3f s in all, %.3f s waiting for the device's batches, %.3f s waiting for the placement of the batch before; "
                 "the record store's memory took %.3f s; the batcher waited %.3f s for the file's loaders, %.3f s for the feeding threads",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - pass1_t0).count(), pass1_wait_s, pass1_join_s, store_alloc_s, sel->waited_for_file(), sel->waited_for_feeders());
        logmsg("debug", buf);
      }
      svdss_ref_free(dref);
      sel.reset();
      for (svdss_bam_filter_t* f : filters) svdss_bam_filter_free(f);
      for (int t = 0; t < T; ++t) extended.insert(extended.end(), per_thread[(size_t)t].begin(), per_thread[(size_t)t].end());
      for (int t = T; t-- > 0;)   // each thread's list goes in front of the others' (clusterer.cpp:24)
        clips.insert(clips.end(), per_thread_clips[(size_t)t].begin(), per_thread_clips[(size_t)t].end());
    }
    logmsg("info", std::to_string(C.unplaced.load()) + "/" + std::to_string(C.s_unplaced.load()) + "/" + std::to_string(C.e_unplaced.load()) +
                       " unplaced SFSs. " + std::to_string(C.unknown.load()) + " erroneus SFSs. " + std::to_string(clips.size()) +
                       " clipped SFSs.");   // clusterer.cpp:26-27
    stage("pass 1: placement");
  }

  // Clusterer::cluster_by_proximity (clusterer.cpp:407-474)
  void cluster_by_proximity() {
    // ---- cluster_by_proximity (clusterer.cpp:407-474)
    if (!extended.empty()) {
      std::sort(extended.begin(), extended.end());
      int maxlen = 0;
      for (const ESFS& s : extended) maxlen = std::max(maxlen, s.re - s.rs);
      const int dist = (int)(maxlen * 1.1);
      std::vector<std::pair<int, int>> intervals;
      int prev_i = 0, prev_e = extended[0].re;
      std::string prev_chrom = extended[0].chrom;
      for (size_t i = 1; i < extended.size(); ++i) {
        const ESFS& s = extended[i];
        if (s.chrom != prev_chrom) {
          prev_chrom = s.chrom; intervals.emplace_back(prev_i, (int)i - 1); prev_i = (int)i; prev_e = s.re; continue;
        }
        if (s.rs - prev_e > dist) { intervals.emplace_back(prev_i, (int)i - 1); prev_e = s.re; prev_i = (int)i; }
      }
      intervals.emplace_back(prev_i, (int)extended.size() - 1);
      std::vector<std::map<std::pair<int, int>, std::vector<ESFS>>> per_thread((size_t)T);
      for (size_t i = 0; i < intervals.size(); ++i) {
        auto& mp = per_thread[i % (size_t)T];   // schedule(static, 1)
        int j = intervals[i].first, low = extended[(size_t)j].rs, high = extended[(size_t)j].re, last_j = j;
        ++j;
        for (; j <= intervals[i].second; ++j) {
          const ESFS& s = extended[(size_t)j];
          if (s.rs <= high) { low = std::min(low, s.rs); high = std::max(high, s.re); }
          else {
            for (int k = last_j; k < j; ++k) mp[{low, high}].push_back(extended[(size_t)k]);
            low = s.rs; high = s.re; last_j = j;
          }
        }
        for (int k = last_j; k <= intervals[i].second; ++k) mp[{low, high}].push_back(extended[(size_t)k]);
      }
      for (int t = 0; t < T; ++t)
        for (auto& kv : per_thread[(size_t)t]) {
          Cluster c;
          c.sfss = kv.second;
          c.chrom = c.sfss[0].chrom;
          clusters.push_back(std::move(c));
        }
    }
    stage("cluster_by_proximity");
  }

  // Clusterer::fill_clusters (clusterer.cpp:477-610): pass 2 over the BAM
  void fill_clusters() {
    // ---- fill_clusters (clusterer.cpp:477-610): pass 2 over the BAM
    {
      std::vector<std::set<std::string>> reads(clusters.size());
      std::vector<int> min_s(clusters.size()), max_e(clusters.size());
      std::vector<char> live(clusters.size(), 0);
      std::vector<std::vector<int>> cov(clusters.size(), std::vector<int>(3, 0));
      std::map<std::string, std::vector<size_t>> by_chrom;   // cluster indices per chrom, sorted by region start
      for (size_t i = 0; i < clusters.size(); ++i) {
        int mn = std::numeric_limits<int>::max(), mx = 0;
        for (const ESFS& s : clusters[i].sfss) { mn = std::min(mn, s.rs); mx = std::max(mx, s.re); reads[i].insert(s.qname); }
        min_s[i] = mn; max_e[i] = mx;
        if (reads[i].size() < (size_t)o.min_cluster_weight) { ++C.small; continue; }
        clusters[i].s = mn; clusters[i].e = mx;
        live[i] = 1;
        by_chrom[clusters[i].chrom].push_back(i);
      }
      std::map<std::string, std::vector<int>> run_max_end;   // per chrom: running maximum of the region ends, same order
      for (auto& kv : by_chrom) {
        std::sort(kv.second.begin(), kv.second.end(), [&](size_t a, size_t b) { return min_s[a] < min_s[b]; });
        std::vector<int>& rm = run_max_end[kv.first];
        int m = 0;
        for (size_t ci : kv.second) { m = std::max(m, max_e[ci]); rm.push_back(m); }
      }
      // per reference id: the clusters of that chromosome (sorted by start) and the running maximum of their ends
      std::vector<const std::vector<size_t>*> tid_clusters(ref_names.size(), nullptr);
      std::vector<const std::vector<int>*> tid_run_max(ref_names.size(), nullptr);
      for (size_t t = 0; t < ref_names.size(); ++t) {
        auto it = by_chrom.find(ref_names[t]);
        if (it == by_chrom.end()) continue;
        tid_clusters[t] = &it->second;
        tid_run_max[t] = &run_max_end[it->first];
      }
      static const char NT16[] = "=ACMGRSVTWYHKDBN";
      std::string qname;
      // Records stay in their raw form: the end position and the two query positions the reference reads off the
      // aligned-pairs vector (bam.cpp:92-134, clusterer.cpp:555-580) are functions of the CIGAR blocks alone, and only
      // the bases of the extracted sub-read are decoded.
      // what one alignment does to one cluster it overlaps; applied in BAM order (sequentially, or collected by worker
      // threads over contiguous record ranges and applied range after range)
      struct Ev { size_t ci; int hp; bool in_reads, unext; std::string name, sub; };
      auto apply = [&](Ev& e) {
        if (e.hp >= 0 && e.hp < 3) ++cov[e.ci][(size_t)e.hp];
        clusters[e.ci].reads.emplace_back(e.in_reads ? 1 : 0, e.hp == 0 ? 3 : e.hp);
        if (!e.in_reads) return;
        if (e.unext) ++C.unextended;
        else clusters[e.ci].subreads.push_back(SubRead{std::move(e.name), std::move(e.sub), e.hp});
      };
      auto process = [&](const BamReader::RawView& rr, std::string& qname, auto&& sink) {
        if (rr.tid < 0 || rr.tid >= (int)ref_names.size() || !tid_clusters[(size_t)rr.tid]) return;
        if (rr.flag & (4 | 2048 | 256)) return;        // clusterer.cpp:535-540: such a record touches no cluster
        if ((unsigned)rr.mapq < o.min_mapq) return;
        const uint8_t* cg = rr.name() + rr.l_name;
        auto cig = [&](uint32_t i) { uint32_t c; memcpy(&c, cg + 4u * i, 4); return c; };
        int32_t ref_len = 0;
        for (uint32_t i = 0; i < rr.n_cigar; ++i) {
          const uint32_t c = cig(i), op = c & 0xf;
          if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += (int32_t)(c >> 4);
        }
        const int a_beg = rr.pos, a_end = rr.pos + (ref_len ? ref_len : 1);   // bam_endpos
        const std::vector<size_t>& cl_ids = *tid_clusters[(size_t)rr.tid];
        // clusters before `first` end at or before the alignment's start: none of them can overlap it
        const std::vector<int>& rm = *tid_run_max[(size_t)rr.tid];
        const size_t first = (size_t)(std::upper_bound(rm.begin(), rm.end(), a_beg) - rm.begin());
        bool have_tags = false;
        int64_t hp = 0;
        for (size_t k = first; k < cl_ids.size(); ++k) {
          const size_t ci = cl_ids[k];
          // region "chrom:min_s-max_e" = 0-based half-open [min_s-1, max_e)
          const int beg0 = std::max(min_s[ci] - 1, 0), end0 = max_e[ci];
          if (beg0 >= a_end) break;   // clusters are sorted by start
          if (!(a_beg < end0 && a_end > beg0)) continue;
          if (!have_tags) {
            have_tags = true;
            BamReader::aux_int(rr.aux(), rr.l_aux, "HP", hp);
            qname.assign((const char*)rr.name(), rr.l_name ? rr.l_name - 1 : 0);
          }
          Ev ev{ci, (int)hp, false, false, std::string(), std::string()};
          if (reads[ci].find(qname) == reads[ci].end()) { sink(ev); continue; }
          ev.in_reads = true;
          // qs: query position of the last aligned (M/=/X) pair with reference position <= min_s;
          // qe: of the first one with reference position >= max_e
          int qs = -1, qe = -1, ref_pos = rr.pos, read_pos = 0;
          for (uint32_t i = 0; i < rr.n_cigar; ++i) {
            const uint32_t c = cig(i), op = c & 0xf;
            const int l = (int)(c >> 4);
            if (op == 0 || op == 7 || op == 8) {
              if (l > 0) {
                if (ref_pos <= min_s[ci]) qs = read_pos + (std::min(min_s[ci], ref_pos + l - 1) - ref_pos);
                if (qe == -1 && ref_pos + l - 1 >= max_e[ci]) qe = read_pos + (std::max(max_e[ci], ref_pos) - ref_pos);
              }
              read_pos += l; ref_pos += l;
            } else if (op == 1 || op == 4) read_pos += l;
            else if (op == 2 || op == 3) ref_pos += l;
          }
          if (qs == -1 || qe == -1) ev.unext = true;
          else {
            if (qs > rr.l_seq) die("corrupt alignment: sub-read start past the end of read " + qname);   // (std::string::substr throws in the reference)
            const int n = std::max(0, std::min(qe - qs + 1, rr.l_seq - qs));
            const uint8_t* sq = rr.seq4();
            std::string sub((size_t)n, 'N');
            for (int i = 0; i < n; ++i) { const int q = qs + i; sub[(size_t)i] = NT16[(sq[q >> 1] >> ((~q & 1) << 2)) & 0xf]; }
            ev.name = qname;
            ev.sub = std::move(sub);
          }
          sink(ev);
        }
      };
      bool from_store = false;
      if (!stores.empty()) {
        // Round 6: the records are in HBM since pass 1 (svdss_bam_store_t; --gpus N: the store of every region in its GPU's).
        // Per stored batch a kernel keeps those that overlap a (merged) cluster region; they come down slim, in file order,
        // and go through `process` on a few threads, what they do to the clusters applied batch after batch -- the order
        // of the single scan.
        struct Item { svdss_bam_store_t* st; int64_t key; size_t dev; };
        std::vector<Item> items;
        bool complete_all = true;
        int64_t n_rec_st = 0, n_bytes_st = 0;
        const int n_phys = std::max(1, svdss_device_count());
        for (size_t g = 0; g < stores.size(); ++g) {
          int32_t complete = 0;
          int64_t nr = 0, nb = 0;
          if (g > 0 && g < region_seam.size() && region_seam[g]) {
            (void)svdss_bam_store_batches(seam_stores[g], &complete, &nr, &nb);
            complete_all = complete_all && complete;
            n_rec_st += nr; n_bytes_st += nb;
            items.push_back(Item{seam_stores[g], 0, g % (size_t)n_phys});
          }
          const int64_t n_b = svdss_bam_store_batches(stores[g], &complete, &nr, &nb);
          complete_all = complete_all && complete && n_b == (g < region_batches.size() ? region_batches[g] : -1);
          n_rec_st += nr; n_bytes_st += nb;
          for (int64_t k = 0; k < n_b; ++k) items.push_back(Item{stores[g], k, g % (size_t)n_phys});
        }
        if (complete_all) {
          from_store = true;
          std::vector<int32_t> rt, rb, re;
          for (size_t t = 0; t < ref_names.size(); ++t) {
            if (!tid_clusters[t]) continue;
            int64_t cb = -1, ce = -1;
            for (size_t ci : *tid_clusters[t]) {
              const int64_t b0 = std::max(min_s[ci] - 1, 0), e0 = max_e[ci];
              if (ce >= 0 && b0 <= ce) { ce = std::max(ce, e0); continue; }
              if (ce >= 0) { rt.push_back((int32_t)t); rb.push_back((int32_t)cb); re.push_back((int32_t)ce); }
              cb = b0; ce = e0;
            }
            if (ce >= 0) { rt.push_back((int32_t)t); rb.push_back((int32_t)cb); re.push_back((int32_t)ce); }
          }
          uint64_t n_down = 0, bytes_down = 0;
          if (!rt.empty() && !items.empty()) {
            std::vector<svdss_bam_filter_t*> rf((size_t)std::min<size_t>((size_t)n_phys, stores.size()), nullptr);   // the regions, on every GPU that holds a store
            for (size_t d = 0; d < rf.size(); ++d)
              check(svdss_bam_filter_create((int32_t)d, (int32_t)std::min<unsigned>(o.min_mapq, 256u), (int32_t)ref_names.size(), nullptr, nullptr, 0, rt.data(), rb.data(),
                                            re.data(), (int64_t)rt.size(), &rf[d]), "svdss_bam_filter_create");
            std::vector<std::vector<Ev>> evs(items.size());
            std::atomic<size_t> next(0);
            std::atomic<uint64_t> a_down(0), a_bytes(0);
            std::mutex err_m;
            std::string err;
            auto work = [&]() {
              std::vector<svdss_bam_batch_t*> batch(rf.size(), nullptr);      // (a batch object belongs to one device)
              std::string nm;
              BamReader::RawView rr;
              for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= items.size()) break;
                const Item& it = items[k];
                const int rc = svdss_bam_store_select(it.st, it.key, rf[it.dev], &batch[it.dev]);
                if (rc != SVDSS_OK) { std::lock_guard<std::mutex> lk(err_m); if (err.empty()) err = std::string(svdss_strerror(rc)) + " " + svdss_last_hip_error(); break; }
                svdss_bam_selection_t r;
                (void)svdss_bam_batch_selection(batch[it.dev], &r);
                auto sink = [&](Ev& e) { evs[k].push_back(std::move(e)); };
                for (int64_t i = 0; i < r.n_selected; ++i) {
                  if (!view_of_record(r.bytes + r.rec_off[i], (size_t)(r.rec_off[i + 1] - r.rec_off[i]), rr, true)) {
                    std::lock_guard<std::mutex> lk(err_m); if (err.empty()) err = "corrupt record in the store"; break;
                  }
                  process(rr, nm, sink);
                }
                a_down += (uint64_t)r.n_selected; a_bytes += (uint64_t)r.n_bytes;
              }
              for (svdss_bam_batch_t* b : batch) if (b) svdss_bam_batch_free(b);
            };
            const size_t Wt = std::max<size_t>(1, std::min<size_t>({(size_t)effective_cpus(), (size_t)8, items.size()}));
            std::vector<std::thread> pool;
            for (size_t w = 1; w < Wt; ++w) pool.emplace_back(work);
            work();
            for (std::thread& th : pool) th.join();
            for (svdss_bam_filter_t* f : rf) svdss_bam_filter_free(f);
            if (!err.empty()) die("pass 2 from the record store: " + err);
            for (size_t k = 0; k < items.size(); ++k)
              for (Ev& e : evs[k]) apply(e);
            n_down = a_down.load(); bytes_down = a_bytes.load();
          }
          logmsg("debug", "pass 2 from the records kept in HBM: " + std::to_string(n_rec_st) + " records (" + std::to_string(n_bytes_st >> 20) + " MB) in " +
                              std::to_string(items.size()) + " batches" + (stores.size() > 1 ? " of " + std::to_string(stores.size()) + " stores" : "") + ", " +
                              std::to_string(rt.size()) + " regions, " + std::to_string(n_down) + " records (" + std::to_string(bytes_down >> 20) + " MB) came down");
        } else
          logmsg("debug", "the record store is incomplete (" + std::to_string(n_bytes_st >> 20) + " MB kept): pass 2 reads the file again");
        // (tens of GB of HBM stay allocated until the run ends: memory handed back is cleared by the driver beside whatever
        // runs next -- here the POA batches -- and `call` has room to spare: it holds no index)
      }
      if (from_store) {
      } else if (cache_ok) {
        stage("pass 2: setup");
        const size_t n = cache_views.size();
        const size_t W = std::max<size_t>(1, std::min<size_t>({(size_t)effective_cpus(), (size_t)32, n / 4096 + 1}));
        std::vector<std::vector<Ev>> evs(W);
        auto range = [&](size_t w) {
          std::string nm;
          auto sink = [&](Ev& e) { evs[w].push_back(std::move(e)); };
          for (size_t i = n * w / W; i < n * (w + 1) / W; ++i) process(cache_views[i], nm, sink);
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < W; ++w) pool.emplace_back(range, w);
        range(0);
        for (std::thread& th : pool) th.join();
        stage("pass 2: scan");
        for (size_t w = 0; w < W; ++w)
          for (Ev& e : evs[w]) apply(e);
        stage("pass 2: apply");
        // (gigabytes of inflated records: released while the DP batches run)
        cache_release = std::thread([v = std::move(cache_views), c = std::move(cache_chunks)]() mutable { v.clear(); c.clear(); });
      } else {
        // the records of pass 1 did not fit in memory.  With a BAI index beside the file (what the reference requires:
        // sam_index_load + one sam_itr_querys per cluster, clusterer.cpp:495-527) only the file chunks around the
        // clusters are read -- all regions turned into one sorted, merged chunk list, read once in file order, which
        // visits the same records in the same order as the full scan below does among those that touch a cluster
        BaiIndex bai;
        bool have_bai = false;
        if (!getenv("SVDSS_CALL_NO_BAI")) {
          // (x.bam.bai, x.bai, x.bam.csi, x.csi: the names htslib's sam_index_load looks for)
          const std::string stem = o.bam.size() > 4 && o.bam.compare(o.bam.size() - 4, 4, ".bam") == 0 ? o.bam.substr(0, o.bam.size() - 4) : std::string();
          have_bai = bai.load(o.bam + ".bai") || (!stem.empty() && bai.load(stem + ".bai")) || bai.load(o.bam + ".csi") ||
                     (!stem.empty() && bai.load(stem + ".csi"));
          if (have_bai && bai.refs.size() != ref_names.size()) have_bai = false;
        }
        std::vector<std::pair<uint64_t, uint64_t>> chunks;
        size_t n_regions = 0;
        if (have_bai) {
          for (size_t t = 0; t < ref_names.size(); ++t) {
            if (!tid_clusters[t]) continue;
            int64_t rb = -1, re = -1;   // current merged region
            for (size_t ci : *tid_clusters[t]) {
              const int64_t b0 = std::max(min_s[ci] - 1, 0), e0 = max_e[ci];
              if (re >= 0 && b0 <= re) { re = std::max(re, e0); continue; }
              if (re >= 0) { bai.query((int)t, rb, re, chunks); ++n_regions; }
              rb = b0; re = e0;
            }
            if (re >= 0) { bai.query((int)t, rb, re, chunks); ++n_regions; }
          }
          BaiIndex::merge(chunks);
          // The index names the file chunks around the clusters; host threads inflate them (~0.3 GB/s each; second session
          // of round 5: all of them, until then one).  When the chunks are a large part of the file, reading ALL of it
          // through the device path (tens of GB/s, only the overlapping records come back) is quicker: beyond 8 % of the
          // file per thread that scans.  SVDSS_CALL_PASS2 = bai | device overrides the estimate.
          if (dev_pass) {
            uint64_t chunk_bytes = 0;
            for (const auto& ch : chunks) chunk_bytes += (ch.second >> 16) - (ch.first >> 16) + 65536;
            struct stat st;
            const uint64_t file_bytes = stat(o.bam.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0;
            const char* p2 = getenv("SVDSS_CALL_PASS2");
            const double share = std::min(0.6, 0.08 * (double)pass2_threads(chunks.size()));
            if (p2 && !strcmp(p2, "device")) have_bai = false;
            else if (!(p2 && !strcmp(p2, "bai")) && file_bytes && (double)chunk_bytes > share * (double)file_bytes) have_bai = false;
          }
        }
        if (have_bai) {
          logmsg("debug", std::string("pass 2 through the ") + (bai.csi ? "CSI" : "BAI") + " index: " + std::to_string(n_regions) + " regions, " + std::to_string(chunks.size()) +
                              " file chunks");
          const size_t Wt = pass2_threads(chunks.size());
          if (Wt <= 1) {
            const std::string e = bam_scan_chunks(o.bam, chunks, [&](const BamReader::RawView& rr) { process(rr, qname, apply); });
            if (!e.empty()) die("error reading " + o.bam + ": " + e);
          } else {
            // Runs of consecutive chunks of about the same size in the file, a few per thread, taken in turn: every run is
            // scanned by one thread with a file handle and an inflater of its own (bam_scan_chunks), what its records do
            // to the clusters is collected and applied run after run -- the order of the single scan.
            std::vector<std::pair<size_t, size_t>> runs;
            {
              uint64_t total = 0;
              for (const auto& ch : chunks) total += (ch.second >> 16) - (ch.first >> 16) + 65536;
              const uint64_t per = std::max<uint64_t>(1, total / (4 * Wt));
              size_t a = 0;
              uint64_t acc = 0;
              for (size_t k = 0; k < chunks.size(); ++k) {
                acc += (chunks[k].second >> 16) - (chunks[k].first >> 16) + 65536;
                if (acc >= per || k + 1 == chunks.size()) { runs.emplace_back(a, k + 1); a = k + 1; acc = 0; }
              }
            }
            std::vector<std::vector<Ev>> evs(runs.size());
            std::vector<std::string> errs(runs.size());
            std::atomic<size_t> next(0);
            auto work = [&]() {
              std::string nm;
              for (;;) {
                const size_t g = next.fetch_add(1);
                if (g >= runs.size()) return;
                const std::vector<std::pair<uint64_t, uint64_t>> sub(chunks.begin() + (ptrdiff_t)runs[g].first, chunks.begin() + (ptrdiff_t)runs[g].second);
                auto sink = [&](Ev& e) { evs[g].push_back(std::move(e)); };
                errs[g] = bam_scan_chunks(o.bam, sub, [&](const BamReader::RawView& rr) { process(rr, nm, sink); });
              }
            };
            std::vector<std::thread> pool;
            for (size_t w = 1; w < Wt; ++w) pool.emplace_back(work);
            work();
            for (std::thread& th : pool) th.join();
            for (size_t g = 0; g < runs.size(); ++g) {
              if (!errs[g].empty()) die("error reading " + o.bam + ": " + errs[g]);
              for (Ev& e : evs[g]) apply(e);
            }
          }
        } else if (dev_pass) {
          // no index: the file again through the device path, the (merged) cluster regions as the filter -- the records
          // that overlap a cluster come back, in file order
          std::vector<int32_t> rt, rb, re;
          for (size_t t = 0; t < ref_names.size(); ++t) {
            if (!tid_clusters[t]) continue;
            int64_t cb = -1, ce = -1;
            for (size_t ci : *tid_clusters[t]) {
              const int64_t b0 = std::max(min_s[ci] - 1, 0), e0 = max_e[ci];
              if (ce >= 0 && b0 <= ce) { ce = std::max(ce, e0); continue; }
              if (ce >= 0) { rt.push_back((int32_t)t); rb.push_back((int32_t)cb); re.push_back((int32_t)ce); }
              cb = b0; ce = e0;
            }
            if (ce >= 0) { rt.push_back((int32_t)t); rb.push_back((int32_t)cb); re.push_back((int32_t)ce); }
          }
          if (!rt.empty()) {
            const int n_phys = std::max(1, svdss_device_count());
            if (bam_cuts.empty()) bam_cuts = plan_bam_regions(o.bam, 1, bam_skip);
            std::vector<svdss_bam_filter_t*> filters;
            std::vector<ShardedBamSelect::Shard> shards;
            for (size_t g = 0; g + 1 < bam_cuts.size(); ++g) {
              svdss_bam_filter_t* f = nullptr;
              check(svdss_bam_filter_create((int32_t)(g % (size_t)n_phys), (int32_t)std::min<unsigned>(o.min_mapq, 256u), (int32_t)ref_names.size(), nullptr, nullptr, 0,
                                            rt.data(), rb.data(), re.data(), (int64_t)rt.size(), &f), "svdss_bam_filter_create");
              filters.push_back(f);
              ShardedBamSelect::Shard sh;
              sh.filter = f; sh.device = (int)(g % (size_t)n_phys);
              shards.push_back(sh);
            }
            {
              ShardedBamSelect sel(o.bam, shards, (int32_t)ref_names.size(), bam_skip, bam_feeders(), bam_batch_bytes(), bam_cuts);
              BamReader::RawView rr;
              while (std::unique_ptr<SelectedBatch> sb = sel.next())
                for (size_t k = 0; k + 1 < sb->off.size(); ++k) {
                  if (!view_of_record(sb->bytes.data() + sb->off[k], (size_t)(sb->off[k + 1] - sb->off[k]), rr, sb->slim)) die("error reading " + o.bam + ": corrupt record");
                  process(rr, qname, apply);
                }
              if (!sel.error().empty()) die("error reading " + o.bam + ": " + sel.error());
            }
            for (svdss_bam_filter_t* f : filters) svdss_bam_filter_free(f);
          }
        } else {
          BamReader bam(o.bam);
          svdss_enable_gpu_inflate(bam);
          if (!bam.ok() || !bam.read_header()) die("cannot read " + o.bam + ": " + bam.error());
          BamReader::RawView rr;   // zero-copy: the record is used where it was inflated, within this iteration only
          int rc;
          while ((rc = bam.next_view(rr)) > 0) process(rr, qname, apply);
          if (rc < 0) die("error reading " + o.bam + ": " + bam.error());
        }
      }
      for (size_t i = 0; i < clusters.size(); ++i) {
        if (!live[i]) { clusters[i].reads.clear(); clusters[i].subreads.clear(); continue; }
        if (clusters[i].size() >= (size_t)o.min_cluster_weight) {
          clusters[i].cov0 = cov[i][0]; clusters[i].cov1 = cov[i][1]; clusters[i].cov2 = cov[i][2];
          clusters[i].cov = cov[i][0] + cov[i][1] + cov[i][2];
        } else ++C.small2;
      }
    }
    stage("pass 2: fill_clusters");
  }

  // Clusterer::store_clusters (clusterer.cpp:613-626), then Caller::split_cluster (caller.cpp:100-255)
  void store_and_split() {
    // ---- store_clusters (clusterer.cpp:613-626): every cluster, also the filtered ones (their coordinates are
    // uninitialised in the reference; 0 here)
    if (!o.clusters.empty()) {
      logmsg("info", "Storing clusters to " + o.clusters);
      FILE* f = fopen(o.clusters.c_str(), "w");
      if (!f) die("cannot write " + o.clusters);
      std::string line;
      for (const Cluster& c : clusters) {
        line = c.chrom + ":" + std::to_string(c.s + 1) + "-" + std::to_string(c.e + 1) + "\t" + std::to_string(c.size());
        for (const SubRead& sr : c.subreads) { line += "\t"; line += sr.name; line += ":"; line += sr.seq; }
        line += "\n";
        fwrite(line.data(), 1, line.size(), f);
      }
      fclose(f);
    }
    logmsg("info", "Calling SVs from " + std::to_string(clusters.size()) + " clusters..");
    // (SVDSS_GPUS_OVERSUBSCRIBE: more shards than GPUs, shard g on GPU g % count -- exercises the sharding on a one-GPU box)
    n_dev = std::max(1, svdss_device_count());
    G = std::max(1, getenv("SVDSS_GPUS_OVERSUBSCRIBE") ? o.gpus : std::min(o.gpus, n_dev));
    // ---- pcall (caller.cpp:311-406): split, then the three GPU batches
    for (size_t i = 0; i < clusters.size(); ++i) {
      if (clusters[i].size() < (size_t)o.min_cluster_weight) continue;
      for (Cluster& cl : split_cluster(clusters[i], o.useht, o.min_ratio)) subs.push_back(Sub{i, std::move(cl)});
    }
    stage("split_cluster");
  }

  // Caller::run_poa for every sub-cluster (caller.cpp:257-308), one GPU batch per device
  void run_poa() {
    consensus.assign(subs.size(), std::string());
    if (!subs.empty()) {
      // the sub-reads of every sub-cluster back to back in abPOA's alphabet: offsets first, then the bytes by T
      // threads (a 30x human sample is ~1 GB of sub-reads: one byte at a time on one thread took seconds)
      std::vector<int64_t> seq_off(1, 0), cl_off(1, 0);
      std::vector<const std::string*> srcs;
      for (const Sub& s : subs) {
        for (const SubRead& sr : s.cl.subreads) {
          srcs.push_back(&sr.seq);
          seq_off.push_back(seq_off.back() + (int64_t)sr.seq.size());
        }
        cl_off.push_back((int64_t)seq_off.size() - 1);
      }
      // (not a vector: its zero-fill of ~1 GB on one thread was a fifth of this stage at 30x; the threads below write every byte)
      struct Flat { std::unique_ptr<uint8_t[]> p; uint8_t* data() const { return p.get(); } uint8_t* begin() const { return p.get(); } } flat{std::unique_ptr<uint8_t[]>(new uint8_t[(size_t)seq_off.back() + 1])};
      {
        uint8_t lut[256];
        for (int c = 0; c < 256; ++c) lut[c] = enc26((char)c);
        const size_t ns = srcs.size(), nt = std::min<size_t>((size_t)T, std::max<size_t>(1, ns / 256));
        auto enc = [&](size_t t) {
          for (size_t k = ns * t / nt; k < ns * (t + 1) / nt; ++k) {
            uint8_t* d = flat.data() + seq_off[k];
            const std::string& q = *srcs[k];
            for (size_t x = 0; x < q.size(); ++x) d[x] = lut[(uint8_t)q[x]];
          }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t) pool.emplace_back(enc, t);
        enc(0);
        for (std::thread& th : pool) th.join();
      }
      // --gpus G: sub-cluster k goes to GPU k % G (no exchange between the GPUs: a sub-cluster is self-contained), the
      // consensus sequences come back in sub-cluster order -- the same bytes as with one GPU
      std::vector<int64_t> lens(subs.size());
      std::vector<uint8_t> cons;
      {
        const size_t nsub = subs.size();
        std::vector<std::vector<uint8_t>> part_cons((size_t)G);
        std::vector<std::vector<int64_t>> part_lens((size_t)G);
        std::vector<std::thread> pool;
        auto run = [&](int g) {
          std::vector<uint8_t> f;
          std::vector<int64_t> so(1, 0), co(1, 0);
          for (size_t k = (size_t)g; k < nsub; k += (size_t)G) {
            for (int64_t sq = cl_off[k]; sq < cl_off[k + 1]; ++sq) {
              f.insert(f.end(), flat.begin() + seq_off[(size_t)sq], flat.begin() + seq_off[(size_t)sq + 1]);
              so.push_back((int64_t)f.size());
            }
            co.push_back((int64_t)so.size() - 1);
          }
          svdss_poa_batch_t* pb = nullptr;
          const auto tp0 = std::chrono::steady_clock::now();
          check(svdss_poa_consensus_batch(G == 1 ? flat.data() : f.data(), G == 1 ? seq_off.data() : so.data(),
                                          G == 1 ? cl_off.data() : co.data(), (int64_t)co.size() - 1, g % n_dev, &pb),
                "svdss_poa_consensus_batch");
          const auto tp1 = std::chrono::steady_clock::now();
          part_lens[(size_t)g].resize(co.size() - 1);
          part_cons[(size_t)g].resize((size_t)svdss_poa_batch_total(pb));
          check(svdss_poa_batch_fetch(pb, part_lens[(size_t)g].data(), part_cons[(size_t)g].data()), "svdss_poa_batch_fetch");
          const auto tp2 = std::chrono::steady_clock::now();
          svdss_poa_batch_free(pb);
          if (o.verbose && getenv("SVDSS_DEBUG"))
            fprintf(stderr, "[call] [debug] POA shard %d: batch %.3f s, fetch %.3f s, free %.3f s\n", g, std::chrono::duration<double>(tp1 - tp0).count(),
                    std::chrono::duration<double>(tp2 - tp1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tp2).count());
        };
        for (int g = 1; g < G; ++g) pool.emplace_back(run, g);
        run(0);
        for (std::thread& th : pool) th.join();
        std::vector<size_t> at((size_t)G, 0), idx((size_t)G, 0);
        for (size_t k = 0; k < nsub; ++k) {
          const size_t g = k % (size_t)G;
          lens[k] = part_lens[g][idx[g]++];
          cons.insert(cons.end(), part_cons[g].begin() + (long)at[g], part_cons[g].begin() + (long)(at[g] + (size_t)lens[k]));
          at[g] += (size_t)lens[k];
        }
      }
      size_t p = 0;
      for (size_t i = 0; i < subs.size(); ++i) {
        consensus[i].resize((size_t)lens[i]);
        for (int64_t k = 0; k < lens[i]; ++k) consensus[i][(size_t)k] = "ACGTN"[cons[p++]];   // caller.cpp:297
      }
    }
    stage("POA");
  }

  // ksw_extd2 of every consensus against its window, SV extraction (caller.cpp:326-406)
  void realign_and_extract() {
    if (!subs.empty()) {
      const int8_t a = 1, b = -9;   // caller.cpp:333-337
      const int8_t mat[25] = {a, b, b, b, 0, b, a, b, b, 0, b, b, a, b, 0, b, b, b, a, 0, 0, 0, 0, 0, 0};
      std::vector<uint8_t> q, t;
      std::vector<int64_t> qo(1, 0), to(1, 0);
      for (size_t i = 0; i < subs.size(); ++i) {
        const Cluster& cl = subs[i].cl;
        const std::string& cs = C.chrom_seqs[cl.chrom];
        for (char ch : consensus[i]) q.push_back(enc26(ch));
        qo.push_back((int64_t)q.size());
        for (int p = cl.s; p <= cl.e && p < (int)cs.size(); ++p) t.push_back(enc26(cs[(size_t)p]));   // caller.cpp:329
        to.push_back((int64_t)t.size());
      }
      std::vector<int32_t> scores(subs.size());
      std::vector<int64_t> ncig(subs.size());
      std::vector<uint32_t> cig;
      {
        const size_t nsub = subs.size();
        std::vector<std::vector<int32_t>> p_sc((size_t)G);
        std::vector<std::vector<int64_t>> p_nc((size_t)G);
        std::vector<std::vector<uint32_t>> p_cg((size_t)G);
        std::vector<std::thread> pool;
        auto run = [&](int g) {
          std::vector<uint8_t> q2, t2;
          std::vector<int64_t> qo2(1, 0), to2(1, 0);
          if (G > 1)
            for (size_t k = (size_t)g; k < nsub; k += (size_t)G) {
              q2.insert(q2.end(), q.begin() + qo[k], q.begin() + qo[k + 1]);
              qo2.push_back((int64_t)q2.size());
              t2.insert(t2.end(), t.begin() + to[k], t.begin() + to[k + 1]);
              to2.push_back((int64_t)t2.size());
            }
          const int64_t np = G == 1 ? (int64_t)nsub : (int64_t)qo2.size() - 1;
          svdss_aln_batch_t* ab = nullptr;
          check(svdss_align_global_batch(G == 1 ? q.data() : q2.data(), G == 1 ? qo.data() : qo2.data(),
                                         G == 1 ? t.data() : t2.data(), G == 1 ? to.data() : to2.data(), np, 5, mat, 16, 2, 41,
                                         1, g % n_dev, &ab), "svdss_align_global_batch");
          p_sc[(size_t)g].resize((size_t)np);
          p_nc[(size_t)g].resize((size_t)np);
          p_cg[(size_t)g].resize((size_t)svdss_aln_batch_total_cigar(ab));
          check(svdss_aln_batch_fetch(ab, p_sc[(size_t)g].data(), p_nc[(size_t)g].data(), p_cg[(size_t)g].data()),
                "svdss_aln_batch_fetch");
          svdss_aln_batch_free(ab);
        };
        for (int g = 1; g < G; ++g) pool.emplace_back(run, g);
        run(0);
        for (std::thread& th : pool) th.join();
        std::vector<size_t> at((size_t)G, 0), idx((size_t)G, 0);
        for (size_t k = 0; k < nsub; ++k) {
          const size_t g = k % (size_t)G;
          scores[k] = p_sc[g][idx[g]];
          ncig[k] = p_nc[g][idx[g]++];
          cig.insert(cig.end(), p_cg[g].begin() + (long)at[g], p_cg[g].begin() + (long)(at[g] + (size_t)ncig[k]));
          at[g] += (size_t)ncig[k];
        }
      }
      std::vector<std::vector<SV>> per_thread((size_t)T);
      sam_rows.resize((size_t)T);
      size_t cp = 0;
      for (size_t i = 0; i < subs.size(); ++i) {
        const Cluster& cl = subs[i].cl;
        const Cluster& parent = clusters[subs[i].parent];
        const std::string& cs = C.chrom_seqs[cl.chrom];
        std::string cigar_str;
        for (int64_t k = 0; k < ncig[i]; ++k) cigar_str += std::to_string(cig[cp + (size_t)k] >> 4) + "MID"[cig[cp + (size_t)k] & 0xf];
        if (!o.poa.empty()) {   // Consensus, caller.hpp:56-69 / caller.cpp:356-357
          const std::string p1 = std::to_string(cl.s + 1);
          sam_rows[subs[i].parent % (size_t)T].push_back(cl.chrom + ":" + p1 + "-" + std::to_string(cl.e + 1) + "\t0\t" + cl.chrom +
                                                         "\t" + p1 + "\t60\t" + cigar_str + "\t*\t0\t0\t" + consensus[i] + "\t*");
        }
        std::string names;
        for (const SubRead& sr : cl.subreads) names += sr.name + ",";
        if (!names.empty()) names.pop_back();
        std::string rvec;
        for (const auto& rd : parent.reads) rvec += std::to_string(rd.first) + ":" + std::to_string(rd.second) + "-";
        if (!rvec.empty()) rvec.pop_back();
        std::vector<SV> local;
        unsigned rpos = (unsigned)cl.s, cpos = 0;
        int nv = 0;
        for (int64_t k = 0; k < ncig[i]; ++k) {
          const unsigned l = cig[cp + (size_t)k] >> 4;
          const char op = "MID"[cig[cp + (size_t)k] & 0xf];
          if (op == 'M') { rpos += l; cpos += l; }
          else if (op == 'I') {
            if (l >= o.min_sv_length) {
              const std::string anchor(1, cs[rpos - 1]);
              SV v = make_sv("INS", cl.chrom, (int)rpos, anchor, anchor + consensus[i].substr(cpos, l), (unsigned)cl.size(),
                             cl.cov, nv, scores[i], (int)l, cigar_str);
              v.reads = names;
              local.push_back(v);
              ++nv;
            }
            cpos += l;
          } else {
            if (l >= o.min_sv_length) {
              SV v = make_sv("DEL", cl.chrom, (int)rpos, cs.substr(rpos - 1, l + 1), std::string(1, cs[rpos - 1]),
                             (unsigned)cl.size(), cl.cov, nv, scores[i], (int)l, cigar_str);
              v.reads = names;
              local.push_back(v);
              ++nv;
            }
            rpos += l;
          }
        }
        cp += (size_t)ncig[i];
        for (SV& v : local) {
          v.ngaps = nv; v.gt = "0/1"; v.gtq = 100;
          v.cov = cl.cov; v.cov0 = cl.cov0; v.cov1 = cl.cov1; v.cov2 = cl.cov2;
          v.rvec = rvec;
          per_thread[subs[i].parent % (size_t)T].push_back(v);
        }
      }
      for (int t = 0; t < T; ++t) svs.insert(svs.begin(), per_thread[(size_t)t].begin(), per_thread[(size_t)t].end());   // caller.cpp:18-22
    }
    stage("realign + SV extraction");
  }

  // sort, clean_dups, filter_sv_chains, sort (caller.cpp:23-28, 409-475)
  void dedup_and_filter() {
    std::sort(svs.begin(), svs.end());   // same libstdc++ std::sort as the reference (caller.cpp:23)
    {   // clean_dups (caller.cpp:409-426)
      std::vector<SV> kept;
      std::string lc, lr, la; int lp = -1;
      for (const SV& v : svs) {
        if (lc != v.chrom || lp != v.s || lr != v.refall || la != v.altall) kept.push_back(v);
        lc = v.chrom; lp = v.s; lr = v.refall; la = v.altall;
      }
      svs.swap(kept);
    }
    if (svs.size() >= 2) {   // filter_sv_chains (caller.cpp:429-475): ratios of adjacent candidates in one batch
      std::vector<size_t> cand;
      std::vector<uint8_t> A, B;
      std::vector<int64_t> ao(1, 0), bo(1, 0);
      for (size_t i = 1; i < svs.size(); ++i) {
        const SV &prev = svs[i - 1], &sv = svs[i];
        if (sv.chrom == prev.chrom && sv.s - prev.e < 2 * sv.l && prev.type == sv.type) {
          const double w_r = std::min((double)sv.w, (double)prev.w) / std::max((double)sv.w, (double)prev.w);
          const double l_r = std::min((double)sv.l, (double)prev.l) / std::max((double)sv.l, (double)prev.l);
          const int d = sv.s - prev.s;
          if (d < 100 && w_r >= 0.9 && l_r >= o.min_ratio) {
            const std::string& x = sv.type == "DEL" ? sv.refall : sv.altall;
            const std::string& y = sv.type == "DEL" ? prev.refall : prev.altall;
            A.insert(A.end(), x.begin(), x.end()); ao.push_back((int64_t)A.size());
            B.insert(B.end(), y.begin(), y.end()); bo.push_back((int64_t)B.size());
            cand.push_back(i);
          }
        }
      }
      std::map<size_t, double> sim;
      if (!cand.empty()) {
        std::vector<double> ratio(cand.size());
        uint8_t dummy = 0;
        check(svdss_indel_ratio_batch(A.empty() ? &dummy : A.data(), ao.data(), B.empty() ? &dummy : B.data(), bo.data(),
                                      (int64_t)cand.size(), 0, ratio.data(), nullptr), "svdss_indel_ratio_batch");
        for (size_t k = 0; k < cand.size(); ++k) sim[cand[k]] = ratio[k];
      }
      std::vector<SV> kept;
      SV prev = svs[0];
      bool reset = false;
      for (size_t i = 1; i < svs.size(); ++i) {
        if (reset) { reset = false; prev = svs[i]; continue; }
        const SV& sv = svs[i];
        auto it = sim.find(i);
        if (it != sim.end() && it->second > 70) { kept.push_back(sv.w > prev.w ? sv : prev); reset = true; continue; }
        kept.push_back(prev);
        prev = sv;
      }
      kept.push_back(prev);
      svs.swap(kept);
    }
    std::sort(svs.begin(), svs.end());
    stage("dups + chain filter");
    // ---- write_vcf (caller.cpp:59-63, 477-550)
  }

  // write_vcf / write_sam / --clipped rows (caller.cpp:36-75, 477-550)
  void write_outputs() {
    std::string out = "##fileformat=VCFv4.2\n##reference=ftp://ftp.1000genomes.ebi.ac.uk/vol1/ftp/data_collections/HGSVC2/"
                      "technical/reference/20200513_hg38_NoALT/hg38.no_alt.fa.gz\n";
    for (const std::string& n : C.chrom_names) out += "##contig=<ID=" + n + ",length=" + std::to_string(C.chrom_seqs[n].size()) + ">\n";
    out += "##FILTER=<ID=PASS,Description=\"All filters passed\">\n";
    for (const auto& f : VCF_INFO) out += std::string("##INFO=<ID=") + f[0] + ",Number=" + f[1] + ",Type=" + f[2] + ",Description=\"" + f[3] + "\">\n";
    out += "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n";
    out += "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">\n";
    out += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tDEFAULT\n";
    for (const SV& v : svs) out += v.line() + "\n";
    fwrite(out.data(), 1, out.size(), stdout);
    fflush(stdout);
    logmsg("info", "Writing " + std::to_string(svs.size()) + " SVs.");
    stage("vcf");
    if (cache_release.joinable()) cache_release.join();
    if (getenv("SVDSS_CLEAN_EXIT")) {   // (otherwise the process ends with _exit)
      for (svdss_bam_store_t* st : stores) svdss_bam_store_free(st);
      for (svdss_bam_store_t* st : seam_stores) svdss_bam_store_free(st);
      stores.clear(); seam_stores.clear();
    }
    // ---- write_sam (caller.cpp:65-75): rows in the order of the reference's per-thread lists, each inserted at the
    // front of the global one (caller.cpp:18-22)
    if (!o.poa.empty()) {
      logmsg("info", "Writing POA alignments to " + o.poa + "..");
      FILE* f = fopen(o.poa.c_str(), "w");
      if (!f) die("cannot write " + o.poa);
      std::string hdr = "@HD\tVN:1.4\n";
      for (const std::string& n : C.chrom_names) hdr += "@SQ\tSN:" + n + "\tLN:" + std::to_string(C.chrom_seqs[n].size()) + "\n";
      fwrite(hdr.data(), 1, hdr.size(), f);
      for (size_t t = sam_rows.size(); t-- > 0;)
        for (const std::string& row : sam_rows[t]) { fwrite(row.data(), 1, row.size(), f); fputc('\n', f); }
      fclose(f);
    }
    if (o.clipped) {   // caller.cpp:36-53: imprecise rows after the VCF, in the Clipper's own order (not sorted)
      logmsg("warning", "Calling imprecise SVs from clipped alignments is experimental");
      std::vector<std::pair<int, int>> called;
      for (const SV& v : svs) called.emplace_back(v.s - 1000, v.e + 1000);
      const std::vector<SV> rows = call_clipped(clips, C.chrom_names, C.chrom_seqs, T, called);
      logmsg("info", "Predicted " + std::to_string(rows.size()) + " SVs from clipped alignments");
      std::string text;
      for (const SV& v : rows) text += v.line() + "\n";
      fwrite(text.data(), 1, text.size(), stdout);
      fflush(stdout);
    }
  }

  int run() {
    load_inputs();
    align_and_extend();
    cluster_by_proximity();
    fill_clusters();
    store_and_split();
    run_poa();
    realign_and_extract();
    dedup_and_filter();
    write_outputs();
    return 0;
  }
};

}  // namespace

int main_call(const CallOptions& o) {
  CallRun run(o);
  return run.run();
}
