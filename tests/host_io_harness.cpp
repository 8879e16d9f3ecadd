// host_io_harness.cpp -- the host-side file readers of the SVDSS binary (csrc/bam_reader.h, bai_index.h,
// fastx_reader.h, rld0.cpp) run over one input file, for tests/test_host_io_robustness.py: built with
// -fsanitize=address,undefined, fed valid files and damaged ones.  A reader may accept a file or refuse it with its
// error message; it may not read outside its buffers, overflow, throw out of main or hang.  Test infrastructure.
//   host_io_harness bam <file> | bai <file.bam> | fastx <file> | fmd <file> | sidecar <file> | sfs <file> | scan <file.bam> | regions <file.bam>
// exit 0: read to a clean end; 1: the reader reported an error (printed); anything else: a finding.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../svdss_amd/csrc/bai_index.h"
#include "../svdss_amd/csrc/bam_reader.h"
#include "../svdss_amd/csrc/bgzf_scanner.h"
#include "../svdss_amd/csrc/bam_device_select.h"
#include "../svdss_amd/csrc/fastx_reader.h"
#include "../svdss_amd/csrc/index_host.h"
#include "../svdss_amd/csrc/rld0.h"
#include "../svdss_amd/csrc/sfs_file.h"

static uint64_t g_sum = 0;   // every byte a reader hands out is read once (ASan sees an out-of-bounds view)
static void touch(const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; ++i) g_sum += b[i];
}

static int run_bam(const std::string& path) {
  for (int pass = 0; pass < 2; ++pass) {   // the zero-copy views, then the copying reader
    BamReader bam(path, 2);
    if (!bam.ok() || !bam.read_header()) { printf("error: %s\n", bam.error().c_str()); return 1; }
    for (const std::string& n : bam.ref_names()) touch(n.data(), n.size());
    long n_rec = 0;
    for (;;) {
      int rc;
      if (pass == 0) {
        BamReader::RawView v;
        rc = bam.next_view(v);
        if (rc == 1) {
          touch(v.name(), v.l_name);
          touch(v.name() + v.l_name, 4u * v.n_cigar);
          touch(v.seq4(), ((size_t)v.l_seq + 1) / 2 + (size_t)v.l_seq);
          touch(v.aux(), v.l_aux);
          int64_t x = 0;
          BamReader::aux_int(v.aux(), v.l_aux, "XF", x);
          BamReader::aux_int(v.aux(), v.l_aux, "HP", x);
          BamRecord r;
          BamReader::materialize(v, r);
          touch(r.seq4.data(), r.seq4.size());
        }
      } else {
        BamRecord r;
        rc = bam.next(r, n_rec % 2 == 0);
        if (rc == 1) {
          touch(r.qname.data(), r.qname.size());
          touch(r.cigar.data(), 4 * r.cigar.size());
          touch(r.seq4.data(), r.seq4.size());
          touch(r.qual.data(), r.qual.size());
          touch(r.aux.data(), r.aux.size());
          int64_t x = 0;
          BamReader::aux_int(r, "XF", x);
        }
      }
      if (rc == 0) break;
      if (rc < 0) { printf("error: %s\n", bam.error().c_str()); return 1; }
      ++n_rec;
    }
    printf("pass %d: %ld records\n", pass, n_rec);
  }
  return 0;
}

// the host side of the device path (round 4): the BAM header probe, the BGZF member scanner with a few loaders and small
// slabs (every member located must lie inside what was read, its deflate stream inside the member), and the record view
// over bytes that came back from a selection -- here: over the members inflated on the host
static int run_scan(const std::string& path) {
  int32_t n_ref = 0;
  int64_t skip = 0;
  std::string err;
  std::vector<std::string> names;
  if (!bam_header_probe(path, n_ref, skip, err, &names)) { printf("error: %s\n", err.c_str()); return 1; }
  for (const std::string& n : names) touch(n.data(), n.size());
  for (size_t slab : {(size_t)64 << 10, (size_t)1 << 20}) {
    BgzfScanner sc(path, BgzfScanner::Hooks(), slab, 3, 6);
    if (!sc.ok()) { printf("error: cannot open\n"); return 1; }
    std::vector<uint8_t> stream;
    BgzfInflater inf;
    long n_members = 0;
    while (std::unique_ptr<CompChunk> c = sc.next()) {
      for (size_t i = 0; i < c->blocks.size(); ++i) {
        const svdss_bgzf_block_t& b = c->blocks[i];
        if (b.coff < 0 || b.clen < 0 || (size_t)(b.coff + b.clen) + 8 > c->n_bytes) { printf("finding: member outside the slab\n"); return 3; }
        touch(c->data + b.coff, (size_t)b.clen);
        const size_t at = stream.size();
        stream.resize(at + (size_t)b.isize);
        if (b.isize && inf.run(c->data + b.coff, (size_t)b.clen, stream.data() + at, (uint32_t)b.isize, c->crc[i])) { printf("error: inflate / crc\n"); return 1; }
        ++n_members;
      }
      sc.recycle(std::move(c));
    }
    if (!sc.error().empty()) { printf("error: %s\n", sc.error().c_str()); return 1; }
    // the records behind the header, as a selection would hand them over
    long n_rec = 0;
    size_t p = (size_t)skip;
    while (p + 4 <= stream.size()) {
      BamReader::RawView v;
      int32_t bs;
      memcpy(&bs, stream.data() + p, 4);
      if (bs < 32 || p + 4 + (size_t)bs > stream.size()) { printf("error: truncated record\n"); return 1; }
      if (!view_of_record(stream.data() + p, stream.size() - p, v)) { printf("error: corrupt record\n"); return 1; }
      touch(v.name(), v.l_name);
      touch(v.seq4(), ((size_t)v.l_seq + 1) / 2 + (size_t)v.l_seq);
      touch(v.aux(), v.l_aux);
      p += 4 + (size_t)bs;
      ++n_rec;
    }
    printf("slab %zu: %ld members, %ld records\n", slab, n_members, n_rec);
  }
  return 0;
}

// round 5: the file cut into regions at BGZF members (BgzfScanner::member_start_near) and every region scanned on its own
// (`SVDSS search --gpus N`): the members of the regions, in order, are the members of the file
static int run_regions(const std::string& path) {
  auto scan = [&](size_t begin, size_t end, std::vector<uint8_t>& stream, long& n_members) -> std::string {
    BgzfScanner sc(path, BgzfScanner::Hooks(), (size_t)64 << 10, 3, 6, begin, end);
    if (!sc.ok()) return "cannot open";
    BgzfInflater inf;
    while (std::unique_ptr<CompChunk> c = sc.next()) {
      for (size_t i = 0; i < c->blocks.size(); ++i) {
        const svdss_bgzf_block_t& b = c->blocks[i];
        if (b.coff < 0 || b.clen < 0 || (size_t)(b.coff + b.clen) + 8 > c->n_bytes) return "finding: member outside the slab";
        const size_t at = stream.size();
        stream.resize(at + (size_t)b.isize);
        if (b.isize && inf.run(c->data + b.coff, (size_t)b.clen, stream.data() + at, (uint32_t)b.isize, c->crc[i])) return "inflate / crc";
        ++n_members;
      }
      sc.recycle(std::move(c));
    }
    return sc.error();
  };
  std::vector<uint8_t> whole;
  long n_whole = 0;
  const std::string e0 = scan(0, 0, whole, n_whole);
  if (!e0.empty()) { printf("error: %s\n", e0.c_str()); return e0.rfind("finding", 0) == 0 ? 3 : 1; }
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return 1;
  const size_t fsize = (size_t)st.st_size;
  for (int n : {2, 3, 7}) {
    std::vector<size_t> cuts{0};
    for (int g = 1; g < n; ++g) {
      const size_t c = BgzfScanner::member_start_near(path, fsize * (size_t)g / (size_t)n);
      if (c > cuts.back() && c < fsize) cuts.push_back(c);
    }
    cuts.push_back(fsize);
    std::vector<uint8_t> joined;
    long n_joined = 0;
    for (size_t g = 0; g + 1 < cuts.size(); ++g) {
      const std::string e = scan(cuts[g], cuts[g + 1], joined, n_joined);
      if (!e.empty()) { printf("finding: region [%zu, %zu): %s\n", cuts[g], cuts[g + 1], e.c_str()); return 3; }
    }
    if (n_joined != n_whole || joined != whole) { printf("finding: %d regions hold %ld members, the file %ld\n", n, n_joined, n_whole); return 3; }
    printf("%zu regions: %ld members, the file's\n", cuts.size() - 1, n_joined);
  }
  return 0;
}

static int run_bai(const std::string& bam_path) {
  BaiIndex idx;
  if (!idx.load(bam_path + ".bai")) { printf("error: cannot load the index\n"); return 1; }
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  for (int tid = -1; tid <= (int)idx.refs.size(); ++tid)
    for (int64_t beg : {(int64_t)-5, (int64_t)0, (int64_t)1000, (int64_t)70000, (int64_t)1 << 29})
      idx.query(tid, beg, beg + 5000, chunks);
  BaiIndex::merge(chunks);
  long n = 0;
  const std::string err = bam_scan_chunks(bam_path, chunks, [&](const BamReader::RawView& v) {
    touch(v.name(), v.l_name);
    touch(v.seq4(), ((size_t)v.l_seq + 1) / 2);
    touch(v.aux(), v.l_aux);
    ++n;
  });
  if (!err.empty()) { printf("error: %s\n", err.c_str()); return 1; }
  printf("%zu chunks, %ld records\n", chunks.size(), n);
  return 0;
}

static int run_fastx(const std::string& path) {
  FastxReader fx(path);
  if (!fx.ok()) { printf("error: cannot open\n"); return 1; }
  std::string name, seq;
  long n = 0;
  std::vector<std::string> names, seqs;
  while (fx.next(name, seq)) { touch(name.data(), name.size()); touch(seq.data(), seq.size()); ++n; names.push_back(name); seqs.push_back(seq); }
  printf("%ld records\n", n);
  // the mapped, multi-threaded FASTA loader of `SVDSS call` (fastx_reader.h load_fasta_mapped): whenever it takes a file
  // it must give exactly the records above (sequence as it is, and upper-cased)
  for (int threads : {1, 3, 7}) {
    std::vector<std::string> nm, sq, squ;
    const bool took = load_fasta_mapped(path, threads, false, nm, sq);
    if (!took) { printf("mapped loader: declined\n"); break; }
    if (!load_fasta_mapped(path, threads, true, nm, squ)) { printf("finding: the mapped loader took the file once and not twice\n"); return 3; }
    if (nm != names || sq != seqs) { printf("finding: the mapped loader and the line reader disagree (%d threads)\n", threads); return 3; }
    for (size_t i = 0; i < sq.size(); ++i) {
      if (squ[i].size() != sq[i].size()) { printf("finding: upper-cased length\n"); return 3; }
      for (size_t k = 0; k < sq[i].size(); ++k) {
        const char c = sq[i][k];
        if (squ[i][k] != (char)(c - ((c >= 'a' && c <= 'z') ? 32 : 0))) { printf("finding: upper-casing\n"); return 3; }
      }
    }
    if (threads == 7) printf("mapped loader: the same %zu records\n", nm.size());
  }
  return 0;
}

static int run_fmd(const std::string& path) {
  std::vector<uint8_t> bwt;
  uint64_t mcnt[6];
  if (!rld0_is_fmd(path.c_str())) { printf("error: not an rld0 file\n"); return 1; }
  if (rld0_header_counts(path.c_str(), mcnt) != 0) { printf("error: header\n"); return 1; }
  const int rc = rld0_read(path.c_str(), bwt);
  if (rc != 0) { printf("error: rld0_read %d\n", rc); return 1; }
  touch(bwt.data(), bwt.size());
  std::vector<std::vector<uint8_t>> strings;
  if (rld0_strings_of_bwt(bwt.data(), (int64_t)bwt.size(), 2, strings) != 0) { printf("error: not the BWT of a string collection\n"); return 1; }
  std::vector<int64_t> picked;
  if (rld0_pick_strands(strings, picked) != 0) { printf("error: strands\n"); return 1; }
  printf("%zu symbols, %zu strings\n", bwt.size(), strings.size());
  return 0;
}

static int run_sidecar(const std::string& path) {
  // the two sidecars `SVDSS index` can leave beside the .fmd: the records (default) and the full layout
  svdss_index a, b;
  const int r1 = svdss_index_load_records_host(path.c_str(), &a);
  const int r2 = r1 == 0 ? -1 : svdss_index_load_host(path.c_str(), &b);
  if (r1 != 0 && r2 != 0) { printf("error: neither sidecar format loads (%d, %d)\n", r1, r2); return 1; }
  if (r1 == 0) {
    touch(a.records.data(), a.records.size());
    svdss_index built;   // what `search` does with the records when there is no GPU builder: the host builder
    const int rc = svdss_index_build_host(a.records.data(), a.rec_lens.data(), (int32_t)a.rec_lens.size(), 2, &built);
    if (rc != 0) { printf("error: build %d\n", rc); return 1; }
    printf("records: %zu bases, index of %lld symbols\n", a.records.size(), (long long)built.n);
  } else {
    touch(b.text.data(), b.text.size());
    printf("full layout: %lld symbols\n", (long long)b.n);
  }
  return 0;
}

static int run_sfs(const std::string& path) {
  // the piecewise reader against the line-by-line one, with several piece counts
  SfsMap want;
  if (!sfs_parse_lines(path.c_str(), want)) { printf("error: cannot open\n"); return 1; }
  size_t n = 0;
  for (const auto& kv : want) n += kv.second.size();
  for (int threads : {1, 2, 3, 7, 16}) {
    SfsMap got;
    if (!sfs_parse_file(path.c_str(), threads, got)) { printf("finding: the piecewise reader cannot open the file\n"); return 4; }
    bool same = got.size() == want.size();
    for (const auto& kv : want) {
      auto it = got.find(kv.first);
      if (it == got.end() || it->second.size() != kv.second.size()) { same = false; break; }
      for (size_t i = 0; i < kv.second.size(); ++i)
        if (it->second[i].qs != kv.second[i].qs || it->second[i].l != kv.second[i].l || it->second[i].htag != kv.second[i].htag) same = false;
    }
    if (!same) { printf("finding: %d threads read another map (%zu reads vs %zu)\n", threads, got.size(), want.size()); return 4; }
  }
  printf("%zu reads, %zu records\n", want.size(), n);
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: host_io_harness bam|bai|fastx|fmd <file>\n"); return 2; }
  const std::string mode = argv[1], path = argv[2];
  int rc = 2;
  try {
    if (mode == "bam") rc = run_bam(path);
    else if (mode == "bai") rc = run_bai(path);
    else if (mode == "fastx") rc = run_fastx(path);
    else if (mode == "fmd") rc = run_fmd(path);
    else if (mode == "sidecar") rc = run_sidecar(path);
    else if (mode == "sfs") rc = run_sfs(path);
    else if (mode == "scan") rc = run_scan(path);
    else if (mode == "regions") rc = run_regions(path);
  } catch (const std::exception& e) {
    printf("finding: exception out of a reader: %s\n", e.what());
    return 3;
  }
  if (g_sum == 0x5eed) printf(" ");
  return rc;
}
