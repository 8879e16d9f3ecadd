// Test harness (CPU only): the records `SVDSS call`'s second pass would read through a BAI index for a list of
// regions (csrc/bai_index.h), and the records a sequential read of the whole file gives (csrc/bam_reader.h).
//   bai_scan <bam> <bai> tid:beg-end [tid:beg-end ...]   ->  one line per record: name tid pos      (index path)
//   bai_scan <bam> -                                      ->  the same for every record of the file (sequential)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../svdss_amd/csrc/bai_index.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string bam = argv[1], idx = argv[2];
  auto print = [](const BamReader::RawView& v) {
    printf("%.*s\t%d\t%d\n", (int)(v.l_name ? v.l_name - 1 : 0), (const char*)v.name(), v.tid, v.pos);
  };
  if (idx == "-") {
    BamReader r(bam, 2);
    if (!r.ok() || !r.read_header()) { fprintf(stderr, "%s\n", r.error().c_str()); return 1; }
    BamReader::RawView v;
    int rc;
    while ((rc = r.next_view(v)) > 0) print(v);
    if (rc < 0) { fprintf(stderr, "%s\n", r.error().c_str()); return 1; }
    return 0;
  }
  BaiIndex bai;
  if (!bai.load(idx)) { fprintf(stderr, "cannot load %s\n", idx.c_str()); return 1; }
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  for (int i = 3; i < argc; ++i) {
    int tid; long long b, e;
    if (sscanf(argv[i], "%d:%lld-%lld", &tid, &b, &e) != 3) return 2;
    bai.query(tid, b, e, chunks);
  }
  BaiIndex::merge(chunks);
  fprintf(stderr, "%zu chunks\n", chunks.size());
  const std::string err = bam_scan_chunks(bam, chunks, print);
  if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  return 0;
}
