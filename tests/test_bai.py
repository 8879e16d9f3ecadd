"""BAI index reader + chunk-restricted BAM reading (csrc/bai_index.h, `SVDSS call`'s second pass; the reference:
sam_index_load / sam_itr_querys, clusterer.cpp:495-527) on the CPU: for any set of regions the records read through the
index, filtered by overlap, are exactly the overlapping records of a sequential read, in file order -- with records that
straddle BGZF blocks, several references, empty bins, regions past the last record."""
import os
import subprocess

import numpy as np
import pytest

from tests import bam_writer
from tests.common import ROOT

SRC = os.path.join(ROOT, "tests", "native", "bai_scan.cpp")
EXE = os.path.join(ROOT, "tests", "native", "_bai_scan")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(
            os.path.join(ROOT, "svdss_amd", "csrc", "bai_index.h")), os.path.getmtime(os.path.join(ROOT, "svdss_amd", "csrc", "bam_reader.h"))):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", EXE, SRC, "-lz", "-ldl"], check=True)
    return EXE


def _lines(out):
    return [tuple(l.split("\t")) for l in out.splitlines()]


def test_regions_through_the_index_are_the_overlapping_records_of_a_sequential_read(tmp_path, exe):
    rng = np.random.default_rng(12)
    ref_lens = [3_000_000, 40_000, 900_000]
    recs, meta = [], []
    for tid, ln in enumerate(ref_lens):
        n = {0: 900, 1: 0, 2: 300}[tid]                       # the second reference has no record at all
        starts = np.sort(rng.integers(0, ln - 30_000, size=n))
        for k, st in enumerate(starts):
            l = int(rng.integers(400, 26_000))                # records from a fraction of a block to half of one
            seq = "".join("ACGT"[x] for x in rng.integers(0, 4, size=l))
            cig = [("S", 5), ("M", l - 205), ("D", 37), ("M", 200)] if k % 3 == 0 else [("M", l)]
            ref_span = l - 5 + 37 if k % 3 == 0 else l
            name = f"t{tid}r{k}"
            recs.append(bam_writer.record(name, 16 if k % 2 else 0, tid, int(st), 60, cig, seq,
                                          qual=bytes(rng.integers(20, 60, size=l, dtype=np.uint8).tolist())))
            meta.append((name, tid, int(st), int(st) + ref_span))
    bam = tmp_path / "x.bam"
    data = bam_writer.bam([(f"c{t}", l) for t, l in enumerate(ref_lens)], recs)
    bam.write_bytes(data)
    assert len(data) > 40 * 60000                              # dozens of blocks; most records straddle one
    bai = tmp_path / "x.bam.bai"
    bai.write_bytes(bam_writer.bai(data))
    seq_read = _lines(subprocess.run([exe, str(bam), "-"], capture_output=True, text=True, check=True).stdout)
    assert seq_read == [(n, str(t), str(p)) for n, t, p, e in meta]

    def through_index(regions):
        r = subprocess.run([exe, str(bam), str(bai)] + [f"{t}:{b}-{e}" for t, b, e in regions], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return _lines(r.stdout)

    def overlapping(regions):
        return [(n, str(t), str(p)) for n, t, p, e in meta if any(t == rt and p < re and e > rb for rt, rb, re in regions)]

    cases = [[(0, 0, 3_000_000)], [(0, 1_000_000, 1_000_001)], [(0, 16_384, 32_768)], [(2, 899_000, 900_000)],
             [(1, 0, 40_000)], [(0, 2_999_999, 3_000_000), (2, 0, 1)],
             [(0, 500_000, 520_000), (0, 510_000, 700_000), (2, 100_000, 130_000), (0, 2_000_000, 2_000_100)]]
    for _ in range(12):
        k = int(rng.integers(1, 9))
        cases.append(sorted((int(t), int(b), int(b) + int(w)) for t, b, w in
                            zip(rng.choice([0, 0, 2], size=k), rng.integers(0, 880_000, size=k), rng.integers(1, 60_000, size=k))))
    for regions in cases:
        got = through_index(regions)
        want = overlapping(regions)
        # everything that overlaps is read, in file order, once; what else the chunks hold does not overlap
        assert [g for g in got if g in set(want)] == want, regions
        assert len(set(got)) == len(got), regions
    # the index is worth having: a small region reads a small part of the file
    assert len(through_index([(0, 1_000_000, 1_000_001)])) < len(meta) // 10
