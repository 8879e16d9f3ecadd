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

    def through_index(regions, index=None):
        r = subprocess.run([exe, str(bam), str(index or bai)] + [f"{t}:{b}-{e}" for t, b, e in regions], capture_output=True, text=True)
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
    # round 5: CSI indexes (what `samtools index -c` writes; htslib's sam_index_load takes either): the BAI scheme and
    # three others -- a deeper tree, finer and coarser smallest bins
    for ms, dp in ((14, 5), (14, 6), (12, 5), (16, 3)):
        csi = tmp_path / f"x.{ms}.{dp}.csi"
        csi.write_bytes(bam_writer.csi(data, ms, dp))
        for regions in cases:
            got = through_index(regions, csi)
            want = overlapping(regions)
            assert [g for g in got if g in set(want)] == want, (ms, dp, regions)
            assert len(set(got)) == len(got), (ms, dp, regions)
        assert len(through_index([(0, 1_000_000, 1_000_001)], csi)) < len(meta) // 10
    # a damaged CSI is refused or answers with chunks of the file; it is never believed beyond its bytes
    good = bam_writer.csi(data, 14, 5)
    import zlib
    plain = bytearray()
    pos = 0
    while pos + 18 <= len(good):
        bsize = int.from_bytes(good[pos + 16:pos + 18], "little") + 1
        plain += zlib.decompress(good[pos + 18:pos + bsize - 8], -15)
        pos += bsize
    for k in range(60):
        d = bytearray(plain)
        if k % 3 == 0:
            d = d[:int(rng.integers(0, len(d)))]
        else:
            at = int(rng.integers(4, min(len(d), 400)))
            d[at:at + 4] = int(rng.choice([-1, 2**31 - 1, 2**30, 0, 7])).to_bytes(4, "little", signed=True)
        bad = tmp_path / "bad.csi"
        bad.write_bytes(bam_writer.bgzf(bytes(d)))
        r = subprocess.run([exe, str(bam), str(bad), "0:0-3000000"], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 1), r.stderr[-300:]


def test_corrupt_block_headers_are_reported_not_followed(tmp_path, exe):
    """a BSIZE below the fixed header + footer, or an ISIZE above BGZF's 64 KB, must end in
    "bad BGZF block" -- not in a wrapped length or a multi-GB allocation -- in the indexed scan (bai_index.h) and in
    the sequential reader (bam_reader.h)."""
    import struct
    rng = np.random.default_rng(5)
    recs = []
    for k in range(60):
        l = 9000
        seq = "".join("ACGT"[x] for x in rng.integers(0, 4, size=l))
        recs.append(bam_writer.record(f"r{k}", 0, 0, 1000 * k, 60, [("M", l)], seq,
                                      qual=bytes(rng.integers(20, 60, size=l, dtype=np.uint8).tolist())))
    data = bytearray(bam_writer.bam([("c0", 500_000)], recs))
    (tmp_path / "ok.bam").write_bytes(data)
    (tmp_path / "ok.bam.bai").write_bytes(bam_writer.bai(bytes(data)))
    # walk the blocks, pick the third one
    offs, pos = [], 0
    while pos < len(data):
        bsize = struct.unpack_from("<H", data, pos + 16)[0]
        offs.append((pos, bsize + 1))
        pos += bsize + 1
    assert len(offs) > 6
    at, ln = offs[2]
    for name, patch in (("bsize", lambda d: struct.pack_into("<H", d, at + 16, 10)),
                        ("isize", lambda d: struct.pack_into("<I", d, at + ln - 4, 0x7fffffff))):
        bad = bytearray(data)
        patch(bad)
        p = tmp_path / f"{name}.bam"
        p.write_bytes(bad)
        (tmp_path / f"{name}.bam.bai").write_bytes((tmp_path / "ok.bam.bai").read_bytes())
        r = subprocess.run([exe, str(p), str(p) + ".bai", "0:0-500000"], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "bad BGZF block" in r.stderr, (name, r.stderr)
        r = subprocess.run([exe, str(p), "-"], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "BGZF" in r.stderr, (name, r.stderr)
