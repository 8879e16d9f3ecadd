"""SURVEY App. A traps pushed through the `SVDSS call` BINARY with expected outputs derived by
hand from /root/reference/clusterer.cpp and caller.cpp -- not from the Python mirror.

All scenarios share one construction that makes every coordinate derivable on paper: the reference is random except for
a 400-base run of G at [2800, 3200); the sample carries an insertion at 3000, in the middle of the run; a read is
ref[2000:3000] + INS + ref[3000:4000] aligned 1000M <len>I 1000M at 2000, and its SFS is written by hand as
(qs = 995, l = len + 10): last placed base before it q = 994 -> r = 2994, first placed base after it q = 1005 + len ->
r = 3005 (clusterer.cpp:184-203).  The 100 pairs in front of the SFS (r = 2894..2993) and behind it (r = 3006..3105)
lie in the G run: no 7-mer is unique there, so get_unique_kmers falls through to the OUTERMOST clean k-mer
(clusterer.cpp:398-401, trap #20): prekmer = (q 894, r 2894), postkmer = (q, r 3099), and the extended SFS is
rs = 2894, re = 3099 + 7 = 3106 (clusterer.cpp:305-308).  fill_clusters cuts read[qs..qe] with qs = the q at r = 2894
(894) and qe = the q at r = 3106 (1166 + len - 60), both inclusive (clusterer.cpp:564-585): 106 G + INS + 107 G."""
import os
import subprocess

import numpy as np
import pytest

from svdss_amd import synth
from tests import bam_writer
from tests.common import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "svdss_amd", "SVDSS")


def _reference(seed):
    rng = np.random.default_rng(seed)
    ref = rng.integers(1, 5, size=6000).astype(np.uint8)
    ref[2800:3200] = 3
    return synth.to_ascii(ref)


def _ins(seed, n):
    rng = np.random.default_rng(seed)
    return "".join("ACT"[int(x)] for x in rng.integers(0, 3, size=n))    # no G: the insertion cannot slide in the G run


def _read(ref, ins):
    return ref[2000:3000] + ins + ref[3000:4000], [("M", 1000), ("I", len(ins)), ("M", 1000)]


def _call(tmp_path, contigs, recs, sfs_lines, threads, extra=()):
    fa = tmp_path / "ref.fa"
    fa.write_text("".join(f">{n}\n{s}\n" for n, s in contigs))
    recs = sorted(recs, key=lambda r: (r[1], r[2]))
    bam = tmp_path / f"reads{threads}.bam"
    bam.write_bytes(bam_writer.bam([(n, len(s)) for n, s in contigs],
                                   [bam_writer.record(nm, 0, tid, pos, 60, cig, seq, tags) for nm, tid, pos, cig, seq, tags in recs]))
    sfs = tmp_path / "specifics.txt"
    sfs.write_text("".join(sfs_lines))
    cl = tmp_path / f"clusters{threads}.txt"
    r = subprocess.run([BIN, "call", "--reference", str(fa), "--bam", str(bam), "--sfs", str(sfs), "--threads", str(threads),
                        "--min-sv-length", "50", "--min-cluster-weight", "2", "--clusters", str(cl), *extra],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in r.stdout.splitlines() if not l.startswith("#")]
    info = [dict(x.split("=", 1) for x in f[7].split(";") if "=" in x) for f in rows]
    return rows, info, cl.read_text().splitlines()


def test_trap20_kmer_fall_through_sets_the_cluster_borders(tmp_path):
    ref, ins = _reference(1), _ins(2, 60)
    seq, cig = _read(ref, ins)
    recs = [(n, 0, 2000, cig, seq, []) for n in ("r1", "r2")]
    rows, info, clusters = _call(tmp_path, [("chr1", ref)], recs, [f"{n}\t995\t70\t0\t\n" for n in ("r1", "r2")], 4)
    sub = "G" * 106 + ins + "G" * 107
    assert clusters == [f"chr1:2895-3107\t2\tr1:{sub}\tr2:{sub}"]          # clusterer.cpp:613-626: s + 1, e + 1
    # (with (-1, -1) on "no unique k-mer" the cluster would have been [2994, 3005 + 7])
    assert len(rows) == 1 and rows[0][0] == "chr1" and rows[0][1] == "3000" and info[0]["SVTYPE"] == "INS"
    assert info[0]["SVLEN"] == "60" and info[0]["WEIGHT"] == "2" and rows[0][4] == "G" + ins


def test_trap7_cluster_key_without_chromosome(tmp_path):
    """Two chromosomes with the same sequence and the same insertion: the windows (one per chromosome) get the same
    (low, high) key.  With one thread both land in one std::map (clusterer.hpp:177, clusterer.cpp:461,471): ONE cluster,
    chrom of its first SFS (chr1, sorted first), filled from chr1's reads only -- the chr2 call is lost.  With two
    threads window i goes to thread i % 2: two clusters, two calls."""
    ref, ins = _reference(3), _ins(4, 60)
    seq, cig = _read(ref, ins)
    recs = [("a1", 0, 2000, cig, seq, []), ("a2", 0, 2000, cig, seq, []), ("b1", 1, 2000, cig, seq, []), ("b2", 1, 2000, cig, seq, [])]
    sfs = [f"{n}\t995\t70\t0\t\n" for n in ("a1", "a2", "b1", "b2")]
    sub = "G" * 106 + ins + "G" * 107
    rows, info, clusters = _call(tmp_path, [("chr1", ref), ("chr2", ref)], recs, sfs, 1)
    assert clusters == [f"chr1:2895-3107\t2\ta1:{sub}\ta2:{sub}"]
    assert [(f[0], f[1], i["WEIGHT"]) for f, i in zip(rows, info)] == [("chr1", "3000", "2")]
    rows, info, clusters = _call(tmp_path, [("chr1", ref), ("chr2", ref)], recs, sfs, 2)
    assert clusters == [f"chr1:2895-3107\t2\ta1:{sub}\ta2:{sub}", f"chr2:2895-3107\t2\tb1:{sub}\tb2:{sub}"]
    assert [(f[0], f[1], i["WEIGHT"]) for f, i in zip(rows, info)] == [("chr1", "3000", "2"), ("chr2", "3000", "2")]


def test_trap13_zero_based_values_in_a_one_based_region(tmp_path):
    """Coverage counts the primary alignments htslib returns for "chr1:2894-3106" (clusterer.cpp:523-540): the 0-based
    cluster borders printed into a 1-based inclusive region, i.e. 0-based [2893, 3106).  A read whose last base is 2893
    is counted, one ending at 2892 is not; a read starting at 3105 is counted, one starting at 3106 (still inside the
    cluster [2894, 3106]) is not."""
    ref, ins = _reference(5), _ins(6, 60)
    seq, cig = _read(ref, ins)
    recs = [(n, 0, 2000, cig, seq, []) for n in ("r1", "r2")]
    edge = {"endAt2893": 2394, "endAt2892": 2393}                     # 500M: covers [pos, pos + 500)
    for n, pos in edge.items():
        recs.append((n, 0, pos, [("M", 500)], ref[pos:pos + 500], []))
    for n, pos in {"startAt3105": 3105, "startAt3106": 3106}.items():
        recs.append((n, 0, pos, [("M", 500)], ref[pos:pos + 500], []))
    rows, info, clusters = _call(tmp_path, [("chr1", ref)], recs, [f"{n}\t995\t70\t0\t\n" for n in ("r1", "r2")], 4)
    assert len(rows) == 1 and info[0]["WEIGHT"] == "2"
    # (an untagged sub-cluster reports COV1 = COV2 = -1: caller.cpp:121-122)
    assert (info[0]["COV"], info[0]["COV0"], info[0]["COV1"], info[0]["COV2"]) == ("4", "4", "-1", "-1")


def test_trap9_int_truncated_ratios_drop_the_in_between_read(tmp_path):
    """Two tagged haplotypes with insertions of 60 (sub-reads of 273) and 62 (275), plus two untagged reads: one with the
    60-base allele (273: ratio 1.0 to haplotype 1 -> int 1, 273/275 -> int 0: joins haplotype 1) and one with a 61-base
    allele (274: both ratios in [.97, 1) -> int 0, 0 > 0 fails both ways, caller.cpp:162-210): dropped, although it fits
    both.  Weights 4 and 3; the calls are not chained (weight ratio 3/4 < 0.9, caller.cpp:449-453)."""
    ref = _reference(7)
    ins60 = _ins(8, 60)
    ins62 = ins60 + "AC"
    ins61 = ins60 + "A"
    recs, sfs = [], []
    for n, ins, hp in [("h1a", ins60, 1), ("h1b", ins60, 1), ("h1c", ins60, 1), ("h2a", ins62, 2), ("h2b", ins62, 2),
                       ("h2c", ins62, 2), ("u60", ins60, 0), ("u61", ins61, 0)]:
        seq, cig = _read(ref, ins)
        recs.append((n, 0, 2000, cig, seq, [("HP", "C", hp)] if hp else []))
        sfs.append(f"{n}\t995\t{len(ins) + 10}\t{hp}\t\n")
    rows, info, clusters = _call(tmp_path, [("chr1", ref)], recs, sfs, 4)
    assert len(clusters) == 1 and clusters[0].split("\t")[:2] == ["chr1:2895-3107", "8"]
    got = sorted((i["SVLEN"], i["WEIGHT"], tuple(sorted(i["READS"].split(",")))) for i in info)
    assert got == [("60", "4", ("h1a", "h1b", "h1c", "u60")), ("62", "3", ("h2a", "h2b", "h2c"))]
    assert all(f[1] == "3000" for f in rows)


def test_clipped_breakpoints_near_a_called_sv_of_any_chromosome_are_dropped(tmp_path):
    """--clipped (caller.cpp:36-53, clipper.cpp:93-102): the called insertion chr1:3000 puts [2000, 4000] into an interval
    tree that has no chromosome, so the facing soft clips at chr2:3500 (three reads each side) are dropped although chr2
    has no call; those at chr2:4900 are kept and give an imprecise <INS> after the VCF rows (both sides at 4900: every
    lookup of clipper.cpp:104-124 stays inside the lists).  Clip reads: 800M300S ending at B / 300S800M starting at B,
    SFS over the clipped end (clusterer.cpp:207-226)."""
    ref1, ref2, ins = _reference(11), _reference(12), _ins(13, 60)
    seq, cig = _read(ref1, ins)
    recs = [(n, 0, 2000, cig, seq, []) for n in ("r1", "r2")]
    sfs = [f"{n}\t995\t70\t0\t\n" for n in ("r1", "r2")]
    tail = "ACT" * 100
    for b in (3500, 4900):
        for i in range(3):
            recs.append((f"t{b}_{i}", 1, b - 800, [("M", 800), ("S", 300)], ref2[b - 800:b] + tail, []))
            sfs.append(f"t{b}_{i}\t790\t310\t0\t\n")
            recs.append((f"l{b}_{i}", 1, b, [("S", 300), ("M", 800)], tail + ref2[b:b + 800], []))
            sfs.append(f"l{b}_{i}\t0\t310\t0\t\n")
    rows, info, clusters = _call(tmp_path, [("chr1", ref1), ("chr2", ref2)], recs, sfs, 2, extra=("--clipped",))
    assert [(f[0], f[1], f[4] if f[4].startswith("<") else "seq") for f in rows] == [("chr1", "3000", "seq"), ("chr2", "4900", "<INS>")]
    assert rows[1][2] == "INS_chr2:4900-4900_300" and rows[1][3] == ref2[4900] and rows[1][7].endswith(";READS=;IMPRECISE")
    assert info[1]["WEIGHT"] == "3" and info[1]["SVLEN"] == "300" and rows[1][9] == "./.:0"
