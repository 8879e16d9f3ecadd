"""`SVDSS call --clipped` (clipper.cpp, caller.cpp:36-53, clusterer.cpp:207-226,338-345) through the
binary, expected rows derived by hand from the reference's source.  The scenarios have no SFS cluster, so nothing is
sent to the GPU: this runs on the CPU box as well.

One lookup in most scenarios is outside what the reference defines: its search (clipper.cpp:104-124) leaves the array
whenever the query lies left of every centre of the other side (`m - 1` on an unsigned 0 -> index 2^31), which the
leftmost centre of one side or the other always does unless both sides start at the same position (the first test).
The binary returns the first centre there (what the function's comment says it looks for); in the scenarios below
those lookups never produce a row, so every expected ROW follows from defined behaviour of the reference's code.

Construction: a read that ends in a soft clip at reference position B (cigar 2000M500S at B - 2000) with an SFS over
its last 510 bases has a placed base in front of the SFS and none behind it (clusterer.cpp:184-203: refe == -1): a
TRAILING clip (bam_endpos = B, 500).  A read that starts with a soft clip (500S2000M at B) with an SFS over its first
510 bases has refs == -1: a LEADING clip (pos = B, 500)."""
import os
import subprocess

import numpy as np

from svdss_amd import synth
from tests import bam_writer
from tests.common import ROOT

from tests.common import BIN  # noqa: E402
CLIP = "ACGGTCA" * 72            # 504 bases; the first 500 are the clipped part of every read


def _ref(seed, n=20000):
    return synth.to_ascii(np.random.default_rng(seed).integers(1, 5, size=n).astype(np.uint8))


def trailing(name, ref, b, tid=0, clip=500):
    return (name, tid, b - 2000, [("M", 2000), ("S", clip)], ref[b - 2000:b] + CLIP[:clip], []), f"{name}\t1990\t{clip + 10}\t0\t\n"


def leading(name, ref, b, tid=0, clip=500):
    return (name, tid, b, [("S", clip), ("M", 2000)], CLIP[:clip] + ref[b:b + 2000], []), f"{name}\t0\t{clip + 10}\t0\t\n"


def call(tmp_path, contigs, items, threads):
    recs = sorted((r for r, _ in items), key=lambda r: (r[1], r[2]))
    (tmp_path / "ref.fa").write_text("".join(f">{n}\n{s}\n" for n, s in contigs))
    (tmp_path / "reads.bam").write_bytes(bam_writer.bam(
        [(n, len(s)) for n, s in contigs],
        [bam_writer.record(nm, 0, tid, pos, 60, cig, seq, tags) for nm, tid, pos, cig, seq, tags in recs]))
    (tmp_path / "sfs.txt").write_text("".join(s for _, s in items))
    r = subprocess.run([BIN, "call", "--reference", str(tmp_path / "ref.fa"), "--bam", str(tmp_path / "reads.bam"),
                        "--sfs", str(tmp_path / "sfs.txt"), "--threads", str(threads), "--clipped"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Calling imprecise SVs from clipped alignments is experimental" in r.stderr
    return [l for l in r.stdout.splitlines() if not l.startswith("#")], r.stderr


def row(chrom, pos, kind, base, length, weight):
    # sv.cpp:5-27,53-80 for an imprecise record: e = s (one reference base), CIGAR "." (the constructor's default),
    # no reads / RVEC; the fields the reference never initialises (COV0-2, GQ) are 0 here
    svlen = -length if kind == "DEL" else length
    return (f"{chrom}\t{pos}\t{kind}_{chrom}:{pos}-{pos}_{length}\t{base}\t<{kind}>\t.\tPASS\tVARTYPE=SV;SVTYPE={kind};"
            f"SVLEN={svlen};END={pos};WEIGHT={weight};COV=0;COV0=0;COV1=0;COV2=0;AS=0;NV=0;CIGAR=.;RVEC=;READS=;IMPRECISE\t"
            "GT:GQ\t./.:0")


def test_insertion_from_facing_clips(tmp_path):
    """Three trailing and three leading clips at 8000: one breakpoint per side (w = 3, clipper.cpp:17-52), both centres;
    the leading one finds the trailing one at its own position (:111-116: the last entry itself), 0 < 1000 apart
    (:184): <INS> at the trailing position (weights equal, :185), length = the longer clip, weight 3.  The deletion
    loop finds the pair 0 bp apart: out of [2000, 50000] (:206)."""
    ref = _ref(1)
    items = [trailing(f"t{i}", ref, 8000) for i in range(3)] + [leading(f"l{i}", ref, 8000, clip=400 + 50 * i) for i in range(3)]
    for threads in (1, 2, 4):
        rows, err = call(tmp_path, [("chr1", ref)], items, threads)
        assert rows == [row("chr1", 8000, "INS", ref[8000], 500, 3)]
        assert "0/0/0 unplaced SFSs. 0 erroneus SFSs. 6 clipped SFSs." in err      # clusterer.cpp:26-27, :214-217
        assert "Predicted 1 SVs from clipped alignments" in err


def test_deletion_needs_five_reads_and_single_reads_are_dropped(tmp_path):
    """Trailing clips at 3000, leading clips at 6000, a leading breakpoint at 500 with two reads and one at 12000 with a
    single read (dropped, :54-63).  The trailing centre's search among the leading centres [500, 6000] returns 6000
    (:117-121: right of the query, its left neighbour left of it): 3000 bp apart, a deletion of 3001 starting at 3000
    -- with five reads on one side (:211), not with four.  The leading centres look for trailing ones: 500 gets 3000
    (2500 apart: no insertion), 6000 has nothing to its right."""
    ref = _ref(2)
    for n, expect in ((5, True), (4, False)):
        items = ([trailing(f"t{i}", ref, 3000) for i in range(n)] + [leading(f"l{i}", ref, 6000) for i in range(n)]
                 + [leading(f"e{i}", ref, 500) for i in range(2)] + [leading("single", ref, 12000)])
        rows, err = call(tmp_path, [("chr1", ref)], items, 2)
        assert rows == ([row("chr1", 3000, "DEL", ref[3000], 3001, n)] if expect else [])


def test_grouping_is_greedy_in_list_order_and_unsigned(tmp_path):
    """clipper.cpp:66-91.  Leading breakpoints at 6000 (five reads) and 6400 (two): the per-chromosome list comes out of
    a std::unordered_map<uint, ...> filled in record order, i.e. 6400 first (libstdc++ puts a node that opens a bucket
    at the head of its list) -- 6400 becomes the centre and takes the reads of 6000 (w = 7, position 6400).  Leading
    breakpoints at 500 and 700 stay two centres: `centre - 1000` wraps around for a centre below 1000.  Trailing
    clips at 3000 (five reads): deletion 3000..6400 (3401) with weight 7; the insertion loop pairs 500 and 700 with 3000
    (2500 / 2300 apart: nothing)."""
    ref = _ref(3)
    items = ([trailing(f"t{i}", ref, 3000) for i in range(5)] + [leading(f"a{i}", ref, 6000) for i in range(5)]
             + [leading(f"b{i}", ref, 6400) for i in range(2)] + [leading(f"c{i}", ref, 500) for i in range(2)]
             + [leading(f"d{i}", ref, 700) for i in range(2)])
    rows, err = call(tmp_path, [("chr1", ref)], items, 1)
    assert rows == [row("chr1", 3000, "DEL", ref[3000], 3401, 7)]


def test_first_clip_per_read_name_and_positions_without_chromosome(tmp_path):
    """Two quirks.  (1) remove_duplicates keeps the first clip per read NAME and side (:5-15): a second record of the
    same name does not count.  (2) The two loops compare positions only: a leading centre on chr2 at 8000 pairs with
    the trailing centre on chr1 at 8100 (:176-193; found through :117-121, the trailing centre at 2000 being its left
    neighbour) -- an <INS> on the LEADING clip's chromosome at the position of the side with more reads."""
    ref1, ref2 = _ref(4), _ref(5)
    items = ([trailing(f"t{i}", ref1, 8100, tid=0) for i in range(3)] + [trailing(f"u{i}", ref1, 2000, tid=0) for i in range(2)]
             + [leading(f"l{i}", ref2, 8000, tid=1) for i in range(2)])
    items.append(leading("l0", ref2, 9000, tid=1))          # same name as the first leading read: ignored
    rows, err = call(tmp_path, [("chr1", ref1), ("chr2", ref2)], items, 1)
    # leading {chr2, 8000, w 2}, trailing {chr1, 8100, w 3}: 100 apart; s = trailing position (3 > 2), base from chr2.
    # (the deletion loop: 2000 -> leading 8000, 6000 apart but two reads; 8100 -> nothing to its right)
    assert rows == [row("chr2", 8100, "INS", ref2[8100], 500, 3)]
