"""a10-a12: Clusterer host logic (/root/reference/clusterer.cpp) on hand-worked inputs."""
import numpy as np

from svdss_amd import synth
from tests.mirror.clusterer import (Alignment, BAM_CDEL, BAM_CINS, BAM_CMATCH, BAM_CSOFT_CLIP, Clusterer, ExtSFS,
                                 get_aligned_pairs, get_unique_kmers)


def test_aligned_pairs():
    a = Alignment("r", 0, 0, 100, 60, [(2, BAM_CSOFT_CLIP), (3, BAM_CMATCH), (2, BAM_CINS), (2, BAM_CDEL), (1, BAM_CMATCH)], "A" * 8)
    assert get_aligned_pairs(a) == [(0, -1), (1, -1), (2, 100), (3, 101), (4, 102), (5, -1), (6, -1), (-1, 103),
                                    (-1, 104), (7, 105)]
    assert a.endpos() == 106


def test_unique_kmers_first_last_and_fallthrough():
    seq = "ACGTACGTACGTTTGCA" + "G" * 40
    pairs = [(i, i) for i in range(17)]
    # k=4: ACGT x3, CGTA x2, GTAC x2, TACG x2, CGTT, GTTT, TTTG, TTGC, TGCA unique
    assert get_unique_kmers(pairs, 4, False, seq) == (9, 9)      # first unique from the start: CGTT at 9
    assert get_unique_kmers(pairs, 4, True, seq) == (13, 13)     # from the end: TGCA at 13
    assert get_unique_kmers(pairs[:3], 4, True, seq) == (-1, -1)  # fewer than k pairs
    rep = [(i, i) for i in range(20, 40)]                         # all-G flank: no unique k-mer ->
    assert get_unique_kmers(rep, 4, False, seq) == (36, 36)      # outermost clean k-mer examined (App. A#20)
    assert get_unique_kmers(rep, 4, True, seq) == (20, 20)
    gap = [(0, 0), (1, 1), (2, -1), (3, 3), (4, 4), (5, 5), (6, 6), (7, 7)]   # insertion breaks k-mers
    assert get_unique_kmers(gap, 4, False, seq) == (3, 3)


def _mk(sfs, threads=2):
    rng = np.random.default_rng(1)
    ref = synth.to_ascii(rng.integers(1, 5, size=5000).astype(np.uint8))
    return Clusterer(sfs, {"chr1": ref}, ["chr1"], threads=threads), ref


def test_extend_alignment_insertion():
    # read = ref[1000:1400] + 50 inserted + ref[1400:1800]; SFS over the insertion junctions
    c, ref = _mk({"r1": [(395, 60, 1)]})
    ins = "ACGT" * 12 + "AC"
    seq = ref[1000:1400] + ins + ref[1400:1800]
    a = Alignment("r1", 0, 0, 1000, 60, [(400, BAM_CMATCH), (50, BAM_CINS), (400, BAM_CMATCH)], seq)
    out = c.extend_alignment(a)
    assert len(out) == 1
    x = out[0]
    assert x.chrom == "chr1" and x.qname == "r1" and x.htag == 1
    # the extension reaches the nearest unique 7-mers in the 100 pairs before / after the placed SFS
    assert 1394 - 100 <= x.rs <= 1394 and 1405 <= x.re <= 1405 + 100 + 7
    assert x.qs == x.rs - 1000 and x.qe == x.re - 1000 + 50       # query coordinates carry the insertion
    assert c.unplaced == 0


def test_unplaced_and_merge():
    c, ref = _mk({"r1": [(0, 30, 0), (200, 20, 0), (230, 20, 0), (780, 20, 0)]})
    a = Alignment("r1", 0, 0, 2000, 60, [(800, BAM_CMATCH)], ref[2000:2800])
    out = c.extend_alignment(a)
    # first SFS starts at the first base (no placed base before it) and the last one ends at the last base:
    # both are skipped (s_unplaced / e_unplaced); the two middle ones extend into each other and merge
    assert c.s_unplaced == 1 and c.e_unplaced == 1
    assert len(out) == 1 and out[0].rs <= 2199 and out[0].re >= 2250


def test_cluster_by_proximity_and_thread_order():
    c, ref = _mk({}, threads=2)
    c.extended_SFSs = [ExtSFS("chr1", "a", 100, 150, 0, 0, 0), ExtSFS("chr1", "b", 140, 200, 0, 0, 0),
                       ExtSFS("chr1", "c", 230, 260, 0, 0, 0),      # 230 - 150 = 80 > dist = int(60*1.1) = 66: new window
                       ExtSFS("chr1", "d", 1000, 1040, 0, 0, 0),    # (prev_e is the end of the window's FIRST SFS, :418-436)
                       ExtSFS("chr2", "e", 50, 90, 0, 0, 0)]
    groups = c.cluster_by_proximity()
    names = [[s.qname for s in g] for g in groups]
    # windows [a,b] [c] [d] [e] go to threads 0,1,0,1 (static,1); each thread's std::map is keyed by (low, high)
    # WITHOUT the chromosome, so on thread 1 chr2's (50,90) sorts before chr1's (230,260) (App. A#7)
    assert names == [["a", "b"], ["d"], ["e"], ["c"]]


def test_fill_clusters_subreads_and_coverage():
    c, ref = _mk({}, threads=1)
    c.min_cluster_weight = 2
    alns = []
    for i, (pos, hp) in enumerate([(900, 1), (950, 2), (1000, 0), (1190, 0), (3000, 0)]):
        alns.append(Alignment(f"r{i}", 0, 0, pos, 60, [(400, BAM_CMATCH)], ref[pos:pos + 400], {"HP": hp} if hp else {}))
    alns.append(Alignment("lowq", 0, 0, 1000, 5, [(400, BAM_CMATCH)], ref[1000:1400]))
    group = [ExtSFS("chr1", "r0", 1100, 1180, 200, 280, 1), ExtSFS("chr1", "r1", 1110, 1200, 160, 250, 2),
             ExtSFS("chr1", "r3", 1195, 1200, 5, 10, 0)]
    cl = c.fill_clusters([group], alns)[0]
    assert (cl.s, cl.e) == (1100, 1200)
    # r0,r1 span [1100,1200]; r3 starts at 1190 > min_s: counted in coverage but "unextended"; r2 has no SFS here
    assert [sr.name for sr in cl.subreads] == ["r0", "r1"] and c.unextended == 1
    assert cl.subreads[0].seq == ref[1100:1201] and cl.subreads[1].htag == 2
    assert (cl.cov0, cl.cov1, cl.cov2, cl.cov) == (2, 1, 1, 4)       # r2 and r3 untagged; mapq 5 and r4 not counted
    assert cl.reads == [(1, 1), (1, 2), (0, 3), (1, 3)]
    # a single-read group is dropped before any BAM access
    assert c.fill_clusters([[group[0]]], alns)[0].size() == 0 and c.small_clusters == 1
