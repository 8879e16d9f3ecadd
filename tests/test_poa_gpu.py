"""POA consensus kernel against its specification (oracle/svdss_oracle_poa.c): bit-exact consensus,
plus the invariants and the stated tolerance vs the truth at SURVEY 2.3 K3 sizes."""
import numpy as np
import pytest

from tests.mirror import caller
from tests import oracle_lib as O
from tests.test_oracle_poa import edit_distance, mutate

pytestmark = pytest.mark.gpu
LET = np.frombuffer(b"ACGTN", dtype=np.uint8)


def _to_str(a):
    return bytes(LET[a]).decode()


def test_clusters_match_oracle_bit_exact():
    rng = np.random.default_rng(11)
    clusters = []
    for k in range(60):
        length = int(rng.integers(150, 1200))
        t = rng.integers(0, 4, size=length).astype(np.uint8)
        if k % 6 == 0:                                    # a haplotype-specific indel in part of the reads
            alt = np.concatenate([t[:length // 2], rng.integers(0, 4, size=int(rng.integers(30, 200))).astype(np.uint8),
                                  t[length // 2:]])
        else:
            alt = t
        n = int(rng.integers(2, 30))
        reads = [mutate(rng, alt if (i % 3 == 0) else t, float(rng.choice([0.005, 0.02, 0.05]))) for i in range(n)]
        if k % 10 == 3:
            reads[1][5] = 4                               # an N
        clusters.append(reads)
    clusters += [[], [np.array([0, 1, 2, 3], np.uint8)], [np.zeros(0, np.uint8), np.array([1, 1], np.uint8)]]
    got, stats = caller.run_poa(clusters)
    for reads, g in zip(clusters, got):
        assert g == _to_str(O.poa_consensus(reads))
    assert stats["cells"] > 0 and stats["kernel_ms"] > 0


def test_invariants_and_tolerance():
    rng = np.random.default_rng(12)
    t = rng.integers(0, 4, size=3000).astype(np.uint8)
    b = t.copy(); b[1500] = (b[1500] + 1) % 4
    ins = np.concatenate([t[:1000], rng.integers(0, 4, size=300).astype(np.uint8), t[1000:]])
    dl = np.concatenate([t[:1200], t[1700:]])            # 500-bp deletion: band must be abandoned / followed
    noisy = [mutate(rng, t, 0.01) for _ in range(20)]
    got, _ = caller.run_poa([[t, t, t], [t, b, b], [t, ins, ins, ins, t], [t, dl, dl], noisy, [t]])
    assert got[0] == _to_str(t) and got[1] == _to_str(b) and got[2] == _to_str(ins) and got[3] == _to_str(dl)
    assert edit_distance(list(got[4].encode()), list(_to_str(t).encode())) <= max(2, int(0.005 * 3000))
    assert got[5] == _to_str(t)


def test_strings_and_large_cluster():
    rng = np.random.default_rng(13)
    t = rng.integers(0, 4, size=6000).astype(np.uint8)
    reads = [mutate(rng, t, 0.005) for _ in range(12)]
    got, stats = caller.run_poa([[_to_str(r) for r in reads], ["ACGTACGTAC", "ACGTTCGTAC", "ACGTTCGTAC"]])
    assert got[0] == _to_str(O.poa_consensus(reads))
    assert edit_distance(list(got[0].encode()), list(_to_str(t).encode())) <= 30
    assert got[1] == "ACGTTCGTAC"


def _mixed_clusters(seed, n_clusters):
    rng = np.random.default_rng(seed)
    clusters = []
    for k in range(n_clusters):
        length = int(rng.integers(200, 900))
        t = rng.integers(0, 4, size=length).astype(np.uint8)
        if k % 3 == 0:      # some reads skip a stretch: a long deletion edge whose source row leaves the LDS ring
            cut = int(rng.integers(20, 120))
            alt = np.concatenate([t[:length // 3], t[length // 3 + cut:]])
        elif k % 3 == 1:    # some reads carry an insertion
            alt = np.concatenate([t[:length // 2], rng.integers(0, 4, size=int(rng.integers(20, 150))).astype(np.uint8),
                                  t[length // 2:]])
        else:
            alt = t
        n = int(rng.integers(3, 16))
        clusters.append([mutate(rng, alt if (i % 2) else t, float(rng.choice([0.002, 0.01, 0.03]))) for i in range(n)])
    return clusters


def test_lds_and_hbm_kernels_agree_with_the_oracle(monkeypatch):
    """The LDS-resident kernel is the fast path, the HBM kernel its fallback: both must be the specification."""
    clusters = _mixed_clusters(21, 40)
    want = [_to_str(O.poa_consensus(reads)) for reads in clusters]
    got, stats = caller.run_poa(clusters)
    assert got == want
    assert stats["hbm"] <= len(clusters) // 4          # the fast path is the one that ran
    monkeypatch.setenv("SVDSS_POA_HBM", "1")
    got, stats = caller.run_poa(clusters)
    assert got == want and stats["hbm"] == len(clusters)
    monkeypatch.delenv("SVDSS_POA_HBM")
    monkeypatch.setenv("SVDSS_POA_NC", "100")           # LDS graphs sized for the longest read only: most outgrow
    got, stats = caller.run_poa(clusters)               # it and are redone in the roomier second LDS round
    assert got == want


def test_node_with_many_predecessors_falls_back():
    rng = np.random.default_rng(22)
    t = rng.integers(0, 4, size=300).astype(np.uint8)
    reads = [t]
    for i in range(12):     # twelve different insertions at one place: the node after it gets 13 predecessors
        reads.append(np.concatenate([t[:150], rng.integers(0, 4, size=10 + i).astype(np.uint8), t[150:]]))
    # nine deletions of different lengths that end at one node: ten predecessors -- more than a row descriptor of
    # poa_quad.hip (7) or a slow row of poa_wave.hip (8) holds
    dels = [t] + [np.concatenate([t[:150 - k], t[150:]]) for k in range(1, 10)]
    got, stats = caller.run_poa([reads, [t, t], dels])
    assert got[0] == _to_str(O.poa_consensus(reads)) and got[1] == _to_str(t) and got[2] == _to_str(O.poa_consensus(dels))
    # (which stage finishes such a cluster depends on where the alignments put the gaps: poa_quad.hip hands a node with more
    # than 7 predecessors on, poa_wave.hip one with more than 8; tests/test_poa_quad_emu.py pins the hand-over itself)
    assert stats["quad_back"] >= 0 and stats["hbm"] >= 0


def test_band_whose_end_moves_left():
    """Found by tests/fuzz_gpu.py (seed 7, iteration 105; the cluster is tests/golden/poa_fuzz_seed7_105.npz): a short
    read of homopolymer runs against the graph of a longer unrelated one.  The row maximum jumps, the band's end moves
    LEFT and right again, so a column can leave the band and come back: what it held (E1 / E2 as well as H) must read
    as "no path" in between.  Also more reads of that kind, and the band standing still at the last column for a
    hundred rows (query shorter than the graph)."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "poa_fuzz_seed7_105.npz"))
    flat, off = d["reads_flat"], d["reads_off"]
    reads = [np.ascontiguousarray(flat[off[i]:off[i + 1]], dtype=np.uint8) for i in range(len(off) - 1)]
    assert [len(r) for r in reads] == [186, 0, 83]
    rng = np.random.default_rng(105)
    clusters = [reads, [reads[0], reads[2]]]
    for k in range(40):
        long_ = rng.integers(0, 4, size=int(rng.integers(100, 900)), dtype=np.uint8)
        runs = rng.geometric(0.25, size=400)
        short = np.repeat(rng.integers(0, 4, size=400, dtype=np.uint8), runs)[:int(rng.integers(20, len(long_)))]
        other = mutate(rng, long_, 0.1)
        clusters.append([long_, short, other] if k % 2 else [short, long_, short[::-1].copy(), other])
    got, stats = caller.run_poa(clusters)
    for k, (cl, g) in enumerate(zip(clusters, got)):
        assert g == _to_str(O.poa_consensus(cl)), k
    assert stats["hbm"] <= 2
