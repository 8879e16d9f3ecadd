"""The several-sub-clusters-per-wavefront POA kernel (svdss_amd/csrc/poa_quad_core.h) on the CPU wave emulator
(tests/native/wave_emu.h) against the oracle: the same device source hipcc compiles for gfx950, with the cross-lane
primitives as meetings of 64 fibres.  Every sub-cluster the kernel finishes must be the specification bit for bit; what it
hands back (status 3 | reason << 8: the host gives those to poa_wave.hip) must be rare and for a stated reason."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from tests import poaq_emu_lib as Q
from tests.test_oracle_poa import mutate

VARIANTS = [(16, 3), (16, 4), (16, 5), (16, 7), (32, 2), (32, 3), (64, 1), (64, 2)]


def _check(clusters, gw, c, max_back=0, reasons=()):
    got, st, cells = Q.run(clusters, gw, c)
    back = 0
    for k, (cl, g) in enumerate(zip(clusters, got)):
        if g is None:
            back += 1
            assert (int(st[k]) & 255) == 3 and ((int(st[k]) >> 8) & 7) in reasons, (k, hex(int(st[k])))
            continue
        assert np.array_equal(g, O.poa_consensus(cl)), (k, gw, c)
    assert back <= max_back, (back, [hex(int(s)) for s in st if s])
    return back, cells


def _noisy(seed, n_clusters, lo, hi):
    rng = np.random.default_rng(seed)
    clusters = []
    for k in range(n_clusters):
        length = int(rng.integers(lo, hi))
        t = rng.integers(0, 4, size=length).astype(np.uint8)
        if k % 5 == 0:      # a haplotype-specific insertion in part of the reads
            alt = np.concatenate([t[:length // 2], rng.integers(0, 4, size=int(rng.integers(5, 40))).astype(np.uint8), t[length // 2:]])
        elif k % 5 == 1:    # some reads skip a stretch: a deletion edge whose source row has left the ring
            cut = int(rng.integers(3, 30))
            alt = np.concatenate([t[:length // 3], t[length // 3 + cut:]])
        else:
            alt = t
        n = int(rng.integers(1, 12))
        reads = [mutate(rng, alt if (i % 3 == 0) else t, float(rng.choice([0.005, 0.02, 0.06]))) for i in range(n)]
        if k % 7 == 3 and len(reads) > 1 and len(reads[1]) > 5:
            reads[1][5] = 4                               # an N
        clusters.append(reads)
    return clusters


@pytest.mark.parametrize("gw,c", VARIANTS)
def test_emulated_kernel_is_the_specification(gw, c):
    clusters = _noisy(100 + gw + c, 16, 30, 260)
    clusters += [[], [np.array([0, 1, 2, 3], np.uint8)], [np.zeros(0, np.uint8), np.array([1, 1], np.uint8)],
                 [np.array([2], np.uint8)] * 3]
    # wide rows (reason 3), a lost sink (4) and graphs beyond the first allocation (5) are poa_wave.hip's business
    _check(clusters, gw, c, max_back=6, reasons=(3, 4, 5))


def test_groups_of_a_wavefront_do_not_see_each_other():
    """Four very different sub-clusters in one wavefront (lengths 12 .. 400, 1 .. 9 reads): the lock-step rows, the shared
    loops and the per-group masks; then the same sub-clusters in another order."""
    rng = np.random.default_rng(7)
    cl = []
    for L, n in [(12, 9), (400, 3), (150, 1), (77, 6), (300, 2), (33, 4), (5, 5)]:
        t = rng.integers(0, 4, size=L).astype(np.uint8)
        cl.append([mutate(rng, t, 0.04) for _ in range(n)])
    _check(cl, 16, 4)
    _check(cl[::-1], 16, 4)
    _check(cl, 32, 2)


def test_far_predecessors_many_predecessors_and_a_moving_band():
    rng = np.random.default_rng(22)
    t = rng.integers(0, 4, size=200).astype(np.uint8)
    many = [t] + [np.concatenate([t[:100], rng.integers(0, 4, size=4 + i).astype(np.uint8), t[100:]]) for i in range(5)]
    toomany = [t] + [np.concatenate([t[:100 - k], t[100:]]) for k in range(1, 10)]   # nine deletions that end at one node
    dele = [t, np.concatenate([t[:60], t[100:]]), t, np.concatenate([t[:60], t[100:]]), t]
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "poa_fuzz_seed7_105.npz"))
    flat, off = d["reads_flat"], d["reads_off"]
    fz = [np.ascontiguousarray(flat[off[i]:off[i + 1]], dtype=np.uint8) for i in range(len(off) - 1)]
    clusters = [many, dele, fz, [fz[0], fz[2]]]
    for k in range(10):
        long_ = rng.integers(0, 4, size=int(rng.integers(60, 300)), dtype=np.uint8)
        runs = rng.geometric(0.25, size=200)
        short = np.repeat(rng.integers(0, 4, size=200, dtype=np.uint8), runs)[:int(rng.integers(20, len(long_)))]
        other = mutate(rng, long_, 0.1)
        clusters.append([long_, short, other] if k % 2 else [short, long_, short[::-1].copy(), other])
    # (an unrelated short read against a long graph usually loses the sink inside the band -- reason 4: the full-matrix
    # attempt is poa_wave.hip's --, but the rows on the way there are the ones the fixture is about)
    for gw, c in [(16, 4), (16, 7), (32, 3), (64, 2)]:
        back, _ = _check(clusters, gw, c, max_back=13, reasons=(3, 4, 5))
        got, st, _ = Q.run(clusters[:1], gw, c)
        assert got[0] is not None
    # a node with ten predecessors is more than a descriptor holds: handed back with reason 2
    got, st, _ = Q.run([toomany, [t, t]], 16, 4)
    assert got[0] is None and (int(st[0]) >> 8) & 7 == 2
    assert np.array_equal(got[1], t)


def test_bench_shaped_sub_clusters():
    """What bench.py's call-side workload looks like (CallWorkload): 15 reads of ~0.9 kb with 0.5 % substitutions, one
    wavefront of four; the emulated kernel must finish all of them."""
    rng = np.random.default_rng(5)
    cl = []
    for _ in range(4):
        t = rng.integers(0, 4, size=int(rng.integers(600, 900))).astype(np.uint8)
        reads = []
        for _ in range(15):
            r = t.copy()
            e = rng.random(len(r)) < 0.005
            r[e] = (r[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
            reads.append(r)
        cl.append(reads)
    back, cells = _check(cl, 16, 4)
    assert back == 0 and cells > 0
