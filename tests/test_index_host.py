"""Host FM-index builder of the product (svdss_index_build) against the oracle's
independent comparison-sort FMD: same acc, same interval sizes, same symbol
multiset; save/load round trip; argument errors."""
import ctypes as C
import os

import numpy as np
import pytest

import svdss_amd
from svdss_amd import synth
from svdss_amd._lib import lib
from tests import oracle_lib as O


@pytest.fixture(scope="module")
def pair():
    ref = synth.make_reference([120000, 50000, 7], seed=21, repeat_frac=0.1, n_runs=(300, 33))
    return ref, svdss_amd.FMDIndex.build(ref, threads=4), O.OracleFMD.build(ref)


def test_acc_and_size(pair):
    ref, ix, fm = pair
    assert ix.size == fm.n == sum(2 * (len(c) + 1) for c in ref)
    assert (ix.acc == fm.acc).all()
    assert ix.device_bytes >= (ix.size // 128 + 1) * 64


def test_bwt_is_a_permutation_of_the_text(pair):
    ref, ix, fm = pair
    b = ix.bwt()
    assert (np.bincount(b, minlength=6) == np.bincount(fm.bwt(), minlength=6)).all()
    assert (np.bincount(b, minlength=6) == np.diff(ix.acc)).all()


def test_interval_sizes_match_oracle(pair):
    ref, ix, fm = pair
    rng = np.random.default_rng(5)
    for _ in range(300):
        ci = int(rng.integers(0, 2))
        s = int(rng.integers(0, len(ref[ci]) - 64))
        l = int(rng.integers(1, 60))
        w = ref[ci][s:s + l].copy()
        if rng.random() < 0.3:
            w[int(rng.integers(0, l))] = int(rng.integers(1, 6))
        if rng.random() < 0.5:
            w = synth.revcomp(w)
        assert ix.count(w) == fm.count(w)
    # inside the N runs and the 7-base contig
    assert ix.count(np.full(200, 5, np.uint8)) == fm.count(np.full(200, 5, np.uint8)) > 0
    assert ix.count(ref[2]) == fm.count(ref[2]) >= 1


def test_thread_count_does_not_change_counts():
    ref = synth.make_reference([40000], seed=3, repeat_frac=0.2)
    a = svdss_amd.FMDIndex.build(ref, threads=1)
    b = svdss_amd.FMDIndex.build(ref, threads=8)
    assert (a.acc == b.acc).all()
    rng = np.random.default_rng(1)
    for _ in range(100):
        s = int(rng.integers(0, 39000))
        w = ref[0][s:s + int(rng.integers(1, 50))]
        assert a.count(w) == b.count(w)


def test_save_load_roundtrip(pair, tmp_path):
    ref, ix, fm = pair
    p = str(tmp_path / "ref.fa.fmd")
    ix.save(p)
    jx = svdss_amd.FMDIndex.load(p)
    assert jx.size == ix.size and (jx.acc == ix.acc).all()
    assert (jx.bwt() == ix.bwt()).all()
    w = ref[0][100:140]
    assert jx.count(w) == ix.count(w)
    with open(p, "r+b") as fh:
        fh.write(b"garbage!")
    with pytest.raises(svdss_amd.SvdssError):
        svdss_amd.FMDIndex.load(p)
    with pytest.raises(svdss_amd.SvdssError):
        svdss_amd.FMDIndex.load(str(tmp_path / "missing.fmd"))


def test_build_rejects_bad_input():
    bad = np.array([1, 2, 0, 3], dtype=np.uint8)  # '$' inside a record
    with pytest.raises(svdss_amd.SvdssError):
        svdss_amd.FMDIndex.build([bad])
    h = C.c_void_p()
    assert lib.svdss_index_build(None, None, 0, 1, C.byref(h)) != 0
