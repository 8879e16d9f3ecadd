"""The input generators of tests/fuzz_gpu.py (the differential fuzzer that runs on the GPU box) stay runnable: every
kind of reference, read, cluster and byte mixture is drawn, the oracle takes what they produce -- degenerate contigs
and reads included --, and the zlib-side verdict on damaged streams is what the fuzzer expects."""
import zlib

import numpy as np

import svdss_amd
from tests import fuzz_gpu as F
from tests import oracle_lib as O


def test_references_and_reads_are_searchable_by_the_oracle():
    kinds = set()
    for it in range(12):
        rng = np.random.default_rng([5, it])
        contigs = [F._weird_contig(rng, int(rng.choice([60, 3000, 20000]))) for _ in range(int(rng.integers(1, 4)))]
        assert all(c.dtype == np.uint8 and len(c) >= 1 and c.min() >= 1 and c.max() <= 5 for c in contigs)
        reads = F._weird_reads(rng, contigs, 40)
        assert all(r.dtype == np.uint8 for r in reads)
        kinds.update(len(r) for r in reads if len(r) < 2)
        fm = O.OracleFMD.build(contigs)
        flat, offs = svdss_amd.pack_reads(reads)
        c, q, l, e = fm.search_batch(flat, offs, True)
        assert len(c) == len(reads) and (q + l <= np.repeat(np.diff(offs), c)).all()
    assert 0 in kinds and 1 in kinds            # empty and one-base reads were drawn


def test_clusters_have_a_consensus():
    rng = np.random.default_rng(6)
    for _ in range(6):
        reads = F._weird_cluster(rng)
        assert all(r.dtype == np.uint8 and (len(r) == 0 or r.max() <= 4) for r in reads)
        cons = O.poa_consensus(reads)
        assert cons.dtype == np.uint8 and (len(cons) == 0 or cons.max() <= 4)


def test_byte_mixtures_and_the_zlib_verdict():
    rng = np.random.default_rng(7)
    data = F._mixture(rng, 50000)
    assert len(data) == 50000
    raw = zlib.compress(data, 6)[2:-4]
    assert F._zlib_inflate(raw, len(data)) == (True, data)
    assert F._zlib_inflate(raw, len(data) - 1)[0] is False          # wrong ISIZE
    assert F._zlib_inflate(raw[:len(raw) // 2], len(data))[0] is False
    assert F._zlib_inflate(b"\x07" + raw[1:], len(data))[0] is False  # block type 3
