"""The kernel's per-lane code (sfs_core.h + sym_window.h + fmd_layout.h), run on
the CPU through tests/lane_emulator.cpp, against the oracle and the golden
vectors.  This validates the state machine, the register window over the read
and the streaming assembler without a GPU; the -m gpu tests repeat the same
comparisons through the real kernel."""
import numpy as np
import pytest

import svdss_amd
from tests import emulator_lib as E
from tests import oracle_lib as O
from tests.common import from_ascii, load_golden, small_workload, split


def test_golden_through_lane_code():
    for case in load_golden():
        contigs = [from_ascii(c) for c in case["contigs"]]
        ix = svdss_amd.FMDIndex.build(contigs, threads=2)
        reads = [from_ascii(r["read"]) for r in case["reads"]]
        flat, offs = svdss_amd.pack_reads(reads)
        c, q, l, e = E.search(ix, flat, offs, assemble=False)
        for got, rd, ne in zip(split(c, q, l), case["reads"], e.tolist()):
            assert [list(x) for x in got] == rd["sfs"]
            assert ne == rd["n_ext"]
        c, q, l, e = E.search(ix, flat, offs, assemble=True)
        for got, rd in zip(split(c, q, l), case["reads"]):
            assert [list(x) for x in got] == rd["assembled"]


@pytest.mark.parametrize("assemble", [False, True])
def test_random_reads_match_oracle(assemble):
    ref, hap, svs, flat, offs = small_workload(seed=31, n_reads=48, read_len=1200)
    ix = svdss_amd.FMDIndex.build(ref, threads=4)
    fm = O.OracleFMD.build(ref)
    c, q, l, e = E.search(ix, flat, offs, assemble)
    c2, q2, l2, e2 = fm.search_batch(flat, offs, assemble)
    assert (c == c2).all() and (e == e2).all()
    assert (q == q2).all() and (l == l2).all()
    assert c.sum() > 0


def test_unaligned_offsets_and_empty_reads():
    # reads start at arbitrary byte offsets of the concatenated buffer; some are empty
    ref, hap, svs, flat, offs = small_workload(seed=41, n_reads=10, read_len=300, ref_lens=(40000,))
    reads = [flat[offs[i]:offs[i + 1]] for i in range(10)]
    reads.insert(3, np.zeros(0, np.uint8))
    reads.append(np.zeros(0, np.uint8))
    reads.insert(0, ref[0][5:6])
    flat2, offs2 = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref, threads=2)
    fm = O.OracleFMD.build(ref)
    for assemble in (False, True):
        c, q, l, e = E.search(ix, flat2, offs2, assemble)
        c2, q2, l2, e2 = fm.search_batch(flat2, offs2, assemble)
        assert (c == c2).all() and (q == q2).all() and (l == l2).all() and (e == e2).all()
    assert c[4] == 0 and c[-1] == 0
