"""An `.fmd` with NO `.svdss` beside it -- what a user of upstream `SVDSS index` / ropebwt3 hands to `SVDSS search`
(run_svdss:140-147, rb3_fmi_restore at ping_pong.cpp:245) -- on the GPU box: decode the rld0 runs, recover the
records by LF walks, rebuild the index in HBM, verify it against its text, search, compare with the oracle built from
the records themselves.  The file is written by the test's own Python encoder with the sentinels in ropebwt's order
(tests/rld0_py.py), not by csrc/rld0.cpp.  ropebwt3 is not available: the format reading is [UPSTREAM-UNVERIFIED]."""
import os
import subprocess

import numpy as np
import pytest

import svdss_amd
from svdss_amd import synth
from svdss_amd._lib import SvdssError
from tests import oracle_lib as O, rld0_py
from tests.common import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "svdss_amd", "SVDSS")


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = tmp_path_factory.mktemp("fmd")
    ref = synth.make_reference([120000, 40000, 300], seed=71, repeat_frac=0.3, n_runs=(200,))
    strings = []
    for c in ref:
        strings += [c, synth.revcomp(c)]
    (d / "up.fmd").write_bytes(rld0_py.encode_rld0(rld0_py.collection_bwt(strings)))
    hap, _ = synth.implant_svs(ref[:2], 6, seed=72, min_len=50, max_len=400)
    flat, offs, _ = synth.simulate_reads(hap, 300, 4000, 0.005, seed=73)
    return d, ref, flat, offs


def test_import_rebuilds_in_hbm_and_searches_like_the_oracle(case):
    d, ref, flat, offs = case
    assert not os.path.exists(str(d / "up.fmd") + ".svdss")
    ix = svdss_amd.FMDIndex.load(str(d / "up.fmd")).to_device(0)
    v = ix.verify()
    assert v["rows"] == ix.size and v["first_bad"] == -1 and v["bad_order"] == v["bad_bwt"] == v["bad_block"] == 0, v
    fm = O.OracleFMD.build(ref)                      # from the records, not from the imported index
    for assemble in (False, True):
        pp = svdss_amd.PingPong(ix, assemble=assemble)
        got = pp.ping_pong_search(flat, offs)
        pp.close()
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        assert (got.counts == c).all() and (got.n_ext == e).all() and (got.qs == q).all() and (got.len == l).all()
    assert c.sum() > 300


def test_binary_takes_the_fmd_alone(case, tmp_path):
    """`SVDSS search --index up.fmd` without a sidecar writes the text it writes with the index `SVDSS index` built."""
    d, ref, flat, offs = case
    fa, fq = tmp_path / "ref.fa", tmp_path / "reads.fq"
    with open(fa, "w") as fh:
        for i, c in enumerate(ref):
            fh.write(f">c{i}\n{synth.to_ascii(c)}\n")
    with open(fq, "w") as fh:
        for i in range(len(offs) - 1):
            s = synth.to_ascii(flat[offs[i]:offs[i + 1]])
            fh.write(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n")
    own = tmp_path / "own.fmd"
    r = subprocess.run([BIN, "index", "-t", "4", "-d", str(fa), "-o", str(own)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(str(own) + ".svdss")
    a = subprocess.run([BIN, "search", "--index", str(own), "--fastx", str(fq), "--threads", "4"], capture_output=True, text=True, timeout=600)
    b = subprocess.run([BIN, "search", "--index", str(d / "up.fmd"), "--fastx", str(fq), "--threads", "4"], capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    assert a.stdout == b.stdout and a.stdout.count("\n") > 300
    # three replicas made at once from the handle of an imported .fmd (no records: text and suffix array travel through
    # the source's host copy, which the replicating threads share -- ADVICE r3: guarded by one mutex now)
    e = subprocess.run([BIN, "search", "--index", str(d / "up.fmd"), "--fastx", str(fq), "--threads", "4", "--gpus", "3"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, SVDSS_GPUS_OVERSUBSCRIBE="1"))
    assert e.returncode == 0, e.stderr
    assert e.stdout == a.stdout
    # and the .fmd `SVDSS index` wrote, with its sidecar taken away
    os.remove(str(own) + ".svdss")
    c = subprocess.run([BIN, "search", "--index", str(own), "--fastx", str(fq), "--threads", "4"], capture_output=True, text=True, timeout=600)
    assert c.returncode == 0 and c.stdout == a.stdout


def test_fmr_is_diagnosed_by_the_binary(tmp_path):
    (tmp_path / "x.fmr").write_bytes(bytes(range(1, 200)))
    (tmp_path / "r.fq").write_text("@r\nACGT\n+\nIIII\n")
    r = subprocess.run([BIN, "search", "--index", str(tmp_path / "x.fmr"), "--fastx", str(tmp_path / "r.fq")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and ".fmr" in r.stderr


def test_records_sidecar_rebuilds_in_hbm(case, tmp_path):
    """`<fmd>.svdss` as `SVDSS index` writes it since round 3 holds the records only: load() returns at once, the index
    is built where it is made resident (and a replica where IT is), verifies against its text and searches like the
    oracle; the full layout (SVDSS_INDEX_FULL=1 / svdss_index_save) restores to the same bytes."""
    d, ref, flat, offs = case
    full = svdss_amd.FMDIndex.build(ref)
    full.save_fmd(str(tmp_path / "r.fmd"))
    full.save_records(str(tmp_path / "r.fmd.svdss"))
    lazy = svdss_amd.FMDIndex.load(str(tmp_path / "r.fmd"))
    assert lazy.size == full.size and (lazy.acc == full.acc).all()
    lazy.to_device(0)
    assert lazy.kmer_k > 0
    v = lazy.verify()
    assert v["rows"] == lazy.size and v["first_bad"] == -1 and v["bad_order"] == v["bad_bwt"] == v["bad_block"] == 0, v
    lazy.save(str(tmp_path / "a.idx"))
    full.save(str(tmp_path / "b.idx"))
    assert (tmp_path / "a.idx").read_bytes() == (tmp_path / "b.idx").read_bytes()
    fm = O.OracleFMD.build(ref)
    c, q, l, e = fm.search_batch(flat, offs, True)
    pp = svdss_amd.PingPong(lazy, assemble=True)
    got = pp.ping_pong_search(flat, offs)
    pp.close()
    assert (got.counts == c).all() and (got.n_ext == e).all() and (got.qs == q).all() and (got.len == l).all()
    # a replica of an index that came from records is built from the records too
    import ctypes as C
    from svdss_amd._lib import check, lib
    h = C.c_void_p()
    check(lib.svdss_index_replicate(lazy._h, 0, C.byref(h)), "svdss_index_replicate")
    rep = svdss_amd.FMDIndex(h.value)
    assert rep.kmer_k == lazy.kmer_k and rep.verify()["first_bad"] == -1
    pp = svdss_amd.PingPong(rep, assemble=True)
    got = pp.ping_pong_search(flat, offs)
    pp.close()
    assert (got.counts == c).all() and (got.qs == q).all() and (got.len == l).all()
