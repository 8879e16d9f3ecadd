"""The index in HBM checked against its own text, independently of the builders (at product
scale the oracle was built from the BWT of the index under test).  svdss_index_verify_device (csrc/index_verify.hip)
compares adjacent suffix-array rows as strings, BWT[i] with text[SA[i]-1], rank-block counters, '$' rows and the symbol
histogram; it shares no code with index_build.cpp / index_gpu.hip.  Here: it accepts what both builders produce, it
catches every kind of damage, and BASELINE config 2's index is the host builder's byte for byte.  tests/test_scale_gpu.py
runs it on the 2.2e9- and 6.18e9-symbol indexes before it hands their BWT to the oracle (`OracleFMD.from_bwt(ix.bwt())`),
and bench.py on the index it measures, so that the checker no longer inherits an unchecked index.  The index
is what rb3_fmi_restore returns in the reference (ping_pong.cpp:245)."""
import struct

import numpy as np
import pytest

import svdss_amd
from svdss_amd import synth

pytestmark = pytest.mark.gpu

HDR = 8 + 8 + 56 + 8 + 8 + 4 * 4          # FileHeader of index_build.cpp


def _clean(v, n):
    assert v["rows"] == n and v["first_bad"] == -1, v
    assert v["bad_order"] == v["bad_bwt"] == v["bad_range"] == v["bad_block"] == v["bad_dollar"] == 0, v


def _ref_small():
    ref = synth.make_reference([180000, 90000, 700], seed=33, repeat_frac=0.5, divergence=0.0005, n_runs=(500, 30))
    ref.append(np.tile(np.array([1, 2, 3, 4], np.uint8), 3000))
    return ref


@pytest.mark.parametrize("where", ["host", "hbm", "hbm_pieces", "hbm_wide"])
def test_accepts_what_the_builders_produce(monkeypatch, where):
    ref = _ref_small()
    if where == "host":
        monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    if where == "hbm_pieces":
        monkeypatch.setenv("SVDSS_SA_PIECE", "20000")
    if where == "hbm_wide":
        monkeypatch.setenv("SVDSS_FORCE_SA64", "1")
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    v = ix.verify()
    _clean(v, ix.size)
    assert v["max_lcp"] >= 11000           # the tandem repeat: rows compared over thousands of symbols
    assert ix.verify(stride=7)["rows"] == (ix.size + 6) // 7


def test_catches_every_kind_of_damage(monkeypatch, tmp_path):
    """A saved index with one thing wrong at a time, loaded and made resident: two suffix-array rows swapped, an entry
    out of range, a duplicated entry, a BWT bit, a block counter, a '$' row, a text symbol."""
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    ref = synth.make_reference([60000, 20000], seed=4, repeat_frac=0.2, n_runs=(50,))
    good = svdss_amd.FMDIndex.build(ref)
    path = tmp_path / "good.idx"
    good.save(str(path))
    raw = path.read_bytes()
    n, = struct.unpack_from("<q", raw, 8)
    n_blocks, n_dollar = struct.unpack_from("<qq", raw, 8 + 8 + 56)
    sa_wide, = struct.unpack_from("<i", raw, 8 + 8 + 56 + 16 + 8)
    assert n == good.size and not sa_wide
    o_blocks = HDR
    o_dollar = o_blocks + 64 * n_blocks
    o_text = o_dollar + 8 * n_dollar
    o_sa = o_text + n
    assert len(raw) == o_sa + 4 * n
    sa = np.frombuffer(raw, np.uint32, n, o_sa)

    def damaged(edit):
        b = bytearray(raw)
        edit(b)
        p = tmp_path / "bad.idx"
        p.write_bytes(b)
        ix = svdss_amd.FMDIndex.load(str(p)).to_device(0)
        return ix.verify()

    _clean(damaged(lambda b: None), n)
    row = 12345

    def swap(b):
        b[o_sa + 4 * row:o_sa + 4 * row + 4], b[o_sa + 4 * row + 4:o_sa + 4 * row + 8] = \
            b[o_sa + 4 * row + 4:o_sa + 4 * row + 8], b[o_sa + 4 * row:o_sa + 4 * row + 4]
    v = damaged(swap)
    assert v["bad_order"] >= 1 and v["first_bad"] in (row - 1, row, row + 1)
    # (an entry out of range and a '$' row out of order no longer reach the GPU: the loader checks the ranges of what
    # the kernels index with -- tests/test_host_io_robustness.py -- and refuses the file)
    from svdss_amd._lib import SvdssError
    with pytest.raises(SvdssError):
        damaged(lambda b: struct.pack_into("<I", b, o_sa + 4 * 777, n + 5))
    v = damaged(lambda b: struct.pack_into("<I", b, o_sa + 4 * 5000, int(sa[5001])))
    assert v["bad_order"] >= 1
    # a BWT bit plane of block 40, quarter 1 (p0 of row 40*128 + 32 + 3)
    v = damaged(lambda b: b.__setitem__(o_blocks + 64 * 40 + 16 + 4, b[o_blocks + 64 * 40 + 16 + 4] ^ 8))
    assert v["bad_bwt"] >= 1 and v["bad_block"] >= 1
    # the counter of C in block 100
    v = damaged(lambda b: struct.pack_into("<I", b, o_blocks + 64 * 100 + 16, struct.unpack_from("<I", b, o_blocks + 64 * 100 + 16)[0] + 1))
    assert v["bad_block"] >= 1 and v["bad_bwt"] == 0 and v["bad_order"] == 0
    v = damaged(lambda b: struct.pack_into("<q", b, o_dollar + 8, struct.unpack_from("<q", b, o_dollar + 8)[0] + 1))
    assert v["bad_dollar"] >= 1
    with pytest.raises(SvdssError):      # the list out of order: refused by the loader
        damaged(lambda b: struct.pack_into("<q", b, o_dollar + 8, struct.unpack_from("<q", b, o_dollar)[0]))
    # a text symbol: the rows of the suffixes through it are out of order now, and some BWT symbol disagrees
    tpos = 30000
    v = damaged(lambda b: b.__setitem__(o_text + tpos, 1 + (b[o_text + tpos] % 4)))
    assert v["bad_order"] + v["bad_bwt"] >= 1 and v["bad_block"] >= 1


def test_chr20_length_index_built_in_hbm_is_the_host_builders(monkeypatch, tmp_path):
    """BASELINE config 2's reference (64,444,167 bp): the index built in HBM verifies row by row AND is byte for byte
    the file the host builder writes (until now compared at <= 270 kb only)."""
    ref = synth.make_reference([64_444_167], seed=11)
    g = svdss_amd.FMDIndex.build(ref, device=0)
    _clean(g.verify(), g.size)
    g.save(str(tmp_path / "gpu.idx"))
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    svdss_amd.FMDIndex.build(ref, threads=16).save(str(tmp_path / "cpu.idx"))
    import hashlib
    def digest(p):
        h = hashlib.sha256()
        with open(p, "rb") as f:
            while True:
                b = f.read(1 << 24)
                if not b:
                    break
                h.update(b)
        return h.hexdigest()
    assert digest(tmp_path / "gpu.idx") == digest(tmp_path / "cpu.idx")
