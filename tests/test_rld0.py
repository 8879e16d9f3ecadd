"""`.fmd` (ropebwt3 rld0): the writer against an independent pure-Python reading of the published
format and a hand-derived known answer, the reader against the writer, and the import path (decode -> strings ->
rebuild) against the index the records were exported from.  ropebwt3 itself is not available: [UPSTREAM-UNVERIFIED]."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import svdss_amd
from svdss_amd import synth
from svdss_amd._lib import SvdssError, lib


@pytest.fixture(autouse=True)
def _host_builder(monkeypatch):
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")    # these tests run without a GPU


def read_bwt(path):
    n = C.c_int64()
    rc = lib.svdss_fmd_read_bwt(str(path).encode(), None, 0, C.byref(n))
    assert rc == 0
    out = np.zeros(n.value, np.uint8)
    assert lib.svdss_fmd_read_bwt(str(path).encode(), out.ctypes.data, n.value, C.byref(n)) == 0
    return out


def py_decode(path):
    """rld0 as published (rld0.c): header, then small blocks of 8 words -- counts of the previous block (16- or 32-bit,
    type in the top two bits of the first word), then Elias-delta run lengths + 3-bit symbols, most significant bit
    first -- written from the format description, independently of csrc/rld0.cpp."""
    raw = open(path, "rb").read()
    assert raw[:4] == b"RLD\x03"
    a, k, n_frames = struct.unpack_from("<IQQ", raw, 4)
    asize, sbits = a >> 16, a & 0xffff
    assert asize == 6 and sbits == 3
    mcnt = struct.unpack_from("<6Q", raw, 24)
    words = struct.unpack_from(f"<{k}Q", raw, 24 + 48)
    frames = struct.unpack_from(f"<{n_frames * 7}Q", raw, 24 + 48 + 8 * k)
    assert len(raw) == 24 + 48 + 8 * k + 8 * 7 * n_frames
    out, total = [], sum(mcnt)
    block_counts = []
    for shead in range(0, k, 8):
        w0 = words[shead]
        typ = w0 >> 62
        hdr = b"".join(struct.pack("<Q", w) for w in words[shead:shead + 4])
        if typ == 0:
            h = struct.unpack_from("<7H", hdr)
            p = shead + 2
        else:
            h = [x & 0x3fffffff for x in struct.unpack_from("<7I", hdr)]
            p = shead + 4
        block_counts.append(h)
        stail = shead + 8 - (2 if (shead + 8) % (1 << 23) == 0 else 1)
        bits = "".join(f"{words[i]:064b}" for i in range(p, min(stail + 1, k)))
        i = 0
        while len(out) < total:
            z = 0
            while i + z < len(bits) and bits[i + z] == "0" and z < 6:
                z += 1
            if z >= 6 or i + z >= len(bits):
                break
            y = int(bits[i + z:i + 2 * z + 1], 2) - 1
            i += 2 * z + 1
            l = (1 << y) | (int(bits[i:i + y], 2) if y else 0)
            i += y
            c = int(bits[i:i + 3], 2)
            i += 3
            if c >= 6:
                break
            out.extend([c] * l)
    return np.array(out, np.uint8), mcnt, block_counts, frames


def test_known_answer_four_a(tmp_path):
    """BWT of the single record "AAA" and its reverse complement "TTT": text AAA$TTT$.  Sorted suffixes:
    $ (7), $TTT$ (3), A$.. (2), AA$ (1), AAA$ (0), T$ (6), TT$ (5), TTT$ (4) -> BWT = T A A A $ T T $.  Runs: (1,T) (3,A)
    (1,$) (2,T) (1,$).  Delta codes: 1 -> '1', 2 -> '0100', 3 -> '0101'; symbols T=100 A=001 $=000:
    1100 0101001 1000 0100100 1000 -> the first data word starts 1100 0101 0011 0000 1001 0010 00."""
    ix = svdss_amd.FMDIndex.build([np.array([1, 1, 1], np.uint8)])
    assert ix.bwt().tolist() == [4, 1, 1, 1, 0, 4, 4, 0]
    ix.save_fmd(str(tmp_path / "a.fmd"))
    raw = open(tmp_path / "a.fmd", "rb").read()
    a, k, n_frames = struct.unpack_from("<IQQ", raw, 4)
    assert (a >> 16, a & 0xffff) == (6, 3)
    assert struct.unpack_from("<6Q", raw, 24) == (2, 3, 0, 0, 3, 0)
    words = struct.unpack_from(f"<{k}Q", raw, 72)
    assert k == 10                              # one data block + the header of the block after it (2 words)
    assert words[0] == 0 and words[1] == 0      # first block: counts of "the block before" are zero
    assert words[2] >> (64 - 26) == 0b11000101001100001001001000
    assert words[2] & ((1 << 38) - 1) == 0 and all(w == 0 for w in words[3:8])
    # the closing header: total 8, $ 2, A 3, C 0, G 0, T 3, N 0 as 16-bit fields
    assert struct.unpack_from("<7H", struct.pack("<2Q", words[8], words[9])) == (8, 2, 3, 0, 0, 3, 0)
    assert (read_bwt(tmp_path / "a.fmd") == ix.bwt()).all()


@pytest.mark.parametrize("seed", [1, 2])
def test_writer_against_python_reader_and_reader_against_writer(tmp_path, seed):
    ref = synth.make_reference([30000, 9000, 40], seed=seed, repeat_frac=0.3, n_runs=(700, 20))
    ref.append(np.full(5000, 1, np.uint8))          # a long run: a run length above 2^12, a 32-bit block header
    ref.append(np.full(40000, 5, np.uint8))         # and one whose block holds >= 0x4000 symbols
    ix = svdss_amd.FMDIndex.build(ref, threads=4)
    path = tmp_path / "x.fmd"
    ix.save_fmd(str(path))
    bwt = ix.bwt()
    got, mcnt, block_counts, frames = py_decode(path)
    assert (got == bwt).all()
    assert list(mcnt) == np.bincount(bwt, minlength=6).tolist()
    assert any(sum(h[1:]) >= 0x4000 for h in block_counts)
    # every block header holds the symbol counts of the block before it; together they add up to the totals
    tot = np.zeros(6, np.int64)
    for h in block_counts:
        assert h[0] == sum(h[1:])
        tot += np.array(h[1:])
    assert tot.tolist() == list(mcnt)
    # frames: (block offset, counts before that block), non-decreasing, consistent with the block headers
    cum, at = {0: np.zeros(6, np.int64)}, np.zeros(6, np.int64)
    for b, h in enumerate(block_counts[1:], start=1):
        at = at + np.array(h[1:])
        cum[8 * b] = at.copy()
    fr = np.array(frames).reshape(-1, 7)
    assert (np.diff(fr[:, 0]) >= 0).all()
    for row in fr:
        assert (cum[int(row[0])] == row[1:]).all()
    assert (read_bwt(path) == bwt).all()


def test_import_rebuilds_an_equivalent_index(tmp_path):
    ref = synth.make_reference([50000, 20000, 300], seed=5, repeat_frac=0.2, n_runs=(100,))
    ix = svdss_amd.FMDIndex.build(ref, threads=4)
    ix.save_fmd(str(tmp_path / "r.fmd"))
    back = svdss_amd.FMDIndex.load(str(tmp_path / "r.fmd"))      # no .svdss beside it: decode, recover, rebuild
    assert back.size == ix.size and (back.acc == ix.acc).all()
    rng = np.random.default_rng(3)
    for _ in range(300):
        ci = int(rng.integers(0, 2))
        s = int(rng.integers(0, len(ref[ci]) - 70))
        w = ref[ci][s:s + int(rng.integers(1, 60))].copy()
        if rng.random() < 0.4:
            w[int(rng.integers(0, len(w)))] = int(rng.integers(1, 5))
        if rng.random() < 0.5:
            w = synth.revcomp(w)
        assert back.count(w) == ix.count(w)
    # (the rebuilt index concatenates the records in sentinel order, so its BWT is another valid BWT of the same
    # collection: same symbol counts, same interval sizes, not the same bytes)
    back.save_fmd(str(tmp_path / "r2.fmd"))
    assert np.bincount(read_bwt(tmp_path / "r2.fmd"), minlength=6).tolist() == np.bincount(ix.bwt(), minlength=6).tolist()


def test_own_layout_beside_the_fmd_is_preferred(tmp_path):
    ref = synth.make_reference([8000], seed=7)
    ix = svdss_amd.FMDIndex.build(ref)
    ix.save_fmd(str(tmp_path / "i.fmd"))
    ix.save(str(tmp_path / "i.fmd.svdss"))
    a = svdss_amd.FMDIndex.load(str(tmp_path / "i.fmd"))
    assert (a.bwt() == ix.bwt()).all()
    os.utime(tmp_path / "i.fmd.svdss", (1, 1))                  # older than the .fmd: not trusted, import instead
    b = svdss_amd.FMDIndex.load(str(tmp_path / "i.fmd"))
    assert b.size == ix.size and b.count(ref[0][100:140]) == ix.count(ref[0][100:140])


def test_rejects_what_is_not_a_both_strand_index(tmp_path):
    # a BWT whose strings do not come in reverse-complement pairs: "AC$" + "AC$" (two copies of one strand)
    # text AC$AC$ -> suffixes: $ (5), $AC$ (2), AC$ (3), AC$AC$ (0), C$ (4), C$AC$ (1) -> BWT C C $ $ A A
    from svdss_amd._lib import check
    bwt = np.array([2, 2, 0, 0, 1, 1], np.uint8)
    # write it through the library's writer by way of a fake index? the writer is reached through an index object only,
    # so craft the file with the test's own encoder: header + one block
    def delta(x):
        y = x.bit_length() - 1
        z = (y + 1).bit_length() - 1
        return "0" * z + format(y + 1, f"0{z + 1}b") + (format(x ^ (1 << y), f"0{y}b") if y else "")
    bits = "".join(delta(l) + format(c, "03b") for l, c in [(2, 2), (2, 0), (2, 1)])
    word = int(bits.ljust(64, "0"), 2)
    words = [0, 0, word, 0, 0, 0, 0, 0] + list(struct.unpack("<2Q", struct.pack("<7H", 6, 2, 2, 2, 0, 0, 0) + b"\0\0"))
    raw = b"RLD\x03" + struct.pack("<IQQ", 6 << 16 | 3, len(words), 2) + struct.pack("<6Q", 2, 2, 2, 0, 0, 0)
    raw += struct.pack(f"<{len(words)}Q", *words) + struct.pack("<14Q", *([0] * 14))
    (tmp_path / "bad.fmd").write_bytes(raw)
    assert read_bwt(tmp_path / "bad.fmd").tolist() == bwt.tolist()
    with pytest.raises(SvdssError):
        svdss_amd.FMDIndex.load(str(tmp_path / "bad.fmd"))
    (tmp_path / "junk.fmd").write_bytes(b"RLD\x03" + b"\1" * 10)
    with pytest.raises(SvdssError):
        svdss_amd.FMDIndex.load(str(tmp_path / "junk.fmd"))


def test_cache_of_another_fmd_is_not_used(tmp_path):
    """nothing but the mtime tied `<fmd>.svdss` to the .fmd.  A cache with other symbol counts (an .fmd
    swapped under a preserved mtime) or a truncated one must send the load through the import path."""
    ref_a = synth.make_reference([9000], seed=8)
    ref_b = synth.make_reference([7000], seed=9)
    a = svdss_amd.FMDIndex.build(ref_a)
    b = svdss_amd.FMDIndex.build(ref_b)
    a.save_fmd(str(tmp_path / "x.fmd"))
    b.save(str(tmp_path / "x.fmd.svdss"))                      # newer, but the layout of ANOTHER reference
    got = svdss_amd.FMDIndex.load(str(tmp_path / "x.fmd"))
    assert got.size == a.size and (got.acc == a.acc).all()
    assert got.count(ref_a[0][500:560]) == a.count(ref_a[0][500:560]) >= 1
    a.save(str(tmp_path / "x.fmd.svdss"))
    raw = (tmp_path / "x.fmd.svdss").read_bytes()
    (tmp_path / "x.fmd.svdss").write_bytes(raw[:len(raw) // 2])  # truncated cache: fall back, do not fail
    got = svdss_amd.FMDIndex.load(str(tmp_path / "x.fmd"))
    assert got.size == a.size and got.count(ref_a[0][100:160]) == a.count(ref_a[0][100:160])


def test_an_index_in_another_format_is_diagnosed(tmp_path):
    """rb3_fmi_restore (ping_pong.cpp:245) also takes ropebwt3's .fmr; this library does not and says what to do."""
    (tmp_path / "ref.fmr").write_bytes(b"\x01\x02\x03\x04" + bytes(200))
    with pytest.raises(SvdssError) as e:
        svdss_amd.FMDIndex.load(str(tmp_path / "ref.fmr"))
    assert ".fmr" in str(e.value) and "ropebwt3 build" in str(e.value)


def test_reader_takes_64_bit_block_headers_and_bounds_its_reads(tmp_path):
    """A type-2 block (64-bit counts, written when a block covers 2^30 symbols or more) leaves one data word at
    sbits = 3; a data length that is not a whole number of blocks must not be read past."""
    def delta(x):
        y = x.bit_length() - 1
        z = (y + 1).bit_length() - 1
        return "0" * z + format(y + 1, f"0{z + 1}b") + (format(x ^ (1 << y), f"0{y}b") if y else "")
    w_a = int((delta(5) + "001").ljust(64, "0"), 2)             # block 0: AAAAA
    w_b = int((delta(3) + "010" + delta(2) + "000").ljust(64, "0"), 2)   # block 1 (type 2): CCC $$
    blk0 = [0, 0, w_a, 0, 0, 0, 0, 0]
    blk1 = [5 | (2 << 62), 0, 5, 0, 0, 0, 0, w_b]               # counts of block 0: total 5, A 5
    words = blk0 + blk1 + [5, 2, 0, 3]                            # + the start of a closing header, cut short
    raw = b"RLD\x03" + struct.pack("<IQQ", 6 << 16 | 3, len(words), 1) + struct.pack("<6Q", 2, 5, 3, 0, 0, 0)
    raw += struct.pack(f"<{len(words)}Q", *words) + struct.pack("<7Q", *([0] * 7))
    (tmp_path / "t2.fmd").write_bytes(raw)
    assert read_bwt(tmp_path / "t2.fmd").tolist() == [1] * 5 + [2] * 3 + [0] * 2


def test_second_encoder_writes_the_same_bytes_and_ropebwt_sentinel_order_imports(tmp_path):
    """tests/rld0_py.py encodes rld0 in Python from the format description: byte-identical to csrc/rld0.cpp on the
    library's own BWT; and a BWT in ropebwt's sentinel order (string i's '$' before string j's -- not the order of
    this library's index, so not a file its writer could produce) imports to the index of the same records."""
    from tests import rld0_py
    ref = synth.make_reference([30000, 9000, 40], seed=3, repeat_frac=0.2, n_runs=(60,))
    jx = svdss_amd.FMDIndex.build(ref)
    jx.save_fmd(str(tmp_path / "own.fmd"))
    assert (tmp_path / "own.fmd").read_bytes() == rld0_py.encode_rld0(jx.bwt())
    strings = []
    for c in ref:
        strings += [c, synth.revcomp(c)]
    bwt = rld0_py.collection_bwt(strings)
    assert not (bwt == jx.bwt()).all() and np.bincount(bwt, minlength=6).tolist() == np.bincount(jx.bwt(), minlength=6).tolist()
    (tmp_path / "up.fmd").write_bytes(rld0_py.encode_rld0(bwt))
    assert (read_bwt(tmp_path / "up.fmd") == bwt).all()
    assert (py_decode(tmp_path / "up.fmd")[0] == bwt).all()
    ix = svdss_amd.FMDIndex.load(str(tmp_path / "up.fmd"))
    assert ix.size == jx.size and (ix.acc == jx.acc).all() and (ix.bwt() == jx.bwt()).all()
