"""BGZF blocks deflated on the GPU (csrc/deflate.hip, svdss_bgzf_deflate; SURVEY 8(f)3 "later GPU deflate": the
writing half of `SVDSS smooth`'s BAM stream, /root/reference/smoother.cpp:441-494).  The encoder's output is pinned to
the standard the way the inflater is: every member must inflate, with zlib (the library htslib inflates with), to exactly
the bytes that went in -- block types, code lengths folded back to 15 bits, stored quarters, sizes 1 .. 0xff00, many
blocks with a short tail --, libdeflate (what htslib uses when built with it) must agree on a sample of the members,
Python's gzip must read the whole stream as BGZF, the footers must hold, and the GPU
inflater (csrc/inflate.hip) must read what the GPU encoder wrote.  Through the binary: `SVDSS smooth` writes the same
records whether the GPU or the host's deflate packed them."""
import gzip
import io
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from svdss_amd.bgzf import bgzf_blocks, gpu_deflate, gpu_inflate
from tests.common import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "svdss_amd", "SVDSS")


def members(stream):
    """[(header dict, raw deflate bytes, crc, isize)] of a BGZF stream, by the SAM specification's layout"""
    out, pos = [], 0
    while pos < len(stream):
        assert stream[pos:pos + 4] == b"\x1f\x8b\x08\x04"
        xlen, = struct.unpack_from("<H", stream, pos + 10)
        assert xlen == 6 and stream[pos + 12:pos + 16] == b"BC\x02\x00"
        bsize, = struct.unpack_from("<H", stream, pos + 16)
        raw = stream[pos + 18:pos + bsize + 1 - 8]
        crc, isize = struct.unpack_from("<II", stream, pos + bsize + 1 - 8)
        out.append((raw, crc, isize))
        pos += bsize + 1
    assert pos == len(stream)
    return out


def _libdeflate():
    """libdeflate's decompressor (what htslib inflates with when built with it; stricter than zlib about incomplete
    codes) through ctypes, or None when the shared library is not on the machine"""
    import ctypes as C
    for name in ("libdeflate.so.0", "libdeflate.so"):
        try:
            lib = C.CDLL(name)
        except OSError:
            continue
        lib.libdeflate_alloc_decompressor.restype = C.c_void_p
        lib.libdeflate_deflate_decompress.restype = C.c_int
        lib.libdeflate_deflate_decompress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.libdeflate_free_decompressor.argtypes = [C.c_void_p]
        return lib
    return None


def libdeflate_inflate(raw, isize):
    import ctypes as C
    lib = _libdeflate()
    if lib is None:
        return None
    d = lib.libdeflate_alloc_decompressor()
    out = C.create_string_buffer(max(isize, 1))
    got = C.c_size_t()
    rc = lib.libdeflate_deflate_decompress(d, raw, len(raw), out, isize, C.byref(got))
    lib.libdeflate_free_decompressor(d)
    assert rc == 0, rc                                      # LIBDEFLATE_SUCCESS
    return out.raw[:got.value]


def check_roundtrip(data, block_bytes=0xff00):
    data = bytes(data)
    stream = gpu_deflate(data, block_bytes)
    ms = members(stream)
    assert len(ms) == (len(data) + block_bytes - 1) // block_bytes
    back = []
    for i, (raw, crc, isize) in enumerate(ms):
        d = zlib.decompressobj(-15)
        got = d.decompress(raw) + d.flush()
        assert d.eof and d.unused_data == b""               # one complete stream, nothing behind it
        want = data[i * block_bytes:(i + 1) * block_bytes]
        assert got == want, (i, len(got), len(want))
        if i % 7 == 0:                                       # a second, independent inflater
            ld = libdeflate_inflate(raw, len(want))
            assert ld is None or ld == want, i
        assert isize == len(want) and crc == zlib.crc32(want) & 0xffffffff
        assert 18 + len(raw) + 8 <= 65536                    # a BGZF member
        back.append(got)
    assert gzip.GzipFile(fileobj=io.BytesIO(stream)).read() == data   # concatenated gzip members
    return stream, ms


def test_every_kind_of_block_inflates_with_zlib():
    rng = np.random.default_rng(7)
    fib = [1, 1]
    while len(fib) < 32:
        fib.append(fib[-1] + fib[-2])
    cases = {
        "one byte": b"x",
        "two bytes": b"ab",
        "all the same": bytes([9]) * 0xff00,
        "random (stored)": bytes(rng.integers(0, 256, size=0xff00, dtype=np.uint8)),
        "packed bases": bytes(rng.choice([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88],
                                         size=50000).astype(np.uint8)),
        "qualities": bytes(rng.integers(20, 60, size=0xff00, dtype=np.uint8)),
        "binned qualities": bytes(np.array([3, 10, 17, 22, 27, 33, 40], np.uint8)[rng.integers(0, 7, size=40000)]),
        "skewed (geometric)": bytes(np.minimum(255, rng.geometric(0.03, size=0xff00)).astype(np.uint8)),
        "code lengths past 15 bits": b"".join(bytes([i]) * min(c, 9000) for i, c in enumerate(fib))[:0xff00],
        "text": (b"@SQ\tSN:chr1\tLN:248956422\n" * 4000)[:0xff00],
        "quarter boundaries": bytes(range(256)) * 3 + b"z",
    }
    for name, data in cases.items():
        stream, ms = check_roundtrip(data)
        if name in ("packed bases", "binned qualities", "all the same", "text"):
            assert len(stream) < 0.62 * len(data), (name, len(stream), len(data))
        if name == "random (stored)":
            assert len(data) < len(stream) <= len(data) + 26 + 6 * 4
    # sizes around the lane / quarter arithmetic
    for n in (3, 63, 64, 65, 255, 256, 257, 1019, 1020, 1021, 4 * 16320 - 1, 0xff00 - 1):
        check_roundtrip(bytes(rng.integers(60, 70, size=n, dtype=np.uint8)))


def test_runs_become_matches_at_distance_one():
    """Second session of round 5: runs of 4-258 equal bytes are coded as matches at distance 1 (absent qualities are
    15,000 x 0xff per read; as literals a run costs a bit per byte and this repo's own inflater reads one-bit codes at a
    quarter of its rate).  Every inflater must give the bytes back; runs of every length around the thresholds (3, 4,
    258, 259, 261, 262), runs that begin or end at a lane's slice (255 bytes) or a quarter block (16,320 bytes), a run
    as the whole member, runs of several values, and records as the chain bench writes them."""
    rng = np.random.default_rng(11)
    # run lengths 1..600 of a changing byte, separated by one different byte: every remainder modulo 258 and every
    # position relative to the lanes' slices
    parts = []
    for r in list(range(1, 300)) + [515, 516, 517, 518, 519, 520, 600]:
        parts.append(bytes([65 + r % 7]) * r + bytes([200 + r % 5]))
    data = b"".join(parts)
    for off in (0, 1, 254, 255):                       # shifted against the slices
        check_roundtrip((b"x" * off + data)[:0xff00])
    check_roundtrip(bytes([0xff]) * 0xff00)
    check_roundtrip(bytes([0xff]) * (4 * 16320 - 1))
    check_roundtrip(b"a" + bytes([7]) * 16319 + bytes([7]) * 5 + b"b" * 16315 + b"b" * 16320 + b"c" * 3)   # runs across the quarters
    for n in (1, 2, 3, 4, 5, 257, 258, 259, 260, 261, 262, 263, 516, 517, 1000):
        check_roundtrip(bytes([1]) * n)
        check_roundtrip(b"q" + bytes([1]) * n + b"r")
    # the chain bench's records: random packed bases, qualities absent (0xff), a tag
    rec = []
    for k in range(40):
        l = int(rng.integers(9000, 16000))
        rec.append(b"read%07d\0" % k + bytes(rng.choice([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88],
                                                         size=l // 2).astype(np.uint8)) + b"\xff" * l + b"XFC\x02")
    data = b"".join(rec)
    stream, ms = check_roundtrip(data)
    z1 = sum(len(zlib.compress(data[i:i + 0xff00], 1)) for i in range(0, len(data), 0xff00))
    assert len(stream) < 1.05 * z1 and len(stream) < 0.2 * len(data), (len(stream), z1, len(data))
    blocks = bgzf_blocks(stream)
    assert bytes(gpu_inflate(stream, [(c, l, i) for c, l, i, _ in blocks])) == data
    # binned qualities with stretches at the top value (HiFi-like): runs and literals mixed
    q = np.array([3, 10, 17, 22, 27, 33, 40, 93], np.uint8)[rng.integers(0, 8, size=200000)]
    at = rng.integers(0, len(q) - 400, size=300)
    for a in at:
        q[a:a + int(rng.integers(4, 400))] = 93
    stream, ms = check_roundtrip(bytes(q))
    blocks = bgzf_blocks(stream)
    assert bytes(gpu_inflate(stream, [(c, l, i) for c, l, i, _ in blocks])) == bytes(q)
    # two bytes alternating: no runs, nothing may be taken for one
    check_roundtrip(b"ab" * 30000)


def test_many_blocks_short_tail_and_the_gpu_inflater_reads_them():
    rng = np.random.default_rng(8)
    # BAM-like: records of packed bases + qualities + names, 300 blocks and a tail of 777 bytes
    rec = []
    while sum(len(r) for r in rec) < 300 * 0xff00 + 777:
        l = int(rng.integers(9000, 16000))
        rec.append(b"read%07d\0" % len(rec) + bytes(rng.choice([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28], size=l // 2).astype(np.uint8))
                   + bytes(rng.integers(25, 50, size=l, dtype=np.uint8)))
    data = b"".join(rec)[:300 * 0xff00 + 777]
    stream, ms = check_roundtrip(data)
    assert len(ms) == 301 and ms[-1][2] == 777
    # within a few percent of zlib level 1 on this kind of data (no matches to find: Huffman is what compresses it)
    z1 = sum(len(zlib.compress(data[i:i + 0xff00], 1)) for i in range(0, len(data), 0xff00))
    assert len(stream) < 1.08 * z1, (len(stream), z1)
    blocks = bgzf_blocks(stream)
    got = gpu_inflate(stream, [(c, l, i) for c, l, i, _ in blocks])
    assert bytes(got) == data
    # the members written back to back by the library itself (what csrc/bam_writer.h asks for): the same bytes
    assert gpu_deflate(data, dense=True) == stream
    assert gpu_deflate(data[:70001], block_bytes=333, dense=True) == gpu_deflate(data[:70001], block_bytes=333)
    # other block sizes (the writer always uses 0xff00)
    check_roundtrip(data[:200000], block_bytes=4096)
    check_roundtrip(data[:70000], block_bytes=333)


def test_smooth_writes_the_same_records_with_either_deflate(tmp_path):
    """`SVDSS smooth` packs its output on the GPU by default; SVDSS_GPU_DEFLATE=0 keeps the host's deflate.  Different
    compressed bytes, the same BAM."""
    from svdss_amd import synth
    from tests import bam_writer
    rng = np.random.default_rng(3)
    ref = synth.make_reference([400000], seed=5)
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1\n" + synth.to_ascii(ref[0]) + "\n")
    recs = []
    for k in range(120):
        st = int(rng.integers(0, 380000))
        l = int(rng.integers(5000, 15000))
        seq = ref[0][st:st + l].copy()
        e = rng.random(l) < 0.004
        seq[e] = (seq[e] % 4) + 1
        recs.append((st, bam_writer.record(f"r{k}", 0, 0, st, 60, [("M", l)], synth.to_ascii(seq),
                                           qual=bytes(rng.integers(20, 50, size=l, dtype=np.uint8).tolist()))))
    recs.sort(key=lambda r: r[0])
    bam = tmp_path / "in.bam"
    bam.write_bytes(bam_writer.bam([("chr1", 400000)], [r for _, r in recs]))
    outs = {}
    for tag, env in (("gpu", {}), ("host", {"SVDSS_GPU_DEFLATE": "0"})):
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "4"], capture_output=True,
                           timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        outs[tag] = r.stdout
    assert outs["gpu"] != outs["host"]
    a, b = (gzip.GzipFile(fileobj=io.BytesIO(outs[t])).read() for t in ("gpu", "host"))
    assert a == b and len(a) > 1_000_000
    assert outs["gpu"][-28:] == outs["host"][-28:]          # the EOF marker block
    members(outs["gpu"])                                      # well-formed BGZF throughout
