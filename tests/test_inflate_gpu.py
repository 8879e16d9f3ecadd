"""BGZF / raw deflate streams inflated on the GPU (csrc/inflate.hip, svdss_bgzf_inflate) against zlib: every block
type (stored, fixed, dynamic), every compression level and strategy, matches that overlap their own output, maximum
distances, block sizes 0 .. 65536, arbitrary output alignment, corrupt input.  Bit-exact."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def _payloads(rng):
    p = {}
    p["empty"] = b""
    p["one"] = b"A"
    p["text"] = (b"the quick brown fox jumps over the lazy dog. " * 1500)[:65280]
    p["random"] = rng.integers(0, 256, size=65280, dtype=np.uint8).tobytes()
    p["random_max"] = rng.integers(0, 256, size=65536, dtype=np.uint8).tobytes()
    p["nibbles"] = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88],
                                       dtype=np.uint8), size=60000).tobytes()
    p["quals"] = rng.integers(20, 60, size=65000, dtype=np.uint8).tobytes()
    p["binned"] = rng.choice(np.array([2, 10, 20, 30, 40, 93], dtype=np.uint8), p=[.02, .03, .05, .1, .3, .5], size=65536).tobytes()
    p["run"] = b"\x00" * 65536                      # distance 1, length 258 over and over
    p["run3"] = (b"abc" * 22000)[:65536]              # distance 3 < length
    p["far"] = rng.integers(0, 256, size=32768, dtype=np.uint8).tobytes() * 2      # matches at distance 32768
    p["skew"] = rng.choice(np.arange(200, dtype=np.uint8), p=np.r_[0.9, np.full(199, 0.1 / 199)], size=65000).tobytes()  # long codes
    rec = bytearray()
    for i in range(3):                                # BAM-like: core, name, cigar, packed bases, quals
        l = 9000
        rec += struct.pack("<iiiBBHHHiiii", 32 + 8 + 4 + l // 2 + l, 0, 1000 * i, 8, 60, 4680, 1, 0, l, -1, -1, 0)
        rec += b"read%03d\0" % i + struct.pack("<I", l << 4)
        rec += rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x48, 0x84, 0x88], dtype=np.uint8), size=l // 2).tobytes()
        rec += rng.integers(20, 60, size=l, dtype=np.uint8).tobytes()
    p["bam"] = bytes(rec)
    # round 6: runs written as ONE fill by the round engine (csrc/inflate.hip): run lengths around every boundary of the
    # path (2 x 258, the round's 512 bytes, the 4 KB ring, many rings), literals of 0 .. 40 bytes between them, runs at
    # the stream's start and end, a run that ends exactly at the member's size; and BAM records without qualities
    mixed = bytearray()
    for k, n in enumerate([1, 2, 3, 257, 258, 259, 515, 516, 517, 600, 769, 770, 1000, 2047, 4093, 4094, 4095, 4096, 4097, 4098, 4099, 5000,
                           8191, 8192, 8193, 12000]):
        mixed += bytes([int(rng.integers(0, 256))]) * n
        mixed += rng.integers(0, 256, size=int(rng.integers(0, 41)), dtype=np.uint8).tobytes()
    p["runs_mixed"] = bytes(mixed[:65536])
    p["run_start_end"] = b"\xff" * 20000 + rng.integers(0, 256, size=25000, dtype=np.uint8).tobytes() + b"\x07" * 20536
    rec = bytearray()
    for i in range(2):
        l = 15000
        rec += struct.pack("<iiiBBHHHiiii", 32 + 8 + 4 + l // 2 + l, 0, 1000 * i, 8, 60, 4680, 1, 0, l, -1, -1, 0)
        rec += b"read%03d\0" % i + struct.pack("<I", l << 4)
        rec += rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x48, 0x84, 0x88], dtype=np.uint8), size=l // 2).tobytes()
        rec += b"\xff" * l
    p["bam_absent_quals"] = bytes(rec)
    q = np.full(60000, 93, dtype=np.uint8)
    for at in rng.integers(0, 59990, size=300):
        q[at:at + int(rng.integers(1, 9))] = int(rng.integers(5, 60))
    p["hifi_quals"] = q.tobytes()
    return p


def test_every_block_type_level_and_strategy():
    from svdss_amd.bgzf import gpu_inflate
    rng = np.random.default_rng(11)
    pay = _payloads(rng)
    streams, want = [], []
    for name, data in pay.items():
        for level in (0, 1, 6, 9):
            streams.append(_raw(data, level)); want.append(data)
        for strat in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            streams.append(_raw(data, 6, strat)); want.append(data)
        streams.append(_raw(data, 9, memlevel=1)); want.append(data)        # many small deflate blocks per stream
    # back to back in one buffer at odd offsets, outputs back to back too (arbitrary alignment of both)
    comp, blocks = bytearray(), []
    for s, w in zip(streams, want):
        comp += b"\xee" * (len(comp) % 3)
        blocks.append((len(comp), len(s), len(w)))
        comp += s
    out = gpu_inflate(bytes(comp), blocks).tobytes()
    o = 0
    for i, w in enumerate(want):
        assert out[o:o + len(w)] == w, "stream %d" % i
        o += len(w)
    assert o == len(out)


def test_random_mixtures_of_literals_runs_and_copies():
    """400 streams stitched from random pieces -- bytes of alphabets of 2 to 256 symbols (code lengths from 1 bit to
    the one-symbol path), runs, copies of earlier pieces at every distance, BAM-like records -- at random levels and
    strategies: rounds that end at a match, at the byte budget of a round, at the end-of-block code or at a code the
    tables cannot decode, in every combination the pointer-doubling chain has to get right."""
    from svdss_amd.bgzf import gpu_inflate
    rng = np.random.default_rng(2024)
    streams, want = [], []
    for k in range(400):
        target = int(rng.integers(1, 65537))
        buf = bytearray()
        while len(buf) < target:
            kind = int(rng.integers(0, 5))
            n = int(rng.integers(1, 3000))
            if kind == 0:
                a = int(rng.choice([2, 3, 4, 16, 40, 200, 256]))
                buf += rng.integers(0, a, size=n, dtype=np.uint8).tobytes()
            elif kind == 1:
                buf += bytes([int(rng.integers(0, 256))]) * n
            elif kind == 2 and buf:
                d = int(rng.integers(1, min(len(buf), 32768) + 1))
                for _ in range(n):                      # (a copy may overlap itself)
                    buf.append(buf[-d])
            elif kind == 3:
                w = bytes(rng.integers(97, 123, size=int(rng.integers(2, 12)), dtype=np.uint8))
                buf += (w + b" ") * (n // len(w) + 1)
            else:
                p_ = np.r_[0.97, np.full(254, 0.03 / 254)]   # one very short code, many very long ones
                buf += rng.choice(np.arange(255, dtype=np.uint8), p=p_, size=n).tobytes()
        data = bytes(buf[:target])
        level = int(rng.choice([1, 1, 6, 9]))
        strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 6))]
        streams.append(_raw(data, level, strat, memlevel=int(rng.choice([1, 8, 9])))); want.append(data)
    comp, blocks = bytearray(), []
    for s_, w in zip(streams, want):
        comp += b"\x5a" * int(rng.integers(0, 4))
        blocks.append((len(comp), len(s_), len(w)))
        comp += s_
    out = gpu_inflate(bytes(comp), blocks).tobytes()
    o = 0
    for i, w in enumerate(want):
        assert out[o:o + len(w)] == w, "stream %d" % i
        o += len(w)
    assert o == len(out)


def test_what_the_passes_of_288_bit_pieces_have_to_get_right():
    """Round 4's symbol engine (csrc/inflate.hip: every lane walks its own 288 bits, starts settle by resynchronisation):
    codes of ONE length never resynchronise (a pass yields a lane or two, the kernel falls back to rounds); runs and long
    copies make one lane's piece larger than the 4 KB ring (capacity cuts, lane 0 alone too large); dense 3-byte copies
    overflow the queue of 512 matches per pass; copies whose source lies right around a pass's first output byte and around
    the ring's size are read partly from the ring and partly from HBM; several deflate blocks per stream (memLevel 1) end
    passes at end-of-block codes after a few lanes."""
    from svdss_amd.bgzf import gpu_inflate
    rng = np.random.default_rng(404)
    streams, want = [], []

    def add(data, level=6, strat=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
        streams.append(_raw(bytes(data), level, strat, memlevel=memlevel)); want.append(bytes(data))

    for bits in range(1, 9):                                   # 2^bits equiprobable symbols: every code `bits` long
        add(rng.integers(0, 1 << bits, size=60000, dtype=np.uint8), 6, zlib.Z_HUFFMAN_ONLY)
        add(rng.integers(0, 1 << bits, size=60000, dtype=np.uint8), 1)
    add(b"\x07" * 65536); add(b"ab" * 32768); add((b"x" * 300 + b"y") * 217)          # 258-byte copies back to back
    words = [bytes(rng.integers(0, 256, size=3, dtype=np.uint8)) for _ in range(40)]
    add(b"".join(words[int(i)] for i in rng.integers(0, 40, size=21000)), 9)                # a 3-byte copy per ~14 bits
    add(b"".join(words[int(i)] for i in rng.integers(0, 40, size=21000)), 1)
    for base in (1, 40, 60, 64, 70, 200, 1800, 3700, 3830, 4000, 4096, 4200, 8192, 20000, 32768):
        buf = bytearray(rng.integers(0, 256, size=base + 300, dtype=np.uint8).tobytes())
        while len(buf) < 65000:                                  # literals, then a copy from `base` +- a little back
            buf += rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            d = max(1, min(len(buf), base + int(rng.integers(-8, 9))))
            for _ in range(int(rng.integers(3, 70))):
                buf.append(buf[-d])
        add(buf, 9); add(buf, 1)
    for ml in (1, 2, 3):                                         # deflate blocks of a few hundred symbols each
        a = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28], dtype=np.uint8), size=20000).tobytes()
        add(a + rng.integers(30, 50, size=40000, dtype=np.uint8).tobytes(), 6, memlevel=ml)
    comp, blocks = bytearray(), []
    for s_, w in zip(streams, want):
        comp += b"\xa5" * int(rng.integers(0, 4))
        blocks.append((len(comp), len(s_), len(w)))
        comp += s_
    out = gpu_inflate(bytes(comp), blocks).tobytes()
    o = 0
    for i, w in enumerate(want):
        assert out[o:o + len(w)] == w, "stream %d" % i
        o += len(w)
    assert o == len(out)


def test_flipped_bits_anywhere_end_in_an_answer():
    """A bit flipped anywhere in a stream -- headers, code lengths, deep inside the symbols where the lanes of a pass walk
    from guessed starts -- must end in an answer: an error, or bytes (zlib's own, if zlib accepts the stream too; the BGZF
    CRC is the caller's check).  Never a hang, never a write outside the block's output."""
    from svdss_amd._lib import SvdssError
    from svdss_amd.bgzf import gpu_inflate
    rng = np.random.default_rng(77)
    pay = _payloads(rng)
    guard = b"\xcc" * 4096
    n_err = n_same = n_diff = 0
    for name in ("bam", "binned", "skew", "text", "nibbles"):
        good = _raw(pay[name], 1 if name != "text" else 6)
        for k in range(24):
            t = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                t[int(rng.integers(0, len(t)))] ^= 1 << int(rng.integers(0, 8))
            ok = _raw(guard, 0)                                  # stored blocks before and behind: their output must survive
            comp = ok + bytes(t) + ok
            blocks = [(0, len(ok), len(guard)), (len(ok), len(t), len(pay[name])), (len(ok) + len(t), len(ok), len(guard))]
            try:
                z = zlib.decompressobj(-15).decompress(bytes(t))
            except zlib.error:
                z = None
            try:
                out = gpu_inflate(comp, blocks).tobytes()
            except SvdssError as e:
                assert e.bad_block == 1
                n_err += 1
                continue
            assert out[:len(guard)] == guard and out[-len(guard):] == guard
            if z is not None and len(z) == len(pay[name]):
                assert out[len(guard):-len(guard)] == z
                n_same += 1
            else:
                n_diff += 1                                      # (zlib refuses, e.g. a distance beyond its window check, or another size)
    assert n_err > 20 and n_same > 5, (n_err, n_same, n_diff)


def test_a_capped_grid_walks_the_members_with_a_stride():
    """SVDSS_INFLATE_PER_CU=n launches at most n wavefronts per compute unit and lets each inflate several members one
    after the other (csrc/inflate.hip: rings, tables and the lanes' walks start over per member).  In a process of its
    own: the library reads the variable once."""
    import os
    import subprocess
    import sys
    code = r"""
import zlib, numpy as np
from svdss_amd.bgzf import gpu_inflate
rng = np.random.default_rng(5)
streams, want = [], []
for k in range(1500):                       # more members than 1 x 256 wavefronts: every wavefront takes five or six
    n = int(rng.integers(1, 20000))
    kind = k % 4
    if kind == 0: d = rng.integers(0, 40, size=n, dtype=np.uint8).tobytes()
    elif kind == 1: d = bytes(rng.choice(np.array([2, 10, 20, 30, 40, 93], dtype=np.uint8), p=[.02, .03, .05, .1, .3, .5], size=n))
    elif kind == 2: d = (b"run " * n)[:n]
    else: d = b""
    c = zlib.compressobj(int(rng.choice([0, 1, 6])), zlib.DEFLATED, -15)
    streams.append(c.compress(d) + c.flush()); want.append(d)
comp, blocks = bytearray(), []
for s_, w in zip(streams, want):
    blocks.append((len(comp), len(s_), len(w))); comp += s_
out = gpu_inflate(bytes(comp), blocks).tobytes()
assert out == b"".join(want)
print("ok", len(want))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root,
                       env=dict(os.environ, SVDSS_INFLATE_PER_CU="1", PYTHONPATH=root))
    assert p.returncode == 0 and "ok 1500" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_scattered_outputs_do_not_touch_their_neighbours():
    from svdss_amd.bgzf import gpu_inflate
    rng = np.random.default_rng(5)
    datas = [rng.integers(0, 4, size=int(n), dtype=np.uint8).tobytes() for n in (1, 2, 3, 5, 4097, 65535, 7, 16384, 16385)]
    comp, blocks, uoff = bytearray(), [], []
    at = 3
    for d in datas:
        s = _raw(d, 6)
        blocks.append((len(comp), len(s), len(d)))
        comp += s
        uoff.append(at)
        at += len(d) + 5            # 5 bytes between the outputs must stay zero
    out = gpu_inflate(bytes(comp), blocks, uoff=uoff).tobytes()
    for d, u in zip(datas, uoff):
        assert out[u:u + len(d)] == d
        assert out[u - 3:u] == b"\0\0\0"


def test_bgzf_file_blocks():
    """a BGZF file as bgzip / htslib write it (header with BC subfield, footer with CRC32 and ISIZE)"""
    from svdss_amd.bgzf import bgzf_blocks, gpu_inflate
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 8, size=300000, dtype=np.uint8).tobytes()
    data = bytearray()
    for i in range(0, len(raw), 65280):
        blk = raw[i:i + 65280]
        cd = _raw(blk, 6)
        data += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(cd) + 25) + cd
        data += struct.pack("<II", zlib.crc32(blk) & 0xffffffff, len(blk))
    data += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")     # the EOF marker block
    blocks = bgzf_blocks(bytes(data))
    assert blocks[-1][2] == 0
    out = gpu_inflate(bytes(data), [b[:3] for b in blocks]).tobytes()
    assert out == raw
    assert [zlib.crc32(raw[i:i + 65280]) & 0xffffffff for i in range(0, len(raw), 65280)] == [b[3] for b in blocks[:-1]]


def test_corrupt_streams_are_reported_not_trusted():
    from svdss_amd._lib import SvdssError
    from svdss_amd.bgzf import gpu_inflate
    rng = np.random.default_rng(9)
    good = rng.integers(0, 16, size=20000, dtype=np.uint8).tobytes()
    s = bytearray(_raw(good, 6))
    cases = []
    # (a flipped bit in the middle of the symbols can turn one literal into another of the same code length: only the
    # CRC32 of the footer sees that, and the caller checks it on the host -- see BamReader; structural damage is the
    # kernel's to report)
    t = bytearray(s); t[1] ^= 0xff; t[2] ^= 0xff; cases.append((bytes(t), len(good)))     # damaged code-length header
    cases.append((bytes(s), len(good) - 1))                                           # wrong isize (too small)
    cases.append((bytes(s), len(good) + 1))                                           # wrong isize (too large)
    cases.append((bytes(s[:len(s) // 2]), len(good)))                                 # truncated
    cases.append((b"\x07" + bytes(s[1:]), len(good)))                                 # block type 3
    for k, (c, isz) in enumerate(cases):
        ok = _raw(good, 1)
        comp = ok + c
        with pytest.raises(SvdssError) as ei:
            gpu_inflate(comp, [(0, len(ok), len(good)), (len(ok), len(c), isz)])
        assert ei.value.bad_block == 1, k
    # and the undamaged pair inflates
    assert gpu_inflate(ok + bytes(s), [(0, len(ok), len(good)), (len(ok), len(s), len(good))]).tobytes() == good + good


def test_streams_written_by_libdeflate():
    """htslib deflates BGZF blocks with libdeflate when it is built with it (and so does this repo's own BAM writer):
    its streams -- other block splits, other Huffman codes, other match choices than zlib's at every level 1 .. 12 -- must
    inflate to the same bytes on the GPU."""
    import ctypes as C
    from svdss_amd.bgzf import gpu_inflate
    lib = None
    for name in ("libdeflate.so.0", "libdeflate.so"):
        try:
            lib = C.CDLL(name)
            break
        except OSError:
            continue
    if lib is None:
        pytest.skip("libdeflate is not on this machine")
    lib.libdeflate_alloc_compressor.restype = C.c_void_p
    lib.libdeflate_alloc_compressor.argtypes = [C.c_int]
    lib.libdeflate_deflate_compress.restype = C.c_size_t
    lib.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.libdeflate_free_compressor.argtypes = [C.c_void_p]
    rng = np.random.default_rng(21)
    pay = {k: v for k, v in _payloads(rng).items() if v}
    blob, blocks, want = bytearray(), [], bytearray()
    for level in (1, 3, 6, 9, 12):
        comp = lib.libdeflate_alloc_compressor(level)
        for name, data in pay.items():
            out = C.create_string_buffer(len(data) + 1024)
            n = lib.libdeflate_deflate_compress(comp, data, len(data), out, len(out))
            assert n > 0, (level, name)
            blocks.append((len(blob), n, len(data)))
            blob += out.raw[:n]
            want += data
        lib.libdeflate_free_compressor(comp)
    got = gpu_inflate(bytes(blob), blocks)
    assert bytes(got) == bytes(want)
