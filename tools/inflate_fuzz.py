"""Differential fuzzer of the GPU inflate kernel (csrc/inflate.hip) against zlib: random mixtures of literal alphabets,
runs, copies at every distance and BAM-like records, deflated by zlib at random levels / strategies / memLevels (and by
libdeflate when it is on the machine), inflated on the GPU in batches, compared byte by byte.
  python tools/inflate_fuzz.py [seconds] [seed]"""
import ctypes as C
import sys
import time
import zlib

import numpy as np

from svdss_amd.bgzf import gpu_inflate

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ld = None
for name in ("libdeflate.so.0", "libdeflate.so"):
    try:
        ld = C.CDLL(name)
        ld.libdeflate_alloc_compressor.restype = C.c_void_p
        ld.libdeflate_alloc_compressor.argtypes = [C.c_int]
        ld.libdeflate_deflate_compress.restype = C.c_size_t
        ld.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        break
    except OSError:
        continue
comps = {lv: ld.libdeflate_alloc_compressor(lv) for lv in (1, 3, 6, 9, 12)} if ld else {}


def payload():
    target = int(rng.integers(1, 65537))
    buf = bytearray()
    while len(buf) < target:
        kind = int(rng.integers(0, 7))
        n = int(rng.integers(1, 4000))
        if kind == 0:
            a = int(rng.choice([2, 3, 4, 6, 16, 40, 94, 200, 256]))
            buf += rng.integers(0, a, size=n, dtype=np.uint8).tobytes()
        elif kind == 1:
            buf += bytes([int(rng.integers(0, 256))]) * n
        elif kind == 2 and buf:
            d = int(rng.integers(1, min(len(buf), 32768) + 1))
            for _ in range(n):
                buf.append(buf[-d])
        elif kind == 3:
            w = bytes(rng.integers(97, 123, size=int(rng.integers(2, 12)), dtype=np.uint8))
            buf += (w + b" ") * (n // len(w) + 1)
        elif kind == 4:
            p_ = np.r_[0.9 + 0.09 * rng.random(), np.full(254, 1.0)]
            p_[1:] *= (1 - p_[0]) / 254
            buf += rng.choice(np.arange(255, dtype=np.uint8), p=p_ / p_.sum(), size=n).tobytes()
        elif kind == 5:                                      # packed bases + qualities of 2^k values
            buf += rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], dtype=np.uint8), size=n // 3 + 1).tobytes()
            buf += rng.integers(20, 20 + (1 << int(rng.integers(1, 7))), size=n, dtype=np.uint8).tobytes()
        else:
            words = [bytes(rng.integers(0, 256, size=int(rng.integers(3, 9)), dtype=np.uint8)) for _ in range(int(rng.integers(2, 60)))]
            buf += b"".join(words[int(i)] for i in rng.integers(0, len(words), size=n // 4 + 1))
    return bytes(buf[:target])


t0 = time.time()
n_streams = n_bytes = n_rounds = 0
while time.time() - t0 < seconds:
    streams, want = [], []
    for _ in range(192):
        data = payload()
        if comps and rng.random() < 0.35:
            lv = int(rng.choice(list(comps)))
            out = C.create_string_buffer(len(data) + len(data) // 8 + 400)
            k = ld.libdeflate_deflate_compress(comps[lv], data, len(data), out, len(out))
            assert k > 0
            s = out.raw[:k]
        else:
            level = int(rng.choice([1, 1, 2, 4, 6, 9]))
            strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 6))]
            c = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.choice([1, 2, 8, 9])), strat)
            s = c.compress(data) + c.flush()
        streams.append(s); want.append(data)
    comp, blocks = bytearray(), []
    for s, w in zip(streams, want):
        comp += b"\x3c" * int(rng.integers(0, 4))
        blocks.append((len(comp), len(s), len(w)))
        comp += s
    out = gpu_inflate(bytes(comp), blocks).tobytes()
    o = 0
    for i, w in enumerate(want):
        if out[o:o + len(w)] != w:
            open("/tmp/inflate_fuzz_fail.bin", "wb").write(streams[i])
            print("MISMATCH: round %d stream %d (%d bytes), seed %d; deflate stream saved to /tmp/inflate_fuzz_fail.bin" % (n_rounds, i, len(w), seed))
            sys.exit(1)
        o += len(w)
    n_streams += len(want); n_bytes += o; n_rounds += 1
print("inflate fuzz: %d streams, %.1f MB inflated, %d rounds in %.0f s, seed %d, libdeflate %s: no mismatch" % (
    n_streams, n_bytes / 1e6, n_rounds, time.time() - t0, seed, "yes" if ld else "no"))
