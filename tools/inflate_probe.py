"""Developer probe: throughput of the GPU BGZF inflate (svdss_bgzf_inflate) on BAM-like blocks.
  python tools/inflate_probe.py [n_blocks] [level] [kind]     kind: bam (packed bases + random qualities) | binned | skew | absent | hifi | text"""
import ctypes as C
import sys
import time
import zlib
import numpy as np
from svdss_amd._lib import lib, check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = sys.argv[3] if len(sys.argv) > 3 else "bam"
strategy = zlib.Z_HUFFMAN_ONLY if len(sys.argv) > 4 and sys.argv[4] == "huff" else zlib.Z_DEFAULT_STRATEGY   # huff: literals only, what csrc/deflate.hip writes
rng = np.random.default_rng(1)
uniq = 64
raws, comps = [], []
for i in range(uniq):
    if kind == "bam":
        a = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], dtype=np.uint8), size=21760)
        b = rng.integers(20, 60, size=43520, dtype=np.uint8)
        raw = np.concatenate([a, b]).tobytes()
    elif kind == "binned":
        a = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], dtype=np.uint8), size=21760)
        b = rng.choice(np.array([2, 10, 20, 30, 40, 93], dtype=np.uint8), p=[.02, .03, .05, .1, .3, .5], size=43520)
        raw = np.concatenate([a, b]).tobytes()
    elif kind == "skew":
        # qualities over the full range with a long tail of rare values (unbinned HiFi: most bases at the cap): codes
        # of 1 to 13+ bits, the rare ones longer than the direct table's index
        a = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], dtype=np.uint8), size=21760)
        pq = np.r_[np.full(93, 0.35 / 93) * np.linspace(0.05, 1.95, 93), 0.65]
        b = rng.choice(np.arange(94, dtype=np.uint8), p=pq / pq.sum(), size=43520)
        raw = np.concatenate([a, b]).tobytes()
    elif kind in ("absent", "hifi"):
        # whole BAM records as tools/chain_dataset.cpp writes them: core + name + ~90 CIGAR operations, 7,500 bytes of packed
        # bases, then 15,000 qualities -- absent (0xff: what the chain's data set has) or HiFi-like (stretches at the top
        # value, dips of a few bases): RUNS, which a deflate writer codes as matches of up to 258 bytes at distance 1
        parts = []
        while sum(len(p) for p in parts) < 65280:
            hdr = rng.integers(0, 256, size=36 + 18 + 360, dtype=np.uint8)
            a = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], dtype=np.uint8), size=7500)
            if kind == "absent":
                q = np.full(15000, 0xff, dtype=np.uint8)
            else:
                q = np.full(15000, 93, dtype=np.uint8)
                for at in rng.integers(0, 14990, size=60):
                    q[at:at + int(rng.integers(1, 9))] = rng.integers(5, 60, size=1, dtype=np.uint8)[0]
            parts += [hdr, a, q]
        raw = np.concatenate(parts).tobytes()[:65280]
    else:
        raw = (b"the quick brown fox jumps over the lazy dog %d. " % i * 1500)[:65280]
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    comps.append(c.compress(raw) + c.flush())
    raws.append(raw)
comp = bytearray()
rec = np.zeros(n, dtype=[("coff", "<i8"), ("clen", "<i4"), ("isize", "<i4"), ("uoff", "<i8")])
uo = 0
for i in range(n):
    s = comps[i % uniq]
    rec[i] = (len(comp), len(s), len(raws[i % uniq]), uo)
    comp += s
    uo += len(raws[i % uniq])
pin = C.c_void_p()
check(lib.svdss_host_alloc(len(comp), C.byref(pin)), "host_alloc")
C.memmove(pin, bytes(comp), len(comp))
hout = C.c_void_p()
check(lib.svdss_host_alloc(uo, C.byref(hout)), "host_alloc")
d_out = C.c_void_p()
check(lib.svdss_device_alloc(0, uo, C.byref(d_out)), "device_alloc")
obj = C.c_void_p()
bad = C.c_int64()
print(f"{n} blocks, level {level}, {kind}: {len(comp) / 1e6:.1f} MB -> {uo / 1e6:.1f} MB")
for rep in range(4):
    for with_host in (False, True):
        t0 = time.perf_counter()
        check(lib.svdss_bgzf_inflate(C.byref(obj), 0, pin, len(comp), rec.ctypes.data, n, d_out, hout if with_host else None, uo, C.byref(bad)), "inflate")
        w = time.perf_counter() - t0
        k = lib.svdss_inflate_kernel_ms(obj)
        print(f"  {'with' if with_host else 'no  '} copy back: wall {w * 1e3:.1f} ms, kernel {k:.2f} ms = {uo / k / 1e6:.1f} GB/s out ({n / k * 1e3 / 1e3:.0f} k blocks/s)")
out = np.zeros(uo, dtype=np.uint8)
C.memmove(out.ctypes.data, hout, uo)
ok = all(out[int(rec[i]["uoff"]):int(rec[i]["uoff"]) + len(raws[i % uniq])].tobytes() == raws[i % uniq] for i in range(0, n, 97))
print("verified:", ok)
