"""Developer probe: the GPU deflate encoder (csrc/deflate.hip) on BAM-like data -- kernel time, ratio against zlib.
  python tools/deflate_probe.py [n_blocks] [binned | ff | hifi]     (ff: qualities absent, 0xff; hifi: binned with stretches at the top value)"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svdss_amd.bgzf import gpu_deflate  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = sys.argv[2] if len(sys.argv) > 2 else "random"
binned = kind == "binned"
rng = np.random.default_rng(1)
parts = []
total = nb * 0xff00
while sum(len(p) for p in parts) < total:
    l = 15000
    parts.append(b"read%07d\0" % len(parts))
    parts.append(rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], np.uint8),
                            size=l // 2).tobytes())
    if kind == "ff":
        q = np.full(l, 0xff, np.uint8)
    elif kind == "hifi":
        q = np.array([3, 10, 17, 22, 27, 33, 40, 93], np.uint8)[rng.integers(0, 8, size=l)]
        for a in rng.integers(0, l - 300, size=25):
            q[a:a + int(rng.integers(4, 300))] = 93
    else:
        q = (np.array([3, 10, 17, 22, 27, 33, 40], np.uint8)[rng.integers(0, 7, size=l)] if binned
             else rng.integers(20, 60, size=l, dtype=np.uint8))
    parts.append(q.tobytes())
data = b"".join(parts)[:total]
for rep in range(3):
    t0 = time.perf_counter()
    out, st = gpu_deflate(data, return_stats=True)
    wall = time.perf_counter() - t0
    print(f"{nb} blocks, {len(data) / 1e6:.1f} MB -> {len(out) / 1e6:.1f} MB ({len(out) / len(data):.3f}); kernel {st['kernel_ms']:.2f} ms "
          f"= {len(data) / st['kernel_ms'] / 1e6:.1f} GB/s of input; call {wall:.3f} s (Python footers included)", flush=True)
t0 = time.perf_counter()
z1 = sum(len(zlib.compress(data[i:i + 0xff00], 1)) for i in range(0, min(len(data), 200 * 0xff00), 0xff00))
z6 = sum(len(zlib.compress(data[i:i + 0xff00], 6)) for i in range(0, min(len(data), 200 * 0xff00), 0xff00))
n = min(len(data), 200 * 0xff00)
print(f"zlib on the first {n / 1e6:.1f} MB: level 1 ratio {z1 / n:.3f}, level 6 ratio {z6 / n:.3f}")
