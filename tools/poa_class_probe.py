#!/usr/bin/env python3
"""Developer probe: POA throughput for one length class -- n sub-clusters of `depth` reads of lo..hi bases (0.5 % substitutions),
T batches in flight on T batch objects.   python tools/poa_class_probe.py n lo hi depth threads repeats"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svdss_amd._lib import check, lib  # noqa: E402

n, lo, hi, depth, T, R = (int(x) for x in sys.argv[1:7])
rng = np.random.default_rng(3)
seqs, sizes = [], []
for c in range(n):
    t = rng.integers(0, 4, size=int(rng.integers(lo, hi + 1))).astype(np.uint8)
    for _ in range(depth):
        r = t.copy()
        e = rng.random(len(r)) < 0.005
        r[e] = (r[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
        seqs.append(r)
    sizes.append(depth)
seq_off = np.zeros(len(seqs) + 1, dtype=np.int64)
seq_off[1:] = np.cumsum([len(s) for s in seqs])
flat = np.ascontiguousarray(np.concatenate(seqs))
cl_off = np.zeros(n + 1, dtype=np.int64)
cl_off[1:] = np.cumsum(sizes)
handles = [C.c_void_p() for _ in range(T)]


def run(h):
    check(lib.svdss_poa_consensus_batch(flat.ctypes.data, seq_off.ctypes.data, cl_off.ctypes.data, n, 0, C.byref(h)), "poa")


for h in handles:
    run(h)


def work(h):
    for _ in range(R):
        run(h)


t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(h,)) for h in handles]
for t in th:
    t.start()
for t in th:
    t.join()
el = time.perf_counter() - t0
print(f"{n} x {depth} x {lo}-{hi}, {T} thread(s) x {R}: {el * 1e3 / (T * R):.1f} ms per batch; last kernel {lib.svdss_poa_batch_kernel_ms(handles[0]):.1f} ms, "
      f"cells {lib.svdss_poa_batch_cells(handles[0]) / 1e9:.2f} G, handed back {lib.svdss_poa_batch_quad_back(handles[0])}")
