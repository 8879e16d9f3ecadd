#!/usr/bin/env python3
"""T threads, each running the call-side DP of one bench step R times on its own batch objects: does the throughput of
the call side grow when several batches are in flight?   python tools/call_dp_concurrent.py [threads] [repeats] [clusters per batch]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CallWorkload  # noqa: E402
from svdss_amd._lib import check, lib  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NC = int(sys.argv[3]) if len(sys.argv) > 3 else 3395
cw = CallWorkload(NC, seed=99)
ws = [cw] + [cw.clone() for _ in range(T - 1)]
for w in ws:
    w.run(lib, check, 0)   # arenas
def work(w):
    for _ in range(R):
        w.run(lib, check, 0)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(w,)) for w in ws]
for t in th: t.start()
for t in th: t.join()
el = time.perf_counter() - t0
print(f"{T} thread(s) x {R} calls of {NC} clusters: {el * 1e3:.0f} ms total, {el * 1e3 / (T * R):.1f} ms per call = {el * 1e3 / (T * R) * 3395 / NC:.1f} ms per 3395 clusters; last call of thread 0: "
      f"POA kernel {ws[0].last['poa_kernel_ms']:.0f} ms, realign {ws[0].last['realign_kernel_ms']:.0f} ms")
