"""Developer probe: the longest chains of the bench's call-side workload on their own (n sub-clusters of `depth` reads of
`ln` bases, 0.5 % errors): the time of one chain, i.e. the floor of any batch that contains it.
  python tools/poa_long_probe.py [n] [len] [depth]"""
import sys
import time
import numpy as np
from svdss_amd import calldp as caller

n, ln, depth = (int(x) for x in (sys.argv[1:4] + ["16", "2600", "30"][len(sys.argv) - 1:]))
rng = np.random.default_rng(7)
clusters = []
for c in range(n):
    t = rng.integers(0, 4, size=ln).astype(np.uint8)
    reads = []
    for _ in range(depth):
        r = t.copy()
        e = rng.random(len(r)) < 0.005
        r[e] = (r[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
        reads.append(r)
    clusters.append(reads)
for rep in range(3):
    t0 = time.perf_counter()
    cons, poa = caller.run_poa(clusters, device=0)
    print(f"{n} x {depth} x {ln}: wall {(time.perf_counter() - t0) * 1e3:.1f} ms", poa, flush=True)
