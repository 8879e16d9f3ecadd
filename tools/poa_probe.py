"""Developer probe: POA consensus on a few synthetic sub-clusters (prints the batch statistics)."""
import sys
import numpy as np
from svdss_amd import calldp as caller

n, ln, depth = (int(x) for x in (sys.argv[1:4] + ["2", "1200", "12"][len(sys.argv) - 1:]))
rng = np.random.default_rng(99)
clusters = []
for c in range(n):
    t = rng.integers(0, 4, size=ln).astype(np.uint8)
    reads = []
    for _ in range(depth):
        r = t.copy()
        e = rng.random(len(r)) < 0.01
        r[e] = rng.integers(0, 4, size=int(e.sum()), dtype=np.uint8)
        reads.append(r)
    clusters.append(reads)
cons, poa = caller.run_poa(clusters, device=0)
print(poa)
