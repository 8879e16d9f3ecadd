#!/usr/bin/env python3
"""The call-side DP work of one bench step (bench.CallWorkload) on its own: wall and kernel times per stage.
  python tools/call_dp_probe.py [n_clusters] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CallWorkload  # noqa: E402
from svdss_amd._lib import check, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3395
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.time()
cw = CallWorkload(n, seed=99)
print(f"workload: {cw.n_clusters} clusters, {cw.n_sub} sub-clusters, {int(cw.cluster_off[-1])} sub-reads, "
      f"{len(cw.seqs) / 1e6:.1f} MB, built in {time.time() - t0:.1f} s", flush=True)
for r in range(rep):
    t0 = time.perf_counter()
    cw.run(lib, check, 0)
    w = (time.perf_counter() - t0) * 1e3
    L = cw.last
    print(f"run {r}: wall {w:.1f} ms | POA wall {L['poa_wall_ms']:.1f} kernel {L['poa_kernel_ms']:.1f} "
          f"({L['poa_cells'] / L['poa_kernel_ms'] / 1e6:.1f} GCUPS, {L['poa_hbm']} on the HBM kernel) | realign wall "
          f"{L['realign_wall_ms']:.1f} kernel {L['realign_kernel_ms']:.1f} "
          f"({L['realign_cells'] / L['realign_kernel_ms'] / 1e6:.1f} GCUPS) | ratio wall {L['ratio_wall_ms']:.1f}", flush=True)
ok, n_alt = cw.svs_recovered()
print(f"implanted SVs found in the CIGARs: {ok}/{n_alt}")
