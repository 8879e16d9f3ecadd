"""Developer probe: reads/s of svdss_sfs_search_batch when the reads are handed over as HOST buffers (H2D copy of the
reads and D2H copy of the results included), default bench workload."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svdss_amd                      # noqa: E402
from svdss_amd import synth           # noqa: E402
import bench                          # noqa: E402

ref_len, L, n_reads = 64444167, 15000, 128888
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ref = synth.make_reference([ref_len], seed=11)
ix = svdss_amd.FMDIndex.build(ref)
ix.to_device(0)
ref_t = torch.from_numpy(ref[0]).to(dev)
d_reads, d_offs = bench.simulate_reads_gpu(ref_t, n_reads, L, 0.005, seed=13, device=dev)
flat = d_reads.cpu().numpy()[: n_reads * L].copy()
offs = d_offs.cpu().numpy().astype(np.int64)
pp = svdss_amd.PingPong(ix, assemble=True)
for it in range(4):
    t0 = time.perf_counter()
    got = pp.ping_pong_search(flat, offs)
    dt = time.perf_counter() - t0
    print(f"host buffers in, host results out: {dt * 1e3:.1f} ms -> {n_reads / dt / 1e6:.2f} M reads/s "
          f"(kernels {pp.last_kernel_ms:.1f} ms, {len(got.qs)} SFS)", flush=True)
