"""Developer A/B on one box: whole-genome-scale index built once, search kernel timed with and without the
forward-phase outcome in the k-mer table entries (SVDSS_TABLE_FORWARD), same reads."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svdss_amd                      # noqa: E402
from svdss_amd import synth           # noqa: E402
import bench                          # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
L = 15000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
t0 = time.time()
ref = synth.make_reference(bench.GRCH38_PRIMARY, seed=11)
ix = svdss_amd.FMDIndex.build(ref)
print(f"index built in {time.time() - t0:.0f} s", flush=True)
ref_t = torch.from_numpy(np.concatenate(ref)).to(dev)
del ref
d_reads, d_offsets = bench.simulate_reads_gpu(ref_t, n_reads, L, 0.005, seed=13, device=dev)
del ref_t
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
for setting in ("1", "0", "1", "0"):
    os.environ["SVDSS_TABLE_FORWARD"] = setting
    ix.to_device(0)
    pp = svdss_amd.PingPong(ix, assemble=True)
    ks = []
    for it in range(5):
        pp.ping_pong_search_device(d_reads.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L,
                                   stream=stream.cuda_stream, fetch=False)
        if it >= 2:
            ks.append(pp.last_search_kernel_ms)
    print(f"SVDSS_TABLE_FORWARD={setting}: search kernel {np.mean(ks):.2f} ms ({n_reads / np.mean(ks) / 1e3:.2f} M reads/s), "
          f"sfs {pp.last_total}, ext {pp.last_total_ext}", flush=True)
