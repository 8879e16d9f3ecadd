#!/usr/bin/env python3
"""The search launches of bench.py's default workload on their own (no torch kernels of the read simulator around
them inside the profiled region would be better, but the simulator runs once): index built in HBM, one batch of reads,
N searches.  Used under rocprofv3 --pmc (tools/collect_profiles_r02.sh).
  python tools/search_only.py [wg|chr20] [n_reads] [repeats] [families:<fraction>:<divergence>]
(the last argument: a reference in which that fraction of the bases are copies of 40 repeat families, synth.make_family_reference)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import svdss_amd  # noqa: E402
from svdss_amd import synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "wg"
lens = bench.GRCH38_PRIMARY if wl == "wg" else [64_444_167]
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 20 if wl == "wg" else 128888)
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = 15000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if len(sys.argv) > 4 and sys.argv[4].startswith("families:"):
    _, frac, div = sys.argv[4].split(":")
    ref = synth.make_family_reference(lens, seed=11, repeat_frac=float(frac), divergence=float(div))
    wl += " " + sys.argv[4]
else:
    ref = synth.make_reference(lens, seed=11)
ix = svdss_amd.FMDIndex.build(ref, device=0)
starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
ref_t = torch.from_numpy(ref[0] if len(ref) == 1 else np.concatenate(ref)).to(dev)
del ref
err = float(os.environ.get("SEARCH_ONLY_ERR", "0.005"))
d_reads, d_offs = bench.simulate_reads_gpu(ref_t, [(int(s), int(l)) for s, l in zip(starts, lens)], n_reads, L, err,
                                           seed=13, device=dev)
del ref_t
torch.cuda.synchronize()
pp = svdss_amd.PingPong(ix, assemble=True)
st = torch.cuda.Stream(device=dev)
ks = []
for _ in range(rep):
    pp.ping_pong_search_device(d_reads.data_ptr(), d_offs.data_ptr(), n_reads, n_reads * L, stream=st.cuda_stream, fetch=False)
    ks.append(pp.last_search_kernel_ms)
print(f"{wl}: {n_reads} reads, K={ix.kmer_k}, search kernel {np.mean(ks[1:] or ks):.2f} ms "
      f"({n_reads / np.mean(ks[1:] or ks) * 1e3 / 1e6:.2f} M reads/s), segments {pp.last_segments}, "
      f"ext/read {pp.last_total_ext / n_reads:.0f}, assembled SFS/read {pp.last_total / n_reads:.1f}", flush=True)
