"""Developer sweep: search-kernel throughput on the default bench workload vs library environment knobs
(SVDSS_SEGMENTS, SVDSS_BLOCKS), one index build for all settings."""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svdss_amd                      # noqa: E402
from svdss_amd import synth           # noqa: E402
import bench                          # noqa: E402

ref_len, L = 64444167, 15000
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 128888
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ref = synth.make_reference([ref_len], seed=11, repeat_frac=float(os.environ.get("SWEEP_REPEAT_FRAC", "0.03")))
ix = svdss_amd.FMDIndex.build(ref)
ix.to_device(0)
pp = svdss_amd.PingPong(ix, assemble=True)
ref_t = torch.from_numpy(ref[0]).to(dev)
d_reads, d_offsets = bench.simulate_reads_gpu(ref_t, n_reads, L, 0.005, seed=13, device=dev)
stream = torch.cuda.current_stream()
total_syms = n_reads * L
segs = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4", "8", "16"])]
blocks = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["512", "1024", "2048"])]
for seg, blk in itertools.product(segs, blocks):
    if seg > 0:
        os.environ["SVDSS_SEGMENTS"] = str(seg)      # 0: the library's own choice
    else:
        os.environ.pop("SVDSS_SEGMENTS", None)
    if blk > 0:
        os.environ["SVDSS_BLOCKS"] = str(blk)
    else:
        os.environ.pop("SVDSS_BLOCKS", None)
    ks = []
    for it in range(7):
        pp.ping_pong_search_device(d_reads.data_ptr(), d_offsets.data_ptr(), n_reads, total_syms, stream=stream.cuda_stream, fetch=False)
        if it >= 2:
            ks.append(pp.last_kernel_ms)
    print(f"reads={n_reads} seg={seg or pp.last_segments} blocks={blk}: kernel {np.mean(ks):.2f} ms -> {n_reads / np.mean(ks) * 1e3 / 1e6:.2f} M reads/s, "
          f"redone {pp.last_fallbacks}", flush=True)
