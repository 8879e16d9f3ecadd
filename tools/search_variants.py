#!/usr/bin/env python3
"""Developer A/B of search-kernel builds on one box: the whole-genome index and one batch of reads are made once and
kept in /dev/shm; every build of the library (SVDSS_LIB) then restores the index, takes the reads and times the search.
  python tools/search_variants.py prep [n_reads]          index + reads -> /dev/shm/svdss_sv/
  SVDSS_LIB=... python tools/search_variants.py run [repeats]   one line: kernel ms, totals (SFS, extensions)
  python tools/search_variants.py all lib1.so lib2.so ...   prep, then `run` in a process per library, twice round
"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = os.environ.get("SV_WORK", "/dev/shm/svdss_sv")
L = 15000


def prep(n_reads):
    import torch
    import bench
    import svdss_amd
    from svdss_amd import synth
    os.makedirs(W, exist_ok=True)
    t0 = time.time()
    dev = torch.device("cuda", 0)
    lens = bench.GRCH38_PRIMARY
    ref = synth.make_reference(lens, seed=11)
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    ix.save_records(os.path.join(W, "wg.idx"))
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    ref_t = torch.from_numpy(np.concatenate(ref)).to(dev)
    del ref
    d_reads, d_offs = bench.simulate_reads_gpu(ref_t, [(int(s), int(l)) for s, l in zip(starts, lens)], n_reads, L, 0.005, seed=13, device=dev)
    d_reads.cpu().numpy().tofile(os.path.join(W, "reads.bin"))
    d_offs.cpu().numpy().tofile(os.path.join(W, "offs.bin"))
    print(f"prep: index + {n_reads} reads in {time.time() - t0:.0f} s", flush=True)


def run(rep):
    import torch
    import svdss_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ix = svdss_amd.FMDIndex.load(os.path.join(W, "wg.idx"))
    ix.to_device(0)
    offs = np.fromfile(os.path.join(W, "offs.bin"), dtype=np.int64)
    n_reads = len(offs) - 1
    d_reads = torch.from_numpy(np.fromfile(os.path.join(W, "reads.bin"), dtype=np.uint8)).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    pp = svdss_amd.PingPong(ix, assemble=True)
    st = torch.cuda.Stream(device=dev)
    ks = []
    for _ in range(rep):
        pp.ping_pong_search_device(d_reads.data_ptr(), d_offs.data_ptr(), n_reads, int(offs[-1]), stream=st.cuda_stream, fetch=False)
        ks.append(pp.last_search_kernel_ms)
    print(f"{os.path.basename(os.environ.get('SVDSS_LIB', 'libsvdss_hip.so'))}: K={ix.kmer_k} kernel ms {' '.join(f'{k:.2f}' for k in ks)} | min {min(ks):.2f} | "
          f"SFS {pp.last_total} ext {pp.last_total_ext} segments {pp.last_segments}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "run"
    if mode == "prep":
        prep(int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20)
    elif mode == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    else:
        if not os.path.exists(os.path.join(W, "reads.bin")):
            subprocess.check_call([sys.executable, __file__, "prep"] + ([os.environ["SV_READS"]] if os.environ.get("SV_READS") else []))
        for rnd in range(int(os.environ.get("SV_ROUNDS", "2"))):
            for lib in sys.argv[2:]:
                env = dict(os.environ, SVDSS_LIB=os.path.abspath(lib))
                subprocess.call([sys.executable, __file__, "run"], env=env)
