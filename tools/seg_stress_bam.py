#!/usr/bin/env python3
"""Round 4's failing scenario, kept as a tool: `SVDSS search --bam` on a BAM of many small device batches (1 MB), six feeding
threads, eight segments per read, R runs per configuration; prints the md5 of each run's stdout.  The library is taken from
LD_LIBRARY_PATH when given (the binary's RUNPATH comes second): `make -C svdss_amd/csrc hazard` + a copy named
libsvdss_hip.so shows round 4's build.    python tools/seg_stress_bam.py [runs]"""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import svdss_amd  # noqa: E402
from tests import test_bam_device_gpu as T  # noqa: E402
from tests.common import BIN, small_workload  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
work = os.environ.get("SEG_STRESS_DIR", "/tmp/seg_stress")
os.makedirs(work, exist_ok=True)
ref, hap, svs, flat, offs = small_workload(seed=91, n_reads=400, read_len=1500, ref_lens=(150000,))
ix = svdss_amd.FMDIndex.build(ref).to_device(0)
reads = [flat[offs[i]:offs[i + 1]].copy() for i in range(400)]
reads[14] = np.concatenate([reads[14], reads[15], reads[16], reads[17]])
names = [f"m64/{(i * 37) % 400}/ccs" for i in range(400)]
rng = np.random.default_rng(23)
recs = []
for rep in range(8):
    r, _ = T._records([f"{n}/{rep}" for n in names], reads, rng, decoys=(rep % 2 == 0))
    recs += r
bam, fmd = os.path.join(work, "reads.bam"), os.path.join(work, "ref.fmd")
open(bam, "wb").write(T._bgzf_levels(T._raw_bam([("chr1", 150000)], recs), rng, block=60000))
ix.save(fmd)
del ix
base = {"SVDSS_BAM_BATCH_MB": "1", "SVDSS_BAM_SLAB_KB": "64"}
for name, env in [("BS=1 8 segments", dict(base, SVDSS_BS="1", SVDSS_SEGMENTS="8")),
                  ("BS=0 8 segments", dict(base, SVDSS_BS="0", SVDSS_SEGMENTS="8")),
                  ("BS=1 1 segment ", dict(base, SVDSS_BS="1", SVDSS_SEGMENTS="1"))]:
    hs = []
    for rep in range(runs):
        r = subprocess.run([BIN, "search", "--index", fmd, "--bam", bam, "--threads", "4", "--bsize", "100"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        hs.append(hashlib.md5(r.stdout.encode()).hexdigest()[:8] if r.returncode == 0 else f"rc{r.returncode}")
    print(f"{name}: {len(set(hs))} distinct output(s) in {runs} runs: {' '.join(hs)}", flush=True)
