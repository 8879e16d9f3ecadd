#!/bin/bash
# round 5 (second session): the bench's e2e_wg leg (reads with 0.5 % errors from the first 64 Mb of the bench's whole-genome reference, --noputative) by restore variant
cd "$(dirname "$0")/.."
python3 - <<'PY'
import os, sys, subprocess, re, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from svdss_amd import synth
from tools import e2e_search as E
W = "/tmp/e2ewg"; os.makedirs(W, exist_ok=True)
ref = synth.make_reference(bench.GRCH38_PRIMARY, seed=11)
bench.e2e_prepare(W, ref, wg=True)
del ref
ref0 = np.load(W + "/ref0.npy")
bam = W + "/reads.bam"
E.write_bam(bam, "chrS", ref0, 172000, 15000, repeat=6)
os.sync()
exe = "svdss_amd/SVDSS"
subprocess.run([exe, "index", "-d", W + "/wg.fa", "-o", W + "/wg.fmd"], check=True, capture_output=True)
os.remove(W + "/wg.fa")
cfgs = [("warm-up", {}), ("arena (default)", {}), ("no arena", {"SVDSS_INDEX_NO_ARENA": "1"}), ("arena (default)", {})]
for name, env in cfgs:
    for rep in range(2):
        r = subprocess.run([exe, "search", "--index", W + "/wg.fmd", "--bam", bam, "--noputative", "--verbose"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, SVDSS_DEBUG="1", SVDSS_INDEX_VERBOSE="1", **env))
        ix = float(re.search(r"on the device at \+([0-9.]+) s", r.stderr).group(1)); e = float(re.search(r"SFS written at \+([0-9.]+) s", r.stderr).group(1))
        d = re.search(r"device batches, seconds summed: (.*?); the batchers", r.stderr)
        ph = " ".join(m.group(1) + "@" + m.group(2) for m in re.finditer(r"\[index_gpu\] (suffix array \+ rank buffers|suffixes sorted|counters, blocks to the host)\s+at \+([0-9.]+)", r.stderr))
        tb = re.search(r"k-mer table filled at \+([0-9.]+)", r.stderr)
        print(f"   phases: {ph} | table filled +{tb.group(1) if tb else '?'}")
        import hashlib
        km = [float(x) for x in re.findall(r"segmented search \+ stitch: ([0-9.]+) ms", r.stderr)]
        print(f"   output md5 {hashlib.md5(r.stdout.encode()).hexdigest()[:12]}, {len(km)} segmented launches, {sum(km):.0f} ms summed (events on the launch streams)")
        print(f"{name} run {rep}: index resident +{ix:.2f}, streaming {e - ix:.3f} s | {d.group(1) if d else ''}", flush=True)
PY
