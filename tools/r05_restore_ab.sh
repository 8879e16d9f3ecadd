#!/bin/bash
# round 5 (second session): does the way the index is restored change how fast `SVDSS search` streams afterwards?  (A/B of the second session's restore changes)
mkdir -p gpurun_out/r05z5
python - > gpurun_out/r05z5/restore_ab.txt 2>&1 <<'PY'
import os, sys, subprocess, time, re
sys.path.insert(0, os.getcwd())
from tools import e2e_call_wg as W
fa, bam, svs, n, lens = W.write_dataset("/tmp/callwg", 1030000, 3400)
exe = "svdss_amd/SVDSS"
subprocess.run([exe, "index", "-d", fa, "-o", "/tmp/callwg/ref.fmd"], check=True, capture_output=True)
for env in ({}, {"SVDSS_INDEX_SERIAL_READ": "1"}, {}):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "search", "--index", "/tmp/callwg/ref.fmd", "--bam", bam, "--verbose"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, SVDSS_INDEX_VERBOSE="1", **env))
        wall = time.time() - t0
        rd = float(re.search(r"index file read at \+([0-9.]+) s", r.stderr).group(1))
        res = float(re.search(r"on the device at \+([0-9.]+) s", r.stderr).group(1))
        end = float(re.search(r"SFS written at \+([0-9.]+) s", r.stderr).group(1))
        st = re.search(r"device batches, seconds summed: (.*?); the batchers", r.stderr)
        tf = re.search(r"k-mer table filled at \+([0-9.]+)", r.stderr); bd = re.search(r"counters, blocks to the host\s+at \+([0-9.]+)", r.stderr)
        print(f"[build done +{bd.group(1) if bd else '?'}, table filled +{tf.group(1) if tf else '?'}] {env} run {rep}: wall {wall:.2f} s, file read +{rd:.2f}, index resident at +{res:.2f} s, streaming {end - res:.3f} s | {st.group(1) if st else ''}", flush=True)
PY
cat gpurun_out/r05z5/restore_ab.txt
