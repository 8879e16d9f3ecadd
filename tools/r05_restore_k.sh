#!/bin/bash
# round 5: index restore and streaming of `SVDSS search` against the GRCh38-length index by k-mer table order (SVDSS_KMER)
mkdir -p gpurun_out
python - > gpurun_out/r05_restore_k.txt 2>&1 <<'PY'
import os, sys, subprocess, time, re
sys.path.insert(0, os.getcwd())
from tools import e2e_call_wg as W
fa, bam, svs, n, lens = W.write_dataset("/tmp/callwg", 1030000, 3400)
exe = "svdss_amd/SVDSS"
subprocess.run([exe, "index", "-d", fa, "-o", "/tmp/callwg/ref.fmd"], check=True, capture_output=True)
for k in ("16", "15", "14", "16"):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "search", "--index", "/tmp/callwg/ref.fmd", "--bam", bam, "--verbose"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, SVDSS_INDEX_VERBOSE="1", SVDSS_DEBUG="1", SVDSS_KMER=k))
        wall = time.time() - t0
        res = float(re.search(r"on the device at \+([0-9.]+) s", r.stderr).group(1))
        end = float(re.search(r"SFS written at \+([0-9.]+) s", r.stderr).group(1))
        tab = re.search(r"k-mer table of order (\d+).*", r.stderr)
        fill = re.search(r"k-mer table filled at \+([0-9.]+)", r.stderr)
        srt = re.search(r"suffixes sorted\s+at \+([0-9.]+)", r.stderr)
        host = re.search(r"counters, blocks to the host\s+at \+([0-9.]+)", r.stderr)
        print(f"K={k} run {rep}: wall {wall:.2f} s, index resident at +{res:.2f} s, streaming {end - res:.2f} s | sorted at +{srt.group(1) if srt else '?'}, "
              f"build done +{host.group(1) if host else '?'}, table filled +{fill.group(1) if fill else '?'} | {tab.group(0)[:70] if tab else ''}", flush=True)
PY
cat gpurun_out/r05_restore_k.txt
