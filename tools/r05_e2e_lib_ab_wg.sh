#!/bin/bash
# round 5 (second session): as tools/r05_e2e_lib_ab.sh, against a GRCh38-length index (contig 1 begins with the 64 Mb the reads come from)
cd "$(dirname "$0")/.."
W=/tmp/svdss_r04_e2e
R04_ONLY_BUILD=1 python tools/r04_e2e.py 1032000 $W > /dev/null 2>&1
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from bench import GRCH38_PRIMARY
W = "/tmp/svdss_r04_e2e"
first = open(W + "/chr.fa", "rb").read().split(b"\n")[1]
rng = np.random.default_rng(5)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
with open(W + "/wg.fa", "wb") as f:
    for i, n in enumerate(GRCH38_PRIMARY):
        f.write(b">c%d\n" % (i + 1))
        if i == 0:
            f.write(first); n -= len(first)
        for a in range(0, n, 1 << 26):
            f.write(lut[rng.integers(0, 4, size=min(1 << 26, n - a), dtype=np.uint8)].tobytes())
        f.write(b"\n")
PY
svdss_amd/SVDSS index -d $W/wg.fa -o $W/wg.fmd > /dev/null 2>&1
sync
for k in 1 2 3; do
  for lib in new old; do
    if [ $lib = old ]; then export LD_LIBRARY_PATH=$PWD/oldlib; else unset LD_LIBRARY_PATH; fi
    SVDSS_DEBUG=1 svdss_amd/SVDSS search --index $W/wg.fmd --bam $W/reads.bam --noputative --verbose 2>&1 > /dev/null | python3 -c "
import re,sys
s=sys.stdin.read()
ix=float(re.search(r'on the device at \+([0-9.]+) s', s).group(1)); e=float(re.search(r'SFS written at \+([0-9.]+) s', s).group(1))
d=re.search(r'device batches, seconds summed: (.*?); the batchers', s)
print('$lib run $k: index resident +%.3f, streaming %.3f s | %s' % (ix, e-ix, d.group(1) if d else s[-300:]))"
  done
done
