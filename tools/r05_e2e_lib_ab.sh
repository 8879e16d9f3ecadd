#!/bin/bash
# round 5 (second session): `SVDSS search --bam --noputative` on the bench's e2e BAM (1,032,000 x 15 kb, 15.4 GB, chr20-length index), the tree's library against
# the library of commit e070aa5 (before the restore changes) under the same binary, runs interleaved -- is the slower e2e of the last bench runs the code or the box?
cd "$(dirname "$0")/.."
W=/tmp/svdss_r04_e2e
R04_ONLY_BUILD=1 python tools/r04_e2e.py 1032000 $W > /dev/null 2>&1
sync
for k in 1 2 3; do
  for lib in new old; do
    if [ $lib = old ]; then export LD_LIBRARY_PATH=$PWD/oldlib; else unset LD_LIBRARY_PATH; fi
    SVDSS_DEBUG=1 svdss_amd/SVDSS search --index $W/chr.fmd --bam $W/reads.bam --noputative --verbose 2>&1 > /dev/null | python3 -c "
import re,sys
s=sys.stdin.read()
ix=float(re.search(r'on the device at \+([0-9.]+) s', s).group(1)); e=float(re.search(r'SFS written at \+([0-9.]+) s', s).group(1))
d=re.search(r'device batches, seconds summed: (.*?); the batchers', s)
print('$lib run $k: index resident +%.3f, streaming %.3f s | %s' % (ix, e-ix, d.group(1) if d else ''))"
  done
done
