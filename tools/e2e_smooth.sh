#!/bin/bash
# Developer benchmark: `SVDSS smooth` on the BAM written by tools/e2e_search.py (run that first), 1 thread vs many;
# the outputs must be byte-identical.
W=${1:-/tmp/svdss_e2e}
B=$(dirname $0)/../svdss_amd/SVDSS
for t in 1 32; do
  s=$(date +%s.%N)
  $B smooth --reference $W/ref.fa --bam $W/reads.bam --threads $t > $W/smoothed_$t.bam 2> $W/smooth_$t.err || { tail -3 $W/smooth_$t.err; exit 1; }
  e=$(date +%s.%N)
  echo "threads=$t: $(python3 -c "print(round($e - $s, 2))") s, $(stat -c %s $W/smoothed_$t.bam) bytes"
done
cmp $W/smoothed_1.bam $W/smoothed_32.bam && echo "outputs identical"
