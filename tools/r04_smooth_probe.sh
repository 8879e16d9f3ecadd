#!/bin/bash
# round 4: where `SVDSS smooth` spends its time on the bench's BAM (1,032,000 x 15 kb reads): to a file, to /dev/null
cd /root/repo; export PYTHONPATH=/root/repo
W=/tmp/svdss_r04_e2e
R04_ONLY_BUILD=1 python tools/r04_e2e.py 1032000 $W > /dev/null 2>&1
for dst in $W/sm.bam /dev/null; do
  echo "## smooth > $dst"
  s=$(date +%s.%N)
  SVDSS_DEBUG=1 svdss_amd/SVDSS smooth --reference $W/chr.fa --bam $W/reads.bam --threads 16 > $dst 2> gpurun_out/smooth_verbose.txt
  e=$(date +%s.%N)
  python3 -c "print(\"wall\", round($e - $s, 2), \"s\")"
  grep -v "^$" gpurun_out/smooth_verbose.txt | cut -c1-400 | tail -12
done
ls -la $W/sm.bam
