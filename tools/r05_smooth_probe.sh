#!/bin/bash
# round 5: `SVDSS smooth` on the bench's BAM (1,032,000 x 15 kb reads): the device path against the host pipeline, to a file and to /dev/null
export PYTHONPATH=$PWD
W=/tmp/svdss_r05_e2e
mkdir -p gpurun_out
{
R04_ONLY_BUILD=1 timeout 900 python tools/r04_e2e.py 1032000 $W > /dev/null 2>&1
for cfg in "SVDSS_X=1" "SVDSS_BAM_BATCH_MB=64" "SVDSS_BAM_BATCH_MB=192"; do
for dst in /dev/null $W/sm.bam; do rm -f $W/sm.bam; sync
  echo "## $cfg smooth > $dst"
  s=$(date +%s.%N)
  env $cfg SVDSS_DEBUG=1 timeout 300 svdss_amd/SVDSS smooth --reference $W/chr.fa --bam $W/reads.bam --threads 16 > $dst 2> gpurun_out/smooth_verbose.txt
  e=$(date +%s.%N)
  python3 -c "print(\"wall\", round($e - $s, 2), \"s =\", round(1032000 / ($e - $s)), \"reads/s\")"
  grep "smooth\]" gpurun_out/smooth_verbose.txt | cut -c1-420 | tail -6
done
done
ls -la $W/sm.bam
} > gpurun_out/r05_smooth_probe.txt 2>&1
cat gpurun_out/r05_smooth_probe.txt
