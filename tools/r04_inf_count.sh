#!/bin/bash
cd /root/repo; export PYTHONPATH=/root/repo
for lib in $(cd svdss_amd; ls libsvdss_hip_inf_*count.so); do
echo "#### $lib"
for k in "1 bam" "6 bam" "1 binned"; do
  set -- $k
  echo -n "level $1 $2 $3: "
  SVDSS_LIB=/root/repo/svdss_amd/$lib timeout 300 python tools/inflate_probe.py 4096 $1 $2 $3 2>&1 | grep "inflate\]" | head -1
done; done > gpurun_out/inf_count.txt 2>&1
cat gpurun_out/inf_count.txt
