#!/bin/bash
# round 4: variants of the inflate kernel (make infvar) on the probe's kinds of blocks
cd /root/repo; export PYTHONPATH=/root/repo
mkdir -p gpurun_out
for lib in libsvdss_hip.so $(cd svdss_amd; ls libsvdss_hip_inf_*.so); do
  echo "#### $lib"
  for k in "1 bam" "6 bam" "1 binned" "1 skew" "6 bam huff"; do
    set -- $k
    echo -n "level $1 $2 $3: "
    SVDSS_LIB=/root/repo/svdss_amd/$lib timeout 300 python tools/inflate_probe.py 16384 $1 $2 $3 2>&1 | grep "no   copy\|verified" | tail -2 | tr '\n' ' '
    echo
  done
done > gpurun_out/inf_sweep.txt 2>&1
cat gpurun_out/inf_sweep.txt
