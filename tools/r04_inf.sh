#!/bin/bash
# round 4: the inflate kernel with passes -- parity first, then the probe on the kinds of blocks, then the counters
cd /root/repo; export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_inflate_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/inf_tests.txt
cat gpurun_out/inf_tests.txt
for k in "1 bam" "6 bam" "1 binned" "1 skew" "6 bam huff" "6 text"; do
  set -- $k
  echo "== level $1 $2 $3"
  SVDSS_LIB=${INF_LIB:-/root/repo/svdss_amd/libsvdss_hip.so} timeout 300 python tools/inflate_probe.py 16384 $1 $2 $3 2>&1 | grep "no   copy\|verified" | tail -2
  [ -z "$INF_NOCOUNT" ] && SVDSS_LIB=/root/repo/svdss_amd/libsvdss_hip_infcount.so timeout 300 python tools/inflate_probe.py 4096 $1 $2 $3 2>&1 | grep "inflate\]" | head -1
done > gpurun_out/inf_probe.txt 2>&1
cat gpurun_out/inf_probe.txt
