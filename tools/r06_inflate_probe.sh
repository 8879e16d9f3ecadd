#!/bin/bash
# round 6: the inflate kernel on run-heavy members (absent / HiFi-like qualities: matches of 258 bytes at distance 1)
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD
OUT=gpurun_out/${TAG:-r06j}; mkdir -p $OUT
for k in "6 absent" "1 absent" "6 hifi" "6 bam" "6 binned" "6 skew"; do
  set -- $k
  echo "== level $1 $2" >> $OUT/inflate_probe.txt
  timeout 300 python tools/inflate_probe.py 16384 $1 $2 2>&1 | grep "blocks,\|no   copy back\|verified" | tail -3 >> $OUT/inflate_probe.txt
  [ -f svdss_amd/libsvdss_hip_infcount.so ] && SVDSS_LIB=$PWD/svdss_amd/libsvdss_hip_infcount.so timeout 300 python tools/inflate_probe.py 4096 $1 $2 2>&1 | grep "inflate\]" | head -1 >> $OUT/inflate_probe.txt
done
cat $OUT/inflate_probe.txt
