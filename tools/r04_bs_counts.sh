#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04h; mkdir -p $O
export PYTHONPATH=$PWD SVDSS_LIB=$PWD/svdss_amd/libsvdss_hip_count.so SVDSS_DEBUG=1
for d in 0.01 0.05; do for cfg in "0 0" "1 0" "1 64"; do
  set -- $cfg
  echo "== families 0.45:$d SVDSS_BS=$1 SVDSS_BS_AFTER=$2"
  SVDSS_BS=$1 SVDSS_BS_AFTER=$2 timeout 900 python tools/search_only.py wg 262144 1 families:0.45:$d 2>&1 | grep "lane ops\|search kernel" | tail -2
done; done > $O/bs_op_counts.txt 2>&1
cat $O/bs_op_counts.txt
