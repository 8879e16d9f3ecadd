#!/bin/bash
# round 4: the search kernel with / without BS (backward phases on deep intervals by binary search of the suffix array):
# the headline workload (must not get slower) and the repeat-family references
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04h; mkdir -p $O
export PYTHONPATH=$PWD
for cfg in "auto 256" "0 0"; do
  set -- $cfg
  if [ "$1" = auto ]; then unset SVDSS_BS; export SVDSS_INDEX_VERBOSE=1; else export SVDSS_BS=$1; unset SVDSS_INDEX_VERBOSE; fi
  echo "== SVDSS_BS=$1 SVDSS_BS_AFTER=$2, headline workload (wg)"; SVDSS_BS_AFTER=$2 timeout 900 python tools/search_only.py wg 1048576 4 2>&1 | grep "search kernel\|sampled K-mer" | tail -2
  for d in 0.15 0.05 0.01; do
    echo "== SVDSS_BS=$1 SVDSS_BS_AFTER=$2, families 0.45:$d"; SVDSS_BS_AFTER=$2 timeout 900 python tools/search_only.py wg 1048576 3 families:0.45:$d 2>&1 | grep "search kernel\|sampled K-mer" | tail -2
  done
done > $O/search_bs.txt 2>&1
cat $O/search_bs.txt
