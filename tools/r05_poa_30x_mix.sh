#!/bin/bash
# round 5: a 30x batch (30,000 sub-clusters) in waves of mixed chain lengths against waves cut from the sorted list
mkdir -p gpurun_out
{
for cfg in "SVDSS_POA_NO_MIX=1" "X=1" "SVDSS_POA_QUAD_ROWS16=20000" "SVDSS_POA_QUAD=0 SVDSS_POA_NO_MIX=1"; do
  echo "== $cfg"
  env $cfg SVDSS_DEBUG=1 timeout 900 python tools/call_dp_probe.py 20000 3 2>&1 | grep "^run [12]\|round\|implanted" | tail -9
done
} > gpurun_out/r05_poa_30x_mix.txt 2>&1
cat gpurun_out/r05_poa_30x_mix.txt
