#!/bin/bash
# round 5: the call-side DP of a whole 30x genome handed over at once (20,000 clusters -> ~30,000 sub-clusters), per POA variant
mkdir -p gpurun_out
{
for cfg in "SVDSS_POA_QUAD=0" "SVDSS_POA_QUAD_GW=64" "SVDSS_POA_QUAD_GW=32" "SVDSS_POA_QUAD_GW=16" "SVDSS_POA_QUAD_ROWS16=20000" "SVDSS_POA_QUAD_ROWS16=40000 SVDSS_POA_QUAD_ROWS32=60000"; do
  echo "== $cfg"
  env $cfg SVDSS_DEBUG=1 timeout 900 python tools/call_dp_probe.py 20000 2 2>&1 | grep "^workload\|^run 1\|round\|implanted" | tail -6
done
} > gpurun_out/r05_poa_30x.txt 2>&1
cat gpurun_out/r05_poa_30x.txt
