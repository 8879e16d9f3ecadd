#!/bin/bash
# round 5: throughput of the call side alone, T batches in flight, per POA variant
mkdir -p gpurun_out
{
for cfg in "SVDSS_POA_QUAD=0" "SVDSS_POA_QUAD_GW=64" "SVDSS_POA_QUAD_GW=32" "SVDSS_POA_QUAD_GW=16"; do
  for T in 1 3 5; do
    echo -n "$cfg: "
    env $cfg timeout 600 python tools/call_dp_concurrent.py $T 4 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r05_call_tput.txt 2>&1
cat gpurun_out/r05_call_tput.txt
