#!/bin/bash
mkdir -p gpurun_out
{
for cls in "4096 600 700 20" "4096 1000 1300 20"; do
for cfg in "SVDSS_POA_QUAD=0" "SVDSS_POA_QUAD_GW=64" "SVDSS_POA_QUAD_GW=32" "SVDSS_POA_QUAD_GW=16"; do
  for T in 1 3; do
    echo -n "$cfg: "
    env $cfg timeout 300 python tools/poa_class_probe.py $cls $T 3 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
done
} > gpurun_out/r05_poa_class.txt 2>&1
cat gpurun_out/r05_poa_class.txt
