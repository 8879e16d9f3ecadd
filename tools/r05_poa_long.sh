#!/bin/bash
# round 5: lone long sub-clusters (16 x 30 x 2.6 kb), phase timers of the quad stage per group width, and the old kernel
mkdir -p gpurun_out
{
for gw in 16 32 64; do
  echo "== SVDSS_POA_QUAD_GW=$gw"
  SVDSS_DEBUG=1 SVDSS_POA_QUAD_GW=$gw PYTHONPATH=. timeout 600 python tools/poa_long_probe.py 16 2600 30 2>&1 | grep -v amdgpu.ids | tail -4
  SVDSS_DEBUG=1 SVDSS_POA_QUAD_GW=$gw PYTHONPATH=. timeout 600 python tools/poa_long_probe.py 16 900 15 2>&1 | grep -v amdgpu.ids | tail -2
done
echo "== SVDSS_POA_QUAD=0"
SVDSS_DEBUG=1 SVDSS_POA_QUAD=0 PYTHONPATH=. timeout 600 python tools/poa_long_probe.py 16 2600 30 2>&1 | grep -v amdgpu.ids | tail -4
SVDSS_DEBUG=1 SVDSS_POA_QUAD=0 PYTHONPATH=. timeout 600 python tools/poa_long_probe.py 16 900 15 2>&1 | grep -v amdgpu.ids | tail -2
} > gpurun_out/r05_poa_long.txt 2>&1
cat gpurun_out/r05_poa_long.txt
