#!/bin/bash
# round 5: quad POA stage -- GPU tests, lone long sub-clusters (phase timers), one bench step's call-side DP, per group width
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_poa_quad_gpu.py tests/test_poa_gpu.py -x -q 2>&1 | tail -5
for gw in 16 32 64; do
  echo "== SVDSS_POA_QUAD_GW=$gw"
  SVDSS_DEBUG=1 SVDSS_POA_QUAD_GW=$gw PYTHONPATH=. timeout 600 python tools/poa_long_probe.py 16 2600 30 2>&1 | grep -v amdgpu.ids | tail -3
  SVDSS_DEBUG=1 SVDSS_POA_QUAD_GW=$gw timeout 600 python tools/call_dp_probe.py 3395 2 2>&1 | grep -v "amdgpu.ids\|poa_wave\]" | tail -4
done
echo "== SVDSS_POA_QUAD=0"
SVDSS_DEBUG=1 SVDSS_POA_QUAD=0 PYTHONPATH=. timeout 600 python tools/poa_long_probe.py 16 2600 30 2>&1 | grep -v amdgpu.ids | tail -3
} > gpurun_out/r05_poa_both.txt 2>&1
cat gpurun_out/r05_poa_both.txt
