#!/bin/bash
# round 5: the several-sub-clusters-per-wavefront POA stage on the GPU -- tests, then the call-side DP of one bench step with
# and without it (SVDSS_POA_QUAD=0), per group width
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_poa_quad_gpu.py tests/test_poa_gpu.py -x -q 2>&1 | tail -15
for gw in 16 32 64; do
  echo "== SVDSS_POA_QUAD_GW=$gw"
  SVDSS_DEBUG=1 SVDSS_POA_QUAD_GW=$gw timeout 600 python tools/call_dp_probe.py 3395 3 2>&1 | grep -v amdgpu.ids
done
echo "== SVDSS_POA_QUAD=0"
SVDSS_DEBUG=1 SVDSS_POA_QUAD=0 timeout 600 python tools/call_dp_probe.py 3395 3 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_poa_quad.txt 2>&1
tail -60 gpurun_out/r05_poa_quad.txt
