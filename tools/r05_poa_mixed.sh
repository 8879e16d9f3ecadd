#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_poa_quad_gpu.py tests/test_poa_gpu.py -x -q 2>&1 | tail -5
for cfg in "SVDSS_POA_QUAD_SHORT=1000" "SVDSS_POA_QUAD_SHORT=1400" "SVDSS_POA_QUAD_SHORT=0" "SVDSS_POA_QUAD=0"; do
  echo "== $cfg"
  env $cfg SVDSS_DEBUG=1 timeout 600 python tools/call_dp_probe.py 3395 3 2>&1 | grep -v "amdgpu.ids\|poa_wave\]" | tail -4
  env $cfg timeout 600 python tools/call_dp_concurrent.py 3 4 2>&1 | grep -v amdgpu.ids | tail -1
  env $cfg timeout 900 python bench.py --steps 12 --warmup 4 --no-e2e --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config'].get('call_dp', {})
print('bench', d['value'], d['ms_per_step'], {k: c.get(k) for k in ('poa_kernel_ms', 'realign_kernel_ms', 'poa_gcups')})
"
done
} > gpurun_out/r05_poa_mixed.txt 2>&1
cat gpurun_out/r05_poa_mixed.txt
