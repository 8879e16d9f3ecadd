#!/bin/bash
# round 5: the headline with the quad POA stage per group width x call threads
mkdir -p gpurun_out
{
for cfg in "SVDSS_POA_QUAD_GW=16" "SVDSS_POA_QUAD_GW=32" "SVDSS_POA_QUAD=0"; do
for ct in 4 6; do
  echo "== $cfg call-threads $ct"
  env $cfg timeout 900 python bench.py --steps 12 --warmup 4 --no-e2e --no-cpu-baseline --call-threads $ct 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config'].get('call_dp', {})
print(d['value'], d['ms_per_step'], {k: c.get(k) for k in ('poa_kernel_ms', 'realign_kernel_ms', 'poa_gcups')}, d.get('roofline', {}).get('frac'))
"
done
done
} > gpurun_out/r05_bench_quad2.txt 2>&1
cat gpurun_out/r05_bench_quad2.txt
