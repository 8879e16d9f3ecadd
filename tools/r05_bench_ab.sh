#!/bin/bash
# round 5: the headline, POA variants A/B with repeats (the step's run-to-run noise is 2-3 %)
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for cfg in "SVDSS_POA_QUAD=0" "SVDSS_POA_QUAD=1" "SVDSS_POA_QUAD_SHORT=1000" "SVDSS_POA_QUAD_GW=32"; do
  env $cfg timeout 900 python bench.py --steps 12 --warmup 4 --no-e2e --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config'].get('call_dp', {})
print('$cfg', round(d['value']), round(d['ms_per_step'], 1), {k: c.get(k) for k in ('poa_kernel_ms', 'realign_kernel_ms')})
"
done
done
} > gpurun_out/r05_bench_ab.txt 2>&1
cat gpurun_out/r05_bench_ab.txt
