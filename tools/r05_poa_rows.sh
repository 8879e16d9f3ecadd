#!/bin/bash
# round 5: group width by chain length (reads x length): alone, 3 batches in flight, and the bench
mkdir -p gpurun_out
{
for cfg in "SVDSS_POA_QUAD_ROWS16=0" "SVDSS_POA_QUAD_ROWS16=12000" "SVDSS_POA_QUAD_ROWS16=20000" "SVDSS_POA_QUAD_ROWS16=30000" "SVDSS_POA_QUAD_ROWS16=20000 SVDSS_POA_QUAD_ROWS32=45000" "SVDSS_POA_QUAD_ROWS16=40000 SVDSS_POA_QUAD_ROWS32=60000"; do
  echo "== $cfg"
  env $cfg SVDSS_DEBUG=1 timeout 600 python tools/call_dp_probe.py 3395 3 2>&1 | grep "^run 2\|round"  | tail -3
  env $cfg timeout 600 python tools/call_dp_concurrent.py 3 4 2>&1 | grep -v amdgpu.ids | tail -1
done
for cfg in "SVDSS_POA_QUAD_ROWS16=0" "SVDSS_POA_QUAD_ROWS16=20000" "SVDSS_POA_QUAD_ROWS16=20000 SVDSS_POA_QUAD_ROWS32=45000"; do
  echo "== bench $cfg"
  env $cfg timeout 900 python bench.py --steps 12 --warmup 4 --no-e2e --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config'].get('call_dp', {})
print('bench', d['value'], d['ms_per_step'], {k: c.get(k) for k in ('poa_kernel_ms', 'realign_kernel_ms', 'poa_gcups')})
"
done
} > gpurun_out/r05_poa_rows.txt 2>&1
cat gpurun_out/r05_poa_rows.txt
