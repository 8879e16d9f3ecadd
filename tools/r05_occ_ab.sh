#!/bin/bash
# round 5: does the step shrink when the search kernel leaves registers for the call side's wavefronts?
# (search at 96 registers, four workgroups per CU by an LDS pad; POA first stage at <= 128 registers)
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/svdss_amd
run() {
  echo "== $*"
  for rep in 1 2; do
  env "$@" timeout 900 python bench.py --steps 12 --warmup 4 --no-e2e --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config'].get('call_dp', {})
r = d['roofline']
print('bench %.0f reads/s  step %.1f ms | search kernel %.1f ms in the step, %.1f alone | POA %.0f realign %.0f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_on_idle_gpu'], c.get('poa_kernel_ms', 0), c.get('realign_kernel_ms', 0)))
"
  done
}
{
run X=1
run SVDSS_LIB=$L/libsvdss_hip_occ_s4q4.so
run SVDSS_LIB=$L/libsvdss_hip_occ_s5q4.so SVDSS_EXTRA_LDS=5120
run SVDSS_LIB=$L/libsvdss_hip_occ_s5q4.so
run SVDSS_LIB=$L/libsvdss_hip_occ_s5q4.so SVDSS_EXTRA_LDS=5120 SVDSS_POA_QUAD=0
} > gpurun_out/r05_occ_ab.txt 2>&1
cat gpurun_out/r05_occ_ab.txt
