#!/bin/bash
# round 5: the search kernel as workgroups of one wavefront (SV_TPB=64): does the dispatcher fit it better between the call side's wavefronts?
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/svdss_amd
run() {
  echo "== $*"
  for rep in 1 2 3; do
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
run SVDSS_LIB=$L/libsvdss_hip_occ_t64.so
run SVDSS_LIB=$L/libsvdss_hip_occ_t64q4.so
} > gpurun_out/r05_tpb_ab.txt 2>&1
cat gpurun_out/r05_tpb_ab.txt
