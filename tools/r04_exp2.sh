#!/bin/bash
# round 4, second GPU batch: search kernel with 3 / 4 workgroups per CU beside POA builds of 126 / 96 / 80 / 64 registers:
# do the two kernels overlap when both fit a SIMD's register file?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04b; mkdir -p $O
for blocks in 1024 768; do
for v in "" _poa5 _poa6 _poa8; do
  SVDSS_BLOCKS=$blocks SVDSS_LIB=$PWD/svdss_amd/libsvdss_hip$v.so timeout 600 python bench.py --steps 10 --warmup 3 --no-e2e > $O/bench_b${blocks}$v.json 2> $O/bench_b${blocks}$v.err
  python - <<PY
import json
d=json.loads(open("$O/bench_b${blocks}$v.json").read().strip().splitlines()[-1])
print("blocks $blocks lib '$v':", round(d["value"]), "reads/s", round(d["ms_per_step"],1), "ms/step")
PY
done; done | tee $O/summary.txt
PYTHONPATH=$PWD python tools/poa_long_probe.py 16 2600 30 > $O/poa_long_default.txt 2>&1
SVDSS_LIB=$PWD/svdss_amd/libsvdss_hip_poa5.so PYTHONPATH=$PWD python tools/poa_long_probe.py 16 2600 30 > $O/poa_long_poa5.txt 2>&1
