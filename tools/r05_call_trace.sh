#!/bin/bash
# round 5: kernel trace of the call side alone (3 batches in flight): how busy is the GPU, which kernels fill the time
mkdir -p gpurun_out/r05_trace
cd /tmp && export TMPDIR=/tmp
for cfg in "SVDSS_POA_QUAD=0" "SVDSS_POA_QUAD_GW=64"; do
  tag=$(echo $cfg | tr '=' '_')
  env $cfg rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_trace/$tag -o t -- python $GRAFT_REPO_ROOT/tools/call_dp_concurrent.py 3 4 2>&1 | grep -v amdgpu.ids | tail -2
done
cd $GRAFT_REPO_ROOT
for d in gpurun_out/r05_trace/*; do
  echo "== $d"; f=$(find $d -name "*kernel_stats.csv" | head -1); head -8 $f
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]) for r in rows)
# skip the warm-up third
t0 = ev[len(ev) // 3][0]
ev = [e for e in ev if e[0] >= t0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e, _ in ev) - ev[0][0]
print(f"span {span/1e6:.1f} ms, some kernel running {busy/1e6:.1f} ms ({100*busy/span:.0f} %)")
PY
done > gpurun_out/r05_call_trace.txt 2>&1
find gpurun_out/r05_trace -name "*.csv" -size +2M -delete
cat gpurun_out/r05_call_trace.txt
