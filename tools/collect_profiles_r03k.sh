#!/bin/bash
# Round 3, end: the GPU suite, the driver's bench command, kernel statistics of the pipelined step and the call-side DP
# counters with the final POA kernel (tagged chain rows).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/gpu_suite.log 2>&1; tail -9 $O/gpu_suite.log
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
tail -c 600 $O/bench_driver_command.json | head -c 300; echo
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
head -8 $O/kernel_stats.csv | cut -c1-150
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "poa_|align_wave|lcs_" --output-format csv -d $O/pmccall_$i -- python $R/tools/call_dp_probe.py 3395 2 > $O/call_dp_probe_$i.log 2>&1
done
python - <<PY
import csv, glob, re
rows = []
for f in sorted(glob.glob("$O/pmccall_*/**/*counter_collection.csv", recursive=True)):
    acc, n = {}, {}
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70], row["Counter_Name"])
        acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    for (kern, ctr), v in sorted(acc.items()):
        if re.search("poa_|align_|lcs_", kern):
            rows.append((kern, ctr, n[(kern, ctr)], v / n[(kern, ctr)]))
with open("$O/pmc_calldp.csv", "w") as fh:
    fh.write("Kernel,Counter,Dispatches,MeanValuePerDispatch\n")
    for r in rows:
        fh.write("%s,%s,%d,%.1f\n" % r)
print(open("$O/pmc_calldp.csv").read())
PY
rm -rf $O/pmccall_[0-9]*
