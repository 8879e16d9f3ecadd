#!/bin/bash
# Round 3, part b: kernel statistics of the driver's command, the call-side DP counters after the POA row-maximum change,
# the repeat-rich whole-genome datapoints, and `SVDSS smooth` with the GPU deflate encoder against the host's deflate.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
# 1. rocprofv3 --kernel-trace --stats of bench.py (the pipelined step)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
head -12 $O/kernel_stats.csv | cut -c1-160
# 2. call-side DP counters
summarize() {
python - <<PY
import csv, glob, re
rows = []
for f in sorted(glob.glob("$1*/**/*counter_collection.csv", recursive=True)):
    acc, n = {}, {}
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70], row["Counter_Name"])
        acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    for (kern, ctr), v in sorted(acc.items()):
        if re.search("$2", kern):
            rows.append((kern, ctr, n[(kern, ctr)], v / n[(kern, ctr)]))
with open("$3", "w") as fh:
    fh.write("Kernel,Counter,Dispatches,MeanValuePerDispatch\n")
    for r in rows:
        fh.write("%s,%s,%d,%.1f\n" % r)
print(open("$3").read())
PY
}
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "poa_|align_wave|lcs_" --output-format csv -d $O/pmccall_$i -- python $R/tools/call_dp_probe.py 3395 2 > $O/call_dp_probe_$i.log 2>&1
done
summarize "$O/pmccall_" "poa_|align_|lcs_" "$O/pmc_calldp.csv"
rm -rf $O/pmccall_[0-9]*
tail -3 $O/call_dp_probe_1.log
# 3. repeat-rich whole-genome references (45 % family repeats), search + call as in the default bench
for d in 0.15 0.05 0.01; do
  timeout 900 python $R/bench.py --workload wg-families --divergence $d --steps 8 --warmup 2 --no-e2e --cpu-seconds 6 > $O/bench_wg_families_$d.json 2> $O/bench_wg_families_$d.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_wg_families_$d.json").read().strip().splitlines()[-1])
    print("families $d:", round(d["value"]), "reads/s", round(d["ms_per_step"], 1), "ms/step; search kernel on idle GPU", round(d["roofline"]["kernel_ms_on_idle_gpu"], 1), "ms; verified", d.get("verified_reads"))
except Exception as e:
    print("families $d: FAILED", e)
PY
done
# 4. SVDSS smooth, GPU deflate vs host deflate (43,000 x 15 kb reads)
W=/tmp/svdss_e2e
timeout 600 python $R/tools/e2e_search.py 64444167 43000 15000 $W > $O/e2e_search_43k.json 2>&1
for mode in gpu host; do
  if [ $mode = host ]; then export SVDSS_GPU_DEFLATE=0; else unset SVDSS_GPU_DEFLATE; fi
  for rep in 1 2; do
    s=$(date +%s.%N)
    SVDSS_DEBUG=1 $R/svdss_amd/SVDSS smooth --reference $W/ref.fa --bam $W/reads.bam --threads 32 > $W/smoothed_$mode.bam 2> $O/smooth_$mode.err
    e=$(date +%s.%N)
    echo "smooth deflate=$mode run $rep: $(python3 -c "print(round($e - $s, 2))") s, $(stat -c %s $W/smoothed_$mode.bam) bytes" | tee -a $O/smooth_times.txt
  done
done
unset SVDSS_GPU_DEFLATE
python - <<PY | tee -a $O/smooth_times.txt
import gzip
a = gzip.open("$W/smoothed_gpu.bam").read(); b = gzip.open("$W/smoothed_host.bam").read()
print("inflated streams identical:", a == b, len(a))
PY
grep -h "\[smooth\]" $O/smooth_gpu.err | tail -8
