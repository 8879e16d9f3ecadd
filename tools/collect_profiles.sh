#!/bin/bash
# Collects the round's rocprofv3 evidence for the default bench workload on the GPU box (run through gpurun):
#   gpurun_out/final_stats/     rocprofv3 --kernel-trace --stats of `python bench.py --steps 5`
#   gpurun_out/final_pmc_*/     separate --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC / SQ) of the same command
# Summaries are written to gpurun_out/final_kernel_stats.csv and gpurun_out/final_pmc.csv for profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-call-dp"
# counters: the same workload and launches without the torch kernels of the read simulator around them
PMC_CMD="python $R/tools/sweep_search.py 0 0 128888"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_stats -- $CMD > $R/gpurun_out/final_bench_under_rocprof.json 2>/dev/null
cp $(ls $R/gpurun_out/final_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/final_kernel_stats.csv
i=0
for c in "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCC_HIT TCC_MISS TCC_REQ" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex sfs_ --output-format csv -d $R/gpurun_out/final_pmc_$i -- $PMC_CMD > /dev/null 2>&1
done
python - <<PY
import csv, glob
rows = []
for f in sorted(glob.glob("$R/gpurun_out/final_pmc_*/**/*counter_collection.csv", recursive=True)):
    acc, n = {}, {}
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
        acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    for (kern, ctr), v in sorted(acc.items()):
        if "sfs_" in kern:
            rows.append((kern, ctr, n[(kern, ctr)], v / n[(kern, ctr)]))
with open("$R/gpurun_out/final_pmc.csv", "w") as fh:
    fh.write("Kernel,Counter,Dispatches,MeanValuePerDispatch\n")
    for r in rows:
        fh.write("%s,%s,%d,%.1f\n" % r)
print(open("$R/gpurun_out/final_pmc.csv").read())
PY
rm -rf $R/gpurun_out/final_stats $R/gpurun_out/final_pmc_[0-9]*
head -12 $R/gpurun_out/final_kernel_stats.csv | cut -c1-160
tail -c 1500 $R/gpurun_out/final_bench_under_rocprof.json
