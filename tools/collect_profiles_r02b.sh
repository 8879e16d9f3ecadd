#!/bin/bash
# Round-2 (second half) evidence on the GPU box, run through gpurun:
#   gpurun_out/r02b/gpu_suite.log                    python -m pytest tests -m gpu
#   gpurun_out/r02b/bench_default.json               python bench.py (the driver's command)
#   gpurun_out/r02b/bench_under_rocprof.json + kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python bench.py ...
#   gpurun_out/r02b/pmc_calldp.csv                   separate --pmc passes on the call-side DP kernels
#   gpurun_out/r02b/pmc_inflate.csv                  the same for the inflate kernel (tools/inflate_probe.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
summarize() {  # $1 = directory glob prefix, $2 = regex of kernel names to keep, $3 = output csv
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
export PYTHONPATH=$R
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "poa_|align_wave|lcs_ratio" --output-format csv -d $O/pmccall_$i -- python $R/tools/call_dp_probe.py 3395 2 > /dev/null 2>&1
done
summarize "$O/pmccall_" "poa_|align_|lcs_" "$O/pmc_calldp.csv"
rm -rf $O/pmccall_[0-9]*
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "bgzf_inflate" --output-format csv -d $O/pmcinf_$i -- python $R/tools/inflate_probe.py 8192 1 bam > $O/inflate_probe_$i.log 2>&1
done
summarize "$O/pmcinf_" "bgzf_inflate" "$O/pmc_inflate.csv"
rm -rf $O/pmcinf_[0-9]*
head -16 $O/kernel_stats.csv | cut -c1-170
