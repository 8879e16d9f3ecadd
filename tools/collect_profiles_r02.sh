#!/bin/bash
# Round-2 rocprofv3 evidence on the GPU box (run through gpurun), for bench.py's default workload
# (GRCh38 lengths, 1,048,576 reads per step):
#   gpurun_out/r02/bench_under_rocprof.json + kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python bench.py
#   gpurun_out/r02/pmc_wg.csv        separate --pmc passes on the search launches of the same workload
#   gpurun_out/r02/pmc_calldp.csv    the same for the call-side DP kernels (tools/call_dp_probe.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
summarize() {  # $1 = directory glob prefix, $2 = regex of kernel names to keep, $3 = output csv
python - <<PY
import csv, glob, re
rows = []
for f in sorted(glob.glob("$1*/**/*counter_collection.csv", recursive=True)):
    acc, n = {}, {}
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][:70], row["Counter_Name"])
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
for c in "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCC_HIT TCC_MISS TCC_REQ" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcwg_$i -- python $R/tools/search_only.py wg 1048576 3 > $O/search_only_$i.log 2>&1
done
summarize "$O/pmcwg_" "sfs_" "$O/pmc_wg.csv"
rm -rf $O/pmcwg_[0-9]*
# the chr20 workload of round 1 (128,888 reads, several lanes per read) with the same kernel
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcchr20_1 -- python $R/tools/search_only.py chr20 128888 5 > $O/search_only_chr20.log 2>&1
summarize "$O/pmcchr20_" "sfs_" "$O/pmc_chr20.csv"
rm -rf $O/pmcchr20_[0-9]*
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "poa_|align_wave|lcs_ratio" --output-format csv -d $O/pmccall_$i -- python $R/tools/call_dp_probe.py 3395 2 > /dev/null 2>&1
done
summarize "$O/pmccall_" "poa_|align_|lcs_" "$O/pmc_calldp.csv"
rm -rf $O/pmccall_[0-9]*
head -14 $O/kernel_stats.csv | cut -c1-170
tail -c 600 $O/bench_under_rocprof.json
cat $O/search_only_1.log | tail -2
