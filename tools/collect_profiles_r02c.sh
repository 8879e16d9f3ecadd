#!/bin/bash
# Round 2, last part (search kernel with the one-pass decision loop): evidence on the GPU box, run through gpurun.
#   gpurun_out/r02c/gpu_suite.log                       python -m pytest tests -m gpu
#   gpurun_out/r02c/bench_driver_command.json           python bench.py --steps 20 --warmup 5
#   gpurun_out/r02c/bench_under_rocprof.json + kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python bench.py ...
#   gpurun_out/r02c/pmc_wg.csv, pmc_chr20.csv           separate --pmc passes on the search launches (tools/search_only.py)
#   gpurun_out/r02c/op_counts_wg.txt                    counting build (make count): lane operations by type
#   gpurun_out/r02c/pmc_inflate.csv                     the inflate kernel (tools/inflate_probe.py 8192 1 bam)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/gpu_suite.log 2>&1
grep -E "passed|failed" $O/gpu_suite.log | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
tail -c 300 $O/bench_driver_command.json
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
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
i=0
for c in "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcwg_$i -- python $R/tools/search_only.py wg 1048576 3 > $O/search_only_$i.log 2>&1
done
summarize "$O/pmcwg_" "sfs_" "$O/pmc_wg.csv"
rm -rf $O/pmcwg_[0-9]*
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcchr20_1 -- python $R/tools/search_only.py chr20 128888 5 > $O/search_only_chr20.log 2>&1
summarize "$O/pmcchr20_" "sfs_" "$O/pmc_chr20.csv"
rm -rf $O/pmcchr20_[0-9]*
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "bgzf_inflate" --output-format csv -d $O/pmcinf_$i -- python $R/tools/inflate_probe.py 8192 1 bam > $O/inflate_probe_$i.log 2>&1
done
summarize "$O/pmcinf_" "bgzf_inflate" "$O/pmc_inflate.csv"
rm -rf $O/pmcinf_[0-9]*
grep "no   copy" $O/inflate_probe_1.log | tail -1
SVDSS_DEBUG=1 SVDSS_LIB=$R/svdss_amd/libsvdss_hip_count.so timeout 600 python $R/tools/search_only.py wg 1048576 1 2>&1 | grep "^\[svdss\]\|^wg" > $O/op_counts_wg.txt
cat $O/op_counts_wg.txt | cut -c1-250
head -14 $O/kernel_stats.csv | cut -c1-170
tail -2 $O/search_only_1.log
