#!/bin/bash
# Round 3: evidence for the search kernel with the 128-symbol window (read bytes fetched once, a line at a time), run
# through gpurun.   $1 = tag (default r03a)
#   gpurun_out/<tag>/search_only_wg.log          tools/search_only.py wg: kernel time on an idle GPU
#   gpurun_out/<tag>/pmc_wg.csv, pmc_chr20.csv   separate --pmc passes on the search launches
#   gpurun_out/<tag>/op_counts_wg.txt            counting build (make count): lane operations by type
R=$GRAFT_REPO_ROOT
T=${1:-r03a}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 300 python $R/tools/search_only.py wg 1048576 4 > $O/search_only_wg.log 2>&1
tail -1 $O/search_only_wg.log
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
  timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcwg_$i -- python $R/tools/search_only.py wg 1048576 3 > $O/search_only_pmc_$i.log 2>&1
done
summarize "$O/pmcwg_" "sfs_" "$O/pmc_wg.csv"
rm -rf $O/pmcwg_[0-9]*
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcchr20_1 -- python $R/tools/search_only.py chr20 128888 5 > $O/search_only_chr20.log 2>&1
summarize "$O/pmcchr20_" "sfs_" "$O/pmc_chr20.csv"
rm -rf $O/pmcchr20_[0-9]*
tail -1 $O/search_only_chr20.log
SVDSS_DEBUG=1 SVDSS_LIB=$R/svdss_amd/libsvdss_hip_count.so timeout 600 python $R/tools/search_only.py wg 1048576 1 2>&1 | grep "^\[svdss\]\|^wg" > $O/op_counts_wg.txt
cat $O/op_counts_wg.txt | cut -c1-250
