#!/bin/bash
# round 5: instruction counts of the search kernel (one launch = 1,048,576 reads): how much of the step's VALU time is its?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/p1 -- python $R/tools/search_only.py wg 1048576 2 > $O/log1.txt 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/p2 -- python $R/tools/search_only.py wg 1048576 2 > $O/log2.txt 2>&1
python - <<PY
import csv, glob
acc, n = {}, {}
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][-50:], row["Counter_Name"])
        acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
for (kern, ctr), v in sorted(acc.items()):
    print("%s,%s,%d,%.4g" % (kern, ctr, n[(kern, ctr)], v / n[(kern, ctr)]))
PY
rm -rf $O/p1 $O/p2
