#!/bin/bash
# round 6: the search kernel's counters after sv_decide_flat -- instruction counts and, in separate passes, the
# HBM request counters profiles/traffic.json is keyed on (kernel hash); tools/search_only.py wg / chr20
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r06ah}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/p1 -- python $R/tools/search_only.py wg 1048576 2 > $O/log1.txt 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/p2 -- python $R/tools/search_only.py wg 1048576 2 > $O/log2.txt 2>&1
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/p3 -- python $R/tools/search_only.py wg 1048576 3 > $O/log3.txt 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/p4 -- python $R/tools/search_only.py chr20 128888 5 > $O/log4.txt 2>&1
python - <<PY
import csv, glob
for tag, pat in (("wg_insts", "p[12]"), ("wg_requests", "p3"), ("chr20_requests", "p4")):
    acc, n = {}, {}
    for f in sorted(glob.glob("$O/%s/**/*counter_collection.csv" % pat, recursive=True)):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    with open("$O/pmc_%s.csv" % tag, "w") as fh:
        fh.write("Kernel,Counter,Dispatches,MeanValuePerDispatch\n")
        for (kern, ctr), v in sorted(acc.items()):
            fh.write("%s,%s,%d,%.1f\n" % (kern, ctr, n[(kern, ctr)], v / n[(kern, ctr)]))
    print(open("$O/pmc_%s.csv" % tag).read())
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
grep -h "search kernel" $O/log*.txt
cd $R && python -c "
import bench; print('kernel hash', bench.search_kernel_hash())" | tee $O/kernel_hash.txt
