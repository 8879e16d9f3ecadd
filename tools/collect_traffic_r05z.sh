#!/bin/bash
# Round 5, second session: sfs_search.hip changed again (a segment that overflows its region stops at once -- segmented launches only), so
# profiles/traffic.json, keyed by a hash of the kernel sources, is re-collected: HBM request counters of the search kernel.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcwg_1 -- python $R/tools/search_only.py wg 1048576 3 > $O/search_only_pmc_wg.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --kernel-include-regex "sfs_search2" --output-format csv -d $O/pmcchr20_1 -- python $R/tools/search_only.py chr20 128888 5 > $O/search_only_pmc_chr20.log 2>&1
python - <<PY
import csv, glob
for tag in ("wg", "chr20"):
    acc, n = {}, {}
    for f in sorted(glob.glob("$O/pmc%s_1/**/*counter_collection.csv" % tag, recursive=True)):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    with open("$O/pmc_%s_requests.csv" % tag, "w") as fh:
        fh.write("Kernel,Counter,Dispatches,MeanValuePerDispatch\n")
        for (kern, ctr), v in sorted(acc.items()):
            fh.write("%s,%s,%d,%.1f\n" % (kern, ctr, n[(kern, ctr)], v / n[(kern, ctr)]))
    print(open("$O/pmc_%s_requests.csv" % tag).read())
PY
rm -rf $O/pmcwg_1 $O/pmcchr20_1
cd $R && python -c "
import bench; print('kernel hash', bench.search_kernel_hash())" | tee $O/kernel_hash.txt
