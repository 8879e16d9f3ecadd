#!/bin/bash
# Round 4: HBM request counters and operation counts of the search kernel as the tree has it now (profiles/traffic.json is
# keyed by a hash of the kernel's sources), kernel statistics of the pipelined bench, call-side DP counters with the POA
# kernel built for five wavefronts per SIMD.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04k
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
# lane operations by type (counting build) -> useful bytes of this kernel version
SVDSS_LIB=$R/svdss_amd/libsvdss_hip_count.so SVDSS_DEBUG=1 timeout 600 python $R/tools/search_only.py wg 1048576 1 2>&1 | grep "lane ops\|wave-iterations with\|search kernel" > $O/op_counts_wg.txt
cat $O/op_counts_wg.txt
cd $R && python -c "
import bench; print('kernel hash', bench.search_kernel_hash())" | tee $O/kernel_hash.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats_pipelined_bench.csv
rm -rf $O/stats
head -8 $O/kernel_stats_pipelined_bench.csv | cut -c1-150
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
# the inflate kernel on the two kinds of BAM streams (zlib level 1 / literals-only Huffman as csrc/deflate.hip writes)
for a in "1 bam" "1 bam huff" "1 binned" "1 binned huff" "1 skew" "1 skew huff"; do python $R/tools/inflate_probe.py 16384 $a 2>&1 | grep -E "blocks, level|no   copy" | tail -2; done > $O/inflate_probe.txt
cat $O/inflate_probe.txt
