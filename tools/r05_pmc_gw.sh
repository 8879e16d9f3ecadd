#!/bin/bash
# round 5: wave-instructions of the POA first stage per group width (one bench step, 5,093 sub-clusters)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for gw in 16 32; do
  SVDSS_POA_QUAD_GW=$gw timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --kernel-include-regex "poa_quad" --output-format csv -d $O/p$gw -- python $R/tools/call_dp_probe.py 3395 2 > $O/log_$gw.txt 2>&1
done
python - <<PY
import csv, glob
for gw in (16, 32):
    acc, n = {}, {}
    for f in sorted(glob.glob("$O/p%d/**/*counter_collection.csv" % gw, recursive=True)):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0][-30:], row["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    tot = 0
    for (kern, ctr), v in sorted(acc.items()):
        per = v / n[(kern, ctr)] * (1 if True else 1)
        print(gw, kern, ctr, n[(kern, ctr)], "%.4g" % per)
PY
rm -rf $O/p16 $O/p32
