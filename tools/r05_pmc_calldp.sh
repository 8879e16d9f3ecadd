#!/bin/bash
# round 5: SQ instruction counters of the call-side kernels on one bench step's work, new POA stage vs poa_wave.hip (separate --pmc passes)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
i=0
for cfg in "SVDSS_POA_QUAD=1" "SVDSS_POA_QUAD=0"; do
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  env $cfg timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "poa_|align_wave" --output-format csv -d $O/p$i -- python $R/tools/call_dp_probe.py 3395 2 > $O/log_$i.txt 2>&1
done
done
python - <<PY
import csv, glob
acc, n = {}, {}
for i in range(1, 7):
    cfg = "quad" if i <= 3 else "wave"
    for f in sorted(glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True)):
        for row in csv.DictReader(open(f)):
            k = (cfg, row["Kernel_Name"].split("(")[0][-44:], row["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
with open("$O/pmc_calldp.csv", "w") as fh:
    fh.write("Config,Kernel,Counter,Dispatches,MeanValuePerDispatch\n")
    for (cfg, kern, ctr), v in sorted(acc.items()):
        fh.write("%s,%s,%s,%d,%.4g\n" % (cfg, kern, ctr, n[(cfg, kern, ctr)], v / n[(cfg, kern, ctr)]))
print(open("$O/pmc_calldp.csv").read())
PY
rm -rf $O/p[0-9]*
