# developer: SQ counters of the inflate kernel (separate --pmc passes, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R
O=$R/gpurun_out/r02b/pmc_inf
rm -rf $O; mkdir -p $O
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "bgzf_inflate" --output-format csv -d $O/p$i -- python $R/tools/inflate_probe.py ${NB:-1024} 1 bam > $O/log_$i.txt 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k in acc: print(f"  {k}: {acc[k] / 8:.4g} per dispatch")
PY
done
