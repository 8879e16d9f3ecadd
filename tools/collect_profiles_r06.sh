#!/bin/bash
# Round 6: rocprofv3 kernel statistics of (i) the bench's pipelined steps (the driver's command without its CPU and e2e legs) and
# (ii) the three binaries of the chain at the metric's scale (30x).  Output: gpurun_out/$TAG/ (copied to profiles/ by hand).
set -u
TAG=${TAG:-r06p}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
W=${W:-/tmp/svdss_prof30x}
mkdir -p $O $W
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_pipelined_bench.csv
rm -rf /tmp/prof_bench
cd $R
python - <<PY > $O/gen.json 2> $O/gen.err
import json, sys
sys.path.insert(0, ".")
from tools import e2e_call_wg as W
print(json.dumps(W.write_dataset_cxx("$W", ${READS:-6176540}, ${SVS:-20000})[5]))
PY
EXE=$R/svdss_amd/SVDSS
$EXE index -d $W/ref.fa -o $W/ref.fmd > /dev/null 2>&1
$EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16 > $W/smoothed.bam 2> /dev/null
$EXE search --index $W/ref.fmd --bam $W/smoothed.bam > $W/specifics.txt 2> /dev/null
cd /tmp
for st in smooth search call; do
  case $st in
    smooth) CMD="$EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16" ;;
    search) CMD="$EXE search --index $W/ref.fmd --bam $W/smoothed.bam --verbose" ;;
    call) CMD="$EXE call --reference $W/ref.fa --bam $W/reads.bam --sfs $W/specifics.txt --threads 16 --min-sv-length 50 --verbose" ;;
  esac
  sleep 3
  SVDSS_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$st -- $CMD > /dev/null 2> $O/${st}_prof.log
  f=$(find /tmp/prof_$st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${st}_30x_kernel_stats.csv
  rm -rf /tmp/prof_$st
done
rm -rf $W
