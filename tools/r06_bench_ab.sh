#!/bin/bash
# round 6: the pipelined bench (search + call side) with several builds of the library (SVDSS_LIB), same box, alternating
set -u
TAG=${TAG:-r06ag}; OUT=gpurun_out/$TAG
cd "$(dirname "$0")/.."; mkdir -p $OUT
for rnd in 1 2; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    SVDSS_LIB=$PWD/$lib python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-e2e > $OUT/$n.$rnd.json 2> $OUT/$n.$rnd.err
    python - <<PY >> $OUT/summary.txt
import json
d = json.loads(open("$OUT/$n.$rnd.json").read().strip().splitlines()[-1])
print("$n round $rnd: value %.0f ms_per_step %.2f search_ms_per_step %s kernel_alone %s" % (d["value"], d["ms_per_step"], d["config"].get("search_ms_per_step"), d["config"].get("search_kernel_ms_on_idle_gpu")))
PY
  done
done
cat $OUT/summary.txt
