#!/bin/bash
# ms_per_step of bench.py with the search and the call-side DP on disjoint (or partly shared) compute units:
# SVDSS_SEARCH_CUS / SVDSS_CALL_CUS = "first,count" of the 256 mask bits (bit b = CU b / 8 of XCD b % 8)
out=${1:-gpurun_out/cusplit}
shift
mkdir -p $out
for spec in "$@"; do
  s=${spec%%/*}; c=${spec##*/}
  name=$(echo "s${s}_c${c}" | tr ',' '-')
  env ${s:+SVDSS_SEARCH_CUS=$s} ${c:+SVDSS_CALL_CUS=$c} python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $out/$name.json 2> $out/$name.err
  python - $out/$name.json "search=$s call=$c" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[2], "| ms_per_step %.1f" % d["ms_per_step"], "search kernel %.1f" % r["kernel_ms"], "idle %.1f" % r["kernel_ms_on_idle_gpu"])
except Exception as e:
    print(sys.argv[2], "| failed:", e)
PY
done
