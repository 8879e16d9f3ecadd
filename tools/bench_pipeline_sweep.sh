#!/bin/bash
# ms_per_step of bench.py under other pipeline shapes (which stage bounds the step): writes one line per shape
out=${1:-gpurun_out/sweep}
mkdir -p $out
for spec in "s1c3:" "s2c3:--search-threads 2" "s1c2:--call-threads 2" "s1c4:--call-threads 4" "s2c4:--search-threads 2 --call-threads 4" "s1c1:--call-threads 1" "nocall:--no-call-dp" "s2nocall:--no-call-dp --search-threads 2"; do
  name=${spec%%:*}; flags=${spec#*:}
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e $flags > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "ms_per_step %.1f" % d["ms_per_step"], "search kernel %.1f" % r["kernel_ms"], "all search kernels %.1f" % r.get("all_search_kernels_ms", 0),
      "idle %.1f" % r["kernel_ms_on_idle_gpu"], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in (d.get("call_dp") or {}).items() if k.endswith("_ms")})
PY
done
