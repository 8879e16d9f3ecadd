#!/bin/bash
# round 6: the 30x chain laid out as bench.py lays it out (input BAM on /dev/shm, everything else on /tmp), once
set -u
TAG=${TAG:-r06bb}; OUT=gpurun_out/$TAG; W=/tmp/svdss_clb; S=/dev/shm/svdss_clb
cd "$(dirname "$0")/.."; mkdir -p $OUT $W $S
ln -sf $S/reads.bam $W/reads.bam; ln -sf $S/reads.bam.bai $W/reads.bam.bai
CHAIN_KEEP_REF=1 python tools/e2e_call_wg.py chain 6176540 20000 "$W" 1.0 > "$OUT/chain_30x.json" 2> "$OUT/chain_30x.err"
python - <<PY
import json
d = json.load(open("$OUT/chain_30x.json"))
print({k: v for k, v in d.items() if k.endswith("_s") or "recovered" in k})
print(d.get("search_log"))
PY
df -h /tmp | tail -1
rm -rf $W $S
