#!/bin/bash
# round 6: `smooth --gpus N` on the one GPU of the box (regions oversubscribed): the inflated stream against --gpus 1, walls
set -u
TAG=${TAG:-r06ae}; OUT=gpurun_out/$TAG; W=/dev/shm/svdss_sg
cd "$(dirname "$0")/.."; mkdir -p $OUT $W
EXE=$PWD/svdss_amd/SVDSS
timeout 1200 python -m pytest tests/test_smooth_gpu.py tests/test_config3_gpu.py tests/test_bam_device_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
python - <<PY > $OUT/gen.json 2>/dev/null
import json, sys
sys.path.insert(0, ".")
from tools import e2e_call_wg as W
print(json.dumps(W.write_dataset_cxx("$W", ${READS:-1029423}, ${SVS:-3300})[5]))
PY
tm() { local what=$1; shift; local t0=$(date +%s%N); "$@"; local t1=$(date +%s%N); echo "$what: $(( (t1 - t0) / 1000000 )) ms wall" >> "$OUT/walls.txt"; }
for g in 1 2 4 1 2; do
  tm "smooth --gpus $g" env SVDSS_DEBUG=1 SVDSS_GPUS_OVERSUBSCRIBE=1 $EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16 --gpus $g > $W/sm$g.bam 2> $OUT/smooth$g.log
  grep "regions on\|device path" $OUT/smooth$g.log | cut -c1-300 >> $OUT/walls.txt
  echo "gpus $g: $(stat -c %s $W/sm$g.bam) bytes, inflated md5 $(gzip -dc $W/sm$g.bam | md5sum | cut -d' ' -f1)" >> $OUT/walls.txt
done
rm -rf $W
cat $OUT/walls.txt
