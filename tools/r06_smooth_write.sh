#!/bin/bash
# round 6: is `smooth`'s output file slow because of where the bytes come from?  (page-locked coherent / non-coherent memory, writers)
# NOTE: SVDSS_PIN_NONCOHERENT was a knob of the experiment's build (hipHostMallocNonCoherent in svdss_host_alloc); it made no difference and is gone.
set -u
TAG=${TAG:-r06v}; OUT=gpurun_out/$TAG; W=/tmp/svdss_sw
cd "$(dirname "$0")/.."; mkdir -p $OUT $W
EXE=$PWD/svdss_amd/SVDSS
python - <<PY > $OUT/gen.json 2>/dev/null
import json, sys
sys.path.insert(0, ".")
from tools import e2e_call_wg as W
print(json.dumps(W.write_dataset_cxx("$W", ${READS:-6176540}, ${SVS:-20000})[5]))
PY
tm() { local what=$1; shift; sleep 3; local t0=$(date +%s%N); "$@"; local t1=$(date +%s%N); echo "$what: $(( (t1 - t0) / 1000000 )) ms wall" >> "$OUT/walls.txt"; }
for cfg in "SVDSS_X=1" "SVDSS_PIN_NONCOHERENT=1" "SVDSS_PIN_NONCOHERENT=1 SVDSS_SMOOTH_WRITERS=8" "SVDSS_SMOOTH_WRITERS=2" "SVDSS_X=2"; do
  tm "smooth to a file [$cfg]" env SVDSS_DEBUG=1 $cfg $EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16 > $W/sm.bam 2> $OUT/smooth.log
  grep "device path" $OUT/smooth.log | sed 's/.*feeder seconds/feeder seconds/' | cut -c1-220 >> $OUT/walls.txt
  md5sum < $W/sm.bam >> $OUT/walls.txt
done
rm -rf $W
cat $OUT/walls.txt
