#!/bin/bash
# round 6: `search` at 30x by device batch size and feeding threads (SVDSS_BAM_BATCH_MB, SVDSS_SEARCH_FEEDERS)
set -u
TAG=${TAG:-r06ax}; OUT=gpurun_out/$TAG; W=/dev/shm/svdss_sb30
cd "$(dirname "$0")/.."; mkdir -p $OUT $W
EXE=$PWD/svdss_amd/SVDSS
python - <<PY > $OUT/gen.json 2>/dev/null
import json, sys
sys.path.insert(0, ".")
from tools import e2e_call_wg as W
print(json.dumps(W.write_dataset_cxx("$W", ${READS:-6176540}, ${SVS:-20000})[5]))
PY
$EXE index -d $W/ref.fa -o $W/ref.fmd > /dev/null 2>&1
$EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16 > $W/smoothed.bam 2> /dev/null
rm -f $W/reads.bam
for rnd in 1 2 3; do
for cfg in "SVDSS_X=1" "SVDSS_BAM_BATCH_MB=128 SVDSS_SEARCH_FEEDERS=8" "SVDSS_BAM_BATCH_MB=160 SVDSS_SEARCH_FEEDERS=8" "SVDSS_BAM_BATCH_MB=128 SVDSS_SEARCH_FEEDERS=10" "SVDSS_BAM_BATCH_MB=96 SVDSS_SEARCH_FEEDERS=12"; do
  sleep 2
  t0=$(date +%s%N)
  env $cfg $EXE search --index $W/ref.fmd --bam $W/smoothed.bam --verbose > $W/sfs.txt 2> $OUT/search.log
  t1=$(date +%s%N)
  echo "[$cfg]: $(( (t1 - t0) / 1000000 )) ms wall | $(grep -o "done at +[0-9.]* s" $OUT/search.log | head -1) | $(grep -o "device batches, seconds summed: [^;]*" $OUT/search.log) | md5 $(md5sum < $W/sfs.txt | cut -c1-10)" >> $OUT/walls.txt
done
done
rm -rf $W
cat $OUT/walls.txt
