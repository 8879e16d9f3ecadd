#!/bin/bash
# round 6: `search` at 30x (rank blocks alone) by segments per read (SVDSS_SEGMENTS; 0 = the heuristic)
set -u
TAG=${TAG:-r06av}; OUT=gpurun_out/$TAG; W=/dev/shm/svdss_sg30
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
for rnd in 1 2; do
for g in 0 1 2 4 16; do
  sleep 2
  t0=$(date +%s%N)
  env $( [ $g -gt 0 ] && echo SVDSS_SEGMENTS=$g || echo SVDSS_X=1 ) SVDSS_DEBUG=1 $EXE search --index $W/ref.fmd --bam $W/smoothed.bam --verbose > $W/sfs_$g.txt 2> $OUT/search_$g.log
  t1=$(date +%s%N)
  echo "segments $g: $(( (t1 - t0) / 1000000 )) ms wall | $(grep -o "searched in [0-9]* launch(es), [0-9.]* s ([^)]*), done at +[0-9.]* s" $OUT/search_$g.log) | md5 $(md5sum < $W/sfs_$g.txt | cut -c1-10)" >> $OUT/walls.txt
done
done
rm -rf $W
cat $OUT/walls.txt
