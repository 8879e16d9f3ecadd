#!/bin/bash
# round 6: what `smooth` spends before its stream starts (FASTA load, chromosomes up, accuracy pass), 300,000 reads
set -u
TAG=${TAG:-r06am}; OUT=gpurun_out/$TAG; W=/dev/shm/svdss_ss
cd "$(dirname "$0")/.."; mkdir -p $OUT $W
EXE=$PWD/svdss_amd/SVDSS
python - <<PY > $OUT/gen.json 2>/dev/null
import json, sys
sys.path.insert(0, ".")
from tools import e2e_call_wg as W
print(json.dumps(W.write_dataset_cxx("$W", ${READS:-300000}, ${SVS:-1000})[5]))
PY
for k in 1 2 3 4 5 6; do
  if [ $k -ge 3 ]; then export SVDSS_REF_UPLOAD_THREADS=8; fi
  if [ $k -ge 5 ]; then export SVDSS_REF_UPLOAD_THREADS=12; fi
  t0=$(date +%s%N)
  SVDSS_DEBUG=1 $EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16 > $W/sm.bam 2> $OUT/smooth$k.log
  t1=$(date +%s%N)
  echo "run $k: $(( (t1 - t0) / 1000000 )) ms wall | $(grep -o "reference read in [0-9.]* s" $OUT/smooth$k.log) | $(grep -o "chromosomes uploaded at +[0-9.]* s" $OUT/smooth$k.log) | $(grep -o "streaming from +[0-9.]* s" $OUT/smooth$k.log) | $(grep -o "done at +[0-9.]* s" $OUT/smooth$k.log) | md5 $(gzip -dc $W/sm.bam | md5sum | cut -c1-12)" >> $OUT/walls.txt
done
rm -rf $W
cat $OUT/walls.txt
