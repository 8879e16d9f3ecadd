#!/bin/bash
# round 6: does `smooth`'s output file get slower on a fuller /tmp, or right behind other big writes?
set -u
TAG=${TAG:-r06w}; OUT=gpurun_out/$TAG; W=/tmp/svdss_sw
cd "$(dirname "$0")/.."; mkdir -p $OUT $W
EXE=$PWD/svdss_amd/SVDSS
python - <<PY > $OUT/gen.json 2>/dev/null
import json, sys
sys.path.insert(0, ".")
from tools import e2e_call_wg as W
print(json.dumps(W.write_dataset_cxx("$W", ${READS:-6176540}, ${SVS:-20000})[5]))
PY
tm() { local what=$1; shift; local t0=$(date +%s%N); "$@"; local t1=$(date +%s%N); echo "$what: $(( (t1 - t0) / 1000000 )) ms wall" >> "$OUT/walls.txt"; }
sm() { tm "smooth to a file [$1]" env SVDSS_DEBUG=1 $EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16 > $W/sm.bam 2> $OUT/smooth.log; grep "device path" $OUT/smooth.log | sed 's/.*deflate + down/deflate + down/' >> $OUT/walls.txt; df -h /tmp | tail -1 >> $OUT/walls.txt; grep -i "dirty\|writeback" /proc/meminfo | tr '\n' ' ' >> $OUT/walls.txt; echo >> $OUT/walls.txt; }
sm "right behind the generator"
sleep 3; sm "again"
head -c 14000000000 /dev/zero > $W/junk1; sm "right behind 14 GB more in /tmp"
sleep 5; sm "5 s later"
rm -f $W/junk1; sync; sm "junk removed, after sync"
rm -rf $W
cat $OUT/walls.txt
