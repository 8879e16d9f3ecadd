#!/bin/bash
# round 6: `SVDSS search` with the front end beside the index restore -- where the restore's seconds go in either order.
# Generates the 1.03 M-read chain data set, smooths it, then runs search several ways with SVDSS_INDEX_VERBOSE.
set -u
TAG=${TAG:-r06c}
OUT=gpurun_out/$TAG
W=${W:-/tmp/svdss_early_diag}
cd "$(dirname "$0")/.."
mkdir -p "$OUT" "$W"
EXE=$PWD/svdss_amd/SVDSS
python - <<PY > "$OUT/gen.json" 2> "$OUT/gen.err"
import json, sys
sys.path.insert(0, ".")
from tools import e2e_call_wg as W
print(json.dumps(W.write_dataset_cxx("$W", ${READS:-1030000}, ${SVS:-3400})[5]))
PY
$EXE index -d $W/ref.fa -o $W/ref.fmd > /dev/null 2>&1
$EXE smooth --reference $W/ref.fa --bam $W/reads.bam --threads 16 > $W/smoothed.bam 2> /dev/null
run() { local name=$1; shift; sleep 5; local t0=$(date +%s%N); env SVDSS_INDEX_VERBOSE=1 "$@" $EXE search --index $W/ref.fmd --bam $BAM --verbose $EXTRA > $W/out_$name.txt 2> "$OUT/search_$name.log"; local t1=$(date +%s%N); echo "$name: $(( (t1 - t0) / 1000000 )) ms wall" >> "$OUT/walls.txt"; }
BAM=$W/smoothed.bam; EXTRA=""
run first SVDSS_SEARCH_EARLY=0
run early SVDSS_X=1
run early_nolimit SVDSS_NO_KMER_LIMIT=1
run first_k13 SVDSS_SEARCH_EARLY=0 SVDSS_KMER=13
run early_2feeders SVDSS_SEARCH_FEEDERS=2
run early_again SVDSS_X=1
cmp $W/out_first.txt $W/out_early.txt && echo "early == index-first (smoothed, putative)" >> "$OUT/walls.txt"
cmp $W/out_first.txt $W/out_first_k13.txt && echo "K=13 == K=16" >> "$OUT/walls.txt"
# every read searched (the e2e_wg leg's shape): the original BAM, --noputative, text to a file for the comparison
BAM=$W/reads.bam; EXTRA="--noputative"
run np_first SVDSS_SEARCH_EARLY=0
run np_early SVDSS_X=1
run np_early_again SVDSS_X=1
cmp $W/out_np_first.txt $W/out_np_early.txt && echo "early == index-first (every read searched)" >> "$OUT/walls.txt"
rm -rf "$W"
