#!/bin/bash
# round 6: the e2e leg (`SVDSS search --bam`, 1.03 M unsmoothed reads with qualities, chr20-length index) by file loaders / feeders
export PYTHONPATH=.
OUT=gpurun_out/${TAG:-r06az}; mkdir -p $OUT
E2E_REPEAT=6 timeout 1200 python tools/e2e_search.py 64444167 172000 15000 /tmp/e2e 2>&1 | grep "^{" | cut -c1-200 > $OUT/gen.txt
run() {
  for k in 1 2 3; do
    sleep 3
    t0=$(date +%s%N)
    env "$@" SVDSS_SEARCH_EARLY=0 ./svdss_amd/SVDSS search --index /tmp/e2e/ref.fmd --bam /tmp/e2e/reads.bam --noputative --verbose 2> $OUT/log.txt > /dev/null
    t1=$(date +%s%N)
    echo "[$*] $(( (t1 - t0) / 1000000 )) ms wall | $(grep -o "device at +[0-9.]* s" $OUT/log.txt | head -1) | $(grep -o "records read, [0-9]* SFS written at +[0-9.]* s" $OUT/log.txt) | $(grep -o "the batchers waited [0-9.]* s for the file's loaders and [0-9.]* s for the feeding threads" $OUT/log.txt)" >> $OUT/walls.txt
  done
}
run X=1
run SVDSS_BAM_LOADERS=12
run SVDSS_BAM_LOADERS=16
run SVDSS_BAM_LOADERS=12 SVDSS_SEARCH_FEEDERS=8
run SVDSS_SEARCH_FEEDERS=8
run SVDSS_BAM_LOADERS=4
rm -rf /tmp/e2e
cat $OUT/walls.txt
