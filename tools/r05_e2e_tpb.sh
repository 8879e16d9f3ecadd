#!/bin/bash
# round 5: `SVDSS search --bam` streaming with the search kernel as one-wavefront workgroups (LD_PRELOAD of the variant library)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/e2e_tpb
{
timeout 900 python $R/tools/e2e_search.py 64444167 1032000 15000 /tmp/e2e_tpb > /tmp/e2e_tpb/gen.json 2>/tmp/e2e_tpb/gen.err
tail -1 /tmp/e2e_tpb/gen.json
for lib in "" $R/svdss_amd/libsvdss_hip_occ_t64.so; do
  for feeders in 6 8; do
    for rep in 1 2 3; do
      echo -n "lib=${lib##*/} feeders=$feeders: "
      LD_PRELOAD=$lib SVDSS_SEARCH_FEEDERS=$feeders timeout 300 $R/svdss_amd/SVDSS search --index /tmp/e2e_tpb/ref.fmd --bam /tmp/e2e_tpb/reads.bam --noputative --verbose 2>&1 >/dev/null | grep -o "records read.*\|index and k-mer table on the device at +[0-9.]*" | tr '\n' ' '
      echo
    done
  done
done
} > $R/gpurun_out/r05_e2e_tpb.txt 2>&1
cat $R/gpurun_out/r05_e2e_tpb.txt
rm -rf /tmp/e2e_tpb
