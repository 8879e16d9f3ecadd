#!/bin/bash
# round 6: the chain (index -> smooth -> search -> call, run_svdss:136-178) at the metric's own scale: GRCh38 lengths,
# 6,176,540 x 15 kb reads (30x) with 0.5 % errors sub:ins:del = 2:1.5:1.5, 20,000 SVs -- and, first, the 1.03 M-read (4.9x) chain
# of rounds 4-5 on the same generator.  Output: gpurun_out/$TAG/.
set -u
TAG=${TAG:-r06a}
OUT=gpurun_out/$TAG
W=${W:-/tmp/svdss_chain30x}
cd "$(dirname "$0")/.."
mkdir -p "$OUT" "$W"
{ nproc; free -g; df -h /tmp; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > "$OUT/box.txt" 2>&1
if [ -z "${SKIP_SMALL:-}" ]; then
  python tools/e2e_call_wg.py chain ${READS_SMALL:-1030000} ${SVS_SMALL:-3400} "$W" 1.0 > "$OUT/chain_4.9x.json" 2> "$OUT/chain_4.9x.err"
  rm -f $W/reads.bam $W/reads.bam.bai $W/smoothed.bam
fi
CHAIN_KEEP_REF=1 python tools/e2e_call_wg.py chain ${READS:-6176540} ${SVS:-20000} "$W" 1.0 > "$OUT/chain_30x.json" 2> "$OUT/chain_30x.err"
ls -l "$W" > "$OUT/files.txt"; df -h /tmp >> "$OUT/box.txt"; free -g >> "$OUT/box.txt"
if [ -n "${DIAG:-}" ]; then
  EXE=$PWD/svdss_amd/SVDSS
  FA=$W/ref.fa; BAM=$W/reads.bam; SM=$W/smoothed.bam; FMD=$W/ref.fmd; SFS=$W/specifics.txt
  SVDSS_DEBUG=1 $EXE smooth --reference $FA --bam $BAM --threads 16 2> "$OUT/smooth_null.log" > /dev/null
  SVDSS_INDEX_VERBOSE=1 SVDSS_DEBUG=1 $EXE search --index $FMD --bam $SM --verbose > /dev/null 2> "$OUT/search.log"
  SVDSS_DEBUG=1 $EXE call --reference $FA --bam $BAM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > $W/calls2.vcf 2> "$OUT/call.log"
  cmp $W/calls2.vcf $W/calls.vcf && echo "VCF of the second call identical" >> "$OUT/files.txt"
  for cfg in "SVDSS_CALL_STORE=0" "SVDSS_CALL_STORE_INITIAL_MB=2048" "SVDSS_X=1"; do
    sleep 3
    t0=$(date +%s%N)
    env $cfg $EXE call --reference $FA --bam $BAM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > $W/calls4.vcf 2> "$OUT/call_cfg.log"
    t1=$(date +%s%N)
    echo "== $cfg: $(( (t1 - t0) / 1000000 )) ms wall; $(cmp $W/calls4.vcf $W/calls.vcf && echo same VCF)" >> "$OUT/call_cfgs.txt"
    grep "pass 1\|pass 2\|record store" "$OUT/call_cfg.log" | cut -c1-400 >> "$OUT/call_cfgs.txt"
  done
  SVDSS_CALL_PASS2=device SVDSS_CALL_STORE=0 $EXE call --reference $FA --bam $BAM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > $W/calls3.vcf 2> "$OUT/call_pass2_device.log"
  cmp $W/calls3.vcf $W/calls.vcf && echo "VCF with pass 2 on the device identical (the generator's BAI agrees with the file)" >> "$OUT/files.txt"
fi
if [ -n "${DIAG2:-}" ]; then
  EXE=$PWD/svdss_amd/SVDSS
  FA=$W/ref.fa; BAM=$W/reads.bam; SM=$W/smoothed.bam; FMD=$W/ref.fmd; SFS=$W/specifics.txt
  MD5=$(md5sum < $SM); rm -f $SM $W/calls2.vcf $W/calls3.vcf      # (/tmp is 79 GB: one smoothed BAM at a time)
  tm() { local what=$1; shift; sleep 3; local t0=$(date +%s%N); "$@"; local t1=$(date +%s%N); echo "$what: $(( (t1 - t0) / 1000000 )) ms wall" >> "$OUT/walls2.txt"; }
  [ -n "${NO_SMOOTH_CFGS:-}" ] || for cfg in "SVDSS_X=1" "SVDSS_SEARCH_FEEDERS=9" "SVDSS_SEARCH_FEEDERS=12" "SVDSS_BAM_BATCH_MB=128" "SVDSS_BAM_BATCH_MB=192 SVDSS_SEARCH_FEEDERS=8" "SVDSS_SMOOTH_WRITERS=8"; do
    tm "smooth to a file [$cfg]" env SVDSS_DEBUG=1 $cfg $EXE smooth --reference $FA --bam $BAM --threads 16 > $W/sm2.bam 2> "$OUT/smooth_cfg.log"
    grep "device path" "$OUT/smooth_cfg.log" | cut -c1-400 >> "$OUT/walls2.txt"
  done
  [ "$(md5sum < $W/sm2.bam)" = "$MD5" ] && echo "smoothed BAM identical whatever the settings" >> "$OUT/walls2.txt"
  rm -f $W/sm2.bam
  tm "smooth to /dev/null" env SVDSS_DEBUG=1 $EXE smooth --reference $FA --bam $BAM --threads 16 2> "$OUT/smooth_null2.log" > /dev/null
  grep "device path" "$OUT/smooth_null2.log" | cut -c1-400 >> "$OUT/walls2.txt"
  for cfg in ${CALL_CFGS:-"SVDSS_X=1" "SVDSS_CALL_FEEDERS=4" "SVDSS_CALL_FEEDERS=6" "SVDSS_CALL_FEEDERS=6 SVDSS_BAM_BATCH_MB=128" "SVDSS_CALL_FEEDERS=5 SVDSS_BAM_BATCH_MB=192"}; do
    tm "call [$cfg]" env $cfg $EXE call --reference $FA --bam $BAM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > $W/calls5.vcf 2> "$OUT/call_cfg2.log"
    echo "   $(cmp $W/calls5.vcf $W/calls.vcf && echo same VCF)" >> "$OUT/walls2.txt"
    grep "pass 1\|pass 2 \|POA\|poa\]" "$OUT/call_cfg2.log" | cut -c1-330 >> "$OUT/walls2.txt"
  done
  [ -n "${NO_SMOOTH_CFGS:-}" ] && { rm -rf "$W"; exit 0; }
  export TMPDIR=/tmp
  ( cd /tmp && SVDSS_DEBUG=1 SVDSS_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_smooth -- $EXE smooth --reference $FA --bam $BAM --threads 16 > /dev/null 2> $OLDPWD/$OUT/smooth_prof.log )
  f=$(find /tmp/prof_smooth -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/smooth_kernel_stats.csv
fi
rm -rf "$W"
