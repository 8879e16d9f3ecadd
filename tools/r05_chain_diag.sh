#!/bin/bash
# round 5 (second session): where the stages of the chain (index -> smooth -> search -> call at GRCh38 lengths, 1.03 M reads
# with 0.5 % errors) spend their time.  Generates the dataset once (tools/e2e_call_wg.py chain), then runs the stages again
# with their debug logs and under rocprofv3 --kernel-trace --stats.  Output: gpurun_out/r05t/.
set -u
OUT=${OUT:-gpurun_out/r05t}
W=${W:-/tmp/svdss_chain_diag}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
ROOT=$PWD
EXE=$ROOT/svdss_amd/SVDSS
tm() { local what=$1; shift; local t0=$(date +%s%N); "$@"; local t1=$(date +%s%N); echo "$what: $(( (t1 - t0) / 1000000 )) ms wall" >> "$ROOT/$OUT/walls.txt"; }
python tools/e2e_call_wg.py chain ${READS:-1030000} ${SVS:-3400} "$W" ${SCALE:-1.0} > "$OUT/chain.json" 2> "$OUT/chain.err"
ls -l "$W" > "$OUT/files.txt"
FA=$W/ref.fa; BAM=$W/reads.bam; SM=$W/smoothed.bam; FMD=$W/ref.fmd; SFS=$W/specifics.txt
[ -f "$FA" ] || FA=$(ls $W/*.fa | head -1)
[ -f "$BAM" ] || BAM=$(ls $W/*.bam | grep -v smoothed | head -1)
echo "FA=$FA BAM=$BAM" >> "$OUT/files.txt"
for k in 1 2; do
  tm "smooth to a file" env SVDSS_DEBUG=1 $EXE smooth --reference $FA --bam $BAM --threads 16 > $W/sm2.bam 2> "$OUT/smooth_file_$k.log"
  tm "smooth to /dev/null" env SVDSS_DEBUG=1 $EXE smooth --reference $FA --bam $BAM --threads 16 2> "$OUT/smooth_null_$k.log" > /dev/null
done
cmp $W/sm2.bam $SM && echo "smoothed BAM identical to the chain's" >> "$OUT/files.txt"
for k in 1 2; do
  tm "search" env SVDSS_INDEX_VERBOSE=1 $EXE search --index $FMD --bam $SM --verbose > $W/sfs2.txt 2> "$OUT/search_$k.log"
done
cmp $W/sfs2.txt $SFS && echo "SFS identical to the chain's" >> "$OUT/files.txt"
tm "call (smoothed BAM, no index)" $EXE call --reference $FA --bam $SM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > $W/calls.vcf 2> "$OUT/call_1.log"
for cfg in "SVDSS_X=1" "SVDSS_CALL_PASS2=bai" "SVDSS_CALL_PASS2=bai SVDSS_CALL_PASS2_THREADS=1" "SVDSS_CALL_PASS2=device"; do
  tm "call (original BAM + BAI) $cfg" env $cfg $EXE call --reference $FA --bam $BAM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > $W/calls_o.vcf 2> "$OUT/call_orig.log"
  echo "== $cfg: $(cmp $W/calls_o.vcf $W/calls.vcf && echo "VCF identical to the smoothed BAM's")" >> $OUT/call_orig_all.txt
  grep "pass 1\|pass 2" "$OUT/call_orig.log" >> $OUT/call_orig_all.txt
done
# kernel traces (the binaries end with _exit unless SVDSS_CLEAN_EXIT is set: rocprofv3 writes its files at exit)
export TMPDIR=/tmp
for st in search smooth call; do
  case $st in
    search) CMD="$EXE search --index $FMD --bam $SM --verbose" ;;
    smooth) CMD="$EXE smooth --reference $FA --bam $BAM --threads 16" ;;
    call) CMD="$EXE call --reference $FA --bam $SM --sfs $SFS --threads 16 --min-sv-length 50 --verbose" ;;
  esac
  ( cd /tmp && SVDSS_DEBUG=1 SVDSS_CLEAN_EXIT=1 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_$st -- $CMD > /dev/null 2> $ROOT/$OUT/${st}_prof.log )
  f=$(find /tmp/prof_$st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $ROOT/$OUT/${st}_kernel_stats.csv
  python3 tools/trace_summary.py /tmp/prof_$st $([ $st = call ] || echo inflate) > $ROOT/$OUT/${st}_trace_summary.txt 2>&1
done
du -sh "$OUT" >> "$OUT/files.txt"
