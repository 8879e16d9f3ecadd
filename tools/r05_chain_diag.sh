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
  tm "search" $EXE search --index $FMD --bam $SM --verbose > $W/sfs2.txt 2> "$OUT/search_$k.log"
done
cmp $W/sfs2.txt $SFS && echo "SFS identical to the chain's" >> "$OUT/files.txt"
tm "call" $EXE call --reference $FA --bam $SM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > $W/calls.vcf 2> "$OUT/call_1.log"
# kernel traces
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_search -o search -- $EXE search --index $FMD --bam $SM --verbose > /dev/null 2> $ROOT/$OUT/search_prof.log )
( cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_smooth -o smooth -- env SVDSS_DEBUG=1 $EXE smooth --reference $FA --bam $BAM --threads 16 > /dev/null 2> $ROOT/$OUT/smooth_prof.log )
( cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_call -o call -- $EXE call --reference $FA --bam $SM --sfs $SFS --threads 16 --min-sv-length 50 --verbose > /dev/null 2> $ROOT/$OUT/call_prof.log )
# keep the summaries, drop the bulky traces beyond what the merge allows
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
du -sh "$OUT" >> "$OUT/files.txt"
