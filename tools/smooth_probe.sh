#!/bin/bash
# Developer probe: `SVDSS smooth` on the BAM of tools/e2e_search.py (43,000 x 15 kb reads), GPU deflate vs host deflate,
# stage times from SVDSS_DEBUG.   $1 = output directory under gpurun_out/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-smooth}
mkdir -p $O
W=/tmp/svdss_e2e
[ -f $W/reads.bam ] || PYTHONPATH=$R timeout 600 python $R/tools/e2e_search.py 64444167 43000 15000 $W > $O/e2e_search_43k.json 2>&1
for mode in gpu host; do
  if [ $mode = host ]; then export SVDSS_GPU_DEFLATE=0; else unset SVDSS_GPU_DEFLATE; fi
  for rep in 1 2 3; do
    s=$(date +%s.%N)
    SVDSS_DEBUG=1 $R/svdss_amd/SVDSS smooth --reference $W/ref.fa --bam $W/reads.bam --threads 32 > $W/smoothed_$mode.bam 2> $O/smooth_$mode.err
    e=$(date +%s.%N)
    echo "smooth deflate=$mode run $rep: $(python3 -c "print(round($e - $s, 2))") s, $(stat -c %s $W/smoothed_$mode.bam) bytes; $(grep -h 'done at' $O/smooth_$mode.err | sed 's/.*done at/done at/')" | tee -a $O/smooth_times.txt
  done
done
unset SVDSS_GPU_DEFLATE
python3 - <<PY | tee -a $O/smooth_times.txt
import gzip
a = gzip.open("$W/smoothed_gpu.bam").read(); b = gzip.open("$W/smoothed_host.bam").read()
print("inflated streams identical:", a == b, len(a))
PY
