#!/bin/bash
# round 4: kernel-time breakdown of `SVDSS search --bam` on the device path (rocprofv3 --kernel-trace --stats)
cd /root/repo; export PYTHONPATH=/root/repo
W=/tmp/svdss_r04_e2e
R04_ONLY_BUILD=1 python tools/r04_e2e.py 344000 $W > gpurun_out/prof_build.txt 2>&1
ls -la $W >> gpurun_out/prof_build.txt
cd /tmp && export TMPDIR=/tmp
SVDSS_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2e -- /root/repo/svdss_amd/SVDSS search --index $W/chr.fmd --bam $W/reads.bam --noputative --verbose > /dev/null 2> /tmp/prof_e2e.err
tail -5 /tmp/prof_e2e.err > /root/repo/gpurun_out/prof_e2e_err.txt
f=$(find /tmp/prof_e2e -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/r04m_e2e_kernel_stats.csv
head -20 /root/repo/gpurun_out/r04m_e2e_kernel_stats.csv | cut -c1-200
