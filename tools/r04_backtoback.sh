#!/bin/bash
# round 4: consecutive runs of SVDSS search --bam: is the slow second run the teardown of the run before it?
cd /root/repo; export PYTHONPATH=/root/repo
W=/tmp/svdss_r04_e2e
R04_ONLY_BUILD=1 python tools/r04_e2e.py 1032000 $W > /dev/null 2>&1
run() { SVDSS_DEBUG=1 svdss_amd/SVDSS search --index $W/chr.fmd --bam $W/reads.bam --noputative --verbose 2>&1 > /dev/null | grep -o "on the device at +[0-9.]* s\|SFS written at +[0-9.]* s\|batcher waited [0-9.]* s for the file" | tr '\n' ' '; echo; }
echo "## back to back"; for i in 1 2 3 4 5; do run; done
echo "## one second apart"; for i in 1 2 3 4; do sleep 1; run; done
echo "## three seconds apart"; for i in 1 2 3; do sleep 3; run; done
echo "## back to back, SVDSS_CLEAN_EXIT=1"; for i in 1 2 3 4; do SVDSS_CLEAN_EXIT=1 run; done
