#!/bin/bash
# round 4: is the GPU busy while `SVDSS search --bam` streams?  kernel trace -> union of the kernels' intervals, per-kernel sums
cd /root/repo; export PYTHONPATH=/root/repo
W=/tmp/svdss_r04_e2e
R04_ONLY_BUILD=1 python tools/r04_e2e.py 1032000 $W > gpurun_out/trace_build.txt 2>&1
cd /tmp && export TMPDIR=/tmp
SVDSS_INFLATE_PER_CU=${PER_CU:-0} SVDSS_CLEAN_EXIT=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_e2e -- /root/repo/svdss_amd/SVDSS search --index $W/chr.fmd --bam $W/reads.bam --noputative --verbose > /dev/null 2> /tmp/prof_e2e.err
grep "records read\|device path\|device batches" /tmp/prof_e2e.err | cut -c1-600 > /root/repo/gpurun_out/trace_run.txt
python3 - <<'PY' >> /root/repo/gpurun_out/trace_run.txt
import csv, glob
kt = glob.glob("/tmp/prof_e2e/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# streaming part: from the first inflate kernel to the last kernel
t0 = min(s for s, e, n in iv if "inflate" in n); t1 = max(e for s, e, n in iv)
busy = 0; cur_s, cur_e = None, None
for s, e, n in iv:
    if e < t0: continue
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("streaming window %.3f s, some kernel running %.3f s (%.0f %%)" % ((t1 - t0) / 1e9, busy / 1e9, 100.0 * busy / (t1 - t0)))
# time with an inflate kernel running
def union(pred):
    b = 0; cs = ce = None
    for s, e, n in iv:
        if e < t0 or not pred(n): continue
        if ce is None or s > ce:
            if ce is not None: b += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    return (b + (ce - cs if ce else 0)) / 1e9
print("an inflate kernel running %.3f s; a search kernel %.3f s; crc %.3f s; neither inflate nor search %.3f s" % (
    union(lambda n: "inflate" in n), union(lambda n: "sfs_search2" in n), union(lambda n: "crc32" in n), union(lambda n: "inflate" not in n and "sfs_search2" not in n)))
mc = glob.glob("/tmp/prof_e2e/**/*memory_copy_trace.csv", recursive=True)
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    tot = {}
    for r in rows:
        k = r.get("Direction", "?"); tot[k] = tot.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("memory copies, seconds summed by direction:", {k: round(v / 1e9, 3) for k, v in tot.items()}, len(rows), "copies")
PY
cat /root/repo/gpurun_out/trace_run.txt
