"""Summary of a rocprofv3 --kernel-trace (csv) directory: the window from the first to the last kernel, how much of it some
kernel was running, per kernel name the number of launches, the summed and the union time.
   python tools/trace_summary.py DIR [first-kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not kt:
    print("no kernel_trace.csv under", d)
    sys.exit(0)
rows = list(csv.DictReader(open(kt[0])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]) for r in rows)
first = sys.argv[2] if len(sys.argv) > 2 else None
t0 = min(s for s, e, n in iv if first is None or first in n)
t1 = max(e for s, e, n in iv)


def union(pred):
    b = 0
    cs = ce = None
    for s, e, n in iv:
        if e < t0 or not pred(n):
            continue
        s = max(s, t0)
        if ce is None or s > ce:
            if ce is not None:
                b += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return (b + (ce - cs if ce else 0)) / 1e9


print("window %.3f s (from the first %s kernel), some kernel running %.3f s (%.0f %%)" % ((t1 - t0) / 1e9, first or "", union(lambda n: True),
                                                                                   100.0 * union(lambda n: True) / ((t1 - t0) / 1e9)))
tot, cnt = defaultdict(int), defaultdict(int)
for s, e, n in iv:
    if e < t0:
        continue
    tot[n] += e - s
    cnt[n] += 1
print("%-70s %7s %9s %9s %9s" % ("kernel", "calls", "sum s", "union s", "mean ms"))
for n in sorted(tot, key=lambda k: -tot[k])[:25]:
    print("%-70s %7d %9.3f %9.3f %9.3f" % (n, cnt[n], tot[n] / 1e9, union(lambda m: m == n), tot[n] / cnt[n] / 1e6))
# idle gaps
gaps = []
ce = t0
for s, e, n in iv:
    if e < t0:
        continue
    if s > ce:
        gaps.append((s - ce, ce - t0))
    ce = max(ce, e)
gaps.sort(reverse=True)
print("largest gaps with no kernel running (ms at offset ms):", ", ".join("%.1f@%.0f" % (g / 1e6, o / 1e6) for g, o in gaps[:12]))
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    t = defaultdict(int)
    for r in rows:
        t[r.get("Direction", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("memory copies, seconds summed by direction:", {k: round(v / 1e9, 3) for k, v in t.items()}, len(rows), "copies")
