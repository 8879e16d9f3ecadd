#!/usr/bin/env python3
"""Compact per-kernel summary (the --stats table) from a rocprofv3 rocpd SQLite database."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as fh:
    fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0][:80].replace(",", ";")
        fh.write(f"{short},{calls},{tot * 1e3:.0f},{avg * 1e3:.0f},{pct:.3f}\n")
print(open(out).read()[:1500])
