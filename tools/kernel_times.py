"""Developer helper: per-kernel mean duration (ms) from a rocprofv3 kernel_stats CSV."""
import csv
import sys

for row in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in row["Name"]:
        print(f'{row["Name"].split("(")[0][:60]:60s} calls {row["Calls"]:>5s}  mean {float(row["AverageNs"]) / 1e6:9.3f} ms')
