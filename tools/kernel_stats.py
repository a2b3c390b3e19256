#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 --kernel-trace --stats --output-format csv result directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-100s calls %6s total_ms %9.3f avg_us %9.1f %5s%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
