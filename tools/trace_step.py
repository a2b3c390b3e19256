#!/usr/bin/env python3
"""Print the kernel timeline of one bench step from a rocprofv3 kernel trace (csv).
  python tools/trace_step.py <dir with *kernel_trace.csv> [step index]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tpm_kernel" in r["Kernel_Name"]]
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-64s start %8.1f  end %8.1f  dur %7.1f us  q%s grid %s" % (r["Kernel_Name"][:64], s / 1e3, e / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Grid_Size_X"]))
