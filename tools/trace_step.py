#!/usr/bin/env python3
"""tools/trace_step.py <kernel_trace.csv> -- timeline of the last few bench steps out of a rocprofv3 --kernel-trace run:
start/end of every kernel relative to the step's first kernel (us)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void sb::", "").replace("sb::", "")
    return n[:44]
# a step starts at the first delay_kernel / em kernel after a tpm_kernel
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "tpm_kernel" in r["Kernel_Name"]:
        steps.append(cur); cur = []
for st in steps[-3:]:
    t0 = int(st[0]["Start_Timestamp"])
    print("step of %d kernels, %.1f us from first start to last end" % (len(st), (max(int(r["End_Timestamp"]) for r in st) - t0) / 1e3))
    for r in st:
        print("   %8.1f .. %8.1f  (%7.1f)  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
if len(steps) > 4:
    gaps = [(int(b[0]["Start_Timestamp"]) - max(int(r["End_Timestamp"]) for r in a)) / 1e3 for a, b in zip(steps[-6:-1], steps[-5:])]
    print("gap between steps (last end -> next first start) us:", " ".join("%.1f" % g for g in gaps))
