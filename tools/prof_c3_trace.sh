#!/bin/bash
# GPU box: the kernel TRACE of a few C3 steps (start / end of every EM kernel, per stream) -> a per-step timeline on stdout
# (usage: [SB_PIPELINE=0|1] bash tools/prof_c3_trace.sh)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_c3_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT -o c3 -- python3 $REPO/bench.py --no-front --no-chain --no-cpu-baseline --steps 12 --warmup 3 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-200
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
em = [r for r in rows if "em_" in r["Kernel_Name"] or "abundance" in r["Kernel_Name"] or "tpm" in r["Kernel_Name"] or "delay" in r["Kernel_Name"]]
# the last 4 steps' launches: name, queue, start and end relative to the first of them
names = [r["Kernel_Name"] for r in em]
last = [i for i, n in enumerate(names) if "tpm" in n]
import os
back = int(os.environ.get("SB_TRACE_STEPS", "6"))
if len(last) >= back:
    lo = last[-back] + 1
    t0 = int(em[lo]["Start_Timestamp"])
    for r in em[lo:]:
        print("%-60s q%-3s %9.1f %9.1f  %7.1f us" % (r["Kernel_Name"][:60], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
              (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
