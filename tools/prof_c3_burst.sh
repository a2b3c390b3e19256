#!/bin/bash
# GPU box: the START of a pipelined burst of C3 steps (1 warmup step, then 14 timed ones): every EM kernel's queue, start, end
# relative to the burst's first kernel (usage: bash tools/prof_c3_burst.sh)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_c3_burst
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT -o c3 -- python3 $REPO/bench.py --no-front --no-chain --no-cpu-baseline --steps 14 --warmup 1 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
em = [r for r in rows if "em_fused" in r["Kernel_Name"] or "tpm" in r["Kernel_Name"]]
first_tpm = [i for i, r in enumerate(em) if "tpm" in r["Kernel_Name"]][0]
t0 = int(em[first_tpm + 1]["Start_Timestamp"])
for r in em[first_tpm + 1:first_tpm + 1 + 14 * 4]:
    n = r["Kernel_Name"]
    tag = "tpm" if "tpm" in n else ("wave" if "<0, 2" in n else "block" if "<4, 2" in n else "tall")
    print("%-6s q%-3s %9.1f %9.1f  %7.1f us" % (tag, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
