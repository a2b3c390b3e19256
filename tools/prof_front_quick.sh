#!/bin/bash
# GPU box: kernel stats of a few c3-front steps -> gpurun_out/prof_front (usage: bash tools/prof_front_quick.sh [frags])
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_front
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SB_FRONT_FRAGS=${1:-2e8} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fr -- python3 $REPO/bench.py --workload c3-front --no-cpu-baseline --steps 3 --warmup 1 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-400
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if "sb::" in r["Name"] or "rocprim" in r["Name"]:
        print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
