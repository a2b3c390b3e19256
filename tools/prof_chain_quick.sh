#!/bin/bash
# GPU box: kernel stats of a few c3-chain steps -> gpurun_out/prof_chain (usage: bash tools/prof_chain_quick.sh)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_chain
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SB_CHAIN_PCIE=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ch -- python3 $REPO/bench.py --workload c3-chain --no-cpu-baseline --steps 5 --warmup 2 > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-300
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:22]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
