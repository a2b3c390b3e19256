#!/bin/bash
# GPU box: bench line, rocprofv3 kernel stats and PMC passes for the bin-sequence kernel (A8).
# Usage (from the repo root): bash tools/profile_binseq.sh r01
set -u
R=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$R/binseq
mkdir -p $OUT
python tools/bench_binseq.py > $OUT/bench_binseq.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bs -- python3 $REPO/tools/bench_binseq.py --no-cpu-baseline > $OUT/stats.log 2>&1
# one PMC group per pass, never together with a trace domain other than the kernel trace
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$tag -o bs -- python3 $REPO/tools/bench_binseq.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_$tag.log 2>&1
done
cd $REPO
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
pmc = {}
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'binseq' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
stats = None
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'binseq' in r['Name']:
            stats = r
bench = json.loads(open(out + '/bench_binseq.json').read().strip().splitlines()[-1])
json.dump({"bench": bench, "rocprof_kernel_stats": stats, "pmc": pmc}, open(out + '/summary.json', 'w'), indent=1)
print(open(out + '/summary.json').read())
PY
