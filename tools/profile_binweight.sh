#!/bin/bash
# GPU box: PMC passes for the bin-weight kernel (A4).  Usage: bash tools/profile_binweight.sh r01
set -u
R=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$R/binweight
mkdir -p $OUT
python tools/bench_binweight.py > $OUT/bench_binweight.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bw -- python3 $REPO/tools/bench_binweight.py 1000000 > $OUT/stats.log 2>&1
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$tag -o bw -- python3 $REPO/tools/bench_binweight.py 1000000 > $OUT/pmc_$tag.log 2>&1
done
cd $REPO
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
pmc = {}
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'binweight' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
stats = None
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'binweight' in r['Name']:
            stats = r
bench = json.loads(open(out + '/bench_binweight.json').read().strip().splitlines()[-1])
json.dump({"bench_2M_pairs": bench, "rocprof_kernel_stats_1M_pairs": stats, "pmc_1M_pairs": pmc}, open(out + '/summary.json', 'w'), indent=1)
print(open(out + '/summary.json').read())
PY
