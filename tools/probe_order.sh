#!/bin/bash
# GPU box: A/B of plan / scheduling knobs on C3 or C2 (ms per step over repeated runs, per-kind kernel times).
#   bash tools/probe_order.sh [ENV=VALUE ...]     each argument is one configuration to compare with the default
#   W=c2 bash tools/probe_order.sh ...            on C2
run() { echo "== $*"; for i in 1 2 3 4; do env "$@" timeout 300 python bench.py --workload ${W:-c3} --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %.3f ms  %s' % (d['ms_per_step'], {k.split('(')[0][-5:]: round(v, 3) for k, v in d['roofline']['all_kernels_ms'].items()}))"; done; }
run SBGPU_X=default
for cfg in "$@"; do run $cfg; done
