#!/bin/bash
# GPU box: A/B of the plan's ordering knobs, the stream priorities and the wave kind's stream on C3 (ms per step).
run() { echo "== $*"; for i in 1 2 3 4; do env "$@" timeout 300 python bench.py --workload ${W:-c3} --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %.3f ms  %s' % (d['ms_per_step'], {k.split('(')[0][-5:]: round(v, 3) for k, v in d['roofline']['all_kernels_ms'].items()}))"; done; }
run SBGPU_STREAM_PRIORITY=0 SBGPU_CLASS_ORDER=none SBGPU_WAVE_ON_MAIN=0
run SBGPU_STREAM_PRIORITY=1 SBGPU_CLASS_ORDER=pred SBGPU_WAVE_ON_MAIN=0
run SBGPU_STREAM_PRIORITY=1 SBGPU_CLASS_ORDER=pred SBGPU_WAVE_ON_MAIN=1
W=c2 run SBGPU_WAVE_ON_MAIN=0
W=c2 run SBGPU_WAVE_ON_MAIN=1
