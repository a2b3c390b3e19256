#!/bin/bash
# tools/sweep_phases.sh -- C3 step time under different phase limits / lane weights (diagnostic)
cd "$(dirname "$0")/.."
run() {
  SBGPU_PHASES="$1" SBGPU_PHASE_LAMBDA="$2" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ${3:-c3} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('phases %-18s lambda %-14s ms/step %.3f  kinds %s  phases_ms %s' % ('$1','$2',d['ms_per_step'],' '.join('%.3f'%v for v in d['roofline']['all_kernels_ms'].values()),' '.join('%.3f'%v for v in d.get('wave_phase_ms',[]))))"
}
if [ $# -gt 0 ]; then run "$@"; exit; fi
run 0 ""
run 32 t
run 32,128 t,t
run 32,128,512 t,t,t
run 16,64,256 t,t,t
run 32,128,384 t,t,0.25
run 32,128,512 t,t,0.25
run 32,128,256 t,t,0.25
run 64,256 t,0.25
run 32,256 t,0.25
run 32,128,256,512 t,t,t,0.25
run 16,48,128,320 t,t,t,0.25
