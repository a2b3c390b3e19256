#!/bin/bash
# GPU box: the C3 step under the experiments build's tile / schedule switches (usage: bash tools/probe_wave_tiles.sh)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export SBGPU_LIB=$REPO/strawberry_amd/lib/libsbgpu_exp.so
run() { python3 $REPO/bench.py --no-chain --no-front --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k[:28]: round(v,3) for k,v in d['roofline']['all_kernels_ms'].items()})"; }
run default
SBGPU_WAVE_RMULT=1 run rmult1
SBGPU_WAVE_RMULT=4 run rmult4
SBGPU_LIGHT_BLOCK=1 run light_block
SBGPU_MAX_WAVES=4096 run maxwaves4096
SBGPU_PHASES=64 run phases64
SBGPU_STREAM_PRIORITY=0 run noprio
