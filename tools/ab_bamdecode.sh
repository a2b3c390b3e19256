#!/bin/bash
# GPU box: sbgpu_bam_decode_device, one pass against the two-kernel fallback (SBGPU_BAM_TWO_PASS=1), n records resident in HBM
# usage: bash tools/ab_bamdecode.sh [n_records=1e8]
N=${1:-1e8}
for tp in 0 1; do echo "two_pass=$tp"; SBGPU_BAM_TWO_PASS=$tp timeout 600 python tools/bench_bamdecode.py $N --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330; done
