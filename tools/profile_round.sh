#!/bin/bash
# GPU box: the round's evidence.  Usage: bash tools/profile_round.sh r01
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/pytest_gpu.txt
(timeout 900 python bench.py 2>/dev/null) > $OUT/bench_c3.json
(timeout 600 python bench.py --workload c2 2>/dev/null) > $OUT/bench_c2.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 > $OUT/stats_c3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --no-cpu-baseline --steps 10 > $OUT/stats_c2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
cat $OUT/pytest_gpu.txt; cat $OUT/bench_c3.json; echo; cat $OUT/bench_c2.json; echo
find $OUT -name "*.csv" | head -30
