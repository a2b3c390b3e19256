cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_bamdecode_gpu.py -x -q -m gpu 2>&1 | tail -2
for kb in 0 12 16 24 32 48; do
  SBGPU_BAM_STAGE_KB=$kb timeout 300 python tools/bench_bamdecode.py 4e6 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage_kb', $kb, 'ms', round(d['ms_per_call'],3), 'GB/s', round(d['roofline']['achieved']))"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x1/bamprof -o b -- python3 tools/bench_bamdecode.py 4e6 --no-cpu-baseline > /dev/null 2>&1; python3 tools/kernel_stats.py gpurun_out/x1/bamprof 2>/dev/null | head -3
