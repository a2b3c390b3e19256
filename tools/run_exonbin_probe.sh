# GPU box: the exon-bin parity tests, then the chain's kernel times (extra environment switches: "$@")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x1
timeout 1200 python -m pytest tests/test_exonbin_gpu.py tests/test_chain_scale_gpu.py tests/test_collapse_gpu.py tests/test_reference_driver_gpu.py -x -q -m gpu 2>&1 | tail -4
for v in "SBGPU_X=0" "$@"; do
  env $v timeout 600 python bench.py --workload c3-chain --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/x1/chain.err | tail -1 > gpurun_out/x1/chain.json
  python - "$v" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/x1/chain.json').read())
c = d['chain']; print(sys.argv[1], "ms/step %.3f" % c['ms_per_step'], {k: round(v, 3) for k, v in c['kernel_ms'].items()}, "frac %.3f" % c['roofline']['frac'])
PY
done
