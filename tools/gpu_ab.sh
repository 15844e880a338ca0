set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
for ws in 96 192 384 16384; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --ws-mb $ws > gpurun_out/bench_$ws.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/bench_$ws.json'));print($ws,d['value'],d['roofline']['stage_ms'])"
done
