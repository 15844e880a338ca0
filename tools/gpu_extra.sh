set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_a.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/bench_a.json'));print(d['value'],d['roofline']['stage_ms'],d['roofline']['fft_pairs'],d['roofline']['fft_segments'])"
done
