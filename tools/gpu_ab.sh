set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sh in 512 256 512 256; do
SUSHI_HIP_IFFT_SHAPE=$sh timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_a.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/bench_a.json'));print('shape $sh', d['value'],d['roofline']['stage_ms'])"
done
SUSHI_HIP_IFFT_SHAPE=256 timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
