set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_fft.json 2> gpurun_out/bench_fft.err; tail -3 gpurun_out/bench_fft.err; cat gpurun_out/bench_fft.json
timeout 600 python tools/delta_sweep.py --ws-mb 256 1024 4096 > gpurun_out/delta_sweep.log 2>&1; tail -25 gpurun_out/delta_sweep.log
