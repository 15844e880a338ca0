set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
python tools/latency.py > gpurun_out/latency.json 2>gpurun_out/latency.err; cat gpurun_out/latency.json; tail -3 gpurun_out/latency.err
