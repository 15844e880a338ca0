set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1; tail -16 gpurun_out/pytest_gpu.log
