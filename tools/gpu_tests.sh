# GPU tests only (stop at the first failure), then smoke.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; tail -40 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
