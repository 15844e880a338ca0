set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_shapes or prepare_stream or spectra" > gpurun_out/pytest_prop.log 2>&1; tail -30 gpurun_out/pytest_prop.log | grep -v Warning
