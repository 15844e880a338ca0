set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_shapes" > gpurun_out/prop_fail.log 2>&1
grep -v Warning gpurun_out/prop_fail.log | grep -n "Falsifying\|seed=\|L=\|frac=\|u8=\|path=\|scale=\|Error\|assert " | head -40
