set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_load_gpu.py -m gpu -q -k "prepare_stream or spectra or load or random_shapes or known" 2>&1 | tail -2
cat > gpurun_out/prep.py <<'PY'
import numpy as np, torch
from sushi_amd.device import DeviceStream
x = np.random.default_rng(0).random(32_400_000, dtype=np.float32)
for _ in range(3):
    d = DeviceStream(x); torch.cuda.synchronize()
PY
PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_prep -o prep -- python gpurun_out/prep.py > gpurun_out/prep.log 2>&1
cut -d, -f1-4 gpurun_out/prof_prep/prep_kernel_stats.csv | cut -c1-150
