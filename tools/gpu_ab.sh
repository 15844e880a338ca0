set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
python - <<'PY'
import time, numpy as np, torch
from sushi_amd import synth
from sushi_amd.wav import WavStream
import os
pcm = synth.make_dst_pcm(2700, 12000, seed=1)
for mode in ("auto", "host", "auto"):
    os.environ["SUSHI_HIP_LOAD"] = mode
    for st in ("float32", "uint8"):
        torch.cuda.synchronize(); t = time.time()
        w = WavStream.from_samples(pcm, 12000, sample_rate=12000, sample_type=st)
        d = w.device_stream(); torch.cuda.synchronize()
        print(mode, st, "load+prepare s:", round(time.time() - t, 4))
PY
