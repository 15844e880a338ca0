#!/usr/bin/env python3
"""GPU: where a one-shot job's set-up goes (bench.py `setup_ms.streams_upload_prefix_sums_spectra`), step by step in a fresh
process: the context's first use, the library's first call on a tiny stream (code object load, first launches), then the two
2-h streams and the destination's block spectra.  usage: setup_probe.py [warm]   (warm: a tiny stream first)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sushi_amd.device import DeviceStream  # noqa: E402

n = 86_640_000
x = np.random.default_rng(0).random(n, dtype=np.float32)
y = np.random.default_rng(1).random(n, dtype=np.float32)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def ms(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 1), r


out = []
out.append(("torch.cuda.synchronize (what bench.py does before its clock)", ms(lambda: None)[0]))
if len(sys.argv) > 1 and sys.argv[1] == "warm":
    t, tiny = ms(lambda: DeviceStream(x[:65536], device=dev))
    out.append(("tiny stream (64 K samples): library's first call", t))
    out.append(("tiny stream searchable()", ms(lambda: tiny.searchable())[0]))
t, d = ms(lambda: DeviceStream(x, device=dev))
out.append(("destination stream (347 MB over PCIe + prefix sums)", t))
t, s = ms(lambda: DeviceStream(y, device=dev))
out.append(("source stream", t))
out.append(("destination's block spectra", ms(lambda: d.searchable())[0]))
print(sys.argv[1:] or ["cold"], out)
