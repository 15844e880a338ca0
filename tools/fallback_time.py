#!/usr/bin/env python3
"""Dev: time of a batch of tie-saturated searches (periodic stream: every period is a near-tie) at BASELINE
configs[1] sizes -- the searches the FFT path hands to its fallback kernels."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sushi_amd.device import DeviceStream, SearchBatch  # noqa: E402


def main():
    n = 45 * 60 * 12000
    t = np.arange(n)
    for dtype in (np.float32, np.uint8):
        wave = 0.5 + 0.25 * np.sin(2 * np.pi * t / 400.0)              # exactly periodic: 3600 exact ties per search
        x = wave.astype(np.float32) if dtype == np.float32 else np.round(wave * 255).astype(np.uint8)
        d = DeviceStream(x)
        k = 8
        offs = [1000000 + 3000000 * i for i in range(k)]
        b = SearchBatch(d, d, offs, [36000] * k, [o - 720000 for o in offs], [1440001] * k, path="fft")
        b.run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.run(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        idx, score = b.results()
        print("FALLBACK", np.dtype(dtype).name, "searches", k, "finished by fallback", b.fallback_count(),
              "ms per search %.2f" % (dt * 1e3 / k), "first idx", int(idx[0]), "score", float(score[0]))


if __name__ == "__main__":
    main()
