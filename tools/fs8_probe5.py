#!/usr/bin/env python3
"""Dev (SUSHI_HIP_LIB=.../libsushi_hip_dump.so, tools/experiments/r06_dump_scores.patch): the GPU's own f32 scores of every first-half
position of every pair against exact float64 scores, on the tone-burst material."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rank_error_sim import make  # noqa: E402
from sushi_amd import _native  # noqa: E402
from sushi_amd.device import DeviceStream, SearchBatch  # noqa: E402
import torch  # noqa: E402

L = _native.lib()
N, B = L.sushi_hip_fft_size(), L.sushi_hip_fft_block()
H, STEP = N - B, 6
for period in (8.0, 16.0):
    rng = np.random.default_rng(100)
    n = 180000
    dst = make(n, rng, period, True, 0.002, 0.4, 0.5).astype(np.float32)
    m = 9000
    a0 = int(rng.integers(0, n - m))
    src = dst[a0:a0 + m].copy()
    D, S = DeviceStream(dst), DeviceStream(src)
    b = SearchBatch(D, S, [0], [m], [0], [n - m + 1], path="fft", exclusion="never")
    b.run()
    torch.cuda.synchronize()
    dump = b.workspace_view(_native.WS_Y).view(torch.float32).cpu().numpy().reshape(-1, N)
    d = b.diagnostics()
    print("period", period, "all_positions", d["all_positions"], "ratios", d["max_bound_ratio"], d["max_bound_ratio_noncandidate"])
    d64 = dst.astype(np.float64); T = src.astype(np.float64)
    tU = float(T @ T); tn = math.sqrt(tU)
    s2 = np.concatenate([[0.0], np.cumsum(d64 * d64)])
    P = n - m + 1
    for p in range(dump.shape[0]):
        q0 = STEP * p * B
        half = int(os.environ.get("DUMP_HALF", "0"))
        pos = np.arange(H) + half * H
        valid = q0 + pos < P
        p_ok = pos[valid]
        wU = s2[q0 + p_ok + m] - s2[q0 + p_ok]
        span = np.zeros(2 * H + m); piece = d64[q0:q0 + 2 * H + m]; span[:piece.shape[0]] = piece
        nn = 1 << int(math.ceil(math.log2(span.shape[0] + m)))
        ex = np.fft.irfft(np.fft.rfft(span, nn) * np.conj(np.fft.rfft(T, nn)), nn)[half * H:half * H + H][valid]
        exact = np.clip((tU + wU - 2 * ex) / (tn * np.sqrt(wU)), 0, 1)
        got = dump[p, :H][valid].astype(np.float64)
        err = np.abs(got - exact)
        i = int(err.argmax())
        big = np.nonzero(err > 1e-3)[0]
        print("   pair %d: max |gpu f32 score - exact| %.3e at pos %d (gpu %.6f exact %.6f); positions off by > 1e-3: %d%s" % (
            p, err[i], p_ok[i], got[i], exact[i], big.shape[0], (" e.g. " + str(p_ok[big[:12]].tolist())) if big.shape[0] else ""))
