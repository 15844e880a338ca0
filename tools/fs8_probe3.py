#!/usr/bin/env python3
"""Dev: error floor of the stored block spectra against numpy's float64 transform, for tone bursts of several periods."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sushi_amd import _native  # noqa: E402
from sushi_amd.device import DeviceStream  # noqa: E402


def make(n, rng, period, aligned, floor, amp, dc):
    t = np.arange(n, dtype=np.float64)
    base = np.convolve(rng.standard_normal(n + 15), np.ones(16) / 16.0, mode="valid")
    x = dc + floor * base
    step = 4096 if aligned else 4096 + 777
    for b0 in range(0, n - 4096, step * int(rng.integers(2, 5))):
        ln = int(rng.integers(1, 4)) * (4096 if aligned else 3000)
        ph = rng.uniform(0, 2 * np.pi)
        x[b0:b0 + ln] += rng.uniform(amp / 2, amp) * np.sin(2 * np.pi * t[b0:b0 + ln] / period + ph)
    return np.clip(x, 0.0, 1.0)



L = _native.lib()
N, B = L.sushi_hip_fft_size(), L.sushi_hip_fft_block()
H = N - B
slot = np.array([L.sushi_hip_fft_slot_of_bin(f) for f in range(N)])
for period in (8.0, 8.37, 16.0, 5.3, 64.0, 12.0, 9.0, 7.0):
    rng = np.random.default_rng(100)
    n = 180000
    x = make(n, rng, period, True, 0.002, 0.4, 0.5).astype(np.float32)
    d = DeviceStream(x)
    halves = d.spectra().cpu().numpy().astype(np.float64).reshape(-1, N, 2)
    spec = (halves[..., 0] + 1j * halves[..., 1])[:, slot]
    xc = np.zeros(n + 16 * N)
    xc[:n] = x.astype(np.float64) - np.float64(np.float32(x.astype(np.float64).mean()))
    worst = [0.0, 0.0, 0.0]
    scale = None
    for j in range(spec.shape[0] - 1):
        ref = np.fft.fft(xc[j * B:j * B + N] + 1j * xc[j * B + H:j * B + H + N])
        if scale is None:
            scale = 2.0 ** np.round(np.log2(np.abs(spec[j]).max() / np.abs(ref).max()))
        err = spec[j] / scale - ref
        half_ok = 2.0 ** -11 * np.abs(ref) * 1.5
        excess = np.maximum(np.abs(err) - half_ok, 0.0)
        nz = float(np.sqrt(np.sum(np.abs(ref) ** 2)))
        worst[0] = max(worst[0], float(excess.max() / np.abs(ref).max()))
        worst[1] = max(worst[1], float(np.abs(err[0]) / np.abs(ref).max()))
        worst[2] = max(worst[2], float(np.sqrt(np.sum(excess ** 2)) / nz))
    print("period %-5g  max excess over half rounding / max bin %.2e   |err(bin 0)| / max bin %.2e   2-norm of the excess / |Z| %.2e" % (period, worst[0], worst[1], worst[2]), flush=True)
