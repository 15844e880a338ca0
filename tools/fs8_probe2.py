#!/usr/bin/env python3
"""Dev: which property of the fs8burst material makes the ranking stage's f32 scores miss their modelled bound?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sushi_amd.device import DeviceStream, SearchBatch  # noqa: E402


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


cases = [("base", 8.0, True, 0.02, 0.4, 0.5), ("period 8.37", 8.37, True, 0.02, 0.4, 0.5), ("period 16", 16.0, True, 0.02, 0.4, 0.5),
         ("unaligned", 8.0, False, 0.02, 0.4, 0.5), ("floor 0.2", 8.0, True, 0.2, 0.4, 0.5), ("floor 0.002", 8.0, True, 0.002, 0.4, 0.5),
         ("amp 0.1", 8.0, True, 0.02, 0.1, 0.5), ("period 64", 64.0, True, 0.02, 0.4, 0.5), ("period 5.3", 5.3, True, 0.02, 0.4, 0.5)]
for name, period, aligned, floor, amp, dc in cases:
    tot = [0, 0, 0.0, 0.0, 0]
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        n = 180000
        dst = make(n, rng, period, aligned, floor, amp, dc).astype(np.float32)
        offs, lens, wst, npos, parts, pos = [], [], [], [], [], 0
        for k, m in enumerate([300, 4096, 9000, 30000, 50000]):
            a = int(rng.integers(0, n - m))
            piece = dst[a:a + m].astype(np.float64)
            if k % 2:
                piece = piece + rng.standard_normal(m) * 0.01
            parts.append(np.clip(piece, 0, None).astype(dst.dtype))
            w0 = int(rng.integers(0, max(1, a)))
            offs.append(pos); lens.append(m); wst.append(w0); npos.append(n - m - w0 + 1)
            pos += m
        src = np.concatenate(parts)
        b = SearchBatch(DeviceStream(dst), DeviceStream(src), offs, lens, wst, npos, path="fft", exclusion="never")
        b.run()
        b.results()
        d = b.diagnostics(per_search=True)
        tot[0] += d["all_positions"]; tot[1] += d["flagged"]; tot[2] = max(tot[2], d["max_bound_ratio"]); tot[3] = max(tot[3], d["max_bound_ratio_noncandidate"])
        tot[4] += sum(1 for f, k in zip(d["flagged_per_search"], range(5)) if f == 2 and k % 2 == 0)
    print("%-12s all_positions %2d (exact-copy searches among them %d) flagged %2d  max ratio cand %.2f  non-cand %.2f" % (name, tot[0], tot[4], tot[1], tot[2], tot[3]), flush=True)
