#!/usr/bin/env python3
"""Dev: the fs8burst stress material (tools/bound_hunt.py) through the FFT path -- diagnostics per sample type / magnitude, bound model and form."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bound_hunt  # noqa: E402
from sushi_amd.device import DeviceStream, SearchBatch  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "fs8burst"
rng = np.random.default_rng({"drift": 1, "steps": 2, "spikes": 3, "tones": 4, "staircase": 5, "noise": 6, "fs8burst": 7}[kind])
for u8, mag in ((True, 1.0), (False, 1.0), (False, 300.0), (False, 1e-3)):
    n = 180000
    x = bound_hunt.make(kind, n, rng)
    dst = (x * 255 + 0.5).astype(np.uint8) if u8 else (x * mag).astype(np.float32)
    offs, lens, wst, npos, parts, pos = [], [], [], [], [], 0
    for k, m in enumerate([300, 4096, 9000, 30000, 50000]):
        a = int(rng.integers(0, n - m))
        piece = dst[a:a + m].astype(np.float64)
        if k % 2:
            piece = piece + rng.standard_normal(m) * (3.0 if u8 else 0.01 * mag)
        parts.append(np.clip(piece, 0, 255 if u8 else None).astype(dst.dtype))
        w0 = int(rng.integers(0, max(1, a)))
        offs.append(pos); lens.append(m); wst.append(w0); npos.append(n - m - w0 + 1)
        pos += m
    src = np.concatenate(parts)
    D, S = DeviceStream(dst), DeviceStream(src)
    for excl in ("never", "always", "band", "whole"):
        for model in ("worst_case", "statistical"):
            b = SearchBatch(D, S, offs, lens, wst, npos, path="fft", exclusion=excl)
            b.set_bound_model(model)
            b.run()
            idx, score = b.results()
            d = b.diagnostics(per_search=True)
            print("u8" if u8 else "f32 x%g" % mag, excl, model, "all_pos", d["all_positions"], "flagged", d["flagged"], "ratio %.3f nc %.3f" % (d["max_bound_ratio"], d["max_bound_ratio_noncandidate"]),
                  "slb_viol", d["slb_violations"], "slb_ratio %.3f" % d["max_slb_ratio_excluded"], "pairs", d["pairs_transformed"], "of", b.fft_pairs, "flags", d["flagged_per_search"].tolist(), "idx", idx.tolist(), flush=True)
