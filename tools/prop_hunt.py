#!/usr/bin/env python3
"""Dev: random (shape, dtype, path, scale) cases of tests/test_gpu_parity.py::test_random_shapes_property,
printing every case whose result differs from the oracle.  usage: prop_hunt.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.test_gpu_parity import _run_batch, _check_f32, _check_u8  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    master = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    O.build()
    bad = 0
    for c in range(n_cases):
        seed = int(master.integers(0, 2 ** 31 - 1))
        L = int(master.integers(1, 30001)) if master.random() < 0.7 else int(master.integers(1, 200))
        if master.random() < 0.3:
            L = int(master.integers(2048, 60000))
        frac = float(master.random()) if master.random() < 0.8 else float(master.choice([0.0, 1.0]))
        u8 = bool(master.integers(0, 2))
        path = [0, 2, "fft"][int(master.integers(0, 3))]
        scale = float(master.choice([1.0, 1e-3, 1e-6, 40.0]))
        M = max(1, min(L, int(round(frac * L))))
        rng = np.random.default_rng(seed)
        if u8:
            dst = rng.integers(0, 256, L + 5, dtype=np.uint8)
            src = rng.integers(0, 256, M + 3, dtype=np.uint8)
        elif path == "fft":
            dst = (rng.random(L + 5) * scale).astype(np.float32)
            src = (rng.random(M + 3) * scale).astype(np.float32)
        else:
            dst = (0.25 + 0.5 * rng.random(L + 5)).astype(np.float32)
            src = (0.25 + 0.5 * rng.random(M + 3)).astype(np.float32)
        if M >= 8 and L - M >= 1:
            p = int(rng.integers(0, L - M + 1))
            src[1:1 + M] = dst[2 + p:2 + p + M]
            src[1 + M // 2] = dst[0]
        idx, score = _run_batch(dst, src, [1], [M], [2], [L - M + 1], path)
        res = O.match_template_direct(dst[2:2 + L], src[1:1 + M])[0]
        try:
            (_check_u8 if u8 else _check_f32)(res, idx[0], score[0])
        except AssertionError as e:
            bad += 1
            o = int(res.argmin())
            print("BAD seed=%d L=%d M=%d u8=%s path=%s scale=%g: got (%d, %.9g) oracle (%d, %.9g) %s"
                  % (seed, L, M, u8, path, scale, idx[0], score[0], o, res[o], str(e)[:80]), flush=True)
    print("cases", n_cases, "bad", bad)


if __name__ == "__main__":
    main()
