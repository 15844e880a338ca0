#!/usr/bin/env python3
"""Dev: stress of the FFT path's ranking bound on material chosen to break its assumptions: slowly drifting DC,
amplitude steps over four orders of magnitude, sparse spikes, pure tones, quantised steps, any overall magnitude; uint8 and float32.
Every search is compared with the oracle (index / tie rule / score tolerance of tests/test_gpu_parity.py) and the run's largest
error-to-bound ratios (candidates and audited non-candidates) are printed per kind of material.
usage: bound_hunt.py [n_streams] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from sushi_amd.device import DeviceStream, SearchBatch  # noqa: E402
from tests.test_gpu_parity import _check_f32, _check_u8  # noqa: E402

KINDS = ("drift", "steps", "spikes", "tones", "staircase", "noise", "fs8burst")


def make(kind, n, rng):
    t = np.arange(n, dtype=np.float64)
    base = np.convolve(rng.standard_normal(n + 15), np.ones(16) / 16.0, mode="valid")
    if kind == "drift":
        x = 0.5 + 0.1 * base + 0.3 * np.sin(2 * np.pi * t / rng.uniform(20000, 200000)) + 0.1 * np.cumsum(rng.standard_normal(n)) / np.sqrt(n)
    elif kind == "steps":
        gain = 10.0 ** rng.integers(-4, 1, n // 20000 + 1).astype(np.float64)
        x = 0.5 + 0.4 * base * np.repeat(gain, 20000)[:n]
    elif kind == "spikes":
        x = 0.5 + 0.01 * base
        k = rng.integers(0, n, n // 3000)
        x[k] += rng.uniform(-0.45, 0.45, k.shape[0])
    elif kind == "tones":
        x = 0.5 + 0.2 * np.sin(2 * np.pi * t / rng.integers(20, 400)) + 0.1 * np.sin(2 * np.pi * t / rng.uniform(7, 90)) + 0.002 * base
    elif kind == "fs8burst":
        # energy concentrated at EXACTLY Fs/8 -- bin N/8 of a block spectrum, the edge of the low band, whose mirror bin
        # 7N/8 sat inside the band until the band was made symmetric -- in bursts that start on block boundaries (4096 samples),
        # over a quiet low-passed floor: what the split of the rest-of-spectrum bound into the two real blocks' parts must survive
        x = 0.5 + 0.02 * base
        for b0 in range(0, n - 4096, 4096 * int(rng.integers(2, 5))):
            ln = int(rng.integers(1, 4)) * 4096
            ph = rng.uniform(0, 2 * np.pi)
            x[b0:b0 + ln] += rng.uniform(0.15, 0.4) * np.sin(2 * np.pi * t[b0:b0 + ln] / 8.0 + ph)
    elif kind == "staircase":
        x = 0.5 + np.round(base * 8) / 32.0
    else:
        x = 0.5 + 0.25 * base
    return np.clip(x, 0.0, 1.0)


def main():
    n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    master = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    O.build()
    bad = 0
    worst = {}
    for c in range(n_streams):
        kind = KINDS[c % len(KINDS)]
        n = int(master.integers(150000, 400000))
        x = make(kind, n, master)
        u8 = bool(master.integers(0, 2))
        mag = float(master.choice([1.0, 1.0, 1e-3, 300.0]))
        if u8:
            dst = (x * 255 + 0.5).astype(np.uint8)
        else:
            dst = (x * mag).astype(np.float32)
        offs, lens, wst, npos = [], [], [], []
        src_parts = []
        pos = 0
        for k in range(12):
            m = int(master.choice([300, 2000, 4096, 9000, 30000, 70000]))
            m = min(m, n // 3)
            a = int(master.integers(0, n - m))
            piece = dst[a:a + m].astype(np.float64)
            if k % 3:                                             # a noisy copy; else an exact one (ties with repeats possible)
                piece = piece + master.standard_normal(m) * (3.0 if u8 else 0.01 * mag)
            piece = np.clip(piece, 0, 255 if u8 else None).astype(dst.dtype)
            src_parts.append(piece)
            w0 = int(master.integers(0, max(1, a)))
            w1 = int(master.integers(a, n - m)) if a < n - m else a
            offs.append(pos); lens.append(m); wst.append(w0); npos.append(w1 - w0 + 1)
            pos += m
        src = np.concatenate(src_parts)
        b = SearchBatch(DeviceStream(dst), DeviceStream(src), offs, lens, wst, npos, path="fft")
        b.run()
        idx, score = b.results()
        d = b.diagnostics()
        w = worst.setdefault(kind, [0.0, 0.0, 0, 0])
        w[0] = max(w[0], d["max_bound_ratio"]); w[1] = max(w[1], d["max_bound_ratio_noncandidate"])
        w[2] += d["flagged"]; w[3] += d["all_positions"]
        for k in range(len(offs)):
            res = O.match_template(dst[wst[k]:wst[k] + npos[k] + lens[k] - 1], src[offs[k]:offs[k] + lens[k]])[0]
            try:
                (_check_u8 if u8 else _check_f32)(res, idx[k], score[k])
            except AssertionError as e:
                bad += 1
                o = int(res.argmin())
                print("BAD stream %d kind=%s u8=%s mag=%g search %d M=%d P=%d: got (%d, %.9g) oracle (%d, %.9g) %s"
                      % (c, kind, u8, mag, k, lens[k], npos[k], idx[k], score[k], o, res[o], str(e)[:80]), flush=True)
    for kind, (r0, r1, fl, ap) in worst.items():
        print("%-10s max error/bound: candidates %.3f, audited non-candidates %.3f; flagged %d, all_positions %d" % (kind, r0, r1, fl, ap))
    print("streams", n_streams, "bad", bad)


if __name__ == "__main__":
    main()
