#!/usr/bin/env python3
"""How far can the REAL cv2.matchTemplate (wav.py:185) sit from the exactly rounded oracle?  cv2 is absent from this image
and from the GPU boxes, so this runs `oracle.match_template_cv2_model` -- crossCorr's blocking and crossCorr's
arithmetic precision (float64 DFT for float32 streams, FLOAT32 DFT for uint8 streams) through SciPy's FFT -- against the
oracle on searches of BASELINE configs[0] and configs[1] size in both sample types, streams loaded through the oracle's
restatement of WavStream.__init__.  Test infrastructure: reads oracle/, never the product.

usage: cv2_noise_model.py [--searches 24] [--out profiles/r04/cv2_noise_model.json]"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from sushi_amd import synth  # noqa: E402

RTOL, ATOL = 1e-4, 2.5e-7          # the gate of the parity tests: BASELINE.json's 1e-4 relative + one float32 quantum of corr


def measure(name, seconds, window, n_search, sample_type, method, seed):
    rate = 12000
    offset = 1.5
    dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
    src_pcm = synth.make_src_pcm(dst_pcm, int(offset * rate), seed=seed + 1)
    with tempfile.TemporaryDirectory() as d:
        synth.write_wav(os.path.join(d, "dst.wav"), dst_pcm, rate)
        synth.write_wav(os.path.join(d, "src.wav"), src_pcm, rate)
        dst = O.load_wav_stream(os.path.join(d, "dst.wav"), rate, sample_type)
        src = O.load_wav_stream(os.path.join(d, "src.wav"), rate, sample_type)
    events = synth.make_events(n_search, seconds, window + offset, seed=seed + 2)
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, offset, window, seed=seed + 3)
    pick = np.argmin if method == O.SQDIFF_NORMED else np.argmax
    out = {"config": name, "sample_type": sample_type, "method": method, "searches": n_search, "window_s": window,
           "max_idx_diff": 0, "max_gate_ratio": 0.0, "max_rel_score_diff": 0.0, "max_abs_score_diff": 0.0,
           "max_row_abs_diff": 0.0, "median_best_score": None}
    best = []
    for pat, c, w in zip(pats, centres, wins):
        _st, lo, hi = dst.search_bounds(pat.shape[1], c, w)
        img = dst.data[0, lo:hi]
        exact = O.match_template_fft(img, pat[0], method=method)[0]
        model = O.match_template_cv2_model(img, pat[0], method=method)[0]
        ie, im = int(pick(exact)), int(pick(model))
        se, sm = float(exact[ie]), float(model[im])
        # what is held to the gate is the quantity the method minimises: the SQDIFF value, or 1 - the CCOEFF value
        ref = se if method == O.SQDIFF_NORMED else 1.0 - se
        ds = abs(se - sm)
        out["max_idx_diff"] = max(out["max_idx_diff"], abs(ie - im))
        out["max_gate_ratio"] = max(out["max_gate_ratio"], ds / (RTOL * abs(se) + ATOL))
        out["max_rel_score_diff"] = max(out["max_rel_score_diff"], ds / max(abs(ref), 1e-30))
        out["max_abs_score_diff"] = max(out["max_abs_score_diff"], ds)
        out["max_row_abs_diff"] = max(out["max_row_abs_diff"], float(np.abs(exact.astype(np.float64) - model).max()))
        best.append(ref)
    out["median_best_score"] = float(np.median(best))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--searches", type=int, default=24)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "cv2_noise_model.json"))
    args = ap.parse_args()
    O.build()
    rows = []
    for name, seconds, window in (("configs[0] (5 min, CLI small window 1.5 s)", 300.0, 1.5),
                                  ("configs[0] (5 min, max window 30 s)", 300.0, 30.0),
                                  ("configs[1] (45 min, +-60 s)", 2700.0, 60.0)):
        for sample_type in ("float32", "uint8"):
            for method in (O.SQDIFF_NORMED, O.CCOEFF_NORMED):
                r = measure(name, seconds, window, args.searches, sample_type, method, seed=20260924)
                rows.append(r)
                print(json.dumps(r))
    with open(args.out, "w") as f:
        json.dump({"gate": {"rtol": RTOL, "atol": ATOL}, "what": __doc__.split("\n\n")[0], "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
