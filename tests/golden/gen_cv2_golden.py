#!/usr/bin/env python3
"""Writes tests/golden/cv2_match_template.json from the REAL cv2.matchTemplate (the call at reference wav.py:185), on any
machine where `import cv2` works: full result rows for the cases of tests/test_cv2_crosscheck.py::_cases() -- both methods,
both sample types -- as float32 bit patterns, plus the cv2 version.  tests/test_cv2_golden.py then holds the oracle's
restatement to them on every machine (it skips while the fixture does not exist: neither the build image nor the GPU boxes
have OpenCV or a route to it -- profiles/r05/cv2_probe.txt).

usage: python tests/golden/gen_cv2_golden.py        (needs cv2; small rows only, the fixture stays under a megabyte)"""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cases():
    rng = np.random.default_rng(185)
    out = []
    for L, M in [(400, 40), (5000, 700), (30000, 4097)]:
        for dtype in (np.float32, np.uint8):
            if dtype == np.uint8:
                dst = rng.integers(0, 256, L, dtype=np.uint8)
            else:
                dst = (rng.standard_normal(L) * 0.15 + 0.5).clip(0, 1).astype(np.float32)
            p = int(rng.integers(0, L - M + 1))
            src = dst[p:p + M].copy()
            noise = rng.standard_normal(M) * (6.0 if dtype == np.uint8 else 0.02)
            src = (src.astype(np.float64) + noise).clip(0, 255 if dtype == np.uint8 else 1).astype(dtype)
            out.append((dst, src, p))
    return out


def main():
    import cv2
    doc = {"cv2_version": cv2.__version__, "generator": "tests/golden/gen_cv2_golden.py", "cases": []}
    for dst, src, planted in cases():
        entry = {"dtype": str(dst.dtype), "L": int(dst.shape[0]), "M": int(src.shape[0]), "planted": planted,
                 "dst": base64.b64encode(dst.tobytes()).decode(), "src": base64.b64encode(src.tobytes()).decode(), "rows": {}}
        for name, code in (("sqdiff_normed", cv2.TM_SQDIFF_NORMED), ("ccoeff_normed", cv2.TM_CCOEFF_NORMED)):
            row = np.asarray(cv2.matchTemplate(dst.reshape(1, -1), src.reshape(1, -1), code), np.float32).reshape(-1)
            entry["rows"][name] = base64.b64encode(row.tobytes()).decode()
        doc["cases"].append(entry)
    path = os.path.join(ROOT, "tests", "golden", "cv2_match_template.json")
    with open(path, "w") as f:
        json.dump(doc, f)
    print("wrote", path, "cv2", cv2.__version__)


if __name__ == "__main__":
    main()
