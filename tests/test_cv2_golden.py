"""The oracle's restatement of cv2.matchTemplate against rows the REAL call produced (tests/golden/cv2_match_template.json,
written by tests/golden/gen_cv2_golden.py wherever cv2 imports).  While no machine of this build has had OpenCV the fixture does
not exist and this test skips -- DESIGN.md section 5 says "parity unpinned" for exactly that reason; the day the fixture is
committed the oracle is pinned on every machine, at BASELINE.json's gate."""
import base64
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "cv2_match_template.json")
RTOL, ATOL, UINT8_DFT_SLACK = 1e-4, 2.5e-7, 2e-6


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="no cv2-made fixture yet (cv2 is not installable in this image or on the GPU boxes)")
@pytest.mark.parametrize("method", ["sqdiff_normed", "ccoeff_normed"])
def test_oracle_equals_rows_made_by_real_cv2(oracle, method):
    with open(FIXTURE) as f:
        doc = json.load(f)
    assert doc["cases"]
    for c in doc["cases"]:
        dt = np.dtype(c["dtype"])
        dst = np.frombuffer(base64.b64decode(c["dst"]), dt)
        src = np.frombuffer(base64.b64decode(c["src"]), dt)
        ref = np.frombuffer(base64.b64decode(c["rows"][method]), np.float32)
        ours = oracle.match_template(dst, src, method=method)[0]
        assert ours.shape == ref.shape
        err = np.abs(ours.astype(np.float64) - ref.astype(np.float64))
        gate = RTOL * np.abs(ref.astype(np.float64)) + ATOL + (UINT8_DFT_SLACK if dt == np.uint8 else 0.0)
        assert (err <= gate).all(), (c["dtype"], c["L"], c["M"], float((err / gate).max()))
        pick = np.argmin if method == "sqdiff_normed" else np.argmax
        assert abs(int(pick(ours)) - int(pick(ref))) <= 1
