"""The distance between the exactly rounded oracle and what the REAL cv2.matchTemplate may return (wav.py:185), without cv2:
oracle.match_template_cv2_model restates crossCorr's blocking and crossCorr's working precision -- float64 DFT for float32
streams, float32 DFT for uint8 streams -- through SciPy's FFT.  Oracle against oracle: test infrastructure only.
The full table (24 searches per row, configs[0] and configs[1] sizes) is profiles/r04/cv2_noise_model.json, made by
tools/cv2_noise_model.py; DESIGN.md section 4 quotes it."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_optimal_dft_size_known_values(oracle):
    # cv::getOptimalDFTSize's table holds the 5-smooth numbers: spot values incl. the one a 3 s pattern at 12 kHz takes
    assert [oracle.optimal_dft_size(n) for n in (1, 2, 7, 11, 17, 257, 1000, 1025)] == [1, 2, 8, 12, 18, 270, 1000, 1080]
    assert oracle.optimal_dft_size(162000 + 36000 - 1) == 200000
    for n in (3, 97, 4097, 65537, 197999):
        v = oracle.optimal_dft_size(n)
        w = v
        for p in (2, 3, 5):
            while w % p == 0:
                w //= p
        assert v >= n and w == 1


def test_model_blocks_cover_every_position(oracle):
    # several blocks, a ragged last block, a block larger than the row: the model's blocking must not lose a position
    rng = np.random.default_rng(3)
    for L, M in ((300, 7), (5000, 700), (20000, 333), (1000, 999)):
        img = rng.random(L, dtype=np.float32)
        t = rng.random(M, dtype=np.float32)
        a = oracle.match_template_direct(img, t)[0]
        b = oracle.match_template_cv2_model(img, t)[0]
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)


@pytest.mark.parametrize("sample_type", ["float32", "uint8"])
@pytest.mark.parametrize("method", ["sqdiff_normed", "ccoeff_normed"])
def test_distance_to_the_real_call_at_config0_sizes(oracle, sample_type, method):
    import cv2_noise_model as nm
    r = nm.measure("configs[0]", 300.0, 10.0, 6, sample_type, method, seed=20260924)
    assert r["max_idx_diff"] == 0                       # the reported shift does not depend on the DFT's precision
    if sample_type == "float32":
        # float64 DFT: its noise disappears in the float32 rounding of corr -- the model equals the oracle bit for bit
        assert r["max_row_abs_diff"] == 0.0
    else:
        # float32 DFT on corr ~ M * 128^2: a few float32 quanta of corr, i.e. ~1e-6 absolute in score units
        assert r["max_row_abs_diff"] <= 8e-6 and r["max_abs_score_diff"] <= 4e-6


def test_uint8_at_config1_size_exceeds_a_pure_relative_gate(oracle):
    """What DESIGN.md section 4 states: on uint8 streams with +-60 s windows the best SQDIFF score is ~3e-3 and the float32
    DFT moves it by up to ~6e-7 -- 2e-4 RELATIVE.  BASELINE.json's "1e-4 rel" cannot be demanded of real cv2 itself there;
    1e-4 * score + 1e-6 can.  The index is unaffected."""
    import cv2_noise_model as nm
    r = nm.measure("configs[1]", 2700.0, 60.0, 4, "uint8", "sqdiff_normed", seed=20260924)
    assert r["max_idx_diff"] == 0
    assert r["max_abs_score_diff"] <= 1e-4 * r["median_best_score"] + 1e-6
    assert r["max_row_abs_diff"] <= 4e-6
