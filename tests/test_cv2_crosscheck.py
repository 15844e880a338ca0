"""Cross-check against the call the reference actually makes (wav.py:185: cv2.matchTemplate), wherever cv2 imports.

It does not in the build image (no OpenCV, no network: SURVEY F4) -- these tests then SKIP and parity stays
"unpinned at the cv2 boundary" (DESIGN.md section 5).  On any machine with OpenCV they run by themselves and turn the
oracle's restatement -- and, with a GPU, the HIP path -- into a comparison with the real thing.

Tolerance: BASELINE.json's (shift within +-1 sample, score within 1e-4 relative), plus the float32 quantum of cv2's
stored cross term (2.5e-7 absolute).  cv2's crossCorr runs its block DFT in float32 for CV_8U and CV_32F input, so real
cv2 output carries ~1e-6 * corr of noise that neither the oracle (exactly rounded corr) nor the HIP path models: equal
indices are demanded only where the oracle's own row separates the best two positions by more than that noise.
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

RTOL, ATOL = 1e-4, 2.5e-7


def _cases():
    rng = np.random.default_rng(185)
    out = []
    for L, M in [(400, 40), (5000, 700), (30000, 4097), (120000, 12000)]:
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


def _agree(row_a, row_b, method):
    pick = np.argmin if method == "sqdiff_normed" else np.argmax
    ia, ib = int(pick(row_a)), int(pick(row_b))
    assert abs(float(row_a[ia]) - float(row_b[ib])) <= RTOL * abs(float(row_b[ib])) + ATOL + 2e-6
    if ia != ib:                      # acceptable only as a tie inside cv2's own float32-DFT noise
        assert abs(float(row_b[ia]) - float(row_b[ib])) <= 4e-6, (ia, ib)
    return ia, ib


@pytest.mark.parametrize("method", ["sqdiff_normed", "ccoeff_normed"])
def test_oracle_restatement_equals_cv2(oracle, method):
    """oracle/match_template.c against cv2.matchTemplate, whole result rows."""
    for dst, src, planted in _cases():
        ref = oracle.match_template_cv2(dst, src, method)[0]
        ours = oracle.match_template(dst, src, method=method)[0]
        assert ref.shape == ours.shape
        assert np.abs(ours.astype(np.float64) - ref).max() <= RTOL + 4e-6, np.abs(ours - ref).max()
        ia, ib = _agree(ours, ref, method)
        assert abs(ia - planted) <= 1 and abs(ib - planted) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["sqdiff_normed", "ccoeff_normed"])
@pytest.mark.parametrize("path", ["fft", "direct"])
def test_hip_path_equals_cv2(oracle, method, path):
    """The product (through the C ABI) against cv2.matchTemplate + argmin / argmax."""
    from sushi_amd.device import DeviceStream, SearchBatch
    for dst, src, planted in _cases():
        M, P = src.shape[0], dst.shape[0] - src.shape[0] + 1
        b = SearchBatch(DeviceStream(dst), DeviceStream(src), [0], [M], [0], [P], path=path, method=method)
        b.run()
        idx, score = b.results()
        ref = oracle.match_template_cv2(dst, src, method)[0]
        pick = np.argmin if method == "sqdiff_normed" else np.argmax
        ib = int(pick(ref))
        assert abs(int(idx[0]) - ib) <= 1
        assert abs(float(score[0]) - float(ref[ib])) <= RTOL * abs(float(ref[ib])) + ATOL + 2e-6
        if int(idx[0]) != ib:
            assert abs(float(ref[int(idx[0])]) - float(ref[ib])) <= 4e-6
