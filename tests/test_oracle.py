"""The oracle against itself (three independent formulations), its known answers, and the golden
vectors produced by the reference's own wav.py bytecode (tests/golden/gen_find_substream_golden.py)."""
import numpy as np
import pytest


def _rand_case(rng, L, M, dtype):
    if dtype == np.uint8:
        img = rng.integers(0, 256, L, dtype=np.uint8)
        t = rng.integers(0, 256, M, dtype=np.uint8)
    else:
        img = rng.random(L, dtype=np.float32)
        t = rng.random(M, dtype=np.float32)
    return img, t


@pytest.mark.parametrize("L,M", [(1, 1), (5, 5), (64, 1), (257, 33), (1000, 999), (4096, 300), (3001, 1200)])
def test_direct_matches_definition_f32(oracle, L, M):
    rng = np.random.default_rng(L * 7919 + M)
    img, t = _rand_case(rng, L, M, np.float32)
    got = oracle.match_template_direct(img, t, corr_f32=False)[0]
    ref = oracle.definition_sqdiff_normed(img, t)
    ref = np.minimum(ref, 1.0)
    assert got.shape == (L - M + 1,)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1.5e-7)
    # cv2's float32 corr storage moves results by at most one float32 ulp of corr, scaled
    got32 = oracle.match_template_direct(img, t, corr_f32=True)[0]
    assert np.abs(got32 - got).max() <= 4e-7


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
@pytest.mark.parametrize("L,M", [(300, 7), (5000, 700), (20000, 4097)])
def test_fft_matches_direct(oracle, L, M, dtype):
    rng = np.random.default_rng(L + M)
    img, t = _rand_case(rng, L, M, dtype)
    a = oracle.match_template_direct(img, t, corr_f32=False)
    b = oracle.match_template_fft(img, t, corr_f32=False)
    if dtype == np.uint8:
        assert (a == b).all()                    # integer sums: bit-exact
    else:
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)
    assert oracle.argmin_first(a[0]) == int(a.argmin(axis=1)[0])


def test_u8_is_exact_integer_arithmetic(oracle):
    rng = np.random.default_rng(5)
    img, t = _rand_case(rng, 2000, 400, np.uint8)
    got = oracle.match_template_direct(img, t, corr_f32=False)[0]
    i64, t64 = img.astype(np.int64), t.astype(np.int64)
    for p in (0, 17, 1600):
        w = i64[p:p + 400]
        num = float(((t64 - w) ** 2).sum())
        den = np.sqrt(float((t64 * t64).sum())) * np.sqrt(float((w * w).sum()))
        exp = np.float32(num / den) if num < den else np.float32(1.0)
        assert abs(float(got[p]) - float(exp)) <= 1.2e-7


def test_planted_copy_and_first_index_ties(oracle):
    rng = np.random.default_rng(1)
    img = rng.random(3000, dtype=np.float32)
    t = img[1234:1234 + 500].copy()
    r = oracle.match_template(img, t)
    assert int(r.argmin(axis=1)[0]) == 1234 and r[0, 1234] <= 1e-6
    # periodic image: every period is an exact match; NumPy argmin (wav.py:186) takes the first.
    # Dyadic sample values (k/64) and uint8 keep every sum exact, so the ties are exact ties
    # (with arbitrary floats the rounding noise of corr decides which period wins, in cv2 too).
    for period in ((rng.integers(0, 64, 50) / 64.0).astype(np.float32), rng.integers(0, 256, 50, dtype=np.uint8)):
        img = np.tile(period, 40)
        t = img[10:10 + 120].copy()
        r = oracle.match_template(img, t)
        assert int(r.argmin(axis=1)[0]) == 10 and r[0, 10] == 0.0 and r[0, 60] == 0.0
        assert oracle.argmin_first(r[0]) == 10


def test_degenerate_windows_give_one_never_nan(oracle):
    img = np.zeros(100, np.float32)
    img[60:] = 0.5
    t = np.full(10, 0.25, np.float32)
    r = oracle.match_template_direct(img, t)[0]
    assert np.isfinite(r).all()
    assert (r[:50] == 1.0).all()                 # all-zero windows: t == 0 -> 1 (common_matchTemplate)
    assert r.max() <= 1.0 and r.min() >= 0.0
    z = oracle.match_template_direct(np.zeros(20, np.float32), np.zeros(5, np.float32))[0]
    assert (z == 1.0).all()
    u = oracle.match_template_direct(np.zeros(20, np.uint8), np.zeros(5, np.uint8))[0]
    assert (u == 1.0).all()


def test_template_larger_than_image_raises(oracle):
    with pytest.raises(ValueError):
        oracle.match_template_direct(np.zeros(5, np.float32), np.zeros(6, np.float32))


def test_halves_identity(oracle):
    """corr_full[p] = corr_left[p] + corr_right[p + len(left)] at the FFT layer."""
    rng = np.random.default_rng(9)
    img = rng.random(4000, dtype=np.float32)
    t = rng.random(601, dtype=np.float32)
    k = 601 // 2
    full = oracle.cross_correlate_fft(img, t)
    left = oracle.cross_correlate_fft(img, t[:k])
    right = oracle.cross_correlate_fft(img, t[k:])
    P = full.shape[0]
    np.testing.assert_allclose(full, left[:P] + right[k:k + P], rtol=1e-12, atol=1e-9)


def test_golden_index_arithmetic(oracle, golden_index):
    """OracleWavStream window arithmetic == the reference's wav.py:164-188 executed under Py3."""
    cache = {}
    n_ok = 0
    for c in golden_index["cases"]:
        key = (c["sample_rate"], c["framerate"], c["seconds"], c["dtype"])
        if key not in cache:
            cache[key] = oracle.OracleWavStream(np.zeros((1, c["data_len"]), c["dtype"]), c["sample_rate"],
                                                c["sample_count"], c["padding_size"])
        s = cache[key]
        pat = s.get_substream(c["pat_start"], c["pat_end"])
        off = (pat.__array_interface__["data"][0] - s.data.__array_interface__["data"][0]) // pat.itemsize
        assert (off, pat.shape[1]) == (c["pat_off"], c["pat_len"])
        start_time, lo, hi = s.search_bounds(c["pat_len"], c["center"], c["window"])
        assert lo == c["search_off"] and hi - lo == c["search_len"]
        if c["ok"]:
            def stub(search, pattern, corr_f32, c=c):
                r = np.ones((1, search.shape[1] - pattern.shape[1] + 1), np.float32)
                r[0, c["want_min"] % r.shape[1]] = 0.25
                return r
            diff, t = s.find_substream(pat, c["center"], c["window"], matcher=stub)
            assert t == c["time"] and float(diff) == c["diff"]
            n_ok += 1
        else:
            assert c["search_len"] < c["pat_len"]
    assert n_ok > 1000
    for c in golden_index["clip"]:
        assert oracle.clip(c["v"], c["lo"], c["hi"]) == c["out"]


def test_known_answers_by_hand(oracle):
    """Known answers worked out by hand from OpenCV's documented TM_SQDIFF_NORMED formula
    R(x) = sum (T(x') - I(x + x'))^2 / sqrt(sum T(x')^2 * sum I(x + x')^2), exact rationals / surds:
      I = [1 2 3 4], T = [2 3]   -> numerators 2, 0, 2; denominators sqrt(13*5), 13, sqrt(13*25)
      I = [3 0 4 0 3], T = [0 5] -> window [0 4] gives 1/20 (numerator 1, denominator sqrt(25*16)),
                                    the other windows' numerators reach their denominators: clamped to 1
      uint8 I = [10 20 30 40 50], T = [20 30 40] -> 300/sqrt(2900*1400), 0, 300/sqrt(2900*5000)."""
    from fractions import Fraction
    from math import sqrt
    img = np.array([1, 2, 3, 4], np.float32)
    t = np.array([2, 3], np.float32)
    got = oracle.match_template_direct(img, t)[0]
    exp = [2 / sqrt(65.0), 0.0, 2 / sqrt(325.0)]
    np.testing.assert_allclose(got, np.array(exp, np.float32), rtol=0, atol=6e-8)
    assert oracle.argmin_first(got) == 1
    img = np.array([3, 0, 4, 0, 3], np.float32)
    t = np.array([0, 5], np.float32)
    got = oracle.match_template_direct(img, t)[0]
    # windows [3 0]: num 9+25=34 >= den 15 -> 1; [0 4]: num 1, den 20; [4 0]: num 16+25=41 >= 20 -> 1; [0 3]: 4/15
    exp = [1.0, float(Fraction(1, 20)), 1.0, float(Fraction(4, 15))]
    np.testing.assert_allclose(got, np.array(exp, np.float32), rtol=0, atol=6e-8)
    assert oracle.argmin_first(got) == 1
    img = np.array([10, 20, 30, 40, 50], np.uint8)
    t = np.array([20, 30, 40], np.uint8)
    for fn in (oracle.match_template_direct, oracle.match_template_fft):
        got = fn(img, t)[0]
        exp = np.array([300 / sqrt(2900.0 * 1400.0), 0.0, 300 / sqrt(2900.0 * 5000.0)], np.float32)
        assert (got == exp).all()                           # integer sums, correctly rounded quotient
