"""cv2.TM_CCOEFF_NORMED + argmax (the method BASELINE.json's wording names; the reference itself calls TM_SQDIFF_NORMED,
SURVEY F1): the oracle's restatement against the definition, and both HIP paths (FFT ranking + exact evaluation, direct MFMA
kernel) against the oracle."""
import numpy as np
import pytest

CC = "ccoeff_normed"


def _signal(rng, n, dtype):
    x = np.convolve(rng.standard_normal(n + 7), np.ones(8) / 8.0, mode="valid")      # audio-like low pass
    x = 0.5 + 0.35 * x / np.abs(x).max()
    return (x * 255).astype(np.uint8) if dtype == np.uint8 else x.astype(np.float32)


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_oracle_ccoeff_formulations_agree(oracle, dtype):
    rng = np.random.default_rng(3)
    for L, M in [(3000, 1), (3000, 2), (5000, 700), (9000, 4097), (2048, 2048)]:
        img = _signal(rng, L, dtype)
        t = _signal(rng, M, dtype)
        if M > 2 and L > M + 500:
            t = img[500:500 + M].copy()                                  # a planted copy: score 1 at index 500
        a = oracle.match_template_direct(img, t, method=CC)[0]
        b = oracle.match_template_fft(img, t, method=CC)[0]
        d = oracle.definition_ccoeff_normed(img, t)
        assert a.shape == (L - M + 1,) and a.dtype == np.float32
        assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= (0 if dtype == np.uint8 else 2e-6)
        if M > 2:
            assert np.abs(a - d).max() <= 2e-6, (L, M)
            assert (np.abs(a) <= 1.0).all()
        if M > 2 and L > M + 500:
            assert oracle.argmax_first(a) == 500 and a[500] >= 1.0 - 1e-5     # (cv2 keeps corr in float32)


def test_oracle_ccoeff_degenerate_cases(oracle):
    rng = np.random.default_rng(4)
    img = np.zeros(600, np.float32)
    img[100:300] = rng.random(200, dtype=np.float32)
    flat_t = np.full(50, 0.3, np.float32)
    assert (oracle.match_template_direct(img, flat_t, method=CC)[0] == 1.0).all()      # flat template: cv2 returns all ones
    t = rng.random(50, dtype=np.float32)
    r = oracle.match_template_direct(img, t, method=CC)[0]
    assert (r[:50] == 0.0).all() and (r[320:] == 0.0).all()                            # flat windows: t = 0 -> 0
    per = np.tile(rng.random(64, dtype=np.float32), 20)                                # periodic image: ties -> first index
    r = oracle.match_template_direct(per, per[128:128 + 100].copy(), method=CC)[0]
    assert oracle.argmax_first(r) == 0 and int(r.argmax()) == 0
    with pytest.raises(ValueError):
        oracle.match_template_direct(img, t, method="ccorr")


def _check(oracle, dtype, res, idx, score):
    o_idx = oracle.argmax_first(res)
    if dtype == np.uint8:
        assert int(idx) == o_idx
        assert np.float32(score).view(np.uint32) == np.float32(res[o_idx]).view(np.uint32)
    else:
        # the float32 corr cv2 stores may round one ulp apart (the direct kernel's cross term differs from the oracle's
        # double sum in the 8th digit): one ulp of corr over the denominator, < 5e-6 on these signals
        assert abs(float(score) - float(res[o_idx])) <= 1e-4 * abs(float(res[o_idx])) + 5e-6, (score, res[o_idx])
        if int(idx) != o_idx:
            assert abs(float(res[int(idx)]) - float(res[o_idx])) <= 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
@pytest.mark.parametrize("variant", [0, 1, 2, "fft"])
def test_hip_ccoeff_matches_oracle(oracle, dtype, variant):
    from sushi_amd.device import DeviceStream, SearchBatch
    rng = np.random.default_rng(11 + (3 if variant == "fft" else variant))
    dst = _signal(rng, 90000, dtype)
    src = _signal(rng, 30000, dtype)
    src[2000:2000 + 6000] = dst[40000:46000]                               # planted copy
    if dtype == np.float32:
        src[2000:8000] += (rng.standard_normal(6000) * 0.01).astype(np.float32)
    dst[70000:70400] = dst[70000]                                           # a flat stretch: windows with t = 0
    offs = [2000, 2000, 100, 9000, 2500, 15000, 7]
    lens = [6000, 3000, 1, 4097, 700, 12000, 2]
    wst = [30000, 0, 500, 60000, 69900, 100, 1000]
    npos = [20001, 84001, 3000, 25000, 600, 77000, 5000]
    kw = dict(path="fft") if variant == "fft" else dict(path="direct", variant=variant)
    b = SearchBatch(DeviceStream(dst), DeviceStream(src), offs, lens, wst, npos, method=CC, **kw)
    b.run()
    idx, score = b.results()
    if variant == "fft":
        d = b.diagnostics()
        assert d["all_positions"] == 0 and d["max_bound_ratio"] < 1.0 and d["max_bound_ratio_noncandidate"] < 1.0
    for k in range(len(offs)):
        res = oracle.match_template(dst[wst[k]:wst[k] + npos[k] + lens[k] - 1], src[offs[k]:offs[k] + lens[k]], method=CC)[0]
        _check(oracle, dtype, res, idx[k], score[k])
    assert wst[0] + idx[0] == 40000 and score[0] > 0.99
    # the default method on the same handle type is untouched by all this
    b2 = SearchBatch(DeviceStream(dst), DeviceStream(src), offs[:2], lens[:2], wst[:2], npos[:2], **kw)
    b2.run()
    idx2, score2 = b2.results()
    assert wst[0] + idx2[0] == 40000 and score2[0] < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["direct", "fft"])
def test_ccoeff_flat_template_flat_windows_and_unknown_method(oracle, path):
    from sushi_amd.common import SushiError
    from sushi_amd.device import DeviceStream, SearchBatch
    rng = np.random.default_rng(5)
    dst = _signal(rng, 20000, np.uint8)
    src = np.full(5000, 77, np.uint8)                                       # flat template: cv2's result is all ones
    d, s = DeviceStream(dst), DeviceStream(src)
    b = SearchBatch(d, s, [10], [800], [100], [15000], path=path, method=CC)
    b.run()
    idx, score = b.results()
    assert idx[0] == 0 and score[0] == 1.0
    with pytest.raises(SushiError):
        SearchBatch(d, s, [10], [800], [100], [15000], path=path, method="ccorr")
    # a destination that is digital silence but for one stretch: every window outside it has no variance (cv2: 0), the
    # f32 ranking stage cannot bound those (uncertain positions: always candidates), the exact stages settle them
    for dtype in (np.uint8, np.float32):
        dst2 = np.full(60000, 128 if dtype == np.uint8 else 0.5, dtype)
        live = _signal(rng, 9000, dtype)
        dst2[30000:39000] = live
        src2 = live[2000:5000].copy()
        b = SearchBatch(DeviceStream(dst2), DeviceStream(src2), [0, 0], [3000, 1500], [0, 20000], [57001, 30001], path=path, method=CC)
        b.run()
        idx, score = b.results()
        for k, (m, w, p) in enumerate([(3000, 0, 57001), (1500, 20000, 30001)]):
            res = oracle.match_template(dst2[w:w + p + m - 1], src2[:m], method=CC)[0]
            _check(oracle, dtype, res, idx[k], score[k])
        assert idx[0] == 32000 and idx[1] == 12000
    # nothing but silence: every position scores 0, the first one wins
    dst3 = np.full(30000, 99, np.uint8)
    b = SearchBatch(DeviceStream(dst3), DeviceStream(_signal(rng, 4000, np.uint8)), [0], [4000], [0], [26001], path=path, method=CC)
    b.run()
    idx, score = b.results()
    assert idx[0] == 0 and score[0] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("sample_type", ["float32", "uint8"])
def test_ccoeff_at_configs2_size(oracle, sample_type):
    """BASELINE configs[2] sizes (2-h 12 kHz streams, +-120 s: P = 2,880,001) with the method BASELINE.json names, on the
    FFT path; every search against the oracle's FFT port, one of them also against the direct kernel."""
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream
    rate, seconds, off, window = 12000, 7200.0, 11.5, 120.0
    dst_pcm = synth.make_dst_pcm(seconds, rate, seed=31)
    src_pcm = synth.make_src_pcm(dst_pcm, int(off * rate), seed=32)
    dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=sample_type)
    src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=sample_type)
    del dst_pcm, src_pcm
    events = synth.make_events(12, seconds, window + off, seed=33)
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, off, window, seed=34)
    scores, times, pos = dst.find_substreams(pats, centres, wins, with_index=True, method=CC)       # the default path: FFT
    dtype = np.uint8 if sample_type == "uint8" else np.float32
    for k, (s, e) in enumerate(events):
        assert abs((times[k] - s) - off) <= 1.0 / rate + 1e-9 and scores[k] > 0.9
        st, lo, p = dst._window(pats[k].shape[1], centres[k], wins[k])
        res = oracle.match_template_fft(dst.data[:, lo:lo + p + pats[k].shape[1] - 1], pats[k], method=CC)[0]
        _check(oracle, dtype, res, pos[k] - lo, scores[k])
    k = 5
    st, lo, p = dst._window(pats[k].shape[1], centres[k], wins[k])
    b = SearchBatch(dst.device_stream(), src.device_stream(), [src._get_sample_for_time(events[k][0])], [pats[k].shape[1]],
                    [lo], [p], path="direct", method=CC)
    b.run()
    idx_d, score_d = b.results()
    assert lo + int(idx_d[0]) == pos[k] and abs(float(score_d[0]) - float(scores[k])) <= (0.0 if sample_type == "uint8" else 5e-6)


@pytest.mark.gpu
def test_wavstream_find_substreams_with_ccoeff(oracle):
    """The drop-in's batched call with method='ccoeff_normed': the reference's window arithmetic (wav.py:178-184), the
    other cv2 method, arg-max; against the oracle on the same slices."""
    from sushi_amd import synth
    from sushi_amd.wav import WavStream
    rate = 4000
    dst_pcm = synth.make_dst_pcm(120, rate, seed=7)
    src_pcm = synth.make_src_pcm(dst_pcm, int(2.5 * rate), seed=8)
    dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type="uint8")
    src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type="uint8")
    spans = [(10.0, 12.5), (40.0, 41.0), (90.0, 94.0)]
    pats = [src.get_substream(a, b) for a, b in spans]
    centres = [a + 2.0 for a, _ in spans]
    scores, times, pos = dst.find_substreams(pats, centres, [5.0] * 3, with_index=True, method=CC)
    for k, (a, b) in enumerate(spans):
        st, lo, p = dst._window(pats[k].shape[1], centres[k], 5.0)
        res = oracle.match_template(dst.data[:, lo:lo + p + pats[k].shape[1] - 1], pats[k], method=CC)[0]
        j = oracle.argmax_first(res)
        assert pos[k] == lo + j and np.float32(scores[k]).view(np.uint32) == np.float32(res[j]).view(np.uint32)
        assert abs((times[k] - a) - 2.5) <= 1.0 / rate + 1e-9 and scores[k] > 0.9
