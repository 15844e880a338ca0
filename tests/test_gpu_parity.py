"""Parity of the HIP path (through the C ABI) against the CPU oracle.  All tests need an MI355X.

Tolerances (BASELINE.json north_star: shift +-1 sample, float32 score within 1e-4 relative):
  * uint8 streams  : every sum is an exact integer on both sides -> index AND float32 score bit-exact.
  * float32 streams: |score - oracle| <= 1e-4 * oracle + 2.5e-7.  The absolute term is the
    granularity of the reference's own output: cv2 stores the cross-correlation in a float32 Mat
    before forming wndSum2 - 2*corr + templSum2, so one float32 ulp of corr moves the score by
    2 * 2^-23 * corr / sqrt(sum T^2 * sum I^2) <= 2.4e-7 (oracle/match_template.c).  A pure relative
    bound is meaningless where the score itself is 0 (exact copies).
    Index: equal, or -- when two positions are tied to within that quantum -- the oracle's score
    at our index is within 2.5e-7 of the oracle's minimum.  Planted-offset cases additionally
    demand the shift to be within +-1 sample of the planted one.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCORE_RTOL = 1e-4
SCORE_ATOL = 2.5e-7
TIE_ATOL = 2.5e-7


def _score_ok(score, ref):
    return abs(float(score) - float(ref)) <= SCORE_RTOL * float(ref) + SCORE_ATOL


def _check_f32(res_row, idx, score):
    o_idx = int(res_row.argmin())
    o_score = float(res_row[o_idx])
    assert _score_ok(score, o_score), (score, o_score)
    if int(idx) != o_idx:
        assert abs(float(res_row[int(idx)]) - o_score) <= TIE_ATOL, (idx, o_idx, res_row[int(idx)], o_score)


def _check_u8(res_row, idx, score):
    o_idx = int(res_row.argmin())
    assert int(idx) == o_idx
    assert np.float32(score).view(np.uint32) == np.float32(res_row[o_idx]).view(np.uint32)


PATHS = [0, 1, 2, "fft"]          # direct-kernel variants (tile sizes) and the overlap-save FFT path


def _run_batch(dst_row, src_row, offs, lens, wstart, npos, variant=None, want_batch=False, **kw):
    """variant: 0/1/2 = direct kernel with that tile size, 'fft' = FFT path, None = library default."""
    from sushi_amd.device import DeviceStream, SearchBatch
    dst = DeviceStream(dst_row)
    src = DeviceStream(src_row)
    if variant == "fft":
        b = SearchBatch(dst, src, offs, lens, wstart, npos, path="fft", **kw)
    elif variant is None:
        b = SearchBatch(dst, src, offs, lens, wstart, npos, **kw)
    else:
        b = SearchBatch(dst, src, offs, lens, wstart, npos, variant=variant, path="direct", **kw)
    b.run()
    return (b.results(), b) if want_batch else b.results()


@pytest.fixture(params=["fft", "direct"])
def hip_path(request, monkeypatch):
    """Runs a drop-in (WavStream) test once per library path."""
    monkeypatch.setenv("SUSHI_HIP_PATH", request.param)
    return request.param


def test_extension_is_loaded_and_device_is_gfx950():
    from sushi_amd import _native
    L = _native.lib()
    assert L.sushi_hip_device_ok() == 0
    import torch
    assert torch.cuda.is_available()         # loading the library first must not hide the GPU from torch (one HIP runtime)


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
@pytest.mark.parametrize("n", [1, 5, 4095, 4096, 4097, 300001, 5000000])
def test_prepare_stream(dtype, n):
    from sushi_amd.device import DeviceStream
    rng = np.random.default_rng(n)
    x = rng.integers(0, 256, n, dtype=np.uint8) if dtype == np.uint8 else rng.random(n, dtype=np.float32)
    d = DeviceStream(x)
    c = 128.0 if dtype == np.uint8 else 0.5
    xc = (x.astype(np.float32) - np.float32(c))
    assert (d.xc.cpu().numpy() == xc).all()
    s1 = np.concatenate(([0.0], np.cumsum(x.astype(np.float64))))        # of the samples as they are
    s2 = np.concatenate(([0.0], np.cumsum(x.astype(np.float64) ** 2)))
    g1, g2 = d.s1.cpu().numpy(), d.s2.cpu().numpy()
    if dtype == np.uint8:
        assert (g1 == s1).all() and (g2 == s2).all()          # integers: exact in any summation order
    else:
        np.testing.assert_allclose(g1, s1, rtol=0, atol=1e-9 * max(1.0, np.abs(s1).max()))
        np.testing.assert_allclose(g2, s2, rtol=1e-11, atol=1e-9)   # np.cumsum itself rounds sequentially
    # window energies for the FFT path: s2[e] = ubase[e // 4096] + urel[e]
    from sushi_amd import _native
    assert _native.lib().sushi_hip_fft_block() == 4096
    nb = (n + 4095) // 4096
    ubase = d.base.cpu().numpy()[:nb + 1]
    urel = d.urel.cpu().numpy()
    u = np.concatenate(([0.0], np.cumsum(x.astype(np.float64) ** 2)))
    e = np.arange(n + 1)
    rebuilt = ubase[e // 4096] + urel.astype(np.float64)
    tol = 4096 * (255.0 ** 2 if dtype == np.uint8 else 1.0) * 2.0 ** -23
    assert np.abs(rebuilt - u).max() <= tol
    assert ubase[0] == 0 and abs(ubase[nb] - u[n]) <= 1e-12 * max(1.0, u[n])


@pytest.mark.parametrize("variant", PATHS)
@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
@pytest.mark.parametrize("L,M", [(1, 1), (40, 40), (1500, 100), (3000, 700), (20000, 1537), (9000, 4800), (30000, 9000)])
def test_random_search_vs_oracle(oracle, variant, dtype, L, M):
    rng = np.random.default_rng(L * 31 + M + (7 if variant == "fft" else variant))
    if dtype == np.uint8:
        dst = rng.integers(0, 256, L + 77, dtype=np.uint8)
        src = rng.integers(0, 256, M + 13, dtype=np.uint8)
    else:
        dst = rng.random(L + 77, dtype=np.float32)
        src = rng.random(M + 13, dtype=np.float32)
    ws, to = 41, 7
    P = L - M + 1
    idx, score = _run_batch(dst, src, [to], [M], [ws], [P], variant)
    res = oracle.match_template_direct(dst[ws:ws + L], src[to:to + M])[0]
    (_check_u8 if dtype == np.uint8 else _check_f32)(res, idx[0], score[0])


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_ragged_batch_all_variants(oracle, dtype):
    """One launch, many searches of very different sizes (incl. M = 1, P = 1, multi-tile P)."""
    rng = np.random.default_rng(11)
    n_dst, n_src = 120000, 30000
    if dtype == np.uint8:
        dst = rng.integers(0, 256, n_dst, dtype=np.uint8)
        src = rng.integers(0, 256, n_src, dtype=np.uint8)
    else:
        dst = (rng.standard_normal(n_dst) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
        src = (rng.standard_normal(n_src) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
    # plant a few real matches
    for k in range(6):
        a, m = 3000 * k + 17, 400 + 250 * k
        src[a:a + m] = dst[50000 + 1111 * k: 50000 + 1111 * k + m]
    offs, lens, wst, npos = [], [], [], []
    for k in range(40):
        m = int(rng.choice([1, 2, 31, 32, 33, 100, 513, 1024, 2500, 6000]))
        p = int(rng.choice([1, 2, 1023, 1024, 1025, 4097, 20000, 40000]))
        to = int(rng.integers(0, n_src - m))
        ws = int(rng.integers(0, n_dst - (p + m - 1)))
        offs.append(to); lens.append(m); wst.append(ws); npos.append(p)
    for k in range(6):
        a, m = 3000 * k + 17, 400 + 250 * k
        offs.append(a); lens.append(m); wst.append(30000); npos.append(60000)
    refs = [oracle.match_template(dst[w:w + p + m - 1], src[o:o + m])[0]
            for o, m, w, p in zip(offs, lens, wst, npos)]
    for variant in (0, 1, 2, "fft", None):
        idx, score = _run_batch(dst, src, offs, lens, wst, npos, variant)
        for k, res in enumerate(refs):
            (_check_u8 if dtype == np.uint8 else _check_f32)(res, idx[k], score[k])
        for k in range(6):
            assert idx[40 + k] == 20000 + 1111 * k          # planted copies found exactly


@pytest.mark.parametrize("variant", [2, "fft"])
def test_planted_copy_ties_and_degenerate(oracle, variant):
    def run(*args, **kw):
        return _run_batch(*args, variant=variant, **kw)

    rng = np.random.default_rng(1)
    img = rng.random(30000, dtype=np.float32)
    t = img[12345:12345 + 5000].copy()
    idx, score = run(img, t, [0], [5000], [0], [25001])
    assert idx[0] == 12345 and score[0] <= 1e-6
    # exact ties (dyadic floats / uint8): first index wins, like ndarray.argmin (wav.py:186)
    for period in ((rng.integers(0, 64, 50) / 64.0).astype(np.float32), rng.integers(0, 256, 50, dtype=np.uint8)):
        img = np.tile(period, 400)
        idx, score = run(img, img, [10], [120], [0], [img.shape[0] - 119])
        assert idx[0] == 10 and score[0] == 0.0
        idx, score = run(img, img, [10], [120], [23], [img.shape[0] - 119 - 23])
        assert idx[0] == 37 and score[0] == 0.0             # first p with (p + 23) % 50 == 10
    # all-zero windows -> 1.0 (never NaN), all-equal result -> index 0
    z = np.zeros(5000, np.float32)
    t = np.full(100, 0.25, np.float32)
    idx, score = run(z, t, [0], [100], [0], [4901])
    assert idx[0] == 0 and score[0] == 1.0
    zu = np.zeros(5000, np.uint8)
    (idx, score), b = run(zu, zu, [0], [100], [0], [4901], want_batch=True)
    assert idx[0] == 0 and score[0] == 1.0
    if variant == "fft":
        d = b.diagnostics()
        assert d["flagged"] == 1 and d["tiles_dense"] >= 4        # 4901 tied positions: evaluated exactly, tile by tile


@pytest.mark.parametrize("variant", [None, 2])
def test_halves_identity_on_gpu(oracle, variant):
    """Full / left / right searches of sushi.py:450-452: each one individually matches the oracle."""
    rng = np.random.default_rng(4)
    dst = (rng.standard_normal(80000) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
    src = dst[20000:26001].copy() + (rng.standard_normal(6001) * 0.01).astype(np.float32)
    k = 6001 // 2
    offs, lens = [0, 0, k], [6001, k, 6001 - k]
    wst, npos = [1000, 1000, 1000 + k], [60000, 60000, 60000]
    idx, score = _run_batch(dst, src, offs, lens, wst, npos, variant)
    for j in range(3):
        res = oracle.match_template(dst[wst[j]:wst[j] + npos[j] + lens[j] - 1], src[offs[j]:offs[j] + lens[j]])[0]
        _check_f32(res, idx[j], score[j])
    assert idx[0] == 19000 and idx[1] == 19000 and idx[2] == 19000


@pytest.mark.parametrize("sample_type", ["uint8", "float32"])
def test_find_substream_dropin_vs_oracle_with_clipping(oracle, sample_type, hip_path):
    """WavStream.find_substream vs the oracle's wav.py:177-188 restatement, including windows
    clipped at both ends of the stream, negative start times and NumPy slice truncation."""
    from sushi_amd import synth
    from sushi_amd.wav import WavStream
    dst_pcm = synth.make_dst_pcm(40, 12000, seed=5)
    src_pcm = synth.make_src_pcm(dst_pcm, 18000, seed=6)       # dst time = src time + 1.5 s
    dst = WavStream.from_samples(dst_pcm, 12000, sample_type=sample_type)
    src = WavStream.from_samples(src_pcm, 12000, sample_type=sample_type)
    odst = oracle.OracleWavStream(dst.data, dst.sample_rate, dst.sample_count, dst.padding_size)
    cases = [(5.0, 7.5, 6.5, 1.5), (5.0, 7.5, 6.5, 10), (0.0, 2.0, 0.2, 10), (0.0, 1.0, -3.0, 10),
             (36.0, 38.4, 39.5, 10), (33.0, 38.0, 45.0, 10), (10.0, 10.4, 11.5, 30), (20.0, 24.0, 3.0, 60)]
    for (a, b, c, w) in cases:
        pat = src.get_substream(a, b)
        diff, t = dst.find_substream(pat, c, w)
        rdiff, rt = odst.find_substream(pat, c, w)
        assert isinstance(t, float)
        if sample_type == "uint8":
            assert t == rt and np.float32(diff) == np.float32(rdiff)
        else:
            assert _score_ok(diff, rdiff), (diff, rdiff)
            if abs(t - rt) > 1e-12:
                # flat minimum (e.g. windows sliding into the constant padding): a different index is
                # acceptable only if the oracle itself scores it within the tie quantum of its minimum
                start_time, lo, hi = odst.search_bounds(pat.shape[1], c, w)
                row = oracle.match_template(odst.data[:, lo:hi], pat)[0]
                k = int(round((t - start_time) * 12000))
                assert abs(float(row[k]) - float(rdiff)) <= TIE_ATOL, (t, rt, row[k], rdiff)
    # the three searches of sushi.py:450-452 in one batched call, patterns being np.split views
    pat = src.get_substream(12.0, 15.0)
    left, right = np.split(pat, [pat.shape[1] // 2], axis=1)
    off = left.shape[1] / float(src.sample_rate)
    diffs, times = dst.find_substreams([pat, left, right], [13.5, 13.5, 13.5 + off], [10, 10, 10])
    assert abs(times[0] - 13.5) <= 1 / 12000 + 1e-9 and abs(times[1] - 13.5) <= 1 / 12000 + 1e-9
    assert abs(times[2] - off - 13.5) <= 1 / 12000 + 1e-9
    # a pattern that is NOT a view of a live stream takes the upload path and gives the same answer
    d2, t2 = dst.find_substream(pat.copy(), 13.5, 10)
    assert t2 == times[0] and np.float32(d2) == np.float32(diffs[0])
    # pattern longer than the clipped window: cv2.error in the reference, SushiError here
    from sushi_amd import SushiError
    with pytest.raises(SushiError):
        dst.find_substream(src.get_substream(1.0, 13.5), 39.99, 1.5)


@pytest.mark.parametrize("sample_type", ["uint8", "float32"])
def test_config1_global_offset_recovered(oracle, sample_type, tmp_path, hip_path):
    """BASELINE config 0: 50 events, 5-min 12 kHz streams (through real WAV files), +1.5 s offset."""
    import os
    from sushi_amd import synth
    from sushi_amd.wav import WavStream
    dst_pcm = synth.make_dst_pcm(300, 12000, seed=20260924)
    src_pcm = synth.make_src_pcm(dst_pcm, 18000, seed=20260925)
    pd, ps = os.path.join(tmp_path, "dst.wav"), os.path.join(tmp_path, "src.wav")
    synth.write_wav(pd, dst_pcm, 12000)
    synth.write_wav(ps, src_pcm, 12000)
    dst = WavStream(pd, sample_type=sample_type)
    src = WavStream(ps, sample_type=sample_type)
    events = synth.make_events(50, 300, 1.5, seed=7)
    pats = [src.get_substream(s, e) for s, e in events]
    # sequential drop-in calls (small window around the true shift) ...
    for (s, e), p in list(zip(events, pats))[:5]:
        diff, t = dst.find_substream(p, s + 1.5, 1.5)
        assert abs((t - s) - 1.5) <= 1.0 / 12000 + 1e-9
    # ... and all 50 in one launch with the default +-10 s window, centre at the unshifted time
    diffs, times = dst.find_substreams(pats, [s for s, _ in events], [10] * 50)
    odst = oracle.OracleWavStream(dst.data, dst.sample_rate, dst.sample_count, dst.padding_size)
    for k, ((s, e), p) in enumerate(zip(events, pats)):
        assert abs((times[k] - s) - 1.5) <= 1.0 / 12000 + 1e-9
        if k % 10 == 0:
            rdiff, rt = odst.find_substream(p, s, 10, matcher=oracle.match_template_fft)
            assert abs(times[k] - rt) <= 1.0 / 12000 + 1e-12
            if sample_type == "uint8":
                assert np.float32(diffs[k]) == np.float32(rdiff) and times[k] == rt
            else:
                assert _score_ok(diffs[k], rdiff), (diffs[k], rdiff)


def test_full_size_windows_properties(oracle, hip_path):
    """BASELINE config 1 sizes (45-min streams, +-60 s => P = 1,440,001): planted offset recovered
    on every event; a sample of events checked against the FFT oracle."""
    from sushi_amd import synth
    from sushi_amd.wav import WavStream
    dst_pcm = synth.make_dst_pcm(2700, 12000, seed=1)
    off_s = 7.25
    src_pcm = synth.make_src_pcm(dst_pcm, int(off_s * 12000), seed=2)
    dst = WavStream.from_samples(dst_pcm, 12000, sample_type="float32")
    src = WavStream.from_samples(src_pcm, 12000, sample_type="float32")
    events = synth.make_events(24, 2700, 60 + off_s, seed=3)
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, off_s, 60, seed=4)
    diffs, times = dst.find_substreams(pats, centres, wins)
    odst = oracle.OracleWavStream(dst.data, dst.sample_rate, dst.sample_count, dst.padding_size)
    for k, (s, e) in enumerate(events):
        assert abs((times[k] - s) - off_s) <= 1.0 / 12000 + 1e-9
        assert 0.0 <= diffs[k] < 0.05
    for k in (0, 11, 23):
        rdiff, rt = odst.find_substream(pats[k], centres[k], wins[k], matcher=oracle.match_template_fft)
        assert abs(times[k] - rt) <= 1.0 / 12000 + 1e-12
        assert _score_ok(diffs[k], rdiff), (diffs[k], rdiff)


@pytest.mark.parametrize("variant", PATHS)
def test_known_answers_by_hand(variant):
    """The hand-worked vectors of tests/test_oracle.py::test_known_answers_by_hand, through the C ABI."""
    from math import sqrt
    cases = [
        (np.array([1, 2, 3, 4], np.float32), np.array([2, 3], np.float32), 1, 0.0),
        (np.array([3, 0, 4, 0, 3], np.float32), np.array([0, 5], np.float32), 1, 1.0 / 20.0),
        (np.array([10, 20, 30, 40, 50], np.uint8), np.array([20, 30, 40], np.uint8), 1, 0.0),
        # windows [50 40 30]: 2000 / sqrt(5000 * 1400) = 0.756; [40 30 20]: 1100 / sqrt(2900 * 1400) = 0.546;
        # [30 20 10]: 800 / 1400 = 0.571
        (np.array([50, 40, 30, 20, 10], np.uint8), np.array([10, 20, 30], np.uint8), 1,
         1100.0 / sqrt(2900.0 * 1400.0)),
    ]
    for img, t, want_idx, want_score in cases:
        idx, score = _run_batch(img, t, [0], [len(t)], [0], [len(img) - len(t) + 1], variant)
        assert int(idx[0]) == want_idx
        assert abs(float(score[0]) - want_score) <= 6e-8 + 1e-7 * want_score


def test_c_abi_rejects_bad_arguments_with_real_device_pointers():
    """Error behaviour of the boundary (include/sushi_hip.h): negative codes, nothing launched, nothing thrown."""
    import ctypes
    import torch
    from sushi_amd import _native
    from sushi_amd.device import DeviceStream, SearchBatch
    L = _native.lib()
    rng = np.random.default_rng(0)
    d = DeviceStream(rng.random(50000, dtype=np.float32))
    plain = DeviceStream(rng.random(50000, dtype=np.float32))              # never made searchable
    b = SearchBatch(d, d, [100], [5000], [0], [30000], path="fft")
    out_i, out_s = b.out_idx.data_ptr(), b.out_score.data_ptr()
    assert L.sushi_hip_batch_run(b.handle, 2e-5, out_i, out_s, None) == 0
    assert L.sushi_hip_batch_run(b.handle, 0.0, out_i, out_s, None) == -1 and \
        L.sushi_hip_batch_run(b.handle, 2.0, out_i, out_s, None) == -1        # SUSHI_HIP_EINVAL
    assert L.sushi_hip_batch_run(b.handle, 2e-5, None, out_s, None) == -1
    req = b.requests.copy()
    need = L.sushi_hip_batch_bytes(req.ctypes.data, 1, _native.PATH_FFT, -1, 0)
    mem = torch.empty(need + 512, dtype=torch.uint8, device="cuda")
    h = ctypes.c_void_p()

    def create(dst=d.handle, src=d.handle, r=req, ptr=mem.data_ptr(), nbytes=need, path=_native.PATH_FFT):
        return L.sushi_hip_batch_create(dst, src, r.ctypes.data, 1, path, -1, 0, ptr, nbytes, None, ctypes.byref(h))
    assert create(ptr=mem.data_ptr() + 64) == -2                                      # SUSHI_HIP_EALIGN
    assert create(nbytes=4096) == -4                                                  # SUSHI_HIP_ENOSPACE
    assert create(dst=plain.handle) == -1                                             # no spectra: not searchable
    assert create(path=7) == -1
    far = req.copy(); far["n_pos"][0] = 46000                                          # window runs past the stream
    assert L.sushi_hip_batch_bytes(far.ctypes.data, 1, _native.PATH_FFT, -1, 0) > 0 and create(r=far, nbytes=need + 512) in (-1, -4)
    u8 = DeviceStream(rng.integers(0, 255, 60000, dtype=np.uint8))
    assert create(src=u8.handle) == -1                                                # mixed sample types
    assert create() == 0
    L.sushi_hip_batch_destroy(h)
    torch.cuda.synchronize()
    idx, score = b.results()                                                         # the one valid run's result is intact
    assert 0 <= int(idx[0]) < 30000 and 0.0 <= float(score[0]) <= 1.0
    # stream creation: misaligned / short buffers
    raw = torch.rand(5000, device="cuda")
    sb = L.sushi_hip_stream_bytes(5000, _native.F32, 1)
    smem = torch.empty(sb + 512, dtype=torch.uint8, device="cuda")
    sh = ctypes.c_void_p()
    assert L.sushi_hip_stream_create(raw.data_ptr(), _native.F32, 5000, 1, smem.data_ptr() + 16, sb, None, ctypes.byref(sh)) == -2
    assert L.sushi_hip_stream_create(raw.data_ptr(), _native.F32, 5000, 1, smem.data_ptr(), sb - 256, None, ctypes.byref(sh)) == -4
    assert L.sushi_hip_stream_create(raw.data_ptr(), _native.F32, 5000, 1, smem.data_ptr(), sb, None, ctypes.byref(sh)) == 0
    L.sushi_hip_stream_destroy(sh)
    torch.cuda.synchronize()
    # the host layer turns what it can detect before any launch into SushiError
    from sushi_amd.common import SushiError
    with pytest.raises(SushiError):
        SearchBatch(d, d, [100], [5000], [0], [46000], path="fft")                   # window runs past the stream
    with pytest.raises(SushiError):
        SearchBatch(d, DeviceStream(rng.integers(0, 255, 100, dtype=np.uint8)), [0], [10], [0], [10])   # mixed sample types


# ----------------------------------------------------------------------------------------------
# FFT path specifics
# ----------------------------------------------------------------------------------------------

def test_fft_spectra_match_numpy():
    """Block spectra of a searchable stream: block j = DFT_N(xc[jB .. jB+N) + i*xc[jB+H .. jB+H+N)), xc = x - mean(x),
    H = N - B, zeros past the end, one all-zero block behind the last; stored as packed halves times one power of two
    per stream."""
    from sushi_amd import _native
    from sushi_amd.device import DeviceStream
    rng = np.random.default_rng(3)
    n = 5 * 4096 + 1234
    x = rng.random(n, dtype=np.float32)
    d = DeviceStream(x)
    L = _native.lib()
    N, B = L.sushi_hip_fft_size(), L.sushi_hip_fft_block()
    H = N - B
    halves = d.spectra().cpu().numpy().astype(np.float32).reshape(-1, N, 2)
    spec = (halves[..., 0] + 1j * halves[..., 1]).astype(np.complex64)
    slot = np.array([L.sushi_hip_fft_slot_of_bin(f) for f in range(N)])
    assert sorted(slot.tolist()) == list(range(N))            # spectra are stored in the inverse transform's load order
    spec = spec[:, slot]                                      # -> natural bin order
    assert spec.shape[0] == 6 + 1                           # ceil(n / B) blocks + the all-zero block
    assert not spec[6].any()
    xc = np.zeros(16 * N, np.float64)
    xc[:n] = x.astype(np.float64) - np.float64(np.float32(x.astype(np.float64).mean()))   # centred by the stream's mean; zeros past the end
    refs = [np.fft.fft(xc[j * B:j * B + N] + 1j * xc[j * B + H:j * B + H + N]) for j in range(spec.shape[0] - 1)]
    scale = 2.0 ** np.round(np.log2(np.abs(spec[0]).max() / np.abs(refs[0]).max()))   # the stream's power of two
    assert 32 <= np.abs(spec[:-1]).max() <= 32768 * 1.001              # inside the format's range, five binary orders and more above its subnormals
    for j, ref in enumerate(refs):
        d_re, d_im = np.abs(spec[j].real / scale - ref.real), np.abs(spec[j].imag / scale - ref.imag)
        # every component to half precision (11 bits) of its own size, plus the float32 transform's error
        assert (d_re <= 2.0 ** -11 * np.abs(ref.real) + 2e-6 * np.abs(ref).max()).all(), j
        assert (d_im <= 2.0 ** -11 * np.abs(ref.imag) + 2e-6 * np.abs(ref).max()).all(), j


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_fft_every_segment_count_class_and_chunked_patterns(oracle, dtype):
    """Patterns of 1 .. 74 segments in one batch: every class of mac_kernel (up to 6 / 12 / 18 segments) and of
    mac_long_kernel (24 / 30), and patterns beyond 30 segments, which take several accumulating passes; mixed in one
    batch so that waves hold searches of different lengths.  All identical to the oracle."""
    rng = np.random.default_rng(29)
    n_dst, n_src = 700000, 400000
    if dtype == np.uint8:
        dst = rng.integers(0, 256, n_dst, dtype=np.uint8)
        src = rng.integers(0, 256, n_src, dtype=np.uint8)
    else:
        dst = (rng.standard_normal(n_dst) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
        src = (rng.standard_normal(n_src) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
    src[5000:5000 + 160000] = dst[300000:460000]                # a 160,000-sample (40-segment) planted copy
    lens = [1, 4096, 4097, 24576, 24577, 49152, 73728, 73729, 98304, 98305, 122880, 122881, 147456, 147457, 160000,
            300000, 90000, 140000]
    offs = [7, 100, 5000, 5000, 200, 5000, 5000, 300, 5000, 5000, 5000, 400, 5000, 5000, 5000, 60000, 250000, 5000]
    wst = [1000, 250000, 290000, 200000, 0, 250000, 100000, 5000, 250000, 280000, 200000, 30000, 290000, 250000, 200000,
           350000, 123456, 295000]
    npos = [50000, 100001, 20000, 150000, 60001, 100000, 300001, 44444, 80000, 30001, 150000, 70001, 20001, 100000, 200001,
            50001, 77777, 10001]
    assert all(w + p + m - 1 <= n_dst and o + m <= n_src for w, p, m, o in zip(wst, npos, lens, offs))
    (idx, score), b = _run_batch(dst, src, offs, lens, wst, npos, "fft", want_batch=True)
    chk = _check_u8 if dtype == np.uint8 else _check_f32
    for k in range(len(offs)):
        res = oracle.match_template(dst[wst[k]:wst[k] + npos[k] + lens[k] - 1], src[offs[k]:offs[k] + lens[k]])[0]
        chk(res, idx[k], score[k])
    assert wst[14] + idx[14] == 300000 and wst[8] + idx[8] == 300000       # the planted copy, whole and in part
    # ... and by the ranking stage itself, not by the exact fall-back: no search may have missed its error bound.  (Until round 6
    # the accumulating passes of patterns beyond 30 segments added the row's FIRST word to all four bins of an entry -- a hipcc
    # miscompile of `bit_cast<half2>(entry[k])`, sushi_fft.hip add_abs2_entry -- and the bound's violation sent those searches to
    # exact evaluation at every position: right results, thousands of times the work.)
    d = b.diagnostics()
    assert d["all_positions"] == 0 and d["max_bound_ratio"] < 1.0 and d["max_bound_ratio_noncandidate"] < 1.0, d
    # one search per sub-batch: the same bits
    (idx2, score2), _ = _run_batch(dst, src, offs, lens, wst, npos, "fft", want_batch=True, workspace_bytes=1)
    assert (idx2 == idx).all() and (score2.view(np.uint32) == score.view(np.uint32)).all()


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_fft_long_template_stream_end_and_subbatches(oracle, dtype):
    """Templates longer than 16 segments (chunked accumulate), windows that run into the end of the
    stream (zero blocks), and a workspace so small that the batch is split into many sub-batches:
    all identical to the oracle / to the one-sub-batch run."""
    rng = np.random.default_rng(17)
    n_dst, n_src = 400000, 120000
    if dtype == np.uint8:
        dst = rng.integers(0, 256, n_dst, dtype=np.uint8)
        src = rng.integers(0, 256, n_src, dtype=np.uint8)
    else:
        dst = (rng.standard_normal(n_dst) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
        src = (rng.standard_normal(n_src) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
    src[1000:1000 + 70000] = dst[250000:320000]                 # a 70,000-sample (18-segment) planted copy
    offs = [1000, 1000, 500, 90000, 3, 40000]
    lens = [70000, 70000, 66000, 25000, 4097, 8192]
    wst = [200000, 0, 100000, 360000, 390000, 395000 - 8192]
    npos = [n_dst - 200000 - 70000 + 1, 300001, 150000, n_dst - 360000 - 25000 + 1, n_dst - 390000 - 4097 + 1, 5001]
    (idx, score), b = _run_batch(dst, src, offs, lens, wst, npos, "fft", want_batch=True)
    assert idx[0] == 50000 and idx[1] == 250000
    chk = _check_u8 if dtype == np.uint8 else _check_f32
    for k in range(len(offs)):
        res = oracle.match_template(dst[wst[k]:wst[k] + npos[k] + lens[k] - 1], src[offs[k]:offs[k] + lens[k]])[0]
        chk(res, idx[k], score[k])
    # the same batch through the smallest workspace the library accepts -> one search per sub-batch
    (idx2, score2), b2 = _run_batch(dst, src, offs, lens, wst, npos, "fft", want_batch=True, workspace_bytes=1)
    assert b2.ws_bytes < b.ws_bytes
    assert (idx2 == idx).all() and (score2.view(np.uint32) == score.view(np.uint32)).all()
    # and the direct kernel agrees bit for bit on uint8 (exact integers on both paths)
    idx3, score3 = _run_batch(dst, src, offs, lens, wst, npos, 2)
    if dtype == np.uint8:
        assert (idx3 == idx).all() and (score3.view(np.uint32) == score.view(np.uint32)).all()


def test_fft_near_ties_fall_back_to_direct(oracle):
    """Smooth / periodic streams put many positions within `delta` of the minimum: the refinement
    flags the search, the collection pass lists them per tile and they are evaluated exactly -- first index of the
    exact minimum, the same float32 score the direct kernel gives."""
    t = np.arange(60000, dtype=np.float64)
    img = (0.5 + 0.3 * np.sin(2 * np.pi * t / 5000.0)).astype(np.float32)       # very smooth
    tpl = img[20000:26000].copy()
    (idx, score), b = _run_batch(img, tpl, [0], [6000], [0], [54001], "fft", want_batch=True)
    res = oracle.match_template(img, tpl)[0]
    _check_f32(res, idx[0], score[0])
    d = b.diagnostics()
    assert d["flagged"] == 1 and d["all_positions"] == 0 and d["tiles_sparse"] + d["tiles_dense"] >= 1
    idx_d, score_d = _run_batch(img, tpl, [0], [6000], [0], [54001], 2)
    assert idx_d[0] == idx[0] and np.float32(score_d[0]) == np.float32(score[0])


def _planted_config(seconds, rate, window, n_events, off_s, seed, n_oracle, oracle, min_len=1.0, max_len=5.0,
                    sample_type="float32"):
    """Shared body of the full-size configuration tests: planted offset recovered on every event
    (size-independent property), events checked against the FFT oracle (uint8, the reference's default sample type
    sushi.py:769: index and float32 bits equal), both library paths agree."""
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream
    dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
    src_pcm = synth.make_src_pcm(dst_pcm, int(off_s * rate), seed=seed + 1)
    dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=sample_type)
    src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=sample_type)
    del dst_pcm, src_pcm
    u8 = sample_type == "uint8"
    events = synth.make_events(n_events, seconds, window + off_s, seed=seed + 2, min_len=min_len, max_len=max_len)
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, off_s, window, seed=seed + 3)
    diffs, times, positions = dst.find_substreams(pats, centres, wins, with_index=True)        # FFT path
    for k, (s, e) in enumerate(events):
        assert abs((times[k] - s) - off_s) <= 1.0 / rate + 1e-9, (k, times[k] - s)
        assert 0.0 <= diffs[k] < 0.05
    odst = oracle.OracleWavStream(dst.data, dst.sample_rate, dst.sample_count, dst.padding_size)
    for k in list(range(n_events))[:: max(1, n_events // n_oracle)][:n_oracle]:
        rdiff, rt = odst.find_substream(pats[k], centres[k], wins[k], matcher=oracle.match_template_fft)
        if u8:
            assert times[k] == rt and np.float32(diffs[k]).view(np.uint32) == np.float32(rdiff).view(np.uint32), (k, diffs[k], rdiff)
        else:
            assert abs(times[k] - rt) <= 1.0 / rate + 1e-12
            assert _score_ok(diffs[k], rdiff), (diffs[k], rdiff)
    # the direct kernel on a couple of the same searches: same position, same float32 score
    sel = [0, n_events - 1]
    offs = [src._get_sample_for_time(events[k][0]) for k in sel]
    lens = [pats[k].shape[1] for k in sel]
    wst, npos = [], []
    for k in sel:
        _, lo, p = dst._window(pats[k].shape[1], centres[k], wins[k])
        wst.append(lo); npos.append(p)
    b = SearchBatch(dst.device_stream(), src.device_stream(), offs, lens, wst, npos, path="direct")
    b.run()
    idx_d, score_d = b.results()
    for j, k in enumerate(sel):
        assert wst[j] + int(idx_d[j]) == positions[k]
        assert abs(float(score_d[j]) - float(diffs[k])) <= (0.0 if u8 else 2.5e-7)


@pytest.mark.parametrize("sample_type", ["float32", "uint8"])
def test_config3_sizes_two_hour_streams_120s_window(oracle, sample_type):
    """BASELINE configs[2] sizes: 2-h 12 kHz streams, +-120 s (P = 2,880,001); one rank's worth of events
    is 375 -- 48 here (float32; 24 for uint8), every one of them compared with the oracle (0.4 s of CPU each)."""
    n = 48 if sample_type == "float32" else 24
    _planted_config(7200, 12000, 120, n, 11.5, seed=31, n_oracle=n, oracle=oracle, sample_type=sample_type)


@pytest.mark.parametrize("sample_type", ["float32", "uint8"])
def test_config5_sizes_24khz_four_hour_streams(oracle, sample_type):
    """BASELINE configs[4] sizes: 4-h 24 kHz streams (346 M samples, 5.5 GB of block spectra), +-120 s
    (P = 5,760,001), templates up to 5 s = 120,000 samples = 30 segments (mac_long_kernel)."""
    _planted_config(14400, 24000, 120, 8, -17.25 + 40.0, seed=41, n_oracle=8, oracle=oracle, min_len=3.0, max_len=5.0,
                    sample_type=sample_type)


@pytest.mark.parametrize("sample_type", ["float32", "uint8"])
def test_config3_sizes_hard_material_through_the_tile_kernels(oracle, sample_type):
    """BASELINE configs[2] sizes with what bench.py --hard-frac plants: a 2-h stream with digital silence, a held 400 Hz
    tone and a recurring jingle every minute, +-120 s windows.  Events cut from there are tie-saturated (hundreds of
    thousands of positions inside any margin: a +-120 s window holds four silences and four tones): they go through
    collect_kernel + exact_tiles_kernel, and every one of them must give the oracle's FIRST index and its float32 score."""
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream
    rate, seconds, off, window = 12000, 7200.0, 7.25, 120.0
    dst_pcm, spans = synth.make_hard_dst_pcm(seconds, rate, seed=91)
    src_pcm = synth.make_src_pcm(dst_pcm, int(off * rate), seed=92)
    dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=sample_type)
    src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=sample_type)
    del dst_pcm, src_pcm
    events = synth.make_events(72, seconds, window + off, seed=93)
    events, hard = synth.plant_hard_events(events, spans, off, 0.6, seed=94)
    assert hard.sum() >= 40                                   # silence and tone events are tie-saturated, jingles are not
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, off, window, seed=95)
    offs = [src._get_sample_for_time(s) for s, _ in events]
    lens = [p.shape[1] for p in pats]
    wst, npos = [], []
    for m, c, w in zip(lens, centres, wins):
        _, lo, p = dst._window(m, c, w)
        wst.append(lo); npos.append(p)
    assert min(npos) >= 2_000_000
    b = SearchBatch(dst.device_stream(), src.device_stream(), offs, lens, wst, npos, path="fft",
                    workspace_bytes=64 << 30)
    b.run()
    idx, score = b.results()
    d = b.diagnostics(per_search=True)
    assert d["flagged"] >= 16 and d["all_positions"] == 0 and d["tiles_dense"] > 0 and d["tiles_sparse"] > 0
    assert d["max_bound_ratio"] < 1.0 and d["max_bound_ratio_noncandidate"] < 1.0
    chk = _check_u8 if sample_type == "uint8" else _check_f32
    for k in range(len(events)):
        res = oracle.match_template_fft(dst.data[:, wst[k]:wst[k] + npos[k] + lens[k] - 1],
                                        src.data[:, offs[k]:offs[k] + lens[k]])[0]
        chk(res, idx[k], score[k])
        if not hard[k]:
            assert abs((wst[k] + int(idx[k])) - (offs[k] + int(off * rate))) <= 1


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_fft_streams_far_from_the_centring_constant(oracle, dtype):
    """Streams that sit far from the centring constant the exact stages use (uint8 samples 0..6, float32
    samples ~0.19).  The FFT stage centres the destination by the stream's own mean and scales the products by the
    stream's own energy, so its ranking error stays small whatever level the data sits at: no fallback, exact results."""
    rng = np.random.default_rng(23)
    n = 60000
    if dtype == np.uint8:
        dst = rng.integers(0, 7, n, dtype=np.uint8)
    else:
        dst = (np.float32(0.18) + rng.random(n, dtype=np.float32) * np.float32(0.02)).astype(np.float32)
    src = dst[30000:30000 + 5000].copy()
    src[::3] = dst[100:100 + 5000][::3]                     # a noisy copy: minimum well above 0
    (idx, score), b = _run_batch(dst, src, [0, 0], [5000, 2500], [1000, 20000], [50001, 20001], "fft", want_batch=True)
    assert b.fallback_count() == 0
    assert b.ranking_errors().max() < b.delta
    assert b.diagnostics()["max_bound_ratio"] < 0.5        # measured f32 error against the modelled bound
    assert b.diagnostics()["max_bound_ratio_noncandidate"] < 0.5   # the same at positions that were not candidates
    for k, (m, w, p) in enumerate([(5000, 1000, 50001), (2500, 20000, 20001)]):
        res = oracle.match_template(dst[w:w + p + m - 1], src[:m])[0]
        (_check_u8 if dtype == np.uint8 else _check_f32)(res, idx[k], score[k])
    assert idx[0] == 29000 and idx[1] == 10000


def test_fft_ranking_error_is_far_below_delta():
    """The measured |f32 FFT score - exact score| at the result positions of a BASELINE-configs[1]-shaped
    batch stays below the floor `delta` of the margin the exact re-evaluation works with (the margin itself is the
    modelled bound of each pair, checked from both sides by `max_bound_ratio*`)."""
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream
    for sample_type in ("float32", "uint8"):
        dst_pcm = synth.make_dst_pcm(600, 12000, seed=61)
        src_pcm = synth.make_src_pcm(dst_pcm, 4321, seed=62)
        dst = WavStream.from_samples(dst_pcm, 12000, sample_type=sample_type)
        src = WavStream.from_samples(src_pcm, 12000, sample_type=sample_type)
        events = synth.make_events(64, 600, 61, seed=63)
        offs = [src._get_sample_for_time(s) for s, _ in events]
        lens = [src._get_sample_for_time(e) - src._get_sample_for_time(s) for s, e in events]
        wst, npos = [], []
        for (s, e), m in zip(events, lens):
            _, lo, p = dst._window(m, s + 4321 / 12000.0, 60)
            wst.append(lo); npos.append(p)
        b = SearchBatch(dst.device_stream(), src.device_stream(), offs, lens, wst, npos, path="fft")
        b.run()
        err = b.ranking_errors()
        assert b.fallback_count() == 0
        assert err.max() < b.delta, err.max()                 # (spectra and products are kept as halves: ~1e-5 of quantisation noise, modelled per pair)
        dg = b.diagnostics()
        assert 0.0 < dg["max_bound_ratio_noncandidate"] < 0.5      # up to 16 audited non-candidate positions per search, all far inside the bound
        assert 64 * 12 <= dg["audited"] <= 64 * 16, dg["audited"]   # (runs cut by a window's edge give fewer)


@pytest.mark.parametrize("variant", [2, "fft"])
def test_no_match_anywhere_ties_at_one(oracle, variant):
    """A destination that is digital silence followed by a nearly silent passage: every window scores >= 1
    before cv2's clamp, so the result row is all 1.0 and argmin must return index 0 (wav.py:186).  The FFT
    stage has to clamp too, or it would rank the unclamped values and hand back a later index."""
    rng = np.random.default_rng(77)
    dst = np.concatenate((np.zeros(3000, np.float32), (rng.random(30000, dtype=np.float32) * np.float32(1e-3))))
    src = (np.float32(0.3) + rng.random(4000, dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    res = oracle.match_template(dst[:28000 + 3999], src)[0]
    assert res.min() == 1.0 and int(res.argmin()) == 0
    (idx, score), b = _run_batch(dst, src, [0], [4000], [0], [28000], variant, want_batch=True)
    assert idx[0] == 0 and score[0] == 1.0


from hypothesis import given, settings, strategies as st


@settings(max_examples=100, deadline=None, derandomize=True)   # the same examples on every run; tools/prop_hunt.py is the random hunt
@given(seed=st.integers(0, 2 ** 31 - 1), L=st.integers(1, 30000), frac=st.floats(0.0, 1.0), u8=st.booleans(),
       path=st.sampled_from([0, 2, "fft"]), scale=st.sampled_from([1.0, 1e-3, 1e-6, 40.0]))
def test_random_shapes_property(seed, L, frac, u8, path, scale):
    """Any (search length, pattern length, dtype, path): same arg-min and score as the oracle.
    The FFT path (the default) reads the samples as they are in every stage -- including the float64 kernel that
    finishes short patterns and tie-saturated searches of float32 streams -- so it is held to float32 data of any
    magnitude; the direct MFMA kernel accumulates float32(sample - 0.5) products, whose rounding is relative
    to sum |T - 0.5| |I - 0.5| rather than to sum T I, and is held to what WavStream produces: samples around
    the mid level 0.5 (silence maps there, wav.py:148-151), here [0.25, 0.75) (include/sushi_hip.h,
    sushi_hip_match_batch)."""
    from oracle import oracle as O
    O.build()
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
        p = int(rng.integers(0, L - M + 1))                  # plant a noisy copy somewhere
        src[1:1 + M] = dst[2 + p:2 + p + M]
        src[1 + M // 2] = dst[0]
    idx, score = _run_batch(dst, src, [1], [M], [2], [L - M + 1], path)
    res = O.match_template_direct(dst[2:2 + L], src[1:1 + M])[0]
    (_check_u8 if u8 else _check_f32)(res, idx[0], score[0])


# ----------------------------------------------------------------------------------------------
# Error bound of the ranking stage, tie-heavy material, one accumulation order
# ----------------------------------------------------------------------------------------------

def test_quiet_passage_inside_a_loud_block_matches_oracle(oracle):
    """float32 data the WavStream pipeline would never produce, through the same entry points: a passage 1e-4 times
    quieter than its surroundings, longer than the pattern.  The f32 FFT error there is relative to the LOUD block, not
    to the quiet window, so the fixed margin of round 1 would miss the true minimum; the per-pair error bound widens
    the candidate set instead (those pairs are evaluated exactly) and the result is the oracle's."""
    rng = np.random.default_rng(5)
    n = 160000
    dst = rng.standard_normal(n).astype(np.float32)
    quiet = slice(60000, 90000)
    dst[quiet] *= np.float32(1e-4)
    m = 9000
    src = dst[70000:70000 + m].copy()
    src += (rng.standard_normal(m) * 2e-6).astype(np.float32)          # a noisy copy of a stretch of the quiet passage
    src2 = (dst[65000:65000 + m] * np.float32(1.5)).astype(np.float32)  # a scaled copy: minimum > 0, also in the quiet passage
    srcs = np.concatenate((src, src2))
    (idx, score), b = _run_batch(dst, srcs, [0, m], [m, m], [1000, 20000], [140001, 100001], "fft", want_batch=True)
    d = b.diagnostics(per_search=True)
    # At this dynamic range the reference's own arithmetic is noise-limited: cv2 (and the oracle, which restates it)
    # takes the window energy from a float64 integral over the LOUD search image, whose rounding noise is ~1e-2 of the
    # quiet window's sum (T - I)^2.  So: same arg-min as the oracle, and the score held to the textbook definition
    # evaluated in long double (our float64 prefix sums are built block-wise and carry less of that noise).
    for k, (w, p, to) in enumerate([(1000, 140001, 0), (20000, 100001, m)]):
        res = oracle.match_template(dst[w:w + p + m - 1], srcs[to:to + m])[0]
        assert int(idx[k]) == int(res.argmin())
        T = srcs[to:to + m].astype(np.longdouble)
        lo = max(0, int(idx[k]) - 1500)
        best, best_p = None, None
        for q in range(lo, int(idx[k]) + 1500):
            I = dst[w + q:w + q + m].astype(np.longdouble)
            v = float(((T - I) ** 2).sum() / np.sqrt((T * T).sum() * (I * I).sum()))
            if best is None or v < best:
                best, best_p = v, q
        assert best_p == int(idx[k])
        assert abs(float(score[k]) - best) <= 2e-3 * best + 2.5e-7, (score[k], best)
    assert idx[0] == 69000 and idx[1] == 45000
    assert d["max_bound_ratio"] < 1.0                    # every evaluated candidate was inside its modelled bound ...
    assert d["max_bound_ratio_noncandidate"] < 1.0       # ... and so was one position per search that was NOT a candidate
    assert d["all_positions"] == 0


@pytest.mark.parametrize("sample_type", ["float32", "uint8"])
def test_hard_material_silence_tone_jingle_matches_oracle(oracle, sample_type):
    """Digital silence, a held tone with an exact 30-sample period and a jingle that recurs -- what real soundtracks
    hold and filtered noise does not.  Searches cut from there have hundreds to thousands of positions within any margin
    of the minimum; they go through the collection pass + exact tiles and give the oracle's first index and score."""
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream
    rate, seconds, off = 12000, 200.0, 1.75
    dst_pcm, spans = synth.make_hard_dst_pcm(seconds, rate, seed=71, period_s=40.0)
    src_pcm = synth.make_src_pcm(dst_pcm, int(off * rate), seed=72)
    dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=sample_type)
    src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=sample_type)
    events = synth.make_events(24, seconds, 30 + off, seed=73, min_len=1.0, max_len=2.0)
    events, hard = synth.plant_hard_events(events, spans, off, 0.5, seed=74)
    assert hard.sum() >= 10
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, off, 30.0, seed=75)
    offs = [src._get_sample_for_time(s) for s, _ in events]
    lens = [p.shape[1] for p in pats]
    wst, npos = [], []
    for m, c, w in zip(lens, centres, wins):
        _, lo, p = dst._window(m, c, w)
        wst.append(lo); npos.append(p)
    b = SearchBatch(dst.device_stream(), src.device_stream(), offs, lens, wst, npos, path="fft")
    b.run()
    idx, score = b.results()
    d = b.diagnostics(per_search=True)
    assert d["flagged"] >= 4 and d["all_positions"] == 0          # tone / silence searches overflow the lists
    assert d["tiles_sparse"] > 0
    for k in range(len(events)):
        res = oracle.match_template(dst.data[:, wst[k]:wst[k] + npos[k] + lens[k] - 1], src.data[:, offs[k]:offs[k] + lens[k]])[0]
        (_check_u8 if sample_type == "uint8" else _check_f32)(res, idx[k], score[k])
    # a flagged search costs what its candidates cost, not what its window costs: far fewer exact positions than P
    exact_positions = d["candidates"] + 1024 * d["tiles_dense"]
    assert exact_positions < 0.2 * sum(npos[k] for k in range(len(events)) if d["flagged_per_search"][k])


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_runs_of_equal_samples_inside_at_the_ends_and_beyond_the_shortcut(oracle, dtype):
    """Patterns cut from runs of ONE value (digital silence) tie over the whole run: dense tiles whose chunks lie inside a run
    take the run's chunk chains instead of their positions' own sums (exact_tiles_kernel) -- same bits.  Here: tiles inside a
    run and at both of its ends (the run starts and ends off the tile grid), a pattern of more chunks than the shortcut looks at
    (every position evaluated, as before), and a run that reaches the end of the stream (tiles whose windows are cut there)."""
    rng = np.random.default_rng(77)
    n = 600000
    if dtype == np.uint8:
        dst = rng.integers(0, 256, n, dtype=np.uint8)
        c = np.uint8(128)
    else:
        dst = rng.random(n, dtype=np.float32)
        c = np.float32(0.5)
    dst[100000:300000] = c
    dst[560000:] = c
    cuts = [(120000, 150000), (150000, 40000), (565000, 20000)]          # (first sample in dst, length)
    offs, lens, parts = [], [], []
    at = 0
    for a, m in cuts:
        t = dst[a:a + m].copy()
        if dtype == np.uint8:
            k = rng.random(m) < 0.1
            t[k] = t[k] + rng.integers(-1, 2, int(k.sum())).astype(np.uint8)
        else:
            t += (rng.standard_normal(m) * 0.01).astype(np.float32)
        offs.append(at); lens.append(m); parts.append(t)
        at += m
    src = np.concatenate(parts)
    wst = [60000, 60000, 500000]
    npos = [200000, 250000, n - 500000 - 20000 + 1]
    (idx, score), b = _run_batch(dst, src, offs, lens, wst, npos, "fft", want_batch=True)
    d = b.diagnostics()
    assert d["flagged"] == 3 and d["all_positions"] == 0 and d["tiles_dense"] >= 100
    for k in range(3):
        res = oracle.match_template(dst[wst[k]:wst[k] + npos[k] + lens[k] - 1], src[offs[k]:offs[k] + lens[k]])[0]
        (_check_u8 if dtype == np.uint8 else _check_f32)(res, idx[k], score[k])
        # tens of thousands of positions score what the winner scores (to 2e-6): the ties of the run
        assert int((res <= res[int(idx[k])] + 2e-6).sum()) >= (lens[k] == 150000 and 50001 or lens[k] == 40000 and 160001 or 20001)


def test_one_accumulation_order_whatever_route_finds_the_position():
    """The exact value of a position is the same float32, bit for bit, whether the candidate lists (refine_kernel), a
    sparse tile or a dense tile (exact_tiles_kernel) evaluated it: delta = 1 makes every position a candidate and sends
    the same searches through the tiles."""
    from sushi_amd.device import DeviceStream, SearchBatch
    rng = np.random.default_rng(8)
    dst = rng.random(300000, dtype=np.float32)
    src = dst[40000:140000].copy()
    src += (rng.standard_normal(src.shape[0]) * 0.02).astype(np.float32)
    d, s = DeviceStream(dst), DeviceStream(src)
    offs, lens = [1000, 30000, 60000, 5], [30000, 4097, 12000, 700]
    wst, npos = [0, 20000, 50000, 39000], [200001, 100000, 150000, 3000]
    a = SearchBatch(d, s, offs, lens, wst, npos, path="fft")
    a.run()
    ia, sa = a.results()
    assert a.diagnostics()["flagged"] == 0
    bb = SearchBatch(d, s, offs, lens, wst, npos, path="fft", delta=1.0)
    bb.run()
    ib, sb = bb.results()
    dg = bb.diagnostics()
    assert dg["flagged"] == 4 and dg["tiles_dense"] > 100
    assert (ia == ib).all() and (sa.view(np.uint32) == sb.view(np.uint32)).all()
    assert list(ia) == [41000 - 0, 70000 - 20000, 100000 - 50000, 40005 - 39000]


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("M", [700, 3000, 7680])
def test_short_pattern_with_dozens_of_near_tie_candidates(oracle, dtype, M):
    """ADVICE r4 (refine_body): a pattern of at most 15 chunks of 512 samples gets more tasks per round than the round's
    finishing step has thread groups for.  A periodic stream gives ~39 candidates spread over several block pairs (at most 8
    per pair: none overflows), every copy but a LATE one off in one sample: the exact minimum is that late
    copy, and every candidate has to be finished for it to be found."""
    rng = np.random.default_rng(4242 + M)
    period, reps, best = 9000, 40, 30                 # longer than every pattern here: a window meets one copy's quantum only
    if dtype == np.uint8:
        base = rng.integers(1, 239, period, dtype=np.uint8)
    else:
        base = (0.1 + 0.7 * rng.random(period, dtype=np.float32)).astype(np.float32)
    dst = np.tile(base, reps)
    a = 500
    for k in range(reps):
        if k != best:
            j = k * period + a + (37 * k) % M
            # enough to clear the float32 quantum of cv2's stored cross term (2.4e-7 in score; for uint8 streams sum T*I ~ M * 128^2
            # has an ulp of up to 32) and stay inside the candidate margin (2e-5): a score of ~1e-6 .. 4e-6
            dst[j] = dst[j] + (16 if dtype == np.uint8 else np.float32(np.sqrt(1.2e-6 * M)))
    tpl = np.tile(base, 3)[a:a + M].copy()
    P = dst.shape[0] - M + 1
    (idx, score), b = _run_batch(dst, tpl, [0], [M], [0], [P], "fft", want_batch=True)
    res = oracle.match_template(dst, tpl)[0]
    assert int(res.argmin()) == best * period + a
    (_check_u8 if dtype == np.uint8 else _check_f32)(res, idx[0], score[0])
    assert int(idx[0]) == best * period + a
    d = b.diagnostics()
    assert d["flagged"] == 0 and d["all_positions"] == 0


def test_packed_output_is_the_same_results_as_8_byte_records():
    """sushi_hip_batch_set_packed_output: what a rank hands to the all-gather -- (index, score bits) per search -- written by the
    kernel that writes out_idx / out_score, on both paths."""
    import torch
    from sushi_amd.device import DeviceStream, SearchBatch
    rng = np.random.default_rng(9)
    dst = rng.random(60000, dtype=np.float32)
    src = dst[1000:30000].copy()
    for path in ("fft", "direct"):
        b = SearchBatch(DeviceStream(dst), DeviceStream(src), [0, 5000, 9000], [3000, 700, 4097], [500, 2000, 8000], [20000, 30000, 9000],
                        path=path)
        packed = torch.full((5, 2), -7, dtype=torch.int32, device="cuda")
        b.set_packed_output(packed)
        b.run()
        idx, score = b.results()
        p = packed.cpu().numpy()
        assert (p[:3, 0] == idx).all() and (p[:3, 1].view(np.float32) == score).all() and (p[3:] == -7).all()
        assert (idx == [500, 4000, 2000]).all()
        b.set_packed_output(None)
        packed.fill_(-7)
        b.run()
        b.results()
        assert (packed.cpu().numpy() == -7).all()


def test_early_records_hold_the_answer_or_say_flagged():
    """sushi_hip_batch_set_early_output: the kernel that finishes a search from its candidate lists writes the answer as one 16-byte
    record (index, score bits, ready, flagged) into memory the host can poll -- the very values out_idx / out_score get from the
    run's last kernel; a search that goes on to the tile stage says so instead.  A batch of up to four searches (what a drop-in
    find_substream call is) has such records by itself and results() reads them."""
    import torch
    from sushi_amd.device import DeviceStream, SearchBatch
    rng = np.random.default_rng(5)
    t = np.arange(120000, dtype=np.float64)
    smooth = (0.5 + 0.3 * np.sin(2 * np.pi * t / 5000.0)).astype(np.float32)       # ties everywhere: flagged
    noisy = (rng.standard_normal(120000) * 0.2 + 0.5).clip(0, 1).astype(np.float32)
    img = np.concatenate([smooth, noisy])
    offs, lens, wst, npos = [20000, 150000, 200000], [6000, 9000, 4097], [0, 130000, 170000], [54001, 60000, 50001]
    for method in ("sqdiff_normed", "ccoeff_normed"):
        b = SearchBatch(DeviceStream(img), DeviceStream(img), offs, lens, wst, npos, path="fft", method=method)
        assert b._early is not None
        for r in range(3):
            b.run()
            idx, score = b.results()                      # (one search flagged: falls through to the stream's end)
            torch.cuda.synchronize()
            e = b._early.numpy()
            assert (e[:, 2] == 1).all() and list(e[:, 3]) == [1, 0, 0], e
            oi, osc = b.out_idx.cpu().numpy(), b.out_score.cpu().numpy()
            assert (idx == oi).all() and (score.view(np.uint32) == osc.view(np.uint32)).all()
            assert (e[1:, 0] == oi[1:]).all() and (e[1:, 1].view(np.uint32) == osc[1:].view(np.uint32)).all()
            assert wst[1] + oi[1] == offs[1] and wst[2] + oi[2] == offs[2]
        # without the flagged search results() IS the early records (no stream synchronisation: the records are polled)
        b2 = SearchBatch(DeviceStream(img), DeviceStream(img), offs[1:], lens[1:], wst[1:], npos[1:], path="fft", method=method)
        b2.run()
        idx2, score2 = b2.results()
        assert (idx2 == oi[1:]).all() and (score2.view(np.uint32) == osc[1:].view(np.uint32)).all()
    # a flat pattern under TM_CCOEFF_NORMED answers (0, 1.0) from refine_kernel's own early exit
    flat = img.copy(); flat[1000:6000] = 0.5
    b3 = SearchBatch(DeviceStream(flat), DeviceStream(flat), [1000], [5000], [0], [30001], path="fft", method="ccoeff_normed")
    b3.run()
    idx3, score3 = b3.results()
    assert idx3[0] == 0 and score3[0] == 1.0 and list(b3._early.numpy()[0]) == [0, np.float32(1.0).view(np.int32), 1, 0]
