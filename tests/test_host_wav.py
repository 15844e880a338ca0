"""Host side of the drop-in WavStream (no GPU): load pipeline against the oracle's statement-for-
statement restatement of wav.py:108-162, window arithmetic against the reference-generated golden
vectors, error behaviour."""
import os
import struct

import numpy as np
import pytest

from sushi_amd import SushiError, synth
from sushi_amd.wav import DownmixedWavFile, WavStream, _locate


def _write_wav24(path, pcm24, rate, channels):
    b = bytearray()
    for v in np.asarray(pcm24).reshape(-1):
        b += struct.pack('<i', int(v))[:3]
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<L', 36 + len(b)) + b'WAVE')
        f.write(b'fmt ' + struct.pack('<LHHLLHH', 16, 1, channels, rate, rate * channels * 3, channels * 3, 24))
        f.write(b'junk' + struct.pack('<L', 3) + b'abc\0')           # odd-sized chunk to skip
        f.write(b'data' + struct.pack('<L', len(b)))
        f.write(bytes(b))


@pytest.mark.parametrize("rate,channels,sample_rate,seconds", [
    (12000, 1, 12000, 7.3), (12000, 2, 12000, 3.0), (48000, 1, 12000, 2.5), (44100, 2, 12000, 3.21),
    (8000, 3, 12000, 1.7)])
@pytest.mark.parametrize("sample_type", ["float32", "uint8"])
def test_load_pipeline_matches_oracle(tmp_path, oracle, rate, channels, sample_rate, seconds, sample_type):
    rng = np.random.default_rng(rate + channels)
    n = int(seconds * rate)
    pcm = (rng.standard_normal((n, channels)) * 3000).astype(np.int16)
    path = os.path.join(tmp_path, "a.wav")
    synth.write_wav(path, pcm if channels > 1 else pcm[:, 0], rate, channels)
    ours = WavStream(path, sample_rate=sample_rate, sample_type=sample_type)
    ref = oracle.load_wav_stream(path, sample_rate=sample_rate, sample_type=sample_type)
    assert ours.sample_count == ref.sample_count and ours.padding_size == ref.padding_size
    assert ours.sample_rate == ref.sample_rate and ours.duration_seconds == ref.duration_seconds
    assert ours.data.dtype == ref.data.dtype and ours.data.shape == ref.data.shape
    assert (ours.data == ref.data).all()


def test_load_24bit_and_from_samples(tmp_path, oracle):
    rng = np.random.default_rng(3)
    pcm = rng.integers(-2 ** 22, 2 ** 22, size=(24000, 2))
    path = os.path.join(tmp_path, "b.wav")
    _write_wav24(path, pcm, 12000, 2)
    w = DownmixedWavFile(path)
    assert (w.channels_count, w.framerate, w.sample_width, w.frames_count) == (2, 12000, 3, 24000)
    w.close()
    ours = WavStream(path, sample_type="float32")
    ref = oracle.load_wav_stream(path, sample_type="float32")
    assert (ours.data == ref.data).all()
    # from_samples == loading a mono 16-bit file with those samples
    mono = (rng.standard_normal(30000) * 2000).astype(np.int16)
    p2 = os.path.join(tmp_path, "c.wav")
    synth.write_wav(p2, mono, 12000)
    a = WavStream(p2, sample_type="uint8")
    b = WavStream.from_samples(mono, 12000, sample_type="uint8")
    assert (a.data == b.data).all() and a.sample_count == b.sample_count


def test_errors(tmp_path):
    with pytest.raises(SushiError):
        WavStream("nope.wav", sample_type="int16")
    with pytest.raises(IOError):                 # as in the reference: open() is outside its try block
        WavStream(os.path.join(tmp_path, "missing.wav"))
    eight = os.path.join(tmp_path, "eight.wav")   # 8-bit PCM: fails inside the loader -> wrapped (wav.py:158-159)
    with open(eight, "wb") as f:
        f.write(b"RIFF" + struct.pack("<L", 36 + 100) + b"WAVE")
        f.write(b"fmt " + struct.pack("<LHHLLHH", 16, 1, 1, 8000, 8000, 1, 8))
        f.write(b"data" + struct.pack("<L", 100) + b"\x80" * 100)
    with pytest.raises(SushiError) as e:
        WavStream(eight)
    assert "Error while loading" in str(e.value) and "Unsupported sample width" in str(e.value)
    bad = os.path.join(tmp_path, "bad.wav")
    with open(bad, "wb") as f:
        f.write(b"RIFX" + b"\0" * 40)
    with pytest.raises(SushiError) as e:
        WavStream(bad)
    assert "RIFF" in str(e.value)


def _bare_stream(c):
    s = WavStream.__new__(WavStream)
    s.sample_rate, s.sample_count, s.padding_size = c["sample_rate"], c["sample_count"], c["padding_size"]
    s.data = np.zeros((1, c["data_len"]), c["dtype"])
    return s


def test_window_arithmetic_matches_reference_golden(golden_index):
    cache = {}
    for c in golden_index["cases"]:
        key = (c["sample_rate"], c["framerate"], c["seconds"], c["dtype"])
        if key not in cache:
            cache[key] = _bare_stream(c)
        s = cache[key]
        pat = s.get_substream(c["pat_start"], c["pat_end"])
        off = (pat.__array_interface__["data"][0] - s.data.__array_interface__["data"][0]) // pat.itemsize
        assert (off, pat.shape[1]) == (c["pat_off"], c["pat_len"])
        start_time, lo, P = s._window(c["pat_len"], c["center"], c["window"])
        assert lo == c["search_off"]
        assert P == c["search_len"] - c["pat_len"] + 1
        if c["ok"]:
            assert start_time + (c["min_idx"] / float(s.sample_rate)) == c["time"]
        else:
            assert P <= 0


def test_pattern_provenance():
    rng = np.random.default_rng(0)
    a = WavStream.from_samples((rng.standard_normal(40000) * 1000).astype(np.int16), 12000, sample_type="float32")
    b = WavStream.from_samples((rng.standard_normal(40000) * 1000).astype(np.int16), 12000, sample_type="float32")
    pat = a.get_substream(0.5, 1.25)
    st, off, m = _locate(pat)
    assert st is a and off == a._get_sample_for_time(0.5) and m == pat.shape[1]
    left, right = np.split(pat, [pat.shape[1] // 2], axis=1)            # sushi.py:445
    assert _locate(left)[1:] == (off, pat.shape[1] // 2)
    assert _locate(right)[1:] == (off + pat.shape[1] // 2, pat.shape[1] - pat.shape[1] // 2)
    assert _locate(b.get_substream(0, 1))[0] is b
    assert _locate(pat.copy()) is None
    assert _locate(pat[:, ::2]) is None


def test_find_substream_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rng = np.random.default_rng(0)
    a = WavStream.from_samples((rng.standard_normal(40000) * 1000).astype(np.int16), 12000, sample_type="float32")
    with pytest.raises(SushiError):
        a.find_substream(a.get_substream(0.5, 1.0), 0.5, 1.5)


# ---- property tests (hypothesis) of the host window arithmetic against the oracle's wav.py:177-188 restatement ----
from hypothesis import given, settings, strategies as st


@settings(max_examples=300, deadline=None, derandomize=True)
@given(centre=st.floats(min_value=-40.0, max_value=140.0, allow_nan=False),
       window=st.one_of(st.floats(min_value=0.0, max_value=130.0, allow_nan=False), st.integers(0, 130)),
       m=st.integers(1, 70000), rate=st.sampled_from([8000, 12000, 24000]))
def test_window_arithmetic_matches_oracle_for_any_request(centre, window, m, rate):
    """WavStream._window (start time, first sample, result length) == the oracle's search_bounds for any
    centre / window / pattern length, including windows clipped at either end and NumPy slice truncation."""
    from oracle import oracle as O
    from sushi_amd.wav import WavStream
    n = 100 * rate
    w = WavStream.__new__(WavStream)
    w.data = np.zeros((1, n + 20 * rate), np.uint8)
    w.sample_rate, w.sample_count, w.padding_size = rate, n, 10 * rate
    o = O.OracleWavStream(w.data, rate, n, 10 * rate)
    start_time, lo, n_pos = w._window(m, centre, window)
    o_start, o_lo, o_hi = o.search_bounds(m, centre, window)
    assert start_time == o_start and lo == o_lo
    assert n_pos == max(o_hi - o_lo, 0) - m + 1


def test_format_time_reference_vectors():
    """The reference's own known answers (tests/main.py:220-235 FormatTimeTestCase) plus Python 2's rounding of
    halves (common.py:32 `round` under Python 2 rounds half away from zero: 0.005 s -> .01)."""
    from sushi_amd.common import format_time
    assert format_time(0) == '0:00:00.00'
    assert format_time(65) == '0:01:05.00'
    assert format_time(5.559) == '0:00:05.56'
    assert format_time(3600 + 60 * 15 + 35.15) == '1:15:35.15'
    assert format_time(544.997) == '0:09:05.00'
    assert format_time(0.005) == '0:00:00.01' and format_time(0.025) == '0:00:00.03'


def test_truncated_and_placeholder_headers_do_not_size_buffers(tmp_path, monkeypatch):
    """A data size that overstates the file (cut-off copy, 0xFFFFFFFF placeholder of a piped encoder): the raw-frame
    buffers are sized by what the file holds; the frames that exist are decimated as the reference does (a short last
    chunk by its own length, wav.py:127-134) and the part the reference leaves uninitialised (np.empty) is zero."""
    import struct
    from sushi_amd import synth
    from sushi_amd.wav import DownmixedWavFile, WavStream
    monkeypatch.setenv("SUSHI_HIP_LOAD", "host")
    rate = 24000
    pcm = synth.make_dst_pcm(3.5, rate, seed=3)
    full = str(tmp_path / "full.wav")
    synth.write_wav(full, pcm, rate)
    blob = open(full, "rb").read()
    cut = str(tmp_path / "cut.wav")
    keep = 44 + 2 * int(2.3 * rate)
    open(cut, "wb").write(blob[:keep])                               # header still says 3.5 s
    w = DownmixedWavFile(cut)
    assert w.frames_count == pcm.shape[0] and w.frames_available == int(2.3 * rate)
    w.close()
    s_cut = WavStream(cut, sample_rate=12000, sample_type="float32")
    s_full = WavStream(full, sample_rate=12000, sample_type="float32")
    assert s_cut.data.shape == s_full.data.shape and s_cut.sample_count == s_full.sample_count
    # (the values are whatever wav.py:143-151 makes of a stream whose tail and right padding are zeros -- here the
    # medians, and with them the scale, collapse exactly as they would in the reference)
    # a placeholder size on a small file: nothing near 4 GiB of frames is allocated (this would take minutes / fail)
    ph = str(tmp_path / "placeholder.wav")
    open(ph, "wb").write(blob[:40] + struct.pack("<L", 0xFFFFFFFF) + blob[44:44 + 2 * rate])
    w = DownmixedWavFile(ph)
    assert w.frames_count == 0xFFFFFFFF // 2 and w.frames_available == rate
    w.close()
    # an empty data chunk: an error instead of arithmetic on uninitialised memory
    empty = str(tmp_path / "empty.wav")
    open(empty, "wb").write(blob[:40] + struct.pack("<L", 0))
    with pytest.raises(SushiError):
        WavStream(empty, sample_rate=12000, sample_type="float32")
