"""The GPU load pipeline (csrc/sushi_load.hip, sushi_amd/load.py) against the NumPy pipeline
(WavStream._build_host, which tests/test_host_wav.py pins to the oracle's restatement of wav.py:108-162):
bit-identical streams for both sample types, with and without decimation, partial last chunks included."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host_stream(samples, framerate, sample_rate, sample_type):
    from sushi_amd.wav import WavStream
    w = WavStream.__new__(WavStream)
    w._build_host(np.asarray(samples, np.float32), framerate, len(samples), sample_rate, sample_type)
    return w


@pytest.mark.parametrize("sample_type", ["uint8", "float32"])
@pytest.mark.parametrize("framerate,sample_rate,seconds", [(12000, 12000, 33.37), (48000, 12000, 21.5), (44100, 12000, 12.3),
                                                           (8000, 12000, 9.25), (12000, 12000, 0.4), (48000, 24000, 7.0)])
def test_gpu_load_is_bit_identical_to_host_pipeline(sample_type, framerate, sample_rate, seconds):
    from sushi_amd import synth
    from sushi_amd.wav import WavStream
    pcm = synth.make_dst_pcm(seconds, framerate, seed=int(seconds * 100) + framerate)
    dev = WavStream.from_samples(pcm, framerate, sample_rate=sample_rate, sample_type=sample_type)
    assert dev._dev_row is not None                       # the GPU pipeline ran
    ref = _host_stream(pcm.astype(np.float32), framerate, sample_rate, sample_type)
    assert dev.data.dtype == ref.data.dtype and dev.data.shape == ref.data.shape
    assert dev.sample_count == ref.sample_count and dev.padding_size == ref.padding_size
    if sample_type == "uint8":
        assert (dev.data == ref.data).all()
    else:
        assert (dev.data.view(np.uint32) == ref.data.view(np.uint32)).all()
    # the device-resident row feeds the matcher without another upload
    ds = dev.device_stream()
    c = np.float32(128.0 if sample_type == "uint8" else 0.5)
    assert (ds.xc.cpu().numpy() == ref.data[0].astype(np.float32) - c).all()


def test_gpu_load_medians_with_zeros_and_ties():
    """Streams with many exact zeros and repeated values: the radix select picks the same order statistics
    np.median does (zeros belong to both sides, wav.py:145-146)."""
    from sushi_amd.wav import WavStream
    rng = np.random.default_rng(5)
    x = rng.integers(-50, 50, 40001).astype(np.float32) * 16.0
    x[::7] = 0.0
    for sample_type in ("float32", "uint8"):
        dev = WavStream.from_samples(x, 4000, sample_rate=4000, sample_type=sample_type)
        ref = _host_stream(x, 4000, 4000, sample_type)
        assert (dev.data == ref.data).all()


def test_host_pipeline_can_be_forced(monkeypatch):
    from sushi_amd import synth
    from sushi_amd.wav import WavStream
    monkeypatch.setenv("SUSHI_HIP_LOAD", "host")
    w = WavStream.from_samples(synth.make_dst_pcm(3, 12000, seed=1), 12000)
    assert w._dev_row is None
