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


def test_wav_load_budget_of_the_reference_regression_test(tmp_path):
    """regression-tests.py:140-158 (run_wav_test) with the thresholds of tests.example.json:24-27: loading a WAV into a
    WavStream may cost at most 0.7 s of CPU time and 120 MB of resident memory -- measured the reference's way
    (resource.getrusage before / after the constructor), on a 24-minute 48 kHz stereo 16-bit file (276 MB), sample_type
    uint8 at 12 kHz.  The file goes to the GPU in 32 MB pieces and is decoded, downmixed, decimated and normalised there."""
    import resource
    import struct
    from sushi_amd.wav import WavStream
    rate, channels, seconds = 48000, 2, 24 * 60
    path = os.path.join(str(tmp_path), "long.wav")
    rng = np.random.default_rng(3)
    second = (rng.standard_normal((rate, channels)) * 3000).astype('<i2').tobytes()
    n_bytes = len(second) * seconds
    with open(path, "wb") as f:
        f.write(b'RIFF' + struct.pack('<L', 36 + n_bytes) + b'WAVE')
        f.write(b'fmt ' + struct.pack('<LHHLLHH', 16, 1, channels, rate, rate * channels * 2, channels * 2, 16))
        f.write(b'data' + struct.pack('<L', n_bytes))
        for _ in range(seconds):
            f.write(second)
    WavStream(path, sample_rate=12000, sample_type='uint8')           # warm: HIP context, kernels, allocator pools
    before = resource.getrusage(resource.RUSAGE_SELF)
    s = WavStream(path, sample_rate=12000, sample_type='uint8')
    after = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (after.ru_utime + after.ru_stime) - (before.ru_utime + before.ru_stime)
    rss_mb = (after.ru_maxrss - before.ru_maxrss) / 1024.0
    print("WavStream load: %.3f s CPU, +%.1f MB max RSS" % (cpu, rss_mb))
    assert s.data.shape == (1, 20 * rate + seconds * 12000) and s.data.dtype == np.uint8
    assert s._dev_row is not None
    assert cpu <= 0.7, cpu
    assert rss_mb <= 120.0, rss_mb


def test_device_decode_matches_host_decode():
    """sushi_hip_load_decode (wav.py:64-91 on the GPU) against DownmixedWavFile._decode, bit for bit: 16- and 24-bit
    samples, 1 to 6 channels, frame counts that are not multiples of anything."""
    import torch
    from sushi_amd import _native
    from sushi_amd.wav import DownmixedWavFile
    L = _native.lib()
    rng = np.random.default_rng(9)
    for width in (2, 3):
        for channels in (1, 2, 3, 6):
            n = 10007
            blob = rng.integers(0, 256, n * channels * width, dtype=np.uint8).tobytes()
            host = DownmixedWavFile.__new__(DownmixedWavFile)
            host.sample_width, host.channels_count, host._file = width, channels, None
            want = host._decode(blob)
            pcm = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            out = torch.empty(n, dtype=torch.float32, device="cuda")
            rc = L.sushi_hip_load_decode(pcm.data_ptr(), n, channels, width, out.data_ptr(), None)
            assert rc == 0
            got = out.cpu().numpy()
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (width, channels)
