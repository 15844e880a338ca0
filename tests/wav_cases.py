"""Seeded WAV files for the WavStream.__init__ golden (TEST INFRASTRUCTURE).

Shared by tests/golden/gen_wav_init_golden.py (which feeds them to the REFERENCE's own
``WavStream.__init__`` / ``DownmixedWavFile.readframes``, wav.py:64-91,108-162) and by
tests/test_wav_init_golden.py (which feeds the same bytes to sushi_amd.wav.WavStream).
"""
import struct

import numpy as np

CASES = [
    # name, framerate, channels, sample width (bytes), seconds, target sample_rate, sample_type, kind of signal
    {"name": "mono16-12k-u8", "framerate": 12000, "channels": 1, "width": 2, "seconds": 31.4, "sample_rate": 12000,
     "sample_type": "uint8", "signal": "noise", "seed": 1},
    {"name": "mono16-12k-f32", "framerate": 12000, "channels": 1, "width": 2, "seconds": 31.4, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 1},
    {"name": "mono16-12k-f32-b", "framerate": 12000, "channels": 1, "width": 2, "seconds": 17.0, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 5},
    {"name": "mono16-12k-f32-c", "framerate": 12000, "channels": 1, "width": 2, "seconds": 23.77, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 6},
    {"name": "mono16-12k-f32-d", "framerate": 12000, "channels": 1, "width": 2, "seconds": 12.5, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 7},
    {"name": "stereo16-12k-f32", "framerate": 12000, "channels": 2, "width": 2, "seconds": 20.37, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 2},
    {"name": "stereo16-12k-u8", "framerate": 12000, "channels": 2, "width": 2, "seconds": 20.37, "sample_rate": 12000,
     "sample_type": "uint8", "signal": "noise", "seed": 2},
    {"name": "six24-12k-u8", "framerate": 12000, "channels": 6, "width": 3, "seconds": 11.25, "sample_rate": 12000,
     "sample_type": "uint8", "signal": "noise", "seed": 3},
    {"name": "six24-12k-f32", "framerate": 12000, "channels": 6, "width": 3, "seconds": 11.25, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 3},
    {"name": "silence-gaps-f32", "framerate": 12000, "channels": 1, "width": 2, "seconds": 25.0, "sample_rate": 12000,
     "sample_type": "float32", "signal": "gaps", "seed": 4},
    {"name": "silence-gaps-u8", "framerate": 12000, "channels": 2, "width": 2, "seconds": 25.0, "sample_rate": 12000,
     "sample_type": "uint8", "signal": "gaps", "seed": 4},
    {"name": "half-second-u8", "framerate": 12000, "channels": 1, "width": 2, "seconds": 0.5, "sample_rate": 12000,
     "sample_type": "uint8", "signal": "noise", "seed": 8},
    {"name": "mono16-24k-f32", "framerate": 24000, "channels": 1, "width": 2, "seconds": 9.3, "sample_rate": 24000,
     "sample_type": "float32", "signal": "noise", "seed": 9},
    # odd channel counts: the downmixed samples are not exactly representable thirds / fifths / sixths, so the
    # float32-vs-float64 scalar arithmetic of wav.py:145-151 (NumPy 1.x promotion) shows in the result (seeds 100, 106,
    # 118, 121 were picked because the two promotions give different streams there)
    {"name": "ch5-16-12k-f32-100", "framerate": 12000, "channels": 5, "width": 2, "seconds": 26.2, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 100},
    {"name": "ch5-16-12k-f32-106", "framerate": 12000, "channels": 5, "width": 2, "seconds": 26.1, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 106},
    {"name": "ch5-16-12k-u8-118", "framerate": 12000, "channels": 5, "width": 2, "seconds": 26.6, "sample_rate": 12000,
     "sample_type": "uint8", "signal": "noise", "seed": 118},
    {"name": "ch5-16-12k-f32-121", "framerate": 12000, "channels": 5, "width": 2, "seconds": 26.2, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 121},
    {"name": "ch6-16-12k-f32-101", "framerate": 12000, "channels": 6, "width": 2, "seconds": 26.3, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 101},
    {"name": "ch3-16-12k-f32-102", "framerate": 12000, "channels": 3, "width": 2, "seconds": 26.4, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 102},
    # downsample_rate != 1: cv2.resize(INTER_NEAREST) is stubbed by the restated OpenCV index formula in the generator
    {"name": "stereo16-48k-to-12k-f32", "framerate": 48000, "channels": 2, "width": 2, "seconds": 12.6, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 10, "resize_stubbed": True},
    {"name": "stereo16-48k-to-12k-u8", "framerate": 48000, "channels": 2, "width": 2, "seconds": 12.6, "sample_rate": 12000,
     "sample_type": "uint8", "signal": "noise", "seed": 10, "resize_stubbed": True},
    {"name": "mono16-44k1-to-12k-f32", "framerate": 44100, "channels": 1, "width": 2, "seconds": 7.0, "sample_rate": 12000,
     "sample_type": "float32", "signal": "noise", "seed": 11, "resize_stubbed": True},
]


def _signal(n, rate, kind, rng):
    x = rng.standard_normal(n + 7)
    c = np.cumsum(x)
    y = (c[7:] - np.concatenate(([0.0], c[:-8])))[:n] / 8.0
    t = np.arange(n) / float(rate)
    y *= np.abs(np.sin(2 * np.pi * 0.2 * t)) + 0.1
    if kind == "gaps":                                  # digital silence every few seconds
        y[(t % 5.0) > 3.5] = 0.0
    return y / (np.abs(y).max() + 1e-12)


def pcm_frames(case):
    """int array [frames, channels]: int16 range for 2-byte samples, 24-bit range for 3-byte ones."""
    rng = np.random.default_rng(20260924 + case["seed"])
    n = int(round(case["seconds"] * case["framerate"]))
    full = 32767.0 if case["width"] == 2 else 8388607.0
    cols = []
    base = _signal(n, case["framerate"], case["signal"], rng)
    for c in range(case["channels"]):
        y = base * (0.5 + 0.07 * c) + 0.05 * _signal(n, case["framerate"], case["signal"], rng)
        cols.append(np.round(y * 0.6 * full))
    return np.stack(cols, axis=1).astype(np.int64)


def wav_bytes(case):
    frames = pcm_frames(case)
    if case["width"] == 2:
        data = frames.astype('<i2').tobytes()
    else:
        v = frames.astype('<i4').reshape(-1)
        b = v.view(np.uint8).reshape(-1, 4)[:, :3]      # little endian: low three bytes
        data = np.ascontiguousarray(b).tobytes()
    ch, rate, width = case["channels"], case["framerate"], case["width"]
    head = b'RIFF' + struct.pack('<L', 36 + len(data)) + b'WAVE'
    head += b'fmt ' + struct.pack('<LHHLLHH', 16, 1, ch, rate, rate * ch * width, ch * width, 8 * width)
    head += b'data' + struct.pack('<L', len(data))
    return head + data, data
