"""Seeded synthetic inputs of the shapes BASELINE.json names: audio-like PCM
streams with a planted offset, subtitle-event spans, and WAV/SRT writers for the plumbing config.
No dataset is available offline; everything here is generated."""
import math
import struct

import numpy as np


def make_dst_pcm(seconds, rate=12000, seed=0):
    """Gaussian white noise -> 8-tap moving average (audio-like low-pass) x slow |sin| envelope
    -> int16 at half scale.  Returns int16[seconds*rate]."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * rate))
    x = rng.standard_normal(n + 7).astype(np.float32)
    c = np.cumsum(x, dtype=np.float64)
    y = (c[7:] - np.concatenate(([0.0], c[:-8])))[:n] / 8.0
    t = np.arange(n, dtype=np.float64) / rate
    env = np.abs(np.sin(2 * math.pi * 0.2 * t)) + 0.1
    y = y * env
    y = y / (np.abs(y).max() + 1e-12) * (0.5 * 32767.0)
    return np.round(y).astype(np.int16)


def make_src_pcm(dst_pcm, offsets_samples, snr_db=20.0, seed=1):
    """src[t] = dst[t + off(t)] + white noise at `snr_db`.  `offsets_samples` is an int (global
    offset) or a list of (start_sample, offset) pieces (per-chapter offsets); dst times are
    src times + offset, so sushi's expected shift is +offset."""
    rng = np.random.default_rng(seed)
    n = dst_pcm.shape[0]
    if isinstance(offsets_samples, (int, np.integer)):
        pieces = [(0, int(offsets_samples))]
    else:
        pieces = list(offsets_samples)
    src = np.zeros(n, dtype=np.float64)
    for k, (start, off) in enumerate(pieces):
        end = pieces[k + 1][0] if k + 1 < len(pieces) else n
        idx = np.arange(start, end) + off
        valid = (idx >= 0) & (idx < n)
        seg = np.zeros(end - start)
        seg[valid] = dst_pcm[idx[valid]]
        src[start:end] = seg
    p_sig = float(np.mean(dst_pcm.astype(np.float64) ** 2))
    sigma = math.sqrt(p_sig / (10.0 ** (snr_db / 10.0)))
    src += rng.standard_normal(n) * sigma
    return np.clip(np.round(src), -32768, 32767).astype(np.int16)


def make_events(n_events, duration_s, max_abs_offset_s, seed=2, min_len=1.0, max_len=5.0):
    """Sorted event spans (start, end) in seconds: starts uniform in [15, dur-20-max|off|],
    durations U[min_len, max_len]."""
    rng = np.random.default_rng(seed)
    hi = duration_s - 20.0 - max_abs_offset_s
    lo = min(15.0 + max_abs_offset_s, hi)
    starts = np.sort(rng.uniform(lo, hi, n_events))
    lens = rng.uniform(min_len, max_len, n_events)
    return [(float(s), float(s + l)) for s, l in zip(starts, lens)]


def write_wav(path, pcm, rate, channels=1):
    """pcm_s16le WAV; `pcm` is int16[n] (mono) or int16[n, channels]."""
    pcm = np.ascontiguousarray(pcm, dtype='<i2')
    data = pcm.tobytes()
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<L', 36 + len(data)) + b'WAVE')
        f.write(b'fmt ' + struct.pack('<LHHLLHH', 16, 1, channels, rate, rate * channels * 2, channels * 2, 16))
        f.write(b'data' + struct.pack('<L', len(data)))
        f.write(data)


def _srt_time(seconds):
    cs = int(round(seconds * 1000))
    return '%02d:%02d:%02d,%03d' % (cs // 3600000, (cs // 60000) % 60, (cs // 1000) % 60, cs % 1000)


def write_srt(path, events):
    with open(path, 'w') as f:
        for k, (s, e) in enumerate(events):
            f.write('%d\n%s --> %s\nline %d\n\n' % (k + 1, _srt_time(s), _srt_time(e), k + 1))


def explicit_descriptors(src_stream, dst_stream, events, true_offset_s, window_s, seed=3, jitter=0.5):
    """For the raw-kernel configs (BASELINE configs 2/3/5): one search per event, centred at
    event start + true offset + U[-w*jitter, w*jitter] so the optimum is off-centre.
    Returns (patterns, centres, windows) for WavStream.find_substreams on dst_stream."""
    rng = np.random.default_rng(seed)
    patterns, centres, windows = [], [], []
    offs = true_offset_s if callable(true_offset_s) else (lambda t: true_offset_s)
    for (s, e) in events:
        patterns.append(src_stream.get_substream(s, e))
        centres.append(s + offs(s) + float(rng.uniform(-window_s * jitter, window_s * jitter)))
        windows.append(window_s)
    return patterns, centres, windows
