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


def make_hard_dst_pcm(seconds, rate=12000, seed=0, period_s=60.0):
    """make_dst_pcm plus, every `period_s` seconds, what real soundtracks hold and filtered noise does not:
    2.5 s of digital silence, 2.5 s of a held 400 Hz tone (an exact 30-sample period at 12 kHz: every window a
    period apart scores the same) and a 1.5 s jingle that is the same samples at every occurrence.
    Returns (int16 pcm, [(kind, start_s, end_s), ...])."""
    pcm = make_dst_pcm(seconds, rate, seed=seed).copy()
    rng = np.random.default_rng(seed + 1000)
    jingle = make_dst_pcm(1.5, rate, seed=seed + 1001)
    spans = []
    t = 20.0
    n = pcm.shape[0]
    while t + 12.0 < seconds - 20.0:
        a = int(round(t * rate))
        pcm[a:a + int(2.5 * rate)] = 0
        spans.append(("silence", t, t + 2.5))
        b = int(round((t + 5.0) * rate))
        k = np.arange(int(2.5 * rate))
        pcm[b:b + k.shape[0]] = np.round(6000.0 * np.sin(2 * math.pi * k * (400.0 / rate))).astype(np.int16)
        spans.append(("tone", t + 5.0, t + 7.5))
        c = int(round((t + 10.0) * rate))
        pcm[c:c + jingle.shape[0]] = jingle
        spans.append(("jingle", t + 10.0, t + 11.5))
        t += period_s + float(rng.uniform(0.0, 1.0))
    assert pcm.shape[0] == n
    return pcm, spans


def plant_hard_events(events, spans, offset_s, frac, seed=4):
    """Replace a fraction `frac` of `events` (source-time spans) by events cut from the hard spans of the
    destination (`spans` are destination times; source time = destination time - offset).  Returns the re-sorted
    event list and a bool mask of the planted ones."""
    rng = np.random.default_rng(seed)
    n = len(events)
    n_hard = max(1, int(round(frac * n))) if frac > 0 else 0
    chosen = set(np.linspace(0, n - 1, n_hard).astype(int).tolist()) if n_hard else set()
    starts = np.array([s for _, s, _ in spans])
    out = []
    for k, (s, e) in enumerate(events):
        if k in chosen:
            j = int(np.argmin(np.abs(starts - (s + offset_s))))
            kind, a, b = spans[j]
            ns = a - offset_s + float(rng.uniform(0.05, 0.25))
            ne = min(ns + float(rng.uniform(1.0, b - a - 0.4)), b - offset_s - 0.05)
            out.append((ns, ne, True))
        else:
            out.append((s, e, False))
    # an ordinary event whose destination span touches a hard span is a hard search too (its pattern holds part of
    # a tone / a jingle that recurs elsewhere in the window)
    ends = np.array([b for _, _, b in spans])
    for k, (s, e, h) in enumerate(out):
        if not h:
            j = int(np.searchsorted(starts, e + offset_s)) - 1
            if j >= 0 and s + offset_s < ends[j]:
                out[k] = (s, e, True)
    out.sort(key=lambda x: x[0])
    return [(s, e) for s, e, _ in out], np.array([h for _, _, h in out], bool)


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


def _shifted(dst_pcm, off):
    """float64 copy of dst advanced by `off` samples (zeros where that leaves the stream)."""
    n = dst_pcm.shape[0]
    idx = np.arange(n) + int(off)
    valid = (idx >= 0) & (idx < n)
    out = np.zeros(n)
    out[valid] = dst_pcm[idx[valid]]
    return out


def make_src_pcm_other_encode(dst_pcm, offset_samples, rate=12000, gain=0.7, cutoff_hz=4000.0, bits=8, seed=1):
    """What Sushi's source usually IS: another encode of the same programme -- here the destination advanced by the offset, at
    another level (`gain`), band-limited (a 63-tap windowed-sinc low-pass at `cutoff_hz`) and requantised to `bits` bits (the
    codec's noise stands in as quantisation noise: signal-dependent, not white, 6 dB per bit under full scale)."""
    x = _shifted(dst_pcm, offset_samples) * gain
    taps = 63
    k = np.arange(taps) - (taps - 1) / 2.0
    fc = cutoff_hz / rate
    h = 2.0 * fc * np.sinc(2.0 * fc * k) * np.hamming(taps)
    h /= h.sum()
    x = np.convolve(x, h, mode="same")
    step = float(1 << (16 - bits))
    x = np.round(x / step) * step
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def speech_gate(n, rate, seed, on_frac=0.5):
    """bool[n]: alternating stretches of 5-15 s, `on_frac` of the time on -- where a dub's own speech sits."""
    rng = np.random.default_rng(seed)
    gate = np.zeros(n, bool)
    t, on = 0, bool(rng.integers(0, 2))
    while t < n:
        ln = int(rng.uniform(5.0, 15.0) * rate * (2.0 * on_frac if on else 2.0 * (1.0 - on_frac)))
        if on:
            gate[t:t + ln] = True
        t += ln
        on = not on
    return gate


def make_dub_pcm(seconds, offset_samples, rate=12000, seed=0, bed_level=0.5, speech_level=0.8):
    """A dub: both streams share the music-and-effects bed (the source's advanced by the offset, 30 dB of white noise on it); each
    has its OWN speech on the same stretches (half of the time, speech_gate) -- unrelated audio, louder than the bed, exactly where
    subtitles are.  Returns (dst int16, src int16, gate of the DESTINATION bool[n])."""
    n = int(round(seconds * rate))
    bed = make_dst_pcm(seconds, rate, seed=seed).astype(np.float64) * bed_level
    gate = speech_gate(n, rate, seed + 11)
    sp_dst = make_dst_pcm(seconds, rate, seed=seed + 12).astype(np.float64) * speech_level * gate
    sp_src = make_dst_pcm(seconds, rate, seed=seed + 13).astype(np.float64) * speech_level * gate
    dst = bed + sp_dst
    rng = np.random.default_rng(seed + 14)
    src = _shifted(bed, offset_samples) + _shifted(sp_src, offset_samples) + \
        rng.standard_normal(n) * math.sqrt(float(np.mean(bed ** 2)) / 1000.0)
    to16 = lambda v: np.clip(np.round(v), -32768, 32767).astype(np.int16)
    return to16(dst), to16(src), gate


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
