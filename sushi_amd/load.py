"""WavStream's value pipeline on the GPU (reference wav.py:113-156 after the RIFF decode).

Host: RIFF parse + channel downmix (``DownmixedWavFile``), the chunk bookkeeping of wav.py:113-137
(how many samples each one-second chunk becomes) and the 256-bin bucket walks of the radix select.
Device (libsushi_hip.so, csrc/sushi_load.hip): decimation + padding, the histograms behind the two
medians, clip / scale / quantise.  The result is bit-identical to ``wav.WavStream._build_host`` and the
normalised stream never crosses PCIe twice: the device copy is handed to ``DeviceStream`` as is.
"""
import ctypes
import logging
import math

import numpy as np
import torch

from . import _native
from .common import SushiError, py2_round


def _select(L, data, n, side, rank, hist, stream):
    """Key (uint32) of the element of ascending-key rank `rank` among the samples of one side."""
    prefix, mask = 0, 0
    for shift in (24, 16, 8, 0):
        _native.check(L.sushi_hip_load_histogram(data.data_ptr(), n, side, prefix, mask, shift, hist.data_ptr(), stream),
                      "sushi_hip_load_histogram")
        h = hist.cpu().numpy()
        c = np.cumsum(h)
        b = int(np.searchsorted(c, rank, side="right"))
        if b > 255:
            raise SushiError("radix select: rank outside the population")
        rank -= int(c[b - 1]) if b else 0
        prefix |= b << shift
        mask |= 0xFF << shift
    return prefix


def _median(L, data, n, side, hist, stream):
    """np.median(data[data >= 0]) (side 0) / np.median(data[data <= 0]) (side 1) as a Python float."""
    _native.check(L.sushi_hip_load_histogram(data.data_ptr(), n, side, 0, 0, 24, hist.data_ptr(), stream),
                  "sushi_hip_load_histogram")
    m = int(hist.cpu().numpy().sum())
    if m == 0:
        raise SushiError("stream has no samples on one side of zero")

    def value(ascending_rank):
        # side 1 holds keys of -x: ascending x is descending key
        r = ascending_rank if side == 0 else m - 1 - ascending_rank
        v = np.array([_select(L, data, n, side, r, hist, stream)], np.uint32).view(np.float32)[0]
        return v if side == 0 else np.float32(-v)

    if m % 2:
        return float(value(m // 2))
    lo, hi = value(m // 2 - 1), value(m // 2)
    return float(np.float32(lo + hi) / np.float32(2.0))       # np.mean of two float32 values


UPLOAD_CHUNK_BYTES = 32 << 20      # PCM bytes uploaded and decoded per step: bounds the host memory of a load


def decode_file_on_device(wavfile, dev):
    """wav.py:64-91 on the GPU: the data chunk of `wavfile` (a DownmixedWavFile positioned at its first frame) is
    read UPLOAD_CHUNK_BYTES at a time, uploaded, decoded and downmixed by sushi_hip_load_decode into one float32
    mono tensor.  Returns (tensor [frames], frames).  Host memory: one chunk of file bytes."""
    L = _native.lib()
    frame_size = wavfile.frame_size
    if wavfile.sample_width not in (2, 3):
        raise SushiError('Unsupported sample width: {0}'.format(wavfile.sample_width))
    frames_total = int(wavfile.frames_available)           # never more than the file holds, whatever the header says
    frames_per_chunk = max(1, UPLOAD_CHUNK_BYTES // frame_size)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        mono = torch.zeros(max(frames_total, 1), dtype=torch.float32, device=dev)
        stage = bytearray(min(frames_per_chunk, max(frames_total, 1)) * frame_size)     # the one host buffer of the load
        done = 0
        while done < frames_total:
            want = min(frames_per_chunk, frames_total - done)
            nbytes = wavfile.read_bytes_into(memoryview(stage)[:want * frame_size])
            got = nbytes // frame_size
            if got == 0:
                break                                            # file shorter than its header says
            if nbytes != got * frame_size:
                logging.error("Length of audio channels didn't match. This might result in broken output")
            # (a pageable-memory upload returns when the bytes have left `stage`: it can be refilled right away)
            staged = torch.frombuffer(stage, dtype=torch.uint8, count=got * frame_size).to(dev)
            _native.check(L.sushi_hip_load_decode(staged.data_ptr(), got, wavfile.channels_count, wavfile.sample_width,
                                                  mono.data_ptr() + 4 * done, st), "sushi_hip_load_decode")
            done += got
            del staged                                           # stream-ordered free: the kernel above is queued first
    return mono[:max(done, 1)], done                          # the frames that were there (n_raw of the pipeline)


def build_on_device(samples, framerate, frames_count, sample_rate, sample_type, device=None, read_chunk_size=1,
                    padding_seconds=10):
    """-> (host data ndarray (1, L) of dtype uint8/float32, device tensor of the same row, sample_count, padding_size)
    `samples`: downmixed frames, a float32 host array or a float32 CUDA tensor (decode_file_on_device)."""
    if sample_type not in ('float32', 'uint8'):
        raise SushiError('Unknown sample type of WAV stream, must be uint8 or float32')
    L = _native.lib()
    on_device = isinstance(samples, torch.Tensor)
    dev = samples.device if on_device else \
        (torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device))
    total_seconds = frames_count / float(framerate)
    downsample_rate = sample_rate / float(framerate)
    sample_count = math.ceil(total_seconds * sample_rate)
    padding_size = 10 * framerate
    total = int(padding_seconds * 2 * framerate + sample_count)
    chunk = int(read_chunk_size * framerate)
    n_raw = int(samples.shape[0])
    n_full, rest = divmod(n_raw, chunk)
    nl_full = int(py2_round(chunk * downsample_rate))
    nl_rest = int(py2_round(rest * downsample_rate)) if rest else 0
    if downsample_rate != 1 and (nl_full <= 0):
        raise SushiError('sample rate too low for one-second chunks')
    scale_full = 1.0 / (float(nl_full) / float(chunk)) if nl_full > 0 else 0.0
    scale_rest = 1.0 / (float(nl_rest) / float(rest)) if nl_rest > 0 else 0.0
    if total - 2 * padding_size < n_full * nl_full + nl_rest:
        raise SushiError('decimated stream does not fit its buffer')         # np.copyto would raise in the reference
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        raw = samples if on_device else torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float32)).to(dev)
        data = torch.empty(total, dtype=torch.float32, device=dev)
        _native.check(L.sushi_hip_load_resample(raw.data_ptr(), n_raw, chunk, nl_full, scale_full, n_full, rest, nl_rest,
                                                scale_rest, padding_size, total, data.data_ptr(), st),
                      "sushi_hip_load_resample")
        hist = torch.empty(256, dtype=torch.int64, device=dev)
        max_value = _median(L, data, total, 0, hist, st) * 3
        min_value = _median(L, data, total, 1, hist, st) * 3
        lo, hi, rng = np.float32(min_value), np.float32(max_value), np.float32(max_value - min_value)
        u8 = torch.empty(total, dtype=torch.uint8, device=dev) if sample_type == 'uint8' else None
        _native.check(L.sushi_hip_load_normalise(data.data_ptr(), total, ctypes.c_float(lo), ctypes.c_float(hi),
                                                 ctypes.c_float(rng), u8.data_ptr() if u8 is not None else None, st),
                      "sushi_hip_load_normalise")
        row = u8 if u8 is not None else data
        host = row.cpu().numpy().reshape(1, -1)
    return host, row, sample_count, padding_size
