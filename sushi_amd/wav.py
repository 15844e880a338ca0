"""Drop-in for the reference's ``wav`` module (``from wav import WavStream``, sushi.py:19).

Same constructor, attributes and methods as reference wav.py:104-188; ``find_substream`` runs on
the GPU (libsushi_hip.so) instead of ``cv2.matchTemplate`` + ``argmin`` (wav.py:185-186).
Extensions: ``WavStream.from_samples`` (in-memory PCM), ``find_substreams`` (batched).

The load pipeline (wav.py:64-162): the RIFF header walk on the host; PCM decode, channel downmix, decimation,
padding, the two medians, clip / scale / quantise on the GPU when one is present (the file is uploaded in bounded
chunks; sushi_amd/load.py, csrc/sushi_load.hip) and otherwise in NumPy, chunk by chunk (``_build_host``,
bit-identical; it is what the CPU tests compare with the reference-generated goldens).
``SUSHI_HIP_LOAD=host`` forces the NumPy pipeline.
"""
import logging
import math
import os
import struct
import weakref
from time import time

import numpy as np

from .common import SushiError, clip, py2_round

WAVE_FORMAT_PCM = 0x0001
WAVE_FORMAT_EXTENSIBLE = 0xFFFE


class DownmixedWavFile(object):
    """RIFF/WAVE reader with channel-mean downmix to float32 (reference wav.py:15-101)."""
    _file = None

    def __init__(self, path):
        super(DownmixedWavFile, self).__init__()
        self._file = open(path, 'rb')
        try:
            head = self._file.read(12)
            if head[0:4] != b'RIFF':
                raise SushiError('File does not start with RIFF id')
            if head[8:12] != b'WAVE':
                raise SushiError('Not a WAVE file')
            fmt_chunk_read = False
            data_chunk_read = False
            file_size = os.path.getsize(path)
            while True:
                hdr = self._file.read(8)
                if len(hdr) < 8:
                    break
                name, size = hdr[0:4], struct.unpack('<L', hdr[4:8])[0]
                if name == b'fmt ':
                    body = self._file.read(size + (size & 1))
                    self._read_fmt_chunk(body)
                    fmt_chunk_read = True
                elif name == b'data':
                    if not fmt_chunk_read:
                        raise SushiError('Invalid WAV file')
                    self._data_start = self._file.tell()
                    if file_size > 0xFFFFFFFF:
                        # large broken wav
                        self.frames_count = (file_size - self._file.tell()) // self.frame_size
                    else:
                        self.frames_count = size // self.frame_size
                    # what the file really holds: a data size that overstates the file (a truncated file, a
                    # 0xFFFFFFFF placeholder from a piped encoder) must not size any buffer of raw frames
                    self.frames_available = max(0, min(self.frames_count,
                                                       (file_size - self._data_start) // self.frame_size))
                    data_chunk_read = True
                    break
                else:
                    self._file.seek(size + (size & 1), 1)
            if not fmt_chunk_read or not data_chunk_read:
                raise SushiError('Invalid WAV file')
        except Exception:
            self.close()
            raise

    def __del__(self):
        self.close()

    def close(self):
        if self._file:
            self._file.close()
            self._file = None

    def _read_fmt_chunk(self, body):
        wFormatTag, self.channels_count, self.framerate, _avg, _align = struct.unpack('<HHLLH', body[:14])
        if wFormatTag == WAVE_FORMAT_PCM or wFormatTag == WAVE_FORMAT_EXTENSIBLE:  # ignore the rest
            bits_per_sample = struct.unpack('<H', body[14:16])[0]
            self.sample_width = (bits_per_sample + 7) // 8
        else:
            raise SushiError('unknown format: {0}'.format(wFormatTag))
        self.frame_size = self.channels_count * self.sample_width

    def _decode(self, data):
        if self.sample_width == 2:
            unpacked = np.frombuffer(data, dtype='<i2')
        elif self.sample_width == 3:
            raw_bytes = np.frombuffer(data, dtype=np.int8)
            unpacked = np.zeros(len(data) // 3, np.int16)
            unpacked.view(dtype='int8')[0::2] = raw_bytes[1::3]
            unpacked.view(dtype='int8')[1::2] = raw_bytes[2::3]
        else:
            raise SushiError('Unsupported sample width: {0}'.format(self.sample_width))
        unpacked = unpacked.astype('float32')
        if self.channels_count == 1:
            return unpacked
        min_length = len(unpacked) // self.channels_count
        if min_length * self.channels_count != len(unpacked):
            logging.error("Length of audio channels didn't match. This might result in broken output")
        frames = unpacked[:min_length * self.channels_count].reshape(min_length, self.channels_count)
        acc = frames[:, 0].copy()
        for c in range(1, self.channels_count):     # same left-to-right float32 sum as the reference
            acc += frames[:, c]
        acc /= float(self.channels_count)
        return acc

    def readframes(self, count):
        if not count:
            return np.empty(0, np.float32)
        return self._decode(self._file.read(count * self.frame_size))

    def read_bytes_into(self, view):
        """Fill `view` (a writable bytes-like object) with the next raw frame bytes; returns how many were read."""
        return self._file.readinto(view)

    def read_into(self, out, chunk_frames):
        """Decode and downmix the frames from the current position on into the float32 array `out`,
        `chunk_frames` at a time (bounded host memory, like the reference's one-second reads).  Returns the frames read."""
        done = 0
        while done < out.shape[0]:
            data = self.readframes(min(chunk_frames, out.shape[0] - done))
            if data.shape[0] == 0:
                break
            out[done:done + data.shape[0]] = data
            done += data.shape[0]
        return done


_live_streams = weakref.WeakSet()


def torch_device(device):
    import torch
    d = torch.device(device)
    return torch.device("cuda", torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d


def _locate(pattern):
    """If `pattern` is a view into a live WavStream's host data (what get_substream and np.split
    of it return, sushi.py:417,445), return (stream, offset, length); else None."""
    if not isinstance(pattern, np.ndarray) or pattern.ndim != 2 or pattern.shape[0] != 1:
        return None
    if pattern.strides[1] != pattern.itemsize:
        return None
    addr = pattern.__array_interface__['data'][0]
    for s in _live_streams:
        base = s.data.__array_interface__['data'][0]
        if s.data.dtype == pattern.dtype and base <= addr and \
                addr + pattern.shape[1] * pattern.itemsize <= base + s.data.nbytes:
            return s, (addr - base) // pattern.itemsize, pattern.shape[1]
    return None


class WavStream(object):
    READ_CHUNK_SIZE = 1  # one second, seems to be the fastest
    PADDING_SECONDS = 10

    def __init__(self, path, sample_rate=12000, sample_type='uint8', device=None):
        if sample_type not in ('float32', 'uint8'):
            raise SushiError('Unknown sample type of WAV stream, must be uint8 or float32')
        before_read = time()
        stream = DownmixedWavFile(path)
        try:
            if self._use_gpu():
                # decode + downmix on the GPU, the file uploaded in bounded chunks (sushi_amd/load.py)
                from .load import build_on_device, decode_file_on_device
                mono, got = decode_file_on_device(stream, torch_device("cuda" if device is None else device))
                if got == 0:
                    raise SushiError('no audio frames in the data chunk')
                self._dev_row = None
                self.data, self._dev_row, self.sample_count, self.padding_size = build_on_device(
                    mono, stream.framerate, stream.frames_count, sample_rate, sample_type,
                    read_chunk_size=self.READ_CHUNK_SIZE, padding_seconds=self.PADDING_SECONDS)
                self.sample_rate = sample_rate
            else:
                samples = np.zeros(stream.frames_available, np.float32)
                got = stream.read_into(samples, 10 * stream.framerate)
                if got == 0:
                    raise SushiError('no audio frames in the data chunk')
                # a file shorter than its header says: the frames that exist are decimated as the reference does (a
                # short last chunk by its own length, wav.py:127-134); what the reference leaves as uninitialised memory
                # (np.empty, wav.py:119) is zero here
                samples = samples[:got]
                self._build_host(samples, stream.framerate, stream.frames_count, sample_rate, sample_type)
        except Exception as e:
            raise SushiError('Error while loading {0}: {1}'.format(path, e))
        finally:
            stream.close()
        self._device = device
        self._dev = None
        _live_streams.add(self)
        logging.info('Done reading WAV {0} in {1}s'.format(path, time() - before_read))

    @classmethod
    def from_samples(cls, samples, framerate, sample_rate=12000, sample_type='uint8', device=None):
        """Build a stream from downmixed PCM samples already in memory (any real dtype; values as
        DownmixedWavFile would return them), running the same pipeline as the constructor."""
        if sample_type not in ('float32', 'uint8'):
            raise SushiError('Unknown sample type of WAV stream, must be uint8 or float32')
        self = cls.__new__(cls)
        samples = np.asarray(samples).astype(np.float32).reshape(-1)
        self._build(samples, int(framerate), samples.shape[0], sample_rate, sample_type)
        self._device = device
        self._dev = None
        _live_streams.add(self)
        return self

    @classmethod
    def from_prepared(cls, data, sample_rate, sample_count, padding_size, device=None):
        """Wrap a (1, N) row that already is what wav.py:113-156 produces (e.g. a saved ``WavStream.data``)."""
        data = np.ascontiguousarray(data)
        if data.ndim != 2 or data.shape[0] != 1 or data.dtype not in (np.dtype(np.uint8), np.dtype(np.float32)):
            raise SushiError('Unknown sample type of WAV stream, must be uint8 or float32')
        self = cls.__new__(cls)
        self.data, self.sample_rate, self.sample_count, self.padding_size = data, sample_rate, sample_count, padding_size
        self._dev_row = None
        self._device = device
        self._dev = None
        _live_streams.add(self)
        return self

    @staticmethod
    def _use_gpu():
        """The load pipeline runs on the GPU when there is one (SUSHI_HIP_LOAD=host forces NumPy: it is what the CPU
        tests compare with the reference-generated goldens, and what bench.py uses before it forks)."""
        if os.environ.get("SUSHI_HIP_LOAD", "auto") == "host":
            return False
        try:
            import torch
            return torch.cuda.is_available()
        except ImportError:
            return False

    def _build(self, samples, framerate, frames_count, sample_rate, sample_type):
        """wav.py:113-156: on the GPU if there is one (the normalised row then stays in HBM for the
        matching), else in NumPy."""
        self._dev_row = None
        if not self._use_gpu():
            return self._build_host(samples, framerate, frames_count, sample_rate, sample_type)
        from .load import build_on_device
        self.data, self._dev_row, self.sample_count, self.padding_size = build_on_device(
            samples, framerate, frames_count, sample_rate, sample_type,
            read_chunk_size=self.READ_CHUNK_SIZE, padding_seconds=self.PADDING_SECONDS)
        self.sample_rate = sample_rate

    # wav.py:113-156 (value pipeline) in NumPy, whole-stream instead of chunk-by-chunk
    def _build_host(self, samples, framerate, frames_count, sample_rate, sample_type):
        self._dev_row = None
        total_seconds = frames_count / float(framerate)
        downsample_rate = sample_rate / float(framerate)
        self.sample_count = math.ceil(total_seconds * sample_rate)
        self.sample_rate = sample_rate
        self.padding_size = 10 * framerate
        data = np.zeros((1, int(self.PADDING_SECONDS * 2 * framerate + self.sample_count)), np.float32)
        chunk = int(self.READ_CHUNK_SIZE * framerate)
        pos = self.padding_size
        if downsample_rate == 1:
            data[0, pos:pos + samples.shape[0]] = samples
            pos += samples.shape[0]
        else:
            # cv2.resize(..., INTER_NEAREST) per one-second chunk (wav.py:125-137):
            # x_ofs[x] = min(floor(x * (1 / (new_len / len))), len - 1)
            n_full, rest = divmod(samples.shape[0], chunk)
            for length, count, start in ((chunk, n_full, 0), (rest, 1 if rest else 0, n_full * chunk)):
                if count == 0 or length == 0:
                    continue
                new_length = int(py2_round(length * downsample_rate))
                if new_length <= 0:
                    continue
                scale_x = 1.0 / (float(new_length) / float(length))
                sx = np.minimum(np.floor(np.arange(new_length, dtype=np.float64) * scale_x).astype(np.int64),
                                length - 1)
                block = samples[start:start + count * length].reshape(count, length)[:, sx]
                data[0, pos:pos + count * new_length] = block.reshape(-1)
                pos += count * new_length
        # padding the audio from both sides
        data[0][0:self.padding_size].fill(data[0][self.padding_size])
        data[0][-self.padding_size:].fill(data[0][-self.padding_size - 1])
        # normalizing; also clipping the stream by 3*median value from both sides of zero
        max_value = float(np.median(data[data >= 0])) * 3
        min_value = float(np.median(data[data <= 0])) * 3
        np.clip(data, min_value, max_value, out=data)
        data -= min_value
        data /= (max_value - min_value)
        if sample_type == 'uint8':
            data *= 255.0
            data += 0.5
            data = data.astype('uint8')
        self.data = data

    # ------------------------------------------------------------------ reference API
    @property
    def duration_seconds(self):
        return self.sample_count / self.sample_rate

    def get_substream(self, start, end):
        start_off = self._get_sample_for_time(start)
        end_off = self._get_sample_for_time(end)
        return self.data[:, start_off:end_off]

    def _get_sample_for_time(self, timestamp):
        # this function gets REAL sample for time, taking padding into account
        return int(self.sample_rate * timestamp) + self.padding_size

    def find_substream(self, pattern, window_center, window_size):
        diffs, times = self.find_substreams([pattern], [window_center], [window_size])
        return diffs[0], times[0]

    # ------------------------------------------------------------------ batched form
    def device_stream(self):
        """The HBM mirror of self.data (created on first use)."""
        if self._dev is None:
            from .device import DeviceStream
            row = getattr(self, "_dev_row", None)
            if row is not None and (self._device is None or str(row.device) == str(torch_device(self._device))):
                self._dev = DeviceStream(row)           # already in HBM (GPU load pipeline): no H2D
            else:
                self._dev = DeviceStream(self.data[0], device=self._device)
            self._dev_row = None
        return self._dev

    def _window(self, pattern_len, window_center, window_size):
        """wav.py:178-184 -> (start_time, first sample of search_source, result length P)."""
        start_time = clip(window_center - window_size, -self.PADDING_SECONDS, self.duration_seconds)
        end_time = clip(window_center + window_size, 0, self.duration_seconds + self.PADDING_SECONDS)
        start_sample = self._get_sample_for_time(start_time)
        end_sample = self._get_sample_for_time(end_time) + pattern_len
        lo, hi, _ = slice(start_sample, end_sample).indices(self.data.shape[1])   # NumPy slice truncation
        return start_time, lo, max(hi - lo, 0) - pattern_len + 1

    def find_substreams(self, patterns, window_centers, window_sizes, with_index=False, method="sqdiff_normed"):
        """[find_substream(p, c, w) for p, c, w in zip(...)] in one GPU launch.
        Returns (diffs: float32 ndarray, times: list of float); with_index=True appends the absolute
        sample index of every match in self.data (what SpeculativeStream caches).
        method: 'sqdiff_normed' = what the reference's find_substream computes (wav.py:185-186: TM_SQDIFF_NORMED, argmin);
        'ccoeff_normed' = cv2.TM_CCOEFF_NORMED with argmax instead (not used by the reference; the method BASELINE.json names)."""
        from .device import DeviceStream, SearchBatch
        n = len(patterns)
        if not (len(window_centers) == len(window_sizes) == n) or n == 0:
            raise SushiError('find_substreams: need equally many patterns, centres and sizes (>= 1)')
        located = [_locate(p) for p in patterns]
        owners = set(id(l[0]) for l in located if l is not None)
        dst_dev = self.device_stream()
        if all(l is not None for l in located) and len(owners) == 1:
            src_dev = located[0][0].device_stream()
            offs = [l[1] for l in located]
            lens = [l[2] for l in located]
        else:
            # patterns that are not views of one live stream: upload them as one temporary stream
            rows = []
            for p in patterns:
                p = np.asarray(p)
                if p.ndim != 2 or p.shape[0] != 1:
                    raise SushiError('pattern must be a (1, M) array')
                if p.dtype != self.data.dtype:
                    raise SushiError('pattern and stream sample types differ')
                rows.append(np.ascontiguousarray(p[0]))
            lens = [r.shape[0] for r in rows]
            offs = list(np.concatenate(([0], np.cumsum(lens)[:-1])))
            if sum(lens) == 0:
                raise SushiError('empty pattern')
            src_dev = DeviceStream(np.concatenate(rows), device=dst_dev.device)
        start_times, win_start, n_pos = [], [], []
        for m, c, w in zip(lens, window_centers, window_sizes):
            st, lo, p = self._window(m, c, w)
            start_times.append(st)
            win_start.append(lo)
            n_pos.append(p)
        # A drop-in call is a batch of one (a triple of three): its cost is host work and launches, not arithmetic.  Such batches
        # are kept per (source stream, size, method) and RE-PLANNED in place for the next call (sushi_hip_batch_reset: no
        # allocation, one upload) instead of being built and torn down every time.
        batch = None
        pooled = n <= 4 and len(owners) == 1 and all(l is not None for l in located)
        if pooled:
            pool = self.__dict__.setdefault("_small_batches", {})
            from .device import default_path
            key = (id(src_dev), n, method, default_path(), os.environ.get("SUSHI_HIP_EXCLUSION"))   # (what a new batch would read from the environment)
            batch = pool.get(key)
            if batch is not None and (batch.dst is not dst_dev or batch.src is not src_dev or not batch.reset(offs, lens, win_start, n_pos)):
                batch = None
        if batch is None:
            batch = SearchBatch(dst_dev, src_dev, offs, lens, win_start, n_pos, method=method, headroom=4.0 if pooled else 1.0)
            if pooled:
                pool[key] = batch
        batch.run()
        idx, score = batch.results()
        times = [st + (int(k) / float(self.sample_rate)) for st, k in zip(start_times, idx)]
        if with_index:
            return score, times, [int(lo) + int(k) for lo, k in zip(win_start, idx)]
        return score, times
