"""oracle/oracle.py -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE).

PARITY UNPINNED: cv2 (where the arithmetic lives) is absent from this image, the
reference is Python 2 and holds no golden vector for this path.
What *is* pinned: the window/index arithmetic of ``wav.py:177-188`` -- see
``tests/golden/gen_find_substream_golden.py`` which executes the reference's own
``WavStream.find_substream`` / ``get_substream`` bytecode under Python 3 with a
stub ``cv2`` module, and ``tests/test_oracle.py`` which replays those vectors here.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``sushi_amd/`` does.

Restated reference code (all paths relative to /root/reference):
  * ``common.py:41-42``   clip
  * ``wav.py:164-166``    WavStream.duration_seconds
  * ``wav.py:168-175``    get_substream / _get_sample_for_time
  * ``wav.py:177-188``    find_substream
  * ``wav.py:108-162``    WavStream.__init__ value pipeline (load, decimate, pad, normalise, quantise)
  * ``wav.py:15-101``     DownmixedWavFile (RIFF parse + downmix)
  * cv2.matchTemplate(TM_SQDIFF_NORMED) -- ``match_template.c`` (C, direct) and
    ``match_template_fft`` below (NumPy/SciPy overlap-add FFT for big windows).
"""
from __future__ import annotations

import ctypes
import math
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

PADDING_SECONDS = 10  # wav.py:106
READ_CHUNK_SIZE = 1   # wav.py:105


def build(force: bool = False) -> str:
    """Compile match_template.c -> oracle/_build/liboracle.so (gcc)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "match_template.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        lib.oracle_match_sqdiff_normed_f32.argtypes = [vp, i64, vp, i64, vp, ci]
        lib.oracle_match_sqdiff_normed_u8.argtypes = [vp, i64, vp, i64, vp, ci]
        lib.oracle_finish_sqdiff_normed.argtypes = [vp, vp, i64, i64, ctypes.c_double, ctypes.c_double, vp, ci]
        lib.oracle_argmin_f32.argtypes = [vp, i64]
        lib.oracle_argmin_f32.restype = i64
        lib.oracle_definition_sqdiff_normed_f32.argtypes = [vp, i64, vp, i64, vp]
        lib.oracle_match_ccoeff_normed_f32.argtypes = [vp, i64, vp, i64, vp, ci]
        lib.oracle_match_ccoeff_normed_u8.argtypes = [vp, i64, vp, i64, vp, ci]
        lib.oracle_finish_ccoeff_normed.argtypes = [vp, vp, vp, i64, i64, ctypes.c_double, ctypes.c_double, vp, ci]
        lib.oracle_argmax_f32.argtypes = [vp, i64]
        lib.oracle_argmax_f32.restype = i64
        lib.oracle_definition_ccoeff_normed_f32.argtypes = [vp, i64, vp, i64, vp]
        lib.oracle_num_threads.restype = ci
        lib.oracle_set_num_threads.argtypes = [ci]
        lib.oracle_set_num_threads.restype = None
        _lib = lib
    return _lib


def num_threads() -> int:
    return int(_load().oracle_num_threads())


def set_num_threads(n: int) -> None:
    """OpenMP threads of the C restatement in this process (bench.py's CPU leg: one process per core, one thread each)."""
    _load().oracle_set_num_threads(int(n))


# --------------------------------------------------------------------------- the real cv2, when there is one

def cv2_module():
    """`import cv2` if it can be imported, else None.  It cannot in the build image (no OpenCV, no
    network), which is why parity is unpinned at this boundary; wherever it CAN (a GPU box with OpenCV, a
    maintainer's machine) tests/test_cv2_crosscheck.py and bench.py's `parity.cv2` field compare the oracle --
    and the HIP path -- with the call wav.py:185 actually makes."""
    try:
        import cv2
    except Exception:
        return None
    return cv2 if hasattr(cv2, "matchTemplate") else None


def match_template_cv2(search, templ, method: str = "sqdiff_normed") -> np.ndarray:
    """wav.py:185 itself: cv2.matchTemplate(search_source, pattern, cv2.TM_SQDIFF_NORMED) -> (1, P) float32."""
    cv2 = cv2_module()
    if cv2 is None:
        raise RuntimeError("cv2 is not importable here")
    s = _as_row(search)
    t = _as_row(templ, s.dtype)
    code = {"sqdiff_normed": cv2.TM_SQDIFF_NORMED, "ccoeff_normed": cv2.TM_CCOEFF_NORMED}[method]
    return np.asarray(cv2.matchTemplate(s.reshape(1, -1), t.reshape(1, -1), code), np.float32).reshape(1, -1)


# --------------------------------------------------------------------------- matchTemplate

def _as_row(a, dtype=None):
    a = np.asarray(a)
    if a.ndim == 2:
        if a.shape[0] != 1:
            raise ValueError("oracle handles 1 x N images only (wav.py keeps streams as (1, N))")
        a = a[0]
    if dtype is not None and a.dtype != dtype:
        raise TypeError("expected dtype %s, got %s" % (dtype, a.dtype))
    return np.ascontiguousarray(a)


SQDIFF_NORMED, CCOEFF_NORMED = "sqdiff_normed", "ccoeff_normed"      # cv2.TM_SQDIFF_NORMED (what wav.py:185 passes), cv2.TM_CCOEFF_NORMED


def match_template_direct(search, templ, corr_f32: bool = True, method: str = SQDIFF_NORMED) -> np.ndarray:
    """cv2.matchTemplate(search, templ, cv2.TM_SQDIFF_NORMED | TM_CCOEFF_NORMED) -> (1, P) float32, direct O(P*M) in C."""
    s = _as_row(search)
    t = _as_row(templ, s.dtype)
    L, M = s.shape[0], t.shape[0]
    if M <= 0 or L < M:
        raise ValueError("template larger than search image (cv2.error in the reference)")
    out = np.empty(L - M + 1, np.float32)
    lib = _load()
    if method not in (SQDIFF_NORMED, CCOEFF_NORMED):
        raise ValueError("unknown method %r" % (method,))
    stem = "oracle_match_" + method
    if s.dtype == np.float32:
        fn = getattr(lib, stem + "_f32")
    elif s.dtype == np.uint8:
        fn = getattr(lib, stem + "_u8")
    else:
        raise TypeError("sample type must be float32 or uint8 (wav.py:109)")
    rc = fn(s.ctypes.data, L, t.ctypes.data, M, out.ctypes.data, int(corr_f32))
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    return out.reshape(1, -1)


def cross_correlate_fft(search_row: np.ndarray, templ_row: np.ndarray) -> np.ndarray:
    """corr[p] = sum_m T[m] I[p+m] in float64 through SciPy's overlap-add FFT
    (what cv2's crossCorr does with its block DFT).  uint8 inputs are rounded
    back to the exact integer."""
    from scipy.signal import oaconvolve
    s = search_row.astype(np.float64)
    t = templ_row.astype(np.float64)
    corr = oaconvolve(s, t[::-1], mode="valid")
    if search_row.dtype == np.uint8:
        corr = np.rint(corr)
    return corr


def match_template_fft(search, templ, corr_f32: bool = True, method: str = SQDIFF_NORMED) -> np.ndarray:
    """Same result as match_template_direct, O(L log M): FFT cross-correlation +
    the C epilogue (common_matchTemplate restatement)."""
    s = _as_row(search)
    t = _as_row(templ, s.dtype)
    L, M = s.shape[0], t.shape[0]
    if M <= 0 or L < M:
        raise ValueError("template larger than search image (cv2.error in the reference)")
    P = L - M + 1
    corr = np.ascontiguousarray(cross_correlate_fft(s, t))
    s64 = s.astype(np.float64)
    sq = np.zeros(L + 1, np.float64)
    np.cumsum(s64 * s64, out=sq[1:])          # integral(..., sqsum, CV_64F)
    t64 = t.astype(np.float64)
    out = np.empty(P, np.float32)
    if method == CCOEFF_NORMED:
        s1 = np.zeros(L + 1, np.float64)
        np.cumsum(s64, out=s1[1:])            # integral(..., sum, sqsum, CV_64F)
        rc = _load().oracle_finish_ccoeff_normed(corr.ctypes.data, s1.ctypes.data, sq.ctypes.data, P, M,
                                                 float(t64.sum()), float((t64 * t64).sum()),
                                                 out.ctypes.data, int(corr_f32))
    elif method == SQDIFF_NORMED:
        rc = _load().oracle_finish_sqdiff_normed(corr.ctypes.data, sq.ctypes.data, P, M,
                                                 float(t64.sum()), float((t64 * t64).sum()),
                                                 out.ctypes.data, int(corr_f32))
    else:
        raise ValueError("unknown method %r" % (method,))
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    return out.reshape(1, -1)


def optimal_dft_size(n: int) -> int:
    """cv::getOptimalDFTSize: the smallest 2^a 3^b 5^c >= n (OpenCV holds them as a table; the same numbers)."""
    if n <= 1:
        return 1
    best = None
    p5 = 1
    while p5 < 2 * n:
        p35 = p5
        while p35 < 2 * n:
            v = p35
            while v < n:
                v *= 2
            best = v if best is None or v < best else best
            p35 *= 3
        p5 *= 5
    return int(best)


def cross_correlate_cv2_model(search_row: np.ndarray, templ_row: np.ndarray) -> np.ndarray:
    """A NOISE MODEL of cv2's crossCorr (OpenCV imgproc/templmatch.cpp, the function behind wav.py:185), not a restatement
    of its bits: the same blocking and the same ARITHMETIC PRECISION, through SciPy's FFT instead of OpenCV's.

      * block of result columns  = cvRound(4.5 * M), at least 256 - M + 1, at most P;
        DFT length               = getOptimalDFTSize(block + M - 1) (>= 2); block recomputed as length - M + 1
      * working depth (`maxDepth`): CV_64F when the image is deeper than CV_8S (float32 streams), otherwise CV_32F --
        **uint8 streams (the reference's default sample_type, sushi.py:769) go through a float32 DFT**
      * per block: image tile (block + M - 1 samples, zero padded) -> real DFT -> mulSpectrums(.., conjB=true) with the
        template's spectrum -> inverse DFT with DFT_SCALE -> the first `block` values, converted to the CV_32F result

    Returns corr as float32 (what matchTemplate's result matrix holds before common_matchTemplate runs).  Real cv2
    differs from this model in WHICH rounding errors it commits (another FFT factorisation, SIMD reductions), not in
    their size: use it to see how far the real call can sit from the exactly rounded oracle, never as a parity target."""
    import scipy.fft
    s, t = np.asarray(search_row), np.asarray(templ_row)
    L, M = s.shape[0], t.shape[0]
    P = L - M + 1
    work = np.float64 if s.dtype == np.float32 else np.float32          # maxDepth
    cwork = np.complex128 if work == np.float64 else np.complex64
    block = int(round(M * 4.5))                                         # cvRound: half to even, as Python's round
    block = max(block, 256 - M + 1)
    block = min(block, P)
    n_dft = max(optimal_dft_size(block + M - 1), 2)
    block = min(n_dft - M + 1, P)
    tpad = np.zeros(n_dft, work)
    tpad[:M] = t
    tspec = scipy.fft.rfft(tpad).astype(cwork, copy=False)
    out = np.empty(P, np.float32)
    for x in range(0, P, block):
        bsz = min(block, P - x)
        tile = np.zeros(n_dft, work)
        n_in = min(bsz + M - 1, L - x)
        tile[:n_in] = s[x:x + n_in]
        spec = scipy.fft.rfft(tile).astype(cwork, copy=False)
        spec *= np.conj(tspec)                                          # mulSpectrums(a, b, c, 0, conjB = true)
        out[x:x + bsz] = scipy.fft.irfft(spec, n_dft)[:bsz].astype(np.float32)
    return out


def match_template_cv2_model(search, templ, method: str = SQDIFF_NORMED) -> np.ndarray:
    """cv2.matchTemplate as `cross_correlate_cv2_model` + the exact epilogue: what the REAL call may return, to within
    the identity of its rounding errors.  tests/test_cv2_noise_model.py measures its distance to the exactly rounded
    oracle (DESIGN.md section 4 quotes the numbers)."""
    s = _as_row(search)
    t = _as_row(templ, s.dtype)
    L, M = s.shape[0], t.shape[0]
    if M <= 0 or L < M:
        raise ValueError("template larger than search image (cv2.error in the reference)")
    P = L - M + 1
    corr = np.ascontiguousarray(cross_correlate_cv2_model(s, t).astype(np.float64))
    s64 = s.astype(np.float64)
    sq = np.zeros(L + 1, np.float64)
    np.cumsum(s64 * s64, out=sq[1:])
    t64 = t.astype(np.float64)
    out = np.empty(P, np.float32)
    if method == CCOEFF_NORMED:
        s1 = np.zeros(L + 1, np.float64)
        np.cumsum(s64, out=s1[1:])
        rc = _load().oracle_finish_ccoeff_normed(corr.ctypes.data, s1.ctypes.data, sq.ctypes.data, P, M,
                                                 float(t64.sum()), float((t64 * t64).sum()), out.ctypes.data, 1)
    elif method == SQDIFF_NORMED:
        rc = _load().oracle_finish_sqdiff_normed(corr.ctypes.data, sq.ctypes.data, P, M,
                                                 float(t64.sum()), float((t64 * t64).sum()), out.ctypes.data, 1)
    else:
        raise ValueError("unknown method %r" % (method,))
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    return out.reshape(1, -1)


def match_template(search, templ, corr_f32: bool = True, method: str = SQDIFF_NORMED) -> np.ndarray:
    """Dispatch on size: direct C below ~2e9 MACs, FFT above."""
    s = _as_row(search)
    t = _as_row(templ)
    if (s.shape[0] - t.shape[0] + 1) * t.shape[0] <= 2_000_000_000:
        return match_template_direct(s, t, corr_f32, method)
    return match_template_fft(s, t, corr_f32, method)


def definition_ccoeff_normed(search, templ) -> np.ndarray:
    """sum (T-mean T)(I-mean I) / sqrt(sum (T-mean T)^2 sum (I-mean I)^2) in long double, no shared helper."""
    s = _as_row(search).astype(np.float32)
    t = _as_row(templ).astype(np.float32)
    out = np.empty(s.shape[0] - t.shape[0] + 1, np.float64)
    rc = _load().oracle_definition_ccoeff_normed_f32(s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], out.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    return out


def argmax_first(result_row: np.ndarray) -> int:
    """result.argmax(axis=1)[0]: first index of the maximum."""
    r = np.ascontiguousarray(np.asarray(result_row, np.float32).reshape(-1))
    return int(_load().oracle_argmax_f32(r.ctypes.data, r.shape[0]))


def definition_sqdiff_normed(search, templ) -> np.ndarray:
    """The textbook formula in long double, no shared code with the above (small sizes)."""
    s = _as_row(search).astype(np.float32)
    t = _as_row(templ).astype(np.float32)
    out = np.empty(s.shape[0] - t.shape[0] + 1, np.float64)
    rc = _load().oracle_definition_sqdiff_normed_f32(s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0],
                                                     out.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
    return out


def argmin_first(result_row: np.ndarray) -> int:
    r = np.ascontiguousarray(result_row, np.float32)
    return int(_load().oracle_argmin_f32(r.ctypes.data, r.shape[0]))


# --------------------------------------------------------------------------- wav.py restatement

def clip(value, minimum, maximum):
    """common.py:41-42"""
    return max(min(value, maximum), minimum)


class OracleWavStream(object):
    """Holds exactly the state ``wav.py`` WavStream methods read: ``data`` (1, N),
    ``sample_rate``, ``sample_count``, ``padding_size``."""

    PADDING_SECONDS = PADDING_SECONDS
    READ_CHUNK_SIZE = READ_CHUNK_SIZE

    def __init__(self, data, sample_rate, sample_count, padding_size):
        self.data = data
        self.sample_rate = sample_rate
        self.sample_count = sample_count
        self.padding_size = padding_size

    # wav.py:164-166
    @property
    def duration_seconds(self):
        return self.sample_count / self.sample_rate

    # wav.py:173-175
    def _get_sample_for_time(self, timestamp):
        return int(self.sample_rate * timestamp) + self.padding_size

    # wav.py:168-171
    def get_substream(self, start, end):
        start_off = self._get_sample_for_time(start)
        end_off = self._get_sample_for_time(end)
        return self.data[:, start_off:end_off]

    def search_bounds(self, pattern_len, window_center, window_size):
        """wav.py:178-184: (start_time, start_sample, end_sample_after_numpy_truncation)."""
        start_time = clip(window_center - window_size, -self.PADDING_SECONDS, self.duration_seconds)
        end_time = clip(window_center + window_size, 0, self.duration_seconds + self.PADDING_SECONDS)
        start_sample = self._get_sample_for_time(start_time)
        end_sample = self._get_sample_for_time(end_time) + pattern_len
        # NumPy slice semantics of data[:, start_sample:end_sample]
        n = self.data.shape[1]
        lo, hi, _ = slice(start_sample, end_sample).indices(n)
        return start_time, lo, max(hi, lo)

    # wav.py:177-188
    def find_substream(self, pattern, window_center, window_size, corr_f32=True, matcher=None):
        start_time = clip(window_center - window_size, -self.PADDING_SECONDS, self.duration_seconds)
        end_time = clip(window_center + window_size, 0, self.duration_seconds + self.PADDING_SECONDS)

        start_sample = self._get_sample_for_time(start_time)
        end_sample = self._get_sample_for_time(end_time) + len(pattern[0])

        search_source = self.data[:, start_sample:end_sample]
        result = (matcher or match_template)(search_source, pattern, corr_f32)
        min_idx = result.argmin(axis=1)[0]

        return result[0][min_idx], start_time + (min_idx / float(self.sample_rate))


def read_wav_downmixed(path):
    """wav.py:15-101 DownmixedWavFile: returns (framerate, channels, frames_count, reader) where
    reader(count) yields the next `count` frames downmixed to float32 (channel mean)."""
    f = open(path, "rb")
    hdr = f.read(12)
    if hdr[0:4] != b"RIFF":
        raise ValueError("File does not start with RIFF id")
    if hdr[8:12] != b"WAVE":
        raise ValueError("Not a WAVE file")
    file_size = os.path.getsize(path)
    fmt = None
    frames_count = None
    while True:
        ck = f.read(8)
        if len(ck) < 8:
            break
        name, size = ck[0:4], struct.unpack("<L", ck[4:8])[0]
        if name == b"fmt ":
            body = f.read(size + (size & 1))
            tag, channels, framerate, _avg, _align = struct.unpack("<HHLLH", body[:14])
            if tag not in (0x0001, 0xFFFE):
                raise ValueError("unknown format: %d" % tag)
            bits = struct.unpack("<H", body[14:16])[0]
            sample_width = (bits + 7) // 8
            fmt = (channels, framerate, sample_width)
        elif name == b"data":
            channels, framerate, sample_width = fmt
            frame_size = channels * sample_width
            if file_size > 0xFFFFFFFF:
                frames_count = (file_size - f.tell()) // frame_size
            else:
                frames_count = size // frame_size
            break
        else:
            f.seek(size + (size & 1), 1)
    if fmt is None or frames_count is None:
        raise ValueError("Invalid WAV file")
    channels, framerate, sample_width = fmt
    frame_size = channels * sample_width

    def readframes(count):
        if not count:
            return np.empty(0, np.float32)
        data = f.read(count * frame_size)
        if sample_width == 2:
            unpacked = np.frombuffer(data, dtype="<i2")
        elif sample_width == 3:
            raw = np.frombuffer(data, dtype=np.int8)
            unpacked = np.zeros(len(data) // 3, np.int16)
            unpacked.view(np.int8)[0::2] = raw[1::3]
            unpacked.view(np.int8)[1::2] = raw[2::3]
        else:
            raise ValueError("Unsupported sample width: %d" % sample_width)
        unpacked = unpacked.astype("float32")
        if channels == 1:
            return unpacked
        min_length = len(unpacked) // channels
        acc = None
        for i in range(channels):                  # reduce(lambda a, b: a[:n] + b[:n], channels)
            ch = unpacked[i::channels][:min_length]
            acc = ch.copy() if acc is None else acc + ch
        acc /= float(channels)
        return acc

    return framerate, channels, frames_count, readframes, f


def resize_nearest_row(row: np.ndarray, new_length: int) -> np.ndarray:
    """cv2.resize(row.reshape(1,-1), (new_length, 1), interpolation=cv2.INTER_NEAREST)[0]  (wav.py:133).
    OpenCV resizeNN: inv_scale_x = dsize.width / ssize.width; scale_x = 1. / inv_scale_x;
    x_ofs[x] = min(cvFloor(x * scale_x), ssize.width - 1)."""
    n = row.shape[0]
    inv_scale_x = float(new_length) / float(n)
    scale_x = 1.0 / inv_scale_x
    x = np.arange(new_length, dtype=np.float64)
    sx = np.floor(x * scale_x).astype(np.int64)
    np.minimum(sx, n - 1, out=sx)
    return row[sx]


def load_wav_stream(path, sample_rate=12000, sample_type="uint8") -> OracleWavStream:
    """wav.py:108-162, statement for statement (Python 3 spelling)."""
    if sample_type not in ("float32", "uint8"):
        raise ValueError("Unknown sample type of WAV stream, must be uint8 or float32")
    framerate, _channels, frames_count, readframes, fh = read_wav_downmixed(path)
    try:
        total_seconds = frames_count / float(framerate)
        downsample_rate = sample_rate / float(framerate)

        sample_count = math.ceil(total_seconds * sample_rate)
        # wav.py:119 uses np.empty; when decimation leaves the last few samples unwritten the
        # reference reads uninitialised memory there.  Zeros make the restatement deterministic.
        data = np.zeros((1, int(PADDING_SECONDS * 2 * framerate + sample_count)), np.float32)
        padding_size = 10 * framerate
        seconds_read = 0
        samples_read = padding_size
        while seconds_read < total_seconds:
            chunk = readframes(int(READ_CHUNK_SIZE * framerate))
            new_length = int(math.floor(len(chunk) * downsample_rate + 0.5))   # Python 2 round(): half away from zero
            dst_view = data[0][samples_read:samples_read + new_length]
            if downsample_rate != 1:
                chunk = resize_nearest_row(chunk, new_length)
            np.copyto(dst_view, chunk, casting="no")
            samples_read += new_length
            seconds_read += READ_CHUNK_SIZE

        data[0][0:padding_size].fill(data[0][padding_size])
        data[0][-padding_size:].fill(data[0][-padding_size - 1])

        # NumPy 1.x (the reference's era): float32 scalar * int -> float64, and float64 scalars are
        # demoted to float32 only when they meet the float32 array.  Python floats reproduce that
        # under NumPy 2 (weak scalars): (max - min) is formed in double, then rounded once.
        max_value = float(np.median(data[data >= 0])) * 3
        min_value = float(np.median(data[data <= 0])) * 3

        np.clip(data, min_value, max_value, out=data)

        data -= min_value
        data /= (max_value - min_value)

        if sample_type == "uint8":
            data *= 255.0
            data += 0.5
            data = data.astype("uint8")
    finally:
        fh.close()
    return OracleWavStream(data, sample_rate, sample_count, padding_size)
