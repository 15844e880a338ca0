#!/usr/bin/env python3
"""Generate tests/golden/wav_init.json by executing the REFERENCE's own WAV load code.

Runs only in the dev container (needs /root/reference).  What runs is the reference module itself
(`import wav` from /root/reference with a stub `cv2`), not a restatement:

  * ``DownmixedWavFile.readframes`` (wav.py:64-91): 16-bit / 24-bit decode and channel-mean downmix.
    The object is made with ``object.__new__`` and given the attributes ``__init__`` would have parsed
    (the RIFF walk of wav.py:18-51 compares ``bytes`` chunk names with ``str`` literals under Python 3 and
    cannot run; header parsing is pinned separately by tests/test_host_wav.py against the files themselves).
  * ``WavStream.__init__`` (wav.py:108-162) in full: allocation, the one-second chunk loop, padding,
    the two medians, clip, shift, scale, quantise.

Python-2 / NumPy-1 names the bytecode needs are supplied, nothing in the source text is patched:
  ``xrange``, ``reduce``, a Python-2 ``round`` (half away from zero), and a thin ``np`` proxy whose
  ``fromstring`` is ``frombuffer`` (binary ``fromstring`` is gone from NumPy 2), whose ``zeros`` accepts the
  float that ``len(data) / 3`` is under Python 3 (wav.py:72), and -- variant "numpy1" -- whose ``median``
  returns a Python float.  The last one reproduces NumPy 1.x scalar promotion, which is what the reference
  ran on (Python 2.7 => NumPy <= 1.16): there ``np.float32 * 3`` (wav.py:145-146) is a float64 SCALAR,
  ``max_value - min_value`` is formed in double, and both meet the float32 array as values cast to float32
  -- exactly what a Python float does under NumPy 2.  Variant "nep50" runs the same bytecode with NumPy 2's
  own promotion (float32 scalars throughout); it is recorded for information (``data_sha256_nep50``), the
  product follows "numpy1".  ``cv2.resize(..., INTER_NEAREST)`` (wav.py:133, only when the WAV rate differs
  from the target rate) is the oracle's restatement of OpenCV's index formula; those cases say
  ``resize_stubbed``.
"""
import hashlib
import io
import json
import math
import os
import sys
import types
from functools import reduce

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "wav_init.json")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import wav_cases  # noqa: E402
from oracle import oracle as O  # noqa: E402


def py2_round(x, nd=0):
    assert nd == 0
    return float(math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5))


class NumpyProxy(object):
    def __init__(self, legacy_scalars):
        self._legacy = legacy_scalars

    def __getattr__(self, name):
        return getattr(np, name)

    def fromstring(self, data, dtype=float):
        return np.frombuffer(data, dtype=dtype).copy()

    def zeros(self, shape, dtype=float):
        if isinstance(shape, float):
            assert shape == int(shape)
            shape = int(shape)
        return np.zeros(shape, dtype)

    def median(self, a, **kw):
        m = np.median(a, **kw)
        return float(m) if self._legacy else m


def load_reference():
    cv2 = types.ModuleType("cv2")
    cv2.TM_SQDIFF_NORMED = 5
    cv2.INTER_NEAREST = 0
    cv2.matchTemplate = None

    def resize(row2d, dsize, interpolation=None):
        assert interpolation == 0 and dsize[1] == 1 and row2d.shape[0] == 1
        return O.resize_nearest_row(row2d[0], dsize[0]).reshape(1, -1)
    cv2.resize = resize
    sys.modules["cv2"] = cv2
    sys.path.insert(0, REF)
    import wav as refwav
    refwav.xrange = range
    refwav.reduce = reduce
    refwav.round = py2_round
    return refwav


def run_reference(refwav, case, legacy):
    _, data = wav_cases.wav_bytes(case)
    refwav.np = NumpyProxy(legacy)
    real = refwav.DownmixedWavFile

    def open_stub(path):
        f = object.__new__(real)                          # the reference class: readframes / close are its own
        f._file = io.BytesIO(data)
        f.channels_count = case["channels"]
        f.framerate = case["framerate"]
        f.sample_width = case["width"]
        f.frame_size = case["channels"] * case["width"]
        f.frames_count = len(data) // f.frame_size
        return f
    refwav.DownmixedWavFile = open_stub
    try:
        s = refwav.WavStream("<memory>", sample_rate=case["sample_rate"], sample_type=case["sample_type"])
    finally:
        refwav.DownmixedWavFile = real
        refwav.np = np
    return s


def main():
    refwav = load_reference()
    out = []
    for case in wav_cases.CASES:
        blob, _ = wav_cases.wav_bytes(case)
        s = run_reference(refwav, case, legacy=True)
        s2 = run_reference(refwav, case, legacy=False)
        d = s.data
        n = d.shape[1]
        probe = [int(p) for p in np.linspace(0, n - 1, 48).astype(int)]
        rec = {"case": case, "wav_sha256": hashlib.sha256(blob).hexdigest(),
               "sample_count": float(s.sample_count), "padding_size": int(s.padding_size),
               "sample_rate": int(s.sample_rate), "duration_seconds": float(s.duration_seconds),
               "shape": list(d.shape), "dtype": str(d.dtype),
               "data_sha256": hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest(),
               "data_sha256_nep50": hashlib.sha256(np.ascontiguousarray(s2.data).tobytes()).hexdigest(),
               "probe_index": probe, "probe_value": [float(d[0, p]) for p in probe],
               "min": float(d.min()), "max": float(d.max()), "sum": float(d.astype(np.float64).sum())}
        out.append(rec)
        diff = int((s.data != s2.data).sum())
        print("%-26s %s n=%d  nep50 differs in %d samples" % (case["name"], d.dtype, n, diff))
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/gen_wav_init_golden.py",
                   "reference": "wav.py:64-91 (readframes), wav.py:108-162 (WavStream.__init__): reference bytecode, "
                                "NumPy 1.x scalar promotion reproduced (see the generator's docstring)",
                   "cases": out}, f, separators=(",", ":"))
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
