#!/usr/bin/env python3
"""Generate tests/golden/find_substream_index.json by executing the REFERENCE's own code.

Runs only in the dev container (needs /root/reference).  The reference's `wav.py` imports under
Python 3 once a stub `cv2` is provided; `WavStream.__init__` cannot run (Python 2 idioms in the
RIFF reader), but `find_substream`, `get_substream`, `_get_sample_for_time`, `duration_seconds`
(wav.py:164-188) and `common.clip` (common.py:41-42) are pure arithmetic on four attributes, so
we build the object with `object.__new__` and set those attributes ourselves.

What this pins (golden = produced by reference bytecode):
  * the clip / int() truncation / padding arithmetic that turns (window_center, window_size,
    len(pattern)) into the slice handed to cv2.matchTemplate, including NumPy's silent slice
    truncation at the end of the stream and negative start times;
  * the template slice returned by get_substream;
  * the conversion of argmin index back to seconds.
What it does NOT pin: cv2.matchTemplate itself (stubbed -- returns a result whose minimum sits at a
position chosen by the case, so the index->time conversion is exercised).
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "find_substream_index.json")

captured = {}


def _match_template(search, pattern, method):
    assert method == 5
    P = search.shape[1] - pattern.shape[1] + 1
    captured["search_len"] = int(search.shape[1])
    captured["search_off"] = (search.__array_interface__["data"][0] - captured["base"]) // search.itemsize
    captured["pattern_len"] = int(pattern.shape[1])
    if P <= 0:
        raise RuntimeError("cv2.error: template larger than image")
    res = np.ones((1, P), np.float32)
    k = captured["want_min"] % P
    res[0, k] = 0.25
    captured["min_idx"] = int(k)
    return res


cv2 = types.ModuleType("cv2")
cv2.TM_SQDIFF_NORMED = 5
cv2.INTER_NEAREST = 0
cv2.matchTemplate = _match_template
cv2.resize = None
sys.modules["cv2"] = cv2
sys.path.insert(0, REF)
import wav as refwav  # noqa: E402  (the reference module, unmodified)
import common as refcommon  # noqa: E402


def make_stream(sample_rate, framerate, seconds, dtype):
    s = object.__new__(refwav.WavStream)
    import math
    total_seconds = seconds
    s.sample_count = float(math.ceil(total_seconds * sample_rate))  # Py2 math.ceil returns float
    s.sample_rate = sample_rate
    s.padding_size = 10 * framerate
    n = int(refwav.WavStream.PADDING_SECONDS * 2 * framerate + s.sample_count)
    s.data = np.zeros((1, n), dtype)
    return s


def main():
    rng = np.random.default_rng(20260924)
    cases = []
    streams = [
        (12000, 12000, 300.0, "float32"),
        (12000, 12000, 2700.0, "uint8"),
        (12000, 48000, 61.5, "float32"),     # padding_size in WAV frames != downsampled samples
        (24000, 24000, 123.456, "float32"),
        (8000, 8000, 7.3, "uint8"),
    ]
    for (sr, fr, secs, dt) in streams:
        s = make_stream(sr, fr, secs, dt)
        captured["base"] = s.data.__array_interface__["data"][0]
        dur = s.duration_seconds
        centers = [0.0, 1.5, 3.3333333, dur - 0.001, dur, dur + 3.7, dur + 12.0, -2.5, -11.0,
                   float(rng.uniform(0, dur)), 5]
        windows = [1.5, 10, 120, 0.25]
        pat_spans = [(0.0, 1.0), (1.2345, 4.9), (0.5, 0.5 + 0.417), (3, 8), (1.0, 13.5)]
        for c in centers:
            for w in windows:
                for (ps, pe) in pat_spans:
                    pattern = s.get_substream(ps, pe)
                    pat_off = (pattern.__array_interface__["data"][0] - captured["base"]) // pattern.itemsize
                    captured["want_min"] = int(rng.integers(0, 1 << 30))
                    rec = {"sample_rate": sr, "framerate": fr, "seconds": secs, "dtype": dt,
                           "sample_count": s.sample_count, "padding_size": s.padding_size,
                           "data_len": int(s.data.shape[1]), "center": c, "window": w,
                           "pat_start": ps, "pat_end": pe, "pat_off": int(pat_off),
                           "pat_len": int(pattern.shape[1]), "want_min": captured["want_min"]}
                    try:
                        diff, t = s.find_substream(pattern, c, w)
                        rec.update({"ok": True, "search_off": int(captured["search_off"]),
                                    "search_len": captured["search_len"], "min_idx": captured["min_idx"],
                                    "time": float(t), "diff": float(diff)})
                    except RuntimeError:
                        rec.update({"ok": False, "search_off": int(captured["search_off"]),
                                    "search_len": captured["search_len"]})
                    cases.append(rec)
    clip_cases = []
    for _ in range(40):
        v, lo, hi = [float(x) for x in rng.uniform(-50, 50, 3)]
        clip_cases.append({"v": v, "lo": lo, "hi": hi, "out": refcommon.clip(v, lo, hi)})
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/gen_find_substream_golden.py", "reference": "tp7/Sushi wav.py:164-188",
                   "cases": cases, "clip": clip_cases}, f, separators=(",", ":"))
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
