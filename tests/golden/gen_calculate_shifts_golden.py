#!/usr/bin/env python3
"""Generate tests/golden/calculate_shifts.json by executing the REFERENCE's own ``calculate_shifts``.

Runs only in the dev container (needs /root/reference).  The source text of ``calculate_shifts``
(sushi.py:400-508) and ``abs_diff`` is read from the reference file at run time and exec'd in a
namespace that supplies the Python-2 names it uses (``izip``, ``chain``, ``format_time``).  ONE
textual patch is applied, and asserted to apply exactly once: ``len(tv_audio[0])/2`` (sushi.py:445,
integer division in Python 2) becomes ``len(tv_audio[0])//2``.  ``common.format_time`` runs with a
Python-2 ``round`` (half away from zero) injected into its module globals, so the log lines are
the ones Python 2 prints.  Events are the reference's own ``subs.ScriptEventBase``.

Two families of cases:

* ``scripted``: the streams are tests/shifts_fakes.py stand-ins whose answers follow a script that
  forces every branch of the state machine -- small-window commit (:431-443), triple search
  (:450-455), retry around the last uncommitted shift (:457-465), uncommitted pile-up and "back on
  track" (:468-470, :481-493), rewind with the window widened to max_window (:471-479), groups past
  the end of the destination (:424-429) and linking (:498-505).  The golden holds the complete
  sequence of ``find_substream`` calls the reference issued, the resulting per-event shift / diff /
  link, and the log lines.
* ``streams``: the streams are the reference's own ``wav.WavStream`` objects (``find_substream`` /
  ``get_substream`` bytecode, wav.py:164-188) over seeded synthetic audio with per-chapter offsets;
  ``cv2.matchTemplate`` is the CPU oracle (cv2 itself is not installable here).  The golden holds the
  call sequence as (pattern offset, length, centre, window) and the per-event results, to be replayed
  by sushi_amd.shifts.calculate_shifts on oracle-backed streams (CPU) and on the HIP path (GPU).
"""
import json
import logging
import math
import os
import re
import sys
import types
from itertools import chain

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "calculate_shifts.json")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import shifts_fakes  # noqa: E402
from oracle import oracle as O  # noqa: E402
from sushi_amd import synth  # noqa: E402


def py2_round(x, nd=0):
    assert nd == 0
    return float(math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5))


def load_reference():
    cv2 = types.ModuleType("cv2")
    cv2.TM_SQDIFF_NORMED = 5
    cv2.INTER_NEAREST = 0
    cv2.matchTemplate = lambda search, pattern, method: O.match_template(search, pattern)
    cv2.resize = None
    sys.modules["cv2"] = cv2
    sys.path.insert(0, REF)
    import common as refcommon
    import subs as refsubs
    import wav as refwav
    refcommon.round = py2_round                       # format_time: Python 2 rounding
    text = open(os.path.join(REF, "sushi.py")).read()
    ns = {"np": np, "logging": logging, "izip": zip, "chain": chain, "format_time": refcommon.format_time,
          "ALLOWED_ERROR": 0.01}
    assert re.search(r"^ALLOWED_ERROR = 0.01$", text, re.M)
    for name in ("abs_diff", "calculate_shifts"):
        m = re.search(r"^def %s\(.*?(?=^\S)" % name, text, re.S | re.M)
        assert m, name
        src = m.group(0)
        if name == "calculate_shifts":
            assert src.count("len(tv_audio[0])/2") == 1
            src = src.replace("len(tv_audio[0])/2", "len(tv_audio[0])//2")       # sushi.py:445, Python 2 int division
        exec(compile(src, "reference:sushi.py:" + name, "exec"), ns)
    return ns, refsubs, refwav


class LogCapture(logging.Handler):
    def __init__(self):
        super(LogCapture, self).__init__(level=logging.DEBUG)
        self.lines = []

    def emit(self, record):
        self.lines.append([record.levelname, record.getMessage()])


def dump_events(events):
    index = {id(e): k for k, e in enumerate(events)}
    return [{"shift": None if e.shift is None else float(e.shift), "diff": None if e.diff is None else float(e.diff),
             "linked_to": index[id(e._linked_event)] if e.linked else None} for e in events]


def make_groups(events, group_size):
    return [events[k:k + group_size] for k in range(0, len(events), group_size)]


def run_logged(fn, *args):
    cap = LogCapture()
    root = logging.getLogger()
    old = root.level
    root.addHandler(cap)
    root.setLevel(logging.DEBUG)
    try:
        fn(*args)
    finally:
        root.removeHandler(cap)
        root.setLevel(old)
    return cap.lines


def scripted_case(ns, refsubs, sc):
    script = sc["script"]
    src = shifts_fakes.FakeSource(script["sample_rate"], max(sc["starts"]) + sc["length"] + 5)
    dst = shifts_fakes.FakeDestination(script)
    events = [refsubs.ScriptEventBase(k, float(s), float(s) + sc["length"], u"line %d" % k)
              for k, s in enumerate(sc["starts"])]
    groups = make_groups(events, sc.get("group_size", 1))
    lines = run_logged(ns["calculate_shifts"], src, dst, groups, sc["window"], sc["max_window"], sc["rewind"])
    return {"kind": "scripted", "name": sc["name"], "calls": dst.calls, "events": dump_events(events), "log": lines}


STREAM_CASES = [
    # rate, seconds, pieces (chapter start s, offset s), n events, sample type, seed, window, max_window, rewind
    {"name": "chapters-u8", "rate": 2000, "seconds": 120, "pieces": [[0.0, 1.5], [40.0, -2.25], [80.0, 4.0]],
     "n_events": 40, "sample_type": "uint8", "seed": 3, "window": 10, "max_window": 30, "rewind": 5},
    {"name": "chapters-f32", "rate": 2000, "seconds": 120, "pieces": [[0.0, -1.0], [50.0, 12.5], [90.0, 12.5]],
     "n_events": 40, "sample_type": "float32", "seed": 7, "window": 10, "max_window": 30, "rewind": 3},
    {"name": "short-destination", "rate": 2000, "seconds": 80, "pieces": [[0.0, 1.0], [30.0, 14.0]], "dst_seconds": 60,
     "n_events": 24, "sample_type": "float32", "seed": 9, "window": 5, "max_window": 20, "rewind": 2},
]


def stream_inputs(case):
    """Seeded PCM + event spans of one `streams` case (shared with tests/test_shifts_golden.py)."""
    rate, seconds, seed = case["rate"], case["seconds"], case["seed"]
    dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
    pieces = [(int(t * rate), int(round(off * rate))) for t, off in case["pieces"]]
    src_pcm = synth.make_src_pcm(dst_pcm, pieces, seed=seed + 1)
    if "dst_seconds" in case:
        dst_pcm = dst_pcm[:case["dst_seconds"] * rate]
    rng = np.random.default_rng(seed + 2)
    starts = np.sort(rng.uniform(8.0, seconds - 12.0, case["n_events"]))
    spans = []
    for s in starts:
        e = s + float(rng.uniform(1.0, 3.0))
        if spans and s < spans[-1][1] + 0.05:
            continue
        spans.append((float(s), float(e)))
    return dst_pcm, src_pcm, spans


def ref_stream(refwav, host_stream):
    """A reference WavStream object (its own methods) around a value-pipeline output."""
    s = object.__new__(refwav.WavStream)
    s.sample_count = float(host_stream.sample_count)          # Python 2 math.ceil returns a float (wav.py:116)
    s.sample_rate = host_stream.sample_rate
    s.padding_size = host_stream.padding_size
    s.data = host_stream.data
    return s


def stream_case(ns, refsubs, refwav, case):
    os.environ["SUSHI_HIP_LOAD"] = "host"
    from sushi_amd.wav import WavStream
    dst_pcm, src_pcm, spans = stream_inputs(case)
    rate = case["rate"]
    hdst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=case["sample_type"])
    hsrc = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=case["sample_type"])
    rdst, rsrc = ref_stream(refwav, hdst), ref_stream(refwav, hsrc)
    calls = []
    inner = rdst.find_substream
    src_base = rsrc.data.__array_interface__["data"][0]

    def logged(pattern, centre, size):
        off = (pattern.__array_interface__["data"][0] - src_base) // pattern.itemsize
        diff, t = inner(pattern, centre, size)
        calls.append([int(off), int(pattern.shape[1]), float(centre), float(size), float(diff), float(t)])
        return diff, t
    rdst.find_substream = logged
    events = [refsubs.ScriptEventBase(k, s, e, u"line %d" % k) for k, (s, e) in enumerate(spans)]
    lines = run_logged(ns["calculate_shifts"], rsrc, rdst, [[e] for e in events], case["window"], case["max_window"],
                       case["rewind"])
    import hashlib
    return {"kind": "streams", "name": case["name"], "case": case, "calls": calls, "events": dump_events(events),
            "log": lines, "n_spans": len(spans),
            "dst_sha256": hashlib.sha256(hdst.data.tobytes()).hexdigest(),
            "src_sha256": hashlib.sha256(hsrc.data.tobytes()).hexdigest()}


def main():
    O.build()
    ns, refsubs, refwav = load_reference()
    cases = [scripted_case(ns, refsubs, sc) for sc in shifts_fakes.SCENARIOS]
    cases += [stream_case(ns, refsubs, refwav, c) for c in STREAM_CASES]
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/gen_calculate_shifts_golden.py",
                   "reference": "sushi.py:400-508 (calculate_shifts source exec'd; one patch: len(tv_audio[0])/2 -> //2)",
                   "cases": cases}, f, separators=(",", ":"))
    for c in cases:
        levels = sorted(set(l for l, _ in c["log"]))
        print("%-18s %-9s calls %4d  events %3d  linked %2d  log levels %s" % (
            c["name"], c["kind"], len(c["calls"]), len(c["events"]),
            sum(1 for e in c["events"] if e["linked_to"] is not None), levels))
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
