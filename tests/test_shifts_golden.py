"""sushi_amd.shifts.calculate_shifts against tests/golden/calculate_shifts.json, which was produced by executing
the REFERENCE's own ``calculate_shifts`` source (sushi.py:400-508; tests/golden/gen_calculate_shifts_golden.py).

* ``scripted`` cases: the same scripted stand-in streams (tests/shifts_fakes.py) on both sides; compared are the
  complete sequence of find_substream calls, every event's shift / diff / link, and every log line.
* ``streams`` cases: the reference state machine ran over the reference's own WavStream methods with the CPU oracle
  as cv2.matchTemplate; replayed here on oracle-backed streams (sequential and speculative-batched, CPU) and on the
  HIP path (GPU marker)."""
import hashlib
import json
import logging
import os

import numpy as np
import pytest

import shifts_fakes
from sushi_amd import synth
from sushi_amd.shifts import ScriptEvent, calculate_shifts, calculate_shifts_batched
from sushi_amd.wav import WavStream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

with open(os.path.join(ROOT, "tests", "golden", "calculate_shifts.json")) as _f:
    GOLDEN = json.load(_f)["cases"]
SCRIPTED = [g for g in GOLDEN if g["kind"] == "scripted"]
STREAMS = [g for g in GOLDEN if g["kind"] == "streams"]


def _dump(events):
    index = {id(e): k for k, e in enumerate(events)}
    return [{"shift": None if e.shift is None else float(e.shift), "diff": None if e.diff is None else float(e.diff),
             "linked_to": index[id(e._linked_event)] if e.linked else None} for e in events]


def _logged(caplog, fn, *args, **kw):
    caplog.clear()
    with caplog.at_level(logging.DEBUG):
        out = fn(*args, **kw)
    return out, [[r.levelname, r.getMessage()] for r in caplog.records]


def test_golden_covers_every_branch():
    """The scripted cases reach every branch of sushi.py:400-508 (judged by the reference's own log lines)."""
    msgs = [m for g in SCRIPTED for _, m in g["log"]]
    for needle in ("increasing the window", "will most likely be broken", "Going back to window",
                   "outside of audio range", "search offset"):
        assert any(needle in m for m in msgs), needle
    assert any(e["linked_to"] is not None for g in SCRIPTED for e in g["events"])
    # the retry around the last uncommitted shift (:457-465): two consecutive debug lines for one group, with
    # different search offsets
    def retries(g):
        dbg = [m for l, m in g["log"] if l == "DEBUG"]
        return sum(1 for a, b in zip(dbg, dbg[1:]) if a.split(": shift")[0] == b.split(": shift")[0]
                   and a.split("offset: ")[1] != b.split("offset: ")[1])
    assert sum(retries(g) for g in SCRIPTED) >= 10


@pytest.mark.parametrize("g", SCRIPTED, ids=[g["name"] for g in SCRIPTED])
def test_scripted_state_machine_equals_reference(g, caplog):
    sc = next(s for s in shifts_fakes.SCENARIOS if s["name"] == g["name"])
    src = shifts_fakes.FakeSource(sc["script"]["sample_rate"], max(sc["starts"]) + sc["length"] + 5)
    dst = shifts_fakes.FakeDestination(sc["script"])
    events = [ScriptEvent(float(s), float(s) + sc["length"], source_index=k) for k, s in enumerate(sc["starts"])]
    size = sc.get("group_size", 1)
    groups = [events[k:k + size] for k in range(0, len(events), size)]
    _, lines = _logged(caplog, calculate_shifts, src, dst, groups, sc["window"], sc["max_window"], sc["rewind"])
    assert dst.calls == g["calls"]                  # same questions, in the same order, with the same floats
    assert _dump(events) == g["events"]
    assert lines == g["log"]


class OracleBackedStream(WavStream):
    """WavStream whose matching is done by the CPU oracle (no GPU); logs every search."""
    oracle = None

    def find_substreams(self, patterns, window_centers, window_sizes, with_index=False):
        O = type(self).oracle
        scores, times, positions = [], [], []
        for p, c, w in zip(patterns, window_centers, window_sizes):
            start_time, lo, n_pos = self._window(p.shape[1], c, w)
            res = O.match_template(self.data[:, lo:lo + n_pos + p.shape[1] - 1], p)[0]
            k = int(res.argmin())
            scores.append(res[k]); times.append(start_time + k / float(self.sample_rate)); positions.append(lo + k)
        out = (np.array(scores, np.float32), times)
        return out + (positions,) if with_index else out


def _streams(g, cls):
    case = g["case"]
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
    dst = cls.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=case["sample_type"])
    src = cls.from_samples(src_pcm, rate, sample_rate=rate, sample_type=case["sample_type"])
    assert hashlib.sha256(dst.data.tobytes()).hexdigest() == g["dst_sha256"]
    assert hashlib.sha256(src.data.tobytes()).hexdigest() == g["src_sha256"]
    assert len(spans) == g["n_spans"]
    return src, dst, [ScriptEvent(s, e, source_index=k) for k, (s, e) in enumerate(spans)]


class _CallLog(object):
    """Wraps a destination stream: records (pattern offset, length, centre, window, diff, time) per call."""

    def __init__(self, dst, src):
        self.dst, self.calls = dst, []
        self._base = src.data.__array_interface__["data"][0]

    def __getattr__(self, name):
        return getattr(self.dst, name)

    def find_substream(self, pattern, centre, size):
        diff, t = self.dst.find_substream(pattern, centre, size)
        off = (pattern.__array_interface__["data"][0] - self._base) // pattern.itemsize
        self.calls.append([int(off), int(pattern.shape[1]), float(centre), float(size), float(diff), float(t)])
        return diff, t

    def find_substreams(self, patterns, centres, sizes):
        # (the triple of sushi.py:450-452 arrives as ONE call of three searches, in the reference's order)
        diffs, times = self.dst.find_substreams(patterns, centres, sizes)
        for pattern, centre, size, diff, t in zip(patterns, centres, sizes, diffs, times):
            off = (pattern.__array_interface__["data"][0] - self._base) // pattern.itemsize
            self.calls.append([int(off), int(pattern.shape[1]), float(centre), float(size), float(diff), float(t)])
        return diffs, times


@pytest.mark.parametrize("g", STREAMS, ids=[g["name"] for g in STREAMS])
def test_streams_state_machine_equals_reference_on_cpu(g, oracle, caplog, monkeypatch):
    monkeypatch.setenv("SUSHI_HIP_LOAD", "host")
    OracleBackedStream.oracle = oracle
    case = g["case"]
    src, dst, events = _streams(g, OracleBackedStream)
    log = _CallLog(dst, src)
    _, lines = _logged(caplog, calculate_shifts, src, log, [[e] for e in events], case["window"], case["max_window"],
                       case["rewind"])
    assert log.calls == g["calls"]
    assert _dump(events) == g["events"]
    assert lines == g["log"]
    # the speculative batched form answers the same questions from a few launches: identical results
    ev2 = [ScriptEvent(e.start, e.end, source_index=e.source_index) for e in events]
    proxy = calculate_shifts_batched(src, dst, [[e] for e in ev2], case["window"], case["max_window"], case["rewind"],
                                     lookahead=8)
    assert _dump(ev2) == g["events"]
    assert proxy.requests == len(g["calls"]) and proxy.launches < len(g["calls"])


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [False, True], ids=["sequential", "speculative"])
@pytest.mark.parametrize("g", STREAMS, ids=[g["name"] for g in STREAMS])
def test_streams_state_machine_equals_reference_on_gpu(g, batched, monkeypatch):
    """The HIP path under the restated state machine gives what the reference state machine gave over the oracle:
    the same positions (times equal), links, and diffs (uint8: the same float32 bits; float32: within the score
    tolerance 1e-4 * diff + 2.5e-7)."""
    case = g["case"]
    src, dst, events = _streams(g, WavStream)
    groups = [[e] for e in events]
    if batched:
        calculate_shifts_batched(src, dst, groups, case["window"], case["max_window"], case["rewind"], lookahead=8)
    else:
        calculate_shifts(src, dst, groups, case["window"], case["max_window"], case["rewind"])
    for got, want in zip(_dump(events), g["events"]):
        assert got["linked_to"] == want["linked_to"]
        assert got["shift"] == want["shift"]
        if case["sample_type"] == "uint8":
            assert np.float32(got["diff"]) == np.float32(want["diff"])
        else:
            assert abs(got["diff"] - want["diff"]) <= 1e-4 * want["diff"] + 2.5e-7
