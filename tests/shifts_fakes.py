"""Scripted stand-ins for the two streams ``calculate_shifts`` talks to (TEST INFRASTRUCTURE).

Used by tests/golden/gen_calculate_shifts_golden.py (which drives the REFERENCE's own
``calculate_shifts`` source, sushi.py:400-508, with them) and by tests/test_shifts_golden.py (which
drives ``sushi_amd.shifts.calculate_shifts`` with the same script and compares call by call).

``FakeSource.get_substream`` returns a (1, M) float64 view whose values are the absolute sample
numbers, so that ``np.split`` halves (sushi.py:445) still say where in the source they start.
``FakeDestination.find_substream`` answers from a script: a piecewise-constant "true" shift, spans of
the source where nothing matches (the answer is then a deterministic pseudo-random position of the
window), and spans where only one half of the pattern matches.  Every call is logged.
"""
import math

import numpy as np


class FakeSource(object):
    def __init__(self, sample_rate, seconds):
        self.sample_rate = sample_rate
        self._row = np.arange(int(sample_rate * seconds), dtype=np.float64).reshape(1, -1)

    def get_substream(self, start, end):
        return self._row[:, int(self.sample_rate * start):int(self.sample_rate * end)]


def _frac(*xs):
    """Deterministic pseudo-random number in [0, 1) from a few floats/ints (no RNG state)."""
    h = 0.0
    for k, x in enumerate(xs):
        h += math.sin(float(x) * (12.9898 + 78.233 * k)) * 43758.5453
    return h - math.floor(h)


class FakeDestination(object):
    """script = {"duration": s, "shifts": [[from_time, shift], ...], "dead": [[t0, t1], ...],
                 "half_dead": [[t0, t1], ...], "sample_rate": sr}"""

    def __init__(self, script):
        self.script = script
        self.sample_rate = script["sample_rate"]
        self.duration_seconds = script["duration"]
        self.calls = []

    def _true_shift(self, t):
        shift = self.script["shifts"][0][1]
        for t0, s in self.script["shifts"]:
            if t >= t0:
                shift = s
        return shift

    def find_substream(self, pattern, window_center, window_size):
        sr = float(self.sample_rate)
        first, length = int(pattern[0][0]), int(len(pattern[0]))
        self.calls.append([first, length, float(window_center), float(window_size)])
        t_pat = first / sr
        start_time = max(min(window_center - window_size, self.duration_seconds), -10)
        n_pos = max(int(round(2 * window_size * sr)), 1)
        dead = any(t0 <= t_pat < t1 for t0, t1 in self.script.get("dead", ()))
        # a "half dead" span kills searches whose pattern starts inside it; a whole pattern that merely
        # starts before it still matches, so left / right / whole disagree there
        half_dead = any(t0 <= t_pat < t1 for t0, t1 in self.script.get("half_dead", ())) and \
            length < self.script.get("half_len", 0) * sr
        ideal = t_pat + self._true_shift(t_pat)
        k = int(round((ideal - start_time) * sr))
        if not dead and not half_dead and 0 <= k < n_pos:
            return np.float32(0.02 + 0.01 * _frac(first, length)), start_time + k / sr
        k = int(_frac(first, length, window_center, window_size) * n_pos)
        return np.float32(0.5 + 0.4 * _frac(first, window_center)), start_time + k / sr


SCENARIOS = [
    # constant shift: every group commits through the small window (sushi.py:431-443)
    {"name": "constant", "starts": [5 + 4 * k for k in range(12)], "length": 2.0, "window": 10, "max_window": 30,
     "rewind": 5, "script": {"duration": 80.0, "sample_rate": 1000, "shifts": [[0.0, 1.25]]}},
    # shift steps inside the normal window: small window misses, triple search settles (:445-455, :481-493)
    {"name": "steps", "starts": [5 + 4 * k for k in range(20)], "length": 2.0, "window": 10, "max_window": 30,
     "rewind": 5, "script": {"duration": 120.0, "sample_rate": 1000, "shifts": [[0.0, -2.0], [30.0, 3.5], [60.0, 3.5075]]}},
    # a dead span shorter than rewind_thresh: groups pile up as uncommitted, then "back on track" (:468-470, :481-493)
    {"name": "dead-short", "starts": [5 + 4 * k for k in range(20)], "length": 2.0, "window": 10, "max_window": 30,
     "rewind": 5, "script": {"duration": 120.0, "sample_rate": 1000, "shifts": [[0.0, 0.5]], "dead": [[20.0, 31.0]]}},
    # a dead span longer than rewind_thresh: window grows to max_window and the walk rewinds (:471-479);
    # the shift behind it is out of reach of the normal window but not of max_window
    {"name": "rewind", "starts": [5 + 3 * k for k in range(30)], "length": 2.0, "window": 10, "max_window": 30,
     "rewind": 4, "script": {"duration": 150.0, "sample_rate": 1000, "shifts": [[0.0, 1.0], [40.0, 19.0]],
                             "dead": [[38.0, 52.0]]}},
    # shift jumps beyond the normal window: searches around the committed shift fail, the retry around the
    # last uncommitted shift (:457-465) is what gets used once something was found by chance or by rewind
    {"name": "jump", "starts": [5 + 3 * k for k in range(30)], "length": 2.5, "window": 6, "max_window": 40,
     "rewind": 3, "script": {"duration": 150.0, "sample_rate": 1000, "shifts": [[0.0, 0.0], [30.0, 25.0], [70.0, -4.0]]}},
    # halves disagree with the whole pattern (:453, :463): uncommitted until they agree again
    {"name": "halves", "starts": [5 + 4 * k for k in range(18)], "length": 3.0, "window": 10, "max_window": 30,
     "rewind": 6, "script": {"duration": 120.0, "sample_rate": 1000, "shifts": [[0.0, 2.0], [24.0, 4.0]],
                             "half_dead": [[24.0, 40.0]], "half_len": 2.0}},
    # groups past the end of the destination: "outside of audio range" + linking (:424-429, :498-505)
    {"name": "past-end", "starts": [5 + 4 * k for k in range(15)], "length": 2.0, "window": 10, "max_window": 30,
     "rewind": 5, "script": {"duration": 40.0, "sample_rate": 1000, "shifts": [[0.0, 3.0]]}},
    # window not larger than the small window: no probe, straight to the triple search (:431)
    {"name": "tiny-window", "starts": [5 + 4 * k for k in range(10)], "length": 2.0, "window": 1, "max_window": 30,
     "rewind": 5, "script": {"duration": 80.0, "sample_rate": 1000, "shifts": [[0.0, 0.4], [20.0, 0.9]]}},
    # ends unsettled: trailing uncommitted states keep their own shifts (:495-496, chain at :498)
    {"name": "tail-unsettled", "starts": [5 + 4 * k for k in range(12)], "length": 2.0, "window": 10, "max_window": 10,
     "rewind": 3, "script": {"duration": 80.0, "sample_rate": 1000, "shifts": [[0.0, 1.0]], "dead": [[40.0, 80.0]]}},
    # multi-event groups (typesetting merged by prepare_search_groups): span = first start .. last end
    {"name": "groups-of-three", "starts": [5 + 2 * k for k in range(24)], "length": 1.5, "window": 10, "max_window": 30,
     "rewind": 5, "group_size": 3,
     "script": {"duration": 120.0, "sample_rate": 1000, "shifts": [[0.0, -1.0], [25.0, 2.0]], "dead": [[33.0, 36.0]]}},
]
