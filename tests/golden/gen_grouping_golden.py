#!/usr/bin/env python3
"""Generate tests/golden/grouping.json by executing the REFERENCE's own grouping code.

Runs only in the dev container (needs /root/reference).  `sushi.py` does not import under Python 3
(`from itertools import izip`, cv2 behind `wav`), but the functions around `calculate_shifts` --
    smooth_events / running_median        sushi.py:97-117
    detect_groups                         sushi.py:120-127
    groups_from_chapters                  sushi.py:130-161
    split_broken_groups                   sushi.py:164-187
    fix_near_borders                      sushi.py:190-215
    average_shifts                        sushi.py:309-316
    merge_short_lines_into_groups         sushi.py:319-349
    prepare_search_groups                 sushi.py:352-397
-- use no Python-2-only *syntax*, only Python-2 names (izip, xrange, unicode, list-returning filter).
Their source text is read from the reference file at run time and exec'd unmodified in a namespace
that supplies those names; the events are the reference's own `subs.ScriptEventBase`.

The cases replay what `run()` does with them (sushi.py:624-626, 682-704): search-group preparation on
a script with comments / zero-length / duplicate / out-of-range / short lines, and the grouping block
on events whose shifts follow per-chapter offsets with noise, border outliers and one broken chapter.
"""
import json
import logging
import os
import re
import sys
from itertools import takewhile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "grouping.json")
WANTED = ["abs_diff", "running_median", "smooth_events", "detect_groups", "groups_from_chapters",
          "split_broken_groups", "fix_near_borders", "average_shifts", "merge_short_lines_into_groups",
          "prepare_search_groups"]


def load_reference_functions():
    sys.path.insert(0, REF)
    import common as refcommon      # noqa: E402  imports under Python 3
    import subs as refsubs          # noqa: E402
    text = open(os.path.join(REF, "sushi.py")).read()
    ns = {
        "np": np, "logging": logging, "takewhile": takewhile,
        "izip": zip, "xrange": range, "unicode": str,
        "filter": lambda f, it: list(__import__("builtins").filter(f, it)),      # Python 2's filter returns a list
        "SushiError": refcommon.SushiError, "format_time": refcommon.format_time,
        "ensure_static_collection": refcommon.ensure_static_collection,
        "ALLOWED_ERROR": 0.01, "MAX_GROUP_STD": 0.025,          # sushi.py:39-40
    }
    for name in WANTED:
        m = re.search(r"^def %s\(.*?(?=^\S)" % name, text, re.S | re.M)
        assert m, name
        exec(compile(m.group(0), "reference:sushi.py:" + name, "exec"), ns)
    for const, value in (("ALLOWED_ERROR", 0.01), ("MAX_GROUP_STD", 0.025)):
        assert re.search(r"^%s = %s$" % (const, value), text, re.M), const
    return ns, refsubs


def make_event_class(refsubs):
    """The reference's ScriptEventBase plus the one attribute prepare_search_groups reads from the
    concrete script classes (is_comment, subs.py)."""
    class RefEvent(refsubs.ScriptEventBase):
        def __init__(self, idx, start, end, is_comment=False):
            super(RefEvent, self).__init__(idx, start, end, u"line %d" % idx)
            self.is_comment = is_comment

        def __str__(self):
            return u"event %d" % self.source_index
    return RefEvent


def dump_events(events):
    index = {id(e): k for k, e in enumerate(events)}
    return [{"shift": float(e.shift), "diff": float(e.diff),
             "linked_to": index[id(e._linked_event)] if e.linked else None} for e in events]


def dump_groups(groups, events):
    index = {id(e): k for k, e in enumerate(events)}
    return [[index[id(e)] for e in g] for g in groups]


def case_prepare(ns, RefEvent, rng, n, duration, chapters, max_ts_duration, max_ts_distance):
    spans, t = [], 10.0
    for k in range(n):
        kind = rng.random()
        t += float(rng.uniform(0.0, 6.0))
        if kind < 0.10:
            spans.append((t, t, False))                         # zero duration
        elif kind < 0.20:
            spans.append((t, t + float(rng.uniform(0.5, 4.0)), True))   # comment
        elif kind < 0.45:
            d = float(rng.uniform(0.05, 0.5))                   # short (typesetting) lines close together
            spans.append((t, t + d, False))
            t -= float(rng.uniform(0.0, 0.04))
        else:
            spans.append((t, t + float(rng.uniform(1.0, 5.0)), False))
        if kind > 0.93 and spans:
            spans.append(spans[-1])                             # exact duplicate of the previous span
    spans.sort(key=lambda s: s[0])
    events = [RefEvent(i, s, e, c) for i, (s, e, c) in enumerate(spans)]
    groups = ns["prepare_search_groups"](events, duration, list(chapters), max_ts_duration, max_ts_distance)
    return {"kind": "prepare", "spans": [[s, e, bool(c)] for s, e, c in spans], "source_duration": duration,
            "chapters": list(chapters), "max_ts_duration": max_ts_duration, "max_ts_distance": max_ts_distance,
            "groups": dump_groups(groups, events), "events": dump_events(events)}


def case_grouping(ns, RefEvent, rng, n, chapters, offsets, use_chapters, smooth_radius, break_chapter=None):
    dur = chapters[-1] + 300.0 if chapters else 1500.0
    starts = np.sort(rng.uniform(5.0, dur - 10.0, n))
    events = [RefEvent(i, float(s), float(s + rng.uniform(1.0, 5.0))) for i, s in enumerate(starts)]
    bounds = list(chapters[1:]) + [1e18]
    shifts, diffs, linked = [], [], []
    for e in events:
        c = 0
        while e.end > bounds[c]:
            c += 1
        sh = offsets[c] + float(rng.normal(0.0, 0.002))
        if break_chapter is not None and c == break_chapter and e.start > (chapters[c] + bounds[c]) / 2 - 40:
            sh += 0.8                                           # the chapter list is wrong: a second offset inside
        df = float(rng.uniform(0.01, 0.05))
        if rng.random() < 0.04:
            sh += float(rng.uniform(-3, 3)); df = float(rng.uniform(0.4, 0.9))   # a failed search somewhere
        shifts.append(sh); diffs.append(df)
    for k in (0, 1, n - 1):                                     # bad lines at the borders
        diffs[k] = 0.7; shifts[k] += 2.0
    last = None
    for k, e in enumerate(events):
        if last is not None and rng.random() < 0.08:
            e.link_event(last); linked.append(k)
        else:
            e.set_shift(shifts[k], diffs[k]); last = e
    inputs = {"spans": [[e.start, e.end] for e in events], "shifts": shifts, "diffs": diffs,
              "linked": [[k, events.index(events[k]._linked_event)] for k in linked]}
    # sushi.py:682-704
    if use_chapters and chapters:
        groups = ns["groups_from_chapters"](events, list(chapters))
        for g in groups:
            ns["fix_near_borders"](g)
            ns["smooth_events"]([x for x in g if not x.linked], smooth_radius)
        groups = ns["split_broken_groups"](groups)
    else:
        ns["fix_near_borders"](events)
        ns["smooth_events"]([x for x in events if not x.linked], smooth_radius)
        groups = ns["detect_groups"](events)
    before_avg = dump_events(events)
    averages = [float(ns["average_shifts"](g)) for g in groups]
    return {"kind": "grouping", "inputs": inputs, "chapters": list(chapters), "use_chapters": bool(use_chapters),
            "smooth_radius": smooth_radius, "groups": dump_groups(groups, events), "events_before_average": before_avg,
            "averages": averages, "events": dump_events(events)}


def main():
    logging.disable(logging.CRITICAL)
    ns, refsubs = load_reference_functions()
    RefEvent = make_event_class(refsubs)
    rng = np.random.default_rng(20260924)
    cases = []
    for n, dur, ch in [(40, 200.0, [0.0]), (120, 420.0, [0.0, 150.0, 300.0]), (150, 520.0, [0.0, 120.0, 250.0, 400.0])]:
        for mtd, mdist in [(0.42, 1.0), (1.0, 0.05)]:
            cases.append(case_prepare(ns, RefEvent, rng, n, dur, ch, mtd, mdist))
    five = [0.0, 300.0, 600.0, 900.0, 1200.0]
    offs = [-30.0, -12.0, 3.0, 17.0, 30.0]                      # BASELINE configs[3]
    for radius in (0, 3):
        cases.append(case_grouping(ns, RefEvent, rng, 160, five, offs, True, radius))
        cases.append(case_grouping(ns, RefEvent, rng, 160, five, offs, False, radius))
        cases.append(case_grouping(ns, RefEvent, rng, 140, five, offs, True, radius, break_chapter=2))
    cases.append(case_grouping(ns, RefEvent, rng, 60, [], [1.5], False, 3))
    cases.append(case_grouping(ns, RefEvent, rng, 25, [0.0, 400.0], [0.0, 0.004], True, 1))
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/gen_grouping_golden.py", "reference": "sushi.py (functions exec'd from source)",
                   "cases": cases}, f, separators=(",", ":"))
    print(OUT, len(cases), "cases", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
