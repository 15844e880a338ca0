"""sushi_amd/grouping.py against tests/golden/grouping.json -- outputs of the reference's own function
bodies (tests/golden/gen_grouping_golden.py): search-group preparation (sushi.py:352-397) and the
--grouping block (sushi.py:682-704), the host code either side of calculate_shifts."""
import json
import os

import numpy as np
import pytest

from sushi_amd import grouping
from sushi_amd.shifts import ScriptEvent

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "grouping.json")) as f:
        return json.load(f)["cases"]


def _state(events):
    index = {id(e): k for k, e in enumerate(events)}
    return [(float(e.shift), float(e.diff), index[id(e._linked_event)] if e.linked else None) for e in events]


def _groups(groups, events):
    index = {id(e): k for k, e in enumerate(events)}
    return [[index[id(e)] for e in g] for g in groups]


def _expect(dumped):
    return [(d["shift"], d["diff"], d["linked_to"]) for d in dumped]


def test_prepare_search_groups_matches_reference(golden):
    cases = [c for c in golden if c["kind"] == "prepare"]
    assert len(cases) >= 6
    for c in cases:
        events = [ScriptEvent(s, e, source_index=i, is_comment=com) for i, (s, e, com) in enumerate(c["spans"])]
        groups = grouping.prepare_search_groups(events, c["source_duration"], c["chapters"], c["max_ts_duration"],
                                                c["max_ts_distance"])
        assert _groups(groups, events) == c["groups"]
        assert _state(events) == _expect(c["events"])            # who follows whom, bit for bit


def test_grouping_block_matches_reference(golden):
    cases = [c for c in golden if c["kind"] == "grouping"]
    assert len(cases) >= 8
    for c in cases:
        inp = c["inputs"]
        events = [ScriptEvent(s, e, source_index=i) for i, (s, e) in enumerate(inp["spans"])]
        follows = dict((k, t) for k, t in inp["linked"])
        for k, e in enumerate(events):
            if k in follows:
                e.link_event(events[follows[k]])
            else:
                e.set_shift(inp["shifts"][k], inp["diffs"][k])
        # the block of sushi.py:682-704 in two steps, to compare the state before the averaging as well
        if c["use_chapters"] and c["chapters"]:
            groups = grouping.groups_from_chapters(events, c["chapters"])
            for g in groups:
                grouping.fix_near_borders(g)
                grouping.smooth_events([x for x in g if not x.linked], c["smooth_radius"])
            groups = grouping.split_broken_groups(groups)
        else:
            grouping.fix_near_borders(events)
            grouping.smooth_events([x for x in events if not x.linked], c["smooth_radius"])
            groups = grouping.detect_groups(events)
        assert _groups(groups, events) == c["groups"]
        assert _state(events) == _expect(c["events_before_average"])
        averages = [float(grouping.average_shifts(g)) for g in groups]
        assert averages == c["averages"]                         # same NumPy calls in the same order: bit-identical
        assert _state(events) == _expect(c["events"])


def test_group_shifts_is_the_whole_block(golden):
    c = next(c for c in golden if c["kind"] == "grouping" and c["use_chapters"] and c["smooth_radius"] == 3)
    inp = c["inputs"]
    events = [ScriptEvent(s, e, source_index=i) for i, (s, e) in enumerate(inp["spans"])]
    follows = dict((k, t) for k, t in inp["linked"])
    for k, e in enumerate(events):
        if k in follows:
            e.link_event(events[follows[k]])
        else:
            e.set_shift(inp["shifts"][k], inp["diffs"][k])
    groups = grouping.group_shifts(events, c["chapters"], smooth_radius=3)
    assert _groups(groups, events) == c["groups"]
    assert _state(events) == _expect(c["events"])


def test_running_median_window_shrinks_at_the_ends():
    v = [5.0, 1.0, 9.0, 2.0, 8.0, 3.0, 7.0]
    assert grouping.running_median(v, 1) == v
    assert grouping.running_median(v, 5) == [5.0, 5.0, 5.0, 3.0, 7.0, 7.0, 7.0]
    with pytest.raises(Exception):
        grouping.running_median(v, 4)


def test_detect_groups_cuts_at_allowed_error():
    ev = [ScriptEvent(float(k), k + 1.0) for k in range(5)]
    for e, s in zip(ev, [0.0, 0.009, 0.0195, 0.5, 0.5099]):
        e.set_shift(s, 0.1)
    assert [len(g) for g in grouping.detect_groups(ev)] == [2, 1, 2]       # 0.009 -> 0.0195 is 0.0105 > 0.01
