"""calculate_shifts (reference sushi.py:400-508 restated) and its speculative batched form.

CPU tests run the state machine on WavStreams whose matching is done by the oracle (the product's
host logic with a stand-in backend); the `gpu` tests run the same scenarios on the HIP path and
compare with the oracle-backed run."""
import logging

import numpy as np
import pytest

from sushi_amd import synth
from sushi_amd.shifts import (ScriptEvent, SpeculativeStream, calculate_shifts, calculate_shifts_batched)
from sushi_amd.wav import WavStream


class OracleBackedStream(WavStream):
    """WavStream whose find_substreams is answered by the CPU oracle (no GPU)."""
    oracle = None
    calls = 0
    searches = 0

    def find_substreams(self, patterns, window_centers, window_sizes, with_index=False):
        O = type(self).oracle
        type(self).calls += 1
        type(self).searches += len(patterns)    # (a triple of sushi.py:450-452 arrives as one call of three)
        scores, times, positions = [], [], []
        for p, c, w in zip(patterns, window_centers, window_sizes):
            start_time, lo, n_pos = self._window(p.shape[1], c, w)
            assert n_pos >= 1
            res = O.match_template(self.data[:, lo:lo + n_pos + p.shape[1] - 1], p)[0]
            k = int(res.argmin())
            scores.append(res[k]); times.append(start_time + k / float(self.sample_rate)); positions.append(lo + k)
        out = (np.array(scores, np.float32), times)
        return out + (positions,) if with_index else out


def _scenario(rate, seconds, pieces_s, n_events, sample_type, cls, seed=0, min_len=1.0, max_len=3.0):
    """dst + src with piecewise-constant offsets (chapters) and sorted, non-overlapping events."""
    dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
    pieces = [(int(t * rate), int(round(off * rate))) for t, off in pieces_s]
    src_pcm = synth.make_src_pcm(dst_pcm, pieces, seed=seed + 1)
    dst = cls.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=sample_type)
    src = cls.from_samples(src_pcm, rate, sample_rate=rate, sample_type=sample_type)
    rng = np.random.default_rng(seed + 2)
    starts = np.sort(rng.uniform(8.0, seconds - 12.0, n_events))
    events = []
    for s in starts:
        e = s + float(rng.uniform(min_len, max_len))
        if events and s < events[-1].end + 0.05:
            continue
        events.append(ScriptEvent(float(s), float(e)))

    def true_offset(t):
        off = pieces_s[0][1]
        for start, o in pieces_s:
            if t >= start:
                off = o
        return off
    return src, dst, events, true_offset


def _run(fn, src, dst, events, *args, **kw):
    groups = [[e] for e in events]
    out = fn(src, dst, groups, *args, **kw)
    return [(e.shift, e.diff, e.linked) for e in events], out


def _fresh(events):
    return [ScriptEvent(e.start, e.end) for e in events]


PIECES = [(0.0, 1.5), (40.0, -2.25), (80.0, 4.0)]


@pytest.mark.parametrize("sample_type", ["uint8", "float32"])
def test_speculative_equals_sequential_on_cpu(oracle, sample_type):
    OracleBackedStream.oracle = oracle
    src, dst, events, true_off = _scenario(2000, 120, PIECES, 45, sample_type, OracleBackedStream, seed=3)
    ev_a, ev_b = _fresh(events), _fresh(events)
    OracleBackedStream.calls = OracleBackedStream.searches = 0
    ra, _ = _run(calculate_shifts, src, dst, ev_a, 10, 30, 5)
    sequential_calls = OracleBackedStream.searches
    OracleBackedStream.calls = 0
    rb, proxy = _run(calculate_shifts_batched, src, dst, ev_b, 10, 30, 5, lookahead=16)
    assert ra == rb                                   # bit-identical shifts and diffs, same links
    assert proxy.requests == sequential_calls         # the state machine asked the same questions
    assert proxy.launches == OracleBackedStream.calls
    assert proxy.launches <= 0.35 * sequential_calls  # ... but far fewer launches answered them
    # and the shifts are the planted ones (events that do not straddle a chapter boundary)
    for e in ev_b:
        if all(not (e.start - 4.5 < t < e.end + 4.5) for t, _ in PIECES[1:]):
            assert abs(e.shift - true_off(e.start)) <= 1.0 / 2000 + 1e-9, (e.start, e.shift)


def test_cache_serves_subwindows_only_when_argmin_inside(oracle):
    OracleBackedStream.oracle = oracle
    src, dst, events, _ = _scenario(2000, 60, [(0.0, 1.0)], 6, "float32", OracleBackedStream, seed=5)
    groups = [[e] for e in events]
    proxy = SpeculativeStream(dst, src, groups, lookahead=0)
    pat = src.get_substream(events[2].start, events[2].end)
    c = events[2].start + 1.0
    d1, t1 = proxy.find_substream(pat, c, 10)                # launch
    d2, t2 = proxy.find_substream(pat, c + 0.3, 1.5)         # sub-window containing the match: cache hit
    assert proxy.launches == 1 and proxy.hits == 1 and (d1, t1) == (d2, t2)
    rd, rt = dst.find_substream(pat, c + 0.3, 1.5)
    assert (np.float32(rd), rt) == (np.float32(d2), t2)
    d3, t3 = proxy.find_substream(pat, c + 6.0, 1.5)         # sub-window that excludes the match: new launch
    assert proxy.launches == 2
    rd, rt = dst.find_substream(pat, c + 6.0, 1.5)
    assert (np.float32(rd), rt) == (np.float32(d3), t3)
    # patterns that are not views of the source stream are passed straight through
    d4, t4 = proxy.find_substream(pat.copy(), c, 10)
    assert (np.float32(d4), t4) == (np.float32(d1), t1)


def test_out_of_range_groups_are_linked_and_rewind_widens_window(oracle, caplog):
    """sushi.py:424-429 (groups past the end of the destination are linked to the last shifted event)
    and :471-479 (after rewind_thresh unsettled groups the window grows to max_window and the walk
    restarts at the first uncommitted group)."""
    OracleBackedStream.oracle = oracle
    rate = 2000
    dst_pcm = synth.make_dst_pcm(80, rate, seed=9)
    src_pcm = synth.make_src_pcm(dst_pcm, [(0, int(1.0 * rate)), (30 * rate, int(14.0 * rate))], seed=10)
    dst = OracleBackedStream.from_samples(dst_pcm[:60 * rate], rate, sample_rate=rate, sample_type="float32")
    src = OracleBackedStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type="float32")
    events = [ScriptEvent(float(s), float(s) + 2.0) for s in (10, 14, 18, 22, 33, 36, 39, 42, 70, 74)]
    ev_a, ev_b = _fresh(events), _fresh(events)
    with caplog.at_level(logging.WARNING):
        ra, _ = _run(calculate_shifts, src, dst, ev_a, 5, 20, 2)
    assert any("increasing the window" in r.getMessage() for r in caplog.records)
    rb, proxy = _run(calculate_shifts_batched, src, dst, ev_b, 5, 20, 2, lookahead=8)
    assert ra == rb
    assert abs(ev_a[0].shift - 1.0) < 1e-3 and abs(ev_a[5].shift - 14.0) < 1e-3
    assert ev_a[-1].linked and ev_a[-2].linked                 # start + shift beyond the 60 s destination
    assert ev_a[-1].shift == ev_a[-3].shift


@pytest.mark.gpu
@pytest.mark.parametrize("sample_type", ["uint8", "float32"])
def test_config4_chapters_on_gpu_matches_oracle_run(oracle, sample_type):
    """BASELINE configs[3] in small: per-chapter offsets, the batched state machine on the HIP path
    gives the shifts the oracle-backed sequential run gives (uint8: identical; float32: the same
    positions, diffs within the score tolerance)."""
    OracleBackedStream.oracle = oracle
    pieces = [(0.0, -3.0), (60.0, 1.25), (120.0, 6.5), (180.0, -0.75), (240.0, 3.0)]
    src, dst, events, true_off = _scenario(12000, 300, pieces, 60, sample_type, WavStream, seed=21)
    osrc = OracleBackedStream.__new__(OracleBackedStream); osrc.__dict__.update(src.__dict__)
    odst = OracleBackedStream.__new__(OracleBackedStream); odst.__dict__.update(dst.__dict__)
    ev_gpu, ev_cpu = _fresh(events), _fresh(events)
    r_gpu, proxy = _run(calculate_shifts_batched, src, dst, ev_gpu, 10, 30, 5)
    r_cpu, _ = _run(calculate_shifts_batched, osrc, odst, ev_cpu, 10, 30, 5)
    assert proxy.launches <= 3 * len(pieces)            # a couple of launches per chapter, not one per request
    for (sg, dg, lg), (sc, dc, lc), e in zip(r_gpu, r_cpu, events):
        assert lg == lc
        assert abs(sg - sc) <= 1.0 / 12000 + 1e-9, (e.start, sg, sc)
        if sample_type == "uint8":
            assert sg == sc and np.float32(dg) == np.float32(dc)
        else:
            assert abs(float(dg) - float(dc)) <= 1e-4 * float(dc) + 2.5e-7
    # sequential drop-in calls (one launch per find_substream) give the same as the batched form
    ev_seq = _fresh(events[:12])
    r_seq, _ = _run(calculate_shifts, src, dst, ev_seq, 10, 30, 5)
    assert r_seq == r_gpu[:12]


def test_lookahead_grows_while_the_shift_holds(oracle):
    """A constant offset: the speculation depth doubles after every fully used batch, so 80 groups need
    a handful of launches; results equal the sequential run."""
    OracleBackedStream.oracle = oracle
    src, dst, events, _ = _scenario(2000, 200, [(0.0, 2.5)], 90, "uint8", OracleBackedStream, seed=11, max_len=2.0)
    ev_a, ev_b = _fresh(events), _fresh(events)
    ra, _ = _run(calculate_shifts, src, dst, ev_a, 10, 30, 5)
    rb, proxy = _run(calculate_shifts_batched, src, dst, ev_b, 10, 30, 5, lookahead=4)
    assert ra == rb
    assert len(events) >= 40 and proxy.launches <= 6           # 4 + 8 + 16 + 32 instead of n / 4
    assert proxy.lookahead > 4


def _pipeline(src, dst, events, chapters, window=10, max_window=30, rewind=5, shifts=calculate_shifts_batched):
    """sushi.run's audio path (sushi.py:624-626, 664-670, 682-704): search groups, shifts, grouping."""
    from sushi_amd import grouping
    groups = grouping.prepare_search_groups(events, src.duration_seconds, chapters, 1001.0 / 24000.0 * 10, 2)
    shifts(src, dst, groups, window, max_window, rewind)
    return grouping.group_shifts(events, chapters, smooth_radius=3)


def _config4_events(rng, seconds, n, first=8.0, tail=12.0):
    """A script with what prepare_search_groups has to deal with: dialogue, a comment, a zero-length line,
    an exact duplicate and a run of short typesetting lines."""
    starts = np.sort(rng.uniform(first, seconds - tail, n))
    events = []
    for s in starts:
        if events and s < events[-1].end + 0.05:
            continue
        events.append(ScriptEvent(float(s), float(s + rng.uniform(1.0, 3.0))))
    events.insert(5, ScriptEvent(events[5].start, events[5].end))                      # duplicate of the next line
    events.insert(11, ScriptEvent(events[11].start - 0.01, events[11].start - 0.01))   # zero length
    events.insert(17, ScriptEvent(events[17].start - 0.02, events[17].start + 1.0, is_comment=True))
    t0 = events[23].end + 0.01
    for k in range(4):                                                                  # typesetting: 0.3 s lines 0.1 s apart
        events.insert(24 + k, ScriptEvent(t0 + 0.4 * k, t0 + 0.4 * k + 0.3))
    events.sort(key=lambda e: e.start)
    return events


def test_config4_pipeline_on_cpu_recovers_chapter_offsets(oracle):
    """BASELINE configs[3] end to end with the oracle as the matching backend: every event ends up within one
    sample of its chapter's planted offset, one group per chapter."""
    OracleBackedStream.oracle = oracle
    pieces = [(0.0, -3.0), (50.0, 1.25), (100.0, 6.5)]
    src, dst, _, true_off = _scenario(4000, 150, pieces, 10, "uint8", OracleBackedStream, seed=33)
    events = _config4_events(np.random.default_rng(5), 150, 45)
    chapters = [t for t, _ in pieces]
    groups = _pipeline(src, dst, events, chapters)
    assert len(groups) == len(pieces)
    for e in events:
        if true_off(e.start) == true_off(e.end):                       # a line across a chapter mark follows its END (sushi.py:137)
            assert abs(e.shift - true_off(e.start)) <= 1.5 / 4000, (e.start, e.shift)


@pytest.mark.gpu
@pytest.mark.parametrize("sample_type", ["uint8", "float32"])
def test_config4_pipeline_on_gpu_matches_oracle_run(oracle, sample_type):
    """BASELINE configs[3] end to end: prepare_search_groups -> calculate_shifts (batched, HIP path) -> the
    --grouping block, against the same pipeline with the oracle doing the matching, and against the planted
    per-chapter offsets (five chapters, -30 ... +30 s scaled to the test's stream)."""
    OracleBackedStream.oracle = oracle
    pieces = [(0.0, -6.0), (60.0, -2.4), (120.0, 0.6), (180.0, 3.4), (240.0, 6.0)]
    src, dst, _, true_off = _scenario(12000, 300, pieces, 10, sample_type, WavStream, seed=41)
    osrc = OracleBackedStream.__new__(OracleBackedStream); osrc.__dict__.update(src.__dict__)
    odst = OracleBackedStream.__new__(OracleBackedStream); odst.__dict__.update(dst.__dict__)
    ev_gpu = _config4_events(np.random.default_rng(9), 300, 70)
    ev_cpu = [ScriptEvent(e.start, e.end, is_comment=e.is_comment) for e in ev_gpu]
    chapters = [t for t, _ in pieces]
    g_gpu = _pipeline(src, dst, ev_gpu, chapters)
    g_cpu = _pipeline(osrc, odst, ev_cpu, chapters)
    assert [len(g) for g in g_gpu] == [len(g) for g in g_cpu] and len(g_gpu) == len(pieces)
    for a, b in zip(ev_gpu, ev_cpu):
        assert a.linked == b.linked
        if sample_type == "uint8":
            assert a.shift == b.shift                                   # same positions, same scores: same averages
        else:
            assert abs(a.shift - b.shift) <= 1e-6                       # score-weighted averages of identical positions
        if true_off(a.start) == true_off(a.end):                       # a line across a chapter mark follows its END (sushi.py:137)
            assert abs(a.shift - true_off(a.start)) <= 1.5 / 12000, (a.start, a.shift)


@pytest.mark.gpu
def test_config4_stated_sizes_on_gpu_matches_oracle_run(oracle):
    """BASELINE configs[3] at the sizes SURVEY 8(d) states: five chapters with offsets -30, -12, +3, +17, +30 s,
    12 kHz streams, small window 10 s, maximum window 60 s (so that every jump between chapters is reachable),
    through prepare_search_groups -> calculate_shifts (batched, HIP path) -> the --grouping block; against the same
    pipeline with the oracle doing the matching and against the planted offsets."""
    OracleBackedStream.oracle = oracle
    offsets = (-30.0, -12.0, 3.0, 17.0, 30.0)
    chapter_s = 300.0
    pieces = [(k * chapter_s, off) for k, off in enumerate(offsets)]
    seconds = int(chapter_s * len(offsets))
    src, dst, _, true_off = _scenario(12000, seconds, pieces, 10, "float32", WavStream, seed=43)
    osrc = OracleBackedStream.__new__(OracleBackedStream); osrc.__dict__.update(src.__dict__)
    odst = OracleBackedStream.__new__(OracleBackedStream); odst.__dict__.update(dst.__dict__)
    ev_gpu = _config4_events(np.random.default_rng(10), seconds, 140, first=36.0, tail=48.0)
    ev_cpu = [ScriptEvent(e.start, e.end, is_comment=e.is_comment) for e in ev_gpu]
    chapters = [t for t, _ in pieces]
    g_gpu = _pipeline(src, dst, ev_gpu, chapters, window=10, max_window=60)
    g_cpu = _pipeline(osrc, odst, ev_cpu, chapters, window=10, max_window=60, shifts=calculate_shifts)   # sequential: no speculated searches to pay for
    assert [len(g) for g in g_gpu] == [len(g) for g in g_cpu] and len(g_gpu) == len(pieces)
    for a, b in zip(ev_gpu, ev_cpu):
        assert a.linked == b.linked
        assert abs(a.shift - b.shift) <= 1e-6                           # score-weighted averages of identical positions
        if true_off(a.start) == true_off(a.end):
            assert abs(a.shift - true_off(a.start)) <= 1.5 / 12000, (a.start, a.shift)
