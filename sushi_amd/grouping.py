"""What ``sushi.run`` does around ``calculate_shifts``: choosing what to search for
(``prepare_search_groups``, reference sushi.py:352-397, called at :624-626) and turning the per-event
shifts the searches produced into per-group shifts (the ``--grouping`` block, sushi.py:682-704).

Host-side, O(events) list work -- restated for Python 3 so that BASELINE configs[3] (per-chapter
offsets, grouped search) runs end to end on GPU results.  Behaviour is pinned by
``tests/golden/grouping.json``, which ``tests/golden/gen_grouping_golden.py`` produces by executing the
reference's own function bodies; every function below names the lines it restates.

Events are anything with the ``subs.ScriptEventBase`` surface (subs.py:14-80): ``start``, ``end``,
``duration``, ``shift``, ``diff``, ``linked``, ``set_shift``, ``link_event``, ``get_link_chain_end`` and,
for ``prepare_search_groups``, ``is_comment`` -- e.g. ``sushi_amd.shifts.ScriptEvent``.
"""
import logging

import numpy as np

from .common import SushiError, format_time

ALLOWED_ERROR = 0.01      # sushi.py:39  seconds: two shifts closer than this are "the same"
MAX_GROUP_STD = 0.025     # sushi.py:40  a chapter whose shifts spread more than this is not one group
_FAR_FUTURE = 36000000000  # sushi.py:133: the closing pseudo-chapter


def running_median(values, window_size):
    """sushi.py:97-107: centred median whose window shrinks symmetrically towards both ends."""
    if window_size % 2 != 1:
        raise SushiError('Median window size should be odd')
    half = window_size // 2
    n = len(values)
    out = []
    for k in range(n):
        r = min(half, k, n - 1 - k)
        out.append(np.median(values[k - r:k + r + 1]))
    return out


def smooth_events(events, radius):
    """sushi.py:110-117: replace every shift by the running median over 2*radius+1 events (diffs kept)."""
    if not radius:
        return
    smoothed = running_median([e.shift for e in events], 2 * radius + 1)
    for e, s in zip(events, smoothed):
        e.set_shift(s, e.diff)


def detect_groups(events):
    """sushi.py:120-127: cut the sequence wherever consecutive shifts differ by more than ALLOWED_ERROR."""
    groups = []
    for e in events:
        if not groups or abs(e.shift - groups[-1][-1].shift) > ALLOWED_ERROR:
            groups.append([])
        groups[-1].append(e)
    if not groups:
        # the reference lets next() on the empty iterator raise StopIteration here (sushi.py:122); an exception of
        # our own cannot be swallowed by an enclosing for loop or generator
        raise SushiError('No events to group')
    return groups


def groups_from_chapters(events, times):
    """sushi.py:130-161: one group per chapter (an event belongs to the chapter its END falls in); a chapter
    made of linked events only is dissolved into the groups of the events they follow."""
    logging.info(u'Chapter start points: {0}'.format([format_time(t) for t in times]))
    limits = iter(list(times[1:]) + [_FAR_FUTURE])
    limit = next(limits)
    groups = [[]]
    for e in events:
        if e.end > limit:
            groups.append([])
            while e.end > limit:
                limit = next(limits)
        groups[-1].append(e)
    groups = [g for g in groups if g]
    orphaned = [g for g in groups if all(e.linked for e in g)]
    if orphaned:
        for g in orphaned:
            for e in g:
                target = e.get_link_chain_end()
                home = next(h for h in groups if any(x is target for x in h))
                home.append(e)
            del g[:]
        groups = [g for g in groups if g]
        for g in groups:
            g.sort(key=lambda e: e.start)
    return groups


def split_broken_groups(groups):
    """sushi.py:164-187: a chapter whose shifts are inconsistent falls back to detect_groups; after that,
    neighbouring pieces that agree (last/first shift and joint spread) are glued together again."""
    pieces = []
    any_broken = False
    for g in groups:
        spread = np.std([e.shift for e in g])
        if spread > MAX_GROUP_STD:
            logging.warning(u'Shift is not consistent between {0} and {1}, most likely chapters are wrong (std: {2}). '
                            u'Switching to automatic grouping.'.format(format_time(g[0].start), format_time(g[-1].end),
                                                                       spread))
            pieces.extend(detect_groups(g))
            any_broken = True
        else:
            pieces.append(g)
    if not any_broken:
        return pieces
    merged = [list(pieces[0])]
    for g in pieces[1:]:
        apart = abs(merged[-1][-1].shift - g[0].shift) >= ALLOWED_ERROR
        if apart or np.std([e.shift for e in g + merged[-1]]) >= MAX_GROUP_STD:
            merged.append([])
        merged[-1].extend(g)
    return merged


def fix_near_borders(events):
    """sushi.py:190-215: at either end of the list, events whose diff is out of proportion (not within
    (0.2, 5) times the smaller of the overall median diff and the median of the first ten) follow the first
    sane event after them."""
    def fix_border(seq, median_diff):
        limit = min(np.median([x.diff for x in seq[:10]]), median_diff)
        broken = []
        for e in seq:
            if 0.2 < (e.diff / limit) < 5:
                for b in broken:
                    b.link_event(e)
                return len(broken)
            broken.append(e)
        return 0

    median_diff = np.median([x.diff for x in events])
    fixed = fix_border(events, median_diff)
    if fixed:
        logging.info('Fixing {0} border events right after {1}'.format(fixed, format_time(events[0].start)))
    fixed = fix_border(list(reversed(events)), median_diff)
    if fixed:
        logging.info('Fixing {0} border events right before {1}'.format(fixed, format_time(events[-1].end)))


def average_shifts(events):
    """sushi.py:309-316: every unlinked event of the group takes the (1 - diff)-weighted mean shift."""
    own = [e for e in events if not e.linked]
    avg = np.average([e.shift for e in own], weights=[1 - e.diff for e in own])
    for e in own:
        e.set_shift(avg, e.diff)
    return avg


def merge_short_lines_into_groups(events, chapter_times, max_ts_duration, max_ts_distance):
    """sushi.py:319-349: a line longer than max_ts_duration is searched for on its own; a short one collects the
    short lines that start within max_ts_distance of the running end of its group and end before the next
    chapter (typesetting: many short overlapping lines are one template)."""
    events = list(events)
    limits = iter(list(chapter_times[1:]) + [100000000])
    next_chapter = next(limits)
    taken = set()
    out = []
    for k, e in enumerate(events):
        if k in taken:
            continue
        while e.end > next_chapter:
            next_chapter = next(limits)
        if e.duration > max_ts_duration:
            out.append([e])
            taken.add(k)
            continue
        group, group_end = [e], e.end
        i = k + 1
        while i < len(events) and abs(group_end - events[i].start) < max_ts_distance:
            cand = events[i]
            if cand.end < next_chapter and cand.duration <= max_ts_duration:
                taken.add(i)
                group.append(cand)
                group_end = max(group_end, cand.end)
            i += 1
        out.append(group)
    return out


def prepare_search_groups(events, source_duration, chapter_times, max_ts_duration, max_ts_distance):
    """sushi.py:352-397: link what will not be searched for (comments and zero-length lines to the next event,
    lines centred past the end of the audio to the last searched one, exact duplicates to their first
    occurrence), merge short lines, and link groups that lie inside an earlier group to it."""
    last_own = None
    for k, e in enumerate(events):
        follows_next = False
        if e.is_comment:
            follows_next = True
        elif (e.start + e.duration / 2.0) > source_duration:
            logging.info('Event time outside of audio range, ignoring: %s' % e)
            e.link_event(last_own)
            continue
        elif e.end == e.start:
            logging.info('{0}: skipped because zero duration'.format(format_time(e.start)))
            follows_next = True
        if follows_next:
            e.link_event(events[k + 1] if k + 1 < len(events) else last_own)
            continue
        twin = None
        for x in reversed(events[:k]):           # scripts are sorted by start: only the run of equal starts
            if x.start != e.start:
                break
            if not x.linked and x.end == e.end:
                twin = x
                break
        if twin is not None:
            e.link_event(twin)
        else:
            last_own = e
    groups = merge_short_lines_into_groups([e for e in events if not e.linked], chapter_times, max_ts_duration,
                                           max_ts_distance)
    kept = []
    for k, g in enumerate(groups):
        outer = next((x for x in reversed(groups[:k]) if x[0].start <= g[0].start and x[-1].end >= g[-1].end), None)
        if outer is None:
            kept.append(g)
        else:
            for e in g:
                e.link_event(outer[0])
    return kept


def group_shifts(events, chapter_times=None, smooth_radius=3):
    """The ``--grouping`` block of sushi.run (sushi.py:682-704; --smooth-radius defaults to 3, :755): returns the groups,
    every unlinked event of a group carrying the group's weighted mean shift afterwards."""
    if chapter_times:
        groups = groups_from_chapters(events, chapter_times)
        for g in groups:
            fix_near_borders(g)
            smooth_events([x for x in g if not x.linked], smooth_radius)
        groups = split_broken_groups(groups)
    else:
        fix_near_borders(events)
        smooth_events([x for x in events if not x.linked], smooth_radius)
        groups = detect_groups(events)
    for g in groups:
        start_shift, end_shift = g[0].shift, g[-1].shift
        avg = average_shifts(g)
        logging.info(u'Group (start: {0}, end: {1}, lines: {2}), shifts (start: {3}, end: {4}, average: {5})'
                     .format(format_time(g[0].start), format_time(g[-1].end), len(g), start_shift, end_shift, avg))
    return groups
