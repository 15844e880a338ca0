"""The caller of the hot path: ``sushi.calculate_shifts`` (reference sushi.py:400-508), restated for
Python 3, plus a speculative batched form that feeds it from a few large GPU launches.

``calculate_shifts`` walks the search groups sequentially: every call of
``dst_stream.find_substream`` is centred on the shift the *previous* group committed
(sushi.py:420,432,450-452,459-462), so the reference issues 1-7 dependent searches per group.
On the GPU one search of a +-10 s window costs ~10 us of device time but a launch + readback
costs far more, so the dependent chain -- not the arithmetic -- would set the pace.

``SpeculativeStream`` breaks the chain without changing a single result.  It stands in for the
destination ``WavStream`` and answers ``find_substream`` from a cache of exact results:

* a search's answer is an absolute position in the destination stream plus the score *of that
  position*; both are independent of which window found them.  So a cached result for the same
  pattern over window W answers a request over any window W' inside W, provided the cached arg-min
  lies in W' (the first index of the minimum of a superset that falls inside the subset is the first
  index of the subset's minimum);
* on a miss the proxy runs ONE batched launch (``WavStream.find_substreams``) holding the request
  plus the searches the state machine will ask for next if the shift stays what it is now: the same
  offset and window for the following ``lookahead`` groups (full pattern; and both halves unless it
  is the small-window probe of sushi.py:431-432).

Where the shift is piecewise constant (chapters, sushi.py:120-187) that is one launch per piece.
A wrong guess costs a wasted speculative search, never a different answer: the state machine below
consumes exactly the values the sequential calls would have produced.
"""
import logging

import numpy as np

from .common import SushiError, format_time

ALLOWED_ERROR = 0.01      # sushi.py:39
SMALL_WINDOW = 1.5        # sushi.py:410


class ScriptEvent(object):
    """The part of subs.ScriptEventBase (subs.py:14-80) that calculate_shifts and the grouping code
    (sushi_amd/grouping.py) touch: a time span, a shift/diff pair, and links to another event whose shift
    it follows."""

    def __init__(self, start, end, source_index=0, text=u'', is_comment=False):
        self.source_index = source_index
        self.start = start
        self.end = end
        self.text = text
        self.is_comment = is_comment      # subs.py: AssEvent / SrtEvent attribute read by prepare_search_groups
        self._shift = 0
        self._diff = 1
        self._linked_event = None

    @property
    def linked(self):
        return self._linked_event is not None

    @property
    def shift(self):
        return self._linked_event.shift if self.linked else self._shift

    @property
    def diff(self):
        return self._linked_event.diff if self.linked else self._diff

    @property
    def duration(self):
        return self.end - self.start

    def set_shift(self, shift, audio_diff):
        assert not self.linked, 'Cannot set shift of a linked event'
        self._shift = shift
        self._diff = audio_diff

    def get_link_chain_end(self):
        return self._linked_event.get_link_chain_end() if self.linked else self

    def link_event(self, other):
        assert other.get_link_chain_end() is not self, 'Circular link detected'
        self._linked_event = other

    def resolve_link(self):
        assert self.linked, 'Cannot resolve unlinked events'
        self._shift, self._diff = self._linked_event.shift, self._linked_event.diff
        self._linked_event = None

    def __str__(self):
        return u'{0}-{1} {2}'.format(format_time(self.start), format_time(self.end), self.text)


def _span_state(group, shift=None, diff=None):
    return {"start_time": group[0].start, "end_time": group[-1].end, "shift": shift, "diff": diff}


def _log_shift(state):
    logging.info('{0}-{1}: shift: {2:0.10f}, diff: {3:0.10f}'.format(
        format_time(state["start_time"]), format_time(state["end_time"]), state["shift"], state["diff"]))


def _triple_search(dst_stream, whole, left, right, centre, window, right_offset):
    """The three searches of sushi.py:450-452 / 460-462 and their agreement test (:453 / :463)."""
    batched = getattr(dst_stream, "find_substreams", None)
    if batched is not None:
        # the three searches are independent of each other: one launch (one pass of every kernel over three searches) instead
        # of three launches one after the other -- the same three results (tools/latency.py: a triple costs ~1.1 x a single call)
        diffs, times = batched([whole, left, right], [centre, centre, centre + right_offset], [window] * 3)
        diff, whole_time, left_time, right_time = diffs[0], times[0], times[1], times[2] - right_offset
    else:
        diff, whole_time = dst_stream.find_substream(whole, centre, window)
        left_time = dst_stream.find_substream(left, centre, window)[1]
        right_time = dst_stream.find_substream(right, centre + right_offset, window)[1] - right_offset
    agree = abs(left_time - right_time) <= ALLOWED_ERROR and abs(whole_time - left_time) <= ALLOWED_ERROR
    return diff, whole_time, left_time, right_time, agree


def calculate_shifts(src_stream, dst_stream, groups_list, normal_window, max_window, rewind_thresh):
    """sushi.py:400-508.  Same arguments, same effect on the events of ``groups_list`` (set_shift /
    link_event), same log lines.  ``dst_stream`` needs ``duration_seconds`` and ``find_substream``."""
    committed, pending = [], []           # group states: settled / found but not yet trusted (:412-413)
    window = normal_window
    at = 0
    while at < len(groups_list):
        group = groups_list[at]
        whole = src_stream.get_substream(group[0].start, group[-1].end)              # :417
        t0 = group[0].start
        state = _span_state(group)
        base_shift = committed[-1]["shift"] if committed else 0                       # :420
        diff = found = None

        if not pending:
            if t0 + base_shift > dst_stream.duration_seconds:                         # :424-429
                # this and every later group starts past the end of the destination audio
                for g in groups_list[at:]:
                    committed.append(_span_state(g))
                    logging.info("{0}-{1}: outside of audio range".format(format_time(g[0].start),
                                                                          format_time(g[-1].end)))
                break
            if SMALL_WINDOW < window:                                                 # :431-432
                diff, found = dst_stream.find_substream(whole, t0 + base_shift, SMALL_WINDOW)
            if found is not None and abs((found - t0) - base_shift) <= ALLOWED_ERROR:  # :434-443
                state.update({"shift": found - t0, "diff": diff})
                committed.append(state)
                _log_shift(state)
                if window != normal_window:
                    logging.info("Going back to window {0} from {1}".format(normal_window, window))
                    window = normal_window
                at += 1
                continue

        left, right = np.split(whole, [len(whole[0]) // 2], axis=1)                   # :445 (Python 2 int division)
        right_offset = len(left[0]) / float(src_stream.sample_rate)                   # :446
        settled = False
        if t0 + base_shift < dst_stream.duration_seconds:                             # :449-455
            diff, found, lt, rt, settled = _triple_search(dst_stream, whole, left, right, t0 + base_shift,
                                                          window, right_offset)
            logging.debug('{0}-{1}: shift: {2:0.5f} [{3:0.5f}, {4:0.5f}], search offset: {5:0.6f}'.format(
                format_time(state["start_time"]), format_time(state["end_time"]), found - t0, lt - t0, rt - t0,
                base_shift))
        if not settled and pending and pending[-1]["shift"] is not None \
                and t0 + pending[-1]["shift"] < dst_stream.duration_seconds:          # :457-465
            retry_shift = pending[-1]["shift"]
            diff, found, lt, rt, settled = _triple_search(dst_stream, whole, left, right, t0 + retry_shift,
                                                          window, right_offset)
            logging.debug('{0}-{1}: shift: {2:0.5f} [{3:0.5f}, {4:0.5f}], search offset: {5:0.6f}'.format(
                format_time(state["start_time"]), format_time(state["end_time"]), found - t0, lt - t0, rt - t0,
                retry_shift))

        shift = found - t0                                                            # :467 (TypeError if nothing ran, as in the reference)
        if not settled:                                                               # :468-479
            state.update({"shift": shift, "diff": diff})
            pending.append(state)
            at += 1
            if rewind_thresh == len(pending) and window < max_window:
                logging.warning("Detected possibly broken segment starting at {0}, increasing the window from {1} "
                                "to {2}".format(format_time(pending[0]["start_time"]), window, max_window))
                window = max_window
                at = len(committed)
                del pending[:]
            continue

        if pending:                                                                   # :482-485
            logging.warning("Events from {0} to {1} will most likely be broken!".format(
                format_time(pending[0]["start_time"]), format_time(pending[-1]["end_time"])))
        pending.append(state)                                                         # :487-493
        for st in pending:
            st.update({"shift": shift, "diff": diff})
            _log_shift(st)
        committed.extend(pending)
        del pending[:]
        at += 1

    for st in pending:                                                                # :495-496
        _log_shift(st)

    for at, (group, st) in enumerate(zip(groups_list, committed + pending)):          # :498-508
        if st["shift"] is None:
            for earlier in reversed(groups_list[:at]):
                target = next((x for x in reversed(earlier) if not x.linked), None)
                if target:
                    for e in group:
                        e.link_event(target)
                    break
        else:
            for e in group:
                e.set_shift(st["shift"], st["diff"])


class SpeculativeStream(object):
    """Stands in for the destination stream inside ``calculate_shifts`` (module docstring).

    ``dst`` must offer ``find_substreams(patterns, centres, windows, with_index=True)``,
    ``_window(pattern_len, centre, window)`` -> (start_time, first sample, P) and ``sample_rate`` /
    ``duration_seconds`` (``sushi_amd.wav.WavStream`` does)."""

    def __init__(self, dst, src, groups_list, lookahead=64, max_lookahead=4096):
        self.dst = dst
        self.src = src
        self.lookahead = int(lookahead)            # groups speculated per launch; doubles while guesses hold
        self.base_lookahead = int(lookahead)
        self.max_lookahead = max(int(max_lookahead), int(lookahead))
        self._last_launch = None                   # (first group, groups speculated) of the previous launch
        self.launches = 0          # batched launches issued
        self.searches = 0          # searches computed (useful + speculative)
        self.requests = 0          # find_substream calls answered
        self.hits = 0
        self._cache = {}           # (pattern offset, length) -> [(first sample, P, abs arg-min sample, score)]
        self._groups = []          # per group: (start, whole, left, right, right_offset)
        self._role = {}            # (pattern offset, length) -> (group index, 0 whole / 1 left / 2 right)
        base = src.data.__array_interface__['data'][0]
        itemsize = src.data.itemsize
        for gi, group in enumerate(groups_list):
            whole = src.get_substream(group[0].start, group[-1].end)
            left, right = np.split(whole, [len(whole[0]) // 2], axis=1)
            right_offset = len(left[0]) / float(src.sample_rate)
            self._groups.append((group[0].start, whole, left, right, right_offset))
            for role, pat in enumerate((whole, left, right)):
                key = ((pat.__array_interface__['data'][0] - base) // itemsize, pat.shape[1])
                self._role.setdefault(key, (gi, role))
        self._src_base, self._src_itemsize = base, itemsize
        self._src_nbytes = src.data.nbytes

    # ---- what calculate_shifts reads ---------------------------------------------------------
    @property
    def duration_seconds(self):
        return self.dst.duration_seconds

    @property
    def sample_rate(self):
        return self.dst.sample_rate

    def find_substream(self, pattern, window_center, window_size):
        self.requests += 1
        key = self._key(pattern)
        if key is None:                      # not a view of the source stream: nothing to speculate on
            return self.dst.find_substream(pattern, window_center, window_size)
        start_time, first, n_pos = self.dst._window(pattern.shape[1], window_center, window_size)
        if n_pos < 1:
            raise SushiError('pattern is longer than the search window (cv2.error in the reference)')
        got = self._lookup(key, first, n_pos)
        if got is None:
            self._launch(key, pattern, window_center, window_size)
            got = self._lookup(key, first, n_pos)
        else:
            self.hits += 1
        pos, score = got
        # wav.py:188 with min_idx = pos - first
        return score, start_time + ((pos - first) / float(self.dst.sample_rate))

    # ---- internals ---------------------------------------------------------------------------
    def _key(self, pattern):
        if not isinstance(pattern, np.ndarray) or pattern.ndim != 2 or pattern.shape[0] != 1 \
                or pattern.dtype != self.src.data.dtype or pattern.strides[1] != pattern.itemsize:
            return None
        addr = pattern.__array_interface__['data'][0]
        if not (self._src_base <= addr and addr + pattern.shape[1] * pattern.itemsize <= self._src_base + self._src_nbytes):
            return None
        return ((addr - self._src_base) // self._src_itemsize, pattern.shape[1])

    def _lookup(self, key, first, n_pos):
        for (f0, p0, pos, score) in self._cache.get(key, ()):
            if f0 <= first and first + n_pos <= f0 + p0 and first <= pos < first + n_pos:
                return pos, score
        return None

    def _launch(self, key, pattern, centre, window):
        pats, centres, wins, keys = [pattern], [centre], [window], [key]
        role = self._role.get(key)
        if role is not None and self.lookahead > 0:
            gi, which = role
            # the previous guess held for every group it covered -> speculate twice as far this time;
            # it broke early (a chapter boundary, a noisy group) -> back to the base depth
            if self._last_launch is not None:
                first, count = self._last_launch
                self.lookahead = min(2 * self.lookahead, self.max_lookahead) if gi >= first + count \
                    else self.base_lookahead
            self._last_launch = (gi, min(self.lookahead, len(self._groups) - gi))
            start_g, _, _, _, roff_g = self._groups[gi]
            offset = centre - start_g - (roff_g if which == 2 else 0.0)     # the shift being probed
            probe_only = (which == 0 and window == SMALL_WINDOW)            # sushi.py:431-432 asks for nothing else
            for h in range(gi, min(gi + self.lookahead, len(self._groups))):
                start_h, whole, left, right, roff = self._groups[h]
                if start_h + offset > self.dst.duration_seconds:
                    break
                cands = [(whole, start_h + offset)] if probe_only else \
                    [(whole, start_h + offset), (left, start_h + offset), (right, start_h + offset + roff)]
                for pat, c in cands:
                    k = self._key(pat)
                    if pat.shape[1] < 1 or (k == key and c == centre):
                        continue
                    _, first, n_pos = self.dst._window(pat.shape[1], c, window)
                    if n_pos < 1 or self._lookup_window(k, first, n_pos):
                        continue
                    pats.append(pat); centres.append(c); wins.append(window); keys.append(k)
        scores, _times, positions = self.dst.find_substreams(pats, centres, wins, with_index=True)
        self.launches += 1
        self.searches += len(pats)
        for k, pat, c, w, score, pos in zip(keys, pats, centres, wins, scores, positions):
            _, first, n_pos = self.dst._window(pat.shape[1], c, w)
            self._cache.setdefault(k, []).append((first, n_pos, int(pos), score))

    def _lookup_window(self, key, first, n_pos):
        """True if some cached search of this pattern already covers [first, first + n_pos)."""
        return any(f0 <= first and first + n_pos <= f0 + p0 for (f0, p0, _, _) in self._cache.get(key, ()))


def calculate_shifts_batched(src_stream, dst_stream, groups_list, normal_window, max_window, rewind_thresh,
                             lookahead=64, max_lookahead=4096):
    """calculate_shifts with the destination stream behind a SpeculativeStream: identical results,
    a handful of batched GPU launches instead of thousands of dependent ones.  Returns the proxy
    (its counters tell how the speculation went)."""
    proxy = SpeculativeStream(dst_stream, src_stream, groups_list, lookahead=lookahead, max_lookahead=max_lookahead)
    calculate_shifts(src_stream, proxy, groups_list, normal_window, max_window, rewind_thresh)
    return proxy
