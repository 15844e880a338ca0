#!/usr/bin/env python3
"""GPU dev tool: what the drop-in costs per call -- sequential WavStream.find_substream calls (the
reference's calling pattern), the sequential calculate_shifts, and the speculative batched form, on a
config-1-like scenario (5-min 12 kHz streams, +1.5 s offset, CLI default windows 10 / 30)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sushi_amd import synth
    from sushi_amd.shifts import ScriptEvent, calculate_shifts, calculate_shifts_batched
    from sushi_amd.wav import WavStream
    out = {}
    for sample_type in ("uint8", "float32"):
        dst_pcm = synth.make_dst_pcm(300, 12000, seed=1)
        src_pcm = synth.make_src_pcm(dst_pcm, 18000, seed=2)
        dst = WavStream.from_samples(dst_pcm, 12000, sample_type=sample_type)
        src = WavStream.from_samples(src_pcm, 12000, sample_type=sample_type)
        spans = synth.make_events(200, 300, 1.5, seed=3, min_len=1.0, max_len=4.0)
        pats = [src.get_substream(s, e) for s, e in spans]
        dst.find_substream(pats[0], spans[0][0], 10)                   # warm up (spectra, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for (s, e), p in zip(spans, pats):
            dst.find_substream(p, s + 1.5, 1.5)
        t_small = (time.perf_counter() - t0) / len(spans)
        t0 = time.perf_counter()
        for (s, e), p in zip(spans[:50], pats[:50]):
            dst.find_substream(p, s, 10)
        t_w10 = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        dst.find_substreams(pats, [s for s, _ in spans], [10] * len(spans))
        t_batch = time.perf_counter() - t0
        # the triple of sushi.py:450-452 (whole / left half / right half of a group) the way calculate_shifts issues it here:
        # one launch of three searches -- against the same three as consecutive calls, and against one call
        import numpy as np
        trip = {}
        for w in (10, 30):
            halves = [np.split(p, [p.shape[1] // 2], axis=1) for p in pats[:50]]
            t0 = time.perf_counter()
            got = []
            for (s, e), p, (l, r) in zip(spans[:50], pats[:50], halves):
                ro = l.shape[1] / 12000.0
                got.append(dst.find_substreams([p, l, r], [s, s, s + ro], [w] * 3)[1])
            trip["w%d_one_launch_ms" % w] = (time.perf_counter() - t0) / 50 * 1e3
            t0 = time.perf_counter()
            ref = []
            for (s, e), p, (l, r) in zip(spans[:50], pats[:50], halves):
                ro = l.shape[1] / 12000.0
                ref.append([dst.find_substream(p, s, w)[1], dst.find_substream(l, s, w)[1], dst.find_substream(r, s + ro, w)[1]])
            trip["w%d_three_calls_ms" % w] = (time.perf_counter() - t0) / 50 * 1e3
            assert got == ref
            t0 = time.perf_counter()
            for (s, e), p in zip(spans[:50], pats[:50]):
                dst.find_substream(p, s, w)
            trip["w%d_single_ms" % w] = (time.perf_counter() - t0) / 50 * 1e3
            trip["w%d_triple_over_single" % w] = trip["w%d_one_launch_ms" % w] / trip["w%d_single_ms" % w]
        ev = [ScriptEvent(s, e) for s, e in spans]
        t0 = time.perf_counter()
        calculate_shifts(src, dst, [[x] for x in ev], 10, 30, 5)
        t_seq = time.perf_counter() - t0
        warm = [ScriptEvent(s, e) for s, e in spans]                    # first batched launch of the process: workspace
        calculate_shifts_batched(src, dst, [[x] for x in warm], 10, 30, 5)   # allocation, not what is being measured
        ev2 = [ScriptEvent(s, e) for s, e in spans]
        t0 = time.perf_counter()
        proxy = calculate_shifts_batched(src, dst, [[x] for x in ev2], 10, 30, 5)
        t_spec = time.perf_counter() - t0
        assert [(a.shift, a.diff) for a in ev] == [(a.shift, a.diff) for a in ev2]
        out[sample_type] = {"find_substream_w1.5_ms": t_small * 1e3, "find_substream_w10_ms": t_w10 * 1e3,
                            "find_substreams_200x_w10_ms": t_batch * 1e3, "triple": trip,
                            "calculate_shifts_sequential_200_groups_ms": t_seq * 1e3,
                            "calculate_shifts_speculative_200_groups_ms": t_spec * 1e3,
                            "speculative_launches": proxy.launches, "requests": proxy.requests}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
