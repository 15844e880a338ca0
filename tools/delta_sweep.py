#!/usr/bin/env python3
"""GPU dev tool: how small can the FFT path's `delta` get before its results differ from the direct
kernel's?  Runs the bench workload (fewer events) through the direct kernel once and through the FFT
path for a ladder of deltas; prints mismatches, fallbacks and step times.  The smallest delta with
zero mismatches bounds the f32 FFT score error from above ("delta")."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=200)
    ap.add_argument("--minutes", type=float, default=45.0)
    ap.add_argument("--window", type=float, default=60.0)
    ap.add_argument("--sample-type", default="float32")
    ap.add_argument("--ws-mb", type=int, nargs="*", default=[1024])
    args = ap.parse_args()
    import torch
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream
    rate, off = 12000, 7.25
    seconds = args.minutes * 60
    dst_pcm = synth.make_dst_pcm(seconds, rate, seed=11)
    src_pcm = synth.make_src_pcm(dst_pcm, int(off * rate), seed=12)
    dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=args.sample_type)
    src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=args.sample_type)
    events = synth.make_events(args.events, seconds, args.window + off, seed=13)
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, off, args.window, seed=14)
    offs = [src._get_sample_for_time(s) for s, _ in events]
    lens = [p.shape[1] for p in pats]
    wst, npos = [], []
    for m, c, w in zip(lens, centres, wins):
        _, lo, p = dst._window(m, c, w)
        wst.append(lo); npos.append(p)
    dd, sd = dst.device_stream(), src.device_stream()

    def timed(b, reps=3):
        b.run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            b.run()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    ref = SearchBatch(dd, sd, offs, lens, wst, npos, variant=2, path="direct")
    ms_direct = timed(ref, 1)
    ridx, rscore = ref.results()
    print(json.dumps({"path": "direct", "ms": ms_direct}))
    for ws in args.ws_mb:
        for delta in (2e-5, 2e-6, 2e-7, 5e-8, 1e-8, 1e-9):
            b = SearchBatch(dd, sd, offs, lens, wst, npos, path="fft", delta=delta, workspace_bytes=ws << 20)
            ms = timed(b)
            idx, score = b.results()
            bad = int((idx != ridx).sum())
            sdiff = float(np.abs(score.astype(np.float64) - rscore).max())
            print(json.dumps({"path": "fft", "ws_mb": ws, "delta": delta, "ms": ms, "idx_mismatch": bad,
                              "max_idx_diff": int(np.abs(idx.astype(np.int64) - ridx).max()),
                              "max_score_diff": sdiff, "fallbacks": b.fallback_count()}))


if __name__ == "__main__":
    main()
