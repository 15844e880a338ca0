#!/usr/bin/env python3
"""GPU: where one drop-in `find_substream` call's 0.2 ms goes -- host arithmetic, batch creation (plan + descriptor upload),
the launch chain of run(), the two result copies -- on the latency workload of tools/latency.py (5-min streams, +-10 s)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sushi_amd import synth
    from sushi_amd.device import SearchBatch
    from sushi_amd.wav import WavStream, _locate
    dst_pcm = synth.make_dst_pcm(300, 12000, seed=1)
    src_pcm = synth.make_src_pcm(dst_pcm, 18000, seed=2)
    dst = WavStream.from_samples(dst_pcm, 12000, sample_type="float32")
    src = WavStream.from_samples(src_pcm, 12000, sample_type="float32")
    spans = synth.make_events(200, 300, 1.5, seed=3, min_len=1.0, max_len=4.0)
    pats = [src.get_substream(s, e) for s, e in spans]
    dst.find_substream(pats[0], spans[0][0], 10)
    torch.cuda.synchronize()
    ddev, sdev = dst.device_stream(), src.device_stream()
    acc = {"locate_and_window": 0.0, "batch_create": 0.0, "run_call_returns": 0.0, "wait_for_the_gpu": 0.0, "two_result_copies": 0.0}
    t_all = time.perf_counter()
    for (s, e), p in zip(spans, pats):
        t0 = time.perf_counter()
        owner, off, m = _locate(p)
        st, lo, npos = dst._window(m, s, 10)
        t1 = time.perf_counter()
        b = SearchBatch(ddev, sdev, [off], [m], [lo], [npos])
        t2 = time.perf_counter()
        b.run()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        b.results()
        t5 = time.perf_counter()
        for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += d
    total = time.perf_counter() - t_all
    out = {k: round(v / len(spans) * 1e6, 1) for k, v in acc.items()}
    out["total_us_per_call"] = round(total / len(spans) * 1e6, 1)
    t0 = time.perf_counter()
    for (s, e), p in zip(spans, pats):
        dst.find_substream(p, s, 10)
    out["find_substream_us_per_call"] = round((time.perf_counter() - t0) / len(spans) * 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
