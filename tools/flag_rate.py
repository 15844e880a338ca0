#!/usr/bin/env python3
"""Dev: how many searches of the latency scenario (5-min streams, 200 events, +-10 s and +-1.5 s windows) the FFT
path hands to its fallback kernels, per sample type."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sushi_amd import synth  # noqa: E402
from sushi_amd.device import SearchBatch  # noqa: E402
from sushi_amd.wav import WavStream  # noqa: E402


def main():
    for sample_type in ("uint8", "float32"):
        dst_pcm = synth.make_dst_pcm(300, 12000, seed=1)
        src_pcm = synth.make_src_pcm(dst_pcm, 18000, seed=2)
        dst = WavStream.from_samples(dst_pcm, 12000, sample_type=sample_type)
        src = WavStream.from_samples(src_pcm, 12000, sample_type=sample_type)
        spans = synth.make_events(200, 300, 1.5, seed=3, min_len=1.0, max_len=4.0)
        for window, centre_off in ((10, 0.0), (1.5, 1.5)):
            offs, lens, wst, npos = [], [], [], []
            for s, e in spans:
                p = src.get_substream(s, e)
                _, lo, n = dst._window(p.shape[1], s + centre_off, window)
                offs.append(src._get_sample_for_time(s)); lens.append(p.shape[1]); wst.append(lo); npos.append(n)
            b = SearchBatch(dst.device_stream(), src.device_stream(), offs, lens, wst, npos, path="fft")
            b.run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.run(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("FLAGS", sample_type, "window", window, "searches", len(spans), "fallback", b.fallback_count(),
                  "batch ms %.3f" % (dt * 1e3), "min len", min(lens))


if __name__ == "__main__":
    main()
