#!/usr/bin/env python3
"""Dev: do two halves of the bench job overlap when they run on two HIP streams?  (mac_kernel<1024> is store-bound, bound_low_kernel
VALU-bound: complementary -- if the hardware co-schedules them, a pipelined step is shorter than a sequential one.)
Needs SUSHI_BENCH_CACHE (bench.py's stream cache)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sushi_amd import synth  # noqa: E402
from sushi_amd.wav import WavStream  # noqa: E402

cfg = bench.CONFIGS[2]
rate, seconds, n_total = cfg["rate"], cfg["minutes"] * 60.0, cfg["events"]
OFFSET = 7.25
seed = 20260924 + 2
z = np.load(os.path.join(os.environ["SUSHI_BENCH_CACHE"], "c2_%g_%d_float32_%g_0_20_0.npz" % (cfg["minutes"], rate, OFFSET)), allow_pickle=False)
dst = WavStream.from_prepared(z["dst"], rate, int(z["sample_count"]), int(z["padding_size"]))
src = WavStream.from_prepared(z["src"], rate, int(z["sample_count"]), int(z["padding_size"]))
events = synth.make_events(n_total, seconds, cfg["window"] + OFFSET, seed=seed + 2)
pats, centres, wins = synth.explicit_descriptors(src, dst, events, OFFSET, cfg["window"], seed=seed + 3)
offs = [src._get_sample_for_time(s) for s, _ in events]
lens = [p.shape[1] for p in pats]
wst, npos = [], []
for m, c, w in zip(lens, centres, wins):
    st, lo, p = dst._window(m, c, w)
    wst.append(lo); npos.append(p)
import torch  # noqa: E402
from sushi_amd.device import SearchBatch  # noqa: E402
dev = torch.device("cuda", 0)
dst._device = src._device = dev
D, S = dst.device_stream(), src.device_stream()
D.searchable()


def mk(lo, hi):
    return SearchBatch(D, S, offs[lo:hi], lens[lo:hi], wst[lo:hi], npos[lo:hi], path="fft", workspace_bytes=160 << 30)


whole = mk(0, n_total)
streams = [torch.cuda.Stream(dev) for _ in range(4)]


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cuts(fracs):
    c = np.concatenate(([0.0], np.cumsum(fracs)))
    c = (c / c[-1] * n_total).astype(int)
    return [(int(c[i]), int(c[i + 1])) for i in range(len(fracs))]


out = {"whole_ms": round(timed(lambda: whole.run()), 3)}
configs = {"2x(50,50)": ([0.5, 0.5], 2), "2x(40,60)": ([0.4, 0.6], 2), "2x(30,70)": ([0.3, 0.7], 2), "3 on 3": ([1, 1, 1], 3),
           "4 on 2": ([1, 1, 1, 1], 2), "4 on 4": ([1, 1, 1, 1], 4), "6 on 2": ([1] * 6, 2), "8 on 2": ([1] * 8, 2), "8 on 4": ([1] * 8, 4),
           "6 on 3": ([1] * 6, 3)}
for name, (fr, ns) in configs.items():
    bs = [mk(lo, hi) for lo, hi in cuts(fr)]

    def par():
        for i, b in enumerate(bs):
            b.run(hip_stream=streams[i % ns].cuda_stream)
    out[name] = round(timed(par), 3)
    del bs
    torch.cuda.empty_cache()
print(json.dumps(out))
