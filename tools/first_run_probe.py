#!/usr/bin/env python3
"""Dev: what does the FIRST run of a batch cost next to the runs after it?  (A one-shot job -- sushi.py: two WavStream loads, one
calculate_shifts pass -- IS a first run: the exclusion's form is voted on, the lanes' streams and events are created, ...)
Needs SUSHI_BENCH_CACHE (bench.py's stream cache of BASELINE configs[2])."""
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
from sushi_amd.device import warm_up  # noqa: E402
warm_ms = warm_up(dev)               # (what a one-shot job does first thing, behind its demux)
dst._device = src._device = dev
D, S = dst.device_stream(), src.device_stream()
D.searchable()
torch.cuda.synchronize()
out = {"warm_up_ms": round(warm_ms, 1)}
if os.environ.get("PROBE_PREWARM"):
    # every kernel of the hot path once, on a small batch: what device.warm_up() could do behind the demux
    t0 = time.perf_counter()
    for excl in ("band", "whole", "never"):
        w = SearchBatch(D, S, offs[:16], lens[:16], [o - 100000 for o in offs[:16]], [200001] * 16, path="fft", exclusion=excl)
        w.run(); w.results()
        del w
    torch.cuda.synchronize()
    out["prewarm_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
for tag in ("first_batch", "second_batch"):
    t0 = time.perf_counter()
    b = SearchBatch(D, S, offs, lens, wst, npos, path="fft", workspace_bytes=160 << 30)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    runs, stages = [], []
    from sushi_amd import _native
    for r in range(4):
        _native.profile_begin()
        t = time.perf_counter()
        b.run()
        t_launch = time.perf_counter() - t
        torch.cuda.synchronize()
        runs.append((round((time.perf_counter() - t) * 1e3, 3), round(t_launch * 1e3, 3)))
        stages.append({k: round(float(v), 2) for k, v in zip(_native.STAGE_NAMES, _native.profile_end(1)[0])})
    out[tag] = {"create_ms": round((t1 - t0) * 1e3, 3), "run_ms_total_and_host_launch": runs, "stage_ms": stages[:2], "lanes": b.lanes, "sub_batches": b.sub_batches}
    del b
print(json.dumps(out))
