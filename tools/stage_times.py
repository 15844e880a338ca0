#!/usr/bin/env python3
"""Per-stage HIP-event times of the bench workload WITHOUT any check of the results: for ablation builds of the library
(SUSHI_HIP_LIB=... built with -DSUSHI_DEV_MAC_ABL=n, whose outputs are garbage).  Needs the stream cache bench.py writes
(SUSHI_BENCH_CACHE) so that it starts in seconds.  usage: stage_times.py [--config 2] [--steps 5] [--tag name]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sushi_amd import synth, _native  # noqa: E402
from sushi_amd.wav import WavStream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--tag", default="")
args = ap.parse_args()
cfg = bench.CONFIGS[args.config]
rate, seconds, n_total = cfg["rate"], cfg["minutes"] * 60.0, cfg["events"]
OFFSET = 7.25                                              # bench.py's default planted offset
seed = 20260924 + args.config
cache = os.environ["SUSHI_BENCH_CACHE"]
z = np.load(os.path.join(cache, "c%d_%g_%d_float32_%g_0_20_0.npz" % (args.config, cfg["minutes"], rate, OFFSET)), allow_pickle=False)
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
from sushi_amd.device import SearchBatch, DEFAULT_DELTA  # noqa: E402
dev = torch.device("cuda", 0)
dst._device = src._device = dev
b = SearchBatch(dst.device_stream(), src.device_stream(), offs, lens, wst, npos, path="fft", delta=DEFAULT_DELTA,
                workspace_bytes=160 << 30)
b.run(); torch.cuda.synchronize()
_native.profile_begin()
for _ in range(args.steps):
    b.run()
torch.cuda.synchronize()
ms = _native.profile_end(args.steps).mean(axis=0)
print(json.dumps({"tag": args.tag, "lib": os.environ.get("SUSHI_HIP_LIB", "product"),
                  "stage_ms": {n: round(float(v), 3) for n, v in zip(_native.STAGE_NAMES, ms)}}))
