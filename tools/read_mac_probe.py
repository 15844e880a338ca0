#!/usr/bin/env python3
"""Reads the per-wave records a PROBE build of mac_kernel leaves in the batch workspace (tools/experiments/r05_product_probe.patch,
r05_store_wave_probe_on_top.patch: every 4099th workgroup, one 32-byte record per wave) after ONE run of the bench workload, and
prints per segment-count class what a producer wave and a store wave spent: microseconds from the kernel's start of work to the
wave's end (s_memtime ticks of 10 ns), inside macq_reserve, per round; the store wave's steps, idle steps and slots taken.
Needs the stream cache bench.py writes (SUSHI_BENCH_CACHE).  usage: SUSHI_HIP_LIB=.../libsushi_hip_probe_sw.so read_mac_probe.py [--config 2]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sushi_amd import synth  # noqa: E402
from sushi_amd.wav import WavStream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--tag", default="")
args = ap.parse_args()
cfg = bench.CONFIGS[args.config]
rate, seconds, n_total = cfg["rate"], cfg["minutes"] * 60.0, cfg["events"]
OFFSET = 7.25
seed = 20260924 + args.config
cache = os.environ["SUSHI_BENCH_CACHE"]
z = np.load(os.path.join(cache, "c%d_%g_%d_float32_%g_0.npz" % (args.config, cfg["minutes"], rate, OFFSET)), allow_pickle=True)
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
b = SearchBatch(dst.device_stream(), src.device_stream(), offs, lens, wst, npos, path="fft", delta=DEFAULT_DELTA, workspace_bytes=160 << 30)
b.run(); torch.cuda.synchronize()
mem = b._mem[: (b._mem.numel() // 8) * 8].view(torch.int64)
MAGIC = 0x4d41435150524f42                     # the first two words of a record, little-endian
hits = (mem == MAGIC).nonzero().flatten()
recs = []
for h in hits.tolist():
    w = mem[h:h + 4].cpu().numpy().view(np.uint32)
    recs.append(dict(wg=int(w[2]), wave=int(w[3] & 0xff), cls=int(w[3] >> 8), ticks=int(w[4]), r1=int(w[5]), r2=int(w[6]), r3=int(w[7])))
out = {"tag": args.tag, "lib": os.environ.get("SUSHI_HIP_LIB", "product"), "records": len(recs)}
for cls in sorted({r["cls"] for r in recs}):
    prod = [r for r in recs if r["cls"] == cls and r["wave"] < 4]
    cons = [r for r in recs if r["cls"] == cls and r["wave"] == 4]
    e = {"producer_waves": len(prod), "producer_us_median": float(np.median([r["ticks"] for r in prod])) / 100.0}
    if any(r["r2"] for r in prod):
        e["reserve_us_median"] = float(np.median([r["r1"] for r in prod])) / 100.0
        e["rounds_median"] = float(np.median([r["r2"] for r in prod]))
        e["pushes_median"] = float(np.median([r["r3"] for r in prod]))
        e["us_per_round"] = e["producer_us_median"] / max(e["rounds_median"], 1.0)
    if cons:
        e["store_wave_us_median"] = float(np.median([r["ticks"] for r in cons])) / 100.0
        e["store_wave_steps_median"] = float(np.median([r["r1"] for r in cons]))
        e["store_wave_idle_steps_median"] = float(np.median([r["r2"] for r in cons]))
        e["store_wave_taken_median"] = float(np.median([r["r3"] for r in cons]))
    out["class_%d" % cls] = e
print(json.dumps(out))
