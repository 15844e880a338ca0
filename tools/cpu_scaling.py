#!/usr/bin/env python3
"""How the oracle leg of bench.py scales with worker processes on this host (the GPU boxes report 256 logical CPUs and
deliver far less): events/s for 1, 2, 4, ... workers on a configs[1]-shaped sample.  CPU only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SUSHI_HIP_LOAD"] = "host"
import bench  # noqa: E402
from sushi_amd import synth  # noqa: E402
from sushi_amd.wav import WavStream  # noqa: E402

rate, seconds = 12000, 900
dst_pcm = synth.make_dst_pcm(seconds, rate, seed=1)
src_pcm = synth.make_src_pcm(dst_pcm, int(7.25 * rate), seed=2)
dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type="float32")
src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type="float32")
events = synth.make_events(512, seconds, 67.25, seed=3)
pats, centres, wins = synth.explicit_descriptors(src, dst, events, 7.25, 60, seed=4)
offs = [src._get_sample_for_time(s) for s, _ in events]
lens = [p.shape[1] for p in pats]
wst, npos = [], []
for m, c, w in zip(lens, centres, wins):
    st, lo, p = dst._window(m, c, w)
    wst.append(lo); npos.append(p)
print(json.dumps({"usable_cores": bench.usable_cores(), "cpu_count": os.cpu_count(), "loadavg": os.getloadavg()}))
for workers in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if workers > (os.cpu_count() or 1):
        break
    t = time.time()
    cpu, res = bench.oracle_leg(dst.data[0], src.data[0], offs, lens, wst, npos, "sqdiff_normed", forced=[], timed=True,
                                min_sample=8, budget_s=4.0, workers=workers)
    print(json.dumps({"workers": workers, "events_per_s": round(cpu["value"], 2), "searches": len(res),
                      "per_search_alone": round(cpu["per_search_s_alone"], 4), "per_search_loaded": round(cpu["per_search_s_loaded"], 4),
                      "wall": round(time.time() - t, 1)}), flush=True)
