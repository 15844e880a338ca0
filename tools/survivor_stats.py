#!/usr/bin/env python3
"""GPU: which searches' pairs does the band-split bound leave?  The bench's job at BASELINE configs[2]; per pattern length (segments)
the pairs with slb <= the search's final score, for both forms."""
import os, sys
os.environ["SUSHI_HIP_LOAD"] = "host"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sushi_amd import synth
from sushi_amd.device import DeviceStream, SearchBatch
from sushi_amd.wav import WavStream
from sushi_amd.distributed import fft_layout_host

rate, seconds, n_ev, window, off = 12000, 7200.0, 3000, 120.0, 7.25
snr = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
seed = 20260924 + 2
dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
src_pcm = synth.make_src_pcm(dst_pcm, int(round(off * rate)), snr_db=snr, seed=seed + 1)
dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type="float32")
src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type="float32")
events = synth.make_events(n_ev, seconds, window + off, seed=seed + 2)
pats, centres, wins = synth.explicit_descriptors(src, dst, events, off, window, seed=seed + 3)
offs = [src._get_sample_for_time(s) for s, _ in events]
lens = [p.shape[1] for p in pats]
wst, npos = [], []
for m, c, w in zip(lens, centres, wins):
    _, lo, p = dst._window(m, c, w)
    wst.append(lo); npos.append(p)
D, S = DeviceStream(dst.data[0]), DeviceStream(src.data[0])
pairs, segs = fft_layout_host(wst, npos, lens)
first = np.concatenate(([0], np.cumsum(pairs)))
for form in ("band", "whole"):
    b = SearchBatch(D, S, offs, lens, wst, npos, path="fft", exclusion=form, workspace_bytes=160 << 30)
    b.run()
    idx, score = b.results()
    slb, acc = b.pair_bounds()
    left = np.array([int((slb[first[k]:first[k + 1]] <= score[k] * 1.000001 + 1e-7).sum()) for k in range(n_ev)])
    print(form, "snr", snr, "pairs left in all:", int(left.sum()), "of", int(pairs.sum()), "transformed", b.diagnostics()["pairs_transformed"])
    for lo_s, hi_s in ((1, 3), (4, 4), (5, 6), (7, 9), (10, 12), (13, 15)):
        m = (segs >= lo_s) & (segs <= hi_s)
        if m.any():
            print("   segments %2d-%2d: %4d searches, pairs left per search: mean %.2f  median %d  p90 %d  max %d ; searches with only their own pair left: %.0f %%" % (
                lo_s, hi_s, int(m.sum()), left[m].mean(), int(np.median(left[m])), int(np.percentile(left[m], 90)), int(left[m].max()), 100.0 * np.mean(left[m] <= 1)))
    del b
