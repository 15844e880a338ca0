import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sushi_amd import synth
from sushi_amd.wav import WavStream
dst_pcm = synth.make_dst_pcm(600, 12000, seed=1)
src_pcm = synth.make_src_pcm(dst_pcm, 18000, seed=2)
dst = WavStream.from_samples(dst_pcm, 12000, sample_type="float32", device="cuda:0")
src = WavStream.from_samples(src_pcm, 12000, sample_type="float32", device="cuda:0")
pats = [src.get_substream(10.0 + 2.5 * k, 13.0 + 2.5 * k) for k in range(200)]
for k in range(50):
    dst.find_substream(pats[k], 10.0 + 2.5 * k, 10)
import time
t = time.perf_counter()
for k in range(200):
    dst.find_substream(pats[k], 10.0 + 2.5 * k, 10)
print("us per call", (time.perf_counter() - t) / 200 * 1e6)
pr = cProfile.Profile(); pr.enable()
for k in range(200):
    dst.find_substream(pats[k], 10.0 + 2.5 * k, 10)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
