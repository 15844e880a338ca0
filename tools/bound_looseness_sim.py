#!/usr/bin/env python3
"""CPU study (numpy, float64; no GPU, no library): how loose may the pair bound of DESIGN.md 3.2 be before pairs stop being excluded?
For a sample of bench-shaped searches (synthetic streams of sushi_amd/synth.py, +-120 s windows) and every block pair but the one that
holds the match: the largest |cross term| the pair may have before one of its positions could beat the match ("margin"), against
  * the pair's true largest |cross term| (the ideal bound),
  * sum over the sixteen decimated shares of max |A| after THREE in-wave passes (what bound_kernel computes),
  * the same after TWO passes with the third bounded: sum of the four moduli at a position / four maxima / 2 sqrt(sum of four squares),
  * an energy (Parseval) bound, sqrt(sum |y|^2) -- which needs no transform at all, and turns out 40-60 x the ideal.
The pair's complex transform is stood in for by two runs of 16384 consecutive correlation values (valid positions + what wraps
around): the statistics, not the bits.  usage: tools/bound_looseness_sim.py   (about a minute)"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sushi_amd import synth
from sushi_amd.wav import WavStream
rate = 12000; seconds = 1500.0; window = 120.0; offset = 7.25
dst_pcm = synth.make_dst_pcm(seconds, rate, seed=1)
src_pcm = synth.make_src_pcm(dst_pcm, int(round(offset * rate)), seed=2)
dst = WavStream.from_samples(dst_pcm, rate, sample_type="float32")
src = WavStream.from_samples(src_pcm, rate, sample_type="float32")
events = synth.make_events(24, seconds, window + offset, seed=3)
pats, centres, wins = synth.explicit_descriptors(src, dst, events, offset, window, seed=4)
d = dst.data[0].astype(np.float64); mu = d.mean()
N = 16384; H = 12288
rows = []
for pat, c, w in zip(pats, centres, wins):
    T = np.asarray(pat[0], dtype=np.float64); M = T.shape[0]
    _, lo, P = dst._window(M, c, w)
    W = d[lo:lo + P + M - 1]; Wc = W - mu
    n = 1 << int(math.ceil(math.log2(W.shape[0] + M)))
    corrc = np.fft.irfft(np.fft.rfft(Wc, n) * np.conj(np.fft.rfft(T, n)), n)[:P]
    cs = np.concatenate(([0.0], np.cumsum(W * W))); W2 = cs[M:M + P] - cs[:P]
    t = float(T @ T); sT = float(T.sum())
    score = (t + W2 - 2 * (corrc + mu * sT)) / np.sqrt(t * W2); U = score.min()
    thr = (t + W2 - U * np.sqrt(t * W2)) / 2 - mu * sT
    for a in range(0, P - 2 * N, 2 * H):
        if a <= score.argmin() < a + 2 * H: continue
        # a stand-in for the pair's complex output: two runs of N consecutive correlation values (valid + what wraps around)
        yc = corrc[a:a + N] + 1j * corrc[a + H:a + H + N]
        Y = np.fft.fft(yc)
        ideal = max(np.abs(corrc[a:a + 2 * H]).max(), 1e-30)
        A = np.stack([np.fft.ifft(Y[r::16]) / 16 for r in range(16)])          # y[n] = sum_r W^(rn) A_r[n mod 1024]
        B3 = np.abs(A).max(axis=1).sum()
        G = np.stack([[np.fft.ifft(Y[r::16][q::4]) / 64 for q in range(4)] for r in range(16)])   # A_r[m] = sum_q tw G_rq[m mod 256]
        B2 = np.abs(G).sum(axis=1).max(axis=1).sum()
        B2l = np.abs(G).max(axis=2).sum()
        B2cs = (2 * np.sqrt((np.abs(G) ** 2).sum(axis=1))).max(axis=1).sum()
        Bpar = math.sqrt(float(np.sum(np.abs(yc) ** 2)))
        tmin = float(thr[a:a + 2 * H].min())
        rows.append((tmin / ideal, B3 / ideal, B2 / ideal, B2l / ideal, M, B2cs / ideal, Bpar / ideal))
r = np.array(rows)
print("pairs", len(r)); print("margin over the ideal bound: min %.2f  1 %% %.2f  10 %% %.2f  median %.2f" % (r[:,0].min(), np.percentile(r[:,0],1), np.percentile(r[:,0],10), np.median(r[:,0]))); print("looseness of the energy bound: median %.1f" % np.median(r[:,6])); print("looseness of the Cauchy-Schwarz form: median %.2f p99 %.2f" % (np.median(r[:,5]), np.percentile(r[:,5],99)))
print("looseness over the ideal: three passes %.2f (p99 %.2f), two passes + sum of four at a position %.2f (p99 %.2f), two passes + four maxima %.2f" %
      (np.median(r[:,1]), np.percentile(r[:,1],99), np.median(r[:,2]), np.percentile(r[:,2],99), np.median(r[:,3])))
for name, col in (("ideal", None), ("three passes", 1), ("two passes, sum at a position", 2), ("two passes, four maxima", 3), ("two passes, 2 sqrt(sum of four squares)", 5), ("energy (Parseval), no passes", 6)):
    L = 1.0 if col is None else r[:, col]
    print("  excluded with %-32s %.1f %%" % (name, 100.0 * np.mean(r[:,0] > L)))
