#!/usr/bin/env python3
"""CPU study (numpy float64): the band-split pair bound.  For bench-shaped searches, real overlap-save arithmetic
(N = 16384, B = 4096, pair grid 6 blocks): per pair that does not hold the match,
  margin  = largest |cross term| the pair may have before one of its positions could beat the match,
  B3      = sum over the 16 decimated shares of max |A| (what bound_kernel computes today),
  Bband   = sqrt(2) * sum over the G groups of max |A_g| of the LOW band resampled on a 2L grid
            + sum_s |Tt_s|_out |Z_{6I+s}|_out  (Cauchy-Schwarz over the bins outside the band).
usage: tools/band_bound_sim.py [snr_db] [n_search]"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sushi_amd import synth
from sushi_amd.wav import WavStream
snr = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nsearch = int(sys.argv[2]) if len(sys.argv) > 2 else 10
unrelated = snr < -100
rate = 12000; seconds = 900.0; window = 120.0; offset = 7.25
dst_pcm = synth.make_dst_pcm(seconds, rate, seed=1)
src_pcm = synth.make_src_pcm(synth.make_dst_pcm(seconds, rate, seed=77) if unrelated else dst_pcm, int(round(offset * rate)), snr_db=20.0 if unrelated else snr, seed=2)
dst = WavStream.from_samples(dst_pcm, rate, sample_type="float32")
src = WavStream.from_samples(src_pcm, rate, sample_type="float32")
events = synth.make_events(nsearch, seconds, window + offset, seed=3)
pats, centres, wins = synth.explicit_descriptors(src, dst, events, offset, window, seed=4)
d = dst.data[0].astype(np.float64); mu = d.mean(); dc = d - mu
N = 16384; B = 4096; H = N - B; STEP = 6
nb = (len(d) + B - 1) // B
def zspec(j):
    a = np.zeros(N); b = np.zeros(N)
    s = dc[j * B: j * B + N]; a[:len(s)] = s
    s = dc[j * B + H: j * B + H + N]; b[:len(s)] = s
    return np.fft.fft(a + 1j * b)
zcache = {}
def Z(j):
    if j not in zcache: zcache[j] = zspec(j)
    return zcache[j]
bands = {"N/2": 4096, "N/4": 2048, "N/8": 1024}     # half-width of the band
def band_mask(hw):
    m = np.zeros(N, bool); m[:hw] = True; m[N - hw:] = True; return m
rows = []
for pat, c, w in zip(pats, centres, wins):
    T = np.asarray(pat[0], dtype=np.float64); M = T.shape[0]
    _, lo, P = dst._window(M, c, w)
    W = d[lo:lo + P + M - 1]; Wc = W - mu
    n = 1 << int(math.ceil(math.log2(W.shape[0] + M)))
    corrc = np.fft.irfft(np.fft.rfft(Wc, n) * np.conj(np.fft.rfft(T, n)), n)[:P]
    cs = np.concatenate(([0.0], np.cumsum(W * W))); W2 = cs[M:M + P] - cs[:P]
    t = float(T @ T); sT = float(T.sum())
    score = (t + W2 - 2 * (corrc + mu * sT)) / np.sqrt(t * W2); U = score.min(); amin = int(score.argmin())
    thr = (t + W2 - U * np.sqrt(t * W2)) / 2 - mu * sT          # a position beats U iff its cross term > thr
    nseg = (M + B - 1) // B
    Tt = []
    for s in range(nseg):
        seg = np.zeros(N); x = T[s * B:(s + 1) * B]; seg[:len(x)] = x
        Tt.append(np.conj(np.fft.fft(seg)) / N)
    pair0 = lo // (STEP * B); pair1 = (lo + P - 1) // (STEP * B)
    for I in range(pair0, pair1 + 1):
        q0 = I * STEP * B
        p_lo = max(q0 - lo, 0); p_hi = min(q0 + 2 * H - lo, P)
        if p_hi <= p_lo: continue
        if p_lo <= amin < p_hi: continue
        Y = np.zeros(N, complex)
        for s in range(nseg): Y += Tt[s] * Z(I * STEP + s)
        y = np.fft.ifft(Y) * N                                  # Re y[r] = cross term of position q0 + r, Im: q0 + H + r
        # sanity on the first pair of the first search
        tmin = float(thr[p_lo:p_hi].min())
        A = np.stack([np.fft.ifft(Y[r::16]) * (N / 16) for r in range(16)])
        B3 = np.abs(A).max(axis=1).sum()
        rec = [tmin, max(np.abs(y.real[:H]).max(), np.abs(y.imag[:H]).max()), B3, M]
        for name, hw in bands.items():
            m = band_mask(hw)
            L2 = 4 * hw                                          # 2x oversampled grid for a band of 2 hw bins
            V = np.zeros(L2, complex); V[:hw] = Y[:hw]; V[L2 - hw:] = Y[N - hw:]
            G = L2 // 1024
            Ag = np.stack([np.fft.ifft(V[g::G]) * (L2 / G) for g in range(G)])
            Blow = math.sqrt(2.0) * np.abs(Ag).max(axis=1).sum()
            ylow_true = np.abs(np.fft.ifft(np.where(m, Y, 0)) * N).max()
            Bhigh = sum(math.sqrt(float(np.sum(np.abs(Tt[s][~m]) ** 2))) * math.sqrt(float(np.sum(np.abs(Z(I * STEP + s)[~m]) ** 2))) for s in range(nseg))
            yhigh_true = np.abs(np.fft.ifft(np.where(m, 0, Y)) * N).max()
            rec += [Blow, Bhigh, ylow_true, yhigh_true]
        rows.append(rec)
r = np.array(rows)
print("snr", "unrelated" if unrelated else snr, "pairs", len(r))
print("excluded: ideal %.2f %%  three-pass (today) %.2f %%" % (100 * np.mean(r[:, 0] > r[:, 1]), 100 * np.mean(r[:, 0] > r[:, 2])))
k = 4
for name in bands:
    Bl, Bh, yl, yh = r[:, k], r[:, k + 1], r[:, k + 2], r[:, k + 3]; k += 4
    print("band %-4s excluded %.2f %%   median Blow/margin %.3f  Bhigh/margin %.3f  (true low max/margin %.3f, true high max/margin %.3f)  looseness low %.2f high %.2f" % (
        name, 100 * np.mean(r[:, 0] > Bl + Bh), np.median(Bl / r[:, 0]), np.median(Bh / r[:, 0]), np.median(yl / r[:, 0]), np.median(yh / r[:, 0]),
        np.median(Bl / yl), np.median(Bh / yh)))
    short = r[:, 3] < 24576
    print("           patterns under 6 segments: excluded %.2f %%; others %.2f %%" % (100 * np.mean((r[:, 0] > Bl + Bh)[short]) if short.any() else -1, 100 * np.mean((r[:, 0] > Bl + Bh)[~short]) if (~short).any() else -1))
