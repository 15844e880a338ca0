#!/usr/bin/env python3
"""CPU study: the ranking stage's f32 scores of every position of every pair against exact float64 scores, with the stage's
arithmetic emulated -- float32 transforms (scipy.fft on complex64), packed-half storage of Z, Tt and Y -- and the pair's modelled error
bound (sushi_fft.hip pair_error_model).  Which material breaks the model?  usage: rank_error_sim.py period floor [seed]"""
import math
import os
import sys

import numpy as np
import scipy.fft as sfft

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N, B, H, STEP, KE, KQ = 16384, 4096, 12288, 6, 32.0, 8.0


def make(n, rng, period, aligned, floor, amp, dc):
    t = np.arange(n, dtype=np.float64)
    base = np.convolve(rng.standard_normal(n + 15), np.ones(16) / 16.0, mode="valid")
    x = dc + floor * base
    step = 4096 if aligned else 4096 + 777
    for b0 in range(0, n - 4096, step * int(rng.integers(2, 5))):
        ln = int(rng.integers(1, 4)) * (4096 if aligned else 3000)
        ph = rng.uniform(0, 2 * np.pi)
        x[b0:b0 + ln] += rng.uniform(amp / 2, amp) * np.sin(2 * np.pi * t[b0:b0 + ln] / period + ph)
    return np.clip(x, 0.0, 1.0)


def pow2_under(target, bound):
    if not bound > 0:
        return 1.0
    return math.ldexp(1.0, max(-60, min(60, int(math.floor(math.log2(target / bound))))))


def half(z):
    return (z.real.astype(np.float16).astype(np.float32)) + 1j * (z.imag.astype(np.float16).astype(np.float32))


def main():
    period = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    f32fft = (sys.argv[4] if len(sys.argv) > 4 else "f32") == "f32"
    rng = np.random.default_rng(seed)
    n = 180000
    dst = make(n, rng, period, True, floor, 0.4, 0.5).astype(np.float32)
    d64 = dst.astype(np.float64)
    mean = float(np.float32(d64.mean()))
    xc32 = (dst - np.float32(mean)).astype(np.float32)
    xc = np.concatenate([xc32.astype(np.float64), np.zeros(16 * N)])
    nb = (n + B - 1) // B
    be = np.array([float(np.sum(xc[j * B:(j + 1) * B] ** 2)) for j in range(nb + 8)])
    e7 = max(float(be[j:j + 7].sum()) for j in range(nb))
    sz = pow2_under(32768.0, 181.02 * math.sqrt(e7))
    fft = (lambda v: sfft.fft(v.astype(np.complex64))) if f32fft else (lambda v: np.fft.fft(v.astype(np.complex128)))
    ifft = (lambda v: sfft.ifft(v.astype(np.complex64)) * np.float32(N)) if f32fft else (lambda v: np.fft.ifft(v.astype(np.complex128)) * N)
    Zs = {}

    def Z(j):
        if j not in Zs:
            Zs[j] = half(fft(xc[j * B:j * B + N] + 1j * xc[j * B + H:j * B + H + N]) * np.float32(sz))
        return Zs[j]
    s2 = np.concatenate([[0.0], np.cumsum(d64 * d64)])
    worst = 0.0
    for k, m in enumerate([4096, 9000, 30000, 50000]):
        a0 = int(rng.integers(0, n - m))
        T = d64[a0:a0 + m].copy()
        if k % 2:
            T = np.clip(T + rng.standard_normal(m) * 0.01, 0, None).astype(np.float32).astype(np.float64)
        tU = float(T @ T); tn = math.sqrt(tU); sT = float(T.sum())
        n_seg = (m + B - 1) // B
        st = pow2_under(8192.0, 64.0 * tn / N)
        sy = pow2_under(32768.0, (64.0 * math.sqrt(n_seg) * tn / N) * (169.33 * math.sqrt(e7)))
        Tts = []
        for s in range(n_seg):
            seg = np.zeros(N); piece = T[s * B:(s + 1) * B]; seg[:piece.shape[0]] = piece
            Tts.append(half(np.conj(fft(seg)) / np.float32(N) * np.float32(st)))
        P = n - m + 1
        for pairI in range(0, (P + 2 * H - 1) // (2 * H)):
            Y = np.zeros(N, np.complex64)
            for s in range(n_seg):
                Y = Y + (Tts[s] * Z(STEP * pairI + s)).astype(np.complex64)
            Y = half(Y * np.float32(sy / (st * sz)))
            y = ifft(Y)
            q2 = float(np.sum(np.abs(Y.astype(np.complex128)) ** 2))
            q0 = STEP * pairI * B
            pos = np.arange(2 * H)
            valid = q0 + pos < P
            cross = np.concatenate([y.real[:H], y.imag[:H]]).astype(np.float64) / sy + mean * sT
            p_ok = pos[valid]
            wU = s2[q0 + p_ok + m] - s2[q0 + p_ok]
            score32 = ((np.float32(tU) + wU.astype(np.float32) - np.float32(2.0) * cross[valid].astype(np.float32)) /
                       (np.float32(tn) * np.sqrt(wU.astype(np.float32)))).astype(np.float64)
            # exact scores of the same positions (float64 direct sums via FFT in float64 over this pair's span)
            span = np.zeros(2 * H + m); piece = d64[q0:q0 + 2 * H + m]; span[:piece.shape[0]] = piece
            nn = 1 << int(math.ceil(math.log2(span.shape[0] + m)))
            ex = np.fft.irfft(np.fft.rfft(span, nn) * np.conj(np.fft.rfft(T, nn)), nn)[:2 * H][valid[:2 * H]]
            exact = (tU + wU - 2 * ex) / (tn * np.sqrt(wU))
            sp = xc[q0:(STEP * pairI + n_seg + 6) * B]
            zn_c = math.sqrt(float(sp @ sp)); sp_u = d64[q0:min(n, (STEP * pairI + n_seg + 6) * B)]; zn = math.sqrt(float(sp_u @ sp_u))
            max_rs = float((1.0 / np.sqrt(wU)).max())
            sigma = math.sqrt(q2 * 7.9472862e-8 * 3 + N * 1.2e-15) / sy
            eps = 2.0 ** -24
            model = eps * max_rs * (2 * KE * zn_c + 16 * zn * zn / tn) + 2 * KQ * sigma * max_rs / tn
            err = float(np.abs(score32 - exact).max())
            worst = max(worst, err / model)
            if err / model > 0.8:
                i = int(np.abs(score32 - exact).argmax())
                print("  M %d pair %d: err %.3e model %.3e (KE part %.2e, window part %.2e, halves part %.2e) ratio %.2f at pos %d score %.4f" % (
                    m, pairI, err, model, eps * max_rs * 2 * KE * zn_c, eps * max_rs * 16 * zn * zn / tn, 2 * KQ * sigma * max_rs / tn, err / model, p_ok[i], exact[i]))
    print("period %g floor %g seed %d %s transforms: worst err / model %.3f" % (period, floor, seed, "float32" if f32fft else "float64", worst))


if __name__ == "__main__":
    main()
