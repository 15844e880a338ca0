#!/usr/bin/env python3
"""CPU study (numpy, float64 transforms; no GPU): what rounding the block spectra, the pattern spectra and their products to packed
halves does to a pair's cross terms, source by source, against the ranking stage's model of it (sushi_fft.hip pair_error_model:
8 standard deviations of independent roundings, variance from the energy of the row) and against the worst case the excluded
side now uses (slb_one: c |T| sqrt(8) zn_c).  On the stress materials of tools/bound_hunt.py.
usage: tools/half_rounding_sim.py [kind] [u8|f32] [mag]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bound_hunt  # noqa: E402

N, B, H, STEP = 16384, 4096, 12288, 6


def pow2_under(target, bound):
    if not bound > 0:
        return 1.0
    k = int(math.floor(math.log2(target / bound)))
    return math.ldexp(1.0, max(-60, min(60, k)))


def half(z):
    return (z.real.astype(np.float16).astype(np.float64)) + 1j * (z.imag.astype(np.float16).astype(np.float64))


def pair_terms(xc, T, pairI, e7, rz=True, rt=True, ry=True):
    """Re / Im y of pair `pairI` from the stored (optionally rounded) spectra, and the row energy the model sees."""
    M = T.shape[0]
    n_seg = (M + B - 1) // B
    tn = math.sqrt(float(T @ T))
    sz = pow2_under(32768.0, 181.02 * math.sqrt(e7))
    st = pow2_under(8192.0, 64.0 * tn / N)
    sy = pow2_under(32768.0, (64.0 * math.sqrt(n_seg) * tn / N) * (169.33 * math.sqrt(e7)))
    Y = np.zeros(N, complex)
    for s in range(n_seg):
        seg = np.zeros(N); piece = T[s * B:(s + 1) * B]; seg[:piece.shape[0]] = piece
        Tt = np.conj(np.fft.fft(seg)) / N * st
        j = STEP * pairI + s
        a = np.zeros(N); b = np.zeros(N)
        xa = xc[j * B:j * B + N]; xb = xc[j * B + H:j * B + H + N]
        a[:xa.shape[0]] = xa; b[:xb.shape[0]] = xb
        Z = np.fft.fft(a + 1j * b) * sz
        if rt: Tt = half(Tt)
        if rz: Z = half(Z)
        Y += Tt * Z
    Y *= sy / (st * sz)
    if ry: Y = half(Y)
    y = np.fft.ifft(Y) * N / sy
    return y, float(np.sum(np.abs(Y) ** 2)), sy, tn, n_seg


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fs8burst"
    u8 = (sys.argv[2] if len(sys.argv) > 2 else "u8") == "u8"
    mag = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rng = np.random.default_rng(7)
    n = 180000
    x = bound_hunt.make(kind, n, rng)
    dst = (x * 255 + 0.5).astype(np.uint8).astype(np.float64) if u8 else (x * mag).astype(np.float32).astype(np.float64)
    mean = float(np.float32(dst.mean()))
    xc = dst - mean
    # largest centred energy of 7 consecutive blocks
    nb = (n + B - 1) // B
    be = np.array([float(np.sum(xc[j * B:(j + 1) * B] ** 2)) for j in range(nb)])
    e7 = max(float(be[j:j + 7].sum()) for j in range(nb))
    worst = {}
    for m in (4096, 9000, 30000):
        a0 = int(rng.integers(0, n - m))
        T = dst[a0:a0 + m].copy()
        n_pairs = (n - m) // (2 * H) - 1
        for pairI in range(0, max(1, n_pairs)):
            exact, _, _, tn, n_seg = pair_terms(xc, T, pairI, e7, False, False, False)
            span = xc[STEP * pairI * B:(STEP * pairI + n_seg + 6) * B]
            zn_c = math.sqrt(float(span @ span))
            for name, flags in (("Z", (True, False, False)), ("T", (False, True, False)), ("Y", (False, False, True)), ("all", (True, True, True))):
                y, q2, sy, _, _ = pair_terms(xc, T, pairI, e7, *flags)
                err = max(np.abs((y - exact).real[:H]).max(), np.abs((y - exact).imag[:H]).max())
                sigma = math.sqrt(q2 * 7.9472862e-8 * 3 + N * 1.2e-15) / sy
                wc = (3 * 4.8829e-4 * 1.001 + 4e-5) * 2.8284272 * tn * zn_c
                key = (m, name)
                w = worst.setdefault(key, [0.0, 0.0, 0.0])
                w[0] = max(w[0], err / (8 * sigma)); w[1] = max(w[1], err / wc); w[2] = max(w[2], err / (tn * zn_c + 1e-300))
    print(kind, "u8" if u8 else "f32 x%g" % mag)
    for (m, name), w in sorted(worst.items()):
        print("  M %6d  rounding %-3s  max |err| / (8 sigma model) %.3f   / worst-case bound %.4f   / (|T| zn_c) %.2e" % (m, name, w[0], w[1], w[2]))


if __name__ == "__main__":
    main()
