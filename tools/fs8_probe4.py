#!/usr/bin/env python3
"""Dev: the GPU's stored pattern spectra and products of one search on the tone-burst material against a float64 reference of the
same quantities (the reference built from the GPU's own stored halves one stage up, so that each stage is judged alone)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rank_error_sim import make  # noqa: E402
from sushi_amd import _native  # noqa: E402
from sushi_amd.device import DeviceStream, SearchBatch  # noqa: E402

L = _native.lib()
N, B = L.sushi_hip_fft_size(), L.sushi_hip_fft_block()
H, STEP = N - B, 6
slot = np.array([L.sushi_hip_fft_slot_of_bin(f) for f in range(N)])


def c64(t):
    h = t.cpu().numpy().astype(np.float64).reshape(-1, N, 2)
    return (h[..., 0] + 1j * h[..., 1])[:, slot]


for period in (8.0, 16.0):
    rng = np.random.default_rng(100)
    n = 180000
    dst = make(n, rng, period, True, 0.002, 0.4, 0.5).astype(np.float32)
    m = 9000
    a0 = int(rng.integers(0, n - m))
    src = dst[a0:a0 + m].copy()
    D, S = DeviceStream(dst), DeviceStream(src)
    b = SearchBatch(D, S, [0], [m], [0], [n - m + 1], path="fft", exclusion="never")
    b.run()
    idx, score = b.results()
    d = b.diagnostics()
    Z = c64(D.spectra())
    Tt = c64(b.workspace_view(_native.WS_TSPEC))
    Y = c64(b.workspace_view(_native.WS_Y))
    n_seg = Tt.shape[0]
    print("period", period, "idx", idx, "planted", a0, "score", score, "all_positions", d["all_positions"], "ratios", d["max_bound_ratio"], d["max_bound_ratio_noncandidate"], "pairs", Y.shape[0])
    # pattern spectra against float64 of the samples
    T = src.astype(np.float64)
    ref_t = []
    for s in range(n_seg):
        seg = np.zeros(N); piece = T[s * B:(s + 1) * B]; seg[:piece.shape[0]] = piece
        ref_t.append(np.conj(np.fft.fft(seg)) / N)
    ref_t = np.array(ref_t)
    st = 2.0 ** np.round(np.log2(np.abs(Tt).max() / np.abs(ref_t).max()))
    # stored U = (Re Tt, -Im Tt): the scaled FORWARD transform itself
    e1 = np.abs(np.conj(Tt) / st - ref_t); e2 = np.abs(Tt / st - ref_t)
    eT = np.minimum(e1, e2)
    conjugated = e1.max() < e2.max()
    print("  tspec: scale 2^%d, stored as %s; max |err| / max bin %.2e; excess over half rounding / max bin %.2e" % (
        int(np.log2(st)), "conj" if conjugated else "plain", eT.max() / np.abs(ref_t).max(), np.maximum(eT - 1.5 * 2.0 ** -11 * np.abs(ref_t), 0).max() / np.abs(ref_t).max()))
    Tt_use = np.conj(Tt) if conjugated else Tt
    # products against float64 products of the STORED factors
    worst = (0.0, -1, 0.0)
    for p in range(Y.shape[0]):
        ref = np.zeros(N, complex)
        for s in range(n_seg):
            j = STEP * p + s
            ref += Tt_use[s] * (Z[j] if j < Z.shape[0] else 0)
        if np.abs(ref).max() == 0:
            continue
        sc = 2.0 ** np.round(np.log2(np.abs(Y[p]).max() / np.abs(ref).max()))
        err = np.abs(Y[p] / sc - ref)
        ex = np.maximum(err - 1.5 * 2.0 ** -11 * np.abs(ref), 0)
        f = int(ex.argmax())
        r = float(ex.max() / np.abs(ref).max())
        if r > worst[0]:
            worst = (r, p, f, float(np.abs(ref[f]) / np.abs(ref).max()), float(np.log2(sc)))
    print("  Y: worst excess over half rounding / max bin %.2e at pair %d bin %d (|ref bin| / max %.2e, scale 2^%g)" % worst)
    # the GPU's stored Y through a float64 inverse transform and float32 score arithmetic, against exact scores
    d64 = dst.astype(np.float64)
    mean = float(np.float32(d64.mean()))
    tU = float(T @ T); tn = math.sqrt(tU); sT = float(T.sum())
    s2 = np.concatenate([[0.0], np.cumsum(d64 * d64)])
    xc = np.concatenate([d64 - mean, np.zeros(16 * N)])
    P = n - m + 1
    # y scale: Y = sy / (st sz) sum Tt Z
    nb = (n + B - 1) // B
    be = np.array([float(np.sum(xc[j * B:(j + 1) * B] ** 2)) for j in range(nb + 8)])
    e7 = max(float(be[j:j + 7].sum()) for j in range(nb))
    from rank_error_sim import pow2_under
    sy = pow2_under(32768.0, (64.0 * math.sqrt(n_seg) * tn / N) * (169.33 * math.sqrt(e7)))
    for p in range(Y.shape[0]):
        y = np.fft.ifft(Y[p]) * N / sy
        q0 = STEP * p * B
        pos = np.arange(2 * H)
        valid = q0 + pos < P
        cross = np.concatenate([y.real[:H], y.imag[:H]]) + mean * sT
        p_ok = pos[valid]
        wU = s2[q0 + p_ok + m] - s2[q0 + p_ok]
        score = (tU + wU - 2.0 * cross[valid]) / (tn * np.sqrt(wU))
        span = np.zeros(2 * H + m); piece = d64[q0:q0 + 2 * H + m]; span[:piece.shape[0]] = piece
        nn = 1 << int(math.ceil(math.log2(span.shape[0] + m)))
        ex = np.fft.irfft(np.fft.rfft(span, nn) * np.conj(np.fft.rfft(T, nn)), nn)[:2 * H][valid]
        exact = (tU + wU - 2 * ex) / (tn * np.sqrt(wU))
        sp = xc[q0:(STEP * p + n_seg + 6) * B]
        zn_c = math.sqrt(float(sp @ sp)); sp_u = d64[q0:min(n, (STEP * p + n_seg + 6) * B)]; zn = math.sqrt(float(sp_u @ sp_u))
        max_rs = float((1.0 / np.sqrt(wU)).max())
        q2 = float(np.sum(np.abs(Y[p]) ** 2))
        sigma = math.sqrt(q2 * 7.9472862e-8 * 3 + N * 1.2e-15) / sy
        eps = 2.0 ** -24
        model = eps * max_rs * (2 * 32.0 * zn_c + 16 * zn * zn / tn) + 2 * 8.0 * sigma * max_rs / tn
        err = np.abs(score - exact)
        i = int(err.argmax())
        print("    pair %d: float64 transform of the GPU's Y: max score err %.3e at pos %d (exact %.5f), model %.3e (KE %.2e win %.2e halves %.2e), ratio %.2f; y scale 2^%g" % (
            p, err[i], p_ok[i], exact[i], model, eps * max_rs * 2 * 32.0 * zn_c, eps * max_rs * 16 * zn * zn / tn, 2 * 8.0 * sigma * max_rs / tn, err[i] / model, math.log2(sy)))
