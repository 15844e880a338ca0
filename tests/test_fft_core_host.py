"""The FFT path's device code that can run on the CPU: sushi_amd/csrc/fft_core.hpp (workgroup
FFT, emulated one thread at a time) and mac_core.hpp (ring-buffered frequency-domain
multiply-accumulate), compiled with g++ and checked against float64 definitions."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_and_run(tmp_path, name):
    exe = os.path.join(tmp_path, name)
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(HERE, name + ".cpp"), "-o", exe])
    return subprocess.run([exe], capture_output=True, text=True)


def test_workgroup_fft_on_host(tmp_path):
    r = _build_and_run(tmp_path, "host_fft_check")
    assert r.returncode == 0, r.stdout + r.stderr
    errs = [float(x) for x in r.stdout.split()]          # forward / inverse: 8192 points (512 x 16), 16384 points (1024 x 16),
    assert len(errs) == 6 and max(errs) < 8e-7           # 16384 points wave plan; relative L2 error of an f32 FFT of these lengths


def test_mac_ring_on_host(tmp_path):
    r = _build_and_run(tmp_path, "host_mac_check")
    assert r.returncode == 0, r.stdout + r.stderr
    assert float(r.stdout.split()[0]) < 1e-5


def test_mfma_first_pass_on_host(tmp_path):
    """The inverse transform with its first pass on the matrix pipe, emulated lane by lane (stored order -> 16-byte loads ->
    v_perm_b32 / v_permlane32_swap -> v_mfma_f32_16x16x32_f16 -> the wave plan's passes 2 .. 4) against a float64 FFT."""
    r = _build_and_run(tmp_path, "host_mfma_check")
    assert r.returncode == 0, r.stdout + r.stderr
    assert float(r.stdout.split()[0]) < 8e-7


def test_twiddle_table_is_correctly_rounded():
    from sushi_amd import build
    path = build.write_twiddles()
    vals = np.array([float(t.strip().rstrip("f")) for t in open(path).read().replace("\n", " ").split(",") if t.strip()],
                    dtype=np.float64)
    assert vals.shape[0] == 2 * 16384
    k = np.arange(16384)
    assert (vals[0::2].astype(np.float32) == np.cos(2 * np.pi * k / 16384).astype(np.float32)).all()
    assert (vals[1::2].astype(np.float32) == (-np.sin(2 * np.pi * k / 16384)).astype(np.float32)).all()


@pytest.mark.parametrize("ratio", [2, 4])
def test_overlap_save_formulation_matches_definition(oracle, ratio):
    """NumPy model of the FFT path's block arithmetic (packing of two real blocks H = N - B apart into one complex
    block, segmenting of patterns, the absolute pair grid, which output goes where) against the definition, for
    N = 2 B (one valid block per half) and N = 4 B (three: the product's geometry).  The HIP kernels implement exactly
    this layout (csrc/sushi_common.hpp fft_layout)."""
    B = 256
    N = ratio * B
    H = N - B
    STEP = 2 * H // B
    rng = np.random.default_rng(1)
    n_dst = 20000
    xc = ((rng.standard_normal(n_dst) * 0.2 + 0.5).clip(0, 1).astype(np.float32) - np.float32(0.5))
    tc = ((rng.standard_normal(5000) * 0.2 + 0.5).clip(0, 1).astype(np.float32) - np.float32(0.5))
    J = -(-n_dst // B)
    pad = np.zeros((J + 3) * B + 2 * N + H, np.float32)
    pad[:n_dst] = xc
    Z = np.stack([np.fft.fft(pad[j * B:j * B + N].astype(np.complex128) + 1j * pad[j * B + H:j * B + H + N])
                  for j in range(J)])

    def getz(j):
        return Z[j] if j < J else np.zeros(N, np.complex128)

    for (to, M, w, P) in [(100, 1300, 3000, 9000), (0, 255, 0, 100), (7, 256, 255, 513), (50, 3000, 1234, 13000),
                          (0, 700, n_dst - 1500, 801), (3, 1, 1535, 2)]:
        pair0, pair_last = (w // B) // STEP, ((w + P - 1) // B) // STEP
        S = -(-M // B)
        tt = []
        for s in range(S):
            seg = np.zeros(N)
            ln = min(B, M - s * B)
            seg[:ln] = tc[to + s * B:to + s * B + ln]
            tt.append(np.conj(np.fft.fft(seg)) / N)
        corr = np.full(P, np.nan)
        for I in range(pair0, pair_last + 1):
            Y = sum(tt[s] * getz(STEP * I + s) for s in range(S))
            y = np.fft.ifft(Y) * N
            for half, vals in ((0, y.real), (1, y.imag)):
                p = STEP * I * B + half * H + np.arange(H) - w
                ok = (p >= 0) & (p < P)
                corr[p[ok]] = vals[:H][ok]
        ref = np.array([np.dot(tc[to:to + M].astype(np.float64), pad[w + p:w + p + M].astype(np.float64))
                        for p in range(P)])
        assert not np.isnan(corr).any()
        assert np.abs(corr - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
