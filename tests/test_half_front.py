"""bound_kernel's transform passes in packed halves (fft_core.hpp fft_wave_half_front) against the float32 passes
(fft_wave_mfma_front), on the GPU, by a standalone HIP program over the same header (tools/ubench/half_front_check.hip, built by
__graft_entry__.build()): the largest difference of any A_n1[k2] must be inside the allowance bound_kernel adds for the halves'
rounding (0.29 x the largest pass-1 value), and nothing may be non-finite.  It is also the test that found hipcc 7.2's miscompile
of permlane swaps whose results are bit-cast to half vectors (the workaround is in fft_core.hpp)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "ubench", "half_front_check")


@pytest.mark.gpu
def test_packed_half_passes_equal_the_float32_passes_within_their_allowance():
    if not os.path.exists(EXE):
        pytest.skip("tools/ubench/half_front_check is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    d = json.loads(lines[-1])
    low = json.loads([l for l in lines if "low_max_abs_diff" in l][-1])
    assert out.returncode == 0, d
    assert d["non_finite"] == 0
    assert d["max_abs_diff"] <= d["bound_on_diff"]
    assert d["diff_over_max"] < 0.01                      # measured: 0.0011
    assert abs(d["max_abs_A_half"] - d["max_abs_A_f32"]) <= 0.005 * d["max_abs_A_f32"]
    # the low-band group transform (bound_low_kernel) against a float64 DFT of the same 512 bins: layout, operands and passes
    assert low["low_non_finite"] == 0
    assert low["low_max_abs_diff"] <= low["low_bound_on_diff"]
    assert low["low_diff_over_max"] < 0.01
