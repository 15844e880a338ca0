"""The FFT path's ranking bound on material chosen to break the assumptions behind it -- the f32 transform
error model and, since the products are kept as packed halves, the quantisation model: slowly drifting DC, amplitude steps over
four orders of magnitude, sparse spikes, pure tones, quantised staircases, bursts of a tone at exactly Fs/8 (the edge of the low band) on block boundaries; uint8 and float32 at several magnitudes.  Every search
must equal the oracle and no evaluated position -- candidate or audited non-candidate -- may be further from its f32 score
than the pair's bound (a violation would show as `all_positions`).  tools/bound_hunt.py is the long version."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", ["drift", "steps", "spikes", "tones", "staircase", "noise", "fs8burst"])
def test_bound_holds_on_adversarial_material(oracle, kind):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bound_hunt
    from sushi_amd.device import DeviceStream, SearchBatch
    from test_gpu_parity import _check_f32, _check_u8
    rng = np.random.default_rng({"drift": 1, "steps": 2, "spikes": 3, "tones": 4, "staircase": 5, "noise": 6, "fs8burst": 7}[kind])
    for u8, mag in ((True, 1.0), (False, 1.0), (False, 300.0), (False, 1e-3)):
        n = 180000
        x = bound_hunt.make(kind, n, rng)
        dst = (x * 255 + 0.5).astype(np.uint8) if u8 else (x * mag).astype(np.float32)
        offs, lens, wst, npos, parts, pos = [], [], [], [], [], 0
        for k, m in enumerate([300, 4096, 9000, 30000, 50000]):
            a = int(rng.integers(0, n - m))
            piece = dst[a:a + m].astype(np.float64)
            if k % 2:
                piece = piece + rng.standard_normal(m) * (3.0 if u8 else 0.01 * mag)
            parts.append(np.clip(piece, 0, 255 if u8 else None).astype(dst.dtype))
            w0 = int(rng.integers(0, max(1, a)))
            offs.append(pos); lens.append(m); wst.append(w0); npos.append(n - m - w0 + 1)
            pos += m
        src = np.concatenate(parts)
        b = SearchBatch(DeviceStream(dst), DeviceStream(src), offs, lens, wst, npos, path="fft")
        b.run()
        idx, score = b.results()
        d = b.diagnostics()
        assert d["all_positions"] == 0 and d["max_bound_ratio"] < 1.0 and d["max_bound_ratio_noncandidate"] < 1.0, d
        for k in range(len(offs)):
            res = oracle.match_template(dst[wst[k]:wst[k] + npos[k] + lens[k] - 1], src[offs[k]:offs[k] + lens[k]])[0]
            (_check_u8 if u8 else _check_f32)(res, idx[k], score[k])
        # both forms of the exclusion, several runs each (every run transforms ANOTHER excluded pair of every second search and holds
        # its lower bound to what it really scores): no bound above a real score, the same results
        for form in ("band", "whole"):
            bf = SearchBatch(b.dst, b.src, offs, lens, wst, npos, path="fft", exclusion=form)
            for _ in range(4):
                bf.run()
                i2, s2 = bf.results()
                df = bf.diagnostics()
                assert df["slb_violations"] == 0 and df["all_positions"] == 0 and df["max_slb_ratio_excluded"] < 1.0, (form, df)
                assert (i2 == idx).all() and (s2.view(np.uint32) == score.view(np.uint32)).all(), form
