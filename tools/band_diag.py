#!/usr/bin/env python3
"""GPU: how sharp is the band-split pair bound next to the whole-row one?  For a batch of searches run both forms, print the
diagnostics and the per-pair bounds (slb; acc[0] = the cross term's bound in the units of the stored products)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PAIR = 6 * 4096

def stream(n, seed, lowpass=8):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n + lowpass)
    c = np.cumsum(x)
    y = (c[lowpass:] - c[:-lowpass]) / lowpass
    y = y / np.abs(y).max() * 0.35 + 0.5
    return y.astype(np.float32)

def main():
    from sushi_amd.device import DeviceStream, SearchBatch
    n = 40 * PAIR
    dst = stream(n, 1)
    rng = np.random.default_rng(2)
    src = (dst + rng.standard_normal(n).astype(np.float32) * 0.02).clip(0, 1).astype(np.float32)
    offs, lens, wst, npos = [], [], [], []
    for k in range(12):
        m = int(rng.integers(12000, 60000))
        a = int(rng.integers(5 * PAIR, n - 5 * PAIR - m))
        ws = a - int(rng.integers(PAIR, 4 * PAIR))
        p = 8 * PAIR + int(rng.integers(0, 5000))
        offs.append(a); lens.append(m); wst.append(ws); npos.append(min(p, n - ws - m + 1))
    D, S = DeviceStream(dst), DeviceStream(src)
    out = {}
    for form in ("whole", "band"):
        b = SearchBatch(D, S, offs, lens, wst, npos, path="fft", exclusion=form)
        b.run()
        idx, score = b.results()
        d = b.diagnostics()
        slb, acc = b.pair_bounds()
        out[form] = (slb, acc, idx, score)
        print(form, {k: d[k] for k in ("pairs_transformed", "excluded_audited", "max_slb_ratio_excluded", "slb_violations", "band", "flagged")}, "pairs", b.fft_pairs)
    sw, aw = out["whole"][0], out["whole"][1]
    sb, ab = out["band"][0], out["band"][1]
    print("scores", out["band"][3][:6])
    print("idx equal", (out["band"][2] == out["whole"][2]).all())
    fin = np.isfinite(sw) & np.isfinite(sb)
    print("pairs with finite slb: whole %d band %d" % (np.isfinite(sw).sum(), np.isfinite(sb).sum()))
    print("slb whole: median %.4f  band: median %.4f" % (np.median(sw[fin]), np.median(sb[fin])))
    print("acc0 whole (B): median %.1f; band (B_low): median %.1f; ratio of medians %.3f" % (np.median(aw[:, 0]), np.median(ab[:, 0]), np.median(ab[:, 0]) / np.median(aw[:, 0])))
    for i in range(0, min(24, len(sw))):
        print(i, "slb whole %.4f band %.4f  B whole %.1f  B_low %.1f  q_low %.3g" % (sw[i], sb[i], aw[i, 0], ab[i, 0], ab[i, 1]))

if __name__ == "__main__":
    main()
