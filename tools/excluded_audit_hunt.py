#!/usr/bin/env python3
"""GPU: the audit of the pair exclusion, at length.  Every run of a batch transforms, for every audited search,
one pair its lower bound had EXCLUDED -- a different one every run -- and holds the bound to what that pair really scores
(SushiHipBatchDiag.excluded_audited / max_slb_ratio_excluded / slb_violations).  Here: SUSHI_HIP_AUDIT_EVERY=1 (every search, every
run), many runs, both forms of the exclusion, on (a) the bench's own job at BASELINE configs[2] sizes and (b) the stress materials of
tools/bound_hunt.py with windows of +-60 s.  One JSON line per (material, form); a violation anywhere is an error.
usage: excluded_audit_hunt.py [runs_big] [runs_stress]   ->  profiles/r05/excluded_audit_hunt.jsonl"""
import json
import os
import sys

os.environ["SUSHI_HIP_AUDIT_EVERY"] = "1"
os.environ["SUSHI_HIP_LOAD"] = "host"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from sushi_amd import synth  # noqa: E402
from sushi_amd.device import DeviceStream, SearchBatch  # noqa: E402
from sushi_amd.wav import WavStream  # noqa: E402
import bound_hunt  # noqa: E402


def hunt(name, D, S, offs, lens, wst, npos, runs, method="sqdiff_normed"):
    out = []
    ref = None
    for form in ("band", "whole"):
        b = SearchBatch(D, S, offs, lens, wst, npos, path="fft", exclusion=form, method=method, workspace_bytes=160 << 30)
        tot, worst, viol, allp, tot2 = 0, 0.0, 0, 0, 0
        for r in range(runs):
            b.run()
            idx, score = b.results()
            d = b.diagnostics()
            tot2 += d["second_look_audited"]
            tot += d["excluded_audited"]; worst = max(worst, d["max_slb_ratio_excluded"]); viol += d["slb_violations"]; allp += d["all_positions"]
            if ref is None:
                ref = (idx.copy(), score.copy().view(np.uint32))
            assert (idx == ref[0]).all() and (score.view(np.uint32) == ref[1]).all(), (name, form, r)
        line = {"material": name, "form": form, "method": method, "searches": len(offs), "pairs": b.fft_pairs, "runs": runs,
                "excluded_pairs_audited": int(tot), "of_them_excluded_by_the_second_look": int(tot2), "max_slb_ratio_excluded": float(worst), "slb_violations": int(viol),
                "all_positions": int(allp), "pairs_transformed_last_run": int(d["pairs_transformed"])}
        print(json.dumps(line), flush=True)
        out.append(line)
        del b
        torch.cuda.empty_cache()
    return out


def main():
    runs_big = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    runs_stress = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    lines = []
    # (a) the bench's job: 3000 events, 2-h streams, +-120 s
    rate, seconds, n_ev, window, off = 12000, 7200.0, 3000, 120.0, 7.25
    seed = 20260924 + 2
    dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
    src_pcm = synth.make_src_pcm(dst_pcm, int(round(off * rate)), seed=seed + 1)
    for st in ("float32", "uint8"):
        dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=st)
        src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=st)
        events = synth.make_events(n_ev, seconds, window + off, seed=seed + 2)
        pats, centres, wins = synth.explicit_descriptors(src, dst, events, off, window, seed=seed + 3)
        offs = [src._get_sample_for_time(s) for s, _ in events]
        lens = [p.shape[1] for p in pats]
        wst, npos = [], []
        for m, c, w in zip(lens, centres, wins):
            _, lo, p = dst._window(m, c, w)
            wst.append(lo); npos.append(p)
        D, S = DeviceStream(dst.data[0]), DeviceStream(src.data[0])
        lines += hunt("bench configs[2] " + st, D, S, offs, lens, wst, npos, runs_big if st == "float32" else max(4, runs_big // 4))
        if st == "float32":
            lines += hunt("bench configs[2] float32", D, S, offs, lens, wst, npos, max(4, runs_big // 4), method="ccoeff_normed")
        del D, S, dst, src
        torch.cuda.empty_cache()
    # (b) the stress materials, windows of +-60 s (59 pairs each)
    rng = np.random.default_rng(5)
    for kind in bound_hunt.KINDS:
        n = 3_000_000
        x = bound_hunt.make(kind, n, rng)
        for u8 in (False, True):
            dst = (x * 255 + 0.5).astype(np.uint8) if u8 else x.astype(np.float32)
            offs, lens, wst, npos, parts, pos = [], [], [], [], [], 0
            for k in range(24):
                m = int(rng.choice([2000, 9000, 30000, 60000]))
                a = int(rng.integers(750000, n - 750000 - m))
                piece = dst[a:a + m].astype(np.float64) + rng.standard_normal(m) * (3.0 if u8 else 0.01)
                parts.append(np.clip(piece, 0, 255 if u8 else None).astype(dst.dtype))
                w0 = a - int(rng.integers(100000, 720000))
                offs.append(pos); lens.append(m); wst.append(w0); npos.append(1_440_001 if w0 + 1_440_001 + m <= n else n - m - w0 + 1)
                pos += m
            src = np.concatenate(parts)
            lines += hunt("stress %s %s" % (kind, "uint8" if u8 else "float32"), DeviceStream(dst), DeviceStream(src), offs, lens, wst, npos, runs_stress)
    tot = sum(l["excluded_pairs_audited"] for l in lines)
    tot2 = sum(l["of_them_excluded_by_the_second_look"] for l in lines)
    worst = max(l["max_slb_ratio_excluded"] for l in lines)
    viol = sum(l["slb_violations"] for l in lines)
    print(json.dumps({"total_excluded_pairs_audited": tot, "of_them_excluded_by_the_second_look": tot2, "max_slb_ratio_excluded": worst, "slb_violations": viol}))
    sys.exit(1 if viol else 0)


if __name__ == "__main__":
    main()
