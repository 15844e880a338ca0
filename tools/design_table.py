#!/usr/bin/env python3
"""Rewrites the round-6 numbers table of DESIGN.md (between its BEGIN/END markers) from profiles/r06/final/bench_*_n1.json, so that the
table is the committed evidence and nothing else.  usage: tools/design_table.py [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "profiles", "r06", "final")


def L(n):
    with open(os.path.join(O, "bench_%s_n1.json" % n)) as f:
        return json.load(f)


def k(v):
    return "%.1f k" % (v / 1e3)


def lanes(r, whole=False):
    return "the one-sub-batch cut" if whole else "%d : %d" % (r["sub_batches"], r["lanes"])


def pairs(g):
    return "{:,}".format(g["pairs_transformed"])


def stages(r):
    st = r["stage_ms"]
    return " / ".join("%.2f" % st[x] for x in ("tspec", "mac", "bound", "ifft", "refine", "finish"))


rows = []
d = L("cfg2"); r = d["roofline"]; g = r["diagnostics"]; p = d["parity"]
clean_ms = d["ms_per_step"]
rows.append("| **configs[2] float32** (target ≥ 1000) | **%s** | **%.2f** | %s | **%.3f** (%.2f of 8 TB/s); %.1f GB = **%.2f × algorithmic** at %.2f TB/s | %s (of %s; %s) | band | %d, %s (max \\|Δscore\\| %.1e); CPU port %.1f events/s on %d cores |" % (
    k(d["value"]), d["ms_per_step"], lanes(r), r["frac"], r["achieved"] / 1e3, r["step_traffic_bytes"] / 1e9, r["step_traffic_over_algorithmic"],
    r["traffic"] / 1e3, pairs(g), "{:,}".format(r["fft_pairs"]), "{:,}".format(g["excluded_audited"]), p["oracle_sample_searches"],
    p["max_idx_err_vs_oracle_sample"], p["max_abs_score_err_vs_oracle_sample"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
d = L("lanes1"); r = d["roofline"]; g = r["diagnostics"]; p = d["parity"]
rows.append("| … the same box, one sub-batch on one stream (`SUSHI_HIP_LANES=1:1`: round 5's way) | %s | %.2f | 1 : 1 — stages %s | %.3f | %s | band | %d, %s |" % (
    k(d["value"]), d["ms_per_step"], stages(r), r["frac"], pairs(g), p["oracle_sample_searches"], p["max_idx_err_vs_oracle_sample"]))


def simple(name, label, form, traffic=True, whole=False, extra_oracle=""):
    d = L(name); r = d["roofline"]; g = r["diagnostics"]; p = d["parity"]
    t = "%.3f" % r["frac"]
    if traffic and r.get("step_traffic_bytes"):
        t += "; %.1f GB" % (r["step_traffic_bytes"] / 1e9)
    return "| %s | %s | %.2f | %s | %s | %s | %s | %d, %s%s |" % (label, k(d["value"]), d["ms_per_step"], lanes(r, whole), t, pairs(g), form,
                                                          p["oracle_sample_searches"], p["max_idx_err_vs_oracle_sample"], extra_oracle)


rows.append(simple("stat20", "… with round 5's statistical bound (`SUSHI_HIP_BOUND_MODEL=statistical`)", "band", traffic=False))
rows.append(simple("cfg2u8", "configs[2] uint8 (its algorithmic bytes are ¼)", "band", extra_oracle=" (bit-exact)"))
d = L("hard")
rows.append(simple("hard", "configs[2], 5 %% tie-saturated events (%.2f × the clean step)" % (d["ms_per_step"] / clean_ms), "band", extra_oracle=" (all 150 hard + every flagged one)"))
rows.append(simple("cc", "configs[2] `TM_CCOEFF_NORMED` + argmax", "band"))
rows.append(simple("cfg1", "configs[1] float32 (59,555 pairs)", "band"))
d = L("cfg4")
rows.append(simple("cfg4", "configs[4] float32 (24 kHz, 4 h, 5000 events; 1,176,852 pairs)", "band", extra_oracle="; CPU port %.1f events/s" % d["cpu_baseline"]["value"]))
rows.append(simple("encode", "**what Sushi's inputs look like**: `--source encode` (another encode: gain 0.7, 4 kHz low-pass, requantised to 8 bits)", "band"))
rows.append(simple("encodeu8", "… the same as uint8 streams", "band"))
rows.append(simple("partial", "`--source partial` (second half of the programme is another cut: 1455 of 3000 events find nothing)", "band; the dense form for the second half's searches only"))
rows.append(simple("dub", "`--source dub` (shared music bed, each stream's OWN speech on half of the time: 1902 of 3000 events under speech; 126.4 k events/s before the dense searches were regrouped, §3.2)", "band; the searches under speech in the dense form, in items of their own"))
rows.append(simple("snr12", "configs[2], source noise 12 dB", "band"))
rows.append(simple("snr6", "… 6 dB", "band"))
rows.append(simple("stat6", "… 6 dB with the statistical bound", "band"))
rows.append(simple("snr0", "… 0 dB", "whole rows (4 % of the votes)", whole=True))
rows.append(simple("unrelated", "source UNRELATED to the destination", "none (AUTO has suspended the exclusion)", whole=True))
rows.append(simple("whole", "`--exclusion whole`", "whole rows", whole=True))
rows.append(simple("never", "`--exclusion never`", "–", whole=True))
d = L("cfg2")
head = "| workload | events/s | ms per step | sub-batches : lanes | whole-step roofline fraction; PMC traffic | pairs transformed (of; audited) | form | oracle sample: searches, max index error |\n|---|---|---|---|---|---|---|---|\n"
tail = "\nFirst run of the batch (`first_step_ms`) %.2f ms; set-up %.1f + %.1f ms; one-shot %s events/s (%s with the process's start-up, %.0f ms on this box, on the critical path).\n" % (
    d["first_step_ms"], d["setup_ms"]["streams_upload_prefix_sums_spectra"], d["setup_ms"]["batch_plan_allocate_upload"], k(d["one_shot_events_per_s"]),
    k(d["one_shot_incl_process_start_events_per_s"]), d["process_start_ms"])
se = d["shard_emulation"]["by_world_size"]
tail += "\n| G | slowest shard (ms) | mean shard (ms) | implied speed-up over one GPU | work of the slowest shard over the mean | launch chain of one shard's `run()` on the host |\n|---|---|---|---|---|---|\n"
for gk, v in se.items():
    hl = [s["host_launch_ms"] for s in v["shards"]]
    tail += "| %s | %.2f | %.2f | %.2f%s | %.2f | %.2f – %.2f ms |\n" % (gk, v["max_shard_ms"], v["mean_shard_ms"], v["implied_speedup_over_one_gpu"],
                                                                   " (%.2f M events/s, gather excluded)" % (v["implied_events_per_s_gather_excluded"] / 1e6) if gk == "8" else "",
                                                                   max(s["work_over_mean"] for s in v["shards"]), min(hl), max(hl))
block = head + "\n".join(rows) + "\n" + tail
BEGIN, END = "<!-- BEGIN r06 table (tools/design_table.py) -->\n", "<!-- END r06 table -->\n"
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a, z = s.index(BEGIN) + len(BEGIN), s.index(END)
if "--check" in sys.argv:
    sys.exit(0 if s[a:z] == block else 1)
open(path, "w").write(s[:a] + block + s[z:])
print("DESIGN.md: table rewritten from", O)
