#!/usr/bin/env python3
"""Dev: from a rocprofv3 kernel trace (…_kernel_trace.csv) of bench.py, how busy the GPU is inside a step on lanes: for the last steps,
the span from the fill kernel to the unpack kernel, the union of the kernels' intervals inside it, their summed durations (concurrency)
and the longest gaps.  usage: tools/timeline.py kt_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
fills = [i for i, e in enumerate(ev) if "fill_ranges_kernel" in e[2]]
unpacks = [i for i, e in enumerate(ev) if "unpack_keys_kernel" in e[2]]
out = []
for f in fills[-4:]:
    u = next((j for j in unpacks if j > f), None)
    if u is None:
        continue
    t0, t1 = ev[f][0], ev[u][1]
    ks = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    ks.sort()
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for s, e, n in ks:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, n in ks)
    gaps.sort(reverse=True)
    # time at each level of concurrency, and who runs ALONE for how long
    pts = sorted([(s_, 1, n) for s_, e_, n in ks] + [(e_, -1, n) for s_, e_, n in ks])
    level, last, at, alone, running = 0, t0, {}, {}, {}
    for t, dlt, n in pts:
        if t > last and level > 0:
            at[level] = at.get(level, 0) + (t - last)
            if level == 1:
                who = next(iter(running))
                alone[who] = alone.get(who, 0) + (t - last)
        last = t
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:28]
        if dlt > 0:
            running[short] = running.get(short, 0) + 1
        else:
            running[short] -= 1
            if running[short] == 0:
                del running[short]
        level += dlt
    out.append({"time_at_concurrency_ms": {k_: round(v / 1e6, 2) for k_, v in sorted(at.items())},
                "alone_ms": {k_: round(v / 1e6, 2) for k_, v in sorted(alone.items(), key=lambda x: -x[1])[:8]},
                "span_ms": round((t1 - t0) / 1e6, 3), "busy_ms": round(busy / 1e6, 3), "sum_of_kernels_ms": round(tot / 1e6, 3),
                "mean_concurrency": round(tot / busy, 2), "kernels": len(ks),
                "largest_gaps_us": [(round(g / 1e3, 1), n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]) for g, n in gaps[:4]]})
for o in out:
    print(o)
