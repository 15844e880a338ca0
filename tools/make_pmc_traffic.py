#!/usr/bin/env python3
"""profiles/<round>/<summary>.csv -> profiles/pmc_traffic.json, the per-launch HBM bytes bench.py reports as
roofline.traffic (rocprofv3 PMC passes cannot run inside the timed bench process).
usage: make_pmc_traffic.py summary.csv pmc_traffic.json <workload key> <source commit>
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reads exactly
half of the bytes of a wide (16 B/lane) coalesced stream -> doubled for mac_kernel and ifft_kernel, whose bulk reads
(block spectra rows; Y) are dwordx4 loads.  The entry records the digest of the kernel sources it was measured on
(bench.kernel_source_digest): bench.py reports the traffic only while the digest still matches."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

summary, out, workload, commit = sys.argv[1:5]
# optional: a TCP summary (tools/summarize_pmc.py over a TCP_TCC_WRITE_REQ_sum pass) and "kernel=bytes" pairs, for kernels whose
# WRITE_SIZE pass is missing: write bytes from the L1s' 64-byte write requests, or from the kernel's own store count where that
# is exact -- recorded as such per kernel ("write_source")
tcp_summary = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
computed = dict(a.split("=") for a in sys.argv[6:])
rows = {}                                                  # template instances of one kernel (mac_kernel<0..5>: one dispatch
for r in csv.DictReader(open(summary)):                    # each per step) add up to that kernel's bytes per step
    k = r['kernel'].split('<')[0]
    if k in rows:
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            if r.get(c) and rows[k].get(c):
                rows[k][c] = str(float(rows[k][c]) + float(r[c]))
        rows[k]['dispatches'] = str(max(int(rows[k]['dispatches']), int(r['dispatches'])))
    else:
        rows[k] = dict(r)
tcp = {}
if tcp_summary:
    for r in csv.DictReader(open(tcp_summary)):
        if r.get('TCP_TCC_WRITE_REQ_sum'):
            tcp[r['kernel'].split('<')[0]] = float(r['TCP_TCC_WRITE_REQ_sum']) * 64.0
res = {}
for k, fetch_scale in (('ifft_kernel', 2.0), ('mac_kernel', 2.0), ('mac_long_kernel', 2.0), ('bound_kernel', 2.0), ('tspec_kernel', 1.0),
                       ('refine_kernel', 1.0), ('collect_kernel', 2.0), ('exact_tiles_kernel', 1.0), ('slb_kernel', 1.0),
                       ('pilot_kernel', 1.0), ('survivor_kernel', 1.0)):
    if k not in rows or not rows[k].get('FETCH_SIZE'):
        continue
    if rows[k].get('WRITE_SIZE'):
        wb, ws = float(rows[k]['WRITE_SIZE']) * 1024, "WRITE_SIZE"
    elif k in tcp:
        wb, ws = tcp[k], "TCP_TCC_WRITE_REQ_sum x 64 B (no WRITE_SIZE pass)"
    elif k in computed:
        wb, ws = float(computed[k]), "the kernel's own stores, counted (no WRITE_SIZE pass)"
    else:
        continue
    res[k] = {"fetch_bytes": float(rows[k]['FETCH_SIZE']) * 1024 * fetch_scale, "write_bytes": wb, "write_source": ws,
              "fetch_scale_applied": fetch_scale, "dispatches_averaged": int(rows[k]['dispatches'])}
try:
    allw = json.load(open(out))
except Exception:
    allw = {}
allw[workload] = {"source": summary, "source_commit": commit, "kernel_source_digest": bench.kernel_source_digest(),
                  "kernels": res}
json.dump(allw, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(allw[workload], indent=1))
