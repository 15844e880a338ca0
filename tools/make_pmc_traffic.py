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
rows = {}                                                  # template instances of one kernel (mac_kernel<1024>, <4096>) add up to
runs_env = int(os.environ.get("PMC_RUNS", "0"))            # that kernel's bytes per step: each instance's average x its dispatches per step
for r in csv.DictReader(open(summary)):
    k = r['kernel'].split('<')[0]
    w = (int(r['dispatches']) / runs_env) if runs_env else 1.0
    cur = rows.setdefault(k, {'kernel': k, 'dispatches': 0, 'FETCH_SIZE': None, 'WRITE_SIZE': None})
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        if r.get(c):
            cur[c] = (cur[c] or 0.0) + float(r[c]) * w
    cur['dispatches'] = cur['dispatches'] + int(r['dispatches'])
tcp = {}
if tcp_summary:
    for r in csv.DictReader(open(tcp_summary)):
        if r.get('TCP_TCC_WRITE_REQ_sum'):
            tcp[r['kernel'].split('<')[0]] = float(r['TCP_TCC_WRITE_REQ_sum']) * 64.0
res = {}
# PMC_RUNS=<batch runs of the profiled process>: bytes PER STEP = (average per dispatch) x dispatches / runs -- several kernels are
# launched more than once per step (ifft_kernel and mac_list_kernel for the pairs looked at first and for the ones the bound
# left), with very different sizes; without it the average dispatch stands for the step (round 4's entries)
runs = int(os.environ.get("PMC_RUNS", "0"))
for k, fetch_scale in (('ifft_kernel', 2.0), ('ifft_list_kernel', 2.0), ('mac_kernel', 2.0), ('mac_long_kernel', 2.0), ('bound_kernel', 2.0),
                       ('bound_low_kernel', 2.0), ('bound_low_exact_kernel', 1.0), ('mac_list_kernel', 2.0), ('mac_rows_kernel', 2.0), ('tspec_kernel', 1.0),
                       ('slb_list_kernel', 1.0), ('survivor2_kernel', 1.0),
                       ('refine_kernel', 1.0), ('collect_kernel', 2.0), ('exact_tiles_kernel', 1.0), ('slb_kernel', 1.0),
                       ('pilot_kernel', 1.0), ('survivor_kernel', 1.0)):
    if k not in rows or rows[k].get('FETCH_SIZE') is None:
        continue
    if rows[k].get('WRITE_SIZE') is not None:
        wb, ws = float(rows[k]['WRITE_SIZE']) * 1024, "WRITE_SIZE"
    elif k in tcp:
        wb, ws = tcp[k], "TCP_TCC_WRITE_REQ_sum x 64 B (no WRITE_SIZE pass)"
    elif k in computed:
        wb, ws = float(computed[k]), "the kernel's own stores, counted (no WRITE_SIZE pass)"
    else:
        continue
    per_step = (int(rows[k]['dispatches']) / runs) if runs else 1.0
    res[k] = {"fetch_bytes": float(rows[k]['FETCH_SIZE']) * 1024 * fetch_scale, "write_bytes": wb, "write_source": ws,
              "fetch_scale_applied": fetch_scale, "dispatches_averaged": int(rows[k]['dispatches']),
              "dispatches_per_step": per_step}
try:
    allw = json.load(open(out))
except Exception:
    allw = {}
allw[workload] = {"source": summary, "source_commit": commit, "kernel_source_digest": bench.kernel_source_digest(),
                  "kernels": res}
json.dump(allw, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(allw[workload], indent=1))
