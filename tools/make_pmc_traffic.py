#!/usr/bin/env python3
"""profiles/<round>/fft_pmc_summary.csv -> profiles/pmc_traffic.json, the per-launch HBM bytes bench.py
reports as roofline.traffic (rocprofv3 PMC passes cannot run inside the timed bench process).
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
reads exactly half of the bytes of a wide (16 B/lane) coalesced stream -> doubled for mac_kernel and
ifft_kernel, whose bulk reads (block spectra; Y) are dwordx4 loads (ifft: 2 x 6.14 GB = 12.3 GB against the
11.6 GB of Y it must read plus 64 KiB of window energies per pair, mostly L2 hits)."""
import csv
import json
import sys

summary, out, workload = sys.argv[1], sys.argv[2], sys.argv[3]
rows = {r['kernel'].split('<')[0]: r for r in csv.DictReader(open(summary))}       # template arguments dropped
res = {}
for k, fetch_scale in (('ifft_kernel', 2.0), ('mac_kernel', 2.0), ('tspec_kernel', 1.0), ('refine_kernel', 1.0)):
    if k in rows and rows[k].get('FETCH_SIZE') and rows[k].get('WRITE_SIZE'):
        res[k] = {"fetch_bytes": float(rows[k]['FETCH_SIZE']) * 1024 * fetch_scale,
                  "write_bytes": float(rows[k]['WRITE_SIZE']) * 1024,
                  "fetch_scale_applied": fetch_scale, "dispatches_averaged": int(rows[k]['dispatches'])}
try:
    allw = json.load(open(out))
except Exception:
    allw = {}
allw[workload] = {"source": summary, "kernels": res}
json.dump(allw, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(allw[workload], indent=1))
