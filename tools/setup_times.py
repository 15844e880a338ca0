#!/usr/bin/env python3
"""Where a one-shot job's time before its first step goes (BASELINE configs[2], a FRESH process): host-to-device copies of the
two streams, HBM allocations, stream preparation kernels, the host-side plan of the batch (twice: sushi_hip_batch_bytes, then
sushi_hip_batch_create), the workspace allocation, descriptor uploads, the first run (code objects page in) and a steady
run -- for several workspace caps.  Needs the stream cache bench.py writes (SUSHI_BENCH_CACHE).
usage: setup_times.py [--config 2] [--ws-mb 0 8192 6144 4096 3072]"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t_import = time.perf_counter()
import bench  # noqa: E402
from sushi_amd import synth, _native  # noqa: E402
from sushi_amd.wav import WavStream  # noqa: E402
import torch  # noqa: E402
from sushi_amd.device import SearchBatch, DeviceStream, DEFAULT_DELTA  # noqa: E402
t_import = time.perf_counter() - t_import

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--ws-mb", type=int, nargs="*", default=[0, 8192, 6144, 4096, 3072])
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
cfg = bench.CONFIGS[args.config]
rate, seconds, n_total = cfg["rate"], cfg["minutes"] * 60.0, cfg["events"]
OFFSET = 7.25
seed = 20260924 + args.config
z = np.load(os.path.join(os.environ["SUSHI_BENCH_CACHE"], "c%d_%g_%d_float32_%g_0.npz" % (args.config, cfg["minutes"], rate, OFFSET)),
            allow_pickle=True)
dst = WavStream.from_prepared(z["dst"], rate, int(z["sample_count"]), int(z["padding_size"]))
src = WavStream.from_prepared(z["src"], rate, int(z["sample_count"]), int(z["padding_size"]))
events = synth.make_events(n_total, seconds, cfg["window"] + OFFSET, seed=seed + 2)
pats, centres, wins = synth.explicit_descriptors(src, dst, events, OFFSET, cfg["window"], seed=seed + 3)
offs = [src._get_sample_for_time(s) for s, _ in events]
lens = [p.shape[1] for p in pats]
wst, npos = [], []
for m, c, w in zip(lens, centres, wins):
    st, lo, p = dst._window(m, c, w)
    wst.append(lo); npos.append(p)
dev = torch.device("cuda", 0)
out = {"import_s": round(t_import, 2)}


def tick(label, t0, sync=True):
    if sync:
        torch.cuda.synchronize(dev)
    out[label] = round((time.perf_counter() - t0) * 1e3, 2)


t0 = time.perf_counter(); torch.cuda.init(); torch.empty(1, device=dev); tick("hip_context_ms", t0)
L = _native.lib()
t0 = time.perf_counter(); d_raw = torch.from_numpy(dst.data[0]).to(dev); s_raw = torch.from_numpy(src.data[0]).to(dev); tick("h2d_two_streams_pageable_ms", t0)
t0 = time.perf_counter(); ddev = DeviceStream(d_raw); sdev = DeviceStream(s_raw); tick("two_stream_creates_alloc_plus_kernels_ms", t0)
t0 = time.perf_counter(); ddev.searchable(); tick("spectra_alloc_plus_kernel_ms", t0)
# the same again with the allocator warm: what the kernels alone take
del ddev, sdev
t0 = time.perf_counter(); ddev = DeviceStream(d_raw); sdev = DeviceStream(s_raw); ddev.searchable(); tick("stream_prep_again_allocator_warm_ms", t0)
req = np.zeros(n_total, dtype=_native.REQUEST_DTYPE)
req["tmpl_off"], req["win_start"], req["tmpl_len"], req["n_pos"] = offs, wst, lens, npos
rows = []
for ws_mb in args.ws_mb:
    cap = (160 << 30) if ws_mb == 0 else ws_mb << 20
    r = {"ws_cap_mb": ws_mb}
    t0 = time.perf_counter()
    need = int(L.sushi_hip_batch_bytes(req.ctypes.data, n_total, _native.PATH_FFT, -1, cap))
    r["batch_bytes_host_plan_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    r["need_gb"] = round(need / 2 ** 30, 2)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    b = SearchBatch(ddev, sdev, offs, lens, wst, npos, path="fft", delta=DEFAULT_DELTA, workspace_bytes=cap)
    torch.cuda.synchronize(dev)
    r["searchbatch_total_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    r["sub_batches"] = b.sub_batches
    t0 = time.perf_counter(); b.run(); torch.cuda.synchronize(dev)
    r["first_run_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.run()
    torch.cuda.synchronize(dev)
    r["steady_run_ms"] = round((time.perf_counter() - t0) * 1e3 / args.steps, 3)
    idx = b.out_idx.cpu().numpy()
    r["idx_checksum"] = int(idx.astype(np.int64).sum())
    rows.append(r)
    del b
    torch.cuda.empty_cache()          # the next cap allocates afresh, as a one-shot job would
out["caps"] = rows
print(json.dumps(out))
