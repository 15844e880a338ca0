#!/usr/bin/env python3
"""bench.py -- events/s of the batched template match on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic searches whose streams (and, for
the FFT path, the destination stream's block spectra -- built once per stream, like the prefix
sums) and descriptors are already resident in HBM: sushi_hip_match_batch_fft (template DFTs,
frequency-domain multiply-accumulate, inverse DFTs + scoring, exact refinement, unpack; default)
or sushi_hip_match_batch (--path direct: the exact-f32 MFMA kernel), and for N > 1 the all-gather
of (index, score).  Workload at every N: BASELINE.json configs[1] per GPU
(1000 events, 45-min 12 kHz float32 src/dst, +-60 s window => P = 1,440,001 positions, templates
U[1,5] s) -- weak scaling, streams replicated, events sharded in contiguous blocks.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launch stream;
`cpu_baseline` (rank 0, N = 1 only) times the CPU oracle (an FFT port of cv2.matchTemplate; cv2
itself is not installable here) on a bounded sample of the same searches.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "subtitle events/sec (matchTemplate+argmax) + max |shift err| vs cv2, 1/2/4/8 GPU"
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X dense f32 (vector == matrix) peak, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0

_cpu_ctx = {}


def _cpu_one(k):
    """One search on the CPU oracle (FFT port).  Returns (idx, score, seconds)."""
    from oracle import oracle as O
    c = _cpu_ctx
    t0 = time.perf_counter()
    off, m, ws, p = c["offs"][k], c["lens"][k], c["wst"][k], c["npos"][k]
    res = O.match_template_fft(c["dst"][ws:ws + p + m - 1], c["src"][off:off + m])[0]
    idx = int(res.argmin())
    return idx, float(res[idx]), time.perf_counter() - t0


def cpu_baseline(dst_row, src_row, offs, lens, wst, npos, budget_s=25.0):
    """Time the oracle on a bounded sample of the workload, before CUDA is initialised (fork)."""
    import multiprocessing as mp
    _cpu_ctx.update(dst=dst_row, src=src_row, offs=offs, lens=lens, wst=wst, npos=npos)
    n = len(offs)
    t0 = time.perf_counter()
    first = _cpu_one(0)                                   # also the 1-core figure
    one_core = 1.0 / max(first[2], 1e-9)
    cores = max(1, os.cpu_count() or 1)
    per_search = first[2]
    sample = int(max(2, min(n, cores * max(1, int(budget_s / max(per_search, 1e-3)) - 1))))
    sample = min(sample, 4 * cores, n)
    ks = list(np.linspace(0, n - 1, sample).astype(int))
    results = {}
    used = 1
    wall = None
    if cores > 1 and sample > 1:
        try:
            ctx = mp.get_context("fork")
            used = min(cores, sample)
            with ctx.Pool(used) as pool:
                t1 = time.perf_counter()
                out = pool.map(_cpu_one, ks, chunksize=1)
                wall = time.perf_counter() - t1
            results = {k: o for k, o in zip(ks, out)}
        except Exception:
            results, used, wall = {}, 1, None
    if wall is None:
        ks = ks[:max(2, min(len(ks), int(budget_s / max(per_search, 1e-3))))]
        t1 = time.perf_counter()
        results = {k: _cpu_one(k) for k in ks}
        wall = time.perf_counter() - t1
        used = 1
    return {"value": len(results) / wall, "unit": "events/s", "cores": used, "kind": "port",
            "sample": "%d of the %d searches of this workload (evenly spaced), NumPy/SciPy float64 overlap-add FFT "
                      "restatement of cv2.matchTemplate(TM_SQDIFF_NORMED)+argmin, one process per core"
                      % (len(results), n),
            "value_1core": one_core, "seconds": time.perf_counter() - t0}, results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)   # the shader clock takes ~3 steps to ramp (tools/gpu_clock.sh)
    ap.add_argument("--events", type=int, default=1000, help="events per GPU (BASELINE configs[1]: 1000)")
    ap.add_argument("--minutes", type=float, default=45.0)
    ap.add_argument("--window", type=float, default=60.0)
    ap.add_argument("--rate", type=int, default=12000)
    ap.add_argument("--sample-type", default="float32")
    ap.add_argument("--offset", type=float, default=7.25, help="planted src->dst offset in seconds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--path", choices=("fft", "direct"), default="fft")
    ap.add_argument("--ws-mb", type=int, default=None, help="FFT path scratch per batch (MiB)")
    ap.add_argument("--delta", type=float, default=None)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from sushi_amd import synth
    from sushi_amd.wav import WavStream

    # ---- synthetic inputs (identical on every rank: streams are replicated) ---------------------
    # The streams are built with the NumPy load pipeline: the CPU baseline below forks worker processes,
    # which must happen before this process initialises HIP (the GPU load pipeline would do that).
    os.environ["SUSHI_HIP_LOAD"] = "host"
    seconds = args.minutes * 60.0
    seed = 20260924 + 1
    dst_pcm = synth.make_dst_pcm(seconds, args.rate, seed=seed)
    src_pcm = synth.make_src_pcm(dst_pcm, int(round(args.offset * args.rate)), seed=seed + 1)
    dst = WavStream.from_samples(dst_pcm, args.rate, sample_rate=args.rate, sample_type=args.sample_type)
    src = WavStream.from_samples(src_pcm, args.rate, sample_rate=args.rate, sample_type=args.sample_type)
    del dst_pcm, src_pcm
    n_total = args.events * world
    events = synth.make_events(n_total, seconds, args.window + abs(args.offset), seed=seed + 2)
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, args.offset, args.window, seed=seed + 3)
    pad = src.padding_size
    offs = [src._get_sample_for_time(s) for s, _ in events]
    lens = [p.shape[1] for p in pats]
    start_times, wst, npos = [], [], []
    for m, c, w in zip(lens, centres, wins):
        st, lo, p = dst._window(m, c, w)
        start_times.append(st); wst.append(lo); npos.append(p)
    del pats

    # ---- CPU baseline first (rank 0, N = 1): fork a pool before CUDA exists in this process ----
    cpu = None
    cpu_results = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cpu_results = cpu_baseline(dst.data[0], src.data[0], offs, lens, wst, npos)

    import torch
    import torch.distributed as dist
    from sushi_amd.device import SearchBatch
    from sushi_amd.distributed import ShardedSearch

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    dst._device = src._device = dev
    ddev, sdev = dst.device_stream(), src.device_stream()

    from sushi_amd import _native
    from sushi_amd.device import DEFAULT_DELTA

    def make_batch(lo, hi):
        return SearchBatch(ddev, sdev, offs[lo:hi], lens[lo:hi], wst[lo:hi], npos[lo:hi], variant=args.variant,
                           path=args.path, delta=DEFAULT_DELTA if args.delta is None else args.delta,
                           workspace_bytes=None if args.ws_mb is None else args.ws_mb << 20)

    sharded = ShardedSearch(n_total, make_batch)
    batch = sharded.batch

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    from sushi_amd.distributed import gather_results

    def step(ev_start=None, ev_end=None):
        if ev_start is not None:
            ev_start.record()             # the stream the kernels are launched on (torch's current stream)
        idx, score = batch.run()
        if ev_end is not None:
            ev_end.record()
        if world > 1:
            idx, score = gather_results(idx, score, n_total)
        return idx, score

    # one untimed verification pass before anything is timed: a batch that does not recover the planted offset is
    # not worth measuring (it also pages the kernels in; the W warm-up steps below are the contract's)
    v_idx, _ = step()
    sync()
    v_times = np.array(start_times) + v_idx.cpu().numpy().astype(np.float64) / float(args.rate)
    v_err = np.abs((v_times - np.array([s for s, _ in events])) - args.offset) * args.rate
    if float(v_err.max()) > 1.0:
        raise SystemExit("verification pass: planted offset not recovered (max error %.3f samples)" % float(v_err.max()))
    for _ in range(args.warmup):
        step()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    sync()
    if args.path == "fft":
        _native.profile_begin()           # per-stage HIP events on the launch stream, read after the final sync
    t0 = time.perf_counter()
    for k in range(args.steps):
        idx_all, score_all = step(starts[k], ends[k])
    sync()
    elapsed = time.perf_counter() - t0
    stage_ms = _native.profile_end(args.steps).mean(axis=0) if args.path == "fft" else None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))
    idx_all = idx_all.cpu().numpy()
    score_all = score_all.cpu().numpy()

    if rank == 0:
        # parity on the whole job: planted offset recovered to +-1 sample on every event
        times = np.array(start_times) + idx_all.astype(np.float64) / float(args.rate)
        shift_err = np.abs((times - np.array([s for s, _ in events])) - args.offset) * args.rate
        max_shift_err_vs_planted = float(shift_err.max())
        max_idx_err_vs_oracle = None
        max_rel_score_err = None
        max_abs_score_err = None
        if cpu_results:
            ie = [abs(int(idx_all[k]) - r[0]) for k, r in cpu_results.items()]
            # excess over the parity bound |d| <= 1e-4*score + 2.5e-7 (tests/test_gpu_parity.py); <= 1 passes
            se = [abs(float(score_all[k]) - r[1]) / (1e-4 * r[1] + 2.5e-7) for k, r in cpu_results.items()]
            ae = [abs(float(score_all[k]) - r[1]) for k, r in cpu_results.items()]
            max_idx_err_vs_oracle, max_rel_score_err, max_abs_score_err = int(max(ie)), float(max(se)), float(max(ae))
        value = n_total * args.steps / elapsed
        flops_launch = batch.flops
        if args.path == "fft":
            # dominant kernel of the step = the stage with the largest HIP-event time; its duration is
            # the sum over the step's sub-batch launches of that kernel
            stages = {n: float(v) for n, v in zip(_native.STAGE_NAMES, stage_ms)}
            dom = max(stages, key=stages.get)
            dom_ms = stages[dom]
            kname = {"tspec": "tspec_kernel", "mac": "mac_kernel", "ifft": "ifft_kernel",
                     "refine": "refine_kernel", "finish": "match_flagged_kernel+unpack_keys_kernel"}[dom]
            achieved = batch.algorithmic_bytes / (dom_ms * 1e-3) / 1e9
            # HBM bytes of that kernel per launch from the committed rocprofv3 PMC passes of this very
            # workload (profiles/pmc_traffic.json, made by tools/make_pmc_traffic.py); None if absent
            traffic, traffic_bytes = None, None
            wl_key = "configs1/fft/%s/%d/w%g/m%g" % (args.sample_type, args.events, args.window, args.minutes)
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    kern = json.load(f)[wl_key]["kernels"][kname]
                traffic_bytes = kern["fetch_bytes"] + kern["write_bytes"]
                traffic = traffic_bytes / (dom_ms * 1e-3) / 1e9
            except Exception:
                pass
            roofline = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBPS, "traffic": traffic,
                        "traffic_bytes_per_launch": traffic_bytes, "traffic_key": wl_key,
                        "kernel": kname, "kernel_ms": dom_ms, "stage_ms": stages,
                        "step_kernels_ms": kernel_ms,
                        "step_hbm_achieved_GBps": batch.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9,
                        "algorithmic_bytes_per_launch": batch.algorithmic_bytes,
                        "direct_form_flop_per_launch": flops_launch,
                        "direct_form_equivalent_TFLOPs": flops_launch / (kernel_ms * 1e-3) / 1e12,
                        "fft_pairs": batch.fft_pairs, "fft_segments": batch.fft_segs,
                        "workspace_bytes": batch.ws_bytes, "delta": batch.delta,
                        "searches_finished_by_direct_kernel": batch.fallback_count(),
                        # what a bare streaming kernel reaches on this part (tools/ubench/hbm_bw.hip,
                        # profiles/r01/hbm_bw.jsonl): the practical ceiling under the 8 TB/s peak
                        "stream_ceiling_GBps": {"read": 6300.0, "write": 5300.0, "copy": 5500.0}}
        else:
            achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
            hbm_achieved = batch.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9
            roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                        "kernel": "match_sqdiff_f32_kernel", "kernel_ms": kernel_ms,
                        "algorithmic_flop_per_launch": flops_launch,
                        "algorithmic_bytes_per_launch": batch.algorithmic_bytes,
                        "hbm_achieved_GBps": hbm_achieved, "hbm_frac": hbm_achieved / PEAK_HBM_GBPS}
        out = {
            "metric": METRIC, "value": value, "unit": "events/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.sample_type == "float32" else "u8",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d-event batch per GPU, %g-min %d Hz %s src/dst, +-%g s window "
                                   "(P=%d positions, template U[1,5] s), explicit centres"
                                   % (args.events, args.minutes, args.rate, args.sample_type, args.window,
                                      int(np.median(npos))),
                       "events_per_gpu": args.events, "global_events": n_total, "window_s": args.window,
                       "stream_minutes": args.minutes, "sample_rate": args.rate, "sample_type": args.sample_type,
                       "method": "TM_SQDIFF_NORMED+argmin (what wav.py:185-186 does; see SURVEY F1)",
                       "path": ("overlap-save FFT (f32) + exact float64 re-evaluation of the near-minimum positions"
                                if args.path == "fft" else "direct exact-f32 MFMA sliding dot product"),
                       "parallelism": "events sharded in contiguous blocks over %d GPU(s), streams replicated, "
                                      "one all-gather of (idx, score)" % world,
                       "kernel_variant": batch.variant},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": {"max_shift_err_samples_vs_planted": max_shift_err_vs_planted,
                       "max_idx_err_vs_oracle_sample": max_idx_err_vs_oracle,
                       "max_score_err_over_tolerance_vs_oracle_sample": max_rel_score_err,
                       "max_abs_score_err_vs_oracle_sample": max_abs_score_err,
                       "score_tolerance": "1e-4*score + 2.5e-7 (one float32 ulp of cv2's stored corr)"},
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
