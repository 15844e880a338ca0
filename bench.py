#!/usr/bin/env python3
"""bench.py -- events/s of the batched template match on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over the whole job's batch of synthetic searches, with the streams (and the
destination stream's block spectra -- built once per stream, like the prefix sums) and the descriptors already
resident in HBM: sushi_hip_match_batch_fft (pattern DFTs, frequency-domain multiply-accumulate, inverse DFTs +
scoring, exact refinement, unpack; default) or sushi_hip_match_batch (--path direct: the exact-f32 MFMA kernel),
and for N > 1 the all-gather of (index, score).

Workload (--config, default 2 = the configuration BASELINE.json's north_star target is quoted on):
  1  BASELINE configs[1]: 1000 events, 45-min 12 kHz streams, +-60 s window  (P = 1,440,001 positions)
  2  BASELINE configs[2]: 3000 events, 2-h 12 kHz streams,  +-120 s window  (P = 2,880,001 positions)
  4  BASELINE configs[4]: 5000 events, 4-h 24 kHz streams,  +-120 s window  (P = 5,760,001 positions)
Patterns U[1,5] s, float32 streams (--sample-type uint8 for the reference's default type).  The job is the same at
every N: the time-sorted events are sharded in contiguous blocks over the N ranks (strong scaling), the two streams
are replicated, one all-gather of 8 bytes per event ends the step.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launch stream; `cpu_baseline`
(rank 0, N = 1 only) times the CPU oracle (an FFT port of cv2.matchTemplate; cv2 itself is not installable here)
on a bounded sample of the same searches, and `parity` compares the GPU results of that sample with it.
"""
import argparse
import hashlib
import json
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

CONFIGS = {
    1: {"events": 1000, "minutes": 45.0, "window": 60.0, "rate": 12000, "label": "BASELINE configs[1]"},
    2: {"events": 3000, "minutes": 120.0, "window": 120.0, "rate": 12000,
        "label": "BASELINE configs[2] (the north_star target's configuration)"},
    4: {"events": 5000, "minutes": 240.0, "window": 120.0, "rate": 24000, "label": "BASELINE configs[4]"},
}

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


def cpu_baseline(dst_row, src_row, offs, lens, wst, npos, budget_s=25.0, min_sample=64):
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
    sample = min(max(min(sample, 2 * cores), min_sample), n)
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


def kernel_source_digest():
    """sha256 over the kernel sources: profiles/pmc_traffic.json records the digest its PMC passes were taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sushi_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # ~40 ms per step at the default config: a 2 s timed region
    ap.add_argument("--warmup", type=int, default=3)   # the shader clock takes ~3 steps to ramp (tools/gpu_clock.sh)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2, help="BASELINE.json configs[] index")
    ap.add_argument("--events", type=int, default=None, help="events of the whole job (overrides --config)")
    ap.add_argument("--minutes", type=float, default=None)
    ap.add_argument("--window", type=float, default=None)
    ap.add_argument("--rate", type=int, default=None)
    ap.add_argument("--sample-type", default="float32")
    ap.add_argument("--offset", type=float, default=7.25, help="planted src->dst offset in seconds")
    ap.add_argument("--hard-frac", type=float, default=0.0,
                    help="fraction of the events cut from digital silence / a held tone / a repeated jingle "
                         "(tie-saturated searches); 0 = the BASELINE workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64, help="searches of the workload the oracle is run on, at least")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--path", choices=("fft", "direct"), default="fft")
    ap.add_argument("--ws-mb", type=int, default=None, help="FFT path scratch per batch (MiB); default: what one "
                                                            "sub-batch for the whole shard needs, at most 160 GiB")
    ap.add_argument("--delta", type=float, default=None)
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    custom = False
    for key in ("events", "minutes", "window", "rate"):
        v = getattr(args, key)
        if v is not None and v != cfg[key]:
            cfg[key] = v
            custom = True
    label = ("custom sizes (based on %s)" % cfg["label"]) if custom else cfg["label"]

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
    rate = cfg["rate"]
    seconds = cfg["minutes"] * 60.0
    seed = 20260924 + args.config
    # SUSHI_BENCH_CACHE=<dir>: keep the normalised streams of a workload between runs on one box (the profiling
    # scripts run this file several times; generating 2 x 86 M samples takes longer than the measurement)
    cache = os.environ.get("SUSHI_BENCH_CACHE")
    tag = "c%d_%g_%d_%s_%g_%g" % (args.config, cfg["minutes"], rate, args.sample_type, args.offset, args.hard_frac)
    cpath = os.path.join(cache, tag + ".npz") if cache else None
    hard_spans = []
    if cpath and os.path.exists(cpath):
        z = np.load(cpath, allow_pickle=True)
        dst = WavStream.from_prepared(z["dst"], rate, int(z["sample_count"]), int(z["padding_size"]))
        src = WavStream.from_prepared(z["src"], rate, int(z["sample_count"]), int(z["padding_size"]))
        hard_spans = [tuple(x) for x in z["hard_spans"].tolist()]
        hard_spans = [(k, float(a), float(b)) for k, a, b in hard_spans]
    else:
        if args.hard_frac > 0:
            dst_pcm, hard_spans = synth.make_hard_dst_pcm(seconds, rate, seed=seed)
        else:
            dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
        src_pcm = synth.make_src_pcm(dst_pcm, int(round(args.offset * rate)), seed=seed + 1)
        dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=args.sample_type)
        src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=args.sample_type)
        del dst_pcm, src_pcm
        if cpath and rank == 0:
            os.makedirs(cache, exist_ok=True)
            np.savez(cpath + ".tmp.npz", dst=dst.data, src=src.data, sample_count=dst.sample_count,
                     padding_size=dst.padding_size, hard_spans=np.array(hard_spans, dtype=object))
            os.replace(cpath + ".tmp.npz", cpath)
    n_total = cfg["events"]
    events = synth.make_events(n_total, seconds, cfg["window"] + abs(args.offset), seed=seed + 2)
    hard_mask = np.zeros(n_total, bool)
    if args.hard_frac > 0:
        events, hard_mask = synth.plant_hard_events(events, hard_spans, args.offset, args.hard_frac, seed=seed + 4)
    pats, centres, wins = synth.explicit_descriptors(src, dst, events, args.offset, cfg["window"], seed=seed + 3)
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
        cpu, cpu_results = cpu_baseline(dst.data[0], src.data[0], offs, lens, wst, npos, min_sample=args.cpu_sample)

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
                           workspace_bytes=(160 << 30) if args.ws_mb is None else args.ws_mb << 20)

    sharded = ShardedSearch(n_total, make_batch, device=dev)
    batch = sharded.batch

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    def step(ev_start=None, ev_end=None):
        if ev_start is not None:
            ev_start.record()             # the stream the kernels are launched on (torch's current stream)
        out = sharded.run_local()
        if ev_end is not None:
            ev_end.record()
        return sharded.gather(*out)

    # one untimed verification pass before anything is timed: a batch that does not recover the planted offset is
    # not worth measuring (it also pages the kernels in; the W warm-up steps below are the contract's)
    v_idx, _ = step()
    sync()
    ev_starts = np.array([s for s, _ in events])
    v_times = np.array(start_times) + v_idx.cpu().numpy().astype(np.float64) / float(rate)
    v_err = np.abs((v_times - ev_starts) - args.offset) * rate
    # An event whose result is more than one sample from the planted offset: with 20 dB of noise on a smooth signal the
    # true minimum can sit a sample beside the planted one (seen at 24 kHz), on top of the < 1 sample that the two
    # truncations of wav.py:173-175 contribute.  Beyond two samples the run is refused outright; between one and two the
    # CPU-baseline leg's oracle (the checker; skipped with --no-cpu-baseline) has to find the very same position.
    off_planted = [int(k) for k in np.nonzero((v_err > 1.0) & ~hard_mask)[0]]
    use_oracle = world == 1 and not args.no_cpu_baseline
    beyond_planted = {"events": len(off_planted), "confirmed_by_oracle": 0 if use_oracle else None}
    if off_planted and rank == 0:
        worst = float(v_err[~hard_mask].max())
        if len(off_planted) > 32 or worst > 2.0:
            raise SystemExit("verification pass: planted offset not recovered on %d events (max error %.3f samples)"
                             % (len(off_planted), worst))
        if use_oracle:
            _cpu_ctx.update(dst=dst.data[0], src=src.data[0], offs=offs, lens=lens, wst=wst, npos=npos)
            for k in off_planted:
                o_idx, o_score, _ = _cpu_one(k)
                if o_idx != int(v_idx[k]):
                    raise SystemExit("verification pass: event %d: position %d, oracle %d, planted offset missed by %.3f "
                                     "samples" % (k, int(v_idx[k]), o_idx, float(v_err[k])))
                beyond_planted["confirmed_by_oracle"] += 1
    for _ in range(args.warmup):
        step()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    sync()
    if args.path == "fft" and batch is not None:
        _native.profile_begin()           # per-stage HIP events on the launch stream, read after the final sync
    t0 = time.perf_counter()
    for k in range(args.steps):
        idx_all, score_all = step(starts[k], ends[k])
    sync()
    elapsed = time.perf_counter() - t0
    stage_ms = None
    if args.path == "fft" and batch is not None:
        stage_ms = _native.profile_end(args.steps).mean(axis=0)
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        local_elapsed, elapsed = elapsed, float(t.item())
        mine = {"rank": rank, "events": sharded.hi - sharded.lo, "kernels_ms_per_step": kernel_ms,
                "gather_and_wait_ms_per_step": local_elapsed / args.steps * 1e3 - kernel_ms}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    idx_all = idx_all.cpu().numpy()
    score_all = score_all.cpu().numpy()

    if rank == 0:
        # parity on the whole job: planted offset recovered to +-1 sample on every (ordinary) event
        times = np.array(start_times) + idx_all.astype(np.float64) / float(rate)
        shift_err = np.abs((times - ev_starts) - args.offset) * rate
        max_shift_err_vs_planted = float(shift_err[~hard_mask].max())
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
            kname = _native.STAGE_KERNELS[dom]
            achieved = batch.algorithmic_bytes / (dom_ms * 1e-3) / 1e9
            # HBM bytes of that kernel per launch from the committed rocprofv3 PMC passes of this very workload
            # (profiles/pmc_traffic.json, made by tools/make_pmc_traffic.py).  The entry records the digest of the
            # kernel sources it was measured on: a different digest means the kernels changed since -> null.
            traffic, traffic_bytes, traffic_note = None, None, "no PMC entry for this workload"
            wl_key = "config%d/fft/%s/%d/w%g/m%g/n%d" % (args.config, args.sample_type, n_total, cfg["window"],
                                                        cfg["minutes"], world)
            digest = kernel_source_digest()
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    entry = json.load(f)[wl_key]
                if entry.get("kernel_source_digest") != digest:
                    traffic_note = "stale: PMC passes were taken on kernel sources %s, this is %s" % (
                        entry.get("kernel_source_digest"), digest)
                else:
                    kern = entry["kernels"][kname]
                    traffic_bytes = kern["fetch_bytes"] + kern["write_bytes"]
                    traffic = traffic_bytes / (dom_ms * 1e-3) / 1e9
                    traffic_note = "%s @ %s" % (entry.get("source"), entry.get("source_commit"))
            except Exception:
                pass
            roofline = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBPS, "traffic": traffic,
                        "traffic_bytes_per_launch": traffic_bytes, "traffic_key": wl_key, "traffic_source": traffic_note,
                        "kernel_source_digest": digest,
                        "kernel": kname, "kernel_ms": dom_ms, "launches_per_step": batch.sub_batches,
                        "stage_ms": stages,
                        "step_kernels_ms": kernel_ms,
                        "step_hbm_achieved_GBps": batch.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9,
                        "step_frac": batch.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                        "algorithmic_bytes_per_launch": batch.algorithmic_bytes,
                        "direct_form_flop_per_launch": flops_launch,
                        "direct_form_equivalent_TFLOPs": flops_launch / (kernel_ms * 1e-3) / 1e12,
                        "fft_pairs": batch.fft_pairs, "fft_segments": batch.fft_segs,
                        "workspace_bytes": batch.ws_bytes, "delta": batch.delta,
                        "diagnostics": batch.diagnostics(),
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
            "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.sample_type == "float32" else "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %d events, %g-min %d Hz %s src/dst, +-%g s window (P=%d positions, "
                                   "patterns U[1,5] s), explicit centres; the whole job every step"
                                   % (label, n_total, cfg["minutes"], rate, args.sample_type, cfg["window"],
                                      int(np.median(npos))),
                       "baseline_config_index": args.config, "global_events": n_total,
                       "events_per_gpu": [shard[1] - shard[0] for shard in sharded.all_bounds()],
                       "window_s": cfg["window"], "stream_minutes": cfg["minutes"], "sample_rate": rate,
                       "sample_type": args.sample_type, "hard_events": int(hard_mask.sum()),
                       "method": "TM_SQDIFF_NORMED+argmin (what wav.py:185-186 does; see SURVEY F1)",
                       "path": ("overlap-save FFT (f32) + exact float64 re-evaluation of the near-minimum positions"
                                if args.path == "fft" else "direct exact-f32 MFMA sliding dot product"),
                       "parallelism": "events sharded in contiguous blocks over %d GPU(s), streams replicated, "
                                      "one all-gather of (idx, score)" % world,
                       "kernel_variant": batch.variant},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": {"max_shift_err_samples_vs_planted": max_shift_err_vs_planted,
                       "events_beyond_one_sample_of_planted": beyond_planted,
                       "oracle_sample_searches": len(cpu_results),
                       "max_idx_err_vs_oracle_sample": max_idx_err_vs_oracle,
                       "max_score_err_over_tolerance_vs_oracle_sample": max_rel_score_err,
                       "max_abs_score_err_vs_oracle_sample": max_abs_score_err,
                       "score_tolerance": "1e-4*score + 2.5e-7 (one float32 ulp of cv2's stored corr)"},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
