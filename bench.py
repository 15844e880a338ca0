#!/usr/bin/env python3
"""bench.py -- events/s of the batched template match on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over the whole job's batch of synthetic searches, with the streams (and the
destination stream's block spectra -- built once per stream, like the prefix sums) and the descriptors already
resident in HBM: sushi_hip_batch_run on the FFT path (pattern DFTs, frequency-domain multiply-accumulate, inverse
DFTs + scoring, exact refinement, unpack; default) or on the direct path (--path direct: the exact-f32 MFMA kernel),
and for N > 1 the all-gather of (index, score).

Workload (--config, default 2 = the configuration BASELINE.json's north_star target is quoted on):
  1  BASELINE configs[1]: 1000 events, 45-min 12 kHz streams, +-60 s window  (P = 1,440,001 positions)
  2  BASELINE configs[2]: 3000 events, 2-h 12 kHz streams,  +-120 s window  (P = 2,880,001 positions)
  4  BASELINE configs[4]: 5000 events, 4-h 24 kHz streams,  +-120 s window  (P = 5,760,001 positions)
Patterns U[1,5] s, float32 streams (--sample-type uint8 for the reference's default type).  --method sqdiff_normed
(what wav.py:185-186 computes; default) or ccoeff_normed (what BASELINE.json's wording names).  The job is the same at
every N: the time-sorted events are sharded in contiguous blocks over the N ranks (strong scaling), the two streams
are replicated, one all-gather of 8 bytes per event ends the step.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launch stream.  The oracle leg (rank 0,
before HIP is initialised in this process: it forks one single-threaded worker per host core) always runs: it is the
line's `parity` block -- an evenly spaced sample of the job's searches plus every search cut from tie-saturated
material, compared with the GPU results -- and, unless --no-cpu-baseline, over a larger sample its wall time is the
`cpu_baseline` (the oracle is an FFT port of cv2.matchTemplate; cv2 itself is not installable here -- where it does
import, `parity.cv2` holds the same comparison against the real call).
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
METHOD_TEXT = {
    "sqdiff_normed": "TM_SQDIFF_NORMED+argmin (what wav.py:185-186 does; see SURVEY F1)",
    "ccoeff_normed": "TM_CCOEFF_NORMED+argmax (the method BASELINE.json's north_star names; the reference calls "
                     "TM_SQDIFF_NORMED)",
}
SCORE_RTOL, SCORE_ATOL = 1e-4, 2.5e-7          # tests/test_gpu_parity.py

_cpu_ctx = {}


def _cpu_init():
    """Worker start: one thread per process (256 processes x an OpenMP team each is what made round 2's leg crawl), and
    a heap that keeps what it frees: every search allocates ~150 MB of NumPy temporaries, and 256 processes faulting
    fresh pages in at once spend their time in the kernel's page allocator instead of the FFT (measured on the 256-core
    box: 13.5 s per search against 0.076 s alone).  glibc serves blocks up to 32 MiB from the heap when told to."""
    from oracle import oracle as O
    O.set_num_threads(1)
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 32 << 20)         # M_MMAP_THRESHOLD: its maximum
        libc.mallopt(-1, 1 << 30)          # M_TRIM_THRESHOLD: do not give the heap back between searches
        libc.mallopt(-2, 64 << 20)         # M_TOP_PAD
    except Exception:
        pass
    try:
        from threadpoolctl import threadpool_limits
        _cpu_ctx["_tp"] = threadpool_limits(limits=1)
    except Exception:
        pass


def _cpu_one(k):
    """One search on the CPU oracle (FFT port).  Returns (idx, score, seconds, cv2 result or None)."""
    from oracle import oracle as O
    c = _cpu_ctx
    t0 = time.perf_counter()
    off, m, ws, p = c["offs"][k], c["lens"][k], c["wst"][k], c["npos"][k]
    method = c.get("method", "sqdiff_normed")
    res = O.match_template_fft(c["dst"][ws:ws + p + m - 1], c["src"][off:off + m], method=method)[0]
    idx = int(res.argmin()) if method == "sqdiff_normed" else O.argmax_first(res)
    dt = time.perf_counter() - t0
    real = None
    if k in c.get("cv2_set", ()):
        r2 = O.match_template_cv2(c["dst"][ws:ws + p + m - 1], c["src"][off:off + m], method)[0]
        i2 = int(r2.argmin()) if method == "sqdiff_normed" else int(r2.argmax())
        real = (i2, float(r2[i2]), float(r2[idx]))
    return idx, float(res[idx]), dt, real


def usable_cores():
    """Host cores this process can actually use: the affinity mask and the cgroup CPU quota, not os.cpu_count() (the GPU
    boxes report 256 logical CPUs and grant a container about three of them: 256 workers then time-share those)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                    # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n), quota


def _spread(n):
    """0 .. n-1 in an order whose every prefix is spread evenly over the range (bit reversal)."""
    bits = max(1, int(n - 1).bit_length())
    order = [int(format(i, "0%db" % bits)[::-1], 2) for i in range(1 << bits)]
    return [i for i in order if i < n]


def oracle_leg(dst_row, src_row, offs, lens, wst, npos, method, forced, timed, min_sample=64, budget_s=25.0, workers=None):
    """The CPU oracle over a sample of the workload, before CUDA is initialised (fork): the searches in `forced`, an
    evenly spread sample of at least `min_sample`, and -- `timed` -- as many more (evenly spread, at least 8 per worker)
    as `budget_s` of wall time allows; their throughput is the cpu_baseline.  Returns (cpu_baseline | None, results)."""
    import multiprocessing as mp
    from oracle import oracle as O
    have_cv2 = O.cv2_module() is not None
    _cpu_ctx.update(dst=dst_row, src=src_row, offs=offs, lens=lens, wst=wst, npos=npos, method=method)
    n = len(offs)
    t0 = time.perf_counter()
    cold = _cpu_one(0)                  # first call: imports, fresh pages
    first = _cpu_one(0)
    per_search = max(first[2], 1e-4)
    cores, quota = usable_cores()
    used = max(1, min(cores if workers is None else workers, n))
    forced = [int(k) for k in forced]
    _forced_set = set(forced)
    rest = [k for k in _spread(n) if k not in _forced_set]
    must = forced + rest[:max(0, min_sample - len(forced))]         # the parity sample: always
    more = rest[max(0, min_sample - len(forced)):] if timed else []
    if have_cv2:                        # the real call on an evenly spread part of the sample (bounded: it is slow too)
        _cpu_ctx["cv2_set"] = set(must[:64])
    results, wall, timed_n = {}, 0.0, 0
    try:
        if used < 2:
            raise RuntimeError("single core")
        ctx = mp.get_context("fork")
        with ctx.Pool(used, initializer=_cpu_init) as pool:
            pool.map(_noop, range(used), chunksize=1)             # workers up before the clock starts
            # rounds of 8 searches per worker until the budget is spent: the first round(s) hold the parity sample
            todo = must + more
            pos = 0
            while pos < len(todo):
                batch = todo[pos:pos + 8 * used]
                if pos >= len(must) and (wall > budget_s or not timed):
                    break
                t1 = time.perf_counter()
                out = pool.map(_cpu_one, batch, chunksize=max(1, len(batch) // (used * 2)))
                dt = time.perf_counter() - t1
                if len(batch) >= used:                            # rounds that fill every worker count for the rate
                    wall += dt
                    timed_n += len(batch)
                results.update(zip(batch, out))
                pos += len(batch)
    except Exception:
        used = 1
        todo = [k for k in must if k not in results]
        t1 = time.perf_counter()
        for k in todo:
            results[k] = _cpu_one(k)
        if timed:
            for k in more:
                if time.perf_counter() - t1 > budget_s:
                    break
                results[k] = _cpu_one(k)
        wall, timed_n = time.perf_counter() - t1, len(results)
    cpu = None
    if timed and timed_n and wall > 0:
        value = timed_n / wall
        one_core = 1.0 / per_search
        busy = sum(r[2] for r in results.values())
        cpu = {"value": value, "unit": "events/s", "cores": used, "logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
               "kind": "port",
               "sample": "%d of the %d searches of this workload (evenly spread%s), NumPy/SciPy float64 overlap-add FFT "
                         "restatement of cv2.matchTemplate(%s), one single-threaded process per core in rounds of 8 "
                         "searches each, %.0f s budget" % (len(results), n, " + every tie-saturated one" if forced else "",
                                                          METHOD_TEXT[method].split(" ")[0], budget_s),
               "value_1core": one_core, "parallel_efficiency": value / (used * one_core),
               # one search alone: the first call (fresh pages) and a repeat (warm heap); the same under load, as the
               # workers timed themselves.  The gap between `alone` and `loaded` is what the processes cost each other
               # (page faults of the NumPy temporaries, memory bandwidth), not arithmetic.
               "per_search_s_alone_cold": cold[2], "per_search_s_alone": per_search,
               "per_search_s_loaded": busy / max(1, len(results)),
               "seconds": time.perf_counter() - t0}
    return cpu, results


def _noop(_):
    return 0


def kernel_source_digest():
    """sha256 over the kernel sources: profiles/pmc_traffic.json records the digest its PMC passes were taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sushi_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")) or (name.endswith(".inc") and not name.startswith("_gen_")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def loaded_library():
    """Path and sha256 of the libsushi_hip.so this process mapped."""
    from sushi_amd import _native
    h = hashlib.sha256()
    try:
        with open(_native.LIB_PATH, "rb") as f:
            for piece in iter(lambda: f.read(1 << 20), b""):
                h.update(piece)
        sha = h.hexdigest()[:16]
    except OSError:
        sha = None
    return {"path": os.path.relpath(_native.LIB_PATH, ROOT), "sha256_16": sha, "env_override": bool(os.environ.get("SUSHI_HIP_LIB"))}


class DryBatch(object):
    """--dry-backend gloo: a stand-in for SearchBatch that answers the planted positions from the host and computes
    nothing -- so that the N > 1 control flow of this file (sharding, gather, per-rank report, max over ranks) can run
    under torch.distributed.run on a machine without GPUs (tests/test_bench_multirank_dry.py).  Never measured."""

    def __init__(self, planted_idx, lo, hi):
        import torch
        self._idx = torch.tensor(np.asarray(planted_idx[lo:hi], np.int32))
        self._score = torch.zeros(hi - lo, dtype=torch.float32)
        self.flops = self.algorithmic_bytes = 1.0
        self.sub_batches, self.variant, self.fft_pairs, self.fft_segs, self.ws_bytes, self.delta = 1, 0, 0, 0, 0, 0.0
        self.lanes = 1

    def run(self):
        return self._idx, self._score

    def diagnostics(self, per_search=False):
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # ~25 ms per step at the default config: a 1.3 s timed region
    ap.add_argument("--warmup", type=int, default=3)   # the shader clock takes ~3 steps to ramp (tools/gpu_clock.sh)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2, help="BASELINE.json configs[] index")
    ap.add_argument("--events", type=int, default=None, help="events of the whole job (overrides --config)")
    ap.add_argument("--minutes", type=float, default=None)
    ap.add_argument("--window", type=float, default=None)
    ap.add_argument("--rate", type=int, default=None)
    ap.add_argument("--sample-type", default="float32")
    ap.add_argument("--method", choices=sorted(METHOD_TEXT), default="sqdiff_normed")
    ap.add_argument("--offset", type=float, default=7.25, help="planted src->dst offset in seconds")
    ap.add_argument("--snr", type=float, default=20.0,
                    help="signal-to-noise ratio (dB) of the source stream = planted copy of the destination + white noise; "
                         "20 = the BASELINE workload (SURVEY 8d)")
    ap.add_argument("--source", choices=("noise", "encode", "dub", "partial"), default="noise",
                    help="what the source stream is beside the planted offset: `noise` = the destination + white noise at --snr (the "
                         "BASELINE workload); `encode` = another encode of it (gain 0.7, 4 kHz low-pass, requantised to 8 bits: "
                         "synth.make_src_pcm_other_encode); `dub` = the same music bed under each stream's OWN speech on half of the "
                         "time (synth.make_dub_pcm); `partial` = the source is the destination (+ noise at --snr) for the first half "
                         "of the programme and OTHER audio for the second half (a different cut: half of the events find nothing) -- "
                         "what Sushi's real inputs look like")
    ap.add_argument("--unrelated", action="store_true",
                    help="the source stream is INDEPENDENT audio of the same kind (no match anywhere: nothing the pair "
                         "exclusion can use) -- the worst case of a data-dependent step; parity is then the oracle sample only")
    ap.add_argument("--exclusion", choices=("auto", "always", "never", "band", "whole"), default=None,
                    help="FFT path: pair exclusion mode of the batch (default: the library's AUTO; band / whole force one form)")
    ap.add_argument("--hard-frac", type=float, default=0.0,
                    help="fraction of the events cut from digital silence / a held tone / a repeated jingle "
                         "(tie-saturated searches); 0 = the BASELINE workload")
    ap.add_argument("--profile-only", action="store_true",
                    help="no oracle leg at all (no forked workers: rocprofv3's counter passes hang waiting for them); the line's "
                         "parity block then holds the planted-offset check only -- for profiler runs, never for a reported line")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the TIMING of the oracle (cpu_baseline: null); the parity sample is still run")
    ap.add_argument("--cpu-sample", type=int, default=64, help="searches of the workload the oracle is run on, at least")
    ap.add_argument("--cpu-workers", type=int, default=None, help="processes of the oracle leg (default: one per host core)")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--path", choices=("fft", "direct"), default="fft")
    ap.add_argument("--ws-mb", type=int, default=None, help="FFT path scratch per batch (MiB); default: what one "
                                                            "sub-batch for the whole shard needs, at most 160 GiB")
    ap.add_argument("--delta", type=float, default=None)
    ap.add_argument("--emulate-shards", type=int, default=8,
                    help="N = 1 only: after the measurement, cut the job into the G = 2, 4, .. work-balanced shards the G-rank run "
                         "would give its ranks (up to this G; 0 = off) and time every shard on THIS GPU -- the slowest shard is "
                         "what a G-GPU step would take, the collective aside (`shard_emulation` in the line)")
    ap.add_argument("--dry-backend", choices=("gloo",), default=None,
                    help="no GPU: run this file's N-rank control flow over gloo with a stand-in batch (tests only)")
    ap.add_argument("--dry-plant-error", type=int, default=0,
                    help="dry runs: the stand-in answers this many samples off the planted position (tests)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    custom = False
    for key in ("events", "minutes", "window", "rate"):
        v = getattr(args, key)
        if v is not None and v != cfg[key]:
            cfg[key] = v
            custom = True
    label = ("custom sizes (based on %s)" % cfg["label"]) if custom else cfg["label"]
    dry = args.dry_backend is not None

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: become the launcher (one process per GPU over RCCL, rendezvous on 127.0.0.1)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.stdout.flush()
            os.execv(sys.executable, cmd)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from sushi_amd import synth
    from sushi_amd.wav import WavStream

    # ---- synthetic inputs (identical on every rank: streams are replicated) ---------------------
    # The streams are built with the NumPy load pipeline: the oracle leg below forks worker processes,
    # which must happen before this process initialises HIP (the GPU load pipeline would do that).
    os.environ["SUSHI_HIP_LOAD"] = "host"
    rate = cfg["rate"]
    seconds = cfg["minutes"] * 60.0
    seed = 20260924 + args.config
    # SUSHI_BENCH_CACHE=<dir>: keep the normalised streams of a workload between runs on one box (the profiling
    # scripts run this file several times; generating 2 x 86 M samples takes longer than the measurement)
    cache = os.environ.get("SUSHI_BENCH_CACHE")
    own_cache = False
    if cache is None and world > 1:
        # N ranks share the host's cores (the GPU boxes grant a 16-core quota): rank 0 generates the two streams ONCE and the
        # others read its file instead of each repeating the work.  One private directory per launch: keyed by the launcher's
        # pid (every rank's parent) and the rendezvous port, created 0700 by rank 0, owner-checked by the readers, removed at exit.
        cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sushi_bench_streams_%d_%d_%s" % (
            os.getuid(), os.getppid(), os.environ.get("MASTER_PORT", "0")))
        own_cache = True
    tag = "c%d_%g_%d_%s_%g_%g_%g_%d%s" % (args.config, cfg["minutes"], rate, args.sample_type, args.offset, args.hard_frac,
                                          args.snr, int(args.unrelated), "" if args.source == "noise" else "_" + args.source)
    cpath = os.path.join(cache, tag + ".npz") if cache else None
    hard_spans = []
    if cpath and rank != 0 and world > 1:
        t_wait = time.perf_counter()
        while not os.path.exists(cpath):                       # written atomically (os.replace) by rank 0
            if time.perf_counter() - t_wait > 1800:
                raise SystemExit("rank %d: no stream file from rank 0 after 30 min (%s)" % (rank, cpath))
            time.sleep(0.2)
        if own_cache:
            st_dir = os.lstat(cache)
            import stat as _stat
            if not _stat.S_ISDIR(st_dir.st_mode) or st_dir.st_uid != os.getuid() or (st_dir.st_mode & 0o077):
                raise SystemExit("rank %d: %s is not a private directory of this user" % (rank, cache))
    if cpath and os.path.exists(cpath):
        z = np.load(cpath, allow_pickle=False)
        dst = WavStream.from_prepared(z["dst"], rate, int(z["sample_count"]), int(z["padding_size"]))
        src = WavStream.from_prepared(z["src"], rate, int(z["sample_count"]), int(z["padding_size"]))
        hard_spans = [(str(k), float(a), float(b)) for k, a, b in zip(z["hard_kind"], z["hard_a"], z["hard_b"])]
    else:
        if args.source != "noise" and (args.hard_frac > 0 or args.unrelated):
            raise SystemExit("--source %s goes with neither --hard-frac nor --unrelated" % args.source)
        base_pcm = None
        if args.source == "dub":
            dst_pcm, src_pcm, _ = synth.make_dub_pcm(seconds, int(round(args.offset * rate)), rate, seed=seed)
        elif args.source == "partial":
            dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
            other = synth.make_dst_pcm(seconds, rate, seed=seed + 7777)
            half = dst_pcm.shape[0] // 2
            src_pcm = synth.make_src_pcm(np.concatenate([dst_pcm[:half + int(round(args.offset * rate))], other[half + int(round(args.offset * rate)):]]),
                                         int(round(args.offset * rate)), snr_db=args.snr, seed=seed + 1)
        elif args.source == "encode":
            dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
            src_pcm = synth.make_src_pcm_other_encode(dst_pcm, int(round(args.offset * rate)), rate, seed=seed + 1)
        else:
            if args.hard_frac > 0:
                dst_pcm, hard_spans = synth.make_hard_dst_pcm(seconds, rate, seed=seed)
            else:
                dst_pcm = synth.make_dst_pcm(seconds, rate, seed=seed)
            # --unrelated: the source is a planted copy of ANOTHER stream of the same kind -- nothing of it is in the destination
            base_pcm = synth.make_dst_pcm(seconds, rate, seed=seed + 7777) if args.unrelated else dst_pcm
            src_pcm = synth.make_src_pcm(base_pcm, int(round(args.offset * rate)), snr_db=args.snr, seed=seed + 1)
        dst = WavStream.from_samples(dst_pcm, rate, sample_rate=rate, sample_type=args.sample_type)
        src = WavStream.from_samples(src_pcm, rate, sample_rate=rate, sample_type=args.sample_type)
        del dst_pcm, src_pcm, base_pcm
        if cpath and rank == 0:
            os.makedirs(cache, mode=0o700, exist_ok=True)
            np.savez(cpath + ".tmp.npz", dst=dst.data, src=src.data, sample_count=dst.sample_count,
                     padding_size=dst.padding_size, hard_kind=np.array([k for k, _, _ in hard_spans], dtype="U16"),
                     hard_a=np.array([a for _, a, _ in hard_spans], np.float64),
                     hard_b=np.array([b for _, _, b in hard_spans], np.float64))
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
    ev_starts = np.array([s for s, _ in events])

    # ---- oracle leg first (rank 0): fork a pool before CUDA exists in this process --------------
    # Always: the parity sample (>= --cpu-sample evenly spaced searches + every search cut from tie-saturated material).
    # N = 1 and not --no-cpu-baseline: over a larger sample, timed = cpu_baseline.
    cpu, cpu_results = None, {}
    if rank == 0 and not dry and not args.profile_only:
        timed = world == 1 and not args.no_cpu_baseline
        cpu, cpu_results = oracle_leg(dst.data[0], src.data[0], offs, lens, wst, npos, args.method,
                                      forced=np.nonzero(hard_mask)[0], timed=timed, min_sample=args.cpu_sample,
                                      workers=args.cpu_workers)

    import torch
    import torch.distributed as dist
    from sushi_amd.distributed import ShardedSearch

    if dry:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group(args.dry_backend)
        planted = [int(round((s + args.offset - st) * rate)) + args.dry_plant_error
                   for (s, _), st in zip(events, start_times)]

        def make_batch(lo, hi):
            return DryBatch(planted, lo, hi)
        setup_ms = None
        process_start_ms = None
    else:
        from sushi_amd import _native
        from sushi_amd.device import DEFAULT_DELTA, SearchBatch
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
        dst._device = src._device = dev
        # the PROCESS's one-time GPU start-up -- the context's first real use, the library's code object: 0.09-0.15 s whatever
        # the job (tools/setup_probe.py) -- on its own clock: a one-shot job starts it first thing (sushi_amd.device.warm_up(
        # background=True)) and has it behind itself when its audio is demuxed and decoded; here it cannot overlap anything
        # (the oracle leg had to fork before HIP existed), so it is timed, reported, and kept out of the job's set-up
        torch.cuda.synchronize(dev)
        t_p = time.perf_counter()
        from sushi_amd.device import warm_up
        warm_up(dev)
        process_start_ms = (time.perf_counter() - t_p) * 1e3
        # set-up a one-shot job pays before its first step, outside every per-step number: the streams cross PCIe and
        # get their prefix sums / block spectra; the batch is planned on the host and its descriptors uploaded
        torch.cuda.synchronize(dev)
        t_s = time.perf_counter()
        ddev, sdev = dst.device_stream(), src.device_stream()
        if args.path == "fft":
            ddev.searchable()
        torch.cuda.synchronize(dev)
        setup_ms = {"streams_upload_prefix_sums_spectra": (time.perf_counter() - t_s) * 1e3}

        def make_batch(lo, hi):
            return SearchBatch(ddev, sdev, offs[lo:hi], lens[lo:hi], wst[lo:hi], npos[lo:hi], variant=args.variant,
                               path=args.path, delta=DEFAULT_DELTA if args.delta is None else args.delta,
                               workspace_bytes=(160 << 30) if args.ws_mb is None else args.ws_mb << 20,
                               method=args.method, exclusion=args.exclusion)

    # blocks of equal WORK per rank: a step is as long as its slowest rank
    from sushi_amd.distributed import search_work          # pure host arithmetic: no library, no GPU (dry runs too)
    work = search_work(wst, npos, lens, args.path) if world > 1 else np.ones(n_total)
    t_b = time.perf_counter()
    sharded = ShardedSearch(n_total, make_batch, device=None if dry else dev, weights=work if world > 1 else None)
    batch = sharded.batch
    if not dry:
        torch.cuda.synchronize(dev)
        setup_ms["batch_plan_allocate_upload"] = (time.perf_counter() - t_b) * 1e3

    def sync():
        if not dry:
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
    sync()
    t_first = time.perf_counter()
    v_idx, _ = step()
    sync()
    first_step_ms = (time.perf_counter() - t_first) * 1e3          # the batch's FIRST run: what a one-shot job's only step is
    v_idx = v_idx.cpu().numpy()
    v_times = np.array(start_times) + v_idx.astype(np.float64) / float(rate)
    v_err = np.abs((v_times - ev_starts) - args.offset) * rate
    # An event whose result is more than one sample from the planted offset: with 20 dB of noise on a smooth signal the
    # true minimum can sit a sample beside the planted one (seen at 24 kHz), on top of the < 1 sample that the two
    # truncations of wav.py:173-175 contribute.  Beyond two samples the run is refused outright; between one and two the
    # oracle has to find the very same position.  Every rank holds the same gathered results and reaches the same
    # verdict by itself (the oracle runs in-process on the few events concerned), so all ranks leave together.
    planted_mask = ~hard_mask if not args.unrelated else np.zeros(n_total, bool)      # --unrelated: no planted answer at all
    if args.source == "partial":
        # the second half of the source is other audio: its events have no planted answer (the oracle sample covers them)
        planted_mask &= np.array([(e_ * rate) < (int(round(seconds * rate)) // 2) for _, e_ in events])
    if args.source == "dub":
        # an event under the dub's own speech has no planted answer either (the louder part of its pattern is not in the
        # destination): the planted check is for the events that lie in the shared bed alone, the oracle sample covers both kinds
        gate = synth.speech_gate(int(round(seconds * rate)), rate, seed + 11)
        cg = np.concatenate(([0], np.cumsum(gate)))
        for k, (s_, e_) in enumerate(events):
            a_, b_ = int((s_ + args.offset) * rate), int((e_ + args.offset) * rate) + 1
            a_, b_ = max(0, min(a_, gate.shape[0])), max(0, min(b_, gate.shape[0]))
            if cg[b_] - cg[a_] > 0:
                planted_mask[k] = False
    off_planted = [int(k) for k in np.nonzero((v_err > 1.0) & planted_mask)[0]]
    beyond_planted = {"events": len(off_planted), "confirmed_by_oracle": 0}
    if off_planted:
        worst = float(v_err[planted_mask].max())
        # (below 20 dB the true minimum wanders further from the planted position: the oracle alone decides then)
        if args.source not in ("dub", "partial") and (len(off_planted) > 32 or (worst > 2.0 and args.snr >= 20.0)):
            raise SystemExit("verification pass: planted offset not recovered on %d events (max error %.3f samples)"
                             % (len(off_planted), worst))
        _cpu_ctx.update(dst=dst.data[0], src=src.data[0], offs=offs, lens=lens, wst=wst, npos=npos, method=args.method)
        _cpu_ctx.pop("cv2_set", None)
        for k in off_planted:
            o = cpu_results.get(k) or _cpu_one(k)
            if o[0] != int(v_idx[k]):
                raise SystemExit("verification pass: event %d: position %d, oracle %d, planted offset missed by %.3f "
                                 "samples" % (k, int(v_idx[k]), o[0], float(v_err[k])))
            beyond_planted["confirmed_by_oracle"] += 1
    for _ in range(args.warmup):
        step()
    timer = (lambda: torch.cuda.Event(enable_timing=True)) if not dry else (lambda: None)
    starts = [timer() for _ in range(args.steps)]
    ends = [timer() for _ in range(args.steps)]
    sync()
    fft_prof = args.path == "fft" and batch is not None and not dry
    if fft_prof:
        _native.profile_begin()           # per-stage HIP events on the launch stream, read after the final sync
    t0 = time.perf_counter()
    for k in range(args.steps):
        idx_all, score_all = step(starts[k], ends[k])
    sync()
    elapsed = time.perf_counter() - t0
    stage_ms = None
    if fft_prof:
        stage_ms = _native.profile_end(args.steps).mean(axis=0)
    kernel_ms = elapsed / args.steps * 1e3 if dry else float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        local_elapsed, elapsed = elapsed, float(t.item())
        mine = {"rank": rank, "events": sharded.hi - sharded.lo, "kernels_ms_per_step": kernel_ms,
                "gather_and_wait_ms_per_step": local_elapsed / args.steps * 1e3 - kernel_ms, "setup_ms": setup_ms}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    idx_all = idx_all.cpu().numpy()
    score_all = score_all.cpu().numpy()

    # ---- what a G-rank run would give each of its ranks, timed on this one GPU ----
    shard_emulation = None
    if world == 1 and not dry and args.emulate_shards >= 2 and not args.profile_only:
        from sushi_amd.distributed import weighted_bounds
        w_all = search_work(wst, npos, lens, args.path)
        shard_emulation = {"label": "single-GPU shard emulation: every shard of the G-rank job timed alone on this GPU with the "
                                    "streams resident; the all-gather (8 B per event) and xGMI are NOT in it",
                           "one_gpu_ms_per_step": kernel_ms, "by_world_size": {}}
        g = 2
        while g <= args.emulate_shards:
            rows = []
            for r, (lo, hi) in enumerate(weighted_bounds(w_all, g)):
                sb = make_batch(lo, hi)
                for _ in range(2):
                    sb.run()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 8
                t_h = time.perf_counter()
                e0.record()
                for _ in range(reps):
                    sb.run()
                e1.record()
                host_ms = (time.perf_counter() - t_h) / reps * 1e3       # the launch chain of one run() on the host
                torch.cuda.synchronize(dev)
                ms = e0.elapsed_time(e1) / reps
                si, _ = sb.results()
                ok = bool((si == idx_all[lo:hi]).all())
                rows.append({"rank": r, "events": hi - lo, "work_over_mean": round(float(w_all[lo:hi].sum() / (w_all.sum() / g)), 4),
                             "ms_per_step": round(ms, 4), "host_launch_ms": round(host_ms, 4), "same_results_as_the_one_gpu_run": ok})
                del sb
            worst = max(x["ms_per_step"] for x in rows)
            shard_emulation["by_world_size"][str(g)] = {
                "max_shard_ms": worst, "mean_shard_ms": round(float(np.mean([x["ms_per_step"] for x in rows])), 4),
                "implied_events_per_s_gather_excluded": n_total / (worst * 1e-3),
                "implied_speedup_over_one_gpu": kernel_ms / worst, "shards": rows}
            torch.cuda.empty_cache()
            g *= 2

    if rank == 0:
        # parity on the whole job: planted offset recovered to +-1 sample on every (ordinary) event
        times = np.array(start_times) + idx_all.astype(np.float64) / float(rate)
        shift_err = np.abs((times - ev_starts) - args.offset) * rate
        max_shift_err_vs_planted = float(shift_err[planted_mask].max()) if planted_mask.any() else None
        diag_ps = None
        if fft_prof and world == 1:
            diag_ps = batch.diagnostics(per_search=True)
            # every search the exact stages had to finish through the tile kernels belongs in the oracle sample: the ones
            # the forced set did not foresee are run here, in-process
            _cpu_ctx.update(dst=dst.data[0], src=src.data[0], offs=offs, lens=lens, wst=wst, npos=npos, method=args.method)
            _cpu_ctx.pop("cv2_set", None)
            late = [int(k) for k in np.nonzero(diag_ps["flagged_per_search"])[0] if int(k) not in cpu_results]
            for k in late[:64]:
                cpu_results[k] = _cpu_one(k)
        parity = {"max_shift_err_samples_vs_planted": max_shift_err_vs_planted,
                  "events_beyond_one_sample_of_planted": beyond_planted,
                  "oracle_sample_searches": len(cpu_results),
                  "score_tolerance": "1e-4*score + 2.5e-7 (one float32 ulp of cv2's stored corr)"}
        ties = 0
        if cpu_results:
            ks = sorted(cpu_results)
            ie = np.array([abs(int(idx_all[k]) - cpu_results[k][0]) for k in ks])
            if args.sample_type == "float32" and ie.max() > 0:
                # a different index is a tie, not an error, if the oracle itself scores the two positions within the float32
                # quantum of cv2's stored cross term (tests/test_gpu_parity.py _check_f32): material that repeats exactly (a
                # recurring jingle, a held tone) ties exactly, the product returns the FIRST such position from exact
                # arithmetic, the oracle's float64 FFT may round a later one a quantum lower
                from oracle import oracle as O
                for j, k in enumerate(ks):
                    if ie[j] == 0:
                        continue
                    off, m, ws, p = offs[k], lens[k], wst[k], npos[k]
                    row = O.match_template_fft(dst.data[0][ws:ws + p + m - 1], src.data[0][off:off + m], method=args.method)[0]
                    if abs(float(row[int(idx_all[k])]) - float(row[cpu_results[k][0]])) <= SCORE_ATOL:
                        ie[j] = 0
                        ties += 1
            ae = np.array([abs(float(score_all[k]) - cpu_results[k][1]) for k in ks])
            # excess over the parity bound |d| <= 1e-4*score + 2.5e-7 (tests/test_gpu_parity.py); <= 1 passes
            se = np.array([abs(float(score_all[k]) - cpu_results[k][1]) / (SCORE_RTOL * abs(cpu_results[k][1]) + SCORE_ATOL)
                           for k in ks])
            parity.update(max_idx_err_vs_oracle_sample=int(ie.max()),
                          max_score_err_over_tolerance_vs_oracle_sample=float(se.max()),
                          max_abs_score_err_vs_oracle_sample=float(ae.max()),
                          oracle_sample_hard_searches=int(sum(1 for k in ks if hard_mask[k])),
                          oracle_sample_exact_ties_at_another_index=ties)
            if diag_ps is not None:
                fl = diag_ps["flagged_per_search"]
                parity["oracle_sample_flagged_searches"] = int(sum(1 for k in ks if fl[k]))
                parity["flagged_searches_not_in_oracle_sample"] = int(sum(1 for k in np.nonzero(fl)[0]
                                                                          if int(k) not in cpu_results))
            real = {k: r[3] for k, r in cpu_results.items() if r[3] is not None}
            if real:
                import cv2
                gi = np.array([abs(int(idx_all[k]) - r[0]) for k, r in real.items()])
                gs = np.array([abs(float(score_all[k]) - r[1]) / (SCORE_RTOL * abs(r[1]) + SCORE_ATOL)
                               for k, r in real.items()])
                oi = np.array([abs(cpu_results[k][0] - r[0]) for k, r in real.items()])
                parity["cv2"] = {"available": True, "version": cv2.__version__, "searches": len(real),
                                 "max_idx_err_gpu_vs_cv2": int(gi.max()),
                                 "max_score_err_over_tolerance_gpu_vs_cv2": float(gs.max()),
                                 "max_idx_err_oracle_vs_cv2": int(oi.max())}
            else:
                parity["cv2"] = {"available": False,
                                 "note": "`import cv2` fails on this machine: parity is against the oracle's restatement"}
        value = n_total * args.steps / elapsed
        flops_launch = batch.flops
        if dry:
            roofline = {"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": None,
                        "traffic": None, "note": "dry run: nothing was computed"}
        elif args.path == "fft":
            # Since round 5 no single kernel dominates the step (three stages of a quarter each, none of which moves the searches'
            # own bytes: they multiply, bound and transform spectra): the roofline figure is the WHOLE STEP's -- the algorithmic
            # bytes of one launch of the hot path over the HIP-event time of all its kernels --, the stage with the largest
            # HIP-event time is reported beside it with its own share and its own measured HBM traffic.
            stages = {n: float(v) for n, v in zip(_native.STAGE_NAMES, stage_ms)}
            dom = max(stages, key=stages.get)
            dom_ms = stages[dom]
            kname = _native.STAGE_KERNELS[dom]
            achieved = batch.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9
            # HBM bytes per step of every kernel from the committed rocprofv3 PMC passes of this very workload
            # (profiles/pmc_traffic.json, made by tools/make_pmc_traffic.py).  The entry records the digest of the
            # kernel sources it was measured on: a different digest means the kernels changed since -> null.
            traffic, dom_traffic_bytes, step_traffic, traffic_note = None, None, None, "no PMC entry for this workload"
            wl_key = "config%d/fft/%s/%d/w%g/m%g/n%d" % (args.config, args.sample_type, n_total, cfg["window"],
                                                        cfg["minutes"], world)
            if args.method != "sqdiff_normed":
                wl_key += "/" + args.method
            if args.hard_frac > 0 or args.offset != 7.25:
                wl_key += "/hard%g/off%g" % (args.hard_frac, args.offset)
            if args.snr != 20.0 or args.unrelated:
                wl_key += "/snr%g%s" % (args.snr, "/unrelated" if args.unrelated else "")
            if args.source != "noise":
                wl_key += "/source-" + args.source
            if args.exclusion not in (None, "auto"):
                wl_key += "/exclusion-" + args.exclusion
            digest = kernel_source_digest()
            per_kernel = None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    entry = json.load(f)[wl_key]
                if os.environ.get("SUSHI_HIP_LIB"):
                    traffic_note = "SUSHI_HIP_LIB names another build of the library: the committed PMC passes are not of it"
                elif entry.get("kernel_source_digest") != digest:
                    traffic_note = "stale: PMC passes were taken on kernel sources %s, this is %s" % (
                        entry.get("kernel_source_digest"), digest)
                else:
                    step_traffic = float(sum(k["fetch_bytes"] + k["write_bytes"] for k in entry["kernels"].values()))
                    traffic = step_traffic / (kernel_ms * 1e-3) / 1e9
                    kerns = [entry["kernels"][kn] for kn in _native.STAGE_KERNEL_SETS[dom] if kn in entry["kernels"]]
                    dom_traffic_bytes = float(sum(kk["fetch_bytes"] + kk["write_bytes"] for kk in kerns)) if kerns else None
                    per_kernel = {kn: {"read": kk["fetch_bytes"], "written": kk["write_bytes"]} for kn, kk in sorted(entry["kernels"].items())}
                    traffic_note = "%s @ %s" % (entry.get("source"), entry.get("source_commit"))
                    # where a kernel's write bytes are not a WRITE_SIZE pass (tools/make_pmc_traffic.py), say what they are
                    others = sorted(set(k.get("write_source", "WRITE_SIZE") for k in entry["kernels"].values()) - {"WRITE_SIZE"})
                    if others:
                        traffic_note += "; reads: FETCH_SIZE; writes: " + " / ".join(others)
            except Exception:
                pass
            roofline = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBPS, "traffic": traffic,
                        "basis": "whole step: algorithmic bytes of one launch of the hot path / HIP-event time of all its kernels; "
                                 "`traffic` = PMC bytes (FETCH_SIZE x 2 + WRITE_SIZE, all kernels of a step) / the same time",
                        "kernel": "all kernels of one sushi_hip_batch_run", "kernel_ms": kernel_ms,
                        "step_traffic_bytes": step_traffic,
                        "step_traffic_over_algorithmic": None if step_traffic is None else
                        step_traffic / batch.algorithmic_bytes,
                        "traffic_bytes_per_kernel_per_step": per_kernel,
                        # (on lanes the stages of different sub-batches run side by side: their HIP-event times add up to more than
                        # the step, so a stage's share is its share of their SUM)
                        "dominant_stage": {"stage": dom, "kernels": kname, "ms": dom_ms,
                                           "share_of_step": dom_ms / max(sum(stages.values()), kernel_ms),
                                           "traffic_bytes_per_step": dom_traffic_bytes,
                                           "traffic_GBps": None if dom_traffic_bytes is None else dom_traffic_bytes / (dom_ms * 1e-3) / 1e9},
                        "traffic_key": wl_key, "traffic_source": traffic_note,
                        "kernel_source_digest": digest,
                        # the binary that actually ran (the digest above is of the sources on disk)
                        "library": loaded_library(),
                        "launches_per_step": batch.sub_batches,
                        "sub_batches": batch.sub_batches, "lanes": batch.lanes,
                        "stage_ms": stages,
                        "stage_ms_basis": "HIP events on the stream each sub-batch runs on, summed over the sub-batches" + (
                            " -- which run side by side on %d HIP streams (lanes): the stages add up to more than the step; "
                            "SUSHI_HIP_LANES=1:1 gives the one-after-the-other times" % batch.lanes if batch.lanes > 1 else ""),
                        "step_kernels_ms": kernel_ms,
                        "step_hbm_achieved_GBps": achieved,
                        "step_frac": achieved / PEAK_HBM_GBPS,
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
                       # sum of the searches' work (block pairs x segments) per rank over the mean: the step is its slowest rank
                       "work_per_gpu_over_mean": [round(float(work[a:b].sum() / (work.sum() / world)), 4)
                                                  for a, b in sharded.all_bounds()],
                       "window_s": cfg["window"], "stream_minutes": cfg["minutes"], "sample_rate": rate,
                       "sample_type": args.sample_type, "hard_events": int(hard_mask.sum()), "events_with_a_planted_answer": int(planted_mask.sum()),
                       "source_snr_db": args.snr if args.source == "noise" else None, "source_unrelated_to_destination": bool(args.unrelated),
                       "source_kind": {"noise": "destination advanced by the offset + white noise",
                                       "encode": "another encode: gain 0.7, 4 kHz low-pass, requantised to 8 bits",
                                       "dub": "shared music bed, each stream's own speech on half of the time",
                                       "partial": "first half of the programme: the destination + white noise; second half: other audio"}[args.source],
                       "method": METHOD_TEXT[args.method],
                       "path": ("overlap-save FFT (f32) + exact float64 re-evaluation of the near-minimum positions"
                                if args.path == "fft" else "direct exact-f32 MFMA sliding dot product"),
                       "parallelism": "events sharded in contiguous blocks over %d GPU(s), streams replicated, "
                                      "one all-gather of (idx, score)" % world,
                       "kernel_variant": batch.variant},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": parity,
            # paid once per job, before the first step; outside `value` (inputs resident in HBM when the timed region starts)
            "setup_ms": setup_ms,
            # ... and once per PROCESS, whatever the job: the context's first use + the library's code object (sushi_amd.device.
            # warm_up; a one-shot job overlaps it with its demux / decode)
            "process_start_ms": process_start_ms,
            # `value` is a RESIDENT-STATE rate (streams, spectra, plan and workspace in HBM, the same job every step).  What a
            # one-shot job -- sushi.py:663-672: two WavStream loads, then one calculate_shifts pass -- gets from this process:
            # events / (set-up + one step), with the process's start-up behind it -- and with the start-up on the critical path
            # (its ONE step is the batch's first run -- the exclusion's form is voted on, nothing is warm --: `first_step_ms`, wall
            # clock around the untimed verification pass; until round 6's last day these two figures took a steady-state step instead,
            # which a first run was 9 - 13 ms away from: the vote's same-address atomics, the lanes' streams made inside the run)
            "first_step_ms": first_step_ms,
            "one_shot_events_per_s": None if not setup_ms else n_total / ((sum(setup_ms.values()) + first_step_ms) * 1e-3),
            "one_shot_incl_process_start_events_per_s": None if not setup_ms else
                n_total / ((process_start_ms + sum(setup_ms.values()) + first_step_ms) * 1e-3),
            "one_shot_with_a_steady_state_step_events_per_s": None if not setup_ms else
                n_total / ((sum(setup_ms.values()) + elapsed / args.steps * 1e3) * 1e-3),
        }
        if dry:
            out["dry_run"] = "control flow only (--dry-backend %s): `value` is not a measurement" % args.dry_backend
        if per_rank is not None:
            out["per_rank"] = per_rank
        if shard_emulation is not None:
            out["shard_emulation"] = shard_emulation
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if own_cache and rank == 0:
            import shutil
            shutil.rmtree(cache, ignore_errors=True)


if __name__ == "__main__":
    main()
