"""The N>1 host path (contiguous event blocks + one all-gather of (idx, score)) on CPU:
two gloo processes, the oracle standing in for the GPU batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sushi_amd.distributed import ShardedSearch, gather_results, max_shard, shard_bounds, weighted_bounds


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000, 3001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) == max_shard(n, w) or n == 0


def test_weighted_bounds_cover_everything_and_balance():
    rng = np.random.default_rng(7)
    for n in (0, 1, 3, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            wt = rng.uniform(1.0, 5.0, n)
            spans = weighted_bounds(wt, w)
            assert len(spans) == w and spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            if n >= w:
                assert all(h > l for l, h in spans)                    # nobody idles while there is work for everyone
    wt = np.ones(3000)
    assert weighted_bounds(wt, 8) == [shard_bounds(3000, r, 8) for r in range(8)]      # equal weights: the count split


def test_blocks_of_equal_work_at_baseline_config2_sizes():
    """SURVEY 8e / VERDICT r3: "balance by sum P*M, not by count".  BASELINE configs[2]'s own descriptors (3000 events, 2-h
    12 kHz streams, +-120 s, patterns U[1,5] s) cut for 8 ranks: the work of a rank -- block pairs x (1 + 0.074 segments),
    what mac_kernel and ifft_kernel walk -- is within 2 % of the mean (cut by COUNT it is off by several per cent)."""
    from sushi_amd import synth
    from sushi_amd.device import search_work
    from sushi_amd.wav import WavStream
    rate, seconds, n_total, window, offset = 12000, 7200.0, 3000, 120.0, 7.25
    n = int(20 * rate + seconds * rate)
    row = np.lib.stride_tricks.as_strided(np.zeros(1, np.float32), shape=(1, n), strides=(0, 0))   # only its SHAPE is read
    dst = WavStream.__new__(WavStream)
    dst.data, dst.sample_rate, dst.sample_count, dst.padding_size = row, rate, int(seconds * rate), 10 * rate
    events = synth.make_events(n_total, seconds, window + offset, seed=20260926 + 2)
    rng = np.random.default_rng(20260926 + 3)
    wst, npos, lens = [], [], []
    for s, e in events:
        m = dst._get_sample_for_time(e) - dst._get_sample_for_time(s)
        c = s + offset + float(rng.uniform(-window * 0.5, window * 0.5))
        _, lo, p = dst._window(m, c, window)
        wst.append(lo); npos.append(p); lens.append(m)
    work = search_work(wst, npos, lens, "fft")
    for world in (2, 4, 8):
        spans = weighted_bounds(work, world)
        per = np.array([work[a:b].sum() for a, b in spans])
        assert per.max() / per.mean() <= 1.02, (world, per / per.mean())
        assert sum(b - a for a, b in spans) == n_total
    by_count = np.array([work[a:b].sum() for a, b in (shard_bounds(n_total, r, 8) for r in range(8))])
    assert by_count.max() / by_count.mean() > per.max() / per.mean()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_q, weighted=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    rng = np.random.default_rng(0)                      # same streams on every rank (replicated)
    dst = rng.random(6000, dtype=np.float32)
    src = rng.random(3000, dtype=np.float32)
    offs = [37 * k for k in range(n_total)]
    lens = [50 + 11 * k for k in range(n_total)]

    class OracleBatch(object):
        def __init__(self, lo, hi):
            self.lo, self.hi = lo, hi

        def run(self):
            idx, sc = [], []
            for k in range(self.lo, self.hi):
                r = O.match_template_direct(dst, src[offs[k]:offs[k] + lens[k]])[0]
                idx.append(int(r.argmin()))
                sc.append(r[idx[-1]])
            return torch.tensor(idx, dtype=torch.int32), torch.tensor(np.array(sc, np.float32))

    sh = ShardedSearch(n_total, OracleBatch, weights=[float(l) for l in lens] if weighted else None)
    sh.run()                                           # (twice: the second step reuses the gather's buffers)
    idx, score = sh.run()
    full = OracleBatch(0, n_total).run()
    ok = bool((idx == full[0]).all()) and bool((score.view(torch.int32) == full[1].view(torch.int32)).all())
    ok = ok and idx.shape[0] == n_total and idx.is_contiguous() and score.is_contiguous()     # (the gather's contract)
    out_q.put((rank, ok, (sh.lo, sh.hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 1, 6])      # uneven blocks, a rank without searches, equal blocks (strided views out)
def test_two_rank_gloo_shard_and_gather(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    spans = sorted(s for _, _, s in res)
    assert spans[0][0] == 0 and spans[-1][1] == n_total and spans[0][1] == spans[1][0]


def test_two_rank_gloo_blocks_of_equal_work():
    """The same with weights: the longer patterns at the end of the list make rank 1's block the shorter one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 9, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    spans = sorted(s for _, _, s in res)
    assert spans[0][0] == 0 and spans[-1][1] == 9 and spans[0][1] == spans[1][0]
    assert spans[0][1] - spans[0][0] > spans[1][1] - spans[1][0]


def test_host_fft_layout_equals_the_library():
    """bench.py's dry runs and the work split use NumPy arithmetic (no library): it must be sushi_hip_fft_layout's."""
    from sushi_amd import _native
    from sushi_amd.distributed import fft_layout_host
    rng = np.random.default_rng(5)
    ws = rng.integers(0, 90_000_000, 400)
    npos = rng.integers(1, 6_000_000, 400)
    m = rng.integers(1, 200_000, 400)
    ws[:4] = [0, 4095, 4096 * 6 - 1, 4096 * 6]
    npos[:4] = [1, 1, 2, 24576]
    pairs, segs = fft_layout_host(ws, npos, m)
    for k in range(400):
        assert _native.fft_layout(ws[k], npos[k], m[k]) == (int(pairs[k]), int(segs[k]))
