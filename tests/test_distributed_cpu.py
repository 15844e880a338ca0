"""The N>1 host path (contiguous event blocks + one all-gather of (idx, score)) on CPU:
two gloo processes, the oracle standing in for the GPU batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sushi_amd.distributed import ShardedSearch, gather_results, max_shard, shard_bounds


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000, 3001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) == max_shard(n, w) or n == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_q):
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

    sh = ShardedSearch(n_total, OracleBatch)
    sh.run()                                           # (twice: the second step reuses the gather's buffers)
    idx, score = sh.run()
    full = OracleBatch(0, n_total).run()
    ok = bool((idx == full[0]).all()) and bool((score.view(torch.int32) == full[1].view(torch.int32)).all())
    ok = ok and idx.shape[0] == n_total
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
