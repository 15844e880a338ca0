"""The N > 1 path on real GPUs (RCCL): needs two devices, skipped on a one-GPU box (the round driver's 8-GPU node runs
it).  What the CPU (gloo) twin in tests/test_distributed_cpu.py cannot check: that everything handed to the collective is
a device tensor -- including the contribution of a rank that has no searches."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from sushi_amd.device import DeviceStream, SearchBatch
    from sushi_amd.distributed import ShardedSearch
    rng = np.random.default_rng(0)                      # the same streams on every rank (replicated)
    dst = rng.random(200000, dtype=np.float32)
    src = dst[20000:120000].copy()
    offs = [1000 + 9000 * k for k in range(n_total)]
    lens = [5000 + 100 * k for k in range(n_total)]
    wst = [0] * n_total
    npos = [150000] * n_total
    d, s = DeviceStream(dst, device=dev), DeviceStream(src, device=dev)

    def make(lo, hi):
        return SearchBatch(d, s, offs[lo:hi], lens[lo:hi], wst[lo:hi], npos[lo:hi])
    sh = ShardedSearch(n_total, make, device=dev)
    idx, score = sh.run()
    torch.cuda.synchronize(dev)
    want = [20000 + o for o in offs]                    # planted copies
    ok = idx.is_cuda and idx.shape[0] == n_total and [int(x) for x in idx.cpu()] == want
    out_q.put((rank, bool(ok), (sh.lo, sh.hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 1])             # 1: rank 1 has nothing to search and still takes part in the gather
def test_two_rank_nccl_shard_and_gather(n_total):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)


def test_a_rank_without_searches_contributes_device_tensors():
    """One device is enough for this part: ShardedSearch with nothing to do allocates its empty result on the GPU."""
    from sushi_amd.distributed import ShardedSearch, pack_results
    dev = torch.device("cuda", 0)
    sh = ShardedSearch(0, lambda lo, hi: None, device=dev)
    idx, score = sh.run_local()
    assert idx.is_cuda and score.is_cuda and idx.numel() == 0
    packed = pack_results(idx, score, 3)
    assert packed.is_cuda and packed.shape == (3, 2) and int(packed[0, 0]) == -1


def test_the_rccl_collective_path_on_one_gpu():
    """a one-GPU box can still load librccl, initialise the `nccl` backend and push DEVICE tensors through
    ShardedSearch.gather's real collective code path (world size 1, the early-out switched off): all_gather_into_tensor on HBM,
    the batch writing its (index, score bits) records into the gather buffer itself from the second step on, and -- ADVICE r5 --
    foreign tensors handed to gather() are gathered as they are, not replaced by the batch's last results."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from sushi_amd.device import DeviceStream, SearchBatch
        from sushi_amd.distributed import ShardedSearch
        rng = np.random.default_rng(0)
        dst = rng.random(200000, dtype=np.float32)
        src = dst[20000:120000].copy()
        n_total = 5
        offs = [1000 + 9000 * k for k in range(n_total)]
        lens = [5000 + 100 * k for k in range(n_total)]
        d, s = DeviceStream(dst, device=dev), DeviceStream(src, device=dev)
        sh = ShardedSearch(n_total, lambda lo, hi: SearchBatch(d, s, offs[lo:hi], lens[lo:hi], [0] * (hi - lo), [150000] * (hi - lo)),
                           device=dev, always_collective=True)
        want = [20000 + o for o in offs]
        for step in range(3):                                  # step 0 packs by hand, steps 1.. let the library write the records
            idx, score = sh.run()
            torch.cuda.synchronize(dev)
            assert idx.is_cuda and score.is_cuda and [int(x) for x in idx.cpu()] == want, step
            assert float(score.max()) < 1e-5
        assert sh._self_packed
        # foreign tensors: gathered as passed in
        fi = torch.arange(100, 100 + n_total, dtype=torch.int32, device=dev)
        fs = torch.full((n_total,), 0.25, dtype=torch.float32, device=dev)
        gi, gs = sh.gather(fi, fs)
        torch.cuda.synchronize(dev)
        assert [int(x) for x in gi.cpu()] == list(range(100, 100 + n_total)) and float(gs.min()) == float(gs.max()) == 0.25
        idx, score = sh.run()                                  # ... and the batch's own results again afterwards
        torch.cuda.synchronize(dev)
        assert [int(x) for x in idx.cpu()] == want
    finally:
        dist.destroy_process_group()
