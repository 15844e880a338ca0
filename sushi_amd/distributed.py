"""Sharding the event batch across the GPUs of one node (one process per GPU, RCCL over xGMI).

The path shards by event: every search reads only the two (replicated) streams, so ranks share
nothing on the data path.  Each rank takes a contiguous block of the (time-sorted) searches --
neighbouring events overlap in the destination stream, which keeps a rank's working set local --
and the only collective is one all-gather of the per-event (index, score) pairs: 8 bytes per event.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(n_items, world_size):
    return (n_items + world_size - 1) // world_size


def pack_results(idx, score, pad_to):
    """(idx int32[n], score float32[n]) -> int32[pad_to, 2] (score bits in column 1); padding = -1."""
    n = idx.shape[0]
    out = torch.full((pad_to, 2), -1, dtype=torch.int32, device=idx.device)
    out[:n, 0] = idx
    out[:n, 1] = score.view(torch.int32)
    return out


def gather_results(idx, score, n_total, group=None):
    """All ranks call this with their local block's results; every rank gets the full
    (idx int32[n_total], score float32[n_total]) in global search order."""
    world = dist.get_world_size(group)
    pad = max_shard(n_total, world)
    packed = pack_results(idx, score, pad)
    full = torch.empty((world * pad, 2), dtype=torch.int32, device=packed.device)
    dist.all_gather_into_tensor(full, packed, group=group)
    full = full.view(world, pad, 2)
    pieces_i, pieces_s = [], []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        pieces_i.append(full[r, :hi - lo, 0])
        pieces_s.append(full[r, :hi - lo, 1])
    return torch.cat(pieces_i), torch.cat(pieces_s).view(torch.float32)


class ShardedSearch(object):
    """Runs this rank's block of a global descriptor list and gathers everyone's results.

    `make_batch(lo, hi)` must return an object with `.run()` -> (idx, score) device tensors for
    searches [lo, hi) (a `SearchBatch` on the GPU; the CPU tests pass a stand-in).  `device`: where a rank
    WITHOUT searches (more ranks than searches) allocates its empty contribution -- RCCL gathers device
    tensors only, so it must be this rank's GPU under the nccl backend (None: CPU, for gloo).

    The gather's buffers are allocated once: a step costs two small copies into the packed block, the one collective,
    and -- when every rank holds the same number of searches (3000 events on 8 GPUs) -- no kernel at all on the way out
    (the results are strided views of the gathered buffer); at 375 events per rank a step is 3.5 ms of kernels, and a
    dozen tiny launches around the collective would be a few per cent of it."""

    def __init__(self, n_total, make_batch, group=None, device=None):
        self.n_total = n_total
        self.group = group
        self.device = device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.lo, self.hi = shard_bounds(n_total, self.rank, self.world)
        self.batch = make_batch(self.lo, self.hi) if self.hi > self.lo else None
        self._packed = self._full = None

    def all_bounds(self):
        return [shard_bounds(self.n_total, r, self.world) for r in range(self.world)]

    def run_local(self):
        """This rank's block: (idx int32[hi - lo], score float32[hi - lo]) on this rank's device."""
        if self.batch is not None:
            return self.batch.run()
        dev = self.device if self.device is not None else torch.device("cpu")
        return torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.float32, device=dev)

    def gather(self, idx, score):
        """The one collective of the path: everyone's (idx, score) in global search order."""
        if self.world == 1:
            return idx, score
        pad = max_shard(self.n_total, self.world)
        if self._packed is None or self._packed.device != idx.device:
            self._packed = torch.full((pad, 2), -1, dtype=torch.int32, device=idx.device)
            self._full = torch.empty((self.world * pad, 2), dtype=torch.int32, device=idx.device)
        n = idx.shape[0]
        self._packed[:n, 0].copy_(idx)
        self._packed[:n, 1].copy_(score.view(torch.int32))
        dist.all_gather_into_tensor(self._full, self._packed, group=self.group)
        if self.n_total == self.world * pad:                  # equal blocks: the gathered buffer IS the result, in order
            return self._full[:, 0], self._full[:, 1].view(torch.float32)
        full = self._full.view(self.world, pad, 2)
        pieces_i, pieces_s = [], []
        for r in range(self.world):
            lo, hi = shard_bounds(self.n_total, r, self.world)
            pieces_i.append(full[r, :hi - lo, 0])
            pieces_s.append(full[r, :hi - lo, 1])
        return torch.cat(pieces_i), torch.cat(pieces_s).view(torch.float32)

    def run(self):
        return self.gather(*self.run_local())
