"""Sharding the event batch across the GPUs of one node (one process per GPU, RCCL over xGMI).

The path shards by event: every search reads only the two (replicated) streams, so ranks share
nothing on the data path.  Each rank takes a contiguous block of the (time-sorted) searches --
neighbouring events overlap in the destination stream, which keeps a rank's working set local --
and the only collective is one all-gather of the per-event (index, score) pairs: 8 bytes per event.
Blocks are cut by WORK, not by count ("balance by sum P*M"): a search costs its block pairs x pattern
segments on the FFT path, and a step is as long as its slowest rank.
"""
import numpy as np
import torch
import torch.distributed as dist


FFT_SEG, FFT_STEP = 4096, 6          # csrc/sushi_common.hpp: samples per block / pattern segment, blocks between block pairs


def fft_layout_host(win_start, n_pos, tmpl_len):
    """(block pairs, pattern segments) of each request on the FFT path: sushi_common.hpp fft_layout in NumPy -- the block
    pairs of a search sit on the absolute pair grid, from the pair holding its first position to the one holding its last.
    No library, no GPU (tests/test_distributed_cpu.py holds it to sushi_hip_fft_layout)."""
    ws = np.asarray(win_start, dtype=np.int64).reshape(-1)
    p = np.asarray(n_pos, dtype=np.int64).reshape(-1)
    m = np.asarray(tmpl_len, dtype=np.int64).reshape(-1)
    pair0 = (ws // FFT_SEG) // FFT_STEP
    pair_last = ((ws + p - 1) // FFT_SEG) // FFT_STEP
    return pair_last - pair0 + 1, (m + FFT_SEG - 1) // FFT_SEG


def search_work(win_start, n_pos, tmpl_len, path="fft"):
    """Relative cost of each search, for cutting a batch into blocks of equal work (weighted_bounds): FFT path -- block pairs
    x (1 + 0.074 pattern segments), what the multiply-accumulate and the bound pass walk (measured stage times at BASELINE
    configs[2]); direct path -- P x M multiply-adds.  Host arithmetic only."""
    if path != "fft":
        return np.asarray(n_pos, np.float64).reshape(-1) * np.asarray(tmpl_len, np.float64).reshape(-1)
    pairs, segs = fft_layout_host(win_start, n_pos, tmpl_len)
    return pairs.astype(np.float64) * (1.0 + 0.074 * segs.astype(np.float64))


def shard_bounds(n_items, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(n_items, world_size):
    return (n_items + world_size - 1) // world_size


def weighted_bounds(weights, world_size):
    """Contiguous blocks [(lo, hi)] * world_size over len(weights) items with nearly equal sums of `weights`
    (all > 0): boundary r is where the running sum crosses r / world_size of the total -- every rank computes the same
    cut from the same descriptors.  No block is empty while there are at least world_size items."""
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    n = w.shape[0]
    if n == 0:
        return [(0, 0)] * world_size
    c = np.concatenate(([0.0], np.cumsum(w)))
    cuts = [0]
    for r in range(1, world_size):
        target = c[-1] * r / world_size
        k = int(np.searchsorted(c, target))                    # first k with c[k] >= target
        if k > 0 and abs(c[k - 1] - target) <= abs(c[k] - target):
            k -= 1                                              # the nearer of the two neighbouring cuts
        lo_allowed = cuts[-1] + 1 if n >= world_size else cuts[-1]
        k = max(k, lo_allowed)
        k = min(k, n - (world_size - r) if n >= world_size else n)
        cuts.append(k)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


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
    `weights` (one positive number per search, the same on every rank): blocks of equal work instead of equal count.

    The gather's buffers are allocated once; a batch that can write its (index, score bits) records itself (SearchBatch.
    set_packed_output) does so from the second step on: a step is then the library's kernels, the one collective and one (equal
    blocks) or two (unequal blocks) small kernels on the way out; at 375 events per rank (BASELINE configs[2] on 8 ranks) a step
    is ~1.6 ms of kernels (bench.py --emulate-shards 8), and a dozen tiny launches around the collective would be several per cent of it.

    `always_collective`: run the collective code path even when the group has ONE rank (tests: RCCL's all_gather_into_tensor on
    device memory, exercised on a one-GPU box).

    CONTRACT of gather() / run(): the two returned tensors are CONTIGUOUS rows of one persistent buffer of this object --
    valid until the next gather(), which overwrites them in place.  Keep results across steps with .clone()."""

    def __init__(self, n_total, make_batch, group=None, device=None, weights=None, always_collective=False):
        self.n_total = n_total
        self.always_collective = bool(always_collective)
        self.group = group
        self.device = device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if weights is not None and len(weights) != n_total:
            raise ValueError("one weight per search")
        self._bounds = (weighted_bounds(weights, self.world) if weights is not None else
                        [shard_bounds(n_total, r, self.world) for r in range(self.world)])
        self.lo, self.hi = self._bounds[self.rank]
        self.batch = make_batch(self.lo, self.hi) if self.hi > self.lo else None
        self._packed = self._full = self._out = self._rows = self._sel = None
        self._self_packed = False

    def all_bounds(self):
        return list(self._bounds)

    def run_local(self):
        """This rank's block: (idx int32[hi - lo], score float32[hi - lo]) on this rank's device."""
        if self.batch is not None:
            return self.batch.run()
        dev = self.device if self.device is not None else torch.device("cpu")
        return torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.float32, device=dev)

    def gather(self, idx, score):
        """The one collective of the path: everyone's (idx, score) in global search order -- of the tensors PASSED IN (this
        rank's block, hi - lo entries).  When they are the batch's own output tensors and the batch leaves its records in the
        gather buffer itself (from the second call on), no copy is made; any other tensors are packed by two small copies."""
        if self.world == 1 and not self.always_collective:
            return idx, score
        pad = max(hi - lo for lo, hi in self._bounds)
        if self._packed is None or self._packed.device != idx.device:
            dev = idx.device
            self._packed = torch.full((pad, 2), -1, dtype=torch.int32, device=dev)
            self._full = torch.empty((self.world * pad, 2), dtype=torch.int32, device=dev)
            self._out = torch.empty((2, self.n_total), dtype=torch.int32, device=dev)
            if self.n_total != self.world * pad:              # unequal blocks: the rows of the gathered buffer that exist
                sel = np.concatenate([np.arange(hi - lo, dtype=np.int64) + r * pad for r, (lo, hi) in enumerate(self._bounds)])
                self._sel = torch.from_numpy(sel).to(dev)
                self._rows = torch.empty((self.n_total, 2), dtype=torch.int32, device=dev)
            # a batch that can write its (index, score bits) records itself does, from now on: a step is then the library's
            # kernels, the one collective and one small kernel on the way out -- no copies in between
            if self.batch is not None and hasattr(self.batch, "set_packed_output") and idx.is_cuda:
                self.batch.set_packed_output(self._packed)
                self._self_packed = True
                n = idx.shape[0]                                  # (this step's results exist already: packed by hand once)
                self._packed[:n, 0].copy_(idx)
                self._packed[:n, 1].copy_(score.view(torch.int32))
        n = idx.shape[0]
        # (the records the batch wrote itself are only THESE results if the caller hands in the batch's own outputs)
        own = self._self_packed and idx.data_ptr() == self.batch.out_idx.data_ptr() and \
            score.data_ptr() == self.batch.out_score.data_ptr()
        if not own:
            self._packed[:n, 0].copy_(idx)
            self._packed[:n, 1].copy_(score.view(torch.int32))
        dist.all_gather_into_tensor(self._full, self._packed, group=self.group)
        rows = self._full
        if self._sel is not None:
            torch.index_select(self._full, 0, self._sel, out=self._rows)
            rows = self._rows
        self._out.copy_(rows.t())                             # de-interleave: two contiguous rows
        return self._out[0], self._out[1].view(torch.float32)

    def run(self):
        return self.gather(*self.run_local())
