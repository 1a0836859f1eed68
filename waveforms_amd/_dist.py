"""Multi-GPU: one process per GPU, channel-block-per-rank.

Channels are independent objects, so the data path has NO
collective: rank r flattens, compiles and samples only its own contiguous block of
channels.  A collective (torch.distributed all_gather == RCCL over xGMI with backend
"nccl") exists only for optional result placement and for timing reductions.
"""
from __future__ import annotations

import numpy as np


def channel_block(n_channels: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced block [start, stop) of rank `rank`."""
    if not 0 <= rank < world:
        raise ValueError('rank out of range')
    base, extra = divmod(n_channels, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class ShardedSampler:
    """The rank-local part of a multi-channel sampling job.

    `make_channel(c)` builds channel c (a Waveform / WaveVStack); only the local block
    is ever built, flattened and uploaded."""

    def __init__(self, n_channels, make_channel, grid, rank, world, plan_only=False, tile=1):
        """`tile`: the job's rows are `tile` copies of n_channels / tile distinct channels (synthetic
        batches); a rank then builds only the distinct channels of its block."""
        from ._sampling import BatchSampler
        self.rank, self.world, self.n_channels = rank, world, n_channels
        self.start, self.stop = channel_block(n_channels, rank, world)
        if tile > 1:
            if (self.stop - self.start) % tile:
                raise ValueError('rows per rank must be a multiple of tile')
            distinct = (self.stop - self.start) // tile
            first = self.start // tile
            self.local = BatchSampler([make_channel(c) for c in range(first, first + distinct)], grid, tile=tile)
        else:
            self.local = BatchSampler([make_channel(c) for c in range(self.start, self.stop)],
                                      grid)
        self.n = self.local.n

    def launch_torch(self, out, accumulate=False):
        return self.local.launch_torch(out, accumulate)


def gather_rows(local, n_channels, group=None):
    """Optional result placement: all-gather the (rows_r, n) blocks of every rank into
    one (n_channels, n) tensor on every rank (RCCL all_gather over xGMI on GPUs, gloo
    on CPU).  Blocks may differ by one row; they are padded to the largest block."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rows = [channel_block(n_channels, r, world) for r in range(world)]
    width = max(b - a for a, b in rows)
    pad = torch.zeros((width, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:b - a] for p, (a, b) in zip(parts, rows)], dim=0)


def max_over_ranks(value: float, device='cpu', group=None) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
