"""Multi-GPU: one process per GPU.

Two ways to cut a job, both without a data-path collective:

* **channel-block-per-rank** (`ShardedSampler`): channels are independent objects; rank r flattens,
  compiles and samples only its own contiguous block of channels.
* **time-slice-per-rank** (`TimeShardedSampler`, `TimeShardedFir`, `TimeShardedIir`): a few very long rows
  (C2: 1 channel x 1e7..1e9 points).  Rank r samples [a_r, b_r) of every row as a SLICE of the caller's grid
  (`wfk_grid.i0`: the slice's times, piece indices and samples are those of the same samples of the whole
  grid).  A FIR stage needs K - 1 neighbouring samples: the sampler is pure, so each rank RECOMPUTES that
  halo itself instead of exchanging it (reference analogue: chunked sampling, waveforms/waveform.py:209-257;
  FIR crop distortion.py:329-337).  An IIR stage is a recurrence: the state zf of rank r - 1 is rank r's zi
  (the reference carries `zi` from chunk to chunk the same way, waveform.py:244-251) -- a (rows, D) message
  per rank boundary, handed on in rank order.

Collectives (torch.distributed: RCCL over xGMI with backend "nccl", gloo on CPU) exist only for optional
RESULT PLACEMENT (`gather_rows`, `gather_rows_to_host`) and for timing reductions.
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


def gather_rows_to_host(local, n_channels, root=0, slab_bytes=2 << 30, group=None):
    """Result placement for jobs whose gathered size exceeds one GPU (C5: 4096 x 1e7 fp64 = 328 GB > 288 GB
    HBM): the row blocks travel to `root` in SLABS of <= `slab_bytes` per rank (dist.gather: RCCL on GPUs, gloo
    on CPU), and root copies every slab into its place of ONE host array -- page-locked, from the library's
    block cache (`_engine.pinned_empty`), so the D2H copy of slab k overlaps the gather of slab k + 1.
    Root's device footprint next to its own block is (1 + 2 * world) slabs, independent of the job size (C5 fp64:
    41 GB block + 17 x 2 GB).
    -> the (n_channels, n) NumPy array on root, None on the other ranks."""
    import torch
    import torch.distributed as dist
    from . import _engine
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = [channel_block(n_channels, r, world) for r in range(world)]
    width = max(b - a for a, b in rows)
    n = local.shape[1]
    itemsize = local.element_size()
    slab = max(1, min(width, int(slab_bytes // max(1, n * itemsize))))
    np_dtype = {torch.float64: np.float64, torch.float32: np.float32,
                torch.complex128: np.complex128, torch.complex64: np.complex64}[local.dtype]
    host = None
    if rank == root:
        host = _engine.pinned_empty((n_channels, n), np_dtype) if local.is_cuda else np.empty((n_channels, n), np_dtype)
    copy_stream = torch.cuda.Stream() if local.is_cuda else None
    # Device memory on root is BOUNDED: one staging slab + a ring of RING receive sets (world slabs each), whatever
    # the number of iterations -- a set is reused only after the D2H copies that read it have completed (an event
    # per set on the copy stream, waited for by the stream the next gather is ordered on).
    RING = 2
    part = torch.zeros((slab, n), dtype=local.dtype, device=local.device)
    ring = [[torch.empty_like(part) for _ in range(world)] for _ in range(RING)] if rank == root else None
    done = [None] * RING
    for it, r0 in enumerate(range(0, width, slab)):
        mine = local[r0:r0 + slab]
        part[:mine.shape[0]] = mine
        if mine.shape[0] < slab:
            part[mine.shape[0]:].zero_()
        parts = None
        if rank == root:
            parts = ring[it % RING]
            if done[it % RING] is not None:
                torch.cuda.current_stream().wait_event(done[it % RING])
        dist.gather(part, parts, dst=root, group=group)
        if rank == root:
            if copy_stream is not None:
                copy_stream.wait_stream(torch.cuda.current_stream())
            for (a, b), p in zip(rows, parts):
                lo, hi = a + r0, min(b, a + r0 + slab)
                if hi <= lo:
                    continue
                dst = torch.from_numpy(host[lo:hi])
                if copy_stream is not None:
                    with torch.cuda.stream(copy_stream):
                        dst.copy_(p[:hi - lo], non_blocking=True)
                else:
                    dst.copy_(p[:hi - lo])
            if copy_stream is not None:
                done[it % RING] = torch.cuda.Event()
                done[it % RING].record(copy_stream)
    if copy_stream is not None:
        copy_stream.synchronize()
    return host


# ---------------------------------------------------------------------------------------------------
# time-slice-per-rank
# ---------------------------------------------------------------------------------------------------
def fir_halo(K: int) -> tuple[int, int]:
    """Samples a FIR output needs to the (left, right) of its own index: out[i] = sum_k ker[k] sig[i + K//2 - k]
    (predistort(ker=), reference distortion.py:329-337) reads sig[i - (K - 1 - K//2) .. i + K//2]."""
    return K - 1 - K // 2, K // 2


class TimeShardedSampler:
    """The rank-local time slice of a job of a few long rows.  `own` = the rank's output samples [start, stop)
    of every row; the plan covers them plus `halo = (left, right)` neighbouring samples (clipped at the ends of
    the grid), which a following FIR stage consumes and discards."""

    def __init__(self, channels, grid, rank, world, halo=(0, 0), function_lib=None):
        from . import _flatten
        from ._sampling import BatchSampler
        if not isinstance(grid, _flatten.wfk_grid):
            grid = _flatten.grid_from_desc(grid)
        self.full = grid
        self.rank, self.world = rank, world
        self.start, self.stop = channel_block(int(grid.n), rank, world)
        self.lo = max(0, self.start - int(halo[0]))
        self.hi = min(int(grid.n), self.stop + int(halo[1]))
        self.grid = _flatten.grid_slice(grid, self.lo, self.hi)
        self.local = BatchSampler(channels, self.grid, function_lib)
        self.n_channels = self.local.n_channels
        self.n = self.local.n                      # samples the local plan writes per row (halo included)
        self.own = slice(self.start - self.lo, self.stop - self.lo)

    def launch_torch(self, out, accumulate=False):
        """out: (n_channels, >= n) device tensor; columns `own` are the rank's samples"""
        return self.local.launch_torch(out, accumulate)

    def close(self):
        self.local.close()


class TimeShardedFir:
    """predistort(wav(t), ker=ker) cut along time: every rank samples its slice plus the FIR halo (recomputed,
    not exchanged) through the sampler -> FIR chain and keeps the outputs of its own samples.  Equal to the same
    columns of the unsharded chain up to the rounding of the transform's block alignment (~1e-15)."""

    def __init__(self, channels, grid, ker, rank, world, dtype=np.float64, function_lib=None):
        from . import _flatten
        from .distortion import SampledFir
        ker = np.asarray(ker, dtype=np.float64)
        if not isinstance(grid, _flatten.wfk_grid):
            grid = _flatten.grid_from_desc(grid)
        self.start, self.stop = channel_block(int(grid.n), rank, world)
        hl, hr = fir_halo(ker.shape[-1])
        self.lo, self.hi = max(0, self.start - hl), min(int(grid.n), self.stop + hr)
        self.grid = _flatten.grid_slice(grid, self.lo, self.hi)
        self.chain = SampledFir(channels, self.grid, ker, dtype, function_lib)
        self.n, self.n_channels = self.chain.n, self.chain.n_channels
        self.own = slice(self.start - self.lo, self.stop - self.lo)

    def launch_torch(self, out):
        """out: (n_channels, >= n) device tensor of the plan dtype; columns `own` are the rank's filtered samples"""
        return self.chain.launch_torch(out)

    def close(self):
        self.chain.close()


class TimeShardedIir:
    """sample(filters=(sos, initial)) / predistort(filters=) cut along time.  A recurrence cannot be split without
    its state: rank r filters its own samples starting from the final state of rank r - 1 (`zi`), received as a
    (rows, D) message, and passes its own final state on -- the sampling itself runs on all ranks at once, the
    filter passes follow one another in rank order (reference: zi carried from chunk to chunk,
    waveforms/waveform.py:244-251)."""

    def __init__(self, sections, sampler: TimeShardedSampler, dtype=np.float64):
        from . import _engine
        if sampler.lo != sampler.start or sampler.hi != sampler.stop:
            raise ValueError('the IIR stage runs on a slice without halo')
        self.sampler = sampler
        self.plan = _engine.IirPlan(sections, sampler.n, sampler.n_channels, dtype)
        self.D = self.plan.state_dim

    def apply_local(self, x, y, state, initial=0.0):
        """This rank's pass alone: x -> y ((rows, >= n) device tensors, y may be x) from `state` (rows, D) -> the
        final state (rows, D), both float64 device tensors."""
        import torch
        zf = torch.empty_like(state)
        stream = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0
        for attempt in range(2):
            ok = self.plan.apply(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), state.data_ptr(), zf.data_ptr(),
                                 initial, stream)
            if self.plan.status(stream) and ok:
                return zf
            if x.data_ptr() == y.data_ptr():
                break                          # (the input is gone: the caller samples again)
        raise RuntimeError('IIR stage timed out (a look-back that never completed); sample again and re-apply')

    def apply_torch(self, x, y, initial=0.0, zi=None, group=None):
        """x -> y on every rank, in rank order.  Rank 0 starts from `zi` ((D,) or (rows, D); default: rest);
        -> the final state (rows, D) on the last rank, None elsewhere."""
        import torch
        import torch.distributed as dist
        rank, world = self.sampler.rank, self.sampler.world
        rows, D = self.sampler.n_channels, self.D
        state = torch.zeros((rows, max(D, 1)), dtype=torch.float64, device=x.device)
        if rank == 0:
            if zi is not None:
                z = np.broadcast_to(np.asarray(zi, dtype=np.float64), (rows, D)).copy()
                state[:, :D] = torch.as_tensor(z, device=x.device)
        elif world > 1:
            if dist.get_backend(group) == 'gloo':      # (rehearsals: CPU transport)
                h = torch.empty(state.shape, dtype=state.dtype)
                dist.recv(h, src=rank - 1, group=group)
                state.copy_(h)
            else:
                dist.recv(state, src=rank - 1, group=group)
        # A rank that cannot produce its state still hands a message on -- a NaN-poisoned one -- so that the ranks
        # behind it raise as well instead of waiting in recv for ever.
        err = None
        if rank > 0 and bool(torch.isnan(state).any()):
            err = RuntimeError('IIR stage: rank %d received a poisoned state (an earlier rank failed)' % rank)
            zf = state
        else:
            try:
                zf = self.apply_local(x, y, state, initial)
            except RuntimeError as exc:
                err = exc
                zf = torch.full_like(state, float('nan'))
        if rank + 1 < world:
            dist.send(zf.cpu() if dist.get_backend(group) == 'gloo' else zf, dst=rank + 1, group=group)
        if err is not None:
            raise err
        if rank + 1 < world:
            return None
        return zf[:, :D]

    def close(self):
        self.plan.close()


def max_over_ranks(value: float, device='cpu', group=None) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
