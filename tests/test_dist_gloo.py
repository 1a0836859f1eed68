"""N > 1 path on CPU: world_size 2, gloo.  Covers what the multi-GPU bench relies on
apart from the kernels: channel-block partitioning, rank-local flatten/compile (host-
only plans: bit-exact piece indices vs the unsharded program), result placement by
all_gather, and the max-over-ranks timing reduction."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_channel_block_partition():
    from waveforms_amd._dist import channel_block
    for n in (0, 1, 5, 256, 257, 4096):
        for world in (1, 2, 3, 8):
            blocks = [channel_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        channel_block(4, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import waveforms_amd as wf
        from waveforms_amd import _engine, _flatten, workloads as wl
        from waveforms_amd._dist import channel_block, gather_rows, max_over_ranks
        nch, grid = 7, ('linspace', 0.0, 6 * wl.SPAN, 40000, False)
        make = lambda c: wl.sum_channel(wf, 6, 1000 + c)
        a, b = channel_block(nch, rank, world)
        local = _flatten.flatten([make(c) for c in range(a, b)])
        plan = _engine.Plan(local, grid=_flatten.grid_from_desc(grid))      # host-only
        whole = _flatten.flatten([make(c) for c in range(nch)])
        ref = _engine.Plan(whole, grid=_flatten.grid_from_desc(grid))
        for i, c in enumerate(range(a, b)):
            assert np.array_equal(plan.member_index(i), ref.member_index(c))
        # result placement: rank r contributes rows filled with its channel numbers
        rows = torch.arange(a, b, dtype=torch.float64)[:, None].repeat(1, 5)
        full = gather_rows(rows, nch)
        assert full.shape == (nch, 5)
        assert torch.equal(full[:, 0], torch.arange(nch, dtype=torch.float64))
        assert max_over_ranks(1.0 + rank) == float(world)
        # chunked placement on root's host: slabs of 2 rows per rank (7 rows over 2 ranks: blocks of 4 and 3)
        from waveforms_amd._dist import gather_rows_to_host
        host = gather_rows_to_host(rows, nch, root=0, slab_bytes=2 * 5 * 8)
        if rank == 0:
            assert host.shape == (nch, 5) and np.array_equal(host[:, 3], np.arange(nch, dtype=np.float64))
        else:
            assert host is None
        # time-slice-per-rank: rank r compiles samples [a_r, b_r) (+ FIR halo) of the SAME grid as a slice
        # (wfk_grid.i0): piece indices are those of the whole plan, shifted and clipped -- bit for bit
        from waveforms_amd._dist import fir_halo
        g = _flatten.grid_from_desc(grid)
        hl, hr = fir_halo(1024)
        sa, sb = channel_block(int(g.n), rank, world)
        lo, hi = max(0, sa - hl), min(int(g.n), sb + hr)
        sl = _flatten.grid_slice(g, lo, hi)
        assert np.array_equal(_flatten.grid_values(sl), _flatten.grid_values(g)[lo:hi])
        part = _engine.Plan(whole, grid=sl)
        for c in range(nch):
            assert np.array_equal(part.member_index(c), np.clip(ref.member_index(c) - lo, 0, hi - lo))
        q.put((rank, 'ok'))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_grid_slices_are_the_same_samples():
    """wfk_grid.i0: a slice of a linspace / arange grid has the whole grid's times bit for bit (the overridden
    last sample of np.linspace(endpoint=True) included), the library's piece indices and the oracle's samples
    are those of the whole grid."""
    import waveforms_amd as wf
    from oracle import c_oracle
    from waveforms_amd import _engine, _flatten, workloads as wl
    w = wl.sum_channel(wf, 9, 3)
    prog = _flatten.flatten([w])
    for desc in (('linspace', 0.0, 9 * wl.SPAN, 30011, True), ('linspace', -3e-8, 2.9e-7, 9973, False),
                 ('arange', 1e-3, 1e-3 + 2.7e-7, 0.25e-9)):
        g = _flatten.grid_from_desc(desc)
        t = wl.make_grid(desc)
        assert np.array_equal(_flatten.grid_values(g), t)
        whole = c_oracle.eval_grid(prog, g)[0]
        ref = _engine.Plan(prog, grid=g)
        n = int(g.n)
        for lo, hi in ((0, n), (0, n // 3), (n // 3, n // 3 + 1), (n // 3, 2 * n // 3 + 17), (n - 5, n), (7, 7)):
            sl = _flatten.grid_slice(g, lo, hi)
            assert np.array_equal(_flatten.grid_values(sl), t[lo:hi]) and np.array_equal(c_oracle.grid_values(sl), t[lo:hi])
            assert np.array_equal(c_oracle.eval_grid(prog, sl)[0], whole[lo:hi])
            assert np.array_equal(_engine.Plan(prog, grid=sl).member_index(0), np.clip(ref.member_index(0) - lo, 0, hi - lo))
            assert np.array_equal(c_oracle.member_index(prog, 0, grid=sl), np.clip(ref.member_index(0) - lo, 0, hi - lo))
    with pytest.raises(ValueError):
        _flatten.grid_slice(g, 5, n + 1)


def _poison_worker(rank, world, port, q):
    """rank 1 of 3 fails in its own pass: ranks 1 AND 2 must raise (rank 2 receives a NaN-poisoned state instead
    of waiting in recv for ever), rank 0 completes"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from waveforms_amd._dist import TimeShardedIir

        class _Sampler:
            n_channels = 3

        class _Stage(TimeShardedIir):
            def __init__(self):
                self.sampler, self.D = _Sampler(), 2
                self.sampler.rank, self.sampler.world = rank, world

            def apply_local(self, x, y, state, initial=0.0):
                if rank == 1:
                    raise RuntimeError('IIR stage timed out')
                return state + 1.0

        x = torch.zeros((3, 8), dtype=torch.float64)
        try:
            zf = _Stage().apply_torch(x, x)
            q.put((rank, 'done' if zf is None else 'state'))
        except RuntimeError as e:
            q.put((rank, 'raised: ' + str(e)[:40]))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_poisons_the_iir_hand_off_instead_of_hanging_it():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_poison_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == 'done' and res[1].startswith('raised: IIR stage timed out'), res
    assert res[2].startswith('raised: IIR stage: rank 2 received a poisoned'), res
