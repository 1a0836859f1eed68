#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" = one pass of the sampler over one batch of synthetic channels with the
output left resident in HBM (that is where the API leaves it: BatchSampler.launch).
Workloads (SURVEY.md §8(d) / BASELINE.json configs):
  sampler256 (default)  256 channels x 1e7 points, 100 gaussian+DRAG pulses per
                        channel (the C4/C5 channel spec), fp64, grid mode
  c2                    1 channel x 100 pulses x 1e7 points, fp64
  c2_duty30 / c2_drag   C2 variants: 30 % duty cycle / built from the DRAG primitive
  c3                    256 WaveVStack channels x 20 pulses x 1e6 points, fp32
  c4                    sampler256 followed by the 1024-tap FIR stage
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); channels are
independent, so each rank samples its own block of 256 channels with no data-path
collective (weak scaling); time = max over ranks between two barriers.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable


def workload(name, channels, points):
    """-> (make_channel(c), grid, dtype, description); channel c is global (all ranks)."""
    import waveforms_amd as wf
    from waveforms_amd import workloads as wl
    if name in ('sampler256', 'c4'):
        return (lambda c: wl.sum_channel(wf, 100, 1000 + c)), wl.c2_grid(points), np.float64, (
            f'{channels} ch/GPU x {points:.0e} pts, 100 gaussian+DRAG pulses/ch '
            f'(SURVEY 8(d) C4/C5 channel spec), grid mode')
    if name == 'c2':
        return (lambda c: wl.c2_channel(wf)), wl.c2_grid(points), np.float64, (
            f'C2: 1 ch x 100 gaussian+DRAG pulses x {points:.0e} pts')
    if name == 'c2_duty30':
        return (lambda c: wl.c2_channel(wf, True)), wl.c2_grid(points, True), np.float64, (
            f'C2 30 % duty: 1 ch x 100 pulses spaced 100 ns x {points:.0e} pts')
    if name == 'c2_drag':
        return (lambda c: wl.c2_drag_channel(wf)), wl.c2_grid(points), np.float64, (
            f'C2 drag() primitive variant: 1 ch x 100 DRAG (type 13) pulses x {points:.0e} pts')
    if name == 'c3':
        return (lambda c: wl.vstack_channel(wf, 20, 100 + c)), wl.c3_grid(points), np.float32, (
            f'C3: {channels} WaveVStack ch/GPU x 20 pulses x {points:.0e} pts')
    raise SystemExit(f'unknown workload {name}')


def _c_oracle_worker(job):
    """One channel of the headline workload through the plain-C oracle (own process)."""
    seed, grid_desc = job
    import waveforms_amd as wf
    from oracle import c_oracle
    from waveforms_amd import _flatten, workloads as wl
    prog = _flatten.flatten([wl.sum_channel(wf, 100, seed)])
    y = c_oracle.eval_grid(prog, _flatten.grid_from_desc(grid_desc))
    return float(y.sum())


def c_oracle_all_cores(grid_desc, n_points, procs=16):
    """The plain-C oracle on `procs` host cores at once (one channel per process, spawned:
    the parent holds a HIP context and must not be forked)."""
    import multiprocessing as mp
    procs = max(1, min(procs, os.cpu_count() or 1))
    ctx = mp.get_context('spawn')
    with ctx.Pool(procs) as pool:
        pool.map(_c_oracle_worker, [(1000 + c, ('linspace', 0.0, 1e-9, 8, False)) for c in range(procs)])
        t0 = time.perf_counter()                      # workers are up and the library is loaded
        pool.map(_c_oracle_worker, [(1000 + c, grid_desc) for c in range(procs)], chunksize=1)
        dt = time.perf_counter() - t0
    return procs * n_points / dt / 1e6, procs


def cpu_baseline(chans, grid_desc, budget_s=12.0):
    """Reference-like NumPy path (oracle/np_oracle.py: same pass structure as the
    reference's calc_parts/_calc/_fill_parts) timed single-threaded on this box's
    host, on as many channels of the same workload as fit the time budget."""
    from oracle import np_oracle
    from waveforms_amd import workloads as wl
    t = wl.make_grid(grid_desc)
    done, t0, outs = 0, time.perf_counter(), []
    for w in chans:
        y = np_oracle.call(w, t)
        if done < 2:
            outs.append(y)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    base = {'value': done * len(t) / dt / 1e6, 'unit': 'Msamples/s', 'cores': 1,
            'kind': 'port',
            'sample': f'{done} channel(s) x {len(t)} pts of the same workload, '
                      f'NumPy restatement of the reference pass structure, '
                      f'{dt:.1f} s, host has {os.cpu_count()} cores'}
    # second CPU figure: the plain-C scalar oracle on the flattened program (1 thread)
    try:
        from oracle import c_oracle
        from waveforms_amd import _flatten
        prog = _flatten.flatten(chans[:2])
        g = _flatten.grid_from_desc(grid_desc)
        t1 = time.perf_counter()
        c_oracle.eval_grid(prog, g)
        dc = time.perf_counter() - t1
        base['c_oracle_msamples_per_s_1thread'] = 2 * len(t) / dc / 1e6
        # third figure: the same C oracle on 16 host cores at once (the box's CPU share)
        rate, procs = c_oracle_all_cores(grid_desc, len(t))
        base['c_oracle_msamples_per_s_multi'] = rate
        base['c_oracle_multi_procs'] = procs
    except Exception as exc:  # the C oracle is optional for the bench
        base['c_oracle_error'] = repr(exc)
    return base, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='sampler256')
    ap.add_argument('--channels', type=int, default=None)
    ap.add_argument('--points', type=float, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gather-rows', type=int, default=0,
                    help='N > 1 only: after the timed steps, also time an RCCL all_gather of this '
                         'many rows per rank (result placement, SURVEY 8(e); reported under '
                         '"gather", never part of `value`: results stay sharded by default)')
    ap.add_argument('--dtype', choices=['f64', 'f32'], default=None,
                    help='override the output dtype of the workload')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the sampler has no CPU path)')
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from waveforms_amd import _engine
    from waveforms_amd._dist import ShardedSampler
    _engine.set_device(local_rank)

    name = args.workload
    channels = args.channels or (1 if name.startswith('c2') else 256)
    points = int(args.points or (10**6 if name == 'c3' else 10**7))
    make_channel, grid, dtype, desc = workload(name, channels, points)
    if args.dtype:
        dtype = np.float64 if args.dtype == 'f64' else np.float32
    # weak scaling: every rank owns a block of `channels` channels of the global job
    sh = ShardedSampler(channels * world, make_channel, grid, rank, world)
    bs = sh.local
    chans = [make_channel(c) for c in range(min(channels, 32))] if rank == 0 else []
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    out = torch.empty((bs.n_channels, bs.n), dtype=tdt, device='cuda')
    fir = None
    if name == 'c4':
        from waveforms_amd.distortion import FirStage
        from waveforms_amd import workloads as wl
        fir = FirStage(wl.c4_kernel(), bs.n, bs.n_channels, dtype)
        out2 = torch.empty_like(out)

    def step():
        bs.launch_torch(out)
        if fir is not None:
            fir.apply_torch(out, out2)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Untimed device pre-warm, before the W warm-up steps: the first ~40 ms of launches after
    # idle run ~10% slower (memory/fabric clocks still ramping; measured per launch in
    # tools/placement_probe2.py), and W = 3 steps of a 3 ms kernel end well inside that ramp.
    t_pre = time.perf_counter()
    n_pre = 0
    while n_pre < 10 or time.perf_counter() - t_pre < 0.25:
        step()
        n_pre += 1
        if n_pre % 10 == 0:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    ev_fir = []
    for a, b in ev:
        a.record()          # events on torch's current stream == the launch stream
        bs.launch_torch(out)
        b.record()
        if fir is not None:
            fir.apply_torch(out, out2)
            c = torch.cuda.Event(enable_timing=True)
            c.record()
            ev_fir.append((b, c))
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    samples_per_step = bs.n_channels * bs.n * world
    elem = np.dtype(dtype).itemsize
    algo_bytes = bs.n_channels * bs.n * elem            # per launch, per GPU
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tfile):
        traffic = json.load(open(tfile)).get(name)

    line = {
        'metric': 'Msamples/s (all channels)',
        'value': samples_per_step * args.steps / elapsed / 1e6,
        'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64' if dtype == np.float64 else 'f32', 'data': 'synthetic',
        'config': {'workload': f'{name}: {desc}', 'channels_per_gpu': bs.n_channels,
                   'points_per_channel': bs.n, 'output': 'device (HBM) buffer',
                   'parallelism': f'channel-block-per-rank x{world}, no collective'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic, 'kernel': 'wfk_sample',
                     'kernel_ms': kern_ms, 'algorithmic_bytes_per_launch': algo_bytes},
    }
    if fir is not None:
        # the FIR stage dominates this workload: report ITS roofline (16 B/sample: read+write)
        fir_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_fir]))
        fb = 2 * algo_bytes
        line['roofline'] = {'bound': 'hbm', 'achieved': fb / (fir_ms * 1e-3) / 1e9,
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': fb / (fir_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            'traffic': traffic, 'kernel': 'fir_fused', 'kernel_ms': fir_ms,
                            'algorithmic_bytes_per_launch': fb,
                            'sampler_kernel_ms': kern_ms}
    if name == 'sampler256' and world == 1 and not args.no_cpu_baseline:
        # C4 = this workload followed by the 1024-tap FIR stage: time the FIR kernel on the
        # buffer just sampled so that the one default line carries both stages
        from waveforms_amd.distortion import FirStage
        from waveforms_amd import workloads as wl
        fst = FirStage(wl.c4_kernel(), bs.n, bs.n_channels, dtype)
        out2 = torch.empty_like(out)
        for _ in range(2):
            fst.apply_torch(out, out2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fst.apply_torch(out, out2)
        e1.record()
        torch.cuda.synchronize()
        fir_ms = e0.elapsed_time(e1) / 5
        line['also'] = {
            'c4_fir_kernel_ms': fir_ms,
            'c4_fir_frac_of_hbm_peak': 2 * algo_bytes / (fir_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'c4_step_ms': kern_ms + fir_ms,
            'c4_msamples_per_s': bs.n_channels * bs.n / ((kern_ms + fir_ms) * 1e-3) / 1e6,
            'note': 'C4 (BASELINE configs[3]) = sampler256 + 1024-tap FIR (fused LDS-FFT kernel)'}
        del out2
        fst.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, outs = cpu_baseline(chans, grid)
        line['cpu_baseline'] = base
        line['speedup_vs_cpu_baseline'] = line['value'] / base['value']
        got = out[:len(outs)].cpu().numpy().astype(np.float64)
        line['max_abs_err_vs_numpy_ref'] = float(
            max(np.max(np.abs(got[i] - outs[i])) for i in range(len(outs))))
    if dist is not None and args.gather_rows > 0:
        # optional result placement over xGMI, timed on its own (a full 20 GB block per rank is
        # gather-bound by > 10x over the compute): bounded slice, extrapolated
        try:
            rows = min(args.gather_rows, bs.n_channels)
            src = out[:rows].contiguous()
            dst = torch.empty((world * rows, bs.n), dtype=out.dtype, device='cuda')
            dist.all_gather_into_tensor(dst, src)
            fence()
            g0 = time.perf_counter()
            for _ in range(3):
                dist.all_gather_into_tensor(dst, src)
            fence()
            g_ms = (time.perf_counter() - g0) / 3 * 1e3
            nbytes = rows * bs.n * elem
            line['gather'] = {
                'rows_per_rank': rows, 'bytes_per_rank': nbytes, 'ms': g_ms,
                'ingress_GBps_per_rank': (world - 1) * nbytes / (g_ms * 1e-3) / 1e9,
                'full_block_est_ms': g_ms * bs.n_channels / rows,
                'note': 'all_gather (RCCL over xGMI) of a row slice; full block extrapolated linearly'}
            del dst
        except Exception as exc:   # placement is optional: never lose the bench line over it
            line['gather'] = {'error': repr(exc)}
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
