#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" = one pass of the hot path over one batch of synthetic channels with the
output left resident in HBM (that is where the API leaves it: BatchSampler.launch).
Workloads (SURVEY.md §8(d) / BASELINE.json configs):
  sampler256 (default)  256 channels x 1e7 points, 100 gaussian+DRAG pulses per
                        channel (the C4/C5 channel spec), fp64, grid mode
  c2                    1 channel x 100 pulses x 1e7 points, fp64
  c2_duty30 / c2_drag   C2 variants: 30 % duty cycle / built from the DRAG primitive
  c3                    256 WaveVStack channels x 20 pulses x 1e6 points, fp32
  c4                    sampler256 followed by the 1024-tap FIR stage
  c5                    C5 per-rank shape: 512 channels x 1e7 points per GPU (seeds 1000+c
                        over the GLOBAL channel index; 4096 channels at --gpus 8), fp64
  far                   256 channels x 2e6 points at 2 GS/s = a 1 ms sequence of back-to-back 10 us
                        pulses under 250-350 MHz carriers (phases of ~2e6 rad: the grid-rounding
                        regime; every sample carries the correction).  far_sparse: the
                        round-2 shape, 200 ns pulses 10 us apart (97 % zero fill)
  awg / awg_duty30      2048 rows x 1e5 points at 2 GS/s (np.arange grid, what Waveform.sample()
                        uses): mixing(gaussian(20 ns), DRAGScaling) pulses back to back (60 samples
                        per pulse) / 100 ns apart (30 % duty); 16 distinct channels per 2048 rows,
                        every row with its own device tables (short-piece tier)
  awg_c4                the awg rows through predistort(wav(t), ker 1024 taps): the sampler fused into
                        the FIR transform at AWG rates (fir_short)

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); channels are
independent, so each rank samples its own block of channels with no data-path
collective (weak scaling); time = max over ranks between two barriers.  Started either
by a launcher that sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun, the driver), or by
`python bench.py --gpus N` itself: the parent then starts N rank processes BEFORE anything
touches the GPU (it never imports torch), waits for them and relays rank 0's JSON line.

Rehearsal switches (no effect on the default line):
  --backend gloo   rendezvous/reductions over gloo instead of RCCL
  --share-gpu      all ranks use device 0 (one-GPU box; needs --backend gloo)
  --plan-only      no GPU at all: every rank flattens + compiles its channel block
                   (host-only plans) and the line reports the sharding (CPU tests)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='sampler256')
    ap.add_argument('--channels', type=int, default=None, help='channels per GPU')
    ap.add_argument('--points', type=float, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-also', action='store_true',
                    help='skip the secondary workloads (c2, c3, c4) of the default line')
    ap.add_argument('--gather-rows', type=int, default=0,
                    help='N > 1 only: after the timed steps, also time an RCCL all_gather of this '
                         'many rows per rank (result placement, SURVEY 8(e); reported under '
                         '"gather", never part of `value`: results stay sharded by default)')
    ap.add_argument('--dtype', choices=['f64', 'f32'], default=None,
                    help='override the output dtype of the workload')
    ap.add_argument('--shard', choices=['channels', 'time'], default='channels',
                    help='N > 1: channel-block-per-rank (default: every rank owns --channels channels, weak scaling) or '
                         'time-slice-per-rank (the SAME --channels rows of --points samples cut along time: strong scaling; '
                         'a job of a few very long rows, SURVEY 8(e))')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl')
    ap.add_argument('--share-gpu', action='store_true')
    ap.add_argument('--plan-only', action='store_true')
    ap.add_argument('--deadline', type=float, default=1500.0,
                    help='self-launched ranks (--gpus N without a rendezvous in the environment): seconds '
                         'after which the parent stops every rank and returns non-zero')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a rendezvous in the environment
# ---------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Start one rank process per GPU and relay rank 0's line.  Runs before any GPU call
    and never imports torch: a process that has initialised the GPU must neither fork
    workers that use it nor be replaced by exec.

    The parent polls ALL children: the first one that exits non-zero (or the overall
    deadline, --deadline seconds) ends the job -- the remaining children, exactly the ones
    started here, are terminated, then killed, and the parent returns non-zero.  Without
    that a rank dying during rendezvous leaves rank 0 blocked in init_process_group until the
    NCCL timeout, and the driver sees a hang instead of an error."""
    import socket
    import tempfile
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (RCCL needs it here)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    deadline = time.time() + args.deadline
    rcs = [None] * len(procs)
    failed = None
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0) and failed is None:
                    failed = f'rank {i} exited with code {rcs[i]}'
        if failed is None and time.time() > deadline:
            failed = f'deadline of {args.deadline:.0f} s passed'
        if failed is not None:
            break
        time.sleep(0.05)
    if failed is not None:
        sys.stderr.write(f'bench.py launcher: {failed}; stopping the other ranks\n')
        for i, p in enumerate(procs):          # the exact children started above
            if rcs[i] is None:
                p.terminate()
        t_kill = time.time() + 5.0
        for i, p in enumerate(procs):
            if rcs[i] is None:
                try:
                    rcs[i] = p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    rcs[i] = p.wait()
    # rank 0's JSON line goes to stdout; anything else it printed there (backend chatter) to stderr
    out0.seek(0)
    for ln in out0.read().decode('utf-8', 'replace').splitlines():
        (sys.stdout if ln.startswith('{') and failed is None else sys.stderr).write(ln + '\n')
    sys.stdout.flush()
    if failed is not None:
        return next((rc for rc in rcs if rc not in (None, 0)), 1) or 1
    return 0


# ---------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------
def far_channel(ns, c, nseg=100, spacing=10e-6, width=200e-9):
    """One channel of a 1 ms sequence: `nseg` gaussian+DRAG pulses 10 us apart, carriers of
    +-(250..350) MHz: phases up to 2 pi * 350e6 * 1e-3 = 2.2e6 rad (routine T1/echo shapes).
    width = spacing / 1.5 makes the pulses contiguous (gaussian(w) lives on +-0.75 w)."""
    import numpy as np
    rng = np.random.default_rng(5000 + c)
    ws = []
    for k in range(nseg):
        A = rng.uniform(0.1, 1)
        f = rng.uniform(250e6, 350e6) * (1 if rng.uniform() < 0.5 else -1)
        phi = rng.uniform(0, 2 * np.pi)
        I, _ = ns.mixing(A * ns.gaussian(width) >> ((k + 0.5) * spacing), freq=f, phase=phi,
                         DRAGScaling=1e-10)
        ws.append(I)
    while len(ws) > 1:
        nxt = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)]
        if len(ws) % 2:
            nxt.append(ws[-1])
        ws = nxt
    return ws[0]


def workload(name, channels, points):
    """-> (make_channel(c), grid, dtype, description); channel c is global (all ranks)."""
    import numpy as np
    import waveforms_amd as wf
    from waveforms_amd import workloads as wl
    if name in ('sampler256', 'c4', 'c5', 'iir_chain', 'iir_chain4'):
        return (lambda c: wl.sum_channel(wf, 100, 1000 + c)), wl.c2_grid(points), np.float64, (
            f'{channels} ch/GPU x {points:.0e} pts, 100 gaussian+DRAG pulses/ch, seeds '
            f'1000+c (SURVEY 8(d) C4/C5 channel spec), grid mode' +
            ('; then sosfilt with two biquads (butter(4, 0.1)): Waveform.sample(filters=), sampler inside the IIR scan' if name == 'iir_chain' else
             '; then four first-order exponential-correction sections: predistort(wav(t), filters), sampler inside the IIR scan'
             if name == 'iir_chain4' else ''))
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
    if name in ('far', 'far_sparse'):
        width = 200e-9 if name == 'far_sparse' else 10e-6 / 1.5
        return (lambda c: far_channel(wf, c, width=width)), ('linspace', 0.0, points / 2e9, points, False), \
            np.float64, (f'{channels} ch/GPU x {points:.0e} pts at 2 GS/s, 100 pulses/ch '
                         f'({"200 ns wide, 10 us apart" if name == "far_sparse" else "10 us each, back to back"}) with '
                         f'250-350 MHz carriers out to t = {points / 2e9 * 1e3:.1f} ms')
    if name in ('awg', 'awg_duty30', 'awg_c4'):
        d30 = name == 'awg_duty30'
        return (lambda c: wl.awg_channel(wf, c, points, 2e9, d30)), wl.awg_grid(points, 2e9), np.float64, (
            f'{channels} rows/GPU x {points:.0e} pts at 2 GS/s (np.arange grid of Waveform.sample), '
            f'mixing(gaussian(20 ns), DRAGScaling) pulses ' + ('100 ns apart (30 % duty)' if d30 else 'back to back') +
            f', 60 samples per pulse; {channels // TILE[name]} distinct channels x {TILE[name]} copies, every row with '
            f'its own device tables')
    if name == 'awg_interp':
        return (lambda c: wl.awg_interp_channel(wf, c, points, 2e9)), wl.awg_grid(points, 2e9), np.float64, (
            f'{channels} rows/GPU x {points:.0e} pts at 2 GS/s: samplingPoints envelopes (301 knots, 8 shapes per channel) under '
            f'carriers, back to back, 60 samples per pulse; {channels // TILE[name]} distinct channels x {TILE[name]} copies')
    if name == 'multitone':
        return (lambda c: wl.multitone_channel(wf, c)), wl.c2_grid(points), np.float64, (
            f'{channels} rows/GPU ({channels // TILE[name]} distinct x {TILE[name]}) x {points:.0e} pts, 100 gaussian pulses per row, 10 tones under every pulse')
    if name.startswith('direct_'):
        shape = name[len('direct_'):]
        return (lambda c: wl.direct_channel(wf, shape, c)), ('linspace', 0.0, wl.DIRECT_T, points, False), np.float64, (
            f'{channels} rows/GPU ({channels // TILE[name]} distinct x {TILE[name]}) x {points:.0e} pts over 3 us: {shape} pulses (direct tier)')
    if name == 'tlist':
        # (the time axis itself is made in run_rank: an explicit array, not a grid descriptor)
        return (lambda c: wl.sum_channel(wf, 100, 1000 + c)), ('tlist', points), np.float64, (
            f'{channels} ch/GPU x {points:.0e} explicit non-uniform times (C2 grid jittered by 0.3 dt, sorted), 100 gaussian+DRAG '
            f'pulses/ch: Waveform.__call__(x) on a non-grid x; 16 B/sample (t is read)')
    raise SystemExit(f'unknown workload {name}')


TILE = {'awg': 128, 'awg_duty30': 128, 'awg_c4': 128, 'awg_interp': 128, 'multitone': 8, 'direct_sinc': 8, 'direct_mollifier': 8, 'direct_interp': 8}     # rows = TILE copies of rows / TILE distinct channels


def iir_shapes():
    """the two IIR cascades of the bench: a two-biquad sosfilt cascade (Waveform.sample(filters=)) and four first-order
    exponential-correction sections (predistort(filters=))"""
    import numpy as np
    from scipy.signal import butter
    return {'two_biquads': [(s_[:3], s_[3:]) for s_ in butter(4, 0.1, output='sos')],
            'four_first_order': [(np.array([1.02, -np.exp(-1 / t_) * 1.01]), np.array([1.0, -np.exp(-1 / t_)]))
                                 for t_ in (50.0, 400.0, 3000.0, 20000.0)]}


def default_shape(name):
    channels = {'c2': 1, 'c2_duty30': 1, 'c2_drag': 1, 'c5': 512, 'awg': 2048, 'awg_duty30': 2048, 'awg_c4': 2048, 'awg_interp': 2048, 'tlist': 64,
                'multitone': 64, 'direct_sinc': 64, 'direct_mollifier': 64, 'direct_interp': 64}.get(name, 256)
    points = {'c3': 10**6, 'far': 2 * 10**6, 'far_sparse': 2 * 10**6, 'awg': 10**5, 'awg_duty30': 10**5, 'awg_c4': 10**5, 'awg_interp': 10**5,
              'tlist': 2 * 10**6}.get(name, 10**7)
    return channels, points


def _c_oracle_worker(job):
    """One channel of the headline workload through the plain-C oracle (own process)."""
    seed, grid_desc = job
    import waveforms_amd as wf
    from oracle import c_oracle
    from waveforms_amd import _flatten, workloads as wl
    prog = _flatten.flatten([wl.sum_channel(wf, 100, seed)])
    y = c_oracle.eval_grid(prog, _flatten.grid_from_desc(grid_desc))
    return float(y.sum())


def host_cpu_share():
    """-> (cores this process may use, cores of the host).  A GPU box hands each GPU a share of the host
    (cgroup quota and / or affinity mask): os.cpu_count() is the host, not the share."""
    total = os.cpu_count() or 1
    share = total
    try:
        share = min(share, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    share = min(share, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    share = min(share, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, share), total


def c_oracle_all_cores(grid_desc, n_points, procs=None):
    """The plain-C oracle on every core this process may use (one channel per process, spawned: the parent
    holds a HIP context and must not be forked).  Bounded at 64 workers: one 1e7-point channel each is
    already 5 GB of outputs in flight."""
    import multiprocessing as mp
    share, _total = host_cpu_share()
    procs = max(1, min(procs or share, share, 64))
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
                      f'{dt:.1f} s, 1 thread; this job may use {host_cpu_share()[0]} of the host\'s {os.cpu_count()} cores'}
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
        # third figure: the same C oracle on every core this process may use (the box's CPU share)
        rate, procs = c_oracle_all_cores(grid_desc, len(t))
        share, total = host_cpu_share()
        base['c_oracle_msamples_per_s_multi'] = rate
        base['c_oracle_multi_procs'] = procs
        base['c_oracle_multi_cores'] = f'{procs} processes = {procs} of the {share} cores this job may use; the host has {total}'
    except Exception as exc:  # the C oracle is optional for the bench
        base['c_oracle_error'] = repr(exc)
    return base, outs


def profile_traffic(key, algo_bytes=None):
    """HBM bytes per launch of this workload's dominant kernel from the committed PMC passes
    (profiles/traffic.json: rocprofv3 --pmc of this same command; counters cannot be read
    from inside the run) -> (bytes | None, source label).  The counters belong to the shape that
    was profiled (`algo_bytes` in the entry): a run of another shape (--channels / --points /
    --dtype) gets null, not a number that is not its own."""
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(tfile):
        return None, None
    t = json.load(open(tfile))
    v = t.get(key)
    if not isinstance(v, dict):
        return None, None
    src = 'profiles/' + v.get('source', 'traffic.json')
    prof = v.get('algo_bytes')
    if algo_bytes is not None and prof is not None and abs(prof - algo_bytes) > 1e-6 * prof:
        return None, (f'{src} was collected on another shape ({prof:.4g} algorithmic bytes per launch, this run '
                      f'{algo_bytes:.4g}): no traffic figure for this run')
    return v.get('bytes'), src + ' (separate rocprofv3 --pmc pass of this shape, not this run)'


RATED_CLOCK_GHZ = 2.4   # MI355X peak engine clock (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2.4 GHz = 78.6 TF


def valu_roofline(key, kernel_ms, samples):
    """Second roofline for kernels bound by fp64 VALU issue, not by HBM (the FIR transform of C4):
    VALU wave instructions per launch from the committed PMC pass x 4 cycles (a wave64 fp64
    instruction occupies its SIMD for 4 cycles) / 1024 SIMDs / the part's RATED clock (2.4 GHz: the
    78.6 TF fp64 vector peak) = the time the kernel needs at 100 % VALU issue; valu_frac = that / the
    kernel time measured in THIS run.  The clock the PMC pass itself sampled (GRBM_GUI_ACTIVE / time;
    the part runs below its rated clock at its power cap) is reported beside it, not used."""
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(tfile):
        return None
    v = json.load(open(tfile)).get(key)
    if not isinstance(v, dict) or 'valu_wave_instr' not in v:
        return None
    prof = v.get('algo_samples')
    if prof is not None and prof != samples:
        return None          # counters of another shape
    issue_ms = v['valu_wave_instr'] * 4 / 1024 / (RATED_CLOCK_GHZ * 1e9) * 1e3
    out = {'bound': 'valu_fp64', 'valu_instr_per_sample': v['valu_wave_instr'] * 64 / samples,
           'issue_limited_ms': issue_ms, 'valu_frac': issue_ms / kernel_ms,
           'clock_ghz': RATED_CLOCK_GHZ, 'clock': 'rated peak engine clock', 'simds': 1024, 'cycles_per_wave_instr': 4,
           'source': 'profiles/' + v.get('source', 'traffic.json') + ' (separate rocprofv3 --pmc pass: SQ_INSTS_VALU, GRBM_GUI_ACTIVE)'}
    if 'shader_clock_ghz' in v:
        sc = v['shader_clock_ghz']
        out['sampled_clock_ghz'] = sc
        out['valu_frac_at_sampled_clock'] = v['valu_wave_instr'] * 4 / 1024 / (sc * 1e9) * 1e3 / kernel_ms
    return out


# ---------------------------------------------------------------------------------------
# plan-only rehearsal (CPU): sharding, rendezvous, reductions -- no kernels
# ---------------------------------------------------------------------------------------
def run_plan_only(args, rank, world):
    import numpy as np
    import torch
    import torch.distributed as dist
    from waveforms_amd import _engine, _flatten
    from waveforms_amd._dist import channel_block
    # launcher rehearsals (tests/test_bench_launcher.py), plan-only runs alone: a rank that dies before the
    # rendezvous, and one that never comes back (the parent's --deadline has to end it)
    if os.environ.get('WFK_BENCH_FAIL_RANK') == str(rank):
        raise SystemExit(7)
    if os.environ.get('WFK_BENCH_HANG_RANK') == str(rank):
        time.sleep(120)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    name = args.workload
    dch, dpts = default_shape(name)
    channels, points = args.channels or dch, int(args.points or dpts)
    make_channel, grid, dtype, desc = workload(name, channels, points)
    a, b = channel_block(channels * world, rank, world)
    tile = TILE.get(name, 1)
    t0 = time.perf_counter()
    prog = _flatten.tile_program(_flatten.flatten([make_channel(c) for c in range(a // tile, a // tile + (b - a) // tile)]), tile)
    plan = _engine.Plan(prog, grid=_flatten.grid_from_desc(grid))
    dt = time.perf_counter() - t0
    info = [float(a), float(b), float(plan.info.n_pieces), float(plan.info.n_fused), dt]
    if world > 1:
        t = torch.tensor(info, dtype=torch.float64)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        infos = [p.tolist() for p in parts]
    else:
        infos = [info]
    if rank == 0:
        print(json.dumps({
            'metric': 'plan-only rehearsal (no kernels run)', 'value': None, 'unit': 'Msamples/s',
            'n_gpus': world, 'plan_only': True, 'backend': 'gloo' if world > 1 else None,
            'config': {'workload': f'{name}: {desc}', 'channels_per_gpu': channels,
                       'points_per_channel': points},
            'kernel': plan.kernel_name(dtype),
            'ranks': [{'rank': r, 'channels': [int(i[0]), int(i[1])], 'pieces': int(i[2]),
                       'fused_terms': int(i[3]), 'plan_s': i[4]} for r, i in enumerate(infos)]}))
    if world > 1:
        dist.destroy_process_group()
    return 0


class TlistBlock:
    """The rank-local channel block of a time-list job: the same interface as _dist.ShardedSampler /
    BatchSampler for the pieces run_rank uses (n, n_channels, plan, launch_torch)."""

    def __init__(self, n_channels, make_channel, t, rank, world):
        from waveforms_amd import _engine, _flatten
        from waveforms_amd._dist import channel_block
        self.start, self.stop = channel_block(n_channels, rank, world)
        self.plan = _engine.Plan(_flatten.flatten([make_channel(c) for c in range(self.start, self.stop)]), t=t)
        self.n, self.n_channels, self.local = self.plan.n, self.plan.n_channels, self

    def launch_torch(self, out, accumulate=False):
        import torch
        from waveforms_amd import _engine
        self.plan.launch(out.data_ptr(), out.stride(0), _engine.OUT_F64 if out.dtype == torch.float64 else _engine.OUT_F32,
                         accumulate, torch.cuda.current_stream(out.device).cuda_stream)
        return out


# ---------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.plan_only:
        return run_plan_only(args, rank, world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the sampler has no CPU path)')
    if args.share_gpu:
        if args.backend != 'gloo' and world > 1:
            raise SystemExit('--share-gpu needs --backend gloo (RCCL wants one device per rank)')
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')

    from waveforms_amd import _engine, workloads as wl
    from waveforms_amd._dist import ShardedSampler
    from waveforms_amd._sampling import BatchSampler
    _engine.set_device(local_rank)

    name = args.workload
    dch, dpts = default_shape(name)
    channels = args.channels or dch
    points = int(args.points or dpts)
    make_channel, grid, dtype, desc = workload(name, channels, points)
    if args.dtype:
        dtype = np.float64 if args.dtype == 'f64' else np.float32
    # weak scaling: every rank owns a block of `channels` channels of the global job
    tile = TILE.get(name, 1)
    if channels % tile:
        raise SystemExit(f'--channels must be a multiple of {tile} for workload {name}')
    time_sharded = args.shard == 'time' and name not in ('tlist', 'c4', 'awg_c4', 'iir_chain', 'iir_chain4')
    if time_sharded:
        from waveforms_amd._dist import TimeShardedSampler
        if channels % tile:
            raise SystemExit(f'--channels must be a multiple of {tile} for workload {name}')
        sh = TimeShardedSampler([make_channel(c) for c in range(channels // tile)] * tile, grid, rank, world)
    elif name == 'tlist':
        sh = TlistBlock(channels * world, make_channel, wl.jittered_times(points), rank, world)
    else:
        sh = ShardedSampler(channels * world, make_channel, grid, rank, world, tile=tile)
    bs = sh.local
    chans = [make_channel(c) for c in range(min(channels // tile, 32))] if rank == 0 else []
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    out = torch.empty((bs.n_channels, bs.n), dtype=tdt, device='cuda')
    fir = chain = None
    if name in ('c4', 'awg_c4'):
        # C4 = predistort(wav(t), ker): ONE kernel when the plan is fully fused (the FIR workgroups
        # evaluate their own input windows), else sampler -> FIR
        from waveforms_amd.distortion import FirStage, SampledFir
        a0, b0 = sh.start, sh.stop
        chain = SampledFir([make_channel(c) for c in range(a0 // tile, b0 // tile)], grid, wl.c4_kernel(), dtype, tile=tile)
        if not chain.fused:
            fir = FirStage(wl.c4_kernel(), bs.n, bs.n_channels, dtype)
            out2 = torch.empty_like(out)
    if name in ('iir_chain', 'iir_chain4'):
        # sample(filters=) / predistort(wav(t), filters): the wave that owns a chunk of the IIR scan evaluates its input
        from waveforms_amd.distortion import SampledIir
        a0, b0 = sh.start, sh.stop
        chain = SampledIir([make_channel(c) for c in range(a0 // tile, b0 // tile)], grid,
                           iir_shapes()['two_biquads' if name == 'iir_chain' else 'four_first_order'], None, dtype, tile=tile)
        if not chain.fused:
            raise SystemExit('iir_chain: the chain did not fuse: ' + chain.why_not)

    def step():
        if chain is not None and chain.fused:
            chain.launch_torch(out)
            return
        bs.launch_torch(out)
        if fir is not None:
            fir.apply_torch(out, out2)

    def allreduce_max(x):
        if dist is None:
            return x
        dev = 'cuda' if args.backend == 'nccl' else 'cpu'
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def prewarm(fn, min_s=0.25):
        # Untimed device pre-warm, before the W warm-up steps: the first ~40 ms of launches
        # after idle run ~10% slower (memory/fabric clocks still ramping; measured per launch in
        # tools/attic/placement_probe2.py), and W = 3 steps of a 3 ms kernel end well inside that ramp.
        t_pre, n_pre = time.perf_counter(), 0
        while n_pre < 10 or time.perf_counter() - t_pre < min_s:
            fn()
            n_pre += 1
            if n_pre % 10 == 0:
                torch.cuda.synchronize()

    prewarm(step)
    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    ev_fir = []
    for a, b in ev:
        a.record()          # events on torch's current stream == the launch stream
        if chain is not None and chain.fused:
            chain.launch_torch(out)
        else:
            bs.launch_torch(out)
        b.record()
        if fir is not None:
            fir.apply_torch(out, out2)
            c = torch.cuda.Event(enable_timing=True)
            c.record()
            ev_fir.append((b, c))
    fence()
    elapsed = allreduce_max(time.perf_counter() - t0)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    samples_per_step = bs.n_channels * bs.n * world
    if time_sharded:          # the ranks' slices differ by at most one sample: the job is channels x points
        samples_per_step = bs.n_channels * points
    elem = np.dtype(dtype).itemsize
    algo_bytes = bs.n_channels * bs.n * (elem + (8 if name == 'tlist' else 0))   # per launch, per GPU (a time list is read: + 8 B/sample)
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    traffic, traffic_src = profile_traffic(name, algo_bytes)

    table_bytes = bs.plan.table_bytes()
    roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic, 'traffic_source': traffic_src,
            'table_bytes_per_launch': table_bytes,
            'frac_incl_tables': (algo_bytes + table_bytes) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'kernel': bs.plan.kernel_name(dtype),
            'kernel_ms': kern_ms, 'algorithmic_bytes_per_launch': algo_bytes,
            'timing': 'HIP events around every launch on the launch stream, mean over the timed steps'}
    if chain is not None and chain.fused:
        # fused chain: the samples never touch HBM; algorithmic traffic = the filtered output only
        roof['kernel'] = chain.plan.kernel_name()
        roof['table_bytes_per_launch'] = chain.plan.table_bytes()
        roof['frac_incl_tables'] = (algo_bytes + roof['table_bytes_per_launch']) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof['note'] = ('sampler fused into the %s: algorithmic bytes = B_out per sample '
                        '(SURVEY 8(d) "fused sampler->FIR: B_out only"); the kernel is bound by fp64 VALU issue, '
                        'not by HBM: see roofline_valu' % ('IIR scan' if name.startswith('iir_chain') else 'FIR transform'))
        rv = valu_roofline(name, kern_ms, bs.n_channels * bs.n)
        if rv is not None:
            roof['roofline_valu'] = rv
    if fir is not None:
        # the FIR stage dominates this workload: report ITS roofline (16 B/sample: read+write)
        fir_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_fir]))
        fb = 2 * algo_bytes
        roof = {'bound': 'hbm', 'achieved': fb / (fir_ms * 1e-3) / 1e9,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': fb / (fir_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'traffic': traffic, 'traffic_source': traffic_src,
                'kernel': 'fir_fused<%s>' % ('double' if dtype == np.float64 else 'float'),
                'kernel_ms': fir_ms, 'algorithmic_bytes_per_launch': fb,
                'sampler_kernel_ms': kern_ms, 'sampler_kernel': bs.plan.kernel_name(dtype)}
    if dist is not None:
        # per-rank roofline of the dominant kernel (every rank times its own launches)
        mine = torch.tensor([roof['kernel_ms']], dtype=torch.float64,
                            device='cuda' if args.backend == 'nccl' else 'cpu')
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per = [float(p.item()) for p in parts]
        ab = roof['algorithmic_bytes_per_launch']
        roof['per_rank'] = [{'rank': r, 'kernel_ms': ms, 'frac': ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                            for r, ms in enumerate(per)]
        worst = max(per)                      # the line's frac is the slowest rank's
        roof['kernel_ms'] = worst
        roof['achieved'] = ab / (worst * 1e-3) / 1e9
        roof['frac'] = roof['achieved'] / HBM_PEAK_GBS

    line = {
        'metric': 'Msamples/s (all channels)',
        'value': samples_per_step * args.steps / elapsed / 1e6,
        'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong' if time_sharded else 'weak', 'vs_baseline': None,
        'dtype': 'f64' if dtype == np.float64 else 'f32', 'data': 'synthetic',
        'config': {'workload': f'{name}: {desc}', 'channels_per_gpu': bs.n_channels,
                   'channels_total': bs.n_channels * (1 if time_sharded else world),
                   'points_per_channel': bs.n, 'output': 'device (HBM) buffer',
                   'parallelism': (f'time-slice-per-rank x{world} (wfk_grid.i0), no data-path collective' if time_sharded
                                   else f'channel-block-per-rank x{world}, no data-path collective'),
                   'backend': None if dist is None else
                   ('nccl (RCCL) world_size=%d' % dist.get_world_size() if args.backend == 'nccl'
                    else 'gloo world_size=%d' % dist.get_world_size())},
        'roofline': roof,
    }

    def timed(fn, steps, warm=3):
        """mean ms per call of fn, HIP events on the launch stream"""
        prewarm(fn, 0.05)
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    if name == 'sampler256' and world == 1 and not args.no_also:
        also = {}
        # C4 = this workload followed by the 1024-tap FIR stage: time the FIR kernel on the
        # buffer just sampled so that the one default line carries both stages
        from waveforms_amd.distortion import FirStage
        fst = FirStage(wl.c4_kernel(), bs.n, bs.n_channels, dtype)
        out2 = torch.empty_like(out)
        fir_ms = timed(lambda: fst.apply_torch(out, out2), 5, 2)
        from waveforms_amd.distortion import SampledFir
        chn = SampledFir([make_channel(c) for c in range(sh.start, sh.stop)], grid, wl.c4_kernel(), dtype)
        chain_ms = timed(lambda: chn.launch_torch(out2), 5, 2)
        also['c4'] = {
            'kernel': 'fir_sampled<double,12>' if chn.fused else 'wfk_sample_lean + fir_fused',
            'fused': chn.fused, 'step_ms': chain_ms,
            'msamples_per_s': bs.n_channels * bs.n / (chain_ms * 1e-3) / 1e6,
            'algorithmic_bytes_per_launch': algo_bytes,
            'frac': algo_bytes / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'unfused': {'fir_kernel': 'fir_fused<double>', 'fir_kernel_ms': fir_ms,
                        'fir_frac_16B_per_sample': 2 * algo_bytes / (fir_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        'step_ms': kern_ms + fir_ms},
            'note': 'C4 (BASELINE configs[3]) = predistort(wav(t), ker 1024 taps): sampler fused into the '
                    'LDS-FFT FIR kernel; frac = 8 B/sample (output only) / kernel time / 8 TB/s; the roof that '
                    'binds is fp64 VALU issue (roofline_valu)'}
        rv = valu_roofline('c4', chain_ms, bs.n_channels * bs.n) if chn.fused else None
        if rv is not None:
            also['c4']['roofline_valu'] = rv
        chn.close()
        fst.close()
        if dtype == np.float64:      # the same chain into a float buffer (4 B/sample)
            chn32 = SampledFir([make_channel(c) for c in range(sh.start, sh.stop)], grid, wl.c4_kernel(), np.float32)
            o32c = torch.empty((bs.n_channels, bs.n), device='cuda', dtype=torch.float32)
            ms32 = timed(lambda: chn32.launch_torch(o32c), 5, 2)
            also['c4']['f32'] = {'kernel': chn32.plan.kernel_name(), 'fused': chn32.fused, 'step_ms': ms32,
                                 'frac': algo_bytes / 2 / (ms32 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            chn32.close()
            del o32c
        # IIR stages of sample(filters=) / predistort(filters=) on the same 256 x 1e7 block (SURVEY 8(f)
        # N1): a two-biquad sosfilt cascade and four first-order (exponential-correction) sections;
        # 16 B/sample algorithmic (read + write), kernel time by HIP events
        try:
            from waveforms_amd import _engine
            stream = torch.cuda.current_stream().cuda_stream
            iir = {}
            shapes = iir_shapes()
            for sname, secs in shapes.items():
                ip = _engine.IirPlan(secs, bs.n, bs.n_channels, dtype)
                zi = torch.zeros((bs.n_channels, ip.state_dim), dtype=torch.float64, device='cuda')
                zf = torch.empty_like(zi)
                ms = timed(lambda: ip.apply(out.data_ptr(), bs.n, out2.data_ptr(), bs.n, zi.data_ptr(),
                                            zf.data_ptr(), 0.0, stream), 5, 2)
                iir[sname] = {'ms': ms, 'msamples_per_s': bs.n_channels * bs.n / (ms * 1e-3) / 1e6,
                              'frac_16B_per_sample': 2 * algo_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                ip.close()
            iir['note'] = 'iir_onepass (single-pass chained scan, x read once); WFK_IIR_ONEPASS=0: three launches'
            also['iir'] = iir
        except Exception as e:      # (the stage is outside the headline path: report, do not fail the line)
            also['iir'] = {'error': repr(e)}
        # the same cascades with the sampler INSIDE the scan (wfk_chain_iir_*): sample(filters=) / predistort(wav(t),
        # filters) device-resident; 8 B/sample algorithmic (the filtered output only)
        try:
            from waveforms_amd.distortion import SampledIir
            ic = {}
            for sname, secs in iir_shapes().items():
                sc = SampledIir([make_channel(c) for c in range(sh.start, sh.stop)], grid, secs, None, dtype)
                ms = timed(lambda: sc.launch_torch(out2), 5, 2)
                ic[sname] = {'kernel': sc.plan.kernel_name(), 'fused': sc.fused, 'ms': ms,
                             'msamples_per_s': bs.n_channels * bs.n / (ms * 1e-3) / 1e6,
                             'frac': algo_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'unfused_ms': kern_ms + also['iir'].get(sname, {}).get('ms', float('nan'))}
                rvi = valu_roofline('iir_chain' if sname == 'two_biquads' else 'iir_chain4', ms, bs.n_channels * bs.n) if sc.fused else None
                if rvi is not None:
                    ic[sname]['valu_frac'] = rvi['valu_frac']
                    ic[sname]['valu_instr_per_sample'] = rvi['valu_instr_per_sample']
                sc.close()
            also['iir_chain'] = ic
        except Exception as e:
            also['iir_chain'] = {'error': repr(e)}
        del out2
        # the same 256 x 1e7 plan launched into a float buffer (4 B/sample)
        if dtype == np.float64:
            o32 = torch.empty((bs.n_channels, bs.n), device='cuda', dtype=torch.float32)
            ms = timed(lambda: bs.launch_torch(o32), 10, 3)
            also['f32'] = {'workload': 'sampler256, float output', 'kernel': bs.plan.kernel_name(np.float32),
                           'kernel_ms': ms, 'msamples_per_s': bs.n_channels * bs.n / (ms * 1e-3) / 1e6,
                           'algorithmic_bytes_per_launch': algo_bytes // 2,
                           'frac': algo_bytes / 2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'dtype': 'f32'}
            del o32
        # BASELINE configs[1] and [2], and the far-from-origin sequence, in the same line (kernel time
        # by HIP events, frac of 8 TB/s)
        for wname in ('c2', 'c3', 'far', 'awg', 'awg_duty30'):
            wch, wpts = default_shape(wname)
            mk, g, dt_, d_ = workload(wname, wch, wpts)
            wt = TILE.get(wname, 1)
            b2 = BatchSampler([mk(c) for c in range(wch // wt)], g, tile=wt)
            o2 = torch.empty((b2.n_channels, b2.n), device='cuda',
                             dtype=torch.float64 if dt_ == np.float64 else torch.float32)
            ms = timed(lambda: b2.launch_torch(o2), 200 if wname == 'c2' else 50, 10)
            nbytes = b2.n_channels * b2.n * np.dtype(dt_).itemsize
            also[wname] = {'workload': d_, 'kernel': b2.plan.kernel_name(dt_), 'kernel_ms': ms,
                           'msamples_per_s': b2.n_channels * b2.n / (ms * 1e-3) / 1e6,
                           'algorithmic_bytes_per_launch': nbytes,
                           'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           'dtype': 'f64' if dt_ == np.float64 else 'f32'}
            if wname.startswith('awg'):
                # at 60 samples per piece the tables are real traffic: report them, and the float launch
                tb = b2.plan.table_bytes()
                also[wname]['table_bytes_per_launch'] = tb
                also[wname]['frac_incl_tables'] = (nbytes + tb) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                o3 = torch.empty((b2.n_channels, b2.n), device='cuda', dtype=torch.float32)
                ms3 = timed(lambda: b2.launch_torch(o3), 50, 10)
                also[wname]['f32'] = {'kernel': b2.plan.kernel_name(np.float32), 'kernel_ms': ms3,
                                      'msamples_per_s': b2.n_channels * b2.n / (ms3 * 1e-3) / 1e6,
                                      'frac': nbytes / 2 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS}
                del o3
            if wname == 'awg':
                # A FRESH sequence: what the host pays before the first launch.  plan_build_ms: flatten + plan creation
                # of ONE new channel (the drop-in call's first wav.sample()); awg_fresh: 2048 rows with distinct device
                # records -- the 16 trees' program tiled 128x with per-row amplitudes, so the compiler sees 2048
                # different rows (building 2048 Python trees would take the bench minutes: that is the user's script,
                # not this library) -- compiled as channel blocks on the host threads the job may use, then launched.
                import time as _time
                from waveforms_amd import _flatten as _fl
                trees = [mk(c) for c in range(wch // wt)]
                gg = _fl.grid_from_desc(g)
                t_a = _time.perf_counter()
                p16 = _fl.flatten(trees)
                t_b = _time.perf_counter()
                each = []                      # every one of the 16 channels as a fresh single-channel plan of its own
                for tr in trees:
                    t_0 = _time.perf_counter()
                    pl1 = _engine.Plan(_fl.flatten([tr]), grid=gg)
                    each.append((_time.perf_counter() - t_0) * 1e3)
                    pl1.close()
                t_c = _time.perf_counter()
                line['config']['plan_build_ms'] = {'flatten_ms': (t_b - t_a) / len(trees) * 1e3,
                                                   'flatten_plus_create_ms': float(np.median(each)),
                                                   'first_ms': each[0], 'max_ms': max(each), 'channels': len(each),
                                                   'channel': '1 x 1e5 pts at 2 GS/s, 1668 pulses'}
                big = _fl.tile_program(p16, wt)
                rows_of_term = np.repeat(np.arange(wt), p16.struct.n_terms)
                big.arrays['tm_amp_re'][:len(rows_of_term)] *= 1.0 + 1e-3 * rows_of_term / wt
                t_d = _time.perf_counter()
                plf = _engine.Plan(big, grid=gg)
                t_e = _time.perf_counter()
                msf = timed(lambda: plf.launch(o2.data_ptr(), b2.n, _engine.OUT_F64, stream=torch.cuda.current_stream().cuda_stream), 20, 5)
                also['awg_fresh'] = {'rows': b2.n_channels, 'kernel': plf.kernel_name(), 'create_s': t_e - t_d,
                                     'flatten_s_extrapolated': (t_b - t_a) * wt, 'build_s': (t_e - t_d) + (t_b - t_a) * wt,
                                     'host_threads': min(16, os.cpu_count() or 1), 'launch_ms': msf,
                                     'frac': nbytes / (msf * 1e-3) / 1e9 / HBM_PEAK_GBS}
                plf.close()
                # the same rows through the 1024-tap FIR: predistort(wav(t), ker) at AWG rates (fir_short)
                chn2 = SampledFir([mk(c) for c in range(wch // wt)], g, wl.c4_kernel(), np.float64, tile=wt)
                o4 = torch.empty_like(o2)
                ms4 = timed(lambda: chn2.launch_torch(o4), 20, 5)
                fst2 = FirStage(wl.c4_kernel(), b2.n, b2.n_channels, np.float64)
                msf = timed(lambda: fst2.apply_torch(o2, o4), 20, 5)
                tb4 = chn2.plan.table_bytes()
                also['awg_c4'] = {'workload': d_ + '; then the 1024-tap FIR of C4 (predistort(wav(t), ker))',
                                  'kernel': chn2.plan.kernel_name(), 'fused': chn2.fused, 'step_ms': ms4,
                                  'msamples_per_s': b2.n_channels * b2.n / (ms4 * 1e-3) / 1e6,
                                  'algorithmic_bytes_per_launch': nbytes,
                                  'frac': nbytes / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  'table_bytes_per_launch': tb4,
                                  'frac_incl_tables': (nbytes + tb4) / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  'unfused': {'sampler_kernel_ms': ms, 'fir_kernel_ms': msf, 'step_ms': ms + msf}}
                rv4 = valu_roofline('awg_c4', ms4, b2.n_channels * b2.n) if chn2.fused else None
                if rv4 is not None:
                    also['awg_c4']['roofline_valu'] = rv4
                fst2.close()
                chn2.close()
                del o4
            del o2
            b2.close()
        # ---- tiers beside the BASELINE configs: explicit non-uniform sample times (the tlist tier: what any `x`
        # that is not np.linspace / np.arange output takes; 16 B/sample, t is read), primitives without a
        # recurrence form (direct tier) and 10-tone multiplexed pulses; kernel time by HIP events
        import waveforms_amd as wfm
        from waveforms_amd import _flatten
        stream = torch.cuda.current_stream().cuda_stream
        tt = wl.jittered_times()
        tch = [wl.sum_channel(wfm, 100, 1000 + c) for c in range(64)]
        tplan = _engine.Plan(_flatten.flatten(tch), t=tt)
        o2 = torch.empty((64, len(tt)), device='cuda', dtype=torch.float64)
        ms = timed(lambda: tplan.launch(o2.data_ptr(), len(tt), _engine.OUT_F64, stream=stream), 20, 3)
        nb = 64 * len(tt) * 16
        also['tlist'] = {'workload': '64 ch x 2e6 explicit non-uniform times (the C2 grid jittered by 0.3 dt, sorted), headline channels '
                                     '(100 gaussian+DRAG pulses): Waveform.__call__(x) on a non-grid x',
                         'kernel': tplan.kernel_name(), 'kernel_ms': ms, 'msamples_per_s': 64 * len(tt) / (ms * 1e-3) / 1e6,
                         'algorithmic_bytes_per_launch': nb, 'bytes_per_sample': 16,
                         'frac': nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'dtype': 'f64',
                         'fused_ops': int(tplan.info.n_fused), 'generic_terms': int(tplan.info.n_generic)}
        tplan.close()
        del o2
        direct = {}
        for shape in ('sinc', 'mollifier', 'interp'):
            b2 = BatchSampler([wl.direct_channel(wfm, shape, c) for c in range(8)], ('linspace', 0.0, wl.DIRECT_T, 10**7, False), tile=8)
            o2 = torch.empty((b2.n_channels, b2.n), device='cuda', dtype=torch.float64)
            ms = timed(lambda: b2.launch_torch(o2), 5, 2)
            nb = b2.n_channels * b2.n * 8
            direct[shape] = {'kernel': b2.plan.kernel_name(), 'kernel_ms': ms, 'msamples_per_s': b2.n_channels * b2.n / (ms * 1e-3) / 1e6,
                             'algorithmic_bytes_per_launch': nb, 'frac': nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'fused_ops': int(b2.plan.info.n_fused), 'generic_terms': int(b2.plan.info.n_generic)}
            b2.close()
            del o2
        direct['workload'] = ('64 rows (8 distinct x 8) x 1e7 pts over 3 us: 25 overlapping sinc pulses / 100 mollifier pulses / '
                              '100 samplingPoints envelopes of 1000 knots under carriers; awg_interp: 2048 rows (16 distinct x 128) '
                              'x 1e5 pts at 2 GS/s, samplingPoints envelopes (301 knots, 8 shapes per channel) under carriers, '
                              '60 samples per pulse')
        b2 = BatchSampler([wl.awg_interp_channel(wfm, c) for c in range(16)], wl.awg_grid(), tile=128)
        o2 = torch.empty((b2.n_channels, b2.n), device='cuda', dtype=torch.float64)
        ms = timed(lambda: b2.launch_torch(o2), 5, 2)
        nb = b2.n_channels * b2.n * 8
        direct['awg_interp'] = {'kernel': b2.plan.kernel_name(), 'kernel_ms': ms, 'msamples_per_s': b2.n_channels * b2.n / (ms * 1e-3) / 1e6,
                                'algorithmic_bytes_per_launch': nb, 'table_bytes_per_launch': int(b2.plan.table_bytes()),
                                'frac': nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'frac_incl_tables': (nb + b2.plan.table_bytes()) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'fused_ops': int(b2.plan.info.n_fused), 'generic_terms': int(b2.plan.info.n_generic)}
        b2.close()
        del o2
        shapes = {}
        # the `also.awg` rows 1 ms from t = 0 (the tail of a 1 ms sequence): carriers with the grid-rounding correction,
        # family 6 of the short tier
        try:
            far_g = ('arange', 1e-3, 1e-3 + 1e5 / 2e9, 1.0 / 2e9)
            b2 = BatchSampler([wl.awg_channel(wfm, c) >> 1e-3 for c in range(16)], far_g, tile=128)
            o2 = torch.empty((b2.n_channels, b2.n), device='cuda', dtype=torch.float64)
            ms = timed(lambda: b2.launch_torch(o2), 20, 5)
            nb = b2.n_channels * b2.n * 8
            shapes['far_1ms'] = {'kernel': b2.plan.kernel_name(), 'kernel_ms': ms, 'msamples_per_s': b2.n_channels * b2.n / (ms * 1e-3) / 1e6,
                                 'frac': nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'fused_ops': int(b2.plan.info.n_fused),
                                 'generic_terms': int(b2.plan.info.n_generic)}
            b2.close()
            del o2
        except Exception as e:
            shapes['far_1ms'] = {'error': repr(e)}
        for shape in ('flat_top', 'linear_chirp', 'ten_tones', 'exp_chirp'):
            b2 = BatchSampler([wl.awg_shape_channel(wfm, shape, c) for c in range(16)], wl.awg_grid(), tile=128)
            o2 = torch.empty((b2.n_channels, b2.n), device='cuda', dtype=torch.float64)
            ms = timed(lambda: b2.launch_torch(o2), 3, 1)
            nb = b2.n_channels * b2.n * 8
            shapes[shape] = {'kernel': b2.plan.kernel_name(), 'kernel_ms': ms, 'msamples_per_s': b2.n_channels * b2.n / (ms * 1e-3) / 1e6,
                             'frac': nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'fused_ops': int(b2.plan.info.n_fused),
                             'generic_terms': int(b2.plan.info.n_generic)}
            b2.close()
            del o2
        shapes['workload'] = ('2048 rows (16 distinct x 128) x 1e5 pts at 2 GS/s, 60-sample pulses back to back: flat tops with erf edges / '
                              'linear chirps (short tier ops), a Gaussian under ten tones, exponential chirps (device libm: evaluated '
                              'pointwise, one sample per lane)')
        also['awg_shapes'] = shapes
        also['direct'] = direct
        b2 = BatchSampler([wl.multitone_channel(wfm, c) for c in range(8)], wl.c2_grid(), tile=8)
        o2 = torch.empty((b2.n_channels, b2.n), device='cuda', dtype=torch.float64)
        ms = timed(lambda: b2.launch_torch(o2), 10, 3)
        nb = b2.n_channels * b2.n * 8
        also['multitone'] = {'workload': '64 rows (8 distinct x 8) x 1e7 pts, 100 gaussian pulses per row, 10 tones under every pulse',
                             'kernel': b2.plan.kernel_name(), 'kernel_ms': ms, 'msamples_per_s': b2.n_channels * b2.n / (ms * 1e-3) / 1e6,
                             'algorithmic_bytes_per_launch': nb, 'frac': nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'dtype': 'f64',
                             'fused_ops': int(b2.plan.info.n_fused), 'generic_terms': int(b2.plan.info.n_generic)}
        b2.close()
        del o2
        line['also'] = also
    if rank == 0 and not args.no_cpu_baseline and name in (
            'sampler256', 'c4', 'c5', 'c2', 'far', 'far_sparse', 'awg', 'awg_duty30'):
        base, outs = cpu_baseline(chans, grid)
        line['cpu_baseline'] = base        # (timed after the last barrier of the timed region: rank 0's host only)
        line['speedup_vs_cpu_baseline'] = line['value'] / base['value']
        got = out[:len(outs)].cpu().numpy().astype(np.float64)
        if time_sharded:      # rank 0 holds the first time slice of every row
            outs = [o[sh.start:sh.stop] for o in outs]
            got = got[:, sh.own]
        line['max_abs_err_vs_numpy_ref'] = float(
            max(np.max(np.abs(got[i] - outs[i])) for i in range(len(outs))))
    if dist is not None and args.gather_rows > 0:
        # optional result placement over xGMI, timed on its own (a full 20 GB block per rank is
        # gather-bound by > 10x over the compute): bounded slice, extrapolated
        try:
            rows = min(args.gather_rows, bs.n_channels)
            src = out[:rows].contiguous()
            if args.backend != 'nccl':
                src = src.cpu()
            dst = torch.empty((world * rows, bs.n), dtype=out.dtype, device=src.device)
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
                'note': ('all_gather (RCCL over xGMI)' if args.backend == 'nccl' else 'all_gather (gloo, host)') +
                        ' of a row slice; full block extrapolated linearly'}
            del dst
        except Exception as exc:   # placement is optional: never lose the bench line over it
            line['gather'] = {'error': repr(exc)}
    if rank == 0:
        print(compact_line(line))
    if dist is not None:
        dist.destroy_process_group()
    return 0


LINE_LIMIT = 7500        # characters; the driver keeps ~8000 of stdout


def compact_line(line):
    """The ONE JSON line, kept inside the driver's stdout window: floats to 5 significant digits; the prose of the
    `also` legs (what each workload is, notes) lives in profiles/workloads.md under the same keys; if the line is still
    too long, derivable figures of the `also` legs go (msamples_per_s = samples / kernel_ms, byte counts)."""
    def rnd(o):
        if isinstance(o, float):
            return float('%.5g' % o) if o == o and abs(o) != float('inf') else o
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [rnd(v) for v in o]
        return o

    def strip(o, keys):
        if isinstance(o, dict):
            return {k: strip(v, keys) for k, v in o.items() if k not in keys}
        return o

    line = rnd(line)
    if 'also' in line:
        line['also'] = strip(line['also'], ('workload', 'note'))
        line['also']['doc'] = 'profiles/workloads.md'
    out = json.dumps(line, separators=(',', ':'))
    for keys in (('algorithmic_bytes_per_launch', 'table_bytes_per_launch'), ('msamples_per_s',), ('fused_ops', 'generic_terms')):
        if len(out) <= LINE_LIMIT or 'also' not in line:
            break
        line['also'] = strip(line['also'], keys)
        out = json.dumps(line, separators=(',', ':'))
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == '__main__':
    sys.exit(main())
