"""Short-piece tier (wfk_short.hip) on AWG-rate grids against vectors of the REAL reference
(tests/golden/awg.npz: pulse trains on 1-5 GS/s np.arange grids, full vectors), the C oracle, and the
standard tiers evaluating the same plans."""
import os

import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl
from waveforms_amd._sampling import BatchSampler

pytestmark = pytest.mark.gpu
AWG = golden_io.npz('awg.npz')


def _case(name):
    build, rate, n = cases.AWG_CASES[name]
    return build(wf, rate), cases._awg_grid(n, rate), rate, n


@pytest.mark.parametrize('name', sorted(cases.AWG_CASES))
def test_short_tier_matches_reference_vectors(name):
    w, grid, rate, n = _case(name)
    want = AWG[name + '.y']
    cplx = np.iscomplexobj(want)
    pk = max(1.0, float(np.abs(want).max()))
    prog = _flatten.flatten([w])
    plan = _engine.Plan(prog, grid=_flatten.grid_from_desc(grid))
    assert plan.kernel_name().startswith('wfk_sample_short<') and plan.info.n_direct == 0
    got = plan.run_host(np.complex128 if cplx else np.float64)[0]
    assert np.max(np.abs(got - want)) <= 1e-9 * pk           # north_star: fp64 max abs err < 1e-9
    assert np.max(np.abs(got - want)) <= 5e-11 * pk          # (measured: grid jitter x carrier, <= 2e-11)
    g32 = plan.run_host(np.complex64 if cplx else np.float32)[0]
    assert np.max(np.abs(g32 - want)) <= 2e-6 * pk           # float output of fp64 arithmetic: rounding only
    # the drop-in call on the caller's own np.arange array (recognised as a grid)
    t = wl.make_grid(grid)
    y = np.asarray(w(t))
    assert y.dtype == want.dtype and np.max(np.abs(y - want)) <= 5e-11 * pk


@pytest.mark.parametrize('name', ['b2b_2g', 'duty30_2g', 'readme_2p4g', 'mixed_2g', 'clip_2g'])
def test_sample_api_and_out_accumulate(name):
    w, grid, rate, n = _case(name)
    want = AWG[name + '.y']
    pk = max(1.0, float(np.abs(want).max()))
    w.start, w.stop, w.sample_rate = 0.0, n / rate, rate
    y = w.sample()                                            # reference waveform.py:173-207
    assert y.shape == want.shape and np.max(np.abs(y - want)) <= 5e-11 * pk
    t = wl.make_grid(grid)
    out = np.full(len(t), 0.25)
    r = w(t, out=out, accumulate=True)                        # out += samples (waveform.py:548-563)
    assert r is out and np.max(np.abs(out - 0.25 - want)) <= 5e-11 * pk
    # chunked: np.linspace(start, stop, size, endpoint=False) per chunk (waveform.py:209-257) -- other
    # grid values than np.arange's, so a sample that sits ON a bound may change sides: compare with the
    # oracle on the very same chunk grids
    chunks = list(w.sample(chunk_size=3000))
    prog = _flatten.flatten([w])
    a, length = float(w.start), 3000 / rate                   # the chunk starts accumulate (waveform.py:224-257)
    for ch in chunks:
        b = float(w.stop) if a + length > w.stop else a + length
        g = _flatten.grid_linspace(a, b, len(ch), endpoint=False)
        assert np.max(np.abs(ch - c_oracle.eval_grid(prog, g)[0])) <= 5e-11 * pk
        a = b
    assert sum(len(c) for c in chunks) == n


def test_accumulate_launch_and_channel_batches():
    chans = [wl.awg_channel(wf, c, 30000, 2e9, duty30=(c % 2 == 1)) for c in range(5)]
    grid = wl.awg_grid(30000, 2e9)
    bs = BatchSampler(chans, grid)
    assert bs.plan.kernel_name().startswith('wfk_sample_short<')
    ref = c_oracle.eval_grid(bs.prog, bs.grid)
    got = bs.to_host(np.float64)
    assert np.max(np.abs(got - ref)) <= 5e-11
    # device-side accumulate into a padded buffer: rows keep their stride, the padding its poison
    stride = 30000 + 24
    buf = _engine.DeviceBuffer(5 * stride * 8)
    init = np.full((5, stride), 3.0)
    buf.upload(init)
    bs.launch(buf.ptr, stride, np.float64, accumulate=True)
    _engine.sync()
    back = buf.download((5, stride), np.float64)
    assert np.max(np.abs(back[:, :30000] - 3.0 - ref)) <= 5e-11
    assert np.array_equal(back[:, 30000:], init[:, 30000:])
    buf.close()
    bs.close()


@pytest.mark.parametrize('name', ['b2b_2g', 'mixed_2g', 'cplx_2g', 'vstack_2g', 'clip_2g', 'readme_1g'])
def test_short_and_standard_tiers_agree(name):
    w, grid, rate, n = _case(name)
    want = AWG[name + '.y']
    cplx = np.iscomplexobj(want)
    prog = _flatten.flatten([w])
    g = _flatten.grid_from_desc(grid)
    a = _engine.Plan(prog, grid=g)
    os.environ['WFK_SHORT'] = '0'
    try:
        b = _engine.Plan(prog, grid=g)
    finally:
        del os.environ['WFK_SHORT']
    assert a.kernel_name().startswith('wfk_sample_short<') and not b.kernel_name().startswith('wfk_sample_short<')
    dt = np.complex128 if cplx else np.float64
    ya, yb = a.run_host(dt)[0], b.run_host(dt)[0]
    pk = max(1.0, float(np.abs(want).max()))
    assert np.max(np.abs(ya - yb)) <= 5e-11 * pk
    assert np.max(np.abs(yb - want)) <= 1e-9 * pk


def test_ragged_ends_offsets_and_tiny_grids():
    # grids that end inside a pulse / inside a gap, units with one slot, n below one row, vstack offset
    w = wl.awg_channel(wf, 7, 6000, 2e9, duty30=True)
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 1007, 1008, 1009, 1024, 2999, 3001):
        g = _flatten.grid_arange(0.0, n / 2e9, 1 / 2e9)
        prog = _flatten.flatten([w])
        plan = _engine.Plan(prog, grid=g)
        got = plan.run_host(np.float64)[0]
        ref = c_oracle.eval_grid(prog, g)[0]
        assert got.shape == ref.shape and np.max(np.abs(got - ref), initial=0.0) <= 5e-11, n
    st = (wf.WaveVStack([wl.awg_channel(wf, c, 6000, 2e9) for c in range(3)]) >> 1.7e-9) - 0.4
    g = _flatten.grid_arange(-40e-9, 2.9e-6, 1 / 2.4e9)
    prog = _flatten.flatten([st])
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name().startswith('wfk_sample_short<')
    assert np.max(np.abs(plan.run_host(np.float64)[0] - c_oracle.eval_grid(prog, g)[0])) <= 5e-11


def test_long_zero_stretches_are_pure_fill_units():
    # 100 us of silence between two bursts: zero stretches far longer than a unit
    a = wl.awg_channel(wf, 3, 4000, 2e9)
    w = a + (a >> 60e-6) + 0.0
    g = _flatten.grid_arange(0.0, 70e-6, 1 / 2e9)
    prog = _flatten.flatten([w])
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name().startswith('wfk_sample_short<')
    got = plan.run_host(np.float64)[0]
    ref = c_oracle.eval_grid(prog, g)[0]
    assert np.max(np.abs(got - ref)) <= 5e-11
    assert np.count_nonzero(got[10000:110000]) == 0


@pytest.mark.parametrize('seed', range(80))
def test_random_awg_script(seed):
    """Random pulse trains on 1-5 GS/s grids (tests/cases.py random_awg_channel: every shape of the
    general fuzz, sums / vstacks / clips / complex amplitudes): short tier alone or mixed with the
    general kernel for the pieces it cannot take (erf edges, mollifiers, unfusable DRAG primitives)."""
    rng = np.random.default_rng(10_000 + seed)
    ch, grid = cases.random_awg_channel(wf, rng)
    prog = _flatten.flatten([ch])
    g = _flatten.grid_from_desc(grid)
    cplx = bool(prog.complex_amp)
    ora = c_oracle.eval_grid(prog, g, cplx)[0]
    pk = max(1.0, float(np.max(np.abs(ora), initial=0.0)))
    plan = _engine.Plan(prog, grid=g)
    got = plan.run_host(np.complex128 if cplx else np.float64)[0]
    assert np.all(np.isfinite(got)) or not np.all(np.isfinite(ora))
    assert np.max(np.abs(got - ora), initial=0.0) <= 1e-9 * pk, (seed, plan.kernel_name())
    got32 = plan.run_host(np.complex64 if cplx else np.float32)[0]
    assert np.max(np.abs(got32 - ora), initial=0.0) <= FP32_TOL * pk, seed
    # accumulate through both launches of a mixed plan: every sample is still written exactly once
    if seed % 5 == 0 and not cplx:
        n = plan.n
        buf = _engine.DeviceBuffer(max(n, 1) * 8)
        buf.upload(np.full(max(n, 1), 2.0))
        plan.launch(buf.ptr, n, _engine.OUT_F64, accumulate=True)
        _engine.sync()
        back = buf.download((max(n, 1), ), np.float64)[:n]
        assert np.max(np.abs(back - 2.0 - ora), initial=0.0) <= 1e-9 * pk, seed
        buf.close()


def test_flat_top_pulses_with_erf_edges_stay_in_the_short_tier():
    """square(width, edge) at AWG rates: 0.5 +- 0.5 erf edges of a handful of samples (reference
    waveform.py:1096-1112).  An edge under ONE carrier (or none) is an own-term op over the edge's host-sampled
    table (family 2); under several carriers the short tier multiplies what the edge piece's ops accumulated by
    m0 + m1 erf(v) per sample (libm erf, family 1; WFK_NO_SHORT_ERFTAB=1: every edge); nothing goes to the general kernel."""
    for rate, edge, tones in ((2e9, 5e-9, 1), (1e9, 3e-9, 1), (2.4e9, 8e-9, 3), (5e9, 2e-9, 2)):
        w = wf.zero()
        for k in range(40):
            env = wf.square(40e-9, edge=edge) >> (30e-9 + 75e-9 * k)
            car = wf.cos(2 * np.pi * (50e6 + 1e6 * k), 0.1 * k)
            for j in range(1, tones):
                car = car + 0.5 * wf.cos(2 * np.pi * (80e6 * j + 3e6 * k), 0.2 * j)
            w = w + (0.3 + 0.01 * k) * env * car
        # a bare flat top, a DRAG-corrected one and a Gaussian-windowed one
        w = w + 0.4 * (wf.square(30e-9, edge=4e-9) >> 3.1e-6)
        I, Q = wf.mixing(wf.square(50e-9, edge=6e-9) >> 3.2e-6, freq=120e6, phase=0.3, DRAGScaling=2e-10)
        w = w + I - 0.3 * Q
        g = _flatten.grid_arange(0.0, 3.4e-6, 1 / rate)
        prog = _flatten.flatten([w])
        plan = _engine.Plan(prog, grid=g)
        assert plan.kernel_name() == 'wfk_sample_short<double,false,false,16,2>', plan.kernel_name()
        assert plan.info.n_direct == 0 and plan.info.n_generic == 0
        ora = c_oracle.eval_grid(prog, g)[0]
        got = plan.run_host(np.float64)[0]
        assert np.max(np.abs(got - ora)) <= 1e-10, rate
        assert np.max(np.abs(plan.run_host(np.float32)[0] - ora)) <= 2e-6
        os.environ['WFK_NO_SHORT_ERFTAB'] = '1'       # the closing-op form of every edge
        try:
            plan = _engine.Plan(prog, grid=g)
        finally:
            del os.environ['WFK_NO_SHORT_ERFTAB']
        assert plan.kernel_name() == 'wfk_sample_short<double,false,false,16,1>', plan.kernel_name()
        assert np.max(np.abs(plan.run_host(np.float64)[0] - ora)) <= 1e-10, rate
    # complex amplitudes and clip on top
    w2 = (0.6 + 0.3j) * (wf.square(40e-9, edge=5e-9) >> 60e-9) * wf.cos(2 * np.pi * 70e6) + 0.2 * (wf.gaussian(20e-9) >> 160e-9)
    g = _flatten.grid_arange(0.0, 0.3e-6, 1 / 2e9)
    prog = _flatten.flatten([w2])
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name(np.complex128).startswith('wfk_sample_short<double,true')
    assert np.max(np.abs(plan.run_host(np.complex128)[0] - c_oracle.eval_grid(prog, g, True)[0])) <= 1e-10


@pytest.mark.parametrize('t0', [0.0, 1e-4])
@pytest.mark.parametrize('jitter', [False, True])
def test_flat_top_edges_as_own_term_ops_over_sampled_tables(t0, jitter):
    """edges that share a table (pulses on the sample grid alike) and edges with a table each (arbitrary pulse times),
    at t = 0 and 100 us from it (where the rounding noise of t - shift exceeds the 2e-11 cap on what two edges may
    differ by and still share): real and complex launches against the C oracle."""
    rate, n = 2e9, 30000
    g = _flatten.grid_arange(t0, t0 + n / rate, 1 / rate)
    rng = np.random.default_rng(11)
    w = wf.zero()
    for k in range(200):
        at = t0 + (k + 0.5) * 70e-9 + (rng.uniform(0, 0.5e-9) if jitter else 0.0)
        amp = rng.uniform(0.2, 1) if k % 3 else complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        car = wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6)) if k % 5 else 1.0
        w = w + ((amp * wf.square(rng.choice([30e-9, 36e-9]), edge=rng.choice([2e-9, 3e-9])) * car) >> at)
    prog = _flatten.flatten([w])
    plan = _engine.Plan(prog, grid=g)
    assert plan.info.n_generic == 0
    if t0 == 0.0:
        assert plan.kernel_name(np.complex128) == 'wfk_sample_short<double,true,false,16,2>'
    ora = c_oracle.eval_grid(prog, g, True)[0]
    assert np.max(np.abs(plan.run_host(np.complex128)[0] - ora)) <= 1e-10
    assert np.max(np.abs(plan.run_host(np.float64)[0] - ora.real)) <= 1e-10


@pytest.mark.parametrize('duty30', [False, True])
def test_bench_shape_2048_rows_full_size(duty30):
    """the `awg` / `awg_duty30` bench workloads as bench.py launches them: 2048 rows x 1e5 points at 2 GS/s,
    16 distinct channels x 128 copies (every row with its own device tables), fp64 and fp32, plus the fused
    chain (fir_short) on the same rows.  Distinct rows against the C oracle over the full row; copies equal
    their originals bit for bit; linearity of the FIR as the size-independent property of the chain."""
    import torch
    from waveforms_amd.distortion import SampledFir
    rows, n, tile = 2048, 100_000, 128
    chans = [wl.awg_channel(wf, c, n, 2e9, duty30) for c in range(rows // tile)]
    grid = wl.awg_grid(n, 2e9)
    bs = BatchSampler(chans, grid, tile=tile)
    assert bs.n_channels == rows and bs.plan.kernel_name().startswith('wfk_sample_short<')
    ref = c_oracle.eval_grid(_flatten.flatten(chans), bs.grid)
    out = torch.empty((rows, n), dtype=torch.float64, device='cuda')
    bs.launch_torch(out)
    torch.cuda.synchronize()
    first = out[:rows // tile].cpu().numpy()
    assert np.max(np.abs(first - ref)) <= 5e-11
    assert bool((out.view(tile, rows // tile, n) == out[:rows // tile].unsqueeze(0)).all())
    o32 = torch.empty((rows, n), dtype=torch.float32, device='cuda')
    bs.launch_torch(o32)
    torch.cuda.synchronize()
    assert float((o32.double() - out).abs().max()) <= 2e-6
    ker = wl.c4_kernel()
    sf = SampledFir(chans, grid, ker, tile=tile)
    assert sf.fused and sf.plan.kernel_name() == 'fir_short<double,12>'
    y = torch.empty_like(out)
    sf.launch_torch(y)
    torch.cuda.synchronize()
    want = np.stack([c_oracle.fir(r, ker) for r in ref[:3]])
    assert np.max(np.abs(y[:3].cpu().numpy() - want)) <= 1e-11
    assert bool((y.view(tile, rows // tile, n) == y[:rows // tile].unsqueeze(0)).all())
    sf.close()
    bs.close()


def test_generic_shapes_at_awg_rates_are_evaluated_pointwise(monkeypatch):
    """Pieces of AWG-rate length whose shapes the short tier does not take (libm chirps, sinc, derivatives of
    mollifiers): the standard tiers would walk every 60-sample piece over whole wave tiles of 1024 samples, so such a
    grid plan is compiled on the grid's own sample times and runs on the time-list tier, one sample per lane
    (wfk_api.cpp: plan_create).  Same numbers as the standard tiers (WFK_NO_POINTWISE_GRID=1) and as the oracle."""
    rng = np.random.default_rng(11)
    W = wl.SPAN
    shapes = [lambda: wf.chirp(8e7, 2.5e8, W, type='exponential') * wf.cosPulse(W),
              lambda: wf.sinc(6 / W) * wf.square(W) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8)),
              lambda: wf.mixing(wf.mollifier(W), freq=rng.uniform(-2e8, 2e8), DRAGScaling=1e-10)[0]]
    chans = [wl._tree_sum([rng.uniform(0.2, 1) * mk() >> ((k + 0.5) * W) for k in range(300)]) for mk in shapes]
    grid = wl.awg_grid(20_000, 2e9)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name() == 'wfk_sample<double,false,true,true,true,1>', plan.kernel_name()
    ref = c_oracle.eval_grid(prog, g)
    pk = float(np.abs(ref).max())
    got = plan.run_host(np.float64)
    assert np.max(np.abs(got - ref)) <= 1e-11 * pk
    assert np.max(np.abs(plan.run_host(np.float32) - ref)) <= FP32_TOL * pk
    # a time slice of the grid (wfk_grid.i0) is those samples of the whole
    sl = _engine.Plan(prog, grid=_flatten.grid_slice(g, 5000, 12000)).run_host(np.float64)
    assert np.array_equal(sl, got[:, 5000:12000])
    monkeypatch.setenv('WFK_NO_POINTWISE_GRID', '1')
    std = _engine.Plan(prog, grid=g)
    assert std.kernel_name().startswith('wfk_sample<double,false,false,'), std.kernel_name()
    assert np.max(np.abs(std.run_host(np.float64) - ref)) <= 1e-9 * pk
    monkeypatch.delenv('WFK_NO_POINTWISE_GRID')
    # long pieces of the same shapes stay on the standard tiers
    long_grid = ('linspace', 0.0, 300 * W, 3_000_000, False)
    assert not _engine.Plan(prog, grid=_flatten.grid_from_desc(long_grid)).kernel_name().startswith('wfk_sample<double,false,true')


def test_linear_chirps_at_awg_rates_run_on_the_short_tier(monkeypatch):
    """chirp(f0, f1, T) pulses of 60-240 samples: a quadratic-phase op of the short tier (z_{k+1} = z_k w_k, w_{k+1} = w_k v),
    under windows with carriers, Gaussian envelopes, complex amplitudes; against the oracle and the same plan with the
    op off (the lean kernel's chirp family / the general kernel)."""
    rng = np.random.default_rng(21)
    W = wl.SPAN

    def pulse(k):
        T = W * float(rng.choice([1, 2, 4]))
        ch = wf.chirp(rng.uniform(-2e8, 2e8), rng.uniform(-3e8, 3e8), T, rng.uniform(0, 6))
        env = [wf.cosPulse(T), wf.square(T), wf.gaussian(T / 3) * wf.square(T), wf.hanning(T) * wf.cos(2 * np.pi * 5e7)][k % 4]
        amp = rng.uniform(0.2, 1) if k % 5 else complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        return amp * ch * env, T
    chans = []
    for c in range(3):
        t, ws = 0.0, []
        while t < 9e-6:
            p, T = pulse(len(ws))
            ws.append(p >> (t + T / 2))
            t += T
        chans.append(wl._tree_sum(ws))
    grid = wl.awg_grid(20_000, 2e9)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name(np.complex128) == 'wfk_sample_short<double,true,false,16,1>' and plan.info.n_generic == 0
    ref = c_oracle.eval_grid(prog, g, True)
    pk = float(np.abs(ref).max())
    got = plan.run_host(np.complex128)
    assert np.max(np.abs(got - ref)) <= 5e-11 * pk
    assert np.max(np.abs(plan.run_host(np.float64) - ref.real)) <= 5e-11 * pk
    assert np.max(np.abs(plan.run_host(np.complex64) - ref)) <= FP32_TOL * pk
    monkeypatch.setenv('WFK_NO_SHORT_CHIRP', '1')
    off = _engine.Plan(prog, grid=g)
    assert not off.kernel_name(np.complex128).startswith('wfk_sample_short<double,true,false,16,1>') or ' + ' in off.kernel_name(np.complex128)
    assert np.max(np.abs(off.run_host(np.complex128) - ref)) <= 1e-9 * pk


@pytest.mark.parametrize('shape', ['flat_top', 'linear_chirp', 'ten_tones', 'exp_chirp'])
def test_bench_awg_shapes_tiled_batches_against_the_oracle(shape):
    """the `also.awg_shapes` workloads as bench.py launches them -- BatchSampler with tiled copies -- at a size the oracle
    finishes in a second: every row (copies included) against its channel, fp64 and fp32"""
    import torch
    n = 30000
    chans = [wl.awg_shape_channel(wf, shape, c, n, 2e9) for c in range(5)]
    grid = wl.awg_grid(n, 2e9)
    ref = c_oracle.eval_grid(_flatten.flatten(chans), _flatten.grid_from_desc(grid))
    pk = max(1.0, float(np.abs(ref).max()))
    for tdt, tol in ((torch.float64, cases.FP64_GRID_TOL), (torch.float32, FP32_TOL)):
        bs = BatchSampler(chans, grid, tile=3)
        out = torch.empty((bs.n_channels, bs.n), dtype=tdt, device='cuda')
        bs.launch_torch(out)
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        assert bs.plan.kernel_name().startswith('wfk_sample_short<'), bs.plan.kernel_name()
        for r in range(bs.n_channels):
            assert np.max(np.abs(got[r] - ref[r % 5])) <= tol * pk, (shape, r, str(tdt))
        bs.close()


@pytest.mark.parametrize('t0', [1e-3, -2e-3, 1e-2])
def test_pulse_trains_milliseconds_from_zero_stay_on_the_short_tier(t0):
    """AWG-rate rows far from t = 0 (a 1 ms sequence at 2 GS/s): W |t| ulp is past the rounding budget, the carriers take
    the per-sample grid-rounding correction -- family 6 of the short tier (wfk_short_dev.h: short_op_corr; before it such
    plans fell to the pointwise tier: 14.2 -> 1.3 ms for 128 rows x 2e6).  Real, complex and float launches, an endpoint
    grid (overridden last sample), a time slice of the grid (wfk_grid.i0 != 0), clip, tiled copies; and the fallback for
    what family 6 does not hold (a channel with a pending shift, flat tops)."""
    import torch
    rate, n = 2e9, 40000
    rng = np.random.default_rng(3)
    def train(cplx):
        w = wf.zero()
        for k in range(n // 60 - 1):
            I, Q = wf.mixing(wf.gaussian(20e-9), freq=rng.uniform(-3e8, 3e8), phase=rng.uniform(0, 6), DRAGScaling=1e-10)
            amp = complex(rng.uniform(-1, 1), rng.uniform(-1, 1)) if (cplx and k % 3 == 0) else rng.uniform(0.2, 1)
            w = w + ((amp * (I if k % 2 else Q)) >> (t0 + (k + 0.5) * 30e-9))
        return w
    chans = [train(False), train(True), train(False)]
    chans[2].min, chans[2].max = -0.3, 0.5
    for grid in (('arange', t0, t0 + n / rate, 1 / rate), ('linspace', t0, t0 + (n - 1) / rate, n, True)):
        prog = _flatten.flatten(chans)
        g = _flatten.grid_from_desc(grid)
        plan = _engine.Plan(prog, grid=g)
        assert plan.kernel_name(np.complex128) == 'wfk_sample_short<double,true,false,16,6>' and plan.info.n_generic == 0, plan.kernel_name(np.complex128)
        ora = c_oracle.eval_grid(prog, g, True)
        pk = max(1.0, float(np.abs(ora).max()))
        assert np.max(np.abs(plan.run_host(np.complex128) - ora)) <= cases.FP64_GRID_TOL * pk
        assert np.max(np.abs(plan.run_host(np.float64) - ora.real)) <= cases.FP64_GRID_TOL * pk
        assert np.max(np.abs(plan.run_host(np.complex64) - ora)) <= FP32_TOL * pk
        assert plan.kernel_name(np.float32) == 'wfk_sample_short<float,false,false,16,6>'
        # a slice of the grid is a grid of its own: the same samples
        lo, hi = 12345, 31000
        ps = _engine.Plan(prog, grid=_flatten.grid_slice(g, lo, hi))
        assert ps.kernel_name() == 'wfk_sample_short<double,false,false,16,6>'
        assert np.max(np.abs(ps.run_host(np.float64) - ora.real[:, lo:hi])) <= cases.FP64_GRID_TOL * pk
        # without the correction (WFK_NO_SHORT_CORR=1) the plan leaves the tier and stays exact
        os.environ['WFK_NO_SHORT_CORR'] = '1'
        try:
            pn = _engine.Plan(prog, grid=g)
        finally:
            del os.environ['WFK_NO_SHORT_CORR']
        assert not pn.kernel_name().endswith(',6>')
        assert np.max(np.abs(pn.run_host(np.float64) - ora.real)) <= cases.FP64_GRID_TOL * pk
    # tiled copies through the batched API
    bs = BatchSampler(chans[:1], ('arange', t0, t0 + n / rate, 1 / rate), tile=3)
    out = torch.empty((bs.n_channels, bs.n), dtype=torch.float64, device='cuda')
    bs.launch_torch(out)
    torch.cuda.synchronize()
    ref0 = c_oracle.eval_grid(_flatten.flatten(chans[:1]), _flatten.grid_arange(t0, t0 + n / rate, 1 / rate))[0]
    assert np.max(np.abs(out.cpu().numpy() - ref0[None, :])) <= cases.FP64_GRID_TOL * max(1.0, np.abs(ref0).max())
    bs.close()
    # what family 6 does not hold goes back to the other tiers, exact all the same
    shifted = wf.WaveVStack([chans[0]]) >> 3e-9
    flat = chans[0] + ((wf.square(30e-9, edge=4e-9) * wf.cos(2 * np.pi * 2.1e8)) >> (t0 + 5e-6))
    for w in (shifted, flat):
        prog = _flatten.flatten([w])
        g = _flatten.grid_arange(t0, t0 + n / rate, 1 / rate)
        plan = _engine.Plan(prog, grid=g)
        assert not plan.kernel_name().endswith(',6>'), plan.kernel_name()
        ora = c_oracle.eval_grid(prog, g)
        assert np.max(np.abs(plan.run_host(np.float64) - ora)) <= cases.FP64_GRID_TOL * max(1.0, float(np.abs(ora).max()))
