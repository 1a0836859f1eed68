"""(a) Carriers far from t = 0 stay on the fused fast path with the per-sample grid-rounding
correction (reference evaluates cos(w*(x - shift)) AT NumPy's rounded grid values,
_waveform.pyx:156,307-308), and (b) the drop-in wav(x) compiles grid mode when x is
bit-identical to a np.linspace / np.arange grid."""
import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import c_oracle, np_oracle
from waveforms_amd import _engine, _flatten, _sampling, workloads as wl

pytestmark = pytest.mark.gpu
SAMPLES = golden_io.npz('samples.npz')


def _far_channel(t_center, f, width=200e-9, seed=0):
    rng = np.random.default_rng(seed)
    I, Q = wf.mixing(rng.uniform(0.3, 1) * wf.gaussian(width) >> t_center, freq=f,
                     phase=rng.uniform(0, 6.28), DRAGScaling=1e-10)
    return I, Q


@pytest.mark.parametrize('t_center,f', [(1e-3, 300e6), (0.99e-3, -347e6), (16e-3, 300e6),
                                        (0.05, 100e6), (-3.3e-3, 410e6)])
def test_far_carriers_stay_fused(t_center, f):
    span = 2e-6
    n = 40001
    chans = list(_far_channel(t_center, f)) + [wf.cos(2 * np.pi * f) * (wf.square(1e-6) >> t_center)]
    for endpoint in (False, True):
        g = _flatten.grid_linspace(t_center - span / 2, t_center + span / 2, n, endpoint)
        prog = _flatten.flatten(chans)
        plan = _engine.Plan(prog, grid=g)
        assert plan.info.n_direct == 0 and plan.info.n_generic == 0 and plan.info.n_fused > 0
        # the correction is taken where the grid-rounding bound exceeds the budget: 2.5e-10 shared by the terms that meet in
        # a piece (three for a pulse as mixing() makes it) -- certainly beyond 2.5e-10, never below a third of it
        x = 2 * np.pi * abs(f) * 1.2e-16 * (abs(t_center) + span / 2 + span)
        need = x > 2.5e-10
        corrected = plan.kernel_name() == 'wfk_sample_lean<double,false,16,true,0>'
        assert corrected or plan.kernel_name() == 'wfk_sample_lean<double,false,16,false,0>', plan.kernel_name()
        assert corrected if need else (not corrected if x <= 2.5e-10 / 3.01 else True)
        got = plan.run_host(np.float64)
        ora = c_oracle.eval_grid(prog, g)
        assert np.max(np.abs(got - ora)) <= 1e-9, (t_center, f, np.max(np.abs(got - ora)))
        # NumPy restatement of the reference on the materialised grid, too
        t = _flatten.grid_values(g)
        ref = np_oracle.call(chans[0], t)
        assert np.max(np.abs(got[0] - ref)) <= 1e-9
        # long enough for carried state: many tiles between exact reseeds
    os_env = __import__('os').environ
    os_env['WFK_DISABLE_CORR'] = '1'
    try:
        plan = _engine.Plan(_flatten.flatten(chans), grid=g)
        assert (plan.info.n_direct > 0) == corrected   # without the correction such factors leave the fast path
        assert np.max(np.abs(plan.run_host(np.float64) - ora)) <= 1e-9
    finally:
        del os_env['WFK_DISABLE_CORR']


def test_far_sequence_full_length():
    # a 1 ms sequence at 2 GS/s: 100 pulses 10 us apart under +-(250..350) MHz carriers
    import bench
    chans = [bench.far_channel(wf, c) for c in range(3)]
    g = _flatten.grid_linspace(0.0, 1e-3, 2_000_000, False)
    prog = _flatten.flatten(chans)
    plan = _engine.Plan(prog, grid=g)
    assert plan.info.n_direct == 0
    got = plan.run_host(np.float64)
    ora = c_oracle.eval_grid(prog, g)
    assert np.max(np.abs(got - ora)) <= 1e-9
    f32 = plan.run_host(np.float32)
    assert np.max(np.abs(f32 - ora)) <= FP32_TOL


@pytest.mark.parametrize('name', ['readme_x', 'readme_y', 'drag_block', 'vstack4', 'vstack_ops', 'mix_env', 'c2_small', 'c3_small', 'cospulse', 'clip', 'complex_amp', 'square_erf'])
def test_call_on_linspace_compiles_grid_mode(name):
    if name not in cases.CASES:
        pytest.skip('case not defined')
    build, grid = cases.CASES[name]
    w = build(wf)
    t = wl.make_grid(grid)
    plan, _owned = _sampling._plan_for_axis(w, t, None)
    try:
        name_ = plan.kernel_name()
        assert name_.startswith(('wfk_sample_lean<', 'wfk_sample_short<')) or name_.split(',')[2] == 'false'   # not the tlist kernel
    finally:
        if _owned:
            plan.close()
    want = SAMPLES[name + '.y']
    got = w(t)
    assert got.dtype == want.dtype
    assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.abs(want).max())


def test_detection_is_exact_and_falls_back():
    w = wl.sum_channel(wf, 6, 1000)
    t = np.linspace(0.0, 6 * wl.SPAN, 50000, endpoint=False)
    assert _engine.detect_grid(t) is not None
    for make in (lambda a: np.linspace(a[0], a[-1], len(a)),          # endpoint=True
                 lambda a: np.arange(0.0, 6 * wl.SPAN, a[1] - a[0]),  # arange
                 lambda a: a + 1e-3):                                 # shifted: not a grid formula? maybe
        tt = make(t)
        g = _engine.detect_grid(tt)
        if g is not None:
            assert np.array_equal(_flatten.grid_values(g), tt)
        assert np.max(np.abs(w(tt) - np_oracle.call(w, tt))) <= 1e-9
    t2 = t.copy()
    t2[12345] = np.nextafter(t2[12345], 1.0)                          # one ulp off: tlist mode
    assert _engine.detect_grid(t2) is None
    assert np.max(np.abs(w(t2) - np_oracle.call(w, t2))) <= 1e-9
    assert _engine.detect_grid(np.sort(np.random.default_rng(1).uniform(0, 1e-6, 5000))) is None
    assert _engine.detect_grid(t[:10]) is None                        # tiny arrays: not worth it


def test_x_of_several_grids_is_sampled_run_by_run(monkeypatch):
    """wav(x) with x = np.concatenate of grids (windows of one sequence, two sample rates), WFK_GRID_RUNS=1: every
    run in grid mode; the result equals the calls on the runs one by one, bit for bit, and the oracle on the
    whole x.  (Default: the whole x as one time list -- faster since round 4 -- checked at the end.)"""
    monkeypatch.setattr(_sampling, '_RUNS_ON', True)
    x_wav = cases.CASES['readme_x'][0](wf)
    parts = [np.linspace(-0.2e-6, 0.4e-6, 60000, endpoint=False), np.arange(0.9e-6, 1.3e-6, 0.25e-10),
             np.linspace(1.95e-6, 2.2e-6, 20001)]
    t = np.concatenate(parts)
    assert _engine.detect_grid(t) is None and len(_engine.detect_grid_runs(t)) == 3
    got = x_wav(t)
    piecewise = np.concatenate([x_wav(p) for p in parts])
    assert np.array_equal(got, piecewise)
    ora = np_oracle.call(x_wav, t)
    pk = np.abs(ora).max()
    assert np.max(np.abs(got - ora)) <= 1e-9 * pk
    # out= / accumulate on that path, a complex waveform, a WaveVStack
    buf = np.full(len(t) + 5, 2.0)
    r = x_wav(t, out=buf, accumulate=True)
    assert r is buf and np.max(np.abs(buf[:len(t)] - 2.0 - ora)) <= 1e-9 * pk and np.all(buf[len(t):] == 2.0)
    wc = (1 + 2j) * (wf.gaussian(200e-9) >> 1.1e-6) * wf.cos(2 * np.pi * 50e6) + wf.gaussian(100e-9)
    gc = wc(t)
    assert gc.dtype == np.complex128 and np.max(np.abs(gc - np_oracle.call(wc, t))) <= 1e-9
    vs = wf.WaveVStack([wf.gaussian(300e-9) >> 1e-6, wf.cos(2 * np.pi * 10e6) * 0.1]) + 0.5
    assert np.max(np.abs(vs(t) - np.real(np_oracle.call(vs, t)))) <= 1e-9
    monkeypatch.setattr(_sampling, '_RUNS_ON', False)
    one_list = x_wav(t)
    assert np.max(np.abs(one_list - ora)) <= 1e-9 * pk


@pytest.mark.parametrize('t0', [1e-4, 1e-3])
def test_negative_amplitudes_far_from_t_zero(t0):
    """A term with a NEGATIVE amplitude milliseconds from t = 0 (found by tools/fuzz_soak.py awgfar through coshPulse's
    23 cos - 22 cos cosh, 3e-9 of peak): the host folded A < 0, B = 0 into the group's reference shift as phi = pi, and the
    moved shift, rounded to a double next to |t|, cost W ulp(t) / 2 of phase -- 7e-12 of relative error per ms on every
    such term.  Positive and negative amplitudes are equally exact now: bare carriers, under exponential and Gaussian
    envelopes, the coshPulse itself; fine grids (lean kernel) and AWG-rate grids."""
    for rate, n in ((2e9, 140), (4e10, 2800)):
        g = _flatten.grid_arange(t0 + 4e-7, t0 + 4e-7 + n / rate, 1 / rate)
        shapes = (lambda a: a * (wf.cos(7.7e8) * wf.square(4.9e-8)) >> (t0 + 4.4e-7),
                  lambda a: a * (wf.exp(1.2e7) * wf.cos(7.7e8) * wf.square(4.9e-8)) >> (t0 + 4.4e-7),
                  lambda a: a * (wf.gaussian(4e-8) * wf.cos(7.7e8, 0.4)) >> (t0 + 4.4e-7),
                  lambda a: a * (wf.coshPulse(4.9e-8, eps=0.6) * wf.cos(7.7e8, 1.1)) >> (t0 + 4.4e-7))
        for mk in shapes:
            for a in (1.0, -1.0, 40.0, -40.0):
                prog = _flatten.flatten([mk(a)])
                ora = c_oracle.eval_grid(prog, g)[0]
                got = _engine.Plan(prog, grid=g).run_host(np.float64)[0]
                pk = float(np.abs(ora).max())
                assert np.max(np.abs(got - ora)) <= 2e-12 * max(pk, abs(a)), (rate, a, np.max(np.abs(got - ora)) / pk)
