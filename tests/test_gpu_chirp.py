"""Fused LINEAR CHIRP op of the lean kernel (family 2): sin(phi0 + 2 pi (a u^2 + f0 u)), reference
_waveform.pyx:323-324, against the C oracle (libm sin per sample) -- alone, under envelopes, multiplied
with carriers (product-to-sum keeps the phase quadratic), with polynomial factors, complex amplitudes,
clip / accumulate, in float; and the plans that must NOT take it (tlist mode, pieces with generic terms,
short pieces, plans with corrected carriers)."""
import os

import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl

pytestmark = pytest.mark.gpu
pi = np.pi


def _check(w, grid, tol=1e-9, want_kernel='wfk_sample_lean<double,false,16,false,2>', cplx=False):
    prog = _flatten.flatten([w] if not isinstance(w, list) else w)
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(prog, grid=g)
    if want_kernel is not None:
        assert plan.kernel_name(np.complex128 if cplx else np.float64) == want_kernel.replace('double,false', 'double,true' if cplx else 'double,false'), plan.kernel_name()
    ora = c_oracle.eval_grid(prog, g, cplx)
    got = plan.run_host(np.complex128 if cplx else np.float64)
    pk = max(1.0, float(np.abs(ora).max()))
    err = float(np.max(np.abs(got - ora)))
    assert err <= tol * pk, err
    g32 = plan.run_host(np.complex64 if cplx else np.float32)
    assert np.max(np.abs(g32 - ora)) <= FP32_TOL * pk
    return plan, err


GRID = ('linspace', 0.0, 4e-6, 2_000_003, False)


def test_bare_chirps_and_sums():
    w = wf.chirp(1e8, 3e8, 1e-6) >> 0.2e-6
    plan, err = _check(w, GRID)
    assert plan.info.n_direct == 0 and plan.info.n_generic == 0
    assert err <= 2e-11                                  # (measured; the budget is 1e-9)
    # down-chirp, phase offset, several back to back, amplitudes
    ws = wf.zero()
    for k in range(6):
        ws = ws + (0.3 + 0.1 * k) * (wf.chirp(2.5e8 - 2e7 * k, 0.4e8 + 1e7 * k, 0.55e-6, 0.3 * k) >> (0.1e-6 + 0.6e-6 * k))
    _check(ws, GRID)


def test_chirp_times_envelopes_carriers_polynomials():
    c = wf.chirp(0.5e8, 2.5e8, 1.5e-6) >> 0.5e-6
    _check(c * (wf.gaussian(1.2e-6) >> 1.25e-6), GRID)                                      # Gaussian envelope
    _check(c * wf.cos(2 * pi * 40e6, 0.7), GRID)                                            # x carrier: two chirps
    _check(c * (wf.gaussian(1.0e-6) >> 1.2e-6) * wf.cos(2 * pi * 25e6) * wf.sin(2 * pi * 10e6), GRID)
    _check((c * (wf.exp(-2e6) >> 0.5e-6)), GRID)                                            # exponential envelope
    _check(c * wf.poly([0.5, 2e5, -3e11]) , GRID)                                           # polynomial factor
    _check(c * (wf.chirp(1e7, 6e7, 1.5e-6) >> 0.5e-6), GRID)                                # chirp x chirp
    I, Q = wf.mixing(c * (wf.gaussian(1.2e-6) >> 1.25e-6), freq=30e6, phase=0.4)            # both quadratures
    _check(I - 0.5 * Q, GRID)


def test_complex_clip_accumulate_batches():
    c = wf.chirp(0.5e8, 2.5e8, 1.5e-6) >> 0.5e-6
    _check((1 + 2j) * c + 0.5j * (c >> 2e-6), GRID, cplx=True)
    w = 1.5 * c
    w.min, w.max = -0.8, 1.1
    _check(w, GRID)
    chans = [(0.2 + 0.1 * k) * (wf.chirp(1e8 + 1e7 * k, 3e8, 1e-6) >> (0.1e-6 * k)) + 0.3 * (wf.gaussian(0.4e-6) >> 3e-6) * wf.cos(2 * pi * 1e8)
             for k in range(5)]
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(GRID)
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name().endswith(',false,2>')
    ora = c_oracle.eval_grid(prog, g)
    n = plan.n
    buf = _engine.DeviceBuffer(5 * n * 8)
    buf.upload(np.full((5, n), 1.0))
    plan.launch(buf.ptr, n, _engine.OUT_F64, accumulate=True)
    _engine.sync()
    assert np.max(np.abs(buf.download((5, n), np.float64) - 1.0 - ora)) <= 1e-9
    buf.close()


def test_plans_that_do_not_take_the_fused_chirp():
    c = wf.chirp(1e8, 3e8, 1e-6) >> 0.2e-6
    # a generic factor in the same piece: the piece is rebuilt without the fused chirp (general kernel)
    w = c * (wf.sinc(2e6) >> 0.7e-6)
    plan, _ = _check(w, GRID, want_kernel=None)
    assert 'lean' not in plan.kernel_name() and plan.info.n_generic > 0
    # tlist mode: libm everywhere
    t = np.sort(np.random.default_rng(3).uniform(0, 4e-6, 50001))
    prog = _flatten.flatten([c])
    got = _engine.Plan(prog, t=t).run_host(np.float64)[0]
    assert np.max(np.abs(got - c_oracle.eval_tlist(prog, t)[0])) <= 1e-11
    # switched off: same numbers through the direct tier
    os.environ['WFK_DISABLE_CHIRP'] = '1'
    try:
        plan2, _ = _check(c, GRID, want_kernel=None)
        assert plan2.info.n_direct > 0
    finally:
        del os.environ['WFK_DISABLE_CHIRP']
    # the other families keep their own instantiations
    assert _engine.Plan(_flatten.flatten([wl.c2_channel(wf)]), grid=_flatten.grid_from_desc(wl.c2_grid(10**6))).kernel_name() \
        == 'wfk_sample_lean<double,false,16,false,0>'
    flat = wf.square(300e-9, edge=40e-9) * wf.cos(2 * pi * 1e8) >> 1e-6
    assert _engine.Plan(_flatten.flatten([flat]), grid=_flatten.grid_from_desc(GRID)).kernel_name().endswith(',false,1>')


def test_reference_chirp_cases_and_goldens():
    SAMPLES = golden_io.npz('samples.npz')
    for name in ('chirp_lin', 'chirp_exp', 'chirp_hyp'):
        build, grid = cases.CASES[name]
        w = build(wf)
        got = w(wl.make_grid(grid))
        want = SAMPLES[name + '.y']
        assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.abs(want).max())
    # fine grid of the same linear chirp: the fused op, against the reference's closed form
    t = np.linspace(0, 10, 4_000_001)
    got = wf.chirp(1, 2, 10, 4, 'linear')(t)
    want = np.where((t >= 0) & (t < 10), np.sin(4 + 2 * np.pi * ((2 - 1) / (2 * 10) * t**2 + 1 * t)), 0.0)
    assert np.max(np.abs(got - want)) <= 1e-9


def test_exponential_and_hyperbolic_chirps_at_awg_rates_are_short_tier_multipliers(monkeypatch):
    """chirp(type='exponential' | 'hyperbolic') pulses of 30-60 samples at 2 GS/s: the phase has no recurrence form, but as
    the closing MULTIPLIER of what the pulse's other factors fuse to they stay on the short tier (family 4: one inline
    exponential step / one log + one sine per sample) instead of the term interpreter with device libm
    (reference EXPONENTIALCHIRP / HYPERBOLICCHIRP, _waveform.pyx:326-332)."""
    rate, n = 2e9, 40_000
    grid = ('arange', 0.0, n / rate, 1.0 / rate)
    rng = np.random.default_rng(5)
    for kind in ('exponential', 'hyperbolic'):
        ws = []
        for k in range(int(n / rate / wl.SPAN)):
            body = wf.chirp(rng.uniform(5e7, 1e8), rng.uniform(1.5e8, 3e8), wl.SPAN, rng.uniform(0, 6), type=kind)
            env = wf.cosPulse(wl.SPAN) if k % 3 else (wf.gaussian(wl.W) >> (wl.SPAN / 2))
            ws.append(rng.uniform(0.2, 1) * body * env >> (k * wl.SPAN))
        w = wl._tree_sum(ws)
        plan, err = _check([w, 0.5 * w + 0.1], grid, tol=1e-9, want_kernel='wfk_sample_short<double,false,false,16,4>')
        assert plan.info.n_generic == 0 and plan.info.n_direct == 0
        assert err <= 5e-11, err                      # (measured; the bound of the grid tiers is 1e-9)
    # complex amplitudes, and the switch that sends such pieces back to the term interpreter
    wc = (0.7 - 0.2j) * w
    _check(wc, grid, want_kernel='wfk_sample_short<double,false,false,16,4>', cplx=True)
    monkeypatch.setenv('WFK_NO_SHORT_XCHIRP', '1')
    off = _engine.Plan(_flatten.flatten([w]), grid=_flatten.grid_from_desc(grid))
    assert '16,4>' not in off.kernel_name()
    ora = c_oracle.eval_grid(_flatten.flatten([w]), _flatten.grid_from_desc(grid))
    assert np.max(np.abs(off.run_host(np.float64) - ora)) <= 1e-9


@pytest.mark.parametrize('t0', [3e-3, -2e-3, 1e-2])
@pytest.mark.parametrize('rate,n', [(2.4e9, 10477), (2e9, 30000), (1e8, 400000)])
def test_linear_chirps_far_from_t_zero(t0, rate, n):
    """Linear chirps milliseconds from t = 0 (tools/chirp_awg_soak.py, seed 5113: 4.6e-9 of peak before the fix).  The
    fusion pass had expanded the quadratic phase about t' = 0 -- K t'^2 + W t' - Psi, three terms of 1e10 rad each 3 ms
    out that cancel to the 80-bit rounding of the host's long double, 5e-9 rad; it is expanded about the chirp's own
    shift now (FceGroup::corg).  Fine-grid pieces (lean family 2) and AWG-rate pieces (short tier), under squares,
    cosine pulses and an extra carrier, real and complex amplitudes."""
    rng = np.random.default_rng(int(abs(t0) * 1e6) + n)
    span = n / rate
    w = wf.zero()
    npulse = 24
    slot = span / npulse
    for k in range(npulse):
        width = slot * rng.uniform(0.5, 0.9)
        f0 = rng.uniform(2e7, 1.2e8) * rate / 2.4e9
        ch = wf.chirp(f0, f0 * rng.uniform(1.3, 2.5), width, rng.uniform(0, 6))
        env = k % 3
        p = ch * wf.square(width * 0.9) if env == 0 else (ch * wf.cosPulse(width) if env == 1 else ch * wf.cosPulse(width) * wf.cos(2 * pi * rng.uniform(-5e7, 5e7) * rate / 2.4e9))
        amp = rng.uniform(0.2, 1) if k % 4 else complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        w = w + ((amp * p) >> (t0 + (k + 0.5) * slot))
    prog = _flatten.flatten([w])
    g = _flatten.grid_from_desc(('arange', t0, t0 + span, 1.0 / rate))
    plan = _engine.Plan(prog, grid=g)        # (fused where max |f| x ulp(t) allows it, device libm on the exact times elsewhere)
    ora = c_oracle.eval_grid(prog, g, True)
    pk = max(1.0, float(np.abs(ora).max()))
    # (the reference's own rounding of t - shift is inside this: a fused chirp is admitted while max |f| x ulp(t) <= 2.5e-10)
    assert np.max(np.abs(plan.run_host(np.complex128) - ora)) <= 6e-10 * pk, plan.kernel_name(np.complex128)
    assert np.max(np.abs(plan.run_host(np.complex64) - ora)) <= FP32_TOL * pk
