"""Pieces with many carriers (frequency-multiplexed pulses) and complex amplitudes: the paths this
exercises are the lean kernel with 5..10 ops per piece (LDS sized per plan), the general kernel
beyond that, the shared-envelope factoring (>= 4 carriers under one Gaussian: closing multiply
op), imaginary-part ops of complex amplitudes, and direct-factor value reuse across terms (erf
edges under several carriers).  All against the C oracle on the same flattened program."""
import numpy as np
import pytest

from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

pytestmark = pytest.mark.gpu
W = 20e-9


def tones(rng, nt, cplx=False):
    out = None
    for _ in range(nt):
        t = rng.uniform(0.05, 0.3) * wf.cos(2 * np.pi * rng.uniform(-300e6, 300e6), rng.uniform(0, 6))
        if cplx:
            t = t * complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        out = t if out is None else out + t
    return out


def channel(rng, nt, env, cplx=False):
    w = wf.zero()
    for k in range(6):
        if env == 'gauss':
            e = wf.gaussian(W)
        elif env == 'mixed':                      # two different envelopes in one piece: no factoring
            e = wf.gaussian(W) + 0.5 * wf.gaussian(0.7 * W)
        elif env == 'flat':
            e = wf.square(W, edge=0.2 * W)        # erf edges: generic terms sharing one direct factor
        else:
            e = wf.cosPulse(W)
        w = w + ((e >> ((k + 0.5) * 1.5 * W)) * tones(rng, nt, cplx))
    return w


@pytest.mark.parametrize('nt', [1, 3, 4, 5, 8, 10, 12])
@pytest.mark.parametrize('env', ['gauss', 'mixed', 'flat', 'cos'])
def test_multi_tone_pieces(nt, env):
    rng = np.random.default_rng(1000 * nt + len(env))
    chans = [channel(rng, nt, env) for _ in range(2)]
    grid = ('linspace', 0.0, 9 * W, 60001, False)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    ora = c_oracle.eval_grid(prog, g)
    pk = max(1.0, float(np.abs(ora).max()))
    plan = _engine.Plan(prog, grid=g)
    assert np.max(np.abs(plan.run_host(np.float64) - ora)) <= 1e-9 * pk
    assert np.max(np.abs(plan.run_host(np.float32) - ora)) <= FP32_TOL * pk
    tl = _engine.Plan(prog, t=c_oracle.grid_values(g)).run_host(np.float64)
    assert np.max(np.abs(tl - ora)) <= 1e-11 * pk


@pytest.mark.parametrize('nt', [1, 2, 4, 6])
def test_complex_amplitudes_fused(nt):
    rng = np.random.default_rng(77 + nt)
    chans = [channel(rng, nt, 'gauss', cplx=True), channel(rng, nt, 'cos', cplx=True)]
    grid = ('linspace', 0.0, 9 * W, 50001, True)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    ora = c_oracle.eval_grid(prog, g, True)
    pk = max(1.0, float(np.abs(ora).max()))
    plan = _engine.Plan(prog, grid=g)
    assert plan.info.n_generic == 0                  # complex amplitudes no longer leave the fused tier
    got = plan.run_host(np.complex128)
    assert np.max(np.abs(got - ora)) <= 1e-9 * pk
    assert np.max(np.abs(plan.run_host(np.complex64) - ora)) <= FP32_TOL * pk
    # a real-output launch of a complex channel keeps the real part (like WaveVStack's .real)
    assert np.max(np.abs(plan.run_host(np.float64) - ora.real)) <= 1e-9 * pk


def test_mixed_plan_lean_plateau_generic_edges():
    """A flat-top pulse with erf edges under several tones: the plateau and the gaps are lean pieces,
    the edges carry generic ERF terms.  The plan runs as TWO launches over disjoint pieces (lean
    kernel + general kernel); every sample is written exactly once."""
    import os
    rng = np.random.default_rng(11)
    tones = None
    for k in range(6):
        tone = rng.uniform(0.05, 0.1) * wf.cos(2 * np.pi * rng.uniform(-300e6, 300e6), rng.uniform(0, 6))
        tones = tone if tones is None else tones + tone
    chans = []
    for c in range(3):
        w = wf.zero()
        for k in range(7):
            w = w + ((wf.square(200e-9, edge=8e-9) >> ((k + 0.5) * 300e-9 + c * 1e-9)) * tones)
        chans.append(w)
    grid = ('linspace', 0.0, 7 * 300e-9, 210013, False)
    g = _flatten.grid_from_desc(grid)
    prog = _flatten.flatten(chans)
    plan = _engine.Plan(prog, grid=g)
    assert ' + ' in plan.kernel_name(), plan.kernel_name()          # lean + general
    want = c_oracle.eval_grid(prog, g)
    for dtype, tol in ((np.float64, 1e-11), (np.float32, FP32_TOL)):
        got = plan.run_host(dtype)
        assert np.max(np.abs(got - want)) <= tol
    # poisoned output buffer: every sample must be overwritten (no sample left to either kernel)
    buf = _engine.DeviceBuffer(want.size * 8)
    buf.upload(np.full(want.shape, np.nan))
    plan.launch(buf.ptr, g.n, _engine.OUT_F64)
    _engine.sync()
    got = buf.download(want.shape, np.float64)
    assert not np.isnan(got).any() and np.max(np.abs(got - want)) <= 1e-11
    # accumulate adds exactly once per sample
    plan.launch(buf.ptr, g.n, _engine.OUT_F64, accumulate=True)
    _engine.sync()
    assert np.max(np.abs(buf.download(want.shape, np.float64) - 2 * want)) <= 2e-11
    buf.close()
    os.environ['WFK_DISABLE_MIXED'] = '1'
    try:
        single = _engine.Plan(prog, grid=g)
        assert ' + ' not in single.kernel_name()
        assert np.max(np.abs(single.run_host(np.float64) - want)) <= 1e-11
    finally:
        del os.environ['WFK_DISABLE_MIXED']


@pytest.mark.parametrize('env', ['gauss', 'flat', 'none'])
def test_tone_loop_on_long_pieces(env, monkeypatch):
    """Runs of bare carriers go through the lean kernel's rolled tone loop (WFK_FCE_BANK): 20 tiles per chunk, the
    phasors advanced tile by tile without the periodic reseed of the envelope recurrences.  Long pieces (hundreds of
    tiles), real and complex amplitudes, float outputs; the same plan with the loop off gives the same numbers."""
    rng = np.random.default_rng({'gauss': 1, 'flat': 2, 'none': 3}[env])
    span = 4e-6

    def chan(cplx):
        w = wf.zero()
        for k in range(3):
            e = {'gauss': wf.gaussian(0.8 * span / 3), 'flat': wf.square(0.7 * span / 3, edge=0.05 * span / 3),
                 'none': wf.square(0.9 * span / 3)}[env]
            w = w + ((e >> ((k + 0.5) * span / 3)) * tones(rng, 7 if k != 1 else 10, cplx))
        return w
    chans = [chan(False), chan(True)]
    grid = ('linspace', 0.0, span, 2_000_000, False)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    ora = c_oracle.eval_grid(prog, g, True)
    pk = max(1.0, float(np.abs(ora).max()))
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name(np.complex128).startswith('wfk_sample_lean<double,true,16,false,1>'), plan.kernel_name(np.complex128)
    got = plan.run_host(np.complex128)
    assert np.max(np.abs(got - ora)) <= 1e-11 * pk
    assert np.max(np.abs(plan.run_host(np.float64) - ora.real)) <= 1e-11 * pk
    assert np.max(np.abs(plan.run_host(np.complex64) - ora)) <= FP32_TOL * pk
    monkeypatch.setenv('WFK_NO_BANK', '1')
    off = _engine.Plan(prog, grid=g).run_host(np.complex128)
    assert np.max(np.abs(off - ora)) <= 1e-11 * pk
    assert np.max(np.abs(off - got)) <= 1e-12 * pk
