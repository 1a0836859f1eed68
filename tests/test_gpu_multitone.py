"""Pieces with many carriers (frequency-multiplexed pulses) and complex amplitudes: the paths this
exercises are the lean kernel with 5..10 ops per piece (LDS sized per plan), the general kernel
beyond that, the shared-envelope factoring (>= 4 carriers under one Gaussian: closing multiply
op), imaginary-part ops of complex amplitudes, and direct-factor value reuse across terms (erf
edges under several carriers).  All against the C oracle on the same flattened program."""
import numpy as np
import pytest

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
    assert np.max(np.abs(plan.run_host(np.float32) - ora)) <= 5e-5 * pk
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
    assert np.max(np.abs(plan.run_host(np.complex64) - ora)) <= 5e-5 * pk
    # a real-output launch of a complex channel keeps the real part (like WaveVStack's .real)
    assert np.max(np.abs(plan.run_host(np.float64) - ora.real)) <= 1e-9 * pk
