"""Exponential factors in the fused tier: EXP (decays, complex exponentials), COSH / SINH (the
reference's coshPulse, waveform.py coshPulse) run as carrier-envelope ops whose envelope state is
exp(a (t - ref)) advanced by a constant ratio, or -- under a Gaussian -- as that Gaussian with its
centre moved.  Against the plain-C oracle (libm exp / cosh / sinh per sample)."""
import os

import numpy as np
import pytest

from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

pytestmark = pytest.mark.gpu
GRID = ('linspace', 0.0, 120e-9, 400_001, False)


def _check(chans, grid=GRID, cplx=False, tol64=1e-12, tol32=FP32_TOL, fused=True, exp_ab=True):
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    want = c_oracle.eval_grid(prog, g, True) if cplx else c_oracle.eval_grid(prog, g)
    pk = max(1.0, float(np.abs(want).max()))
    plan = _engine.Plan(prog, grid=g)
    if fused:
        assert plan.info.n_generic == 0 and 'lean' in plan.kernel_name(), plan.kernel_name()
    got = plan.run_host(np.complex128 if cplx else np.float64)
    assert np.max(np.abs(got - want)) <= tol64 * pk, np.max(np.abs(got - want)) / pk
    got32 = plan.run_host(np.complex64 if cplx else np.float32)
    assert np.max(np.abs(got32 - want)) <= tol32 * pk
    os.environ['WFK_DISABLE_EXPFUSE'] = '1'          # A/B: the same program on the per-sample libm path
    try:
        ref = _engine.Plan(prog, grid=g)
        assert ref.info.n_generic > 0 or not exp_ab
        assert np.max(np.abs(ref.run_host(np.complex128 if cplx else np.float64) - want)) <= 1e-11 * pk
    finally:
        del os.environ['WFK_DISABLE_EXPFUSE']
    return plan


def test_exponential_decay_and_growth():
    decay = (wf.square(80e-9) >> 50e-9) * (wf.exp(-1 / 30e-9) >> 10e-9)
    growth = (wf.square(50e-9) >> 60e-9) * (wf.exp(4e7) >> 60e-9) * 0.3
    _check([decay, growth, decay - 2 * growth + 0.1])


@pytest.mark.parametrize('eps,plateau', [(1.0, 0.0), (3.0, 0.0), (6.0, 25e-9)])
def test_cosh_pulse(eps, plateau):
    p = wf.coshPulse(60e-9, eps=eps, plateau=plateau) >> 60e-9
    I, Q = wf.mixing(p, freq=140e6, phase=0.7, DRAGScaling=2e-10)
    _check([p, p * wf.cos(2 * np.pi * 100e6, 0.3), I, Q])


def test_gaussian_times_exponential_moves_the_centre():
    g = (wf.gaussian(40e-9) >> 50e-9) * (wf.exp(2e7) >> 50e-9)
    h = (wf.gaussian(30e-9) >> 70e-9) * (wf.exp(-5e7) >> 20e-9) * wf.cos(2 * np.pi * 210e6)
    _check([g, h, g + h])


def test_sinh_cosh_products_and_powers():
    w = (wf.square(40e-9) >> 50e-9)
    a = w * (wf.sinh(3e7) >> 50e-9)
    b = w * (wf.cosh(2e7) >> 45e-9) * (wf.exp(-1e7) >> 50e-9)         # cosh x exp: two exponentials
    c = w * (wf.exp(1.5e7) >> 50e-9) ** 2                               # a power of an exponential
    _check([a, b, c, a + b - c])


def test_complex_exponential_carrier():
    z = (wf.square(60e-9) >> 50e-9) * wf.exp(-2e7 + 2j * np.pi * 150e6)
    _check([z], cplx=True)


def test_far_from_origin_and_float_range():
    """a decay referenced 1 ms away: exp(a t' + b) with |a t'| ~ 3e4 on its own, finite only as a
    whole -- the envelope is referenced to the piece; a float launch keeps the state in double where
    the envelope leaves float's range"""
    t0 = 1.0e-3
    d = (wf.square(80e-9) >> (t0 + 50e-9)) * (wf.exp(-1 / 30e-9) >> (t0 + 10e-9))
    _check([d], ('linspace', t0, t0 + 120e-9, 400_001, False), tol64=1e-11)
    steep = (wf.square(100e-9) >> 55e-9) * (wf.exp(1.2e9) >> 105e-9)     # 120 e-foldings across the piece
    _check([steep], tol64=1e-11)


def test_out_of_range_exponentials_stay_on_libm():
    """an exponential that overflows double inside the piece is not fused (the reference gives inf
    there too)"""
    huge = (wf.square(100e-9) >> 55e-9) * (wf.exp(1e11) >> 5e-9)
    prog = _flatten.flatten([huge])
    g = _flatten.grid_from_desc(GRID)
    plan = _engine.Plan(prog, grid=g)
    assert plan.info.n_generic > 0
    want = c_oracle.eval_grid(prog, g)
    got = plan.run_host(np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    assert np.max(np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))) <= 1e-11


@pytest.mark.parametrize('d', [1, 2, 3])
def test_gaussian_derivatives_fuse(d):
    """gaussian(width, d=n) = (-1/s)^n H_n(u/s) exp(-(u/s)^2) (D_GAUSSIAN, reference _waveform.pyx:298-300):
    a polynomial of degree n times the Gaussian envelope -- one fused op; alone, under a carrier,
    times t, through mixing(); d = 4 stays on libm"""
    W = 40e-9
    g = wf.gaussian(W, d=d) >> 60e-9
    scale = (W / 3.33) ** d                       # bring the derivative back to O(1)
    I, Q = wf.mixing(scale * g, freq=120e6, phase=0.3, DRAGScaling=1e-10 if d < 3 else None)
    chans = [scale * g, scale * g * wf.cos(2 * np.pi * 150e6, 0.5), I, Q]
    if d < 3:
        chans.append(scale * g * (wf.poly([0.0, 1e8]) >> 60e-9))
    _check(chans, exp_ab=False)        # (no exponential factor here: the EXPFUSE switch changes nothing)


def test_high_gaussian_derivative_stays_on_libm():
    g = (40e-9 / 3.33) ** 4 * (wf.gaussian(40e-9, d=4) >> 60e-9)
    prog = _flatten.flatten([g])
    gr = _flatten.grid_from_desc(GRID)
    plan = _engine.Plan(prog, grid=gr)
    assert plan.info.n_generic > 0
    want = c_oracle.eval_grid(prog, gr)
    assert np.max(np.abs(plan.run_host(np.float64) - want)) <= 1e-11 * max(1.0, np.abs(want).max())
