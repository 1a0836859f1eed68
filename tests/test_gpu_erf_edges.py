"""Flat-top pulses with erf edges (square(width, edge), reference waveform.py step()/square()) on
oversampled grids: the edge pieces run FUSED -- the erf is a closing multiplier of the piece's
carrier groups, advanced per sample by the midpoint series of its own integral (fce_erfmul) and
reseeded with libm every few tiles.  Everything here is compared with the plain-C oracle, which
calls erf() per sample."""
import os

import numpy as np
import pytest

from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

pytestmark = pytest.mark.gpu


def _tones(rng, nt, lo=-300e6, hi=300e6):
    out = None
    for _ in range(nt):
        t = rng.uniform(0.05, 0.3) * wf.cos(2 * np.pi * rng.uniform(lo, hi), rng.uniform(0, 6))
        out = t if out is None else out + t
    return out


def _check(chans, grid, lean=True, tol64=2e-12, tol32=FP32_TOL, cplx=False):
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    want = c_oracle.eval_grid(prog, g, True) if cplx else c_oracle.eval_grid(prog, g)
    pk = max(1.0, float(np.abs(want).max()))
    plan = _engine.Plan(prog, grid=g)
    if lean:
        assert plan.info.n_generic == 0, plan.kernel_name()
        assert 'lean' in plan.kernel_name() and ' + ' not in plan.kernel_name(), plan.kernel_name()
    got = plan.run_host(np.complex128 if cplx else np.float64)
    assert np.max(np.abs(got - want)) <= tol64 * pk, np.max(np.abs(got - want))
    got32 = plan.run_host(np.complex64 if cplx else np.float32)
    assert np.max(np.abs(got32 - want)) <= tol32 * pk
    # the same program with the erf on the per-sample libm path agrees too (A/B switch)
    os.environ['WFK_DISABLE_ERFMUL'] = '1'
    try:
        ref = _engine.Plan(prog, grid=g)
        assert ref.info.n_generic > 0
        assert np.max(np.abs(ref.run_host(np.complex128 if cplx else np.float64) - want)) <= max(1e-11, tol64) * pk
    finally:
        del os.environ['WFK_DISABLE_ERFMUL']
    return plan


def test_bare_flat_top():
    """square(width, edge) alone: each edge piece = one constant group + the closing erf op."""
    w = wf.zero()
    for k in range(4):
        w = w + 0.7 * (wf.square(30e-9, edge=4e-9) >> (20e-9 + k * 50e-9))
    _check([w, -1.5 * w + 0.25], ('linspace', 0.0, 220e-9, 1_000_003, False))


@pytest.mark.parametrize('nt', [1, 3, 10])
def test_flat_top_under_tones(nt):
    """the multiplexed-readout shape: 0.5 (1 + erf) * sum of carriers -> carriers once, erf once"""
    rng = np.random.default_rng(5 + nt)
    chans = []
    for c in range(3):
        w = wf.zero()
        tones = _tones(rng, nt)
        for k in range(5):
            w = w + ((wf.square(30e-9, edge=4e-9) >> ((k + 0.5) * 60e-9 + c * 1.3e-9)) * tones)
        chans.append(w)
    _check(chans, ('linspace', 0.0, 300e-9, 1_000_000, False))   # (nt = 10: eleven ops per edge piece)


def test_mixing_with_drag_on_flat_top():
    """mixing(square(edge), DRAG): I = env cos - k env' sin; env' of an erf edge is a Gaussian --
    a modulated carrier group plus an unmodulated one (no twin): out = S0 + erf * S1"""
    env = wf.square(40e-9, edge=6e-9) >> 60e-9
    I, Q = wf.mixing(env, freq=137e6, phase=0.4, DRAGScaling=3e-10)
    _check([I, Q, I + 0.3 * Q], ('linspace', 0.0, 120e-9, 700_001, False))


def test_flat_top_times_gaussian_and_neighbours():
    """the modulated group carries a Gaussian envelope; another member overlaps the edge piece"""
    rng = np.random.default_rng(3)
    a = (wf.square(30e-9, edge=5e-9) * wf.gaussian(60e-9) * _tones(rng, 2)) >> 50e-9
    b = (wf.gaussian(20e-9) >> 36e-9) * wf.cos(2 * np.pi * 80e6)
    _check([a + b, wf.WaveVStack([a, b])], ('linspace', 0.0, 100e-9, 600_000, False))


def test_complex_amplitudes_under_erf_edge():
    rng = np.random.default_rng(8)
    w = ((wf.square(30e-9, edge=4e-9) >> 40e-9) * _tones(rng, 3)) * (0.6 - 0.8j) + \
        ((wf.square(20e-9, edge=4e-9) >> 110e-9) * _tones(rng, 2)) * (0.1 + 0.9j)
    _check([w], ('linspace', 0.0, 160e-9, 800_000, False), cplx=True)


def test_far_from_origin_edges_with_corrected_carriers():
    """a flat-top pulse 1 ms from t = 0 under 300 MHz carriers: the corrected-carrier lean variant
    with the closing erf op (times far from the origin: erf argument from (x - shift) / sigma)"""
    rng = np.random.default_rng(21)
    t0 = 1.0e-3
    w = wf.zero()
    tones = _tones(rng, 3, 250e6, 350e6)
    for k in range(3):
        w = w + ((wf.square(30e-9, edge=4e-9) >> (t0 + (k + 0.5) * 60e-9)) * tones)
    _check([w], ('linspace', t0, t0 + 180e-9, 600_000, False), tol64=1e-9)


def test_clip_offset_and_accumulate_on_edge_pieces():
    w = 1.2 * (wf.square(30e-9, edge=4e-9) >> 40e-9) * wf.cos(2 * np.pi * 50e6) + 0.1
    w.min, w.max = -0.4, 0.9
    chans = [w]
    grid = ('linspace', 0.0, 80e-9, 400_000, False)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    want = c_oracle.eval_grid(prog, g)
    plan = _engine.Plan(prog, grid=g)
    assert plan.info.n_generic == 0
    assert np.max(np.abs(plan.run_host(np.float64) - want)) <= 2e-12
    buf = _engine.DeviceBuffer(want.size * 8)
    buf.upload(np.full(want.shape, 0.5))
    plan.launch(buf.ptr, g.n, _engine.OUT_F64, accumulate=True)
    _engine.sync()
    assert np.max(np.abs(buf.download(want.shape, np.float64) - (want + 0.5))) <= 2e-12
    buf.close()


def test_not_eligible_shapes_stay_exact():
    """overlapping edges (two different erfs in one piece) and a squared erf: the erf stays a per-sample
    libm factor (wholly or for the second one).  A grid too coarse for the step series of the lean
    kernel goes to the short-piece tier, whose closing op calls erf per sample."""
    narrow = wf.square(6e-9, edge=4e-9) >> 30e-9                      # width < 2 edge: both erfs in the middle piece
    sq = (wf.square(30e-9, edge=4e-9) >> 30e-9)
    for chans, grid, short in (([narrow * wf.cos(2 * np.pi * 90e6)], ('linspace', 0.0, 60e-9, 500_000, False), False),
                               ([sq * sq], ('linspace', 0.0, 60e-9, 500_000, False), False),
                               ([sq * wf.cos(2 * np.pi * 90e6)], ('linspace', 0.0, 60e-9, 3_001, False), True)):
        prog = _flatten.flatten(chans)
        g = _flatten.grid_from_desc(grid)
        want = c_oracle.eval_grid(prog, g)
        plan = _engine.Plan(prog, grid=g)
        if short:
            assert plan.kernel_name().startswith('wfk_sample_short<')
        else:
            assert plan.info.n_generic > 0
        assert np.max(np.abs(plan.run_host(np.float64) - want)) <= 1e-11


def test_step_series_limit_of_the_grid():
    """h = 64 * step / sigma right at the admission limit (0.09) and just past it"""
    edge = 4e-9
    sigma = edge / 5
    for h, fused in ((0.0899, True), (0.0905, False)):
        step = h * sigma / 64
        n = 200_000
        w = (wf.square(30e-9, edge=edge) >> 50e-9) * wf.cos(2 * np.pi * 200e6, 0.3)
        w = w >> (n * step / 2 - 50e-9)                              # centre the pulse in the window
        prog = _flatten.flatten([w])
        g = _flatten.grid_from_desc(('linspace', 0.0, n * step, n, False))
        plan = _engine.Plan(prog, grid=g)
        assert (plan.info.n_generic == 0) == fused
        want = c_oracle.eval_grid(prog, g)
        assert np.max(np.abs(plan.run_host(np.float64) - want)) <= 3e-12
