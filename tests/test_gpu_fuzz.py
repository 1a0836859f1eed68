"""Differential fuzzing: random pulse scripts x random grids, HIP (grid mode: fusion pass,
lean / general kernels, state carry; tlist mode: device libm) against the C oracle evaluating
the same flattened program.  Seeded; every failure prints its seed."""
import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

pytestmark = pytest.mark.gpu


random_pulse = lambda rng, scale: cases.random_pulse(wf, rng, scale)      # noqa: E731
random_channel = lambda rng: cases.random_channel(wf, rng)                 # noqa: E731


@pytest.mark.parametrize('seed', range(160))
def test_random_script(seed):
    rng = np.random.default_rng(10_000 + seed)
    ch, grid = random_channel(rng)
    prog = _flatten.flatten([ch])
    g = _flatten.grid_from_desc(grid)
    ora = c_oracle.eval_grid(prog, g)[0]
    pk = max(1.0, float(np.max(np.abs(ora)))) if ora.size else 1.0
    plan = _engine.Plan(prog, grid=g)
    got = plan.run_host(np.float64)[0]
    assert np.all(np.isfinite(got)) or not np.all(np.isfinite(ora))
    assert np.max(np.abs(got - ora), initial=0.0) <= 1e-9 * pk, (seed, plan.info.n_fused,
                                                               plan.info.n_generic)
    got32 = plan.run_host(np.float32)[0].astype(np.float64)
    assert np.max(np.abs(got32 - ora), initial=0.0) <= FP32_TOL * pk, seed
    if seed % 4 == 0:
        t = c_oracle.grid_values(g)
        tl = _engine.Plan(prog, t=t).run_host(np.float64)[0]
        assert np.max(np.abs(tl - ora), initial=0.0) <= 1e-11 * pk, seed


def test_random_batches():
    rng = np.random.default_rng(77)
    for _ in range(6):
        scale = 10.0**rng.uniform(-9, -6)
        chans = []
        for _c in range(int(rng.integers(2, 12))):
            ps = [random_pulse(rng, scale) for _ in range(int(rng.integers(1, 5)))]
            chans.append(wf.WaveVStack(ps) if rng.random() < 0.5 else sum(ps[1:], ps[0]))
        a = scale * rng.uniform(-12, -8)
        grid = ('linspace', a, a + scale * rng.uniform(10, 30), int(rng.integers(1000, 200000)),
                False)
        prog = _flatten.flatten(chans)
        g = _flatten.grid_from_desc(grid)
        ora = c_oracle.eval_grid(prog, g)
        got = _engine.Plan(prog, grid=g).run_host(np.float64)
        assert np.max(np.abs(got - ora)) <= 1e-9 * max(1.0, np.abs(ora).max())


# ---- the same random scripts against outputs of the REAL reference (tests/golden/fuzz.npz,
# oracle/make_golden.py): the reference evaluated seeds 0..FUZZ_GOLD-1 on a reduced grid ----
FUZZ = golden_io.npz('fuzz.npz')


@pytest.mark.parametrize('seed', range(cases.FUZZ_GOLD))
def test_random_script_vs_reference_golden(seed):
    ch, grid = cases.fuzz_golden_case(wf, seed)
    want = FUZZ[f'{seed}.y']
    prog = _flatten.flatten([ch])
    g = _flatten.grid_from_desc(grid)
    pk = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
    got = _engine.Plan(prog, grid=g).run_host(np.float64)[0]
    assert np.max(np.abs(got - want), initial=0.0) <= 1e-9 * pk, seed
    t = c_oracle.grid_values(g)
    tl = ch(t)                                   # drop-in __call__ (tlist mode)
    assert np.max(np.abs(np.real(tl) - want), initial=0.0) <= 1e-9 * pk, seed


@pytest.mark.parametrize('seed', range(60))
def test_far_from_origin(seed):
    """Pulses and grid 1e2..1e7 spans away from t = 0: NumPy's grid values are then visibly
    rounded (one ulp of |t|), the reference evaluates AT those rounded times, and a fast carrier
    turns the rounding into phase.  The host takes such factors off the uniform-grid fast paths
    (wfk_compile.cpp: grid_jitter / rate_safe); before that guard 289 of 1500 such scripts were
    off by up to 4e-6."""
    chans, grid = cases.far_from_origin_case(wf, seed)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    ora = c_oracle.eval_grid(prog, g)
    pk = max(1.0, float(np.max(np.abs(ora))))
    got = _engine.Plan(prog, grid=g).run_host(np.float64)
    assert np.max(np.abs(got - ora)) <= 1e-9 * pk, seed


@pytest.mark.parametrize('seed', range(cases.FAR_GOLD))
def test_far_from_origin_vs_reference_golden(seed):
    chans, grid = cases.far_golden_case(wf, seed)
    g = _flatten.grid_from_desc(grid)
    got = _engine.Plan(_flatten.flatten(chans), grid=g).run_host(np.float64)
    t = c_oracle.grid_values(g)
    for c, w in enumerate(chans):
        want = FUZZ[f'far{seed}.{c}']
        pk = max(1.0, float(np.max(np.abs(want))))
        assert np.max(np.abs(got[c] - want)) <= 1e-9 * pk, (seed, c)
        assert np.max(np.abs(np.real(w(t)) - want)) <= 1e-9 * pk, (seed, c)     # drop-in, tlist


def test_float_cancellation_case_stays_inside_the_contract():
    """tools/fuzz_soak.py awg, seed 306405 (the one float-output exception of 54 000 round-4 soak scripts): a piece whose
    terms cancel to ~1e-3 of their size.  Evaluated by the general kernel's float build it measured 9.2e-5 of the peak
    (float accumulators lose the cancellation factor x 6e-8); float outputs of that tier now run under double arithmetic
    (wfk_sample_wide) and round once at the store."""
    rng = np.random.default_rng(10_000 + 306405)
    ch, grid = cases.random_awg_channel(wf, rng)
    prog = _flatten.flatten([ch])
    g = _flatten.grid_from_desc(grid)
    ora = c_oracle.eval_grid(prog, g)[0]
    pk = max(1.0, float(np.abs(ora).max()))
    plan = _engine.Plan(prog, grid=g)
    assert np.max(np.abs(plan.run_host(np.float64)[0] - ora)) <= 1e-9 * pk
    assert 'wfk_sample_wide<' in plan.kernel_name(np.float32)
    e32 = float(np.max(np.abs(plan.run_host(np.float32)[0] - ora)))
    assert e32 <= 2e-7 * pk, e32          # one rounding to float
