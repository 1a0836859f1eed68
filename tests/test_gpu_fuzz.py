"""Differential fuzzing: random pulse scripts x random grids, HIP (grid mode: fusion pass,
lean / general kernels, state carry; tlist mode: device libm) against the C oracle evaluating
the same flattened program.  Seeded; every failure prints its seed."""
import numpy as np
import pytest

import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

pytestmark = pytest.mark.gpu


def random_pulse(rng, scale):
    kind = rng.integers(0, 9)
    w = scale * rng.uniform(0.5, 6.0)
    if kind == 0:
        p = wf.gaussian(w)
    elif kind == 1:
        p = wf.cosPulse(w, plateau=scale * rng.uniform(0, 2) * (rng.random() < 0.4))
    elif kind == 2:
        p = wf.square(w, edge=w * rng.uniform(0.05, 0.3) * (rng.random() < 0.6),
                      type=str(rng.choice(['erf', 'cos', 'linear'])))
    elif kind == 3:
        p = wf.gaussian(w, plateau=scale * rng.uniform(0.1, 2))
    elif kind == 4:
        p = wf.drag(rng.uniform(-2, 2) / scale, w, delta=rng.uniform(-0.1, 0.1) / scale,
                    block_freq=None if rng.random() < 0.3 else rng.uniform(1, 3) / scale,
                    phase=rng.uniform(0, 6), t0=-w / 2)
    elif kind == 5:
        p = wf.coshPulse(w, eps=rng.uniform(0.5, 3))
    elif kind == 6:
        p = wf.gaussian(w) * wf.poly([rng.uniform(-1, 1), rng.uniform(-1, 1) / scale,
                                      rng.uniform(-1, 1) / scale**2])
    elif kind == 7:
        p = wf.D(wf.gaussian(w)) * (scale * 0.3)
    else:
        p = wf.mollifier(w)
    if rng.random() < 0.8:
        f = rng.uniform(-3, 3) / scale
        no_d = kind == 4    # the DRAG primitive has no derivative rule (nor upstream)
        I, Q = wf.mixing(p, freq=f, phase=rng.uniform(0, 6),
                         DRAGScaling=None if (no_d or rng.random() < 0.4)
                         else rng.uniform(-0.05, 0.05) * scale,
                         block_freq=None if (no_d or rng.random() < 0.8)
                         else f + rng.uniform(0.5, 2) / scale)
        p = I if rng.random() < 0.5 else Q
    if rng.random() < 0.3:
        p = p * wf.cos(rng.uniform(0.5, 4) / scale, rng.uniform(0, 6))
    return rng.uniform(0.1, 2.0) * (p >> (scale * rng.uniform(-8, 8)))


def random_channel(rng):
    scale = 10.0**rng.uniform(-9, 0)
    n = int(rng.integers(1, 7))
    pulses = [random_pulse(rng, scale) for _ in range(n)]
    if rng.random() < 0.4:
        ch = wf.WaveVStack(pulses)
        if rng.random() < 0.5:
            ch = (ch + rng.uniform(-0.5, 0.5)) >> (scale * rng.uniform(-1, 1))
    else:
        ch = pulses[0]
        for p in pulses[1:]:
            ch = ch + p
        if rng.random() < 0.2:
            ch = wf.cut(ch, min=-0.4, max=0.6)
    npts = int(rng.integers(1, 60000))
    a = scale * rng.uniform(-14, -6)
    b = a + scale * rng.uniform(2, 30)
    grid = (('linspace', a, b, npts, bool(rng.random() < 0.5)) if rng.random() < 0.7
            else ('arange', a, b, (b - a) / npts))
    return ch, grid


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
    assert np.max(np.abs(got32 - ora), initial=0.0) <= 5e-5 * pk, seed
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
