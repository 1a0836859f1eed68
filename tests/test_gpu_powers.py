"""Powers of factors inside a term (`wav ** n`, reference _waveform.pyx:29-127: `pow` keeps the power on the basic
function; evaluation `_apply(...) ** n`, :130-152).  Powers of a Gaussian are Gaussians and cos^2 / cos^3 are sums of
carriers: such terms run on the fused tiers (lean kernel / short tier / pointwise time lists) instead of device libm."""
import numpy as np
import pytest

from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl

pytestmark = pytest.mark.gpu
W = 30e-9


def chans():
    rng = np.random.default_rng(3)

    def train(mk, n=40):
        return wl._tree_sum([rng.uniform(0.3, 1) * mk(k) >> ((k + 0.5) * 2 * W) for k in range(n)])
    return [
        train(lambda k: (wf.gaussian(W) ** 2) * wf.cos(2e9, 0.1 * k)),
        train(lambda k: wf.gaussian(W) ** 0.5),
        train(lambda k: (wf.gaussian(W) ** 3) * (wf.cos(1.5e9, 0.3 * k) ** 2)),
        train(lambda k: (wf.cos(9e8, 0.2 * k) ** 3) * wf.square(W)),
        train(lambda k: (wf.cosPulse(W) ** 2) * wf.cos(2e9)),
        train(lambda k: ((0.5 + 0.2j) * wf.gaussian(W) ** 2) * wf.cos(1e9) ** 2),
    ]


@pytest.mark.parametrize('grid,tier,tol', [(('linspace', 0.0, 80 * W, 1_200_000, False), 'wfk_sample_lean<', 1e-12),
                                         (('arange', 0.0, 80 * W, 0.5e-9), 'wfk_sample_short<', 5e-11)])
def test_powers_of_gaussians_and_cosines_are_fused(grid, tier, tol):
    cs = chans()
    prog = _flatten.flatten(cs)
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(prog, grid=g)
    assert plan.kernel_name(np.complex128).startswith(tier), plan.kernel_name(np.complex128)
    assert plan.info.n_generic == 0 and plan.info.n_direct == 0
    ref = c_oracle.eval_grid(prog, g, True)
    pk = float(np.abs(ref).max())
    assert np.max(np.abs(plan.run_host(np.complex128) - ref)) <= tol * pk
    assert np.max(np.abs(plan.run_host(np.float64) - ref.real)) <= tol * pk
    assert np.max(np.abs(plan.run_host(np.complex64) - ref)) <= FP32_TOL * pk
    t = c_oracle.grid_values(g)[:200_000]
    t = np.sort(t + np.random.default_rng(0).normal(size=len(t)) * (t[1] - t[0]) * 0.3)
    tl = _engine.Plan(prog, t=t)
    assert tl.info.n_generic == 0
    assert np.max(np.abs(tl.run_host(np.complex128) - c_oracle.eval_tlist(prog, t, want_complex=True))) <= 1e-11 * pk


def test_what_stays_on_the_exact_path():
    """Negative and zero powers, powers of other primitives, cos^4: device libm as before, same numbers."""
    grid = ('linspace', 0.0, 4 * W, 200_001, True)
    g = _flatten.grid_from_desc(grid)
    for w in ((wf.gaussian(W) ** -1) * wf.square(W) >> 2 * W, (wf.cos(2e9) ** 4) * wf.square(W) >> 2 * W,
              (wf.mollifier(W) ** 2) >> 2 * W, (wf.exp(-1 / W) ** 2.5) * (wf.cos(2e9) ** 2) * wf.square(W) >> 2 * W):
        prog = _flatten.flatten([w])
        plan = _engine.Plan(prog, grid=g)
        ref = c_oracle.eval_grid(prog, g)
        fin = np.isfinite(ref)
        got = plan.run_host(np.float64)
        assert np.max(np.abs(got[fin] - ref[fin])) <= 1e-9 * max(1.0, np.abs(ref[fin]).max())
