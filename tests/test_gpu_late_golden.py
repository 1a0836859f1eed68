"""The late round-2 fusions against vectors of the REAL reference (tests/golden/late.npz, generated
by oracle/make_golden.py from tests/cases.py LATE_CASES): flat tops with erf edges, 3- and 10-tone
readout, complex amplitudes, mixing(...DRAG...) of a flat top, a vstack with an overlapping neighbour,
cosh pulses, exponential factors, a flat top 1 ms from the origin -- on 600 k - 800 k point grids,
through the lean kernel, compared at a strided subset plus the samples around every piece edge."""
import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten, workloads as wl

pytestmark = pytest.mark.gpu
LATE = golden_io.npz('late.npz')


@pytest.mark.parametrize('name', sorted(cases.LATE_CASES))
def test_fused_tier_matches_reference_vectors(name):
    build, grid = cases.LATE_CASES[name]
    w = build(wf)
    pick, want, sums = LATE[name + '.pick'], LATE[name + '.y'], LATE[name + '.sum']
    cplx = np.iscomplexobj(want)
    prog = _flatten.flatten([w])
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(prog, grid=g)
    assert plan.info.n_generic == 0 and 'lean' in plan.kernel_name(), plan.kernel_name()
    got = plan.run_host(np.complex128 if cplx else np.float64)[0]
    pk = max(1.0, float(np.abs(want).max()))
    far = abs(grid[1]) > 1e-5
    assert np.max(np.abs(got[pick] - want)) <= (1e-9 if far else 2e-12) * pk
    # whole-array sums of the reference run: sum, sum |y|, max |y|
    assert abs(got.sum() - sums[0]) <= 1e-9 * max(1.0, abs(sums[1]))
    assert abs(np.abs(got).sum() - sums[1].real) <= 1e-9 * max(1.0, abs(sums[1]))
    assert abs(np.abs(got).max() - sums[2].real) <= (1e-9 if far else 1e-11) * pk
    # the drop-in call on the same grid (recognised as a grid, same plan underneath)
    t = wl.make_grid(grid)
    y = np.asarray(w(t))
    assert np.max(np.abs(y[pick] - want)) <= (1e-9 if far else 2e-12) * pk
    # float launch
    g32 = plan.run_host(np.complex64 if cplx else np.float32)[0]
    assert np.max(np.abs(g32[pick] - want)) <= FP32_TOL * pk
