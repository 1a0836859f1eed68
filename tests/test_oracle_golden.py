"""Pin both oracles (oracle/np_oracle.py, oracle/wfk_oracle.c) against golden
vectors produced by running the real reference (oracle/make_golden.py).

This also covers the host-side flattener: the C oracle consumes the
`wfk_program` it emits.  Tolerances: the C oracle uses glibc libm where NumPy
uses its own SIMD ufunc loops (<= few ulp apart), so values are compared at
1e-12 relative to the case's peak; integer piece indices must be identical."""
import numpy as np
import pytest

import cases
import golden_io
import waveforms_amd as wf
from oracle import c_oracle, np_oracle
from waveforms_amd import _flatten, workloads as wl

SAMPLES = golden_io.npz('samples.npz')
API = golden_io.npz('sample_api.npz')
FIR = golden_io.npz('fir.npz')


def close(got, want, rel=1e-12):
    scale = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
    assert got.shape == want.shape
    assert got.dtype == want.dtype
    err = np.max(np.abs(got - want)) if want.size else 0.0
    assert err <= rel * scale, (err, scale)


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_np_oracle_matches_reference(name):
    build, grid = cases.CASES[name]
    w = build(wf)
    t = wl.make_grid(grid)
    close(np_oracle.call(w, t), SAMPLES[name + '.y'], rel=1e-14)


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_c_oracle_matches_reference(name):
    build, grid = cases.CASES[name]
    w = build(wf)
    want = SAMPLES[name + '.y']
    prog = _flatten.flatten([w])
    g = _flatten.grid_from_desc(grid)
    t = wl.make_grid(grid)
    assert np.array_equal(c_oracle.grid_values(g), t)       # bit-exact grid
    cplx = want.dtype == np.complex128
    got = c_oracle.eval_grid(prog, g, cplx)[0]
    close(got, want)
    got_t = c_oracle.eval_tlist(prog, t, cplx)[0]
    assert np.array_equal(got, got_t, equal_nan=True)
    # integer parity: np.searchsorted indices of every member
    m0, m1 = prog.member_range(0)
    idx = np.concatenate([c_oracle.member_index(prog, m, grid=g)
                          for m in range(m0, m1)]) if m1 > m0 else np.zeros(0, np.int64)
    assert np.array_equal(idx, SAMPLES[name + '.idx'])


@pytest.mark.parametrize('name', sorted(cases.sos_cases()))
def test_sample_api_grid(name):
    build, start, stop, rate = cases.sos_cases()[name]
    w = build(wf)
    w.start, w.stop, w.sample_rate = start, stop, rate
    want = API[name]
    close(np_oracle.sample(w), want, rel=1e-14)
    g = _flatten.grid_arange(start, stop, 1 / rate)
    assert g.n == len(want)
    assert np.array_equal(c_oracle.grid_values(g), np.arange(start, stop, 1 / rate))
    close(c_oracle.eval_grid(_flatten.flatten([w]), g)[0], want)


def test_reference_known_answers():
    # reference tests/test_waveform.py:8-35 closed forms
    t = np.linspace(-10, 10, 1001)
    assert np.allclose(np_oracle.call(wf.cos(1), t), np.cos(t), atol=1e-4)
    assert np.allclose(np_oracle.call(wf.sin(1), t), np.sin(t), atol=1e-4)
    s = 2 / (4 * np.sqrt(np.log(2)))
    assert np.allclose(np_oracle.call(wf.gaussian(2), t), np.exp(-(t / s)**2),
                       atol=5e-3)
    assert np.allclose(np_oracle.call(wf.poly([1, -1 / 2, 1 / 6, -1 / 12]), t),
                       np.poly1d([-1 / 12, 1 / 6, -1 / 2, 1])(t))
    # SURVEY.md Appendix E (README config)
    x, y = wl.readme_xy(wf)
    tt = np.linspace(-1e-6, 9e-6, 10001)
    xv, yv = np_oracle.call(x, tt), np_oracle.call(y, tt)
    assert np.array_equal(np.searchsorted(tt, x.bounds),
                          [990, 1010, 1990, 2010, 2990, 3010, 10001])
    assert xv[989] == 0.0 and xv[1010] == 0.0
    assert abs(xv[1000] - 0.5) < 1e-8 and abs(yv[2000] - 1.0) < 1e-8
    assert abs(xv.sum() - 224873786.26300055) < 1e-3


@pytest.mark.parametrize('i', range(10))
def test_fir_oracles(i):
    sig, ker, want = FIR[f'{i}.sig'], FIR[f'{i}.ker'], FIR[f'{i}.out']
    scale = max(1.0, np.abs(want).max())
    assert np.max(np.abs(np_oracle.predistort_fir(sig, ker) - want)) <= 1e-13 * scale
    if len(sig) * len(ker) <= 5e7:
        assert np.max(np.abs(c_oracle.fir(sig, ker) - want)) <= 1e-12 * scale


EDGES = golden_io.npz('edges.npz')


@pytest.mark.parametrize('name', sorted(cases.edge_cases()))
def test_oracles_on_edge_inputs(name):
    """empty / single / off-support / duplicated / non-uniform x against the real reference."""
    build, xs = cases.edge_cases()[name]
    w = build(wf)
    prog = _flatten.flatten([w])
    for k, x in enumerate(xs):
        want = EDGES[f'{name}.{k}']
        close(np_oracle.call(w, x), want, rel=1e-14)
        got = c_oracle.eval_tlist(prog, x, want.dtype == np.complex128)[0]
        close(got, want)


# ---- random pulse scripts: both oracles and the front-end against the REAL reference ----
FUZZ = golden_io.npz('fuzz.npz')
FUZZ_LISTS = golden_io.frontend_lists_named('fuzz_frontend.json')


@pytest.mark.parametrize('seed', range(cases.FUZZ_GOLD))
def test_oracles_match_reference_on_random_scripts(seed):
    from test_frontend_golden import same
    w, grid = cases.fuzz_golden_case(wf, seed)
    want = FUZZ[f'{seed}.y']
    # front-end: the flat list of the script equals the reference's tolist() element by element
    got_l, want_l = w.tolist(), FUZZ_LISTS[str(seed)]
    assert len(got_l) == len(want_l)
    assert all(same(a, b) for a, b in zip(got_l, want_l)), seed
    t = wl.make_grid(grid)
    y = np.asarray(np_oracle.call(w, t))
    close(y.real.astype(np.float64), want, rel=1e-13)
    prog = _flatten.flatten([w])
    g = _flatten.grid_from_desc(grid)
    assert np.array_equal(c_oracle.grid_values(g), t)
    close(c_oracle.eval_grid(prog, g)[0], want, rel=1e-11)


@pytest.mark.parametrize('seed', range(cases.FAR_GOLD))
def test_oracles_match_reference_far_from_origin(seed):
    """Pulses and grid far from t = 0 (NumPy's grid rounding visible through the carriers): the
    oracles evaluate at the rounded grid values exactly like the reference."""
    chans, grid = cases.far_golden_case(wf, seed)
    t = wl.make_grid(grid)
    g = _flatten.grid_from_desc(grid)
    assert np.array_equal(c_oracle.grid_values(g), t)
    for c, w in enumerate(chans):
        want = FUZZ[f'far{seed}.{c}']
        close(np.asarray(np_oracle.call(w, t)).real.astype(np.float64), want, rel=1e-13)
        close(c_oracle.eval_grid(_flatten.flatten([w]), g)[0], want, rel=1e-11)


@pytest.mark.parametrize('name', sorted(cases.LATE_CASES))
def test_oracles_match_reference_on_oversampled_grids(name):
    """tests/golden/late.npz: flat tops with erf edges, multi-tone readout, cosh pulses, exponential
    factors on 600 k - 800 k point grids, evaluated by the real reference (strided subset + every piece
    edge).  These are the shapes the late round-2 fusions evaluate on the device; both oracles are
    pinned on them here."""
    late = golden_io.npz('late.npz')
    build, grid = cases.LATE_CASES[name]
    w = build(wf)
    t = wl.make_grid(grid)
    pick, want = late[name + '.pick'], late[name + '.y']
    pk = max(1.0, float(np.abs(want).max()))
    got_np = np.asarray(np_oracle.call(w, t[pick]))
    assert np.max(np.abs(got_np - want)) <= 1e-13 * pk
    prog = _flatten.flatten([w])
    cplx = np.iscomplexobj(want)
    got_c = c_oracle.eval_tlist(prog, t[pick], cplx)[0]
    far = abs(grid[1]) > 1e-5
    assert np.max(np.abs(got_c - want)) <= (1e-9 if far else 1e-12) * pk


@pytest.mark.parametrize('i', range(len(cases.predistort_cplx_cases())))
def test_predistort_oracle_on_complex_inputs(i):
    """complex signals / kernels / states through predistort (scipy takes them: reference distortion.py:298-337):
    the NumPy oracle against reference-generated vectors (iir.npz, pdc*)"""
    import golden_io
    gold = golden_io.npz('iir.npz')
    n, params, initial, k, _, _, _ = cases.predistort_cplx_cases()[i]
    sig, ker, zi = cases.predistort_cplx_inputs(i)
    filters = None
    if params is not None:
        from waveforms_amd import distortion
        filters = [distortion.exp_decay_filter(A, tau, 1e9) for A, tau in params]
    got, zf = np_oracle.predistort(sig, filters, ker, initial, zi)
    want = gold[f'pdc{i}.out']
    assert got.dtype == want.dtype == np.complex128
    assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.abs(want).max())
    if filters is not None:
        assert np.max(np.abs(zf - gold[f'pdc{i}.zf'])) <= 1e-12


@pytest.mark.parametrize('i', range(len(cases.predistort_high_cases())))
def test_predistort_oracle_combined_order_17_to_20(i):
    """combined order > 16 where the reference's direct form is still accurate (iir.npz, pdh*): the oracle's
    restatement (product polynomials -> one lfilter, as distortion.py:298-321) against the reference's own output"""
    import golden_io
    from waveforms_amd import distortion
    gold = golden_io.npz('iir.npz')
    n, params, initial = cases.predistort_high_cases()[i]
    sig = cases.predistort_high_input(i)
    filters = [distortion.exp_decay_filter(A, tau, 1e9) for A, tau in params]
    got, _ = np_oracle.predistort(sig, filters, None, initial)
    want = gold[f'pdh{i}.out']
    assert got.dtype == want.dtype
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.abs(want).max())
