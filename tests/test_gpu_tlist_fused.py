"""The time-list tier with POINTWISE fused ops (DESIGN: tlist tier): `Waveform.__call__(x)` on an x that is not a
NumPy grid (reference waveforms/waveform.py:529-563 with arbitrary sorted x; `_calc`, _waveform.pyx:134-152).
Every term the host's fusion pass can fuse is evaluated per sample as ONE sincos + ONE exp per carrier-envelope
group instead of a libm call per factor; the rest stays on device libm.  Checked against the C oracle (libm per
factor, the reference's pass structure) and against the same plan with the fusion off."""
import os

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


def run(chans, t, dtype=np.float64, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        plan = _engine.Plan(_flatten.flatten(chans), t=t)
        return plan.run_host(dtype), plan.kernel_name(dtype), plan.info
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def oracle(chans, t, cplx=False):
    return c_oracle.eval_tlist(_flatten.flatten(chans), t, want_complex=cplx)


def test_headline_channels_on_jittered_times():
    t = wl.jittered_times(1_200_000)
    chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(3)]
    got, name, info = run(chans, t)
    assert name == 'wfk_sample<double,false,true,false,false,8>' and info.n_generic == 0 and info.n_fused > 0
    ref = oracle(chans, t)
    assert np.max(np.abs(got - ref)) <= 1e-11
    off, name_off, info_off = run(chans, t, env={'WFK_DISABLE_TLFUSE': '1'})
    assert name_off == 'wfk_sample<double,false,true,true,true,8>' and info_off.n_fused == 0
    assert np.max(np.abs(off - ref)) <= 1e-12                 # (libm per factor: the round-3 path)
    # every carrier through the large-phase reduction instead
    big, _, _ = run(chans, t, env={'WFK_TLSMALL_LIMIT': '0'})
    assert np.max(np.abs(big - ref)) <= 1e-11
    f32, name32, _ = run(chans, t, np.float32)
    assert name32 == 'wfk_sample_wide<false,true,false,false,8>'      # float elements, double arithmetic
    assert np.max(np.abs(f32 - ref)) <= 1e-7


def test_small_calls_take_the_one_sample_per_lane_build():
    x, y = wl.readme_xy(wf)
    t = np.sort(np.linspace(-1e-6, 9e-6, 10001) + np.random.default_rng(0).normal(size=10001) * 1e-11)
    assert _engine.detect_grid(t) is None
    for w in (x, y):
        got, name, info = run([w], t)
        assert name == 'wfk_sample<double,false,true,false,false,1>' and info.n_generic == 0
        ref = oracle([w], t)
        assert np.max(np.abs(got - ref)) <= 1e-11 * np.abs(ref).max()
        assert np.max(np.abs(w(t) - ref[0])) <= 1e-11 * np.abs(ref).max()      # the drop-in call


def test_multitone_shared_envelope_and_complex_groups():
    t = wl.jittered_times(1_100_000)
    chans = [wl.multitone_channel(wf, c, ntones=6, nseg=40) for c in range(2)]
    got, name, info = run(chans, t)
    assert info.n_generic == 0
    ref = oracle(chans, t)
    assert np.max(np.abs(got - ref)) <= 1e-11
    # I + 1j Q channels: imaginary groups into the imaginary accumulators, dropped by a real launch
    I, Q = wf.mixing(wf.gaussian(20e-9) >> 40e-9, freq=123e6, phase=0.4, DRAGScaling=2e-10)
    w = I + 1j * Q + 0.5j * (wf.gaussian(30e-9) >> 90e-9) * wf.cos(2 * np.pi * 77e6)
    tt = np.sort(np.random.default_rng(3).uniform(-10e-9, 140e-9, 50_000))
    gc, namec, infoc = run([w], tt, np.complex128)
    assert infoc.n_generic == 0 and namec == 'wfk_sample<double,true,true,false,false,1>'
    refc = oracle([w], tt, True)
    assert np.max(np.abs(gc - refc)) <= 1e-11
    c64, _, _ = run([w], tt, np.complex64)
    assert np.max(np.abs(c64 - refc)) <= FP32_TOL
    re, _, _ = run([w], tt, np.float64)
    assert np.max(np.abs(re - refc.real)) <= 1e-11


def test_mixed_plans_keep_their_generic_terms_on_libm():
    # flat tops with erf edges: the plateaus fuse (pointwise build), the edge pieces keep ALL their terms on
    # libm (build with the direct tier): two launches over disjoint pieces, every sample written once
    rng = np.random.default_rng(5)
    ws = [(wf.square(40e-9, edge=8e-9) >> ((k + 0.5) * 60e-9)) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6))
          for k in range(20)]
    w = wl._tree_sum(ws)
    t = np.sort(rng.uniform(0, 1.2e-6, 1_100_000))
    got, name, info = run([w], t)
    assert name == 'wfk_sample<double,false,true,false,false,8> + wfk_sample<double,false,true,true,true,8>'
    assert info.n_generic > 0 and info.n_fused > 0
    ref = oracle([w], t)
    assert np.max(np.abs(got - ref)) <= 1e-11
    off, _, _ = run([w], t, env={'WFK_DISABLE_TLFUSE': '1'})
    assert np.max(np.abs(off - ref)) <= 1e-12
    # poisoned output: nothing is left unwritten, nothing written twice (accumulate adds exactly once)
    plan = _engine.Plan(_flatten.flatten([w]), t=t)
    buf = _engine.DeviceBuffer(len(t) * 8)
    buf.upload(np.full(len(t), 7.0))
    plan.launch(buf.ptr, len(t), _engine.OUT_F64, accumulate=True)
    _engine.sync()
    acc = buf.download((len(t), ), np.float64)
    assert np.max(np.abs(acc - 7.0 - ref[0])) <= 1e-11
    buf.close()
    plan.close()


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_every_case_fused_and_unfused_agree_on_scattered_times(name):
    """all of tests/cases.py on times drawn at random inside the case's window (not its grid): the pointwise
    tier, the libm tier and the oracle; exp / cosh envelopes, Gaussian derivatives, the DRAG primitive, clip,
    vstack shifts and offsets, complex amplitudes come from the cases"""
    build, grid = cases.CASES[name]
    g = wl.make_grid(grid)
    if len(g) < 2:
        pytest.skip('single-point grid')
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    t = np.sort(rng.uniform(g[0], g[-1], 3000))
    w = build(wf)
    cplx = np.iscomplexobj(SAMPLES[name + '.y'])
    dt = np.complex128 if cplx else np.float64
    ref = oracle([w], t, cplx)
    pk = max(1.0, float(np.abs(ref).max()))
    got, _, _ = run([w], t, dt)
    assert np.max(np.abs(got - ref)) <= 1e-11 * pk, name
    off, _, _ = run([w], t, dt, env={'WFK_DISABLE_TLFUSE': '1'})
    assert np.max(np.abs(off - ref)) <= 1e-11 * pk, name


def test_far_from_the_origin_is_fused_only_where_the_phase_noise_allows():
    # 0.1 ms out under a 300 MHz carrier: W |t| = 1.9e5 rad -> ulp-level phase noise 4.5e-11, inside the budget the three
    # terms of the pulse share (2.5e-10 / 3): fused; 0.4 ms (1.8e-10) and 4 ms out (1.8e-9): the terms stay on libm at the
    # exact times (what the reference computes)
    for t0, fused in ((0.1e-3, True), (0.4e-3, False), (4e-3, False)):
        I, _ = wf.mixing(wf.gaussian(200e-9) >> (t0 + 300e-9), freq=300e6, phase=0.3, DRAGScaling=1e-10)
        t = np.sort(t0 + np.random.default_rng(1).uniform(0, 600e-9, 40_000))
        got, name, info = run([I], t)
        assert (info.n_generic == 0) == fused, (t0, info.n_generic, info.n_fused)
        ref = oracle([I], t)
        assert np.max(np.abs(got - ref)) <= 1e-9


def test_time_lists_of_zero_one_and_repeated_times():
    I, _ = wf.mixing(wf.gaussian(20e-9) >> 15e-9, freq=150e6, phase=0.3, DRAGScaling=1e-10)
    for t in (np.zeros(0), np.array([15e-9]), np.array([15e-9, 15e-9, 15e-9, 16e-9]),
              np.sort(np.concatenate([np.full(70, 14e-9), np.linspace(0, 30e-9, 130)]))):
        got, name, info = run([I], t)
        assert got.shape == (1, len(t))
        if len(t):
            ref = oracle([I], t)
            assert np.max(np.abs(got - ref)) <= 1e-11
            assert np.max(np.abs(I(t) - ref[0])) <= 1e-11
        else:
            assert I(t).shape == (0, )
