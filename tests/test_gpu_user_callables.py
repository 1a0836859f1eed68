"""Python-callable primitives through the drop-in API on the GPU: function(),
registerBaseFunc, function_lib= (explicit argument, WaveVStack attribute, built-in remap),
against outputs of the real reference (tests/golden/user.npz) and the oracle.
reference: waveform.py:1470-1478, 178/535/679; _waveform.pyx:130-131, 264-271."""
import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import user_lib
import waveforms_amd as wf
from oracle import np_oracle
from waveforms_amd._sampling import BatchSampler

pytestmark = pytest.mark.gpu
USER = golden_io.npz('user.npz')


def _close(got, want, tol=1e-11):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.max(np.abs(got - want)) <= tol * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('name', sorted(cases.USER_CASES))
def test_call_matches_reference(name):
    w, lib, x = cases.USER_CASES[name](wf)
    got = w(x) if lib is None else w(x, function_lib=lib)
    _close(got, USER[name + '.y'])
    # scalar x and out= / accumulate
    k = len(x) // 3
    one = w(float(x[k])) if lib is None else w(float(x[k]), function_lib=lib)
    assert abs(one - USER[name + '.y'][k]) <= 1e-11 * max(1.0, np.abs(USER[name + '.y']).max())
    if not isinstance(w, wf.WaveVStack):
        out = np.ones_like(USER[name + '.y'])
        r = w(x, out=out, accumulate=True, function_lib=lib)
        assert r is out
        _close(out - 1, USER[name + '.y'])
        parts = w(x, frag=True, function_lib=lib)
        idx = np.array([[a, b] for a, b, _ in parts], dtype=np.int64).reshape(-1, 2)
        assert np.array_equal(idx, USER[name + '.frag_idx'])          # integer: bit-exact
        sums = np.array([np.sum(np.asarray(p) * np.ones(b - a)) for a, b, p in parts])
        assert np.max(np.abs(sums - USER[name + '.frag_sum'])) <= 1e-9 * max(1.0, np.abs(sums).max())


def test_sample_with_function_lib():
    w, lib = cases.user_sample_case(wf)
    _close(w.sample(function_lib=lib), USER['sample.full'])
    _close(np.concatenate(list(w.sample(chunk_size=257, function_lib=lib))), USER['sample.chunked'])
    with pytest.raises(KeyError):          # the id function() registered is not in a stale library
        w.sample(function_lib=dict(list(wf._waveform._baseFunc.items())[:17]))


def test_batch_grid_mode_mixes_callables_and_fused_channels():
    # one launch: a channel with a Python callable (general kernel, table factor) next to plain
    # gaussian+carrier channels; grid mode evaluates the callable on the exact NumPy grid
    x = cases._user_x()
    grid = ('linspace', float(x[0]), float(x[-1]), len(x), True)
    chans = [cases._u_tanh(wf)[0], wf.gaussian(40e-9) * wf.cos(2 * np.pi * 60e6),
             cases._u_vstack(wf)[0], cases._u_lib_vstack_attr(wf)[0]]
    got = BatchSampler(chans, grid).to_host(np.float64)
    for c, name in ((0, 'u_tanh'), (2, 'u_vstack'), (3, 'u_lib_vstack_attr')):
        _close(got[c], USER[name + '.y'])
    _close(got[1], np_oracle.call(chans[1], x), 1e-11)
    f32 = BatchSampler(chans, grid).to_host(np.float32)
    assert np.max(np.abs(f32[0] - USER['u_tanh.y'])) <= FP32_TOL


def test_builtin_registry_entries_evaluate_on_the_device():
    base = wf._waveform._baseFunc
    t = np.linspace(-3.0, 3.0, 1001)[::-1].copy()         # any order: a user calls it on anything
    _close(base[wf._waveform.GAUSSIAN](t, 0.8), np.exp(-(t / 0.8)**2), 1e-13)
    _close(base[wf._waveform.COS](t.reshape(7, 11, 13), 5.0), np.cos(5.0 * t.reshape(7, 11, 13)), 1e-13)
    assert abs(base[wf._waveform.ERF](0.3, 1.0) - 0.3286267594591274) <= 1e-14
    # ... so a callable may build on them
    w = wf.function(lambda tt, s: base[2](tt, s)**2 * np.sign(tt), 0.5, start=-1, stop=1)
    x = np.linspace(-1.5, 1.5, 301)
    want = np.where((x >= -1) & (x < 1), np.exp(-2 * (x / 0.5)**2) * np.sign(x), 0.0)
    _close(w(x), want, 1e-12)


def test_large_piece_table():
    # 3e6 samples inside one callable piece: the table travels as one NumPy block
    x = np.linspace(0.0, 1e-3, 3_000_001)
    w = wf.function(cases.uf_tanh, 4e3, 0.5) * wf.cos(2 * np.pi * 1e5)
    want = np_oracle.call(w, x, user_lib.oracle_lib(w, None))
    _close(w(x), want, 1e-11)
