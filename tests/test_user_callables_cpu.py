"""Python-callable primitives without a GPU: the oracle is pinned against what the REAL
reference produced for function() / registerBaseFunc / function_lib= scripts
(tests/golden/user.npz, oracle/make_golden.py), and the product's HOST part -- calling the
user's callable per distinct factor and piece on the exact sample times and handing the values
over as WFK_SAMPLED table factors -- is checked by evaluating the flattened program with the C
oracle (no device involved)."""
import numpy as np
import pytest

import cases
import golden_io
import user_lib
import waveforms_amd as wf
from oracle import c_oracle, np_oracle
from waveforms_amd import _engine, _flatten

USER = golden_io.npz('user.npz')


@pytest.mark.parametrize('name', sorted(cases.USER_CASES))
def test_oracle_matches_reference(name):
    w, lib, x = cases.USER_CASES[name](wf)
    want = USER[name + '.y']
    got = np_oracle.call(w, x, user_lib.oracle_lib(w, lib))
    if not np.iscomplexobj(want):
        got = np.real(got)
    pk = max(1.0, np.abs(want).max())
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-14 * pk


@pytest.mark.parametrize('name', sorted(cases.USER_CASES))
def test_flattened_program_carries_the_callables_values(name):
    w, lib, x = cases.USER_CASES[name](wf)
    want = USER[name + '.y']
    if name == 'u_complex_pow' and _engine.device_count() == 0:
        # a BUILT-IN factor under a complex power is sampled on the device before NumPy applies the power
        # (the product has no CPU evaluation of built-ins): covered by tests/test_gpu_user_callables.py
        pytest.skip('needs the device to sample the built-in factor')
    prog = _flatten.flatten([w], x, lib)
    assert _flatten.SAMPLED in prog.arrays['fc_type'] or name in ('u_lib_remap_builtin', 'u_clip_complex')
    got = c_oracle.eval_tlist(prog, x, want_complex=np.iscomplexobj(want))[0]
    pk = max(1.0, np.abs(want).max())
    assert np.max(np.abs(got - want)) <= 1e-12 * pk
    # the library compiles it (host-only plan here) with bit-exact piece indices
    plan = _engine.Plan(prog, t=x)
    # (cases without a callable factor -- a built-in remapped under another id, a clip of complex built-ins --
    #  fuse completely on a time list: pointwise ops, no direct factor)
    assert plan.info.n_direct > 0 or (name in ('u_lib_remap_builtin', 'u_clip_complex') and plan.info.n_fused > 0)
    assert prog.host_complex == (name in ('u_complex_fn', 'u_complex_pow'))
    grid = _flatten.grid_linspace(x[0], x[-1], len(x))
    assert np.array_equal(_flatten.grid_values(grid), x)
    gprog = _flatten.flatten([w], grid, lib)
    gotg = c_oracle.eval_grid(gprog, grid, want_complex=np.iscomplexobj(want))[0]
    assert np.max(np.abs(gotg - want)) <= 1e-12 * pk


def test_explicit_library_replaces_the_registry():
    # reference semantics (_apply: function_lib[func_id]): an id missing from an explicit
    # library is a KeyError, also for ids that function() registered later
    lib = dict(wf._waveform._baseFunc)
    w = wf.function(cases.uf_bump, 1e-8)
    with pytest.raises(KeyError):
        _flatten.flatten([w], np.linspace(-1e-7, 1e-7, 11), lib)
    del lib[2]
    with pytest.raises(KeyError):
        _flatten.flatten([wf.gaussian(1e-8)], np.linspace(-1e-7, 1e-7, 11), lib)


def test_registry_surface():
    import pickle
    base = wf._waveform._baseFunc
    assert all(i in base for i in range(1, 18))
    tid = wf.registerBaseFunc(cases.uf_const)
    assert base[tid] is cases.uf_const and tid >= 18
    saved = dict(base)                      # other tests register lambdas, which no pickle takes
    try:
        for k in [k for k in base if k >= 18 and k != tid]:
            del base[k]
        blob = wf._waveform.packBaseFunc()
    finally:
        base.update(saved)
    assert pickle.loads(blob)[tid] is cases.uf_const
    wf._waveform.updateBaseFunc(blob)
    # a WaveVStack's library survives pickling (reference waveform.py:823-844)
    w = wf.WaveVStack([wf.gaussian(1.0)])
    w.function_lib = {2: cases.uf_bump}
    w2 = pickle.loads(pickle.dumps(w))
    assert w2.function_lib[2] is cases.uf_bump


def test_complex_valued_callable_expands_into_real_table_factors():
    # amp * (re + i im) * rest = amp re rest + (i amp) im rest: two terms with SAMPLED table factors
    t = np.linspace(0, 1, 5)
    w = wf.function(lambda tt: np.exp(1j * tt)) * 2.0
    prog = _flatten.flatten([w], t)
    assert prog.host_complex and prog.complex_amp
    assert list(prog.arrays['fc_type']) == [_flatten.SAMPLED, _flatten.SAMPLED]
    assert np.allclose(prog.arrays['tm_amp_re'], [2.0, 0.0]) and np.allclose(prog.arrays['tm_amp_im'], [0.0, 2.0])
    got = c_oracle.eval_tlist(prog, t, want_complex=True)[0]
    assert np.max(np.abs(got - 2.0 * np.exp(1j * t))) <= 1e-15


@pytest.mark.parametrize('power', [2 + 1j, np.complex128(2 + 1j), np.complex64(2 + 1j)])
def test_complex_power_in_any_spelling_takes_the_host_path(power):
    # np.complex128 subclasses complex, np.complex64 does not: both must expand over (re, im) like a
    # Python complex power instead of being cast to float64 (which drops the imaginary part)
    t = np.linspace(0.5, 1.5, 7)
    w = wf.function(lambda tt: 1.0 + tt) ** power
    prog = _flatten.flatten([w], t)
    assert prog.host_complex and prog.complex_amp
    got = c_oracle.eval_tlist(prog, t, want_complex=True)[0]
    assert np.max(np.abs(got - (1.0 + t)**complex(power))) <= 1e-13
    # a built-in under such a power must not stay on the native device path either
    if _engine.device_count() == 0:
        with pytest.raises(Exception):
            _flatten.flatten([wf.cos(3.0) ** power], t)
