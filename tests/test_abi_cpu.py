"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/wfk.h declares, compiles plans without a GPU (host-only plans) and gets
the INTEGER part of the contract bit-exact: np.searchsorted piece indices."""
import os
import re

import numpy as np
import pytest

import cases
import golden_io
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten, workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLES = golden_io.npz('samples.npz')


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'wfk.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(wfk_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _engine.lib()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert lib.wfk_abi_version() == 2


def _indices(plan, prog):
    m0, m1 = prog.member_range(0)
    if m1 == m0:
        return np.zeros(0, np.int64)
    return np.concatenate([plan.member_index(m) for m in range(m0, m1)])


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_piece_indices_bit_exact(name):
    build, grid = cases.CASES[name]
    prog = _flatten.flatten([build(wf)])
    want = SAMPLES[name + '.idx']
    plan = _engine.Plan(prog, grid=_flatten.grid_from_desc(grid))
    assert np.array_equal(_indices(plan, prog), want)
    plan2 = _engine.Plan(prog, t=wl.make_grid(grid))
    assert np.array_equal(_indices(plan2, prog), want)
    assert plan.n == plan2.n == len(SAMPLES[name + '.y'])


def test_big_grid_indices():
    big = golden_io.npz('big.npz')
    prog = _flatten.flatten([wl.c2_channel(wf)])
    plan = _engine.Plan(prog, grid=_flatten.grid_from_desc(wl.c2_grid()))
    assert np.array_equal(plan.member_index(0), big['c2.idx'])
    # the whole DRAG workload is absorbed by the host fusion pass
    assert plan.info.n_direct == 0 and plan.info.n_generic == 0 and plan.info.n_fused > 0


def test_callable_primitives_need_an_axis_and_unknown_ids_are_refused():
    # a Python callable is evaluated by the host on the plan's time axis: no axis, no program
    w = wf.function(lambda t: t * 0 + 1.0)
    with pytest.raises(ValueError):
        _flatten.flatten([w])
    prog = _flatten.flatten([w], np.linspace(0, 1, 5))
    assert list(prog.arrays['fc_type']) == [_flatten.SAMPLED]
    # an explicit library replaces the registry (reference _apply: function_lib[func_id])
    with pytest.raises(KeyError):
        _flatten.flatten([wf.cos(1)], np.linspace(0, 1, 5), function_lib={1: lambda t: t})


def test_sample_needs_init():
    with pytest.raises(ValueError):
        wf.cos(1).sample()


def test_vstack_frag_asserts():
    with pytest.raises(AssertionError):
        wf.WaveVStack([wf.cos(1)])(np.linspace(0, 1, 5), frag=True)


@pytest.mark.skipif(_engine.device_count() > 0, reason='GPU present')
def test_launch_without_gpu_fails_loudly():
    with pytest.raises(_engine.EngineError):
        wf.cos(1)(np.linspace(0, 1, 5))


def test_arange_grid_formula():
    rng = np.random.default_rng(0)
    for _ in range(200):
        start = rng.uniform(-1e-5, 1e-5)
        rate = 10**rng.uniform(6, 10)
        stop = start + rng.uniform(1, 5000) / rate
        g = _flatten.grid_arange(start, stop, 1 / rate)
        ref = np.arange(start, stop, 1 / rate)
        assert g.n == len(ref)
        i = np.arange(g.n, dtype=np.float64)
        assert np.array_equal(i * g.step + g.t0, ref)


def test_malformed_programs_are_rejected():
    """The host compiler validates the flattened forest before using it."""
    import copy
    grid = _flatten.grid_linspace(0.0, 1.0, 100)
    good = _flatten.flatten([wf.gaussian(0.5) * wf.cos(3.0) >> 0.5])
    _engine.Plan(good, grid=grid)

    def broken(edit):
        prog = _flatten.flatten([wf.gaussian(0.5) * wf.cos(3.0) >> 0.5])
        edit(prog)
        with pytest.raises(_engine.EngineError):
            _engine.Plan(prog, grid=grid)

    def bad_offsets(p):
        p.arrays['pc_term_off'][1] = 99
    broken(bad_offsets)

    def bad_last_bound(p):
        p.arrays['pc_bound'][-1] = 1.0
    broken(bad_last_bound)

    def descending_bounds(p):
        p.arrays['pc_bound'][0], p.arrays['pc_bound'][1] = 5.0, -5.0
    broken(descending_bounds)

    def bad_argc(p):
        p.arrays['fc_arg_off'][-1] += 1
    broken(bad_argc)

    prog = _flatten.flatten([wf.gaussian(0.5)])
    prog.arrays['fc_type'][0] = 99
    with pytest.raises(NotImplementedError):
        _engine.Plan(prog, grid=grid)
    with pytest.raises(_engine.EngineError):
        _engine.Plan(good, grid=_flatten.wfk_grid(0.0, -1.0, 10, 0, 0.0))


def run_c_consumer(tmp_path):
    """Build tests/c_abi/abi_smoke.c with gcc against include/wfk.h + libwfk_hip.so, run it -> stdout."""
    import subprocess
    exe = tmp_path / 'abi_smoke'
    libdir = os.path.join(ROOT, 'waveforms_amd', 'csrc')
    subprocess.run(['gcc', '-std=c11', '-O1', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'c_abi', 'abi_smoke.c'), '-o', str(exe),
                    '-L', libdir, '-lwfk_hip', '-lm', f'-Wl,-rpath,{libdir}'], check=True)
    torch_lib = ''
    try:   # same runtime-loading order as _engine.lib(): torch's bundled HIP runtime first
        import torch
        torch_lib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    except Exception:
        pass
    env = dict(os.environ, LD_LIBRARY_PATH=torch_lib + ':' + os.environ.get('LD_LIBRARY_PATH', ''))
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    return r.stdout


def test_plain_c_consumer(tmp_path):
    """include/wfk.h is consumable from plain C (gcc), no Python/torch/HIP headers.  (Its device branch runs
    under `-m gpu`: tests/test_gpu_c_abi.py.)"""
    assert 'abi_smoke' in run_c_consumer(tmp_path)


def test_grid_detection_is_exact():
    """wfk_grid_detect: a host x is taken for a grid only if EVERY element equals the NumPy
    formula fl(fl(i*step) + t0) (linspace with/without endpoint, arange); one ulp off anywhere,
    NaN, unsorted or concatenated grids stay in tlist mode."""
    rng = np.random.default_rng(8)
    grids = [np.linspace(-1e-6, 9e-6, 10001), np.linspace(0, 3e-6, 250000, endpoint=False),
             np.arange(-1e-6, 2e-6, 1e-9), np.arange(100000) * 0.5e-9,
             np.linspace(1e-3, 1e-3 + 1e-5, 100000, endpoint=False),
             np.linspace(-7.3e-3, -7.3e-3 + 1e-5, 300000, endpoint=False),
             np.linspace(-121.88426793235047, -121.88402308516527, 1500, endpoint=False),
             np.arange(3e5, dtype=np.float64)]
    for _ in range(30):
        a = rng.uniform(-1, 1) * 10.0**rng.uniform(-9, 2)
        span = 10.0**rng.uniform(-9, 0)
        grids.append(np.linspace(a, a + span, int(rng.integers(16, 50000)), endpoint=bool(rng.random() < 0.5)))
        grids.append(np.arange(a, a + span, span / int(rng.integers(16, 50000))))
    for t in grids:
        g = _engine.detect_grid(t)
        assert g is not None, (t[0], t[-1], len(t))
        assert np.array_equal(_flatten.grid_values(g), t)
        bad = t.copy()
        k = int(rng.integers(1, len(t) - 1))
        bad[k] = np.nextafter(bad[k], np.inf)
        assert _engine.detect_grid(bad) is None
    t = np.linspace(0, 1, 1000)
    for bad in (np.sort(rng.uniform(0, 1, 1000)), np.concatenate([t[:500], t[500:] + 0.5]),
                np.where(np.arange(1000) == 7, np.nan, t), t[::-1].copy(), t[:8], np.zeros(100)):
        assert _engine.detect_grid(np.ascontiguousarray(bad)) is None


def test_grid_run_detection_is_exact():
    """wfk_grid_detect_runs: an x that is several NumPy grids back to back splits into runs, every run verified
    element by element; one ulp off anywhere, a run too short, contiguous chunks without a spacing break
    (np.concatenate of _sample_iter's chunks: not told apart from one grid by spacing, and not one grid bit for
    bit) or random times -> None (the call then takes the time-list tier)."""
    a = np.linspace(0, 1e-6, 50000, endpoint=False)
    b = np.arange(2e-6, 3e-6, 0.5e-10)
    c = np.linspace(3.5e-6, 4e-6, 9000)
    t = np.concatenate([a, b, c])
    runs = _engine.detect_grid_runs(t)
    assert [s for s, _ in runs] == [0, len(a), len(a) + len(b)]
    for (s, g), n in zip(runs, (len(a), len(b), len(c))):
        assert g.n == n and np.array_equal(_flatten.grid_values(g), t[s:s + n])
    bad = t.copy()
    bad[60000] = np.nextafter(bad[60000], 1.0)
    assert _engine.detect_grid_runs(bad) is None
    assert _engine.detect_grid_runs(np.concatenate([a, b[:100], c])) is None           # a run below min_len
    chunks = np.concatenate([np.linspace(k * 1e-6, (k + 1) * 1e-6, 10000, endpoint=False) for k in range(3)])
    assert _engine.detect_grid(chunks) is None and _engine.detect_grid_runs(chunks) is None
    assert _engine.detect_grid_runs(np.sort(np.random.default_rng(0).uniform(0, 1, 100000))) is None
    assert _engine.detect_grid_runs(a) is not None and len(_engine.detect_grid_runs(a)) == 1
    # a run that crosses t = 0 (its spacing wobbles by ulps of i * step there, not of the tiny t)
    z = np.linspace(-0.2e-6, 0.4e-6, 60000, endpoint=False)
    runs = _engine.detect_grid_runs(np.concatenate([z, b]))
    assert [(s, int(g.n)) for s, g in runs] == [(0, len(z)), (len(z), len(b))]
