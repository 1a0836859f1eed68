"""SURVEY 8(f) N4 on the device: trees that come out of the symbolic layer -- the text front-end `wave_eval`
(reference tests/test_waveform.py:141-166, waveform_parser.py), `simplify` (test_waveform.py:81-105,
_waveform.pyx:596-636), `filter` (:638-654), `marker` / `mask` / `|` / `&` (waveform.py:416-476), `interp`
(:1425-1440) and the CLI (__main__.py:17-31) -- SAMPLED through the HIP path and compared with what the real
reference sampled for the twin trees (tests/golden/n4.npz, oracle/make_golden.py: make_n4) and with the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import np_oracle
from waveforms_amd import _engine, _flatten, wave_eval, workloads as wl
from waveforms_amd._sampling import BatchSampler

pytestmark = pytest.mark.gpu

N4 = golden_io.npz('n4.npz')
SAMPLES = golden_io.npz('samples.npz')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP64_TOL = 1e-9       # of max(1, peak): the fp64 contract of BASELINE.json


def close(got, want, tol=FP64_TOL):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    if want.size:
        pk = max(1.0, float(np.max(np.abs(want))))
        assert float(np.max(np.abs(got - want))) <= tol * pk


def device_checks(w, grid, want):
    """the drop-in call (grid detected), the explicit non-grid axis (tlist tier) and a float launch of the
    batched API, all against the reference's vector"""
    t = wl.make_grid(grid)
    close(w(t), want)
    if len(t) > 8:
        # the same times as a list the library cannot take for a grid: every other sample dropped at random
        keep = np.sort(np.random.default_rng(len(t)).choice(len(t), size=len(t) // 2, replace=False))
        assert _engine.detect_grid(np.ascontiguousarray(t[keep])) is None
        close(w(t[keep]), np.asarray(want)[keep])
    bs = BatchSampler([w], grid)
    try:
        if np.iscomplexobj(want):
            got = bs.to_host(np.complex64)[0]
        else:
            got = bs.to_host(np.float32)[0]
        pk = max(1.0, float(np.max(np.abs(want), initial=0.0)))
        assert float(np.max(np.abs(got - want), initial=0.0)) <= FP32_TOL * pk
    finally:
        bs.close()


@pytest.mark.parametrize('name', sorted(cases.PARSER_CASES))
def test_parsed_text_sampled_on_the_device(name):
    text, twin, grid = cases.PARSER_CASES[name]
    w = wave_eval(text)
    assert w == twin(wf) or w.tolist() == twin(wf).simplify().tolist()
    want = N4[f'parse.{name}']
    device_checks(w, grid, want)
    # and against the oracle's evaluation of the very tree the parser built
    t = wl.make_grid(grid)
    ora = np_oracle.call(w, t)
    close(w(t), ora if np.iscomplexobj(want) else np.real(ora), 1e-11)


def test_reference_parser_known_answers_on_the_device():
    # closed forms of the reference's test_parser expressions (tests/test_waveform.py:146-166)
    t = np.linspace(-130, 40, 6801)
    w = wave_eval("(gaussian(10) << 100) + square(20, edge=5, type='linear') * cos(2*pi*23.1)")
    s = 10 / (4 * np.sqrt(np.log(2)))
    g = np.where((t >= -100 - 7.5) & (t < -100 + 7.5), np.exp(-((t + 100) / s)**2), 0.0)
    ramp = np.clip((t + 12.5) / 5, 0, 1) - np.clip((t - 7.5) / 5, 0, 1)
    want = g + ramp * np.cos(2 * np.pi * 23.1 * t)
    assert np.max(np.abs(w(t) - want)) <= 1e-9
    t2 = np.linspace(-10, 10, 1001)
    p = wave_eval("poly((1, -1/2, 1/6, -1/12))")
    assert np.max(np.abs(p(t2) - (1 - t2 / 2 + t2**2 / 6 - t2**3 / 12))) <= 1e-9 * 100
    assert np.array_equal(wave_eval("one()")(t2), np.ones_like(t2))
    assert np.array_equal(wave_eval("zero()")(t2), np.zeros_like(t2))
    assert np.array_equal(wave_eval("pi")(t2), np.full_like(t2, np.pi))


@pytest.mark.parametrize('name', sorted(k[5:] for k in N4.files if k.startswith('simp.')))
def test_simplified_tree_sampled_on_the_device(name):
    build, grid = cases.CASES[name]
    w = build(wf).simplify()
    want = N4[f'simp.{name}']
    t = wl.make_grid(grid)
    close(w(t), want)
    # simplification must not change the values: the reference's UNSIMPLIFIED samples (samples.npz).
    # (Except for a clip: simplify() returns a fresh Waveform WITHOUT min / max, reference waveform.py:384-396.)
    orig = build(wf)
    if np.isfinite(getattr(orig, 'min', -np.inf)) or np.isfinite(getattr(orig, 'max', np.inf)):
        return
    if isinstance(orig, wf.WaveVStack) and not orig.wlist:
        return      # (WaveVStack([]).simplify() is zero() whatever the offset: reference waveform.py:731-733)
    plain = SAMPLES[name + '.y']
    got = w(t)
    if np.iscomplexobj(got) and not np.iscomplexobj(plain):
        assert np.all(got.imag == 0)
        got = got.real
    elif np.iscomplexobj(plain) and not np.iscomplexobj(got):
        assert np.max(np.abs(plain.imag)) <= 1e-12 * max(1.0, np.abs(plain).max())
        plain = plain.real
    pk = max(1.0, float(np.max(np.abs(plain), initial=0.0)))
    assert float(np.max(np.abs(got - plain), initial=0.0)) <= 1e-9 * pk, name


def test_reference_simplify_known_answers_on_the_device():
    # reference tests/test_waveform.py:81-105 (test_simplify, test_simplify2, test_simplify3)
    t = np.linspace(-10, 10, 1001)
    wav = wf.cos(1) * wf.sin(2) * wf.cos(3, 4)
    want = np.cos(t) * np.sin(2 * t) * np.cos(3 * t + 4)
    assert np.allclose(wav(t), want) and np.allclose(wav.simplify()(t), want)
    assert np.max(np.abs(wav.simplify()(t) - want)) <= 1e-9
    t = np.linspace(-2, 2, 1001)
    wav = 1j * (wf.cos(9) >> 1) + 1 * (wf.cos(9) >> 2) - 1j * (wf.cos(9) >> 3)
    assert np.allclose(wav(t), wav.simplify()(t))
    assert np.max(np.abs(wav(t) - wav.simplify()(t))) <= 1e-9
    wav = 2 * (wf.exp(1.01 + 22j)**2 << 1) * wf.exp(1.01 + 22j)
    points = 2 * np.exp((1.01 + 22j) * (t + 1))**2 * np.exp((1.01 + 22j) * t)
    assert np.allclose(wav(t), points) and np.allclose(wav.simplify()(t), points)
    assert np.max(np.abs(wav.simplify()(t) - points)) <= 1e-9 * np.abs(points).max()


@pytest.mark.parametrize('name', sorted(cases.FILTER_CASES))
def test_filtered_tree_sampled_on_the_device(name):
    build, lo, hi, grid = cases.FILTER_CASES[name]
    device_checks(build(wf).filter(lo, hi), grid, N4[f'filter.{name}'])


@pytest.mark.parametrize('name', sorted(cases.INTERP_CASES))
def test_interp_tree_sampled_on_the_device(name):
    build, grid = cases.INTERP_CASES[name]
    device_checks(build(wf), grid, N4[f'interp.{name}'])


def test_marker_mask_or_and_sampled_on_the_device():
    """0 / 1 windows: every sample must be EXACTLY the reference's (piece membership is integer work)."""
    with open(os.path.join(golden_io.GOLDEN, 'logic.json')) as f:
        gold = json.load(f)
    names = cases.n4_logic_names()
    assert len(names) >= 60
    trees, wants, labels = {}, {}, {}
    for name in names:
        w, grid = cases.CASES[name][0](wf), cases.CASES[name][1]
        other = cases.CASES[gold[name]['other']][0](wf)
        for key, tree in (('marker', w.marker), ('mask0', w.mask()), ('mask_e', w.mask(0.37)),
                          ('or', w | other), ('and', w & other)):
            trees.setdefault(grid, []).append(tree)
            wants.setdefault(grid, []).append(N4[f'logic.{name}.{key}'].astype(np.float64))
            labels.setdefault(grid, []).append((name, key))
    for grid, ws in trees.items():
        # one batched launch per grid (all windows of that grid are rows of one plan) ...
        bs = BatchSampler(ws, grid)
        got = bs.to_host(np.float64)
        bs.close()
        for row, want, label in zip(got, wants[grid], labels[grid]):
            assert np.array_equal(row, want), label
        # ... and the drop-in call on a few of them
        t = wl.make_grid(grid)
        for k in range(0, len(ws), 7):
            assert np.array_equal(ws[k](t), wants[grid][k]), labels[grid][k]


def test_mask_gates_a_sampled_waveform():
    # the use the reference makes of mask(): a window `edge` wider than the support, multiplied back in
    w = (wf.gaussian(4) >> 3) * wf.cos(20) + (wf.square(2, edge=0.5) << 5)
    t = np.linspace(-10, 10, 4001)
    gate = w.mask(0.25)
    g = gate(t)
    assert set(np.unique(g)) <= {0.0, 1.0}
    y = w(t)
    assert np.array_equal((w * gate)(t) != 0, (y != 0) & (g != 0)) or np.max(np.abs((w * gate)(t) - y * g)) <= 1e-12
    close((w * gate)(t), np_oracle.call(w * gate, t).real, 1e-12)


@pytest.mark.parametrize('name', sorted(cases.CLI_CASES))
def test_cli_end_to_end(name, tmp_path):
    """`python -m waveforms_amd sample ... EXPR OUT.npy` in its own process (reference __main__.py:17-31):
    parse -> simplify -> sample() on the device -> scale -> np.save."""
    argv, text, twin, start, stop, rate, amp = cases.CLI_CASES[name]
    out = tmp_path / (name + '.npy')
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, '-m', 'waveforms_amd', 'sample', *argv, text, str(out)],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    got = np.load(out)
    want = N4[f'cli.{name}']
    close(got, want)
    w = twin(wf)
    w.start, w.stop, w.sample_rate = start, stop, rate
    close(got, np.real(np_oracle.sample(w)) * amp, 1e-11)


def test_parsed_tree_through_sample_and_chunks():
    # Waveform.sample() and the chunked iterator on a parsed tree (reference waveform.py:173-257)
    argv, text, twin, start, stop, rate, amp = cases.CLI_CASES['cli_awg']
    w = wave_eval(text)
    w.start, w.stop, w.sample_rate = start, stop, rate
    full = w.sample()
    close(full, N4['cli.cli_awg'])
    chunks = np.concatenate(list(w.sample(chunk_size=257)))
    assert chunks.shape == full.shape and np.max(np.abs(chunks - full)) <= 1e-12


@pytest.mark.parametrize('name', sorted(cases.out_nonfinite_cases()))
def test_out_buffer_holding_non_finite_values(name):
    """`wav(x, out=buf)`: the reference zeroes buf with `out *= 0` (waveform.py:551), which keeps NaN / inf;
    the result must be the reference's, element by element (NaN positions included)."""
    build, x, buf = cases.out_nonfinite_cases()[name]
    want = N4[f'outbuf.{name}']
    buf = buf.copy()
    with np.errstate(invalid='ignore'):
        r = build(wf)(x, out=buf)
    assert r is buf and buf.dtype == want.dtype
    assert np.array_equal(np.isnan(buf.real), np.isnan(want.real)) and np.array_equal(np.isnan(buf.imag), np.isnan(want.imag))
    ok = np.isfinite(want)
    assert np.max(np.abs(buf[ok] - want[ok])) <= 1e-12


def test_all_finite_scan():
    for n in (0, 1, 63, 4097, 3_000_001):
        a = np.random.default_rng(n).normal(size=n)
        assert _engine.all_finite(a)
        for bad in (np.nan, np.inf, -np.inf):
            for pos in {0, n // 2, n - 1} if n else ():
                b = a.copy()
                b[pos] = bad
                assert not _engine.all_finite(b)
    c = np.zeros(1000, dtype=np.complex128)
    assert _engine.all_finite(c)
    c[999] = complex(0.0, np.nan)
    assert not _engine.all_finite(c)
