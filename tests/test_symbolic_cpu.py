"""Host-side symbolic layer beyond tree construction: simplify,
equality-by-simplify, marker/mask, interp, WaveVStack.simplify, and the text front-end
`wave_eval` (own parser for the reference's Waveform.g4).  No GPU involved."""
import numpy as np
import pytest
from numpy import e, pi

import cases
import golden_io
import waveforms_amd as wf
from waveforms_amd import wave_eval
from waveforms_amd._ir import wave_sum

SIMPLE = golden_io.frontend_lists_named('frontend_simplified.json')


@pytest.mark.parametrize('name', sorted(SIMPLE))
def test_simplified_tree_matches_reference(name):
    want = SIMPLE[name]
    build = cases.CASES[name][0]
    w = build(wf)
    if want is None:                      # the reference raised as well
        with pytest.raises(Exception):
            w.simplify().tolist()
        return
    got = w.simplify().tolist()
    assert len(got) == len(want)
    for g, x in zip(got, want):
        if isinstance(x, float) and isinstance(g, (float, np.floating)):
            assert g == pytest.approx(x, rel=1e-13, abs=1e-300), name
        else:
            assert g == x, name


def test_reference_test_parser():
    # reference tests/test_waveform.py:141-166 (cannot run upstream without ANTLR/Java)
    assert wave_eval("one()") == wf.one()
    assert wave_eval("zero()") == wf.zero()
    assert wave_eval("pi") == pi
    assert wave_eval("e") == e
    w1 = (wf.gaussian(10) << 100) + wf.square(20, edge=5, type='linear') * wf.cos(2 * pi * 23.1)
    w2 = wave_eval("(gaussian(10) << 100) + square(20, edge=5, type='linear') * cos(2*pi*23.1)")
    w3 = wave_eval("((gaussian(10) << 50) + ((square(20, 5, type='linear') * cos(2*pi*23.1)) >> 50)) << 50")
    w4 = wave_eval("(gaussian(10) << 100) + square(20, 5, 'linear') * cos(2*pi*23.1)")
    assert w1 == w2 and w1 == w3 and w1 == w4
    p1 = wf.poly([1, -1 / 2, 1 / 6, -1 / 12])
    assert p1 == wave_eval("poly([1, -1/2, 1/6, -1/12])")
    assert p1 == wave_eval("poly((1, -1/2, 1/6, -1/12))")


def test_parser_grammar_details():
    # left-associative power; unary minus binds loosest (ANTLR alternative order)
    assert wave_eval("2 ** 3 ** 2") == 64
    assert wave_eval("-2 + 5") == -7
    assert wave_eval("3 * -1 + 2") == -9
    assert wave_eval("2 ^ 3") == 8
    assert wave_eval("1e3 + .5") == 1000.5
    assert wave_eval("(2j) * (2j)") == -4
    assert wave_eval("cos(2) >> 1") == (wf.cos(2) >> 1)
    assert wave_eval("gaussian(width=4, plateau=2)") == wf.gaussian(4, plateau=2)
    assert wave_eval("drag_sin(0.4, 14.0, 0, 0.013, (0.9, 0.55), 0.3, -7.0)").tolist() == \
        wf.drag_sin(0.4, 14.0, 0, 0.013, (0.9, 0.55), 0.3, -7.0).simplify().tolist()
    for bad in ("x = cos(1)", "foo(1)", "cos(1", "cos(1) +", "bar", "cos(1) $ 2", "cos(a=1, 2)"):
        with pytest.raises(SyntaxError):
            wave_eval(bad)


def test_reference_vstack_simplify_and_wave_sum():
    # reference tests/test_wavevstack.py:91-110, 140-143
    w1, w2 = wf.zero(), []
    assert w1 == wf.WaveVStack(w2).simplify()
    for freq in np.linspace(6.1, 6.5, 11) * 1e9:
        pulse = wf.square(1e-6) >> 95e-6
        w1 += pulse * wf.cos(2 * pi * freq)
        w2.append(pulse * wf.cos(2 * pi * freq))
        assert w1 == wf.WaveVStack(w2).simplify()
    rng = np.random.default_rng(0)
    for freq in np.linspace(6.1, 6.5, 3) * 1e9:
        pulse = wf.square(1e-6) >> (95e-6 + rng.normal() * 1e-9)
        w1 += pulse * wf.cos(2 * pi * freq)
        w2.append(pulse * wf.cos(2 * pi * freq))
        assert w1 == wf.WaveVStack(w2).simplify()
    w1 += wf.cos(2 * pi * freq * 0.9)
    w2.append(wf.cos(2 * pi * freq * 0.9))
    assert w1 == wf.WaveVStack(w2).simplify()
    assert wave_sum([((-1.0, np.inf), (((), ()), ((((), ()), ), (0.02, )))),
                     ((-1.0, np.inf), (((), ()), ((((), ()), ), (-0.02, ))))
                     ]) == ((np.inf, ), (((), ()), ))
    wlist = [wf.cos(1), wf.sin(2), wf.gaussian(3), wf.poly([1, -1 / 2, 1 / 6, -1 / 12])]
    tot = wf.zero()
    for w in wlist:
        tot += w
    assert wf.WaveVStack(wlist).simplify() == tot     # reference test_wavevstack.py:16-17
    assert wf.Waveform.fromlist(cases.CASES['ref_tolist'][0](wf).tolist()) == \
        cases.CASES['ref_tolist'][0](wf)              # reference test_waveform.py:50


def test_marker_mask_logic_interp():
    w = (wf.gaussian(4) >> 3) + (wf.square(2) << 5)
    m = w.marker
    assert m.bounds == (-6.0, -4.0, 0.0, 6.0, np.inf) and m.seq[1] != m.seq[0]
    assert (w | wf.zero()) == m and (w & wf.one()) == m
    assert w.mask(0.5).bounds[0] == -6.5
    f = wf.interp([0.0, 1.0, 3.0], [0.0, 2.0, -1.0])
    assert f.bounds == (0.0, 1.0, 3.0, np.inf)
    band = (wf.cos(2) + wf.cos(9) + 1).filter(low=5)
    assert band == wf.cos(9)


# ---- filter-design helpers of distortion.py against reference-generated vectors ----
def test_design_helpers_match_reference():
    import warnings
    import golden_io
    from waveforms_amd import distortion as d
    G = golden_io.npz('design.npz')
    for i, (n, fs, bw, skip) in enumerate(cases.extract_cases()):
        a, b = cases.extract_input(i)
        got = d.extractKernel(a, b, fs, bw, skip)
        assert got.shape == G[f'ek{i}'].shape
        assert np.max(np.abs(got - G[f'ek{i}'])) <= 1e-12 * np.abs(G[f'ek{i}']).max()
    for i, (amp, tau, fs) in enumerate(cases.decay_old_cases()):
        b, a = d.exp_decay_filter_old(amp, tau, fs)
        assert np.allclose(np.concatenate([b, a]), G[f'old{i}'], rtol=1e-13, atol=0)
    for i, (b, a) in enumerate(cases.factor_cases()):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')        # negative gain ** fraction -> nan, as in the reference
            secs = d.factor_filter(b, a)
        got = np.array([list(x) + list(y) for x, y in secs], dtype=complex)
        assert got.shape == G[f'fac{i}'].shape
        assert np.allclose(got, G[f'fac{i}'], rtol=1e-12, atol=1e-15, equal_nan=True)
    got = [d.stable_filter(f, fs) for f, fs in cases.stable_cases()]
    assert got == [bool(x) for x in G['stable']]
    from waveforms_amd.waveform import convolve
    assert convolve(wf.one(), wf.one()) is None      # a stub in the reference too (not exported)


def test_cli_argument_parsing():
    """The reference's CLI surface (waveforms/__main__.py:10-31): sub-command `sample`, option
    letters, defaults, and `--duration` taking effect only against the default stop."""
    from waveforms_amd.__main__ import _build_parser
    p = _build_parser()
    a = p.parse_args(['sample', 'gaussian(10) >> 5', 'out.npy'])
    assert (a.sample_rate, a.start, a.duration, a.stop, a.amplitude) == (44100, 0, -1, 1, 1)
    a = p.parse_args(['sample', '-S', '10', '-a', '2', '-l', '5', '-A', '3', 'one()', 'o.npy'])
    assert (a.sample_rate, a.start, a.duration, a.stop, a.amplitude) == (10, 2, 5, 1, 3)
    assert all(isinstance(v, int) for v in (a.sample_rate, a.start, a.duration, a.stop, a.amplitude))
    a = p.parse_args(['sample', '--sample-rate', '2e9', '--stop', '1e-6', 'one()', 'o.npy'])
    assert a.sample_rate == 2e9 and a.stop == 1e-6
    with pytest.raises(SystemExit):
        p.parse_args(['play', 'one()'])


def test_marker_mask_or_and_match_reference():
    """marker / mask(edge) / | / & on every plain-Waveform case against flat lists produced by the
    real reference (tests/golden/logic.json, oracle/make_golden.py): the window conventions of
    mask() -- which piece of a run sets a bound -- are the reference's, quirks included."""
    import json
    import os
    import cases
    import golden_io
    with open(os.path.join(golden_io.GOLDEN, 'logic.json')) as f:
        gold = json.load(f)
    assert len(gold) >= 60
    for name, v in gold.items():
        w = cases.CASES[name][0](wf)
        other = cases.CASES[v['other']][0](wf)
        for key, got in (('marker', w.marker), ('mask0', w.mask()), ('mask_e', w.mask(0.37)),
                         ('or', w | other), ('and', w & other)):
            assert got.tolist() == [golden_io.dec(x) for x in v[key]], (name, key)


def test_zero_and_one_are_fresh_objects():
    """callers set .start / .stop / .sample_rate / .min / .max on the waveforms they get: what zero()
    and one() return must not be shared (the reference hands out module-level singletons,
    waveform.py:886-896 -- attributes set on one then show up on every later zero())"""
    import waveforms_amd as wf
    z = wf.zero()
    z.start, z.stop, z.sample_rate, z.max = 0.0, 1.0, 10.0, 0.5
    z2 = wf.zero()
    assert z2 is not z and z2.start is None and z2.sample_rate is None and z2.max == float('inf')
    o = wf.one()
    o.min = 0.25
    assert wf.one().min == -float('inf')
    assert wf.zero().tolist() == z2.tolist() and (wf.zero() + wf.one()).seq == wf.one().seq
