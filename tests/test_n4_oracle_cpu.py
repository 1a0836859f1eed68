"""SURVEY 8(f) N4 without a GPU: the oracle (oracle/np_oracle.py) evaluating the trees the PRODUCT's symbolic
layer builds -- wave_eval texts, simplify(), filter(), marker / mask / | / &, interp() -- against what the real
reference sampled for the twin trees (tests/golden/n4.npz, oracle/make_golden.py: make_n4).  Pins both the
oracle and the host front-end on these trees; the device side is tests/test_gpu_frontend.py."""
import numpy as np
import pytest

import cases
import golden_io
import waveforms_amd as wf
from oracle import np_oracle
from waveforms_amd import wave_eval, workloads as wl

N4 = golden_io.npz('n4.npz')
TOL = 1e-12


def close(got, want, tol=TOL):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    if not np.iscomplexobj(want):
        assert np.all(np.imag(got) == 0)
        got = np.real(got)
    pk = max(1.0, float(np.max(np.abs(want), initial=0.0)))
    assert float(np.max(np.abs(got - want), initial=0.0)) <= tol * pk


@pytest.mark.parametrize('name', sorted(cases.PARSER_CASES))
def test_parsed_text_samples_like_the_reference_twin(name):
    text, twin, grid = cases.PARSER_CASES[name]
    w = wave_eval(text)
    assert w.tolist() == twin(wf).simplify().tolist()
    close(np_oracle.call(w, wl.make_grid(grid)), N4[f'parse.{name}'])


@pytest.mark.parametrize('name', sorted(k[5:] for k in N4.files if k.startswith('simp.')))
def test_simplified_tree_samples_like_the_reference(name):
    build, grid = cases.CASES[name]
    t = wl.make_grid(grid)
    close(np_oracle.call(build(wf).simplify(), t), N4[f'simp.{name}'])


@pytest.mark.parametrize('name', sorted(cases.FILTER_CASES))
def test_filtered_tree(name):
    build, lo, hi, grid = cases.FILTER_CASES[name]
    close(np_oracle.call(build(wf).filter(lo, hi), wl.make_grid(grid)), N4[f'filter.{name}'])


@pytest.mark.parametrize('name', sorted(cases.INTERP_CASES))
def test_interp_tree(name):
    build, grid = cases.INTERP_CASES[name]
    close(np_oracle.call(build(wf), wl.make_grid(grid)), N4[f'interp.{name}'])


def test_marker_mask_logic_values():
    import json
    import os
    with open(os.path.join(golden_io.GOLDEN, 'logic.json')) as f:
        gold = json.load(f)
    names = cases.n4_logic_names()
    assert len(names) >= 60
    for name in names:
        w, t = cases.CASES[name][0](wf), wl.make_grid(cases.CASES[name][1])
        other = cases.CASES[gold[name]['other']][0](wf)
        for key, tree in (('marker', w.marker), ('mask0', w.mask()), ('mask_e', w.mask(0.37)),
                          ('or', w | other), ('and', w & other)):
            got = np.real(np_oracle.call(tree, t))
            assert np.array_equal(got, N4[f'logic.{name}.{key}'].astype(np.float64)), (name, key)


@pytest.mark.parametrize('name', sorted(cases.CLI_CASES))
def test_cli_twin(name):
    argv, text, twin, start, stop, rate, amp = cases.CLI_CASES[name]
    w = wave_eval(text)
    assert w.tolist() == twin(wf).simplify().tolist()
    w.start, w.stop, w.sample_rate = start, stop, rate
    close(np_oracle.sample(w) * amp, N4[f'cli.{name}'])
