"""Front-end parity: the symbolic tree this package
builds for a pulse script flattens to exactly the list the reference's
`tolist()` produced for the same script (golden: oracle/make_golden.py)."""
import numpy as np
import pytest

import cases
import golden_io
import waveforms_amd as wf

GOLD = golden_io.frontend_lists()


def same(a, b):
    if isinstance(a, tuple) or isinstance(b, tuple):
        return (isinstance(a, tuple) and isinstance(b, tuple) and len(a) == len(b)
                and all(same(x, y) for x, y in zip(a, b)))
    if a is None or b is None:
        return a is b
    return a == b and isinstance(a, complex) == isinstance(b, complex)


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_flat_list_matches_reference(name):
    build, _ = cases.CASES[name]
    got = build(wf).tolist()
    want = GOLD[name]
    assert len(got) == len(want)
    bad = [i for i, (g, w) in enumerate(zip(got, want)) if not same(g, w)]
    assert not bad, (bad[:5], [got[i] for i in bad[:5]], [want[i] for i in bad[:5]])


def test_reference_golden_list():
    # reference tests/test_waveform.py:38-48 (known answer held by the reference)
    l = cases.CASES['ref_tolist'][0](wf).tolist()
    assert l == [
        np.inf, -np.inf, None, None, None, None, 5, -2.5, 0, 12.5, 1, 1.0, 2,
        1, 3, 2, 3.0028060219661246, 5, 1, 3, 4, 200, 0.0, 42.5, 0, 57.5, 1,
        1.0, 2, 1, 3, 2, 3.0028060219661246, 50, 1, 3, 4, 200, 0.0, np.inf, 0
    ]
    w2 = wf.Waveform.fromlist(l)
    assert w2.tolist() == l
    assert wf.Waveform.fromtree(w2.totree()).tolist() == l


def test_reference_golden_tree():
    # reference tests/test_waveform.py:51-65
    t = cases.CASES['ref_tolist'][0](wf).totree()
    assert t == ((np.inf, -np.inf, None, None, None, None),
                 ((-2.5, ()), (12.5, ((1.0, ((1, (2, 3.0028060219661246, 5)),
                                             (1, (4, 200, 0.0)))), )),
                  (42.5, ()), (57.5, ((1.0, ((1, (2, 3.0028060219661246, 50)),
                                             (1, (4, 200, 0.0)))), )),
                  (np.inf, ())))


def test_reference_golden_vstack_list():
    # reference tests/test_wavevstack.py:29-43
    w = cases.CASES['vstack4'][0](wf)
    l = w.tolist()
    assert l == [
        None, None, 0, 0, None, None, 4, 1, np.inf, 1, 1.0, 1, 1, 3, 4, 1, 0.0,
        1, np.inf, 1, 1.0, 1, 1, 3, 4, 2, 0.7853981633974483, 3, -2.25, 0,
        2.25, 1, 1.0, 1, 1, 3, 2, 0.9008418065898374, 0, np.inf, 0, 1, np.inf,
        4, 1, 0, -0.5, 1, 1, 2, 1, 0, 0.16666666666666666, 1, 2, 2, 1, 0,
        -0.08333333333333333, 1, 3, 2, 1, 0
    ]
    w2 = wf.WaveVStack.fromlist(l)
    assert isinstance(w2, wf.WaveVStack) and w2.wlist == w.wlist


def test_readme_appendix_e():
    # SURVEY.md Appendix E known answers (oracle-captured)
    from waveforms_amd import workloads as wl
    x, y = wl.readme_xy(wf)
    assert x.bounds == (-1e-08, 1e-08, 9.9e-07, 1.01e-06, 1.99e-06, 2.01e-06,
                        np.inf)
    l = x.tolist()
    assert len(l) == 171
    assert l[:20] == [np.inf, -np.inf, None, None, None, None, 7, -1e-08, 0,
                      1e-08, 5, 6283185.307179587, 1, 1, 3, 4,
                      125663706.14359173, -2.5e-08, 6283185.307179587, 2]
