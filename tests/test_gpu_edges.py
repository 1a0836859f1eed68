"""Edge inputs of the drop-in entry points on the device, against vectors produced by the
real reference (tests/golden/edges.npz, oracle/make_golden.py): empty / one-sample /
scalar x, windows that miss every piece, duplicated and non-uniform x, bounds that fall
exactly on samples (in every kernel tier), ns..ks time scales, more pieces than samples in
a tile, complex channels whose window holds no complex piece, clip, +-inf-only bounds, and
ragged sizes around the wave / workgroup / tile widths in every output dtype."""
import numpy as np
import pytest

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import np_oracle
from waveforms_amd import _engine, _flatten

pytestmark = pytest.mark.gpu

EDGES = golden_io.npz('edges.npz')


def close(got, want, tol=1e-9):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype, \
        (got.shape, want.shape, got.dtype, want.dtype)
    if want.size:
        pk = max(1.0, float(np.max(np.abs(want))))
        assert float(np.max(np.abs(got - want))) <= tol * pk


@pytest.mark.parametrize('name', sorted(cases.edge_cases()))
def test_edge_inputs_match_reference(name):
    build, xs = cases.edge_cases()[name]
    w = build(wf)
    for k, x in enumerate(xs):
        close(w(x), EDGES[f'{name}.{k}'])


def test_scalar_x():
    for name in ('pulse', 'vstack', 'complex'):
        w = cases.edge_cases()[name][0](wf)
        for x in (0.0, 0.7, 3, -40.0):
            got = w(x)
            assert np.ndim(got) == 0
            assert abs(got - np_oracle.call(w, np.array([float(x)]))[0]) <= 1e-12


def test_frag_outside_support():
    sq = wf.square(1.0) >> 5.0
    t = np.linspace(0, 4, 4097)
    assert not sq(t).any()
    assert sq(t, frag=True) == []


def test_boundaries_exactly_on_samples_every_tier(monkeypatch):
    # np.searchsorted side='left' puts t == bound in the later piece, in every kernel tier
    w = cases.edge_cases()['aligned'][0](wf)
    t = np.linspace(0, 8, 8 * 64 + 1)          # 1/64 spacing: bounds 0, 2, 3, 4 are samples
    want = EDGES['aligned.0']
    idx = np.searchsorted(t, w.bounds)
    for env in ({}, {'WFK_DISABLE_LEAN': '1'}, {'WFK_DISABLE_FUSE': '1'}, {'WFK_DISABLE_FAST': '1'}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for plan in (_engine.Plan(_flatten.flatten([w]), grid=_flatten.grid_linspace(0, 8, 513)),
                     _engine.Plan(_flatten.flatten([w]), t=t)):
            assert np.array_equal(plan.member_index(0), idx)
            close(plan.run_host(np.float64)[0], want)
            plan.close()
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 1023, 1024, 1025, 4097, 16 * 1024 + 1])
def test_ragged_sizes_all_dtypes(n):
    from waveforms_amd._sampling import BatchSampler
    chans = [cases.edge_cases()['pulse'][0](wf) >> (0.1 * c) for c in range(3)]
    grid = ('linspace', -6.0, 6.0, n, True)
    t = np.linspace(-6.0, 6.0, n)
    want = np.stack([np_oracle.call(w, t) for w in chans])
    bs = BatchSampler(chans, grid)
    close(bs.to_host(np.float64), want)
    g32 = bs.to_host(np.float32)
    assert g32.dtype == np.float32
    assert float(np.max(np.abs(g32 - want))) <= FP32_TOL
    gc = bs.to_host(np.complex128)
    assert gc.dtype == np.complex128 and not gc.imag.any()
    close(gc.real, want)
    bs.close()
