"""AWG-rate grids (1-5 GS/s: the regime of Waveform.sample, reference waveform.py:173-207) on the CPU:
both oracles against vectors of the REAL reference (tests/golden/awg.npz, oracle/make_golden.py awg),
and the host side of the short-piece tier: which plans take it, np.searchsorted indices bit-exact."""
import os

import numpy as np
import pytest

import cases
import golden_io
import waveforms_amd as wf
from oracle import c_oracle, np_oracle
from waveforms_amd import _engine, _flatten, workloads as wl

AWG = golden_io.npz('awg.npz')


def _case(name):
    build, rate, n = cases.AWG_CASES[name]
    return build(wf, rate), cases._awg_grid(n, rate)


def _edges(plan_or_none, prog, g):
    m0, m1 = prog.member_range(0)
    return np.concatenate([c_oracle.member_index(prog, m, grid=g) for m in range(m0, m1)])


@pytest.mark.parametrize('name', sorted(cases.AWG_CASES))
def test_oracles_match_reference_on_awg_grids(name):
    w, grid = _case(name)
    want = AWG[name + '.y']
    t = wl.make_grid(grid)
    assert len(t) == len(want)
    pk = max(1.0, float(np.abs(want).max()))
    got = np_oracle.call(w, t)
    assert got.dtype == want.dtype
    assert np.max(np.abs(got - want)) <= 1e-14 * pk
    prog = _flatten.flatten([w])
    g = _flatten.grid_from_desc(grid)
    assert np.array_equal(c_oracle.grid_values(g), t)       # bit-exact np.arange grid
    cplx = want.dtype == np.complex128
    assert np.max(np.abs(c_oracle.eval_grid(prog, g, cplx)[0] - want)) <= 1e-12 * pk
    assert np.array_equal(_edges(None, prog, g), AWG[name + '.edges'])


@pytest.mark.parametrize('name', sorted(cases.AWG_CASES))
def test_short_tier_is_selected_and_indices_are_exact(name):
    w, grid = _case(name)
    prog = _flatten.flatten([w])
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(prog, grid=g)            # host-only plan when no GPU is visible
    m0, m1 = prog.member_range(0)
    idx = np.concatenate([plan.member_index(m) for m in range(m0, m1)])
    assert np.array_equal(idx, AWG[name + '.edges'])
    # GAUSSIAN / COS / LINEAR / EXP factors fuse at every AWG rate: nothing is left to device libm
    assert plan.info.n_direct == 0 and plan.info.n_generic == 0, (plan.info.n_direct, plan.info.n_generic)
    assert plan.kernel_name().startswith('wfk_sample_short<'), plan.kernel_name()


@pytest.mark.parametrize('rate', [1e9, 2.4e9, 5e9])
def test_gaussian_drag_pulses_fuse_at_awg_rates(rate):
    # round-2 verdict: at 1-5 GS/s n_fused was 0 and the Gaussian went through device libm
    chans = [wl.awg_channel(wf, c, 20000, rate, duty30=(c == 1)) for c in range(2)]
    g = _flatten.grid_from_desc(wl.awg_grid(20000, rate))
    plan = _engine.Plan(_flatten.flatten(chans), grid=g)
    assert plan.info.n_direct == 0 and plan.info.n_generic == 0 and plan.info.n_fused > 0
    assert plan.kernel_name(np.float32).startswith('wfk_sample_short<float,')


def test_long_pieces_keep_the_lean_tier_and_env_overrides():
    prog = _flatten.flatten([wl.c2_channel(wf)])
    g = _flatten.grid_from_desc(wl.c2_grid(10**6))
    assert _engine.Plan(prog, grid=g).kernel_name().startswith('wfk_sample_lean<')
    prog2 = _flatten.flatten([wl.awg_channel(wf, 0, 20000, 2e9)])
    g2 = _flatten.grid_from_desc(wl.awg_grid(20000, 2e9))
    os.environ['WFK_SHORT'] = '0'
    try:
        assert not _engine.Plan(prog2, grid=g2).kernel_name().startswith('wfk_sample_short<')
    finally:
        del os.environ['WFK_SHORT']
    # a linear chirp is an op of the short tier too (quadratic-phase recurrence) ...
    w = wl.awg_channel(wf, 0, 20000, 2e9) + (wf.chirp(1e8, 2e8, 30e-9) >> 5e-6)
    p3 = _engine.Plan(_flatten.flatten([w]), grid=g2)
    assert p3.kernel_name() == 'wfk_sample_short<double,false,false,16,1>' and p3.info.n_generic == 0
    # ... and so is an exponential chirp (a closing multiplier of family 4: one exponential step + one sine per sample)
    w = wl.awg_channel(wf, 0, 20000, 2e9) + (wf.chirp(1e8, 2e8, 30e-9, type='exponential') >> 5e-6)
    p3 = _engine.Plan(_flatten.flatten([w]), grid=g2)
    assert p3.kernel_name() == 'wfk_sample_short<double,false,false,16,4>' and p3.info.n_generic == 0
    # ... pieces the tier cannot take (a sinc pulse: device libm) go to the general kernel in a second launch;
    # the rest of the plan keeps the short tier
    w = wl.awg_channel(wf, 0, 20000, 2e9) + ((wf.sinc(2e8) * wf.square(30e-9)) >> 5e-6)
    p3 = _engine.Plan(_flatten.flatten([w]), grid=g2)
    assert p3.kernel_name().startswith('wfk_sample_short<') and ' + wfk_sample<' in p3.kernel_name()
    assert p3.info.n_generic > 0 and p3.info.n_fused > 0
    # ... and a plan with nothing for it falls back whole
    p4 = _engine.Plan(_flatten.flatten([(wf.square(30e-9, edge=14e-9) >> 1e-6) * wf.sinc(3e7)]), grid=g2)
    assert not p4.kernel_name().startswith('wfk_sample_short<')


def test_tile_program_equals_flattening_the_repeated_list():
    chans = [wl.awg_channel(wf, c, 5000, 2e9) for c in range(3)]
    g = _flatten.grid_from_desc(wl.awg_grid(5000, 2e9))
    a = _flatten.tile_program(_flatten.flatten(chans, g), 4)
    b = _flatten.flatten(chans * 4, g)
    for k in a.arrays:
        assert a.arrays[k].dtype == b.arrays[k].dtype and np.array_equal(a.arrays[k], b.arrays[k]), k


@pytest.mark.parametrize('name', sorted(n for n in cases.AWG_CASES if n != 'cplx_2g'))
def test_chain_oracle_matches_reference_at_awg_rates(name):
    """predistort(wav(t), ker = C4's 1024 taps) on the AWG-rate grids: C oracle (sampler) + direct convolution
    against what the REAL reference produced (tests/golden/awg_c4.npz)"""
    build, rate, n = cases.AWG_CASES[name]
    want = golden_io.npz('awg_c4.npz')[name + '.z']
    g = _flatten.grid_from_desc(cases._awg_grid(n, rate))
    y = c_oracle.eval_grid(_flatten.flatten([build(wf, rate)]), g)[0]
    z = c_oracle.fir(y, wl.c4_kernel())[cases.awg_c4_subset(n)]
    assert z.shape == want.shape
    assert np.max(np.abs(z - want)) <= 1e-12 * max(1.0, np.abs(want).max())


def test_flat_top_edges_share_their_sampled_tables_when_they_sit_on_the_grid_alike():
    """square(width, edge) x one carrier at AWG rates: every edge is an own-term op over a table of m0 + m1 erf(v_k)
    sampled by the host (wfk_compile.cpp: erf_table).  Pulses whose edges fall on the sample grid alike share one table;
    pulses at arbitrary times get a table each; WFK_NO_SHORT_ERFTAB=1 keeps the closing-op form (device libm erf)."""
    rate, n = 2e9, 20000
    g = _flatten.grid_from_desc(wl.awg_grid(n, rate))

    def plan(jitter):
        rng = np.random.default_rng(5)
        w = wf.zero()
        for k in range(150):
            at = (k + 0.5) * 60e-9 + (rng.uniform(0, 0.5e-9) if jitter else 0.0)
            w = w + ((wf.square(36e-9, edge=3e-9) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6))) >> at)
        return _engine.Plan(_flatten.flatten([w]), grid=g)
    aligned, jittered = plan(False), plan(True)
    assert aligned.kernel_name() == jittered.kernel_name() == 'wfk_sample_short<double,false,false,16,2>'
    assert aligned.info.n_generic == 0 and jittered.info.n_generic == 0
    # 300 edges of 12 samples: 16 B per sample when nothing is shared
    assert jittered.table_bytes() - aligned.table_bytes() >= 250 * 12 * 16
    os.environ['WFK_NO_SHORT_ERFTAB'] = '1'
    try:
        assert plan(False).kernel_name() == 'wfk_sample_short<double,false,false,16,1>'
    finally:
        del os.environ['WFK_NO_SHORT_ERFTAB']
