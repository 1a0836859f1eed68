"""Stateless closing multipliers of the lean kernel (family 3): a piece whose terms share ONE
`samplingPoints` table (INTERP, reference _waveform.pyx:309-311: np.interp on linspace knots) or ONE mollifier
(_waveform.pyx:359-363) runs its carriers as fused ops and multiplies what they accumulated by that envelope --
one 16-byte gather or one inline exponential per sample instead of a term-by-term pass of the general kernel.
Checked against the C oracle and against the same plan with the multipliers off (WFK_DISABLE_FMUL=1)."""
import os

import numpy as np
import pytest

from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl
from waveforms_amd.waveform import _window, primitive, INTERP

pytestmark = pytest.mark.gpu
SPAN = wl.SPAN


def run(chans, grid, dtype=np.float64, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        plan = _engine.Plan(_flatten.flatten(chans), grid=_flatten.grid_from_desc(grid))
        return plan.run_host(dtype), plan.kernel_name(dtype), plan.info
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def oracle(chans, grid, cplx=False):
    return c_oracle.eval_grid(_flatten.flatten(chans), _flatten.grid_from_desc(grid), cplx)


def hann(m=1000, a=1.0):
    return wf.samplingPoints(-SPAN / 2, SPAN / 2, np.hanning(m) * a)


GRID = ('linspace', 0.0, 3e-6, 1_500_000, False)


def both_ways(chans, grid=GRID, tol=1e-12, cplx=False, fam3=True):
    dt = np.complex128 if cplx else np.float64
    got, name, info = run(chans, grid, dt)
    if fam3:
        assert name.startswith('wfk_sample_lean<') and name.endswith(',3>'), name
        assert info.n_generic == 0 and info.n_direct == 0
    ref = oracle(chans, grid, cplx)
    pk = max(1.0, np.abs(ref).max())
    err = np.max(np.abs(got - ref))
    assert err <= tol * pk, (err, pk)
    off, name_off, info_off = run(chans, grid, dt, env={'WFK_DISABLE_FMUL': '1'})
    assert ',3>' not in name_off and info_off.n_generic > 0
    assert np.max(np.abs(off - ref)) <= 1e-12 * pk
    return got, ref


def test_bench_shapes():
    """bench.py's `direct_interp` and `direct_mollifier` rows (workloads.direct_channel)."""
    for shape in ('interp', 'mollifier'):
        chans = [wl.direct_channel(wf, shape, c) for c in range(2)]
        both_ways(chans)


def test_table_under_several_carriers_and_next_to_other_terms():
    env = hann()
    chans = [
        (env * (wf.cos(2e9) + 0.3 * wf.cos(3e9, 0.3))) >> 1e-6,                       # two carriers, one multiplier
        ((env * wf.cos(2e9)) >> 1e-6) + (wf.gaussian(20e-9) >> 1.005e-6),             # unmodulated terms behind it
        (hann(17, 0.7) >> 0.5e-6) + (hann(4096) * wf.cos(1e9, 1.0) >> 2e-6),          # coarse and fine tables
        0.25 * env >> 2.5e-6,                                                         # the table alone
    ]
    both_ways(chans)


def test_clamped_continuation_and_offsets():
    """np.interp holds the end values outside [start, stop]; a window wider than the table samples that."""
    pts = tuple(np.linspace(0.2, 1.0, 33) ** 2)
    w = _window(-40e-9, 50e-9, primitive(INTERP, -10e-9, 20e-9, pts)) >> 1e-6
    chans = [w, (w * wf.cos(5e8)) + 0.1, (w >> 1e-6) * 2 - 0.5]
    got, ref = both_ways(chans)
    g = np.linspace(0.0, 3e-6, 1_500_000, endpoint=False)
    left = (g > 1e-6 - 39e-9) & (g < 1e-6 - 11e-9)
    assert np.all(got[0][left] == pts[0]) and np.all(ref[0][left] == pts[0])


def test_mollifier_with_plateau_and_carrier():
    chans = [
        wf.mollifier(20e-9, plateau=30e-9) >> 1e-6,
        (wf.mollifier(40e-9) * wf.cos(2e9, 0.4)) >> 2e-6,
        (0.5 * wf.mollifier(40e-9) * (wf.cos(2e9) + wf.cos(2.5e9))) >> 0.5e-6,
    ]
    both_ways(chans)


def test_complex_amplitudes_and_float_outputs():
    env = hann()
    chans = [((0.5 + 0.3j) * env * wf.cos(2e9)) >> 1e-6, ((0.2 - 0.1j) * wf.mollifier(30e-9) * wf.cos(1e9)) >> 2e-6]
    got, ref = both_ways(chans, cplx=True)
    re, name, _ = run(chans, GRID, np.float64)                                  # real launch of complex channels
    assert name.endswith(',3>') and np.max(np.abs(re - ref.real)) <= 1e-12
    c64, name64, _ = run(chans, GRID, np.complex64)
    assert name64.startswith('wfk_sample_lean<float,true') and name64.endswith(',3>')
    assert np.max(np.abs(c64 - ref)) <= FP32_TOL
    chans = [wl.direct_channel(wf, 'interp'), wl.direct_channel(wf, 'mollifier')]
    f32, name32, _ = run(chans, GRID, np.float32)
    assert name32.startswith('wfk_sample_lean<float,false') and name32.endswith(',3>')
    ref = oracle(chans, GRID)
    assert np.max(np.abs(f32 - ref)) <= FP32_TOL * np.abs(ref).max()


def test_what_the_multiplier_does_not_take():
    """Two different tables in one piece, a table to a power, non-finite table values, derivative orders of the
    mollifier: the general kernel's per-factor paths, as before."""
    env = hann()
    for w in ((env * hann(100)) >> 1e-6,
              (env ** 2) >> 1e-6,
              wf.samplingPoints(-SPAN / 2, SPAN / 2, np.r_[np.hanning(50), np.inf, np.hanning(50)]) >> 1e-6,
              wf.mollifier(20e-9, d=1) >> 1e-6,
              ((env * (wf.cos(2e9) + wf.cos(2.5e9))) >> 1e-6) + (hann(100) >> 1.004e-6)):   # (two tables overlapping, one over two carriers)
        got, name, info = run([w], GRID)
        if info.n_fused:
            assert name.startswith('wfk_sample_lean<double,false,16,false,3> + wfk_sample<'), name   # a mixed plan
        else:
            assert ',3>' not in name, name
        ref = oracle([w], GRID)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), fin)
        assert np.max(np.abs(got[fin] - ref[fin])) <= 1e-9 * max(1.0, np.abs(ref[fin]).max())


def test_far_from_the_origin_and_endpoint_grids():
    env = hann()
    w = (env * wf.cos(2e8, 0.3)) >> 1e-6
    for t0 in (1e-3, -2e-2):
        grid = ('linspace', t0, t0 + 3e-6, 1_000_001, True)
        got, name, info = run([w >> t0], grid)
        ref = oracle([w >> t0], grid)
        assert np.max(np.abs(got - ref)) <= 1e-9


# ---- the same multipliers in the short tier (AWG sample rates: a 30 ns pulse is 60 samples) -----------------
AWG = wl.awg_grid(100_000, 2e9)
AWG_TOL = 5e-11        # of peak, as tests/test_gpu_awg.py: grid jitter x carrier out to t = 50 us (measured <= 2e-11)


def test_optimised_pulses_at_awg_rates_run_on_the_short_tier():
    chans = [wl.awg_interp_channel(wf, c) for c in range(3)]
    got, name, info = run(chans, AWG)
    assert name == 'wfk_sample_short<double,false,false,16,2>' and info.n_generic == 0 and info.n_fused > 0, name
    ref = oracle(chans, AWG)
    assert np.max(np.abs(got - ref)) <= AWG_TOL * np.abs(ref).max()
    # one copy per distinct table: 3 channels x 8 shapes x 302 entries of 16 B (+ the op records), not one per pulse
    plan = _engine.Plan(_flatten.flatten(chans), grid=_flatten.grid_from_desc(AWG))
    assert plan.table_bytes() < 2_000_000            # (records 3 x 1666 x 224 B, tables 3 x 8 x 302 x 16 B; 24 MB with one copy per pulse)
    off, name_off, info_off = run(chans, AWG, env={'WFK_DISABLE_FMUL': '1'})
    assert name_off.startswith('wfk_sample<') and info_off.n_generic > 0
    assert np.max(np.abs(off - ref)) <= AWG_TOL * np.abs(ref).max()
    f32, name32, _ = run(chans, AWG, np.float32)
    assert name32 == 'wfk_sample_short<float,false,false,16,2>'
    assert np.max(np.abs(f32 - ref)) <= FP32_TOL * np.abs(ref).max()


def test_short_tier_mollifiers_complex_amplitudes_offsets_and_mixed_plans(monkeypatch):
    # (a short plan that hands more than 5 % of its samples on is evaluated pointwise as a whole: keep the mixed form here)
    monkeypatch.setenv('WFK_KEEP_MIXED_SHORT', '1')
    rng = np.random.default_rng(5)
    moll = wl._tree_sum([rng.uniform(0.3, 1) * (wf.mollifier(30e-9) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6)))
                         >> ((k + 0.5) * 30e-9) for k in range(1600)])
    cplx = wl._tree_sum([complex(rng.uniform(-1, 1), rng.uniform(-1, 1)) * hann(64) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8))
                         >> ((k + 0.5) * 30e-9) for k in range(1600)])
    chans = [moll + 0.25, cplx, (wl.awg_interp_channel(wf, 7) >> 0.2e-9) - 0.5,
             wl.awg_channel(wf, 3) + (hann(200) * wf.cos(1e9) >> 10e-6) +
             (hann(100) * (wf.cos(2e9) + wf.cos(3e9)) >> 10.004e-6)]   # two tables overlap, one of them over two carriers: generic there
    got, name, info = run(chans, AWG, np.complex128)
    assert name.startswith('wfk_sample_short<double,true,') and ' + wfk_sample<' in name, name     # (a mixed short plan)
    ref = oracle(chans, AWG, True)
    assert np.max(np.abs(got - ref)) <= AWG_TOL * np.abs(ref).max()
    re, _, _ = run(chans, AWG, np.float64)
    assert np.max(np.abs(re - ref.real)) <= AWG_TOL * np.abs(ref).max()
    # windows wider than the table (the clamped continuation), pulses cut by the end of the grid, a clip
    pts = tuple(np.linspace(0.3, 1.0, 40) ** 2)
    wide = wl._tree_sum([_window(-14e-9, 15e-9, primitive(INTERP, -5e-9, 9e-9, pts)) * wf.cos(3e8 * (k % 7), 0.1 * k)
                         >> ((k + 0.5) * 30e-9) for k in range(1700)])
    clipped = wide >> 7e-9
    clipped.max, clipped.min = 0.6, -0.4
    chans = [wide, wf.WaveVStack([wide, moll]), clipped]
    got, name, _ = run(chans, AWG)
    assert name.startswith('wfk_sample_short<'), name
    ref = oracle(chans, AWG)
    assert np.max(np.abs(got - ref)) <= AWG_TOL * max(1.0, np.abs(ref).max())


def test_fir_chain_keeps_its_fused_form_next_to_table_envelopes():
    """predistort(wav(t), ker) at AWG rates: fir_short samples the Gaussian pieces itself; the table / mollifier
    pieces are written by the general kernel first (fir_short has no closing multipliers of this kind)."""
    from waveforms_amd.distortion import SampledFir
    ker = np.random.default_rng(1).normal(size=257)
    ker /= np.abs(ker).sum()
    chans = [wl.awg_channel(wf, 0) + (hann(300) * wf.cos(1e9) >> 20.015e-6), wl.awg_interp_channel(wf, 1, 20_000)]
    grid = wl.awg_grid(20_000, 2e9)
    sf = SampledFir(chans, grid, ker)
    x = oracle(chans, grid)
    want = np.stack([c_oracle.fir(r, ker) for r in x])
    assert np.max(np.abs(sf.to_host() - want)) <= AWG_TOL * np.abs(want).max()
    assert 'fir_short<' in sf.plan.kernel_name() or '+ FIR' in sf.plan.kernel_name()
    sf.close()


def test_overlapping_envelopes_in_short_pieces():
    """Crosstalk-compensated channels: a channel is its own pulses plus scaled copies of its neighbours' -- pulses of
    DIFFERENT table / mollifier envelopes overlapping in time.  In a short piece every envelope x carrier term is an
    own-term op (acc += F (A cos + B sin)), so any number of them share a piece; envelopes over several carriers or
    with complex amplitudes keep the single-envelope forms or go to the exact path."""
    own = [wl.awg_interp_channel(wf, c, 40_000) for c in range(3)]
    rng = np.random.default_rng(8)
    moll = wl._tree_sum([rng.uniform(0.3, 1) * (wf.mollifier(30e-9) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6)))
                         >> ((k + 0.5) * 30e-9 + 11e-9) for k in range(600)])
    chans = [own[0] + 0.07 * (own[1] >> 3e-9) - 0.04 * (own[2] >> 7.5e-9),        # three table trains, shifted against each other
             own[1] + 0.1 * moll,                                                 # tables and mollifiers overlapping
             own[2] + 0.05 * own[0] + (wf.gaussian(20e-9) * wf.cos(1e9) >> 5.01e-6)]
    grid = wl.awg_grid(40_000, 2e9)
    got, name, info = run(chans, grid)
    assert name == 'wfk_sample_short<double,false,false,16,2>' and info.n_generic == 0, (name, info.n_generic)
    ref = oracle(chans, grid)
    assert np.max(np.abs(got - ref)) <= AWG_TOL * np.abs(ref).max()
    off, name_off, info_off = run(chans, grid, env={'WFK_NO_SHORT_MULTI': '1'})
    assert info_off.n_generic > 0
    assert np.max(np.abs(off - ref)) <= AWG_TOL * np.abs(ref).max()
    f32, _, _ = run(chans, grid, np.float32)
    assert np.max(np.abs(f32 - ref)) <= FP32_TOL * np.abs(ref).max()
    # an envelope over TWO carriers next to another envelope: not an own-term op -- the exact path, same numbers
    two = (hann(300) * (wf.cos(2e9) + 0.5 * wf.cos(3e9))) >> 1e-6
    chans = [wl._tree_sum([two >> (k * 30e-9) for k in range(300)]) + 0.1 * (own[0] >> 1e-6)]
    got, name, info = run(chans, grid)
    ref = oracle(chans, grid)
    assert np.max(np.abs(got - ref)) <= AWG_TOL * max(1.0, np.abs(ref).max())


def test_overlapping_envelopes_on_fine_grids():
    """The lean kernel's own-term op (WFK_FCE_OWNMUL): overlapping pulses of different table / mollifier envelopes, each
    over one carrier (or none), complex amplitudes through the single-envelope forms or the exact path."""
    rng = np.random.default_rng(13)

    def train(mk, n, period, t0=0.0):
        return wl._tree_sum([rng.uniform(0.3, 1) * mk(k) >> (t0 + (k + 0.5) * period) for k in range(n)])
    a = train(lambda k: hann(300 + k) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6)), 60, 50e-9)
    b = train(lambda k: hann(64, 0.8) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8)), 55, 54e-9, 7e-9)
    m = train(lambda k: wf.mollifier(40e-9) * wf.cos(2 * np.pi * rng.uniform(-1e8, 1e8), 0.3 * k), 50, 60e-9, 3e-9)
    chans = [a + 0.1 * b, a + 0.05 * b - 0.2 * m, 0.3 * hann(500) * 1.0 >> 1e-6, b + (wf.gaussian(20e-9) * wf.cos(1e9) >> 1.5e-6) + 0.1 * m]
    chans[2] = chans[2] + (0.2 * hann(100) >> 1.004e-6)          # two bare tables (no carrier) overlapping
    grid = ('linspace', 0.0, 3e-6, 1_500_000, False)
    got, name, info = run(chans, grid)
    assert name == 'wfk_sample_lean<double,false,16,false,4>' and info.n_generic == 0, (name, info.n_generic)
    ref = oracle(chans, grid)
    pk = np.abs(ref).max()
    assert np.max(np.abs(got - ref)) <= 1e-12 * pk
    off, name_off, info_off = run(chans, grid, env={'WFK_NO_LEAN_MULTI': '1'})
    assert info_off.n_generic > 0 and np.max(np.abs(off - ref)) <= 1e-12 * pk
    f32, name32, _ = run(chans, grid, np.float32)
    assert name32.startswith('wfk_sample_lean<float,') and np.max(np.abs(f32 - ref)) <= FP32_TOL * pk
    # complex amplitudes: two groups per term -- not an own-term op; same numbers on whatever path
    cplx = [(0.5 + 0.3j) * a + 0.1 * b]
    gotc, _, _ = run(cplx, grid, np.complex128)
    refc = oracle(cplx, grid, True)
    assert np.max(np.abs(gotc - refc)) <= 1e-12 * np.abs(refc).max()


@pytest.mark.parametrize('fine', [True, False])
def test_new_ops_under_vstacks_shifts_offsets_and_clips(fine):
    """Own-term envelope ops, closing multipliers and (AWG rates) chirps inside WaveVStack members with a channel time
    shift, offsets and clips: the same samples as the oracle on the lean kernel / the short tier."""
    rng = np.random.default_rng(31)
    period = 60e-9
    n_p = 40
    a = wl._tree_sum([rng.uniform(0.3, 1) * hann(200) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6))
                      >> ((k + 0.5) * period) for k in range(n_p)])
    b = wl._tree_sum([rng.uniform(0.3, 1) * wf.mollifier(40e-9) * wf.cos(2 * np.pi * rng.uniform(-1e8, 1e8))
                      >> ((k + 0.5) * period + 9e-9) for k in range(n_p)])
    c = wl._tree_sum([rng.uniform(0.3, 1) * wf.chirp(5e7, 2e8, 50e-9) * wf.cosPulse(50e-9)
                      >> ((k + 0.5) * period) for k in range(n_p)])
    stack = (wf.WaveVStack([a, 0.2 * b, 0.1 * c]) + 0.25) >> 3.3e-9
    clipped = (a + 0.3 * b) >> 1.7e-9
    clipped.max, clipped.min = 0.7, -0.6
    chans = [stack, clipped, (a + 0.1 * (a >> 11e-9)) - 0.4]
    span = n_p * period
    grid = ('linspace', 0.0, span, 1_200_000, False) if fine else ('arange', 0.0, span, 0.5e-9)
    got, name, info = run(chans, grid)
    assert name.startswith('wfk_sample_lean<' if fine else 'wfk_sample_short<'), name
    ref = oracle(chans, grid)
    tol = 1e-12 if fine else AWG_TOL
    assert np.max(np.abs(got - ref)) <= tol * max(1.0, np.abs(ref).max())
    f32, _, _ = run(chans, grid, np.float32)
    assert np.max(np.abs(f32 - ref)) <= FP32_TOL * max(1.0, np.abs(ref).max())


def test_time_slices_of_the_new_tiers_are_the_same_samples():
    """wfk_grid.i0: a slice of a grid is those samples of the whole grid -- for the closing multipliers / own-term ops of
    the lean kernel and for the short tier's table and chirp ops (time-sharded sampling)."""
    fine = [wl.direct_channel(wf, 'interp'), wl.direct_channel(wf, 'mollifier')]
    g = _flatten.grid_from_desc(GRID)
    whole = _engine.Plan(_flatten.flatten(fine), grid=g).run_host(np.float64)
    for a, b in ((0, 400_000), (400_000, 1_100_000), (1_100_000, 1_500_000)):
        part = _engine.Plan(_flatten.flatten(fine), grid=_flatten.grid_slice(g, a, b)).run_host(np.float64)
        assert np.max(np.abs(part - whole[:, a:b])) <= 1e-12 * np.abs(whole).max()
    awg = [wl.awg_interp_channel(wf, 0, 40_000), wl.awg_shape_channel(wf, 'linear_chirp', 1, 40_000)]
    g = _flatten.grid_from_desc(wl.awg_grid(40_000, 2e9))
    plan = _engine.Plan(_flatten.flatten(awg), grid=g)
    assert plan.kernel_name().startswith('wfk_sample_short<')
    whole = plan.run_host(np.float64)
    for a, b in ((0, 13_000), (13_000, 40_000)):
        part = _engine.Plan(_flatten.flatten(awg), grid=_flatten.grid_slice(g, a, b)).run_host(np.float64)
        assert np.max(np.abs(part - whole[:, a:b])) <= 1e-11 * np.abs(whole).max()
