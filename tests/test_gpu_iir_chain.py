"""sampler -> IIR (-> FIR) chain through the C-ABI (wfk_chain_iir_*): Waveform.sample(filters=(sos, initial))
(reference waveforms/waveform.py:190-203, chunked :244-251) and predistort(wav(t), filters, ker)
(waveforms/distortion.py:298-337) with the sampler INSIDE the single-pass IIR scan (iir_sampled).  Checked against
the oracle (C sampler) + SciPy's sosfilt / lfilter, the unfused path, and vectors the real reference produced
(tests/golden/iir.npz: iir_fused*)."""
import os

import numpy as np
import pytest
from scipy.signal import butter, lfilter, sosfilt, tf2sos

import cases
import golden_io
from cases import FP32_TOL, FP64_IIR_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl
from waveforms_amd.distortion import SampledIir, exp_decay_filter

pytestmark = pytest.mark.gpu
IIR = golden_io.npz("iir.npz")
TOL = FP64_IIR_TOL      # a blocked scan against SciPy's sequential recurrence (tests/cases.py); contract 1e-9


def _samples(chans, grid):
    return c_oracle.eval_grid(_flatten.flatten(chans), _flatten.grid_from_desc(grid))


def _cascade(sections, x, initial=0.0, zi=None):
    """the reference's filtering of one row: sections one after the other on x - initial, + initial; -> (y, zf)"""
    y = np.asarray(x, dtype=np.float64) - initial
    zf, off = [], 0
    for b, a in sections:
        m = max(len(b), len(a)) - 1
        z0 = np.zeros(m) if zi is None else np.asarray(zi[off:off + m], dtype=np.float64)
        y, z1 = lfilter(b, a, y, zi=z0)
        zf.append(z1)
        off += m
    return y + initial, np.concatenate(zf)


SHAPES = {
    'biquad': [(r[:3], r[3:]) for r in butter(2, 0.04, output='sos')],
    'two_biquads': [(r[:3], r[3:]) for r in butter(4, 0.03, output='sos')],
    'first_order': [exp_decay_filter(0.02, 150e-9, 2e9)],
    'four_first_order': [exp_decay_filter(A, tau, 2e9) for A, tau in ((0.03, 40e-9), (0.01, 900e-9), (-0.005, 20e-6), (0.02, 3e-7))],
    'order3': [butter(3, 0.05)],
    'order4': [butter(4, 0.08)],
}


@pytest.mark.parametrize('shape', sorted(SHAPES))
def test_fused_chain_matches_oracle_and_scipy(shape):
    """every single-pass shape with the sampler inside: fp64 against the C oracle's samples through SciPy"""
    secs = SHAPES[shape]
    chans = [wl.sum_channel(wf, 6, 1000 + c) for c in range(3)]
    # (>= four chunks of the fused scan: 64 lanes x 128 samples in the dot-product form, x 256 in the sweep form that
    #  filters with slow poles -- two_biquads here -- take)
    grid = ('linspace', 0.0, 6 * wl.SPAN, 70001, False)
    si = SampledIir(chans, grid, secs)
    assert si.fused, si.why_not
    assert si.plan.kernel_name().startswith('iir_sampled<double,'), si.plan.kernel_name()
    x = _samples(chans, grid)
    want = np.stack([_cascade(secs, row)[0] for row in x])
    got, zf = si.to_host(return_zf=True)
    assert np.max(np.abs(got - want)) <= TOL * max(1.0, np.abs(want).max())
    wzf = np.stack([_cascade(secs, row)[1] for row in x])
    assert np.max(np.abs(zf - wzf)) <= 1e-10 * max(1.0, np.abs(wzf).max())
    # the unfused path of the same plan type gives the same numbers
    os.environ['WFK_CHAIN_UNFUSED'] = '1'
    try:
        su = SampledIir(chans, grid, secs)
        assert not su.fused and 'WFK_CHAIN_UNFUSED' in su.why_not
        assert np.max(np.abs(su.to_host() - got)) <= TOL
        su.close()
    finally:
        del os.environ['WFK_CHAIN_UNFUSED']
    si.close()


def test_initial_level_and_carried_state():
    """out = F(x - initial) + initial from a given state zi; the final state continues the next grid slice exactly
    as the reference carries zi from chunk to chunk (waveform.py:244-251)"""
    sos = butter(4, 0.02, output='sos')
    secs = [(r[:3], r[3:]) for r in sos]
    chans = [0.3 + wl.sum_channel(wf, 9, 40 + c) for c in range(2)]
    grid = ('linspace', 0.0, 9 * wl.SPAN, 140000, False)
    x = _samples(chans, grid)
    zi = np.array([0.1, -0.2, 0.05, 0.0])
    si = SampledIir(chans, grid, sos)
    assert si.fused, si.why_not
    got, zf = si.to_host(initial=0.3, zi=zi, return_zf=True)
    for r in range(2):
        want, wzf = _cascade(secs, x[r], 0.3, zi)
        assert np.max(np.abs(got[r] - want)) <= TOL
        assert np.max(np.abs(zf[r] - wzf)) <= TOL
    si.close()
    # two halves of the grid, the state handed on == the whole grid
    g = _flatten.grid_from_desc(grid)
    halves, state = [], None
    for lo, hi in ((0, 70001), (70001, 140000)):
        sh = SampledIir(chans, _flatten.grid_slice(g, lo, hi), sos)
        assert sh.fused, sh.why_not
        y, state = sh.to_host(initial=0.3, zi=zi if state is None else state, return_zf=True)
        halves.append(y)
        sh.close()
    assert np.max(np.abs(np.concatenate(halves, axis=1) - got)) <= TOL


@pytest.mark.parametrize('n,endpoint', [(65536, False), (65537, True), (100003, False), (312345, True)])
def test_ragged_rows_piece_edges_inside_chunks_offsets_and_shifts(n, endpoint):
    """pieces far shorter and far longer than a 2048-sample sub-tile, gaps (zero pieces), a constant offset, a vstack
    with a time shift; rows that end inside a chunk / a sub-tile / a lane's block; the overridden last sample of an
    endpoint grid"""
    nseg = max(3, min(100, n // 3000))
    stack = wl.vstack_channel(wf, max(2, nseg // 5), 101) >> 7e-9            # merged members, shifted, with an offset
    stack = stack + 0.25
    chans = [wl.sum_channel(wf, nseg, 1000),                                  # ~1000-3000 samples per piece
             wl.sum_channel(wf, max(2, nseg // 3), 5, spacing=45e-9),         # gaps: zero pieces between the pulses
             stack,
             wl.sum_channel(wf, 3, 9)]                                        # three pieces, then nothing
    grid = ('linspace', 0.0, nseg * wl.SPAN, n, endpoint)
    secs = SHAPES['two_biquads']
    si = SampledIir(chans, grid, secs)
    assert si.fused, si.why_not
    x = _samples(chans, grid)
    want = np.stack([_cascade(secs, row, 0.1)[0] for row in x])
    got, zf = si.to_host(initial=0.1, return_zf=True)
    assert np.max(np.abs(got - want)) <= TOL
    assert np.max(np.abs(zf - np.stack([_cascade(secs, row, 0.1)[1] for row in x]))) <= TOL
    si.close()


def test_run_length_follows_the_scan_form_and_short_rows_fall_back():
    """the dot-product form (transition powers of order 1: butter(4, 0.1)) runs 128 samples per lane and fuses from 32768
    samples on; the sweep form (slow poles: butter(4, 0.03)) runs 256 per lane and needs 65536 -- shorter rows take
    sampler + IIR, same numbers"""
    chans = [wl.sum_channel(wf, 6, 1000 + c) for c in range(2)]
    grid = ('linspace', 0.0, 6 * wl.SPAN, 40000, False)
    x = _samples(chans, grid)
    for sos, fused in ((butter(4, 0.1, output='sos'), True), (butter(4, 0.03, output='sos'), False)):
        secs = [(r[:3], r[3:]) for r in sos]
        si = SampledIir(chans, grid, sos)
        assert si.fused == fused, si.why_not
        if not fused:
            assert 'four chunks' in si.why_not
        else:
            assert si.plan.kernel_name() == 'iir_sampled<double,2,2,true>'
        want = np.stack([_cascade(secs, row)[0] for row in x])
        assert np.max(np.abs(si.to_host() - want)) <= TOL * max(1.0, np.abs(want).max())
        si.close()


@pytest.mark.parametrize('n', [20000, 100000, 4097])
def test_awg_rate_rows_through_the_chain_plan(n):
    """Waveform.sample(filters=) at AWG sample rates (2 GS/s, 60-sample pulses: more than 16 pieces per chunk of the fused
    scan): the chain plan runs wfk_sample_short + the IIR stage -- Gaussian + DRAG trains back to back and at 30 % duty,
    flat tops with erf edges on an offset -- against the oracle's samples through SciPy, every single-pass shape"""
    chans = [wl.awg_channel(wf, c, n, 2e9, duty30=(c == 1)) for c in range(3)]
    chans.append(wl.awg_shape_channel(wf, 'flat_top', 3, n, 2e9) + 0.2)
    grid = wl.awg_grid(n, 2e9)
    x = _samples(chans, grid)
    for shape in ('two_biquads', 'four_first_order', 'order3'):
        secs = SHAPES[shape]
        si = SampledIir(chans, grid, secs)
        assert not si.fused and si.plan.kernel_name().startswith('wfk_sample_short<double,'), (si.plan.kernel_name(), si.why_not)
        got, zf = si.to_host(initial=0.1, return_zf=True)
        want = np.stack([_cascade(secs, row, 0.1)[0] for row in x])
        assert np.max(np.abs(got - want)) <= TOL * max(1.0, np.abs(want).max()), shape
        wzf = np.stack([_cascade(secs, row, 0.1)[1] for row in x])
        assert np.max(np.abs(zf - wzf)) <= TOL * max(1.0, np.abs(wzf).max()), shape
        si.close()


def test_float_rows():
    secs = SHAPES['two_biquads']
    chans = [wl.sum_channel(wf, 20, 300 + c) for c in range(3)]
    grid = ('linspace', 0.0, 20 * wl.SPAN, 150001, False)
    si = SampledIir(chans, grid, secs, dtype=np.float32)
    assert si.fused and si.plan.kernel_name().startswith('iir_sampled<float,'), (si.why_not, si.plan.kernel_name())
    x = _samples(chans, grid)
    want = np.stack([_cascade(secs, row)[0] for row in x])
    got = si.to_host()
    assert got.dtype == np.float32
    assert np.max(np.abs(got - want)) <= FP32_TOL * max(1.0, np.abs(want).max())
    si.close()


@pytest.mark.parametrize('K,per_row', [(33, False), (1024, False), (64, True)])
def test_three_stage_chain_sampler_iir_fir(K, per_row):
    """predistort(wav(t), filters, ker): the IIR pass (sampler inside) writes a workspace, the FIR reads it"""
    secs = [exp_decay_filter(A, tau, 2e9) for A, tau in ((0.03, 40e-9), (0.01, 900e-9))]
    chans = [wl.sum_channel(wf, 12, 77 + c) for c in range(3)]
    grid = ('linspace', 0.0, 12 * wl.SPAN, 90001, False)
    rng = np.random.default_rng(K)
    ker = rng.normal(size=(3, K) if per_row else K)
    ker /= np.abs(ker).sum(axis=-1, keepdims=True)
    si = SampledIir(chans, grid, secs, ker=ker)
    assert si.fused and si.plan.kernel_name().endswith('+ FIR'), (si.why_not, si.plan.kernel_name())
    x = _samples(chans, grid)
    want = np.stack([c_oracle.fir(_cascade(secs, row, 0.2)[0], ker[r] if per_row else ker) for r, row in enumerate(x)])
    assert np.max(np.abs(si.to_host(initial=0.2) - want)) <= TOL
    si.close()


def test_plans_that_cannot_fuse_run_sampler_then_filter():
    secs = SHAPES['biquad']
    grid = ('linspace', 0.0, 6 * wl.SPAN, 40000, False)
    base = wl.sum_channel(wf, 6, 11)
    clipped = wl.sum_channel(wf, 6, 12)
    clipped.max, clipped.min = 0.4, -0.3
    for chans, word in (([base, clipped], 'clip'),
                        ([base * (1 + 0.5j)], 'complex'),
                        ([(wf.square(40e-9, edge=8e-9) >> 60e-9) * wf.cos(2 * np.pi * 90e6), base], 'fused'),
                        # a constant under pulses with gaps: the gap pieces hold one op, the pulse pieces two
                        ([wl.sum_channel(wf, 3, 5, spacing=45e-9) + 0.25], 'shapes')):
        si = SampledIir(chans, grid, secs)
        assert not si.fused and word in si.why_not, si.why_not
        x = _samples(chans, grid).real
        want = np.stack([_cascade(secs, row)[0] for row in x])
        assert np.max(np.abs(si.to_host() - want)) <= 1e-10
        si.close()
    # a cascade whose first pass is not a single-pass shape (three biquads on a big batch: three launches)
    sos6 = butter(6, 0.05, output='sos')
    si = SampledIir([base], ('linspace', 0.0, 6 * wl.SPAN, 2_000_000, False), sos6)
    x = _samples([base], ('linspace', 0.0, 6 * wl.SPAN, 2_000_000, False))
    assert np.max(np.abs(si.to_host()[0] - sosfilt(sos6, x[0]))) <= 1e-10
    si.close()


def test_longer_cascades_fuse_their_first_pass():
    """eight first-order sections = two passes of four: the sampler runs inside the first, the second filters in place"""
    secs = [exp_decay_filter(0.01 * (k + 1), 30e-9 * 3 ** k, 2e9) for k in range(8)]
    chans = [wl.sum_channel(wf, 6, 500 + c) for c in range(2)]
    grid = ('linspace', 0.0, 6 * wl.SPAN, 70001, False)
    si = SampledIir(chans, grid, secs)
    assert si.fused and 'IIR passes' in si.plan.kernel_name(), (si.why_not, si.plan.kernel_name())
    x = _samples(chans, grid)
    want = np.stack([_cascade(secs, row, -0.1)[0] for row in x])
    got, zf = si.to_host(initial=-0.1, return_zf=True)
    assert np.max(np.abs(got - want)) <= 1e-10
    assert np.max(np.abs(zf - np.stack([_cascade(secs, row, -0.1)[1] for row in x]))) <= 1e-10
    si.close()


def test_persistent_waves_on_many_rows():
    """>= 64 rows: a bounded number of persistent waves per row walk the ticket counter"""
    secs = SHAPES['two_biquads']
    distinct = [wl.sum_channel(wf, 5, 900 + c) for c in range(4)]
    grid = ('linspace', 0.0, 5 * wl.SPAN, 120001, False)
    si = SampledIir(distinct, grid, secs, tile=20)           # 80 rows
    assert si.fused and si.n_channels == 80, si.why_not
    x = _samples(distinct, grid)
    want = np.stack([_cascade(secs, row)[0] for row in x])
    got = si.to_host()
    assert np.max(np.abs(got - np.tile(want, (20, 1)))) <= TOL
    si.close()


def test_lookback_timeout_falls_back_to_the_unfused_form():
    secs = SHAPES['two_biquads']
    chans = [wl.sum_channel(wf, 6, 1000 + c) for c in range(3)]
    grid = ('linspace', 0.0, 6 * wl.SPAN, 300_000, False)
    x = _samples(chans, grid)
    want = np.stack([_cascade(secs, row)[0] for row in x])
    si = SampledIir(chans, grid, secs)
    assert si.fused
    buf = _engine.DeviceBuffer(want.nbytes)
    os.environ['WFK_IIR_SPIN'] = '0'
    try:
        assert si.launch(buf.ptr) is True
        assert si.plan.status() is False                                       # reported ...
        assert np.isnan(buf.download(want.shape, np.float64)).any()            # ... and nothing plausible written
        assert not si.fused and 'timed out' in si.plan.why_not or not si.fused
        assert si.launch(buf.ptr) is True and si.plan.status() is True         # sampler + three launches now
        assert np.max(np.abs(buf.download(want.shape, np.float64) - want)) <= 1e-10
        # the drop-in call recovers by itself
        s2 = SampledIir(chans, grid, secs)
        assert np.max(np.abs(s2.to_host() - want)) <= 1e-10
        s2.close()
    finally:
        del os.environ['WFK_IIR_SPIN']
        buf.close()
        si.close()


@pytest.mark.parametrize('name', sorted(k for k in cases.iir_cases() if k.startswith('iir_fused')))
def test_sample_with_filters_takes_the_fused_chain(name):
    """Waveform.sample(filters=) on trees the chain fuses, against the real reference's output"""
    build, start, stop, rate, order, fc, initial = cases.iir_cases()[name]
    w = build(wf)
    w.start, w.stop, w.sample_rate = start, stop, rate
    b, a = butter(order, fc, 'lowpass', fs=rate)
    w.filters = (tf2sos(b, a), initial)
    want = IIR[name + '.full']
    got = w.sample()
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.abs(want).max())
    # and the plan behind it really is the fused one
    grid = _flatten.grid_arange(start, stop, 1 / rate)
    chain = _engine.ChainIirPlan(_flatten.flatten([w], grid), grid, [(r[:3], r[3:]) for r in tf2sos(b, a)])
    assert chain.fused and chain.kernel_name().startswith('iir_sampled<'), (chain.why_not, chain.kernel_name())
    chain.close()
    wantc = IIR[name + '.chunked']
    gotc = np.concatenate(list(w.sample(chunk_size=cases.iir_chunk(name))))
    assert np.max(np.abs(gotc - wantc)) <= 1e-10 * max(1.0, np.abs(wantc).max())
