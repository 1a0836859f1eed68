"""sampler -> FIR chain through the C-ABI (wfk_chain_*, BASELINE configs[3]):
predistort(wav(t), ker=ker) with the sampler fused into the FIR transform.  Checked against
the oracle (C sampler + direct convolution), the unfused two-kernel path, reference fixtures
(big.npz at 1e6 points, c4_full.npz at the full 1e7 points) and FIR properties.
reference: waveforms/waveform.py:529-563 -> waveforms/distortion.py:329-337."""
import numpy as np
import pytest

import golden_io
from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl
from waveforms_amd.distortion import SampledFir

pytestmark = pytest.mark.gpu
BIG = golden_io.npz('big.npz')


def _oracle_chain(chans, grid, ker):
    g = _flatten.grid_from_desc(grid)
    y = c_oracle.eval_grid(_flatten.flatten(chans), g)
    return np.stack([c_oracle.fir(row, ker) for row in y])


def _kernel(K, seed=1):
    ker = np.random.default_rng(seed).normal(size=K)
    return ker / np.abs(ker).sum()


@pytest.mark.parametrize('K', [1, 2, 100, 1024, 1025, 1026, 1537])
def test_fused_chain_matches_oracle(K):
    chans = [wl.sum_channel(wf, 6, 1000 + c) for c in range(3)]
    grid = ('linspace', 0.0, 6 * wl.SPAN, 50001, False)
    ker = _kernel(K)
    sf = SampledFir(chans, grid, ker)
    assert sf.fused, sf.why_not
    got = sf.to_host()
    want = _oracle_chain(chans, grid, ker)
    assert np.max(np.abs(got - want)) <= 1e-12
    sf.close()


def test_many_pieces_per_window_pair_and_ragged_ends():
    # pieces of ~4000 samples: every window pair (7168 samples) crosses piece edges (masked path);
    # n not a multiple of anything; endpoint grid (last sample overridden)
    chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(2)] + \
            [wl.sum_channel(wf, 37, 5, spacing=45e-9)]                      # gaps: zero pieces in between
    for n, endpoint in ((400001, False), (312345, True)):
        grid = ('linspace', 0.0, 100 * wl.SPAN, n, endpoint)
        ker = _kernel(1024, n)
        sf = SampledFir(chans, grid, ker)
        assert sf.fused, sf.why_not
        assert np.max(np.abs(sf.to_host() - _oracle_chain(chans, grid, ker))) <= 1e-12
        sf.close()
    # short grids: one or two window pairs, windows mostly zero padding
    chans = [wl.sum_channel(wf, 2, 1000 + c) for c in range(2)]
    for n, endpoint in ((12345, True), (4097, False), (300, True), (17, False)):
        grid = ('linspace', 0.0, 2 * wl.SPAN, n, endpoint)
        ker = _kernel(min(1024, 2 * n), n)
        sf = SampledFir(chans, grid, ker)
        assert sf.fused, sf.why_not
        assert sf.plan.kernel_name().startswith('fir_sampled<' if n > 4000 else 'fir_short<')    # (300 and 17 points: coarse grid)
        assert np.max(np.abs(sf.to_host() - _oracle_chain(chans, grid, ker))) <= 1e-12
        sf.close()


def test_coarse_grids_fuse_through_the_short_tier():
    # Gaussians narrower than a few window strides (256 samples) cannot be carried along a stride-256
    # chain by the two-multiplier recurrence; such plans are in the short geometry, and the transform's
    # workgroups sample their windows the short tier's way (fir_short): one kernel, same numbers
    chans = [wl.sum_channel(wf, 100, 1000)]
    grid = ('linspace', 0.0, 100 * wl.SPAN, 70001, False)
    ker = _kernel(1024)
    sf = SampledFir(chans, grid, ker)
    assert sf.fused and sf.plan.kernel_name() == 'fir_short<double,12>', (sf.why_not, sf.plan.kernel_name())
    assert np.max(np.abs(sf.to_host() - _oracle_chain(chans, grid, ker))) <= 1e-12
    sf.close()


@pytest.mark.parametrize('rate', [1e9, 2e9, 2.4e9, 5e9])
@pytest.mark.parametrize('duty30', [False, True])
def test_awg_channels_fuse_at_awg_rates(rate, duty30):
    """predistort(wav(t), ker) stays ONE kernel at 1-5 GS/s with 20 ns pulses, K = 1024"""
    n = 60_000
    chans = [wl.awg_channel(wf, c, n, rate, duty30) for c in range(3)]
    grid = wl.awg_grid(n, rate)
    ker = _kernel(1024, int(rate / 1e8))
    sf = SampledFir(chans, grid, ker)
    assert sf.fused and sf.plan.kernel_name() == 'fir_short<double,12>', (sf.why_not, sf.plan.kernel_name())
    want = _oracle_chain(chans, grid, ker)
    assert np.max(np.abs(sf.to_host() - want)) <= 1e-12
    sf.close()
    s32 = SampledFir(chans, grid, ker, np.float32)
    assert s32.fused and s32.plan.kernel_name() == 'fir_short<float,12>'
    assert np.max(np.abs(s32.to_host() - want)) <= FP32_TOL
    s32.close()


@pytest.mark.parametrize('K', [1, 2, 333, 1025, 1026, 1537])
def test_short_chain_kernel_lengths_offsets_and_ragged_ends(K):
    rate = 2e9
    for n, seed in ((50_001, 3), (7169, 4), (6144, 5), (4000, 6), (257, 7)):
        chans = [(wl.awg_channel(wf, seed + c, n, rate, c == 1) + 0.125 * c) >> (c * 0.3e-9) for c in range(2)]
        grid = wl.awg_grid(n, rate)
        ker = _kernel(min(K, 2 * n), n + K)
        sf = SampledFir(chans, grid, ker)
        assert sf.fused and sf.plan.kernel_name().startswith('fir_short<double,'), (n, sf.why_not, sf.plan.kernel_name())
        assert np.max(np.abs(sf.to_host() - _oracle_chain(chans, grid, ker))) <= 1e-11, (n, K)   # (the short tier itself: 4e-12 on these channels)
        sf.close()


def test_short_chain_flat_tops_clip_and_mixed_plans():
    rate, n = 2e9, 40_000
    grid = wl.awg_grid(n, rate)
    ker = _kernel(1024, 11)
    flat = wf.zero()
    for k in range(60):                      # flat-top pulses: erf edges are closing ops of the short tier
        flat = flat + ((wf.square(40e-9, edge=5e-9) >> (100e-9 + 300e-9 * k)) * wf.cos(2 * np.pi * 150e6, 0.1 * k))
    clipped = wf.cut(wl.awg_channel(wf, 1, n, rate), min=-0.3, max=0.45)
    sf = SampledFir([flat, clipped], grid, ker)
    assert sf.fused and sf.plan.kernel_name() == 'fir_short<double,12>', (sf.why_not, sf.plan.kernel_name())
    assert np.max(np.abs(sf.to_host() - _oracle_chain([flat, clipped], grid, ker))) <= 1e-12
    sf.close()
    # pieces the short tier cannot take (a chirp, a mollifier): the general kernel writes just those to the
    # chain's workspace in a launch of its own, the transform's workgroups copy them into their windows
    mixed = [wl.awg_channel(wf, 0, n, rate) + (wf.chirp(1e8, 2e8, 30e-9) >> 5e-6) + 0.2,
             wl.awg_channel(wf, 1, n, rate, True) + (wf.mollifier(40e-9) >> 7.03e-6) + (wf.mollifier(8e-9) >> 3e-9),
             wl.awg_channel(wf, 2, n, rate)]
    for dt, tol in ((np.float64, 1e-12), (np.float32, FP32_TOL)):
        sf = SampledFir(mixed, grid, ker, dt)
        assert sf.fused and 'fir_short<' in sf.plan.kernel_name() and sf.plan.kernel_name().startswith('wfk_sample<'), \
            (sf.why_not, sf.plan.kernel_name())
        assert np.max(np.abs(sf.to_host() - _oracle_chain(mixed, grid, ker))) <= tol
        sf.close()
    # a plan dominated by such pieces is not a short plan at all: two-kernel chain
    sf = SampledFir([(wf.square(30e-9, edge=14e-9) >> 1e-6) * wf.sinc(3e7)], grid, ker)
    assert not sf.fused and '+ FIR' in sf.plan.kernel_name()
    sf.close()


def test_vstack_offset_shift_and_float32():
    chans = [(wf.WaveVStack(wl.pulses(wf, 5, 100 + c)) + 0.25) >> 3e-9 for c in range(3)]
    grid = ('arange', -10e-9, 5 * wl.SPAN + 20e-9, 0.01e-9)
    ker = _kernel(333)
    want = _oracle_chain(chans, grid, ker)
    sf = SampledFir(chans, grid, ker)
    assert sf.fused, sf.why_not
    assert np.max(np.abs(sf.to_host() - want)) <= 1e-12
    sf.close()
    s32 = SampledFir(chans, grid, ker, np.float32)
    assert s32.fused
    got = s32.to_host()
    assert got.dtype == np.float32 and np.max(np.abs(got - want)) <= FP32_TOL
    s32.close()


def test_unfused_fallbacks_give_the_same_numbers():
    grid = ('linspace', -50e-9, 250e-9, 30011, False)
    ker = _kernel(1024)
    cases_ = {
        'erf edges (generic terms)': [wf.square(100e-9, edge=20e-9, type='erf') * wf.cos(2 * np.pi * 80e6) >> 100e-9],
        'clip': [wf.cut(wl.sum_channel(wf, 3, 7), min=-0.2, max=0.3)],
    }
    for why, chans in cases_.items():
        sf = SampledFir(chans, grid, ker)
        assert not sf.fused and sf.why_not, why
        assert np.max(np.abs(sf.to_host() - _oracle_chain(chans, grid, ker))) <= 1e-11, why
        sf.close()
    # a kernel longer than one on-chip transform: segmented FIR passes behind the sampler
    chans = [wl.sum_channel(wf, 4, 3)]
    long_ker = _kernel(2401)
    sf = SampledFir(chans, ('linspace', 0.0, 4 * wl.SPAN, 40000, False), long_ker)
    assert not sf.fused
    assert np.max(np.abs(sf.to_host() - _oracle_chain(chans, ('linspace', 0.0, 4 * wl.SPAN, 40000, False), long_ker))) <= 1e-12
    sf.close()


def test_c4_reference_fixture_1e6():
    ker = wl.c4_kernel()
    grid = ('linspace', 0.0, 100 * wl.SPAN, 10**6, False)
    chans = [wl.sum_channel(wf, 100, 1000 + c) for c in (0, 7)]
    sf = SampledFir(chans, grid, ker)
    assert sf.fused, sf.why_not
    got = sf.to_host()
    for row, c in enumerate((0, 7)):
        pick = BIG[f'c4_{c}.pick']
        assert np.max(np.abs(got[row][pick] - BIG[f'c4_{c}.fir'])) <= 1e-12
        s = BIG[f'c4_{c}.firsum']
        assert abs(got[row].sum() - s[0]) <= 1e-9 and abs(np.abs(got[row]).sum() - s[1]) <= 1e-8
    sf.close()


C4F = golden_io.npz('c4_full.npz')


def test_c4_full_size_256_channels_vs_reference_rows():
    """BASELINE configs[3] at full size: 256 channels x 1e7 points through the fused chain in ONE
    launch; rows 0 and 7 against what the real reference produced at 1e7 points (strided subset,
    every piece edge, both ends), the rest by properties."""
    import torch
    ker = wl.c4_kernel()
    chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(256)]
    sf = SampledFir(chans, wl.c2_grid(), ker)
    assert sf.fused, sf.why_not
    out = torch.empty((256, sf.n), dtype=torch.float64, device='cuda')
    sf.launch_torch(out)
    torch.cuda.synchronize()
    for c in (0, 7):
        pick = torch.as_tensor(C4F[f'{c}.pick'], device='cuda')
        got = out[c][pick].cpu().numpy()
        assert np.max(np.abs(got - C4F[f'{c}.fir'])) <= 1e-12, c
        s = C4F[f'{c}.firsum']
        assert abs(float(out[c].sum()) - s[0]) <= 1e-8 and abs(float(out[c].abs().sum()) - s[1]) <= 1e-7
        assert abs(float(out[c].abs().max()) - s[2]) <= 1e-12
    # determinism, and the same rows from a 2-channel plan (no dependence on batch position)
    out2 = torch.empty_like(out)
    sf.launch_torch(out2)
    assert torch.equal(out, out2)
    small = SampledFir([chans[7], chans[255]], wl.c2_grid(), ker)
    o2 = torch.empty((2, sf.n), dtype=torch.float64, device='cuda')
    small.launch_torch(o2)
    assert torch.equal(o2[0], out[7]) and torch.equal(o2[1], out[255])
    # DC gain: the filtered rows sum to sum(ker) * sum(samples) up to the edge effect of K taps
    assert torch.isfinite(out).all()
    sf.close()
    small.close()


def test_fused_equals_unfused_bitwise_structure_and_linearity():
    import os
    chans = [wl.sum_channel(wf, 20, 1000 + c) for c in range(4)]
    grid = ('linspace', 0.0, 20 * wl.SPAN, 250000, False)
    ker = _kernel(1024, 3)
    sf = SampledFir(chans, grid, ker)
    fused = sf.to_host()
    sf.close()
    os.environ['WFK_CHAIN_UNFUSED'] = '1'
    try:
        su = SampledFir(chans, grid, ker)
        assert not su.fused
        unfused = su.to_host()
        su.close()
    finally:
        del os.environ['WFK_CHAIN_UNFUSED']
    assert np.max(np.abs(fused - unfused)) <= 1e-13
    # linearity in the kernel: FIR(k1 + 2 k2) = FIR(k1) + 2 FIR(k2)
    k1, k2 = _kernel(700, 4), _kernel(700, 5)
    a = SampledFir(chans, grid, k1 + 2 * k2).to_host()
    b = SampledFir(chans, grid, k1).to_host() + 2 * SampledFir(chans, grid, k2).to_host()
    assert np.max(np.abs(a - b)) <= 1e-13
    # impulse kernel: the chain returns the samples themselves (delayed by K//2 - pos)
    imp = np.zeros(9)
    imp[4] = 1.0
    y = c_oracle.eval_grid(_flatten.flatten(chans), _flatten.grid_from_desc(grid))
    assert np.max(np.abs(SampledFir(chans, grid, imp).to_host() - y)) <= 1e-12


def test_erf_channels_take_the_two_kernel_path_and_exponentials_fuse():
    """erf edges are a fused op of the SAMPLER only: a chain over such channels says so and runs
    sampler + FIR as two kernels, bit-identical to calling them one after the other.  Exponential
    envelopes (coshPulse, decays) are seeded by the chain kernel too and stay fused."""
    import torch
    from waveforms_amd._sampling import BatchSampler
    from waveforms_amd.distortion import FirStage, SampledFir
    erf_ch = (wf.square(30e-9, edge=4e-9) >> 50e-9) * wf.cos(2 * np.pi * 1e8)
    exp_ch = [wf.coshPulse(40e-9, eps=2.0) >> 60e-9,
              (wf.square(50e-9) >> 60e-9) * (wf.exp(-4e7) >> 30e-9) * wf.cos(2 * np.pi * 1.5e8, 0.2),
              (wf.gaussian(30e-9) >> 50e-9) * wf.cos(2 * np.pi * 2e8, 0.4)]
    grid = ('linspace', 0.0, 120e-9, 400001, False)
    ker = np.hanning(257)
    ker /= ker.sum()

    def two_kernels(ch):
        bs = BatchSampler(ch, grid)
        raw = torch.empty((len(ch), bs.n), dtype=torch.float64, device='cuda')
        bs.launch_torch(raw)
        fir = FirStage(ker, bs.n, len(ch))
        ref = torch.empty_like(raw)
        fir.apply_torch(raw, ref)
        torch.cuda.synchronize()
        return ref

    sf = SampledFir([erf_ch] + exp_ch, grid, ker)
    assert not sf.fused and 'not fully fused' in sf.why_not
    out = torch.empty((4, sf.n), dtype=torch.float64, device='cuda')
    sf.launch_torch(out)
    torch.cuda.synchronize()
    assert torch.equal(out, two_kernels([erf_ch] + exp_ch))
    sx = SampledFir(exp_ch, grid, ker)
    assert sx.fused, sx.why_not
    out = torch.empty((3, sx.n), dtype=torch.float64, device='cuda')
    sx.launch_torch(out)
    torch.cuda.synchronize()
    assert float((out - two_kernels(exp_ch)).abs().max()) <= 1e-12


def test_short_chain_fuzz_over_random_awg_pulse_trains():
    """random pulse trains on 1-5 GS/s grids (every fusable shape, erf edges, vstacks, clips, offsets): the chain
    against the oracle whichever kernel the plan takes; most of them must take fir_short"""
    import cases
    took = {}
    for seed in range(60):
        rng = np.random.default_rng(40_000 + seed)
        ch, grid = cases.random_awg_channel(wf, rng)
        prog = _flatten.flatten([ch])
        if prog.complex_amp:
            continue                                  # (predistort of a complex signal is not a chain case)
        ker = _kernel(int(rng.choice([1, 7, 300, 1024, 1400])), seed)
        g = _flatten.grid_from_desc(grid)
        y = c_oracle.eval_grid(prog, g)[0]
        want = c_oracle.fir(y, ker)
        sf = SampledFir([ch], grid, ker)
        kn = 'fir_short' if 'fir_short<' in sf.plan.kernel_name() else sf.plan.kernel_name().split('<')[0]
        took[kn] = took.get(kn, 0) + 1
        got = sf.to_host()[0]
        pk = max(1.0, float(np.abs(y).max(initial=0.0)))
        assert np.max(np.abs(got - want), initial=0.0) <= 1e-9 * pk, (seed, kn, sf.why_not)
        sf.close()
    assert took.get('fir_short', 0) >= 40, took


@pytest.mark.parametrize('name', sorted(n for n in __import__('cases').AWG_CASES if n != 'cplx_2g'))
def test_chain_at_awg_rates_against_reference_vectors(name):
    """the chain on the AWG-rate cases against vectors of the REAL reference (awg_c4.npz: predistort of the
    sampled channel with C4's 1024-tap kernel; every 3rd sample + the transform's seams + both ends)"""
    import cases
    build, rate, n = cases.AWG_CASES[name]
    want = golden_io.npz('awg_c4.npz')[name + '.z']
    idx = cases.awg_c4_subset(n)
    for dt, tol in ((np.float64, 1e-11), (np.float32, FP32_TOL)):
        sf = SampledFir([build(wf, rate)], cases._awg_grid(n, rate), wl.c4_kernel(), dt)
        if not name.startswith('readme'):      # (the README pulses are a few long pieces: not a short plan)
            assert sf.fused and 'fir_short<' in sf.plan.kernel_name(), (sf.why_not, sf.plan.kernel_name())
        got = sf.to_host()[0][idx]
        assert np.max(np.abs(got - want)) <= tol * max(1.0, np.abs(want).max())
        sf.close()


def test_one_kernel_per_channel():
    """every AWG line has its own predistortion kernel: ker of shape (n_channels, K) through all three chain
    kernels and the FIR stage alone, against the oracle row by row"""
    from waveforms_amd.distortion import FirStage
    rng = np.random.default_rng(12)
    for chans, grid, kname in (
            ([wl.sum_channel(wf, 6, 1000 + c) for c in range(5)], ('linspace', 0.0, 6 * wl.SPAN, 50001, False), 'fir_sampled<'),
            ([wl.awg_channel(wf, c, 30000, 2e9, c == 2) for c in range(5)], wl.awg_grid(30000, 2e9), 'fir_short<')):
        for K in (1, 300, 1024, 1537):
            kers = rng.normal(size=(len(chans), K)) / K
            g = _flatten.grid_from_desc(grid)
            y = c_oracle.eval_grid(_flatten.flatten(chans), g)
            want = np.stack([c_oracle.fir(r, k) for r, k in zip(y, kers)])
            for dt, tol in ((np.float64, 1e-11), (np.float32, FP32_TOL)):
                sf = SampledFir(chans, grid, kers, dt)
                assert sf.fused and kname in sf.plan.kernel_name(), (sf.why_not, sf.plan.kernel_name())
                assert np.max(np.abs(sf.to_host() - want)) <= tol
                sf.close()
            # the stage alone, and the unfused chain
            st = FirStage(kers, g.n, len(chans))
            dx, dy = _engine.DeviceBuffer(y.nbytes), _engine.DeviceBuffer(y.nbytes)
            dx.upload(np.ascontiguousarray(y))
            st.apply(dx.ptr, g.n, dy.ptr, g.n)
            _engine.sync()
            assert np.max(np.abs(dy.download(y.shape, np.float64) - want)) <= 1e-12
            st.close(); dx.close(); dy.close()
    # a long kernel (two segments of the on-chip transform), per row
    kers = rng.normal(size=(3, 2500)) / 2500
    x = rng.normal(size=(3, 40000))
    st = FirStage(kers, 40000, 3)
    dx, dy = _engine.DeviceBuffer(x.nbytes), _engine.DeviceBuffer(x.nbytes)
    dx.upload(x)
    st.apply(dx.ptr, 40000, dy.ptr, 40000)
    _engine.sync()
    want = np.stack([c_oracle.fir(r, k) for r, k in zip(x, kers)])
    assert np.max(np.abs(dy.download(x.shape, np.float64) - want)) <= 1e-12
    st.close(); dx.close(); dy.close()
    with pytest.raises(ValueError):
        FirStage(kers, 40000, 4)


@pytest.mark.parametrize('t0', [1e-3, -2e-3])
def test_awg_trains_milliseconds_from_zero_through_the_chain(t0):
    """AWG-rate rows far from t = 0 through predistort(wav(t), ker): the sampler's plan carries corrected carriers there
    (family 6 of the short tier), which fir_short does not evaluate -- the chain must run sampler + FIR (tools/chain_soak.py
    far: a family-6 plan inside fir_short was 1e-7 off)"""
    rate, n = 2e9, 30000
    chans = [wl.awg_channel(wf, c, n, rate) >> t0 for c in range(2)]
    grid = ('arange', t0, t0 + n / rate, 1 / rate)
    ker = _kernel(1024)
    sf = SampledFir(chans, grid, ker)
    assert 'fir_short' not in sf.plan.kernel_name(), sf.plan.kernel_name()
    assert sf.plan.kernel_name().startswith('wfk_sample_short<double,false,false,16,6>'), sf.plan.kernel_name()
    want = _oracle_chain(chans, grid, ker)
    assert np.max(np.abs(sf.to_host() - want)) <= 1e-9 * max(1.0, np.abs(want).max())
    sf.close()
