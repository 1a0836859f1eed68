"""FIR stage parity: HIP/rocFFT overlap-save (through the C-ABI) against
  * golden outputs of the real reference's predistort(sig, ker=...) (tests/golden/fir.npz),
  * the oracles (FFT restatement and time-domain restatement),
  * properties at scale: linearity and the impulse response.
fp64: |err| <= 1e-9 * peak (we hold ~1e-13); fp32: 1e-3 relative (we hold 1e-5)."""
import numpy as np
import pytest

import golden_io
from oracle import c_oracle, np_oracle
from waveforms_amd import _engine, distortion

pytestmark = pytest.mark.gpu
FIR = golden_io.npz('fir.npz')
BIG = golden_io.npz('big.npz')


@pytest.fixture(params=['fused', 'rocfft'])
def fir_path(request, monkeypatch):
    """Both FIR implementations behind wfk_fir_*: the fused LDS-FFT kernel (short
    kernels) and the rocFFT overlap-save pipeline (any length)."""
    if request.param == 'rocfft':
        monkeypatch.setenv('WFK_FIR_ROCFFT', '1')
    return request.param


@pytest.mark.parametrize('i', range(10))
def test_fir_matches_reference_vectors(i, fir_path):
    sig, ker, want = FIR[f'{i}.sig'], FIR[f'{i}.ker'], FIR[f'{i}.out']
    got = distortion.predistort(sig, ker=ker)
    assert got.shape == want.shape and got.dtype == np.float64
    scale = max(1.0, np.abs(want).max())
    assert np.max(np.abs(got - want)) <= 1e-12 * scale


def test_fir_edge_semantics(fir_path):
    rng = np.random.default_rng(3)
    for n, k in [(1, 1), (2, 5), (17, 2), (7168, 1024), (7169, 1024), (7170, 1024),
                 (20000, 1), (50, 1024), (9000, 1537), (9000, 1538), (5000, 3000),
                 # long kernels: 2, 3, 4 accumulated segments of the fused kernel; 5 -> rocFFT
                 (30000, 2401), (30001, 3075), (12345, 4611), (40000, 6148), (9000, 6149),
                 (300, 2401), (1, 5000)]:
        sig, ker = rng.normal(size=n), rng.normal(size=k)
        want = np_oracle.predistort_fir(sig, ker)
        got = distortion.predistort(sig, ker=ker)
        assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.abs(want).max()), (n, k)
        if n * k < 2e7:
            td = c_oracle.fir(sig, ker)
            assert np.max(np.abs(got - td)) <= 1e-12 * max(1.0, np.abs(td).max())
    # ker=None passes the signal through
    s = rng.normal(size=10)
    assert np.array_equal(distortion.predistort(s), s)


def test_fir_batch_fp32_and_properties(fir_path):
    import torch
    rng = np.random.default_rng(5)
    n, batch, K = 300_000, 5, 1024
    ker = rng.normal(size=K)
    ker /= np.abs(ker).sum()
    x = rng.normal(size=(batch, n))
    want = np.stack([np_oracle.predistort_fir(r, ker) for r in x])
    st = distortion.FirStage(ker, n, batch, np.float64)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    st.apply_torch(xd, yd)
    torch.cuda.synchronize()
    assert np.max(np.abs(yd.cpu().numpy() - want)) <= 1e-13
    # linearity: F(a x1 + b x2) == a F(x1) + b F(x2)
    zd = torch.empty_like(xd)
    st.apply_torch(2.0 * xd + 3.0 * torch.flip(xd, [0]), zd)
    torch.cuda.synchronize()
    lin = 2.0 * yd + 3.0 * torch.flip(yd, [0])
    assert float((zd - lin).abs().max()) <= 1e-12
    # impulse at position p reproduces the kernel, centred at K//2
    imp = torch.zeros_like(xd)
    p = 123_456
    imp[:, p] = 1.0
    st.apply_torch(imp, zd)
    torch.cuda.synchronize()
    got = zd[0, p - K // 2: p - K // 2 + K].cpu().numpy()
    assert np.max(np.abs(got - ker)) <= 1e-15
    # fp32
    st32 = distortion.FirStage(ker, n, batch, np.float32)
    x32 = xd.float()
    y32 = torch.empty_like(x32)
    st32.apply_torch(x32, y32)
    torch.cuda.synchronize()
    assert np.max(np.abs(y32.cpu().numpy().astype(np.float64) - want)) <= 1e-5


def test_c4_sampler_then_fir_against_reference():
    """C4 at 1e6 points: sample a 100-pulse channel on the device, FIR it on the
    device, compare with reference subsets (oracle/make_golden.py)."""
    import torch
    import waveforms_amd as wf
    from waveforms_amd import workloads as wl
    from waveforms_amd._sampling import BatchSampler
    chans = [wl.sum_channel(wf, 100, 1000 + c) for c in (0, 7)]
    grid = ('linspace', 0.0, 100 * wl.SPAN, 10**6, False)
    bs = BatchSampler(chans, grid)
    sig = torch.empty((2, bs.n), dtype=torch.float64, device='cuda')
    bs.launch_torch(sig)
    st = distortion.FirStage(wl.c4_kernel(), bs.n, 2, np.float64)
    out = torch.empty_like(sig)
    st.apply_torch(sig, out)
    torch.cuda.synchronize()
    sig_h, out_h = sig.cpu().numpy(), out.cpu().numpy()
    for row, c in enumerate((0, 7)):
        pick = BIG[f'c4_{c}.pick']
        assert np.max(np.abs(sig_h[row][pick] - BIG[f'c4_{c}.y'])) <= 1e-9
        assert np.max(np.abs(out_h[row][pick] - BIG[f'c4_{c}.fir'])) <= 1e-9
        assert abs(out_h[row].sum() - BIG[f'c4_{c}.firsum'][0]) <= 1e-6


def test_long_kernel_segments_batch_and_fp32():
    """K = 2401 (the Z-line kernel of a 400 ns time constant at 2 GS/s) takes two accumulated
    passes of the fused kernel; rows, strides and fp32 as for the short kernels."""
    import torch
    rng = np.random.default_rng(11)
    n, batch, K = 200_001, 3, 2401
    ker = rng.normal(size=K)
    ker /= np.abs(ker).sum()
    x = rng.normal(size=(batch, n))
    want = np.stack([np_oracle.predistort_fir(r, ker) for r in x])
    for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        st = distortion.FirStage(ker, n, batch, dt)
        tdt = torch.float64 if dt == np.float64 else torch.float32
        xd = torch.from_numpy(x).to('cuda', tdt)
        yd = torch.full((batch, n + 13), 7.0, dtype=tdt, device='cuda')   # padded rows
        st.apply_torch(xd, yd)
        torch.cuda.synchronize()
        got = yd.cpu().numpy().astype(np.float64)
        assert np.max(np.abs(got[:, :n] - want)) <= tol
        assert np.all(got[:, n:] == 7.0)                                  # padding untouched
        st.close()
