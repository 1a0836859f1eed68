"""BASELINE.json configurations at their full sizes, against reference-generated
subsets (tests/golden/big.npz: strided picks + piece-boundary neighbourhoods, sums,
norms, integer piece indices) and size-independent properties."""
import numpy as np
import pytest

import golden_io
from cases import FP32_TOL
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten, workloads as wl
from waveforms_amd._sampling import BatchSampler

pytestmark = pytest.mark.gpu
BIG = golden_io.npz('big.npz')


def check_subset(name, y, tol):
    pick, want = BIG[name + '.pick'], BIG[name + '.y']
    assert np.max(np.abs(y[pick] - want)) <= tol
    s = BIG[name + '.sum']
    assert abs(y.sum() - s[0]) <= tol * len(y) ** 0.5 * 10
    assert abs(np.abs(y).sum() - s[1]) <= tol * len(y)
    assert abs(np.sqrt((y * y).sum()) - s[2]) <= tol * 1e3
    assert abs(np.abs(y).max() - s[3]) <= tol


@pytest.mark.parametrize('duty30', [False, True])
def test_c2_1e7_fp64(duty30):
    """C2: 1 channel x 100 gaussian+DRAG pulses x 1e7 points, fp64 (< 1e-9 abs)."""
    name = 'c2_duty30' if duty30 else 'c2'
    w = wl.c2_channel(wf, duty30)
    bs = BatchSampler([w], wl.c2_grid(duty30=duty30))
    assert bs.plan.info.n_generic == 0
    assert np.array_equal(bs.plan.member_index(0), BIG[name + '.idx'])   # integer parity
    y = bs.to_host(np.float64)[0]
    check_subset(name, y, 1e-9)
    if duty30:   # 70 % of the samples lie in zero pieces: exactly 0.0
        idx = BIG[name + '.idx']
        assert not y[:idx[0]].any() and not y[idx[1]:idx[2]].any()
    # the drop-in call (tlist mode, device libm) agrees on a window
    t = wl.make_grid(wl.c2_grid(duty30=duty30))[4_000_000:4_200_000]
    assert np.max(np.abs(w(t) - y[4_000_000:4_200_000])) <= 1e-9


def test_c3_256ch_1e6_fp32():
    """C3: 256 WaveVStack channels x 20 pulses x 1e6 points, fp32 out (1e-3 rel)."""
    import torch
    chans = wl.c3_channels(wf, 256)
    bs = BatchSampler(chans, wl.c3_grid())
    out = torch.empty((256, bs.n), dtype=torch.float32, device='cuda')
    bs.launch_torch(out)
    torch.cuda.synchronize()
    for c in (0, 1, 255):
        y = out[c].cpu().numpy().astype(np.float64)
        pick, want = BIG[f'c3_{c}.pick'], BIG[f'c3_{c}.y']
        assert np.max(np.abs(y[pick] - want)) <= 1e-3 * np.abs(want).max()
        assert np.max(np.abs(y[pick] - want)) <= FP32_TOL          # what we actually hold
    # fp64 launch of the same plan: linearity  2*x == x + x  via accumulate
    o64 = torch.zeros((256, bs.n), dtype=torch.float64, device='cuda')
    bs.launch_torch(o64)
    bs.launch_torch(o64, accumulate=True)
    ref = torch.empty_like(o64)
    bs.launch_torch(ref)
    torch.cuda.synchronize()
    assert torch.equal(o64, ref + ref)
    assert float((ref.float() - out).abs().max()) <= FP32_TOL


def test_sampler_256ch_1e7_properties():
    """Headline workload at full size: every channel equals its own single-channel
    plan, on probes; checksum of checksums is reproducible run to run."""
    import torch
    nch = 256
    chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
    bs = BatchSampler(chans, wl.c2_grid())
    out = torch.empty((nch, bs.n), dtype=torch.float64, device='cuda')
    bs.launch_torch(out)
    torch.cuda.synchronize()
    s1 = out.sum(dim=1)
    bs.launch_torch(out)
    torch.cuda.synchronize()
    assert torch.equal(s1, out.sum(dim=1))                    # deterministic
    assert bool(torch.isfinite(out).all())
    assert float(out.abs().max()) <= 1.2
    for c in (0, 7):                                          # C4 sampler rows, 1e7 grid
        # (a 1-channel plan chunks the time axis differently, so its exact-reseed
        # points differ: equal to rounding, not bitwise)
        one = BatchSampler([chans[c]], wl.c2_grid()).to_host(np.float64)[0]
        assert np.max(np.abs(one - out[c].cpu().numpy())) <= 1e-12
    probe = torch.tensor([0, 1234567, 9_999_999], device='cuda')
    from oracle import np_oracle
    t = wl.make_grid(wl.c2_grid())[probe.cpu().numpy()]
    for c in (3, 200):
        want = np_oracle.call(chans[c], np.sort(t))
        assert np.max(np.abs(out[c][probe].cpu().numpy() - want)) <= 1e-9
