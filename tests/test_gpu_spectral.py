"""FFT-domain operations of distortion.py on the device, against
golden outputs of the real reference (tests/golden/spectral.npz).  The reference has no
tests for these functions: parity is pinned by reference-generated vectors only."""
import numpy as np
import pytest

import cases
import golden_io
from waveforms_amd import distortion
import waveforms_amd as wf

pytestmark = pytest.mark.gpu
SPEC = golden_io.npz('spectral.npz')


@pytest.mark.parametrize('i', range(len(cases.spectral_cases())))
def test_reflection_and_shift(i):
    n, A, tau, fs = cases.spectral_cases()[i]
    sig = cases.spectral_input(i)
    scale = max(1.0, np.abs(sig).max())
    got = distortion.reflection(sig, A, tau, fs)
    assert np.max(np.abs(got - SPEC[f'{i}.refl'])) <= 1e-11 * scale
    # round trip: exact except for the Nyquist bin of an even-length signal, whose
    # imaginary part `.real` discards -- in the reference just the same (1.6e-7 here)
    back = distortion.correct_reflection(got, A, tau, fs)
    fr = np.fft.fftfreq(n, 1 / fs)
    Hf = distortion.reflection_filter(fr, A, tau)
    ref_back = np.fft.ifft(np.fft.fft(SPEC[f'{i}.refl']) / Hf).real
    assert np.max(np.abs(back - ref_back)) <= 1e-11 * scale
    assert np.max(np.abs(back - sig)) <= 1e-6 * scale
    corr = distortion.correct_reflection(sig, A, tau, fs)
    assert np.max(np.abs(corr - SPEC[f'{i}.corr'])) <= 1e-11 * scale
    sh = distortion.shift(sig, 3.3 / fs * (1 if i % 2 else -1), 1 / fs)
    assert np.max(np.abs(sh - SPEC[f'{i}.shift'])) <= 1e-12 * scale


def test_kernel_design_and_symbolic_branch():
    zk = distortion.zDistortKernel(1e-9, [(50e-9, 0.02), (400e-9, -0.01)])
    assert np.allclose(zk, SPEC['zker'], rtol=1e-12, atol=1e-15)
    w = wf.gaussian(10e-9) >> 20e-9
    c = distortion.correct_reflection(w, 0.1, 5e-9)
    assert isinstance(c, wf.Waveform)
    assert c == 1 / 0.9 * w - 0.1 / 0.9 * (w >> 5e-9)
    with pytest.raises(ValueError):
        distortion.correct_reflection(np.zeros(8), 0.1, 1e-9)


def test_shift_keeps_the_callers_dtype():
    # upstream is np.convolve(signal, ker, 'same') + an integer roll with np.zeros_like(signal)
    # (distortion.py:22-38): complex stays complex, integers stay integers when only whole samples move
    rng = np.random.default_rng(5)
    z = rng.normal(size=3001) + 1j * rng.normal(size=3001)
    for delay in (2.3, -4.7, 0.4):
        pts, delta = int(delay // 1.0), delay - int(delay // 1.0)
        want = np.convolve(z, np.array([0, 1 - delta, delta]), mode='same')
        if pts:
            r = np.zeros_like(want)
            if pts < 0:
                r[:pts] = want[-pts:]
            else:
                r[pts:] = want[:-pts]
            want = r
        got = distortion.shift(z, delay, 1.0)
        assert got.dtype == np.complex128 and np.max(np.abs(got - want)) <= 1e-13
    k = np.arange(10)
    got = distortion.shift(k, 3.0, 1.0)
    assert got.dtype == k.dtype and list(got) == [0, 0, 0, 0, 1, 2, 3, 4, 5, 6]
