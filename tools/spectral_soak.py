"""reflection / correct_reflection / shift on random lengths (even, odd, prime, large) against the
reference formulas in NumPy (distortion.py:12-39, 208-223).  GPU box.  usage: spectral_soak.py [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from waveforms_amd import distortion as d

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(123)
bad = []
for it in range(count):
    n = int(rng.choice([1, 2, 3, 5, 7, 64, 97, 1000, 1009, 4096, 10007, 65536, 65537, 100003,
                        int(rng.integers(1, 300000))]))
    fs = 10.0**rng.uniform(8, 9.5)
    A, tau = rng.uniform(-0.5, 0.5), rng.uniform(0.3, 50) / fs
    sig = np.cumsum(rng.normal(size=n)) / 30 + rng.normal(size=n) * 0.01
    try:
        freq = np.fft.fftfreq(n, 1 / fs)
        H = d.reflection_filter(freq, A, tau)
        want = np.fft.ifft(np.fft.fft(sig) * H).real
        got = d.reflection(sig, A, tau, fs)
        sc = max(1.0, float(np.abs(want).max()))
        e1 = float(np.max(np.abs(got - want))) / sc
        want2 = np.fft.ifft(np.fft.fft(sig) / H).real
        e2 = float(np.max(np.abs(d.correct_reflection(sig, A, tau, fs) - want2))) / max(1.0, float(np.abs(want2).max()))
        delay = rng.uniform(-5, 5) / fs
        dt = 1 / fs
        pts = int(delay // dt); delta = delay / dt - pts
        ref = sig.copy()
        if delta > 0:
            ker = np.array([0, 1 - delta, delta])
            ref = np.convolve(np.concatenate([np.zeros(3), sig, np.zeros(3)]), ker)[3 + 1:3 + 1 + n]
        if pts != 0:
            r2 = np.zeros_like(ref)
            if abs(pts) < n:
                if pts < 0: r2[:pts] = ref[-pts:]
                else: r2[pts:] = ref[:-pts]
            ref = r2
        e3 = float(np.max(np.abs(d.shift(sig, delay, dt) - ref), initial=0.0)) / sc
        if not (e1 <= 1e-10 and e2 <= 1e-9 and e3 <= 1e-12):
            bad.append((it, n, e1, e2, e3)); print('FAIL', bad[-1], flush=True)
    except Exception as ex:
        bad.append((it, n, repr(ex))); print('ERROR', bad[-1], flush=True)
print('done', count, 'rounds;', len(bad), 'failures', bad[:8])
