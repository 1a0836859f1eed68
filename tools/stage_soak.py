"""Long differential fuzz of the filter stages (not part of the suite): FIR against the time-domain
definition (np.convolve) and IIR against scipy (sosfilt / lfilter with initial state), random
sizes, tap counts (incl. multi-segment kernels and the rocFFT fallback) and filter shapes.
usage: python tools/stage_soak.py [count]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from scipy import signal
from cases import FP64_FIR_TOL, FP64_IIR_RANDOM_TOL
from waveforms_amd import distortion

count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(4242)
bad, t0 = [], time.time()
for it in range(count):
    n = int(rng.integers(1, 40000))
    K = int(rng.choice([1, 2, 3, 17, 255, 1024, 1537, 1538, 2401, 3075, 4611, 6148, 6149,
                        int(rng.integers(1, 7000))]))
    sig, ker = rng.normal(size=n), rng.normal(size=K)
    want = np.convolve(np.concatenate([np.zeros(K), sig, np.zeros(K)]), ker)[K + K // 2:K + K // 2 + n]
    got = distortion.predistort(sig, ker=ker)
    e = float(np.max(np.abs(got - want))) / max(1.0, float(np.abs(want).max()))
    if not e <= FP64_FIR_TOL:
        bad.append(('fir', n, K, e)); print('FAIL', bad[-1], flush=True)
    # IIR: random stable cascade
    nsec = int(rng.integers(1, 9))
    if rng.random() < 0.5:
        z = rng.uniform(-0.95, 0.95, size=2 * nsec) * np.exp(1j * rng.uniform(0, 0.5, size=2 * nsec))
        r = rng.uniform(0.2, 0.9995, size=nsec); th = rng.uniform(0, 3.1, size=nsec)
        p = np.concatenate([r * np.exp(1j * th), r * np.exp(-1j * th)])
        zz = rng.uniform(-1, 1, size=nsec); zz = np.concatenate([zz, zz[::-1] * 0.5])
        sos = signal.zpk2sos(zz, p, rng.uniform(0.1, 2))
        secs = [(row[:3], row[3:]) for row in sos]
        zi0 = rng.normal(size=(len(sos), 2)) * 0.1
        want, zf = signal.sosfilt(sos, sig, zi=zi0)
        got, gzf = distortion.iir_host(sig, secs, zi=zi0.reshape(-1))
        gzf = np.asarray(gzf).reshape(-1, 2)
    else:
        order = int(rng.integers(1, 10))
        p = rng.uniform(0.1, 0.999, size=order) * rng.choice([-1, 1], size=order)
        a = np.poly(p); b = rng.normal(size=order + 1)
        zi0 = rng.normal(size=order) * 0.1
        want, zf = signal.lfilter(b, a, sig, zi=zi0)
        got, gzf = distortion.iir_host(sig, [(b, a)], zi=zi0)
    sc = max(1.0, float(np.abs(want).max()))
    e = float(np.max(np.abs(got - want))) / sc
    ez = float(np.max(np.abs(np.asarray(gzf).reshape(-1) - np.asarray(zf).reshape(-1)))) / sc
    if not (e <= FP64_IIR_RANDOM_TOL and ez <= FP64_IIR_RANDOM_TOL):
        bad.append(('iir', n, nsec, e, ez)); print('FAIL', bad[-1], flush=True)
    if it % 100 == 99:
        print(f'{it + 1} rounds, {len(bad)} failures, {time.time() - t0:.0f} s', flush=True)
print('done', count, 'rounds;', len(bad), 'failures', bad[:10], '(bounds: FIR %g, IIR %g of peak)' % (FP64_FIR_TOL, FP64_IIR_RANDOM_TOL))
