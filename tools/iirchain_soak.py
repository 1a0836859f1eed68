"""Long differential fuzz of the sampler -> IIR (-> FIR) chain (not part of the suite): random BASELINE-style pulse
trains on fine grids (whatever the chain does with them: sampler inside the IIR scan, or sampler then filter) and random
trains at AWG rates, through SampledIir with random cascades (Butterworth SOS, exponential-correction sections, single
sections of order 3 / 4), random initial levels and states, fp64 and fp32, against the C oracle (sampler) + SciPy.
usage: python tools/iirchain_soak.py [first_seed] [count] [far]   (far: the trains 0.1 .. 3 ms from t = 0; the sampler's own bound then applies)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from scipy.signal import butter, lfilter
import cases
from cases import FP32_TOL, FP64_IIR_TOL, FP64_IIR_ORDER34_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _flatten, workloads as wl
from waveforms_amd.distortion import SampledIir, exp_decay_filter

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
FAR = len(sys.argv) > 3 and sys.argv[3] == 'far'
TOL = FP64_IIR_TOL   # the IIR stages' fp64 bound (tests/cases.py); contract 1e-9
took, bad, t0, worst = {}, [], time.time(), 0.0


def cascade(secs, x, initial, zi):
    y = np.asarray(x, dtype=np.float64) - initial
    off, zf = 0, []
    for b, a in secs:
        m = max(len(b), len(a)) - 1
        y, z1 = lfilter(b, a, y, zi=zi[off:off + m])
        zf.append(z1)
        off += m
    return y + initial, np.concatenate(zf)


for seed in range(first, first + count):
    rng = np.random.default_rng(90_000 + seed)
    try:
        kind = seed % 4
        if kind == 0:
            ch, grid = cases.random_awg_channel(wf, rng)
        elif kind == 1:
            ch, grid = cases.random_channel(wf, rng)
        else:      # the shape the fused scan is for: pulse trains on fine grids, 8192-sample chunks crossing piece edges
            nseg = int(rng.integers(3, 40))
            n = int(rng.integers(33000, 400000))
            ch = wl.sum_channel(wf, nseg, int(rng.integers(1 << 30)), spacing=wl.SPAN * float(rng.choice([1.0, 1.0, 1.5])))
            grid = ('linspace', 0.0, nseg * wl.SPAN * 1.5, n, bool(rng.integers(2)))
        if FAR:
            T_ = float(rng.choice([1e-4, 1e-3, 3e-3, -2e-3]))
            ch = ch >> T_
            grid = (grid[0], grid[1] + T_, grid[2] + T_) + tuple(grid[3:])
        nch = int(rng.integers(1, 4))
        chans = [ch] + [ch * float(rng.uniform(0.2, 1.5)) for _ in range(nch - 1)]
        prog = _flatten.flatten(chans)
        if prog.complex_amp or prog.host_complex:
            continue
        g = _flatten.grid_from_desc(grid)
        if g.n < 2:
            continue
        shape = int(rng.integers(5))
        if shape == 0:
            secs = [(r[:3], r[3:]) for r in butter(int(rng.choice([2, 4])), float(rng.uniform(0.01, 0.3)), output='sos')]
        elif shape == 1:
            secs = [exp_decay_filter(float(rng.uniform(-0.05, 0.05)), float(10 ** rng.uniform(-8.5, -5.5)), 1e9) for _ in range(int(rng.integers(1, 5)))]
        elif shape == 2:
            secs = [butter(int(rng.choice([3, 4])), float(rng.uniform(0.02, 0.2)))]
        elif shape == 3:
            secs = [(r[:3], r[3:]) for r in butter(int(rng.choice([6, 8])), float(rng.uniform(0.05, 0.3)), output='sos')]
        else:
            secs = [exp_decay_filter(float(rng.uniform(-0.03, 0.03)), float(10 ** rng.uniform(-8, -6)), 1e9) for _ in range(int(rng.integers(5, 9)))]
        D = sum(max(len(b), len(a)) - 1 for b, a in secs)
        initial = float(rng.choice([0.0, rng.uniform(-0.5, 0.5)]))
        zi = rng.normal(size=D) * float(rng.choice([0.0, 0.1]))
        x = c_oracle.eval_grid(prog, g)
        want, wzf = zip(*[cascade(secs, r, initial, zi) for r in x])
        want, wzf = np.stack(want), np.stack(wzf)
        pk = max(1.0, float(np.abs(want).max(initial=0.0)))
        tol64 = FP64_IIR_ORDER34_TOL if any(max(len(b), len(a)) - 1 >= 3 for b, a in secs) else TOL
        if FAR: tol64 = max(tol64, cases.FP64_GRID_TOL)      # (milliseconds out the sampler itself uses its budget)
        for dt, tol in ((np.float64, tol64), (np.float32, FP32_TOL)):
            si = SampledIir(chans, grid, secs, None, dt)
            kn = si.plan.kernel_name().split('<')[0] + (' (+passes)' if 'IIR passes' in si.plan.kernel_name() else '')
            took[kn] = took.get(kn, 0) + 1
            got, zf = si.to_host(initial=initial, zi=zi, return_zf=True)
            si.close()
            e = float(np.max(np.abs(got - want), initial=0.0)) / pk
            ez = float(np.max(np.abs(zf - wzf), initial=0.0)) / max(1.0, float(np.abs(wzf).max(initial=0.0)))
            if dt is np.float64:
                worst = max(worst, e)
            if not e <= tol or (dt is np.float64 and not ez <= 1e-8):
                bad.append((seed, dt.__name__, kn, e, ez))
                print('FAIL', bad[-1], flush=True)
    except Exception as ex:     # noqa: BLE001
        bad.append((seed, repr(ex)[:200]))
        print('FAIL', bad[-1], flush=True)
    if (seed - first) % 100 == 99:
        print('...', seed - first + 1, 'rounds,', len(bad), 'failures, worst fp64 %.3g of peak, %.0f s' % (worst, time.time() - t0), flush=True)
print('done %d rounds; %d failures %s; worst fp64 %.3g of peak (bound %g; single sections of order 3 / 4: %g); kernels %s' % (count, len(bad), bad[:10], worst, TOL, FP64_IIR_ORDER34_TOL, took))
