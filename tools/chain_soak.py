"""Long differential fuzz of the sampler -> FIR chain (not part of the suite): random pulse trains on 1-5 GS/s
grids (tests/cases.py random_awg_channel) and random BASELINE-style channels through SampledFir, fp64 and fp32,
against the C oracle (sampler) + direct convolution.   usage: python tools/chain_soak.py [first_seed] [count] [far]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import cases
from cases import FP32_TOL, FP64_GRID_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _flatten
from waveforms_amd.distortion import SampledFir

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
FAR = len(sys.argv) > 3 and sys.argv[3] == 'far'
took, bad, t0 = {}, [], time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(70_000 + seed)
    try:
        ch, grid = cases.random_awg_channel(wf, rng) if seed % 4 else cases.random_channel(wf, rng)
        if FAR:        # the same trains 0.1 .. 3 ms from t = 0 (corrected carriers: family 6 plans must stay out of fir_short)
            T_ = float(rng.choice([1e-4, 1e-3, 3e-3, -2e-3]))
            ch = ch >> T_
            grid = (grid[0], grid[1] + T_, grid[2] + T_) + tuple(grid[3:])
        nch = int(rng.integers(1, 4))
        chans = [ch] + [(ch * float(rng.uniform(0.2, 1.5)) + float(rng.uniform(-0.3, 0.3))) for _ in range(nch - 1)]
        prog = _flatten.flatten(chans)
        if prog.complex_amp or prog.host_complex:
            continue
        g = _flatten.grid_from_desc(grid)
        if g.n < 2:
            continue
        K = int(rng.choice([1, 2, 9, 128, 1000, 1024, 1025, 1500, 1537, 3000]))
        ker = rng.normal(size=K); ker /= np.abs(ker).sum()
        y = c_oracle.eval_grid(prog, g)
        want = np.stack([c_oracle.fir(r, ker) for r in y])
        pk = max(1.0, float(np.abs(y).max(initial=0.0)))
        for dt, tol in ((np.float64, FP64_GRID_TOL), (np.float32, FP32_TOL)):      # (fp32: fuzz_soak's bound for the sampler alone)
            sf = SampledFir(chans, grid, ker, dt)
            kn = sf.plan.kernel_name()
            kn = 'hybrid fir_short' if (' + fir_short' in kn) else kn.split('<')[0].split(' ')[0] + (' + FIR' if '+ FIR' in kn else '')
            took[kn] = took.get(kn, 0) + 1
            e = float(np.max(np.abs(sf.to_host() - want), initial=0.0))
            sf.close()
            if not e <= tol * pk:
                bad.append((seed, dt.__name__, kn, e))
                print('FAIL', bad[-1], flush=True)
    except Exception as ex:     # noqa: BLE001
        bad.append((seed, repr(ex)[:200]))
        print('FAIL', bad[-1], flush=True)
    if (seed - first) % 200 == 199:
        print('...', seed - first + 1, 'rounds,', len(bad), 'failures, %.0f s' % (time.time() - t0), flush=True)
print('done %d rounds; %d failures %s kernels %s' % (count, len(bad), bad[:10], took))
