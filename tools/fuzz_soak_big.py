"""Like fuzz_soak.py with grids of up to 3e6 points (thousands of wave tiles per piece: chunking,
state carry across tiles, periodic reseeding) and small batches."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import cases
from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
OFFSET = len(sys.argv) > 3 and sys.argv[3] == 'offset'
bad, t0 = [], time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(77_000 + seed)
    chans = []
    ch, grid = cases.random_channel(wf, rng)
    nch = int(rng.integers(1, 4))
    scale = (grid[2] - grid[1])
    chans = [ch] + [cases.random_channel(wf, rng)[0] for _ in range(nch - 1)]
    npts = int(rng.integers(100000, 3000000))
    if OFFSET:
        # everything (pulses and grid) far from t = 0: offsets of 1e2..1e7 grid spans
        off = (grid[2] - grid[1]) * 10.0**rng.uniform(2, 7) * (1 if rng.random() < 0.5 else -1)
        chans = [c >> off for c in chans]
        grid = (grid[0], grid[1] + off, grid[2] + off) + tuple(grid[3:])
        npts = int(rng.integers(1000, 300000))
    grid = ('linspace', grid[1], grid[2], npts, bool(rng.random() < 0.5))
    try:
        prog = _flatten.flatten(chans)
        g = _flatten.grid_from_desc(grid)
        ora = c_oracle.eval_grid(prog, g)
        pk = max(1.0, float(np.max(np.abs(ora))))
        plan = _engine.Plan(prog, grid=g)
        e64 = float(np.max(np.abs(plan.run_host(np.float64) - ora)))
        e32 = float(np.max(np.abs(plan.run_host(np.float32).astype(np.float64) - ora)))
        if not (e64 <= 1e-9 * pk and e32 <= FP32_TOL * pk):
            bad.append((seed, npts, e64 / pk, e32 / pk, plan.info.n_fused, plan.info.n_generic))
            print('FAIL', bad[-1], flush=True)
        plan.close()
    except NotImplementedError:
        pass
    except Exception as e:
        bad.append((seed, repr(e))); print('ERROR', bad[-1], flush=True)
    if (seed - first) % 50 == 49:
        print(f'{seed - first + 1} scripts, {len(bad)} failures, {time.time() - t0:.0f} s', flush=True)
print('done', count, 'scripts;', len(bad), 'failures', bad[:10])
