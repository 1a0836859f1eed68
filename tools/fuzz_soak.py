"""Long differential fuzz (not part of the suite): random scripts x grids, HIP vs the C oracle.
usage: python tools/fuzz_soak.py [first_seed] [count] [awg | far | awgfar]   -> prints failing seeds
(awg: pulse trains on 1-5 GS/s grids, tests/cases.py random_awg_channel; also counts the tiers taken;
 far: the same scripts and grids moved 10 us .. 10 ms away from t = 0 -- the generators place everything within a few
 pulse widths of the origin, where no expansion about t = 0 can lose anything)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import cases
from cases import FP32_TOL, FP32_FAR_TOL, FP64_GRID_TOL, FP64_TLIST_FUSED_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
awg = len(sys.argv) > 3 and sys.argv[3] in ('awg', 'awgfar')
far = len(sys.argv) > 3 and sys.argv[3] in ('far', 'awgfar')
tiers = {}
bad, t0 = [], time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(10_000 + seed)
    try:
        ch, grid = cases.random_awg_channel(wf, rng) if awg else cases.random_channel(wf, rng)
        if far:
            T = float(rng.choice([1e-5, 1e-4, 1e-3, 3e-3, -2e-3, 1e-2]))
            ch = ch >> T
            grid = (grid[0], grid[1] + T, grid[2] + T) + tuple(grid[3:])
        prog = _flatten.flatten([ch])
        g = _flatten.grid_from_desc(grid)
        ora = c_oracle.eval_grid(prog, g)[0]
        pk = max(1.0, float(np.max(np.abs(ora)))) if ora.size else 1.0
        plan = _engine.Plan(prog, grid=g)
        kn = plan.kernel_name().split('<')[0]
        tiers[kn] = tiers.get(kn, 0) + 1
        cplx = bool(plan.prog.complex_amp) and awg
        if cplx:
            ora = c_oracle.eval_grid(prog, g, True)[0]
            got = plan.run_host(np.complex128)[0]
            e = float(np.max(np.abs(got - ora), initial=0.0))
            if not e <= FP64_GRID_TOL * max(1.0, float(np.max(np.abs(ora), initial=0.0))):
                bad.append((seed, 'complex', e))
                print('FAIL', bad[-1], flush=True)
            continue
        got = plan.run_host(np.float64)[0]
        e64 = float(np.max(np.abs(got - ora), initial=0.0))
        if g.n >= 4:
            # a random time slice of the grid is a grid of its own (wfk_grid.i0 != 0: what time sharding launches)
            lo = int(rng.integers(0, g.n - 2)); hi = int(rng.integers(lo + 1, g.n + 1))
            ps = _engine.Plan(prog, grid=_flatten.grid_slice(g, lo, hi))
            es = float(np.max(np.abs(ps.run_host(np.float64)[0] - ora[lo:hi]), initial=0.0))
            kn_s = ps.kernel_name().split('<')[0]
            ps.close()
            if not es <= FP64_GRID_TOL * pk:
                bad.append((seed, 'slice', lo, hi, es / pk, kn_s))
                print('FAIL', bad[-1], flush=True)
        got32 = plan.run_host(np.float32)[0].astype(np.float64)
        e32 = float(np.max(np.abs(got32 - ora), initial=0.0))
        tl = _engine.Plan(prog, t=c_oracle.grid_values(g)).run_host(np.float64)[0]
        etl = float(np.max(np.abs(tl - ora), initial=0.0))
        if not (e64 <= FP64_GRID_TOL * pk and e32 <= (FP32_FAR_TOL if far else FP32_TOL) * pk and etl <= FP64_TLIST_FUSED_TOL * pk) or \
                (np.all(np.isfinite(ora)) and not np.all(np.isfinite(got))):
            bad.append((seed, e64 / pk, e32 / pk, etl / pk))
            print('FAIL', bad[-1], flush=True)
    except NotImplementedError as e:
        pass
    except Exception as e:
        bad.append((seed, repr(e)))
        print('ERROR', bad[-1], flush=True)
    if (seed - first) % 250 == 249:
        print(f'{seed - first + 1} scripts, {len(bad)} failures, {time.time() - t0:.0f} s', flush=True)
print('done', count, 'scripts;', len(bad), 'failures', bad[:10], 'tiers', tiers)
