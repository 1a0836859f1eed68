import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import cases
import waveforms_amd as wf
from oracle import c_oracle, np_oracle
from waveforms_amd import _engine, _flatten, workloads as wl

for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(77_000 + seed)
    ch, grid = cases.random_channel(wf, rng)
    nch = int(rng.integers(1, 4))
    chans = [ch] + [cases.random_channel(wf, rng)[0] for _ in range(nch - 1)]
    npts = int(rng.integers(100000, 3000000))
    if os.environ.get('OFFSET'):
        off = (grid[2] - grid[1]) * 10.0**rng.uniform(2, 7) * (1 if rng.random() < 0.5 else -1)
        chans = [c >> off for c in chans]
        grid = (grid[0], grid[1] + off, grid[2] + off) + tuple(grid[3:])
        npts = int(rng.integers(1000, 300000))
    grid = ('linspace', grid[1], grid[2], npts, bool(rng.random() < 0.5))
    g = _flatten.grid_from_desc(grid)
    t = wl.make_grid(grid)
    print('seed', seed, 'grid', grid, 'nch', nch)
    for c, w in enumerate(chans):
        prog = _flatten.flatten([w])
        ora = c_oracle.eval_grid(prog, g)[0]
        npo = np.real(np_oracle.call(w, t))
        pk = max(1.0, float(np.abs(ora).max()))
        plan = _engine.Plan(prog, grid=g)
        gpu = plan.run_host(np.float64)[0]
        g32 = plan.run_host(np.float32)[0].astype(np.float64)
        tl = _engine.Plan(prog, t=t).run_host(np.float64)[0]
        i = int(np.argmax(np.abs(gpu - ora)))
        print(' ch', c, 'peak %.3g' % pk, 'fused/generic/fast/direct', plan.info.n_fused, plan.info.n_generic, plan.info.n_fast, plan.info.n_direct,
              '| gpu-C %.2e  gpu-np %.2e  C-np %.2e  tlist-C %.2e  f32-C %.2e' % (
                  np.abs(gpu - ora).max() / pk, np.abs(gpu - npo).max() / pk, np.abs(ora - npo).max() / pk,
                  np.abs(tl - ora).max() / pk, np.abs(g32 - ora).max() / pk),
              '| worst at i=%d t=%.6g' % (i, t[i]))
        if np.abs(gpu - ora).max() / pk > 1e-9 or np.abs(g32 - ora).max() / pk > 5e-5:
            print('   script:', w.tolist()[:60])
        if os.environ.get('WHERE') and np.abs(gpu - ora).max() / pk > 1e-9:
            badi = np.nonzero(np.abs(gpu - ora) > 1e-9 * pk)[0]
            print('   bad count', len(badi), 'first', badi[:8], 'last', badi[-3:])
            i0 = badi[0]
            print('   around first: gpu', gpu[i0 - 2:i0 + 3], '\n                 C  ', ora[i0 - 2:i0 + 3])
            m0, m1 = prog.member_range(0)
            for m in range(m0, min(m1, m0 + 6)):
                print('   member', m, 'idx', plan.member_index(m)[:12])
