"""one seed of tools/fuzz_soak.py awg in detail: python tools/attic/seed_diag.py 306405"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import cases
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten
seed = int(sys.argv[1])
rng = np.random.default_rng(10_000 + seed)
ch, grid = cases.random_awg_channel(wf, rng)
prog = _flatten.flatten([ch])
g = _flatten.grid_from_desc(grid)
ora = c_oracle.eval_grid(prog, g)[0]
plan = _engine.Plan(prog, grid=g)
print(plan.kernel_name(np.float32), grid, 'peak', np.abs(ora).max(), 'n', len(ora), 'complex', plan.prog.complex_amp)
for env in ({}, {'WFK_SHORT': '0'}, {'WFK_DISABLE_FUSE': '1'}):
    os.environ.update(env)
    p = _engine.Plan(prog, grid=g)
    for k in env: del os.environ[k]
    g32 = p.run_host(np.float32)[0].astype(np.float64)
    g64 = p.run_host(np.float64)[0]
    d = np.abs(g32 - ora); i = int(np.argmax(d))
    print(env, p.kernel_name(np.float32), 'e32 %.3g at %d: ora %.9g got32 %.9g got64 %.9g ; e64 %.3g' % (d.max(), i, ora[i], g32[i], g64[i], np.abs(g64 - ora).max()))
    lo = max(0, i - 3)
    print('   ora', ora[lo:i + 4]); print('   g32', g32[lo:i + 4])
print(type(ch).__name__, getattr(ch, 'min', None), getattr(ch, 'max', None), getattr(ch, 'offset', None))
