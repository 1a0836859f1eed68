import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import cases, golden_io
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, _sampling
FUZZ = golden_io.npz('fuzz.npz')
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 31
chans, grid = cases.far_golden_case(wf, seed)
g = _flatten.grid_from_desc(grid)
t = c_oracle.grid_values(g)
plan = _engine.Plan(_flatten.flatten(chans), grid=g)
got = plan.run_host(np.float64)
print(grid, plan.kernel_name(), plan.info.n_direct, plan.info.n_generic, plan.info.n_fused)
for c, w in enumerate(chans):
    want = FUZZ[f'far{seed}.{c}']
    pk = max(1.0, np.abs(want).max())
    p2, _ = _sampling._plan_for_axis(w, t, None)
    dr = np.real(w(t))
    print(c, 'batch err %.2e  drop-in err %.2e  pk %.2e  drop-in kernel %s' % (np.abs(got[c]-want).max()/pk, np.abs(dr-want).max()/pk, pk, p2.kernel_name()))
    k = int(np.argmax(np.abs(dr - want)))
    print('   worst at', k, dr[k], want[k])
    for b, e in zip(w.bounds if hasattr(w, 'bounds') else [], w.seq if hasattr(w, 'seq') else []):
        print('   ', b, str(e)[:300])
print('--- channel 0 alone, grid plan')
for env in ('0', '1'):
    os.environ['WFK_DISABLE_CORR'] = env
    prog = _flatten.flatten([chans[0]])
    p = _engine.Plan(prog, grid=g)
    y = p.run_host(np.float64)[0]
    o = c_oracle.eval_grid(prog, g)[0]
    e = np.abs(y - o)
    print('DISABLE_CORR', env, p.kernel_name(), 'err %.2e' % e.max(), 'at', int(np.argmax(e)), 'first piece sample', np.nonzero(o)[0][:1])
    if env == '0':
        nz = np.nonzero(o)[0]
        for k in list(nz[:3]) + [1024, 1025, 1088, 1165]:
            print('    ', k, y[k] - o[k])
