"""README x waveform on np.linspace(-1e-6, 9e-6, n, endpoint=True) through every tier against the C oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import cases
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400001
w = cases.CASES['readme_x'][0](wf)
for endpoint in (True, False):
    g = _flatten.grid_from_desc(('linspace', -1e-6, 9e-6, n, endpoint))
    prog = _flatten.flatten([w])
    ora = c_oracle.eval_grid(prog, g)[0]
    for env in ({}, {'WFK_SHORT': '0'}, {'WFK_SHORT': '0', 'WFK_DISABLE_LEAN': '1'}, {'WFK_DISABLE_FUSE': '1'}, {'WFK_DISABLE_FAST': '1'}):
        os.environ.update(env)
        plan = _engine.Plan(prog, grid=g)
        got = plan.run_host(np.float64)[0]
        for k in env: del os.environ[k]
        d = np.abs(got - ora)
        i = int(np.argmax(d))
        print(endpoint, env, plan.kernel_name(), 'max err %.3g at %d (t=%.6g) ora %.6g got %.6g' % (d.max(), i, _flatten.grid_values(g)[i], ora[i], got[i]),
              'idx', plan.member_index(0)[:8], flush=True)
