import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, waveforms_amd as wf, os
from waveforms_amd import _flatten,_engine
from oracle import c_oracle
import cases
build,grid=cases.CASES['deriv_misc']
prog=_flatten.flatten([build(wf)]); g=_flatten.grid_from_desc(grid)
want=c_oracle.eval_grid(prog,g)[0]
plan=_engine.Plan(prog,grid=g); print(plan.kernel_name())
got=plan.run_host(np.float64)[0]
bad=np.nonzero(np.abs(got-want)>1e-9)[0]
print(len(bad), bad[:20], bad[-20:])
print(got[bad[:5]], want[bad[:5]])
