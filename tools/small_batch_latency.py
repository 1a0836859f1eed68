"""A typical experiment-sized batch: 16 channels x 20 pulses x 2e5 points -> host arrays.
Where does the time go (build / flatten / plan / launch+copy)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import waveforms_amd as wf
from waveforms_amd import workloads as wl, _flatten, _engine
from waveforms_amd._sampling import BatchSampler
from oracle import np_oracle
c = time.perf_counter
grid = ('linspace', 0.0, 20 * wl.SPAN, 200000, False)
for rep in range(3):
    t0 = c(); chans = [wl.sum_channel(wf, 20, 500 + k) for k in range(16)]
    t1 = c(); prog = _flatten.flatten(chans)
    t2 = c(); plan = _engine.Plan(prog, grid=_flatten.grid_from_desc(grid))
    t3 = c(); out = plan.run_host(np.float64)
    t4 = c(); plan.close()
    print('build %.2f  flatten %.2f  plan %.2f  run+copy %.2f  total %.2f ms' % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3))
t = wl.make_grid(grid)
t0 = c(); ref = [np_oracle.call(w, t) for w in chans]; print('numpy restatement of the reference: %.1f ms' % ((c() - t0) * 1e3))
print('max err', max(float(np.abs(out[k] - ref[k]).max()) for k in range(16)))
