"""Breakdown of the drop-in call on small inputs: flatten / plan create / index query / run / close."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import waveforms_amd as wf
from waveforms_amd import workloads as wl, _flatten, _engine, _sampling

def bench(w, t, grid=None, n=200):
    acc = dict(flatten=0, create=0, live=0, run=0, close=0)
    for it in range(n + 5):
        c = time.perf_counter
        t0 = c(); prog = _flatten.flatten([w])
        t1 = c(); plan = _engine.Plan(prog, t=t) if grid is None else _engine.Plan(prog, grid=_flatten.grid_from_desc(grid))
        t2 = c(); live = _sampling._live_pieces(plan, 0, w.seq); dt = _sampling._result_dtype(live)
        t3 = c(); res = plan.run_host(dt)
        t4 = c(); plan.close()
        t5 = c()
        if it >= 5:
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[k] += v
    return {k: round(v / n * 1e6, 1) for k, v in acc.items()}

x, y = wl.readme_xy(wf)
t = np.linspace(-1e-6, 9e-6, 10001)
print('README tlist 1e4 ', bench(x, t))
print('README grid  1e4 ', bench(x, None, ('linspace', -1e-6, 9e-6, 10001, True)))
w = wl.sum_channel(wf, 100, 1000)
t2 = np.linspace(0, 100 * wl.SPAN, 100001)
print('100p   tlist 1e5 ', bench(w, t2, n=50))
print('100p   grid  1e5 ', bench(w, None, ('linspace', 0.0, 100 * wl.SPAN, 100001, True), n=50))
