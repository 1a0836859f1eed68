"""Where the time of a drop-in call on 1e7 points goes (C2 channel): tlist `wav(t)` and grid `wav.sample()`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import waveforms_amd as wf
from waveforms_amd import workloads as wl, _flatten, _engine

w = wl.c2_channel(wf)
n = 10**7
t = np.linspace(0, 100 * wl.SPAN, n, endpoint=False)
c = time.perf_counter
for rep in range(3):
    t0 = c(); prog = _flatten.flatten([w])
    t1 = c(); plan = _engine.Plan(prog, t=t)
    t2 = c(); out = np.empty((1, n))
    t3 = c(); _engine.check(_engine.lib().wfk_plan_run_host(plan._h, out.ctypes.data, n, 0))
    t4 = c(); _engine.check(_engine.lib().wfk_plan_run_host(plan._h, out.ctypes.data, n, 0))
    t5 = c(); plan.close()
    print('tlist: flatten %.2f  plan(create+upload t) %.2f  empty %.3f  run_host(fresh out) %.2f  run_host(touched out) %.2f ms'
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
g = _flatten.grid_from_desc(('linspace', 0.0, 100 * wl.SPAN, n, False))
for rep in range(3):
    t1 = c(); plan = _engine.Plan(prog, grid=g)
    t2 = c(); out = np.empty((1, n))
    t3 = c(); _engine.check(_engine.lib().wfk_plan_run_host(plan._h, out.ctypes.data, n, 0))
    t4 = c(); _engine.check(_engine.lib().wfk_plan_run_host(plan._h, out.ctypes.data, n, 0))
    t5 = c(); plan.close()
    print('grid : plan %.2f  run_host(fresh out) %.2f  run_host(touched out) %.2f ms' % ((t2 - t1) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
t0 = c(); y = w(t); print('w(t) total %.2f ms' % ((c() - t0) * 1e3))
t0 = c(); y = w(t); print('w(t) total %.2f ms' % ((c() - t0) * 1e3))
# round 3: the drop-in call with out= (reused buffer), without, and sample()
out = np.zeros(n)
for rep in range(4):
    t0 = c(); r = w(t, out=out); t1 = c()
    print('w(t, out=out) %.2f ms' % ((t1 - t0) * 1e3))
for rep in range(4):
    t0 = c(); y = w(t); t1 = c()
    print('w(t) %.2f ms' % ((t1 - t0) * 1e3))
    del y
t0 = c(); gd = _engine.detect_grid(t); print('detect_grid %.2f ms' % ((c() - t0) * 1e3))
w.start, w.stop, w.sample_rate = 0.0, 100 * wl.SPAN, n / (100 * wl.SPAN)
for rep in range(3):
    t0 = c(); y = w.sample(); print('w.sample() %.2f ms, n = %d' % ((c() - t0) * 1e3, len(y)))
    del y
for m in (10001, 100000):
    tt = np.linspace(0, 100 * wl.SPAN, m, endpoint=False)
    w(tt)
    t0 = c()
    for _ in range(50):
        w(tt)
    print('w(t) on %d points: %.1f us' % (m, (c() - t0) / 50 * 1e6))
# round 4: a NON-grid x of 1e7 points (jittered) through the drop-in call: upload of x + the time-list tier + copy back;
# the finite scan that out= costs; an x of four grids back to back (sampled run by run)
tj = wl.jittered_times(n)
for rep in range(3):
    t0 = c(); y = w(tj); print('w(jittered x) %.2f ms' % ((c() - t0) * 1e3)); del y
t0 = c(); ok = _engine.all_finite(out); print('all_finite(1e7 doubles) %.2f ms' % ((c() - t0) * 1e3))
tr = np.concatenate([np.linspace(k * 0.8e-6, (k * 0.8 + 0.7) * 1e-6, n // 4, endpoint=False) for k in range(4)])
t0 = c(); runs = _engine.detect_grid_runs(tr); print('detect_grid_runs %.2f ms -> %d runs' % ((c() - t0) * 1e3, len(runs)))
for rep in range(3):
    t0 = c(); y = w(tr); print('w(4 grids back to back) %.2f ms' % ((c() - t0) * 1e3)); del y
