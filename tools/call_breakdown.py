"""Where a small drop-in call spends its time: python tools/call_breakdown.py [tree]   (tree: a checkout holding waveforms_amd/)"""
import sys, os, time
tree = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, tree)
import numpy as np
import waveforms_amd as wf
from waveforms_amd import workloads as wl, _flatten, _engine
print('tree', os.path.dirname(wf.__file__))
w = wl.c2_channel(wf)
x, y = wl.readme_xy(wf)
c = time.perf_counter
for name, ww, tt in (('readme 10001', x, np.linspace(-1e-6, 9e-6, 10001)), ('c2 1e4', w, np.linspace(0, 100 * wl.SPAN, 10001, endpoint=False)),
                     ('c2 1e5', w, np.linspace(0, 100 * wl.SPAN, 100000, endpoint=False))):
    ww(tt)
    N = 100
    t0 = c()
    for _ in range(N): ww(tt)
    tot = (c() - t0) / N
    t0 = c()
    for _ in range(N): g = _engine.detect_grid(tt)
    td = (c() - t0) / N
    t0 = c()
    for _ in range(N): prog = _flatten.flatten([ww], g)
    tf = (c() - t0) / N
    t0 = c()
    for _ in range(N):
        plan = _engine.Plan(prog, grid=g); 
        plan.close()
    tp = (c() - t0) / N
    plan = _engine.Plan(prog, grid=g)
    t0 = c()
    for _ in range(N): plan.run_host(np.float64)
    tr = (c() - t0) / N
    print('%-14s call %.0f us = detect %.0f + flatten %.0f + plan %.0f + run_host %.0f (+ %.0f)  %s' %
          (name, tot * 1e6, td * 1e6, tf * 1e6, tp * 1e6, tr * 1e6, (tot - td - tf - tp - tr) * 1e6, plan.kernel_name()), flush=True)
