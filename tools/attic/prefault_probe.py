"""Does pre-faulting a fresh 80 MB NumPy output (madvise MADV_POPULATE_WRITE / touching one byte per
page) make the D2H copy into it cheaper than letting the DMA fault it in?"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import waveforms_amd as wf
from waveforms_amd import workloads as wl, _flatten, _engine
libc = ctypes.CDLL('libc.so.6', use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 10**7
prog = _flatten.flatten([wl.c2_channel(wf)])
plan = _engine.Plan(prog, grid=_flatten.grid_from_desc(('linspace', 0.0, 100 * wl.SPAN, n, False)))
run = lambda out: _engine.check(_engine.lib().wfk_plan_run_host(plan._h, out.ctypes.data, n, 0))
c = time.perf_counter
keep = []
for mode in ('plain', 'populate', 'touch', 'plain', 'populate', 'touch'):
    out = np.empty(n); keep.append(out)
    t0 = c()
    if mode == 'populate':
        a = out.ctypes.data & ~4095
        rc = libc.madvise(a, out.nbytes + (out.ctypes.data - a), 23)
        if rc != 0: print('madvise errno', ctypes.get_errno())
    elif mode == 'touch':
        out.view(np.uint8)[::4096] = 0
    t1 = c(); run(out); t2 = c()
    print('%-9s prefault %.2f ms  run_host %.2f ms  total %.2f ms' % (mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
