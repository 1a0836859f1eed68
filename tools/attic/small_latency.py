"""Drop-in call latency on small inputs (C1: README sequence, 10 001 points): where does the
time go between Python and the device?  Prints per-call wall times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import waveforms_amd as wf
from waveforms_amd import workloads as wl, _flatten, _engine
from oracle import np_oracle

x, y = wl.readme_xy(wf)
t = np.linspace(-1e-6, 9e-6, 10001)
def timeit(f, n=200):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print('wav(t) drop-in        %8.1f us' % timeit(lambda: x(t)))
print('flatten only          %8.1f us' % timeit(lambda: _flatten.flatten([x])))
print('numpy oracle (CPU)    %8.1f us' % timeit(lambda: np_oracle.call(x, t), 50))
w = wl.sum_channel(wf, 100, 1000)
t2 = np.linspace(0, 100 * wl.SPAN, 100001)
print('100-pulse wav(t) 1e5  %8.1f us' % timeit(lambda: w(t2), 50))
print('  numpy oracle (CPU)  %8.1f us' % timeit(lambda: np_oracle.call(w, t2), 10))
w.start, w.stop, w.sample_rate = 0.0, 100 * wl.SPAN, 1e5 / (100 * wl.SPAN)
print('  wav.sample() grid   %8.1f us' % timeit(lambda: w.sample(), 50))
