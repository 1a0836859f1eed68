"""20 000 drop-in calls of varying size: device memory in use before/after (block cache bounded)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd.distortion import predistort
free0, total = torch.cuda.mem_get_info()
rng = np.random.default_rng(0)
w = wl.sum_channel(wf, 10, 3)
for it in range(20000):
    n = int(rng.integers(1, 200000))
    t = np.linspace(0, 10 * wl.SPAN, n)
    y = w(t)
    if it % 50 == 0:
        predistort(y, ker=np.ones(int(rng.integers(1, 3000))) / 7)
        predistort(y, filters=[([0.1, 0.0], [1.0, -0.9])])
    if it % 5000 == 0:
        free, _ = torch.cuda.mem_get_info()
        print(it, 'device memory in use by this process (delta): %.1f MB' % ((free0 - free) / 1e6), flush=True)
free, _ = torch.cuda.mem_get_info()
print('end: %.1f MB' % ((free0 - free) / 1e6))
