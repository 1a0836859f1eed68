"""VALU instructions per sample of the general kernel's per-factor fast paths, one launch per shape (run under
rocprofv3 --pmc SQ_INSTS_VALU: dispatches appear in this order).  64 rows x 1e6 points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler
n = 10**6
T = 3e-6
env = lambda k: wf.samplingPoints(-wl.SPAN / 2, wl.SPAN / 2, np.hanning(1000)) >> ((k + 0.5) * wl.SPAN)
shapes = {
    'interp x cos': lambda: wl._tree_sum([env(k) * wf.cos(2 * np.pi * 1e8, 0.1 * k) for k in range(100)]),
    'interp alone': lambda: wl._tree_sum([env(k) for k in range(100)]),
    'mollifier': lambda: wl._tree_sum([wf.mollifier(wl.W) >> ((k + 0.5) * wl.SPAN) for k in range(100)]),
    'sinc x square (1 term)': lambda: wl._tree_sum([(wf.sinc(4 / wl.W) * wf.square(wl.SPAN)) >> ((k + 0.5) * wl.SPAN) for k in range(100)]),
}
for name, mk in shapes.items():
    bs = BatchSampler([mk()] * 64, ('linspace', 0.0, T, n, False))
    out = torch.empty((64, n), dtype=torch.float64, device='cuda')
    bs.launch_torch(out)
    torch.cuda.synchronize()
    i = bs.plan.info
    print(name, bs.plan.kernel_name(), 'fast', i.n_fast, 'direct', i.n_direct, 'generic', i.n_generic, flush=True)
    bs.close()
