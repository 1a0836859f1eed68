"""Flat-top pulses with erf edges under ONE carrier on a fine grid (lean kernel family 1: erf closing op, no tone bank):
    python tools/flattop_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd._sampling import BatchSampler
nch, n, T = 64, 10**7, 3e-6
rng = np.random.default_rng(0)
chans = []
for c in range(nch):
    w = wf.zero()
    for k in range(50):
        w = w + ((wf.square(30e-9, edge=4e-9) >> ((k + 0.5) * 60e-9)) * (rng.uniform(0.2, 1) * wf.cos(2 * np.pi * rng.uniform(-300e6, 300e6), rng.uniform(0, 6))))
    chans.append(w)
bs = BatchSampler(chans, ('linspace', 0.0, T, n, False))
out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
for _ in range(3): bs.launch_torch(out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): bs.launch_torch(out)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
i = bs.plan.info
print(f'flat tops, one carrier, {nch} x {n}: {ms:.3f} ms ({nch * n * 8 / ms * 1e-9 / 8 * 100:.1f}% of 8 TB/s)  {bs.plan.kernel_name()}  fused {i.n_fused} generic {i.n_generic}  lib={os.environ.get("WFK_LIB", "tree")}')
