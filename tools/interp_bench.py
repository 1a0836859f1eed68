"""Arbitrary-envelope pulses (samplingPoints = INTERP primitive, e.g. optimal-control shapes) under a
carrier: throughput of the direct-primitive tier.  python tools/interp_bench.py [knots]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd._sampling import BatchSampler
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nch, n, W = 64, 10**7, 30e-9
rng = np.random.default_rng(0)
chans = []
for c in range(nch):
    w = wf.zero()
    for k in range(100):
        env = wf.samplingPoints(-W / 2, W / 2, np.hanning(m) * rng.uniform(0.5, 1))
        I, _ = wf.mixing(env >> ((k + 0.5) * W), freq=rng.uniform(-200e6, 200e6), phase=rng.uniform(0, 6))
        w = w + I
    chans.append(w)
bs = BatchSampler(chans, ('linspace', 0.0, 100 * W, n, False))
out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
for _ in range(3): bs.launch_torch(out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): bs.launch_torch(out)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
i = bs.plan.info
print(f'INTERP envelopes ({m} knots) x carrier, {nch} x {n}: {ms:.2f} ms = {nch * n / ms * 1e-6:.1f} Gsamples/s '
      f'({nch * n * 8 / ms * 1e-9 / 8 * 100:.1f}% of 8 TB/s); fused {i.n_fused} generic {i.n_generic} fast {i.n_fast} direct {i.n_direct}')
