"""INTERP envelopes in tlist mode (arbitrary x): kernel-only time of one 100-pulse channel, 4e6 points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten
W, n = 30e-9, 4 * 10**6
rng = np.random.default_rng(0)
w = wf.zero()
for k in range(100):
    env = wf.samplingPoints(-W / 2, W / 2, np.hanning(1000) * rng.uniform(0.5, 1))
    I, _ = wf.mixing(env >> ((k + 0.5) * W), freq=rng.uniform(-200e6, 200e6), phase=rng.uniform(0, 6))
    w = w + I
t = np.linspace(0, 100 * W, n, endpoint=False)
plan = _engine.Plan(_flatten.flatten([w]), t=t)
out = torch.empty(n, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): plan.launch(out.data_ptr(), n, _engine.OUT_F64, False, st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): plan.launch(out.data_ptr(), n, _engine.OUT_F64, False, st)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f'tlist INTERP x carrier, 1 x {n}: {ms * 1e3:.1f} us = {n / ms * 1e-6:.1f} Gsamples/s')
