"""tlist mode (arbitrary sorted sample times: Waveform.__call__(x) with a non-grid x): device libm per
factor and sample.  64 headline-style channels x 2e6 jittered times.  python tools/tlist_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten, workloads as wl
nch, n = 64, 2 * 10**6
chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
g = wl.make_grid(wl.c2_grid(n))
rng = np.random.default_rng(0)
t = np.sort(g + rng.normal(size=n) * (g[1] - g[0]) * 0.3)
prog = _flatten.flatten(chans)
plan = _engine.Plan(prog, t=t)
out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
f = lambda: plan.launch(out.data_ptr(), n, _engine.OUT_F64, stream=st)
for _ in range(2): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f'tlist {nch} x {n}: {ms:.2f} ms = {nch * n / ms * 1e-6:.0f} Gsamples/s  {plan.kernel_name()}')
