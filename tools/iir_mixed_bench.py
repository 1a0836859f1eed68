import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from scipy.signal import butter
from waveforms_amd import _engine
def first(t): return (np.array([1.02, -np.exp(-1 / t) * 1.01]), np.array([1.0, -np.exp(-1 / t)]))
sos = butter(4, 0.1, output='sos')
bq = [(s[:3], s[3:]) for s in sos]
b3, a3 = butter(3, 0.2)
shapes = {'8 first-order': [first(t) for t in (30, 80, 200, 500, 1200, 3000, 8000, 20000)],
          'mixed 1,1,1,2,2': [first(50), first(400), first(3000)] + bq,
          'two order-3': [(b3, a3), (b3, a3)],
          '6 first-order': [first(t) for t in (30, 80, 200, 500, 1200, 3000)]}
rows, n = 64, 10**7
x = torch.randn((rows, n), dtype=torch.float64, device='cuda'); y = torch.empty_like(x)
for name, secs in shapes.items():
    plan = _engine.IirPlan(secs, n, rows, np.float64)
    zi = torch.zeros((rows, plan.state_dim), dtype=torch.float64, device='cuda'); zf = torch.empty_like(zi)
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: plan.apply(x.data_ptr(), n, y.data_ptr(), n, zi.data_ptr(), zf.data_ptr(), 0.0, st)
    for _ in range(2): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): step()
    b.record(); torch.cuda.synchronize()
    print(f'{name:18s} 64 x 1e7: {a.elapsed_time(b)/5:8.3f} ms', flush=True)
    plan.close()
