"""IIR single-section (lfilter) timing by order: python tools/iir_bench_lf.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from waveforms_amd import _engine
rows, n = 64, 10**7
x = torch.randn((rows, n), dtype=torch.float64, device='cuda'); y = torch.empty_like(x)
for order in (4, 6, 7, 8, 9, 10):
    b, a = butter(order, 0.2)
    plan = _engine.IirPlan([(b, a)], n, rows, np.float64)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: plan.apply(x.data_ptr(), n, y.data_ptr(), n, None, None, 0.0, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print(f'lfilter order {order}: {e0.elapsed_time(e1) / 5:.3f} ms')
    plan.close()
