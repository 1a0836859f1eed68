"""single-pass IIR: persistent waves per row (WFK_IIR_OP_DEPTH) vs one chunk per workgroup, few long rows.
python tools/iir_depth.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from waveforms_amd import _engine

def bench(rows, n, nsec, depth):
    os.environ['WFK_IIR_OP_DEPTH'] = str(depth)
    sos = butter(2 * nsec, 0.1, output='sos')
    plan = _engine.IirPlan([(s[:3], s[3:]) for s in sos], n, rows, np.float64)
    x = torch.randn((rows, n), dtype=torch.float64, device='cuda')
    y = torch.empty_like(x)
    zi = torch.zeros((rows, plan.state_dim), dtype=torch.float64, device='cuda')
    zf = torch.empty_like(zi)
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: plan.apply(x.data_ptr(), n, y.data_ptr(), n, zi.data_ptr(), zf.data_ptr(), 0.0, st)
    for _ in range(3): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = max(5, min(100, int(2e9 / (rows * n))))
    a.record()
    for _ in range(R): step()
    b.record(); torch.cuda.synchronize()
    plan.close()
    return a.elapsed_time(b) / R

def three(rows, n, nsec):
    os.environ['WFK_IIR_ONEPASS'] = '0'
    try:
        return bench(rows, n, nsec, 0)
    finally:
        del os.environ['WFK_IIR_ONEPASS']

for rows, n in ((8, 10**7), (12, 10**7), (16, 10**7), (16, 10**6), (24, 10**7), (32, 10**7), (32, 10**6), (48, 10**7), (64, 10**7), (64, 10**6),
                (128, 10**7), (256, 10**7), (512, 10**6), (1024, 10**6), (4096, 10**5)):
    line = f'{rows:4d} x {n:.0e}, 2 biquads: three-launch {three(rows, n, 2):7.3f}  single-pass'
    for depth in (0, 2, 3, 4, 6, 9, 12, 18, 24, 36):
        if depth and not (1500 <= depth * rows <= 4700): continue
        line += f'  d{depth} {bench(rows, n, 2, depth):7.3f}'
    print(line, flush=True)
