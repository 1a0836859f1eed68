"""IIR stage: three-launch block scan vs single-pass chained scan over shapes (same process, same box),
and what the library picks by itself.
python tools/iir_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from waveforms_amd import _engine


def bench(rows, n, nsec, dtype, onepass):
    if onepass is None:
        os.environ.pop('WFK_IIR_ONEPASS', None)
    else:
        os.environ['WFK_IIR_ONEPASS'] = '1' if onepass else '0'
    sos = butter(2 * nsec, 0.1, output='sos')
    plan = _engine.IirPlan([(s[:3], s[3:]) for s in sos], n, rows, dtype)
    td = torch.float64 if dtype == np.float64 else torch.float32
    torch.manual_seed(5)
    x = torch.randn((rows, n), dtype=td, device="cuda")
    y = torch.empty_like(x)
    zi = torch.zeros((rows, plan.state_dim), dtype=torch.float64, device='cuda')
    zf = torch.empty_like(zi)
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: plan.apply(x.data_ptr(), n, y.data_ptr(), n, zi.data_ptr(), zf.data_ptr(), 0.0, st)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = max(5, min(200, int(2e9 / (rows * n))))
    a.record()
    for _ in range(R):
        step()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / R
    plan.close()
    return ms, y


for dtype in (np.float64, np.float32):
    for nsec in (1, 2):
        for rows, n in ((1, 10**7), (4, 10**7), (8, 10**6), (16, 10**7), (32, 10**7), (64, 10**5), (64, 10**6), (64, 10**7), (256, 10**6),
                        (256, 10**7), (1024, 10**6), (4096, 10**5)):
            if rows * n * (8 if dtype == np.float64 else 4) * 2 > 60e9:
                continue
            t3, y3 = bench(rows, n, nsec, dtype, False)
            t1, y1 = bench(rows, n, nsec, dtype, True)
            td, _ = bench(rows, n, nsec, dtype, None)
            err = float((y1.double() - y3.double()).abs().max())
            b = 8 if dtype == np.float64 else 4
            print(f'{np.dtype(dtype).name} {nsec} biquad(s) {rows:5d} x {n:.0e}: three-launch {t3:8.4f} ms ({rows*n*2*b/t3*1e-9/8*100:4.1f}%)  '
                  f'single-pass {t1:8.4f} ms ({rows*n*2*b/t1*1e-9/8*100:4.1f}%)  ratio {t3/t1:5.2f}  default {td:8.4f} ms  max|diff| {err:.1e}', flush=True)
