"""IIR stage timing: rows x n fp64, SOS cascade, in place on the device.
usage: python tools/iir_bench.py [rows] [n] [sections] [first]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from waveforms_amd import _engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**7
nsec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
first = len(sys.argv) > 4 and sys.argv[4] == 'first'     # cascade of first-order sections instead of biquads
if first:
    taus = [50.0, 400.0, 3000.0, 20000.0][:nsec]
    secs = [(np.array([1.02, -np.exp(-1 / t) * 1.01]), np.array([1.0, -np.exp(-1 / t)])) for t in taus]
else:
    sos = butter(2 * nsec, 0.1, output='sos')
    secs = [(s[:3], s[3:]) for s in sos]
plan = _engine.IirPlan(secs, n, rows, np.float64)
D = plan.state_dim
x = torch.randn((rows, n), dtype=torch.float64, device='cuda')
y = torch.empty_like(x)
zi = torch.zeros((rows, D), dtype=torch.float64, device='cuda')
zf = torch.empty_like(zi)
stream = torch.cuda.current_stream().cuda_stream
def step():
    plan.apply(x.data_ptr(), n, y.data_ptr(), n, zi.data_ptr(), zf.data_ptr(), 0.0, stream)
for _ in range(5):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
R = 20
for _ in range(R):
    step()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / R
print(f'iir {rows}x{n} fp64, {nsec} {"first-order sections" if first else "biquads"}: {ms:.3f} ms  {rows * n / ms * 1e-6:.1f} Gsamples/s  '
      f'algorithmic 16 B/sample -> {rows * n * 16 / ms * 1e-9:.2f} TB/s ({rows * n * 16 / ms * 1e-9 / 8 * 100:.1f}% of 8 TB/s)')
