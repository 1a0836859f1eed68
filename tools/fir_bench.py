"""FIR stage timing (C4's dominant kernel): rows x n, K taps, out of place on the device.
usage: python tools/fir_bench.py [rows] [n] [K] [f64|f32]      (WFK_LIB=... selects an A/B build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveforms_amd import _engine
from waveforms_amd.distortion import FirStage

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**7
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dt = np.float32 if (len(sys.argv) > 4 and sys.argv[4] == 'f32') else np.float64
ker = np.random.default_rng(1).normal(size=K); ker /= np.abs(ker).sum()
st = FirStage(ker, n, rows, dt)
tdt = torch.float64 if dt == np.float64 else torch.float32
x = torch.randn((rows, n), dtype=tdt, device='cuda')
y = torch.empty_like(x)
for _ in range(8):
    st.apply_torch(x, y)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
R = 10
for _ in range(R):
    st.apply_torch(x, y)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / R
by = 2 * rows * n * x.element_size()
print(f'fir {rows}x{n} {x.dtype} K={K} lib={os.environ.get("WFK_LIB", "default")}: {ms:.3f} ms  '
      f'{rows * n / ms * 1e-6:.1f} Gsamples/s  algorithmic {by / ms * 1e-9:.2f} TB/s ({by / ms * 1e-9 / 8 * 100:.1f}% of 8 TB/s)')
