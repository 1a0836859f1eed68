"""Complex-valued channels (I + 1j*Q of every pulse): complex128 output, 16 B/sample."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler
nch, n = 128, 10**7
def chan(seed):
    rng = np.random.default_rng(seed)
    ws = []
    for k in range(100):
        A, f, ph = rng.uniform(0.1, 1), rng.uniform(-200e6, 200e6), rng.uniform(0, 2 * np.pi)
        I, Q = wf.mixing(A * wf.gaussian(wl.W) >> ((k + 0.5) * wl.SPAN), freq=f, phase=ph, DRAGScaling=1e-10)
        ws.append(I + 1j * Q)
    while len(ws) > 1:
        ws = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)] + ([ws[-1]] if len(ws) % 2 else [])
    return ws[0]
bs = BatchSampler([chan(1000 + c) for c in range(nch)], wl.c2_grid(n))
out = torch.empty((nch, n), dtype=torch.complex128, device='cuda')
for _ in range(3): bs.launch_torch(out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): bs.launch_torch(out)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
i = bs.plan.info
print(f'complex128 {nch} x {n}: {ms:.3f} ms = {nch * n / ms * 1e-6:.0f} Gsamples/s = {nch * n * 16 / ms * 1e-9:.2f} TB/s '
      f'({nch * n * 16 / ms * 1e-9 / 8 * 100:.1f}% of 8 TB/s); fused {i.n_fused} generic {i.n_generic}')
