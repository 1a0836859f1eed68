"""Primitives without a fused form (sinc, chirp, mollifier, d-th Gaussian derivative): the general
kernel's direct (device libm per sample) tier.  python tools/direct_tier_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd._sampling import BatchSampler
nch, n, T = 64, 10**7, 3e-6
W = 20e-9
shapes = {
    'sinc': lambda k: wf.sinc(4 / W) >> ((k + 0.5) * 1.5 * W * 4),
    'linear chirp': lambda k: (wf.chirp(1e8, 3e8, 1.2 * W) >> (k * 1.5 * W)),
    'mollifier': lambda k: wf.mollifier(W) >> ((k + 0.5) * 1.5 * W),
    'gaussian d=1': lambda k: wf.gaussian(W, d=1) >> ((k + 0.5) * 1.5 * W),
    'gaussian (fused, for scale)': lambda k: wf.gaussian(W) >> ((k + 0.5) * 1.5 * W),
}
for name, mk in shapes.items():
    npul = 25 if name == 'sinc' else 100
    ws = [mk(k) for k in range(npul)]
    while len(ws) > 1:
        ws = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)] + ([ws[-1]] if len(ws) % 2 else [])
    bs = BatchSampler([ws[0]] * nch, ('linspace', 0.0, T, n, False))
    out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
    for _ in range(2): bs.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): bs.launch_torch(out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    i = bs.plan.info
    print(f'{name:28s} {nch} x {n}: {ms:7.2f} ms = {nch * n / ms * 1e-6:6.0f} Gsamples/s  fused {i.n_fused} generic {i.n_generic}  {bs.plan.kernel_name()}', flush=True)
    bs.close(); del out
