"""Fully fusable multi-tone pulses: every Gaussian pulse carries NT tones (frequency-multiplexed
drive): NT fused ops per piece.  python tools/multitone_bench.py [NT]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nch, n = 64, 10**7
rng = np.random.default_rng(0)
chans = []
for c in range(nch):
    ws = []
    for k in range(100):
        tones = None
        for j in range(NT):
            tone = rng.uniform(0.05, 0.2) * wf.cos(2 * np.pi * rng.uniform(-300e6, 300e6), rng.uniform(0, 6))
            tones = tone if tones is None else tones + tone
        ws.append((wf.gaussian(wl.W) >> ((k + 0.5) * wl.SPAN)) * tones)
    while len(ws) > 1:
        ws = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)] + ([ws[-1]] if len(ws) % 2 else [])
    chans.append(ws[0])
bs = BatchSampler(chans, wl.c2_grid(n))
out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
for _ in range(3): bs.launch_torch(out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): bs.launch_torch(out)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
i = bs.plan.info
print(f'{NT} tones/pulse, {nch} x {n}: {ms:.2f} ms = {nch * n / ms * 1e-6:.0f} Gsamples/s ({nch * n * 8 / ms * 1e-9 / 8 * 100:.1f}% of 8 TB/s); fused {i.n_fused} generic {i.n_generic}')
