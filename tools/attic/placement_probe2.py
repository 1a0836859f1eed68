"""Per-launch kernel times of the headline plan inside one process (is the spread per
launch, per buffer, or per process?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler

nch, n = 256, 10**7
chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
bs = BatchSampler(chans, wl.c2_grid(n))
out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
for mode in ('events around every launch', 'back to back'):
    for _ in range(3):
        bs.launch_torch(out)
    torch.cuda.synchronize()
    if mode.startswith('events'):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); bs.launch_torch(out); b.record()
        torch.cuda.synchronize()
        print(mode, ' '.join(f'{a.elapsed_time(b):.2f}' for a, b in ev), flush=True)
    else:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            bs.launch_torch(out)
        b.record(); torch.cuda.synchronize()
        print(mode, f'{a.elapsed_time(b) / 30:.3f}', flush=True)
