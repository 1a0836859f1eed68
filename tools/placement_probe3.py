"""Time drift vs buffer identity for the headline plan (one process)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler

nch, n = 256, 10**7
chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
bs = BatchSampler(chans, wl.c2_grid(n))

def series(out, groups=8, per=10):
    res = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(per):
            bs.launch_torch(out)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / per)
    return ' '.join(f'{x:.2f}' for x in res)

outs = []
for i in range(4):
    out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
    outs.append(out)
    print(f'buf{i} 0x{out.data_ptr():x}:', series(out), flush=True)
for i, out in enumerate(outs):
    print(f'buf{i} again:', series(out, 3), flush=True)
