"""Row-stride alignment vs buffer placement for the headline plan: 3 fresh buffers per stride."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler

nch, n = 256, 10**7
chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
bs = BatchSampler(chans, wl.c2_grid(n))

def t(out, per=10):
    for _ in range(3):
        bs.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(per):
        bs.launch_torch(out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / per

def up(x, m):
    return (x + m - 1) // m * m

for label, stride in [('n', n), ('8 KB tile', up(n, 1024)), ('64 KB', up(n, 8192)), ('2 MB', up(n, 262144)),
                      ('2 MB', up(n, 262144)), ('n', n), ('1 GB/8', up(n, 16 * 1024 * 1024))]:
    keep = []
    res = []
    for i in range(3):
        out = torch.empty((nch, stride), dtype=torch.float64, device='cuda')
        keep.append(out)
        res.append(t(out))
    print(f'stride {label:10s} ({stride}): ' + ' '.join(f'{x:.3f}' for x in res), flush=True)
    del keep, out
    torch.cuda.empty_cache()
