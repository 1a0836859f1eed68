"""Does the sampler's kernel time depend on WHICH buffer it writes?  
Times the headline plan into several separately allocated output tensors of one process,
plus row-padded variants (ch_stride > n)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler

nch, n = 256, 10**7
chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
bs = BatchSampler(chans, wl.c2_grid(n))

def timeit(out, reps=10):
    for _ in range(2):
        bs.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        bs.launch_torch(out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

bufs = []
for i in range(6):
    out = torch.empty((nch, n), dtype=torch.float64, device='cuda')
    bufs.append(out)
    print(f'buffer {i} ptr=0x{out.data_ptr():x}  {timeit(out):.3f} ms', flush=True)
print('again, same buffers:', ' '.join(f'{timeit(o):.3f}' for o in bufs), flush=True)
del bufs
torch.cuda.empty_cache()
for pad in (0, 512, 4096, 65536, 262144 - (n % 262144)):
    out = torch.empty((nch, n + pad), dtype=torch.float64, device='cuda')
    print(f'row stride n+{pad}: {timeit(out):.3f} ms', flush=True)
    del out
