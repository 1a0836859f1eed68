"""Does the sampler's write rate depend on the row stride / base alignment of the output buffer,
or on the tiles-per-chunk of the plan?  (71 % vs 81 % between boxes.)
Same plan, same process; the output is a view into one large allocation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler

nch, n = 256, 10**7
chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
pool = torch.empty(nch * (n + 2**21) + 2**22, dtype=torch.float64, device='cuda')

def timed(bs, out, R=20):
    for _ in range(8): bs.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R): bs.launch_torch(out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / R

bs = BatchSampler(chans, wl.c2_grid(n))
base_ms = None
for name, stride, off in (('stride n (80 000 000 B)', n, 0), ('n + 8 (64 B)', n + 8, 0), ('n + 512 (4 KB)', n + 512, 0),
                          ('2 MB multiple', ((n * 8 + 2**21 - 1) // 2**21) * 2**21 // 8, 0),
                          ('2 MB multiple + 4 KB', ((n * 8 + 2**21 - 1) // 2**21) * 2**21 // 8 + 512, 0),
                          ('n, base + 64 B', n, 8), ('n, base + 4 KB', n, 512), ('n + 2^18 + 64', n + 2**18 + 64, 0)):
    out = torch.as_strided(pool, (nch, n), (stride, 1), off)
    ms = timed(bs, out)
    print(f'{name:28s} {ms:.3f} ms  {nch * n * 8 / ms * 1e-9:.2f} TB/s', flush=True)
bs.close()
for tpc in (2, 4, 6, 8, 12, 16):
    os.environ['WFK_TPC'] = str(tpc)
    b2 = BatchSampler(chans, wl.c2_grid(n))
    out = torch.as_strided(pool, (nch, n), (n, 1), 0)
    print(f'tiles per chunk {tpc:2d}: {timed(b2, out):.3f} ms', flush=True)
    b2.close()
