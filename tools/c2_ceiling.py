"""What an 80 MB launch can reach at all: torch.fill_ / zero_ of the C2 output (1e7 fp64) timed like
also.c2 (200 launches back to back between two events), next to the C2 sampler launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler

n = 10**7
out = torch.empty((1, n), dtype=torch.float64, device='cuda')
def timed(fn, R=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / R * 1e3
print('fill_  %.1f us' % timed(lambda: out.fill_(1.5)))
print('zero_  %.1f us' % timed(lambda: out.zero_()))
x = torch.empty_like(out)
print('copy_  %.1f us (read + write)' % timed(lambda: out.copy_(x)))
bs = BatchSampler([wl.c2_channel(wf)], wl.c2_grid(n))
print('c2 sampler %.1f us' % timed(lambda: bs.launch_torch(out)))
for tpc in (1, 2, 3, 4, 6, 8):
    os.environ['WFK_TPC'] = str(tpc)
    b2 = BatchSampler([wl.c2_channel(wf)], wl.c2_grid(n))
    print('  tiles per chunk %d: %.1f us' % (tpc, timed(lambda: b2.launch_torch(out))))
    b2.close()
e = torch.empty((1, 16), dtype=torch.float64, device='cuda')
print('empty-ish launch (16 elements fill_) %.1f us' % timed(lambda: e.fill_(1.0)))
