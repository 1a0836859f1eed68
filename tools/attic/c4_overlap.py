"""C4 chain (sampler -> 1024-tap FIR): serial vs row-block pipelining on two streams
(sampler of block k+1 concurrent with the FIR of block k).  usage: c4_overlap.py [blocks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler
from waveforms_amd.distortion import FirStage

nch, n = 256, 10**7
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = nch // B
chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
grid = wl.c2_grid(n)
whole = BatchSampler(chans, grid)
parts = [BatchSampler(chans[b * rows:(b + 1) * rows], grid) for b in range(B)]
fir_all = FirStage(wl.c4_kernel(), n, nch, np.float64)
fir_blk = FirStage(wl.c4_kernel(), n, rows, np.float64)
x = torch.empty((nch, n), dtype=torch.float64, device='cuda')
y = torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def serial():
    whole.launch_torch(x)
    fir_all.apply_torch(x, y)

def piped():
    evs = []
    for b in range(B):
        with torch.cuda.stream(s1):
            parts[b].launch_torch(x[b * rows:(b + 1) * rows])
            e = torch.cuda.Event(); e.record(s1); evs.append(e)
        with torch.cuda.stream(s2):
            s2.wait_event(evs[b])
            fir_blk.apply_torch(x[b * rows:(b + 1) * rows], y[b * rows:(b + 1) * rows])
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)

def t(f, R=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R):
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / R
serial(); torch.cuda.synchronize(); ref = y.clone()
piped(); torch.cuda.synchronize()
print('max diff piped vs serial', float((y - ref).abs().max()))
print(f'serial {t(serial):.3f} ms   piped x{B} {t(piped):.3f} ms')
