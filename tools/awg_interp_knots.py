"""wfk_sample_short on optimised-pulse trains (workloads.awg_interp_channel) against the number of knots per envelope:
is the per-lane table gather bound by the bytes it touches or by the number of lanes that gather?
    python tools/awg_interp_knots.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler

for knots in (16, 61, 301, 1201):
    bs = BatchSampler([wl.awg_interp_channel(wf, c, knots=knots) for c in range(16)], wl.awg_grid(), tile=128)
    out = torch.empty((bs.n_channels, bs.n), dtype=torch.float64, device='cuda')
    for _ in range(2): bs.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): bs.launch_torch(out)
    b.record(); torch.cuda.synchronize()
    print(f'{knots:5d} knots ({knots / 60:.2f} per sample): {a.elapsed_time(b) / 10:.3f} ms  {bs.plan.kernel_name()}  tables {bs.plan.table_bytes() / 1e6:.1f} MB', flush=True)
    bs.close()
bs = BatchSampler([wl.awg_channel(wf, c) for c in range(16)], wl.awg_grid(), tile=128)
out = torch.empty((bs.n_channels, bs.n), dtype=torch.float64, device='cuda')
for _ in range(2): bs.launch_torch(out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): bs.launch_torch(out)
b.record(); torch.cuda.synchronize()
print(f'gaussian+DRAG pulses (no table): {a.elapsed_time(b) / 10:.3f} ms  {bs.plan.kernel_name()}', flush=True)
