#!/usr/bin/env python3
"""Kernel time of the AWG-rate workload (rows x n_pts at `rate`), short tier vs the standard tiers.
    python tools/awg_bench.py [rows] [n_pts] [rate_GSps] [duty30]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten, workloads as wl

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100000
rate = float(sys.argv[3]) * 1e9 if len(sys.argv) > 3 else 2e9
duty30 = len(sys.argv) > 4 and sys.argv[4] == '1'
only = sys.argv[5] if len(sys.argv) > 5 else ''
distinct = min(rows, 16)
t0 = time.time()
chans = [wl.awg_channel(wf, c, n, rate, duty30) for c in range(distinct)]
g = _flatten.grid_from_desc(wl.awg_grid(n, rate))
prog = _flatten.tile_program(_flatten.flatten(chans, g), rows // distinct)
t1 = time.time()
plan = _engine.Plan(prog, grid=g)
t2 = time.time()
print(f'front-end {t1 - t0:.2f} s, plan {t2 - t1:.2f} s, kernel {plan.kernel_name()}, pieces {plan.info.n_pieces}, '
      f'units {plan.info.n_tiles}, table doubles {plan.info.param_doubles}')
for dt, kind in ((torch.float64, _engine.OUT_F64), (torch.float32, _engine.OUT_F32)):
    if only and only not in str(dt):
        continue
    out = torch.empty((plan.n_channels, plan.n), dtype=dt, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    t_pre = time.time()
    while time.time() - t_pre < 0.4:            # clocks ramp for ~0.1 s after idle
        for _ in range(20):
            plan.launch(out.data_ptr(), plan.n, kind, False, st)
        torch.cuda.synchronize()
    reps = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            plan.launch(out.data_ptr(), plan.n, kind, False, st)
        e1.record()
        torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) / 50)
    ms = sorted(reps)[len(reps) // 2]
    nb = out.numel() * out.element_size()
    print(f'{dt}: {ms:.4f} ms (min {min(reps):.4f} max {max(reps):.4f})  {out.numel() / ms / 1e6:.1f} Gsamples/s  {nb / ms / 1e6:.0f} GB/s  frac {nb / ms / 1e6 / 8000:.3f}')
    del out
