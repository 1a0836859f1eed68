#!/usr/bin/env python3
"""Flat-top pulses (square(40 ns, edge 5 ns) x carrier, 75 ns apart) at 2 GS/s: the erf edges in the short
tier (closing op, libm erf per edge sample) against the standard tiers (WFK_SHORT=0).
    python tools/awg_flattop_bench.py [rows] [n_pts]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100000
rate = 2e9


def channel(c):
    rng = np.random.default_rng(8000 + c)
    ws = []
    for k in range(int(n / rate / 75e-9)):
        ws.append(rng.uniform(0.2, 1) * (wf.square(40e-9, edge=5e-9) >> (35e-9 + 75e-9 * k)) *
                  wf.cos(2 * np.pi * rng.uniform(30e6, 200e6), rng.uniform(0, 6)))
    while len(ws) > 1:
        ws = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)] + ([ws[-1]] if len(ws) % 2 else [])
    return ws[0]


g = _flatten.grid_arange(0.0, n / rate, 1 / rate)
prog = _flatten.tile_program(_flatten.flatten([channel(c) for c in range(8)], g), rows // 8)
out = torch.empty((rows, n), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for mode in ('1', '0'):
    os.environ['WFK_SHORT'] = mode
    plan = _engine.Plan(prog, grid=g)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        plan.launch(out.data_ptr(), n, _engine.OUT_F64, False, st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.launch(out.data_ptr(), n, _engine.OUT_F64, False, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'WFK_SHORT={mode}: {ms:.3f} ms  {rows * n / ms / 1e6:.1f} Gsamples/s  frac {rows * n * 8 / ms / 1e6 / 8000:.3f}  {plan.kernel_name()}', flush=True)
    plan.close()
