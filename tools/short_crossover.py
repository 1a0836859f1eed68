#!/usr/bin/env python3
"""Where the short tier (wfk_short.hip) stops paying: back-to-back gaussian+DRAG pulses of growing
length at 2 GS/s, sampled with the short tier forced (WFK_SHORT=1) and forbidden (WFK_SHORT=0).
    python tools/short_crossover.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten, workloads as wl

rate, n, rows = 2e9, 10**6, 256


def channel(c, width):
    rng = np.random.default_rng(9000 + c)
    span = 1.5 * width
    nseg = int(n / rate / span)
    ws = []
    for k in range(nseg):
        A, f, phi = rng.uniform(0.1, 1), rng.uniform(-200e6, 200e6), rng.uniform(0, 2 * np.pi)
        I, _ = wf.mixing(A * wf.gaussian(width) >> ((k + 0.5) * span), freq=f, phase=phi, DRAGScaling=1e-10)
        ws.append(I)
    while len(ws) > 1:
        nxt = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)]
        if len(ws) % 2:
            nxt.append(ws[-1])
        ws = nxt
    return ws[0]


def timed(plan, out):
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(10):
            plan.launch(out.data_ptr(), plan.n, _engine.OUT_F64, False, st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        plan.launch(out.data_ptr(), plan.n, _engine.OUT_F64, False, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30


g = _flatten.grid_from_desc(('arange', 0.0, n / rate, 1.0 / rate))
out = torch.empty((rows, n), dtype=torch.float64, device='cuda')
for L in (30, 60, 120, 240, 480, 960, 1536, 2048, 3072, 4096, 8192, 16384):
    width = L / 1.5 / rate
    distinct = 8
    prog = _flatten.tile_program(_flatten.flatten([channel(c, width) for c in range(distinct)], g), rows // distinct)
    res = {}
    for mode in ('1', '0'):
        os.environ['WFK_SHORT'] = mode
        plan = _engine.Plan(prog, grid=g)
        res[mode] = (timed(plan, out), plan.kernel_name())
        plan.close()
    print(f'{L:6d} samples/piece: short {res["1"][0]:.4f} ms ({res["1"][1]}), standard {res["0"][0]:.4f} ms ({res["0"][1]})', flush=True)
