#!/usr/bin/env python3
"""Short tier (wfk_short.hip) against the C oracle on AWG-rate grids: parity + which kernel ran.
    python tools/awg_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl
from waveforms_amd._sampling import BatchSampler

worst = 0.0
for rate in (1e9, 2e9, 2.4e9, 5e9):
    for duty30 in (False, True):
        n = 100000
        chans = [wl.awg_channel(wf, c, n, rate, duty30) for c in range(3)]
        grid = wl.awg_grid(n, rate)
        bs = BatchSampler(chans, grid)
        ref = c_oracle.eval_grid(bs.prog, bs.grid)
        got = bs.to_host(np.float64)
        e64 = np.abs(got - ref).max()
        got32 = bs.to_host(np.float32)
        e32 = np.abs(got32 - ref).max()
        worst = max(worst, e64)
        i = bs.plan.info
        print(f'{rate / 1e9:.1f} GS/s duty30={duty30}: {bs.plan.kernel_name()} units={i.n_tiles} '
              f'direct={i.n_direct} generic={i.n_generic} max|d| f64 {e64:.3g} f32 {e32:.3g} peak {np.abs(ref).max():.3g}')
        bs.close()
print('worst fp64', worst)
assert worst <= 1e-10   # (grid jitter: ulp(1e-4 s) x 1.3e9 rad/s; the budget is 1e-9)
