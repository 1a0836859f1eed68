"""Pulse shapes at an AWG sample rate (2048 rows x 1e5 points at 2 GS/s, 60-sample pulses back to back, 16 distinct rows
x 128): which tier takes them and what it costs.    python tools/awg_shapes_bench.py [shape ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd._sampling import BatchSampler
W = wl.SPAN
SHAPES = {
    'gaussian+drag': lambda rng: wf.mixing(wf.gaussian(20e-9), freq=rng.uniform(-2e8, 2e8), phase=rng.uniform(0, 6), DRAGScaling=1e-10)[0],
    'flat top (erf)': lambda rng: wf.square(0.6 * W, edge=0.1 * W) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6)),
    'linear chirp': lambda rng: wf.chirp(rng.uniform(5e7, 1e8), rng.uniform(1.5e8, 3e8), W) * wf.cosPulse(W),
    'exp chirp': lambda rng: wf.chirp(rng.uniform(5e7, 1e8), rng.uniform(1.5e8, 3e8), W, type='exponential') * wf.cosPulse(W),
    'sinc': lambda rng: wf.sinc(6 / W) * wf.square(W) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8)),
    'mollifier d1 (drag)': lambda rng: wf.mixing(wf.mollifier(W), freq=rng.uniform(-2e8, 2e8), DRAGScaling=1e-10)[0],
    'gaussian**2': lambda rng: (wf.gaussian(20e-9) ** 2) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8)),
    'ten tones': lambda rng: wf.gaussian(20e-9) * sum((rng.uniform(0.05, 0.2) * wf.cos(2 * np.pi * rng.uniform(-3e8, 3e8), rng.uniform(0, 6)) for _ in range(9)),
                                                   0.1 * wf.cos(2 * np.pi * 1e8)),
}
def _mixed(rng):      # one pulse in ten of a libm shape among Gaussian pulses: a mixed short plan
    if rng.integers(10) == 0:
        return wf.chirp(rng.uniform(5e7, 1e8), rng.uniform(1.5e8, 3e8), W, type='exponential') * wf.cosPulse(W)
    return wf.mixing(wf.gaussian(20e-9), freq=rng.uniform(-2e8, 2e8), phase=rng.uniform(0, 6), DRAGScaling=1e-10)[0]


SHAPES['1 in 10 exp chirp'] = _mixed
SHAPES['far (t0 = 1 ms)'] = SHAPES['gaussian+drag']
SHAPES['two overlapping trains'] = lambda rng: wf.gaussian(20e-9) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8)) + ((wf.gaussian(20e-9) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8))) >> (W / 2))
names = sys.argv[1:] or list(SHAPES)
n_pts, rate = 100_000, 2e9
nseg = int(n_pts / rate / W)
for name in names:
    chans = []
    for c in range(16):
        rng = np.random.default_rng(900 + c)
        t00 = 1e-3 if name.startswith('far') else 0.0
        chans.append(wl._tree_sum([rng.uniform(0.2, 1) * SHAPES[name](rng) >> (t00 + (k + 0.5) * W) for k in range(nseg)]))
    grid = ('arange', 1e-3, 1e-3 + n_pts / rate, 1.0 / rate) if name.startswith('far') else wl.awg_grid(n_pts, rate)
    bs = BatchSampler(chans, grid, tile=128)
    out = torch.empty((bs.n_channels, bs.n), dtype=torch.float64, device='cuda')
    for _ in range(2): bs.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): bs.launch_torch(out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    i = bs.plan.info
    print(f'{name:22s} {ms:8.3f} ms  {bs.n_channels * bs.n * 8 / ms * 1e-9 / 8 * 100:5.1f}% of 8 TB/s  {bs.plan.kernel_name()}  fused {i.n_fused} generic {i.n_generic}', flush=True)
    bs.close()
