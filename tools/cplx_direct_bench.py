"""complex128 / complex64 output of a plan on the DIRECT tier (complex amplitudes over primitives without a fast path:
exponential chirps, erf on a coarse grid): the build of the general kernel that sits at its 256-register limit.
    python tools/cplx_direct_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import workloads as wl, _flatten
from waveforms_amd._sampling import BatchSampler
rng = np.random.default_rng(0)
ws = [((0.5 + 0.3j) * wf.chirp(1e8, 3e8, 1.2 * wl.W, type='exponential') * (wf.square(wl.SPAN, edge=4e-9) >> 0)) >> ((k + 0.5) * wl.SPAN) for k in range(100)]
w = wl._tree_sum(ws)
n = 2 * 10**6
grid = ('linspace', 0.0, 3e-6, n, False)
# `slice` as first argument: the same samples as the second half of a grid twice as long (wfk_grid.i0 != 0: the *_slice builds)
SLICE = len(sys.argv) > 1 and sys.argv[1] == 'slice'
if SLICE:
    full = _flatten.grid_from_desc(('linspace', -3e-6, 3e-6, 2 * n, False))
    grid = _flatten.grid_slice(full, n, 2 * n)
for dt, tdt in ((np.complex128, torch.complex128), (np.complex64, torch.complex64), (np.float64, torch.float64)):
    bs = BatchSampler([w] * 32, grid)
    out = torch.empty((32, n), dtype=tdt, device='cuda')
    for _ in range(2): bs.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): bs.launch_torch(out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    ref = c_oracle.eval_grid(_flatten.flatten([w]), grid if SLICE else _flatten.grid_from_desc(('linspace', 0.0, 3e-6, n, False)), True)[0][:200000]
    got = out[0, :200000].cpu().numpy()
    err = np.max(np.abs(got - (ref if np.iscomplexobj(got) else ref.real)))
    print(f'{np.dtype(dt).name}: {ms:.3f} ms  {bs.plan.kernel_name(dt)}  max err {err:.2e}  lib={os.environ.get("WFK_LIB", "tree")}', flush=True)
    bs.close()
