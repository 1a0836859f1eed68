"""Channel blocks compiled on host threads (wfk_compile_blocks) against the plan compiled in one piece and against the C
oracle: random batches of 32-72 AWG-rate rows of mixed shapes -- Gaussian + DRAG trains, 30 % duty, samplingPoints envelopes
(pool tables that move with their block), flat tops (sampled edge tables), ten tones, chirps -- random thread counts.
    python tools/block_soak.py [n_cases] [seed0]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from cases import FP64_GRID_TOL, FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = blocks = 0
worst = worst_ab = 0.0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    rate = float(rng.choice([1e9, 2e9, 2.4e9]))
    n = int(rng.integers(9000, 20000))
    rows = int(rng.integers(32, 73))
    kinds = rng.choice(['awg', 'duty', 'interp', 'flat_top', 'ten_tones', 'linear_chirp', 'exp_chirp'], size=int(rng.integers(1, 4)), replace=False)
    chans = []
    for c in range(rows):
        k = kinds[c % len(kinds)]
        if k == 'awg': w = wl.awg_channel(wf, 1000 * case + c, n, rate)
        elif k == 'duty': w = wl.awg_channel(wf, 1000 * case + c, n, rate, duty30=True)
        elif k == 'interp': w = wl.awg_interp_channel(wf, 1000 * case + c, n, rate)
        else: w = wl.awg_shape_channel(wf, k, 1000 * case + c, n, rate)
        if rng.integers(4) == 0: w = w + float(rng.uniform(-0.3, 0.3))
        chans.append(w)
    grid = wl.awg_grid(n, rate)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)

    def plan(threads):
        os.environ['WFK_COMPILE_THREADS'] = str(threads)
        try:
            return _engine.Plan(prog, grid=g)
        finally:
            del os.environ['WFK_COMPILE_THREADS']
    th = int(rng.choice([2, 3, 4, 5, 8, 16]))
    one, many = plan(1), plan(th)
    blocks += prog.struct.n_pieces >= 8192 and rows >= 2 * th
    a, b = one.run_host(np.float64), many.run_host(np.float64)
    pick = sorted(set(int(v) for v in rng.integers(0, rows, size=6)) | {0, rows - 1})
    ref = c_oracle.eval_grid(_flatten.flatten([chans[i] for i in pick]), g)
    pk = max(1.0, float(np.abs(ref).max()))
    err = float(np.max(np.abs(b[pick] - ref))) / pk
    dab = float(np.max(np.abs(a - b))) / pk
    e32 = float(np.max(np.abs(many.run_host(np.float32)[pick] - ref))) / pk
    worst, worst_ab = max(worst, err), max(worst_ab, dab)
    same_name = one.kernel_name() == many.kernel_name()
    # (sampled flat-top edges: a block shares edge tables among its own channels only -- equal to the sharing bound)
    if not (err <= FP64_GRID_TOL and dab <= 1e-10 and e32 <= FP32_TOL and same_name):
        fails += 1
        print(f'FAIL case {seed0 + case}: vs oracle {err:.2e}, one piece vs blocks {dab:.2e}, float {e32:.2e}, {one.kernel_name()} / {many.kernel_name()}  rows {rows} threads {th} kinds {list(kinds)}', flush=True)
    one.close(); many.close()
    if case % 10 == 9:
        print(f'.. {case + 1} cases, {blocks} compiled as blocks, worst {worst:.2e} vs oracle, {worst_ab:.2e} one piece vs blocks, {fails} failures', flush=True)
print(f'{n_cases} cases (seeds {seed0}..{seed0 + n_cases - 1}): {blocks} compiled as blocks, worst {worst:.2e} of peak vs the oracle, {worst_ab:.2e} one piece vs blocks, {fails} failures')
sys.exit(1 if fails else 0)
