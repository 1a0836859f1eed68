"""Random terms with powers on their factors (gaussian ** p, cos ** 2 | 3, exp ** p, products of them) near and far
from t = 0, on fine grids, AWG-rate grids and time lists, against the C oracle.   python tools/powers_soak.py [n] [seed0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cases import FP64_GRID_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = fused = 0
worst = 0.0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    t0 = float(rng.choice([0.0, 0.0, -2e-6, 5e-4, -3e-3]))
    awg = case % 3 == 2
    if awg:
        rate = float(rng.choice([1e9, 2e9, 5e9]))
        n = int(rng.integers(2000, 50000))
        span = n / rate
        grid = ('arange', t0, t0 + span, 1.0 / rate)
    else:
        n = int(rng.integers(5000, 300000))
        span = float(10 ** rng.uniform(-6.5, -5))
        grid = ('linspace', t0, t0 + span, n, bool(rng.integers(2)))
    chans = []
    for c in range(int(rng.integers(1, 4))):
        w = wf.zero()
        npulse = int(rng.integers(2, 40)) if awg else int(rng.integers(1, 6))
        for k in range(npulse):
            width = span / npulse * rng.uniform(0.3, 1.5)
            centre = t0 + (k + 0.5) * span / npulse
            p = float(rng.choice([2, 3, 0.5, 1.5, rng.uniform(0.3, 4)]))
            env = wf.gaussian(width) ** p if rng.integers(3) else wf.square(width)
            f = rng.uniform(-3, 3) * 10 / width
            car = wf.cos(2 * np.pi * f, rng.uniform(0, 6)) ** int(rng.choice([1, 2, 3]))
            if rng.integers(3) == 0:
                car = car * wf.cos(2 * np.pi * rng.uniform(-1, 1) * 5 / width) ** int(rng.choice([1, 2]))
            if rng.integers(4) == 0:
                env = env * (wf.exp(-rng.uniform(0.2, 3) / width) ** float(rng.choice([1, 2, 0.5])))
            amp = rng.uniform(0.1, 1) if rng.integers(4) else complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
            w = w + ((amp * env * car) >> centre)
        chans.append(w)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    cplx = bool(rng.integers(2))
    dt = np.complex128 if cplx else np.float64
    plan = _engine.Plan(prog, grid=g)
    fused += plan.info.n_generic == 0
    got = plan.run_host(dt)
    ref = c_oracle.eval_grid(prog, g, cplx)
    pk = max(1.0, float(np.abs(ref).max()))
    err = float(np.max(np.abs(got - ref))) / pk
    if case % 5 == 0:
        t = c_oracle.grid_values(g)[:20000]
        tl = _engine.Plan(prog, t=t).run_host(dt)
        err = max(err, float(np.max(np.abs(tl - c_oracle.eval_tlist(prog, t, want_complex=cplx)))) / pk)
    worst = max(worst, err)
    if not err <= FP64_GRID_TOL:
        fails += 1
        print(f'FAIL case {seed0 + case}: err {err:.3e} of peak {pk:.3g}  {grid[0]} n {n} t0 {t0:.3g}  {plan.kernel_name(dt)}', flush=True)
    if case % 100 == 99:
        print(f'.. {case + 1} cases, {fused} without generic terms, worst {worst:.2e}, {fails} failures', flush=True)
print(f'{n_cases} cases (seeds {seed0}..{seed0 + n_cases - 1}): {fused} fully fused, worst error {worst:.2e} of peak, {fails} failures')
sys.exit(1 if fails else 0)
