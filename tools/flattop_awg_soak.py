"""Random flat-top pulse trains at AWG sample rates -- square(width, edge) under 0 / 1 / several carriers, real and complex
amplitudes, pulses on the sample grid or at arbitrary times, near t = 0 and up to 1 ms from it, stacked with Gaussian
pulses, shifted, clipped: the short tier's sampled-edge own-term op and its closing-op form against the C
oracle, in fp64 (real + complex) and fp32.     python tools/flattop_awg_soak.py [n_cases] [seed0]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from cases import FP64_GRID_TOL, FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = own = 0
worst = worst32 = 0.0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    rate = float(rng.choice([1e9, 2e9, 2.4e9, 5e9]))
    n = int(rng.integers(2000, 60000))
    t0 = float(rng.choice([0.0, 0.0, 0.0, -3e-6, 1e-5, 1e-4, 1e-3]))
    grid = ('arange', t0, t0 + n / rate, 1.0 / rate)
    span = n / rate
    aligned = bool(rng.integers(2))
    chans = []
    for c in range(int(rng.integers(1, 4))):
        w = wf.zero()
        npulse = int(rng.integers(5, 120))
        slot = span / npulse
        for k in range(npulse):
            width = slot * rng.uniform(0.3, 0.8)
            edge = float(rng.choice([0.0, 0.05, 0.1, 0.2])) * width
            at = t0 + (k + 0.5) * slot
            if aligned:
                at = round((at - t0) * rate) / rate + t0
                width = max(2, round(width * rate)) / rate
            kind = rng.integers(6)
            env = wf.square(width, edge=edge) if edge > 0 else wf.square(width)
            nt = int(rng.choice([0, 1, 1, 1, 2, 5]))
            car = None
            for _ in range(nt):
                tone = rng.uniform(0.1, 1) * wf.cos(2 * np.pi * rng.uniform(-3e8, 3e8), rng.uniform(0, 6))
                car = tone if car is None else car + tone
            p = env if car is None else env * car
            if kind == 0:
                p = p + 0.3 * (wf.gaussian(width * 0.6) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8)))
            elif kind == 1 and edge > 0:
                I, Q = wf.mixing(env, freq=rng.uniform(-2e8, 2e8), phase=rng.uniform(0, 6), DRAGScaling=rng.uniform(-3e-10, 3e-10))
                p = I - 0.4 * Q
            amp = rng.uniform(0.1, 1) if rng.integers(4) else complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
            w = w + ((amp * p) >> at)
        if rng.integers(5) == 0:
            w = w >> rng.uniform(-2, 2) / rate
        chans.append(w)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    cplx = bool(rng.integers(2))
    dt = np.complex128 if cplx else np.float64
    ref = c_oracle.eval_grid(prog, g, cplx)
    pk = max(1.0, float(np.abs(ref).max()))
    err = 0.0
    for form in (None, '1'):
        if form: os.environ['WFK_NO_SHORT_ERFTAB'] = form
        try:
            plan = _engine.Plan(prog, grid=g)
        finally:
            os.environ.pop('WFK_NO_SHORT_ERFTAB', None)
        name = plan.kernel_name(dt)
        if form is None: own += name.startswith('wfk_sample_short<') and name.endswith(',2>')
        got = plan.run_host(dt)
        err = max(err, float(np.max(np.abs(got - ref))) / pk)
        if form is None:
            e32 = float(np.max(np.abs(plan.run_host(np.complex64 if cplx else np.float32) - ref))) / pk
            worst32 = max(worst32, e32)
            if not e32 <= FP32_TOL:
                fails += 1
                print(f'FAIL32 case {seed0 + case}: err {e32:.3e}  {name}', flush=True)
        plan.close()
    worst = max(worst, err)
    if not err <= FP64_GRID_TOL:
        fails += 1
        print(f'FAIL case {seed0 + case}: err {err:.3e} of peak {pk:.3g}  rate {rate:g} n {n} t0 {t0:g} aligned {aligned}  {name}', flush=True)
    if case % 100 == 99:
        print(f'.. {case + 1} cases, {own} on family 2, worst {worst:.2e} (float {worst32:.2e}), {fails} failures', flush=True)
print(f'{n_cases} cases (seeds {seed0}..{seed0 + n_cases - 1}): {own} with sampled-edge ops (family 2), worst error {worst:.2e} of peak (float {worst32:.2e}), {fails} failures')
sys.exit(1 if fails else 0)
