"""Random chirp trains at AWG sample rates -- linear, exponential and hyperbolic chirps under cosine pulses, squares,
Gaussians, with and without an extra carrier, real and complex amplitudes, near t = 0 and away from it, among plain
Gaussian pulses: the short tier's chirp op (family 1), its exponential / hyperbolic closing multipliers (family 4) and
whatever falls back to the pointwise tier, against the C oracle in fp64 (real + complex) and fp32.
    python tools/chirp_awg_soak.py [n_cases] [seed0]"""
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
fails = 0
worst = worst32 = 0.0
tiers = {}
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    rate = float(rng.choice([1e9, 2e9, 2.4e9, 5e9]))
    n = int(rng.integers(2000, 40000))
    t0 = float(rng.choice([0.0, 0.0, -2e-6, 1e-5, 3e-4, 1e-3, 3e-3, -2e-3, 1e-2]))
    grid = ('arange', t0, t0 + n / rate, 1.0 / rate)
    span = n / rate
    chans = []
    for c in range(int(rng.integers(1, 3))):
        w = wf.zero()
        npulse = int(rng.integers(4, 80))
        slot = span / npulse
        for k in range(npulse):
            width = slot * rng.uniform(0.4, 0.95)
            at = t0 + (k + 0.5) * slot
            kind = str(rng.choice(['linear', 'exponential', 'hyperbolic', 'none']))
            f0 = rng.uniform(2e7, 1.5e8) * float(rng.choice([1, 1, -1]))
            f1 = f0 * rng.uniform(1.2, 3.0)
            if kind == 'none':
                p = wf.gaussian(width) * wf.cos(2 * np.pi * f0, rng.uniform(0, 6))
            else:
                ch = wf.chirp(abs(f0), abs(f1), width, rng.uniform(0, 6), kind) if kind != 'linear' else wf.chirp(f0, f1, width, rng.uniform(0, 6))
                env = int(rng.integers(4))
                if env == 0: p = ch * wf.cosPulse(width)
                elif env == 1: p = ch * wf.square(width * 0.9)
                elif env == 2: p = ch * wf.gaussian(width * 0.8)
                else: p = ch * wf.cosPulse(width) * wf.cos(2 * np.pi * rng.uniform(-1e8, 1e8))
            amp = rng.uniform(0.1, 1) if rng.integers(4) else complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
            w = w + ((amp * p) >> at)
        chans.append(w)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    cplx = bool(rng.integers(2))
    dt = np.complex128 if cplx else np.float64
    plan = _engine.Plan(prog, grid=g)
    name = plan.kernel_name(dt)
    tiers[name.split('<')[0] + (name[name.rfind(','):] if name.startswith('wfk_sample_short<') else '')] = tiers.get(name.split('<')[0] + (name[name.rfind(','):] if name.startswith('wfk_sample_short<') else ''), 0) + 1
    ref = c_oracle.eval_grid(prog, g, cplx)
    pk = max(1.0, float(np.abs(ref).max()))
    err = float(np.max(np.abs(plan.run_host(dt) - ref))) / pk
    e32 = float(np.max(np.abs(plan.run_host(np.complex64 if cplx else np.float32) - ref))) / pk
    plan.close()
    worst, worst32 = max(worst, err), max(worst32, e32)
    if not (err <= FP64_GRID_TOL and e32 <= FP32_TOL):
        fails += 1
        print(f'FAIL case {seed0 + case}: err {err:.3e} float {e32:.3e} of peak {pk:.3g}  rate {rate:g} n {n} t0 {t0:g}  {name}', flush=True)
    if case % 100 == 99:
        print(f'.. {case + 1} cases, worst {worst:.2e} (float {worst32:.2e}), {fails} failures, tiers {tiers}', flush=True)
print(f'{n_cases} cases (seeds {seed0}..{seed0 + n_cases - 1}): worst error {worst:.2e} of peak (float {worst32:.2e}), {fails} failures; tiers {tiers}')
sys.exit(1 if fails else 0)
