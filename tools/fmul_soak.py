"""Random INTERP tables / mollifiers under random carriers on random grids: the lean kernel's stateless closing
multipliers (family 3) against the C oracle.     python tools/fmul_soak.py [n_cases] [seed0]
Prints one line per failure and a summary; exit code 1 on any failure."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cases import FP64_GRID_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten
from waveforms_amd.waveform import _window, primitive, INTERP

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = fam3 = 0
worst = 0.0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.integers(2000, 400000))
    t0 = float(rng.choice([0.0, 0.0, -1e-6, 3e-4, -2e-3])) + rng.uniform(-1e-7, 1e-7)
    span = float(10 ** rng.uniform(-6.5, -4.5))
    endpoint = bool(rng.integers(2))
    grid = ('linspace', t0, t0 + span, n, endpoint)
    awg = case % 3 == 2          # every third case at an AWG sample rate: trains of short pulses (the short tier)
    if awg:
        rate = float(rng.choice([1e9, 2e9, 5e9]))
        n = int(rng.integers(2000, 60000))
        span = n / rate
        grid = ('arange', t0, t0 + span, 1.0 / rate)
    chans = []
    for c in range(int(rng.integers(1, 4))):
        w = wf.zero()
        npulse = int(rng.integers(20, 200)) if awg else int(rng.integers(1, 6))
        for k in range(npulse):
            width = span * 10 ** rng.uniform(-2.0, -0.3)
            centre = t0 + rng.uniform(0.0, 1.0) * span
            if awg:
                width = rng.uniform(8, 120) / rate
                if case % 2:
                    width = min(width, 0.95 * span / npulse)      # back to back at most: no two envelopes in one piece
                centre = t0 + (k + rng.uniform(0.3, 0.7)) * span / npulse
            kind = rng.integers(4)
            if kind <= 1:
                m = int(rng.choice([2, 3, 5, 17, 100, 1000, 5000]))
                pts = rng.normal(size=m) if kind == 0 else np.hanning(m) * rng.uniform(0.2, 2)
                if rng.integers(3) == 0:
                    a, b = sorted(rng.uniform(-0.5, 0.5, size=2) * width)      # window and table do not coincide
                    if b - a < 1e-3 * width:
                        b = a + 1e-3 * width
                    env = _window(-width / 2, width / 2, primitive(INTERP, float(a), float(b), tuple(pts)))
                else:
                    env = wf.samplingPoints(-width / 2, width / 2, pts)
            elif kind == 2:
                env = wf.mollifier(width)
            else:
                env = wf.mollifier(width / 2, plateau=width / 2)
            nc = int(rng.integers(0, 4))
            car = None
            for _ in range(nc):
                f = rng.uniform(-3, 3) * 30 / width
                tone = rng.uniform(0.1, 1) * wf.cos(2 * np.pi * f, rng.uniform(0, 6))
                car = tone if car is None else car + tone
            p = env if car is None else env * car
            amp = rng.uniform(0.1, 1) if rng.integers(4) else complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
            w = w + ((amp * p) >> centre)
        if rng.integers(4) == 0:
            w = w + (rng.uniform(0.1, 1) * wf.gaussian(span * 0.05) >> (t0 + 0.5 * span))
        chans.append(w)
    prog = _flatten.flatten(chans)
    g = _flatten.grid_from_desc(grid)
    cplx = bool(rng.integers(2))
    plan = _engine.Plan(prog, grid=g)
    name = plan.kernel_name(np.complex128 if cplx else np.float64)
    fam3 += ',3>' in name or ',4>' in name or (name.startswith('wfk_sample_short<') and plan.info.n_generic == 0)
    got = plan.run_host(np.complex128 if cplx else np.float64)
    ref = c_oracle.eval_grid(prog, g, cplx)
    pk = max(1.0, float(np.abs(ref).max()))
    err = float(np.max(np.abs(got - ref))) / pk
    worst = max(worst, err)
    if not err <= FP64_GRID_TOL:
        fails += 1
        print(f'FAIL case {seed0 + case}: err {err:.3e} of peak {pk:.3g}  n {n} span {span:.3g} t0 {t0:.3g}  {name}', flush=True)
    if case % 50 == 49:
        print(f'.. {case + 1} cases, {fam3} on family 3, worst {worst:.2e}, {fails} failures', flush=True)
print(f'{n_cases} cases (seeds {seed0}..{seed0 + n_cases - 1}): {fam3} with a family-3 lean launch or a pure short-tier plan, worst error {worst:.2e} of peak, {fails} failures')
sys.exit(1 if fails else 0)
