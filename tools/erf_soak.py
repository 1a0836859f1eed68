"""Differential soak of the fused erf edges (DESIGN 3.2 "Erf edges"): random flat-top scripts --
square(width, edge) alone, under 1..12 tones, times Gaussians, through mixing() with DRAG, summed
with neighbours, stacked, clipped, shifted, with complex amplitudes -- on grids whose lane stride
64*dt/sigma spans both sides of the admission limit.  HIP (fp64, fp32) vs the C oracle (libm erf
per sample), and vs the same plan with WFK_DISABLE_ERFMUL=1.
usage: python tools/erf_soak.py [first_seed] [count]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from cases import FP32_TOL
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def tones(rng, nt, cplx):
    out = None
    for _ in range(nt):
        t = rng.uniform(0.05, 0.4) * wf.cos(2 * np.pi * rng.uniform(-400e6, 400e6), rng.uniform(0, 6))
        if cplx and rng.random() < 0.5:
            t = t * complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        out = t if out is None else out + t
    return out


def script(rng):
    origin = 0.0 if rng.random() < 0.8 else rng.choice([1e-4, 1e-3, -3e-4])
    edge = rng.uniform(2e-9, 12e-9)
    npulse = int(rng.integers(1, 6))
    cplx = rng.random() < 0.15
    span = 0.0
    w = wf.zero()
    for k in range(npulse):
        width = rng.uniform(0.5, 8.0) * edge if rng.random() < 0.2 else rng.uniform(3, 10) * edge
        e = edge * (1.0 if rng.random() < 0.7 else rng.uniform(0.5, 1.5))
        centre = span + width / 2 + 2 * e + rng.uniform(0, 3 * e)
        env = wf.square(width, edge=e, type='erf' if rng.random() < 0.9 else 'cos')
        kind = rng.integers(0, 6)
        if kind == 0:
            p = rng.uniform(0.2, 1.0) * env
        elif kind == 1:
            p = env * tones(rng, int(rng.integers(1, 13)), cplx)
        elif kind == 2:
            I, Q = wf.mixing(env, freq=rng.uniform(-300e6, 300e6), phase=rng.uniform(0, 6),
                             DRAGScaling=rng.uniform(-4e-10, 4e-10))
            p = I if rng.random() < 0.5 else I + rng.uniform(-1, 1) * Q
        elif kind == 3:
            p = env * wf.gaussian(rng.uniform(2, 6) * width) * tones(rng, int(rng.integers(1, 4)), cplx)
        elif kind == 4:
            p = env * env if rng.random() < 0.3 else env * (wf.square(width * 0.6, edge=e * 0.7))
        else:
            p = env * tones(rng, 2, cplx) + 0.3 * (wf.gaussian(3 * e) >> rng.uniform(-width / 2, width / 2))
        w = w + (p >> (origin + centre))
        span = centre + width / 2 + (0 if rng.random() < 0.15 else 2 * e)       # sometimes overlapping the next
    if rng.random() < 0.2:
        w = w + rng.uniform(-0.2, 0.2)
    chans = [w]
    if rng.random() < 0.25:
        other = (wf.gaussian(rng.uniform(10e-9, 40e-9)) >> (origin + rng.uniform(0, span))) * wf.cos(2 * np.pi * 91e6)
        vs = wf.WaveVStack([w, other]) + rng.uniform(-0.1, 0.1)
        if rng.random() < 0.5:
            vs = vs >> rng.uniform(-3e-9, 3e-9)
        chans = [vs]
    elif rng.random() < 0.2 and not cplx:
        w.min, w.max = sorted(rng.uniform(-0.6, 0.9, 2))
    sigma = edge / 5
    h = 10 ** rng.uniform(-2.5, -0.7)                                  # 0.003 .. 0.2 around the 0.09 limit
    step = h * sigma / 64
    n = int(min(2_000_000, max(2000, (span + 4 * edge) / step)))
    t0 = origin - 2 * edge
    grid = ('linspace', t0, t0 + n * step, n, bool(rng.random() < 0.3))
    return chans, grid, cplx


bad, t0 = [], time.time()
nfused = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(77_000 + seed)
    try:
        chans, grid, cplx = script(rng)
        prog = _flatten.flatten(chans)
        g = _flatten.grid_from_desc(grid)
        ora = c_oracle.eval_grid(prog, g, True) if cplx else c_oracle.eval_grid(prog, g)
        pk = max(1.0, float(np.max(np.abs(ora))))
        far = abs(grid[1]) > 1e-5
        tol = 1e-9 if far else 5e-12
        plan = _engine.Plan(prog, grid=g)
        nfused += plan.info.n_generic == 0
        got = plan.run_host(np.complex128 if cplx else np.float64)
        e64 = float(np.max(np.abs(got - ora)))
        e32 = float(np.max(np.abs(plan.run_host(np.complex64 if cplx else np.float32) - ora)))
        os.environ['WFK_DISABLE_ERFMUL'] = '1'
        try:
            eab = float(np.max(np.abs(_engine.Plan(prog, grid=g).run_host(np.complex128 if cplx else np.float64) - got)))
        finally:
            del os.environ['WFK_DISABLE_ERFMUL']
        if not (e64 <= tol * pk and e32 <= FP32_TOL * pk and eab <= 2 * tol * pk) or not np.all(np.isfinite(got)):
            bad.append((seed, e64 / pk, e32 / pk, eab / pk, plan.kernel_name()))
            print('FAIL', bad[-1], flush=True)
    except Exception as e:
        bad.append((seed, repr(e)))
        print('ERROR', bad[-1], flush=True)
    if (seed - first) % 100 == 99:
        print(f'{seed - first + 1} scripts ({nfused} without generic terms), {len(bad)} failures, {time.time() - t0:.0f} s', flush=True)
print('done', count, 'scripts;', nfused, 'fully fused;', len(bad), 'failures', bad[:10])
