"""Every constructor with random parameters.  Part 1 (build container, CPU): the NumPy and C
oracles against the REAL reference.  Part 2 (`gpu`): the HIP path (grid mode, tlist mode, fp32)
against the oracle.   usage: python tools/prims_soak.py [count] [gpu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import waveforms_amd as ours
from oracle import np_oracle, c_oracle
from waveforms_amd import _flatten
from cases import FP32_TOL, FP32_FAR_TOL, FP64_GRID_TOL, FP64_TLIST_FUSED_TOL

count = int(sys.argv[1]) if len(sys.argv) > 1 else 500
gpu = len(sys.argv) > 2 and sys.argv[2] in ('gpu', 'gpufar')
far = len(sys.argv) > 2 and sys.argv[2] == 'gpufar'     # the scripts and their grids moved 10 us .. 10 ms from t = 0
ref = None
if not gpu:
    import make_golden
    ref = make_golden.import_reference()
    import waveforms.waveform as refw


def build(ns, nsw, rng):
    s = 10.0**rng.uniform(-9, -3)
    k = int(rng.integers(0, 16))
    w = s * rng.uniform(1, 20)
    win = ns.square(4 * w)                       # window that keeps unbounded primitives finite
    if k == 0: p = ns.chirp(rng.uniform(0.1, 2) / s, rng.uniform(0.1, 2) / s, w, phi0=rng.uniform(0, 6), type='linear') 
    elif k == 1: p = ns.chirp(rng.uniform(0.1, 2) / s, rng.uniform(0.1, 2) / s, w, phi0=rng.uniform(0, 6), type='exponential')
    elif k == 2: p = ns.chirp(rng.uniform(0.1, 2) / s, rng.uniform(0.1, 2) / s, w, phi0=rng.uniform(0, 6), type='hyperbolic')
    elif k == 3: p = ns.sinc(rng.uniform(0.1, 3) / s) * win
    elif k == 4: p = ns.exp(rng.uniform(-0.3, 0.3) / s) * win
    elif k == 5: p = ns.cosh(rng.uniform(0.05, 0.3) / s) * win
    elif k == 6: p = ns.sinh(rng.uniform(0.05, 0.3) / s) * win
    elif k == 7: p = ns.coshPulse(w, eps=rng.uniform(0.3, 4), plateau=s * rng.uniform(0, 5) * (rng.random() < 0.5))
    elif k == 8: p = ns.mollifier(w, plateau=s * rng.uniform(0, 5) * (rng.random() < 0.5), d=int(rng.integers(0, 4)))
    elif k == 9: p = ns.hanning(w, plateau=s * rng.uniform(0, 5) * (rng.random() < 0.5))
    elif k == 10: p = ns.general_cosine(w, *rng.uniform(-1, 1, size=int(rng.integers(1, 5))))
    elif k == 11:
        x = np.sort(rng.uniform(-2 * w, 2 * w, size=int(rng.integers(2, 30))))
        p = ns.interp(x, rng.normal(size=len(x)))
    elif k == 12: p = ns.samplingPoints(-w, w, rng.normal(size=int(rng.integers(2, 200))))
    elif k == 13: p = ns.sign() * win
    elif k == 14: p = ns.step(w * 0.3, type=str(rng.choice(['erf', 'cos', 'linear']))) * win
    else: p = nsw.slepian(w, *rng.uniform(0, 1, size=int(rng.integers(1, 4)))) if hasattr(nsw, 'slepian') else ns.gaussian(w)
    if rng.random() < 0.5:
        p = p * ns.cos(rng.uniform(0.5, 5) / s, rng.uniform(0, 6))
    p = rng.uniform(0.2, 2) * (p >> (s * rng.uniform(-3, 3)))
    cz = rng.random()
    if cz < 0.15:
        p = p * complex(rng.uniform(-1, 1), rng.uniform(-1, 1))           # complex amplitudes
    elif cz < 0.3:
        p = p * ns.exp(1j * rng.uniform(0.5, 5) / s)                       # cos + 1j sin carrier
    elif cz < 0.4:
        p = p + 1j * (ns.gaussian(w) >> (s * rng.uniform(-1, 1)))
    n = int(rng.integers(10, 30000))
    a = -3 * w + s * rng.uniform(-1, 1)
    return p, ('linspace', a, a + 6 * w, n, bool(rng.random() < 0.5))


bad = []
for it in range(count):
    try:
        import waveforms_amd.waveform as oursw
        w, grid = build(ours, oursw, np.random.default_rng(88_000 + it))
        if far:
            T_ = float(np.random.default_rng(99_000 + it).choice([1e-5, 1e-4, 1e-3, 3e-3, -2e-3, 1e-2]))
            w = w >> T_
            grid = (grid[0], grid[1] + T_, grid[2] + T_) + tuple(grid[3:])
        g = _flatten.grid_from_desc(grid)
        t = c_oracle.grid_values(g)
        want = np.asarray(np_oracle.call(w, t))
        pk = max(1.0, float(np.abs(want).max()))
        if gpu:
            from waveforms_amd import _engine
            plan = _engine.Plan(_flatten.flatten([w]), grid=g)
            cplx = np.iscomplexobj(want)
            e1 = float(np.max(np.abs(plan.run_host(np.complex128 if cplx else np.float64)[0] - want))) / pk
            got = w(t)
            if got.dtype != want.dtype:
                bad.append((it, 'dtype', got.dtype, want.dtype)); print('FAIL', bad[-1], flush=True)
            e2 = float(np.max(np.abs(got - want))) / pk
            e3 = float(np.max(np.abs(plan.run_host(np.complex64 if cplx else np.float32)[0] - want))) / pk
            # the same times as an explicit list (time-list tier: pointwise fused ops / device libm)
            ptl = _engine.Plan(_flatten.flatten([w]), t=t)
            e4 = float(np.max(np.abs(ptl.run_host(np.complex128 if cplx else np.float64)[0] - want))) / pk
            if not (e1 <= FP64_GRID_TOL and e2 <= FP64_GRID_TOL and e3 <= (FP32_FAR_TOL if far else FP32_TOL) and e4 <= FP64_TLIST_FUSED_TOL):
                bad.append((it, e1, e2, e3, e4)); print('FAIL', bad[-1], flush=True)
        else:
            wr, _ = build(ref, refw, np.random.default_rng(88_000 + it))
            got = np.asarray(wr(t))
            if got.dtype != want.dtype:
                bad.append((it, 'dtype', got.dtype, want.dtype)); print('FAIL', bad[-1], flush=True)
            e1 = float(np.max(np.abs(got - want))) / pk
            e2 = float(np.max(np.abs(c_oracle.eval_grid(_flatten.flatten([w]), g, np.iscomplexobj(got))[0] - got))) / pk
            if not (e1 <= 1e-12 and e2 <= 1e-10):
                bad.append((it, e1, e2)); print('FAIL', bad[-1], flush=True)
    except NotImplementedError as ex:
        pass
    except Exception as ex:
        bad.append((it, repr(ex))); print('ERROR', bad[-1], flush=True)
print('done', count, 'scripts;', len(bad), 'failures', bad[:8], '(GPU vs oracle)' if gpu else '(oracles vs reference)')
