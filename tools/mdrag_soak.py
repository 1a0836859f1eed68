"""multy_drag (ids 16/17) random-parameter soak.  Part 1 (build container, CPU): the NumPy oracle
against the REAL reference.  Part 2 (GPU box, `gpu` argument): the HIP path against the oracle.
usage: python tools/mdrag_soak.py [count] [gpu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import waveforms_amd as ours
from oracle import np_oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 500
gpu = len(sys.argv) > 2 and sys.argv[2] == 'gpu'
ref = None
if not gpu:
    import make_golden
    ref = make_golden.import_reference()


def build(ns, rng):
    scale = 10.0**rng.uniform(-9, -5)
    width = scale * rng.uniform(5, 60)
    plateau = 0.0 if rng.random() < 0.5 else scale * rng.uniform(1, 40)
    freq = rng.uniform(-0.3, 0.3) / scale
    delta = 0.0 if rng.random() < 0.4 else rng.uniform(-0.02, 0.02) / scale
    nb = int(rng.integers(0, 4))
    bf = None if nb == 0 else tuple(float(x) for x in rng.uniform(0.05, 0.6, size=nb) / scale * rng.choice([-1, 1], size=nb))
    if nb == 1 and rng.random() < 0.5:
        bf = bf[0]
    phase, t0 = rng.uniform(0, 6), scale * rng.uniform(-20, 20)
    if rng.random() < 0.5:
        w = ns.drag_sin(freq, width, plateau=plateau, delta=delta, block_freq=bf, phase=phase, t0=t0)
    else:
        w = ns.drag_sinx(freq, width, plateau=plateau, delta=delta, block_freq=bf, phase=phase, t0=t0,
                         tab=rng.uniform(0.2, 0.9))
    n = int(rng.integers(50, 20000))
    t = np.linspace(t0 - 0.1 * width, t0 + width + plateau + 0.1 * width, n)
    return w * rng.uniform(0.2, 1.5), t


bad = []
for it in range(count):
    try:
        w, t = build(ours, np.random.default_rng(66_000 + it))
        want = np.real(np_oracle.call(w, t))
        pk = max(1.0, float(np.abs(want).max()))
        if gpu:
            got = np.real(w(t))
            tol = 1e-9
        else:
            wr, tr = build(ref, np.random.default_rng(66_000 + it))
            got = np.real(wr(tr))
            tol = 1e-12
        e = float(np.max(np.abs(got - want))) / pk
        if not e <= tol:
            bad.append((it, e)); print('FAIL', bad[-1], flush=True)
    except Exception as ex:
        bad.append((it, repr(ex))); print('ERROR', bad[-1], flush=True)
print('done', count, 'pulses;', len(bad), 'failures', bad[:8], '(GPU vs oracle)' if gpu else '(oracle vs reference)')
