"""WaveVStack sums (stack + stack with equal / different pending shifts, + Waveform, + number, nested) against the REAL reference,
build container only (oracle/make_golden.import_reference): wlist, offset and shift element for element.   python tools/vstack_soak.py"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import make_golden
R = make_golden.import_reference()
import waveforms_amd as A
def build(ns, rng):
    mk = lambda: (ns.gaussian(float(rng.uniform(1, 3))) >> float(rng.uniform(-2, 2))) * float(rng.uniform(0.2, 1))
    s1 = ns.WaveVStack([mk(), mk()]) >> float(rng.choice([0.0, 0.5]))
    s2 = ns.WaveVStack([mk()]) >> float(rng.choice([0.0, 0.5, 1.25]))
    s1 = s1 + float(rng.uniform(-1, 1))
    outs = [s1 + s2, s1 + mk(), s1 + 0.25, (s1 + s2) + (s2 + 0.5), 0.5 + s1, (s1 >> 0.3) + (s2 >> 0.3), (s1 + mk()) + s2]
    return outs
bad = 0
for seed in range(300):
    a = build(A, np.random.default_rng(seed)); r = build(R, np.random.default_rng(seed))
    for x, y in zip(a, r):
        same = (x.shift == y.shift and x.offset == y.offset and len(x.wlist) == len(y.wlist)
                and all(tuple(p[0]) == tuple(q[0]) and p[1] == q[1] for p, q in zip(x.wlist, y.wlist)))
        if not same:
            bad += 1
print('mismatches', bad)
