import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, waveforms_amd as wf, cases
from waveforms_amd import _flatten
from waveforms_amd.distortion import SampledFir
why={}
for seed in range(60):
    rng = np.random.default_rng(40_000 + seed)
    ch, grid = cases.random_awg_channel(wf, rng)
    prog = _flatten.flatten([ch])
    if prog.complex_amp: continue
    K=int(rng.choice([1, 7, 300, 1024, 1400]))
    ker = np.ones(K)/K
    sf = SampledFir([ch], grid, ker)
    g=_flatten.grid_from_desc(grid)
    why.setdefault(sf.why_not[:90],[]).append((seed,int(g.n),K))
    sf.close()
for k,v in why.items(): print(repr(k), len(v), v[:6])
