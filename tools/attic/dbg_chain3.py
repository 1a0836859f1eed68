import sys, os; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, waveforms_amd as wf, cases
from oracle import c_oracle
from waveforms_amd import _flatten
from waveforms_amd.distortion import SampledFir
seed=29703
rng = np.random.default_rng(70_000 + seed)
ch, grid = cases.random_awg_channel(wf, rng) if seed % 4 else cases.random_channel(wf, rng)
nch = int(rng.integers(1, 4))
chans = [ch] + [(ch * float(rng.uniform(0.2, 1.5)) + float(rng.uniform(-0.3, 0.3))) for _ in range(nch - 1)]
prog = _flatten.flatten(chans); g = _flatten.grid_from_desc(grid)
K = int(rng.choice([1, 2, 9, 128, 1000, 1024, 1025, 1500, 1537, 3000]))
ker = rng.normal(size=K); ker /= np.abs(ker).sum()
y = c_oracle.eval_grid(prog, g); want = np.stack([c_oracle.fir(r, ker) for r in y])
print('n', g.n, 'K', K, 'rows', len(chans), 'peak y', np.abs(y).max(), 'peak z', np.abs(want).max())
for env in ('0','1'):
    os.environ['WFK_CHAIN_UNFUSED']=env
    for dt in (np.float64, np.float32):
        sf = SampledFir(chans, grid, ker, dt); got=sf.to_host(); e=np.abs(got-want)
        print('unfused' if env=='1' else 'fused', dt.__name__, sf.plan.kernel_name()[:60], 'max err', e.max(), 'at', np.unravel_index(e.argmax(), e.shape), 'want there', want[np.unravel_index(e.argmax(), e.shape)])
        sf.close()
# float32 sampler alone
from waveforms_amd._sampling import BatchSampler
bs=BatchSampler(chans, grid); s32=bs.to_host(np.float32).astype(np.float64); print('sampler f32 max err', np.abs(s32-y).max())
