import sys, os; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, waveforms_amd as wf
from waveforms_amd import _flatten, workloads as wl
from waveforms_amd.distortion import SampledFir
from waveforms_amd._sampling import BatchSampler
from oracle import c_oracle
n, rate = 50001, 2e9
chans = [(wl.awg_channel(wf, 3 + c, n, rate, c == 1) + 0.125 * c) >> (c * 0.3e-9) for c in range(2)]
grid = wl.awg_grid(n, rate)
ker = np.array([1.0])
y = c_oracle.eval_grid(_flatten.flatten(chans), _flatten.grid_from_desc(grid))
sf = SampledFir(chans, grid, ker); got = sf.to_host(); print(sf.plan.kernel_name())
e = np.abs(got - y); print('fused vs oracle', e.max(), np.argmax(e, axis=1), np.abs(y).max())
smp = BatchSampler(chans, grid).to_host(np.float64); print('sampler vs oracle', np.abs(smp - y).max())
os.environ['WFK_CHAIN_UNFUSED'] = '1'
su = SampledFir(chans, grid, ker); gu = su.to_host(); print(su.plan.kernel_name())
print('unfused vs oracle', np.abs(gu - y).max(), 'fused vs unfused', np.abs(gu - got).max())
