import sys; sys.path.insert(0, '.')
import numpy as np
from waveforms_amd import cosPulse, gaussian, mixing, WaveVStack
from waveforms_amd.distortion import predistort

pulse = gaussian(20e-9) >> 40e-9
I, Q = mixing(pulse, freq=120e6, phase=0.3, DRAGScaling=1e-10)
t = np.linspace(0, 100e-9, 2001)
y = I(t)
I.start, I.stop, I.sample_rate = 0, 100e-9, 2e9
y2 = I.sample()
z = predistort(y2, ker=np.ones(5) / 5)
import torch
from waveforms_amd._sampling import BatchSampler
from waveforms_amd.distortion import FirStage
bs = BatchSampler([I, Q, WaveVStack([I, Q])], ('linspace', 0.0, 100e-9, 10**6, False))
out = torch.empty((bs.n_channels, bs.n), dtype=torch.float64, device='cuda')
bs.launch_torch(out)
fir = FirStage(np.ones(1024) / 1024, bs.n, bs.n_channels)
o2 = torch.empty_like(out)
fir.apply_torch(out, o2)
torch.cuda.synchronize()
print(y.shape, y2.shape, z.shape, float(out.abs().max()), float(o2.abs().max()))
