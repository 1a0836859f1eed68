import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from scipy.signal import butter
from waveforms_amd import _engine
n, batch = 10**7, 64
sec=[(r[:3], r[3:]) for r in butter(4, 0.03, output='sos')]
plan=_engine.IirPlan(sec, n, batch, np.float64)
x=torch.randn((batch,n),dtype=torch.float64,device='cuda'); y=torch.empty_like(x)
for _ in range(2): plan.apply(x.data_ptr(), n, y.data_ptr(), n)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): plan.apply(x.data_ptr(), n, y.data_ptr(), n)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
print('IIR 2 biquads fp64: %d x %g: %.2f ms, %.1f Gsamples/s, %.2f TB/s algorithmic(16B)'%(batch,n,dt*1e3,batch*n/dt/1e9,batch*n*16/dt/1e12))
