import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd.distortion import SampledIir
from scipy.signal import butter
chans=[wl.sum_channel(wf,100,1000+c) for c in range(4)]
si=SampledIir(chans, wl.c2_grid(10**7), butter(4,0.1,output='sos'))
print(si.fused, si.plan.kernel_name())
y=si.to_host()
print(y.shape)
