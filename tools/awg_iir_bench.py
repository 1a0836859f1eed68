"""sample(filters=) at AWG sample rates: 2048 rows x 1e5 points at 2 GS/s (the also.awg rows) through butter(4, 0.1) as two
biquads -- which path SampledIir takes there (60-sample pieces: more than 16 per chunk of the fused scan) and what it costs.
    python tools/awg_iir_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
import waveforms_amd as wf
from waveforms_amd import workloads as wl
from waveforms_amd.distortion import SampledIir
from waveforms_amd._sampling import BatchSampler

sos = butter(4, 0.1, output='sos')
for name, mk in (('gaussian+drag, back to back', lambda c: wl.awg_channel(wf, c)), ('30 % duty', lambda c: wl.awg_channel(wf, c, duty30=True))):
    chans = [mk(c) for c in range(16)]
    si = SampledIir(chans, wl.awg_grid(), sos, tile=128)
    out = torch.empty((si.n_channels, si.n), dtype=torch.float64, device='cuda')
    for _ in range(2): si.launch_torch(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): si.launch_torch(out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f'{name:30s} {ms:7.3f} ms  {si.n_channels * si.n * 8 / ms * 1e-9 / 8 * 100:5.1f}% of 8 TB/s on 8 B/sample  fused {si.fused}  {si.plan.kernel_name()}  {si.why_not}', flush=True)
    si.close()
