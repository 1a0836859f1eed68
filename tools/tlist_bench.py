"""tlist mode (arbitrary sorted sample times: Waveform.__call__(x) with a non-grid x).
    python tools/tlist_bench.py [headline|flattop|multitone|readme] [nch] [n]
headline: 100 gaussian+DRAG pulses per channel (every term fuses: pointwise ops only);
flattop : square(width, edge) pulses under carriers (erf edges stay generic: the build with the direct tier);
multitone: 10 tones under every gaussian pulse; readme: the README sequence tiled.
Prints kernel time, Gsamples/s, fraction of the 16 B/sample roof and max |err| against the C oracle on a slice.
A/B: WFK_DISABLE_TLFUSE=1 (every factor on device libm), WFK_LIB=_ab/libwfk_base.so (the round-3 kernels)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl
shape = sys.argv[1] if len(sys.argv) > 1 else 'headline'
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 2 * 10**6
t = wl.jittered_times(n)
rng = np.random.default_rng(1)
if shape == 'headline':
    chans = [wl.sum_channel(wf, 100, 1000 + c) for c in range(nch)]
elif shape == 'multitone':
    chans = [wl.multitone_channel(wf, c) for c in range(min(nch, 8))] * max(1, nch // 8)
elif shape == 'flattop':
    chans = []
    for c in range(min(nch, 8)):
        ws = [(wf.square(40e-9, edge=8e-9) >> ((k + 0.5) * 60e-9)) * wf.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6)) * rng.uniform(0.2, 1)
              for k in range(50)]
        chans.append(wl._tree_sum(ws))
    chans = chans * max(1, nch // 8)
else:
    x, y = wl.readme_xy(wf)
    chans = [x, y] * (nch // 2)
    t = np.sort(np.linspace(-1e-6, 9e-6, n) + rng.normal(size=n) * 1e-12)
prog = _flatten.flatten(chans)
plan = _engine.Plan(prog, t=t)
out = torch.empty((len(chans), n), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
f = lambda: plan.launch(out.data_ptr(), n, _engine.OUT_F64, stream=st)
for _ in range(3): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
m = min(n, 300_000)
sl = slice(n // 3, n // 3 + m)
ref = c_oracle.eval_tlist(_flatten.flatten(chans[:2]), t[sl])
got = out[:2, sl].cpu().numpy()
i = plan.info
print(f'tlist {shape} {len(chans)} x {n}: {ms:.3f} ms = {len(chans) * n / ms * 1e-6:.0f} Gsamples/s  frac(16 B) '
      f'{len(chans) * n * 16 / ms * 1e-6 / 8000:.3f}  {plan.kernel_name()} fused {i.n_fused} generic {i.n_generic} direct {i.n_direct}  '
      f'max|err| {np.max(np.abs(got - ref)):.2e} (peak {np.abs(ref).max():.2f})  lib={os.environ.get("WFK_LIB", "tree")} '
      f'tlfuse={"off" if os.environ.get("WFK_DISABLE_TLFUSE") == "1" else "on"}', flush=True)
