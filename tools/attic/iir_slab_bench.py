"""IIR stage processed in TIME SLABS that fit the 256 MB Infinity Cache: pass A and pass C of a
slab run back to back, so the second read of x can be served on chip instead of from HBM.
usage: python tools/iir_slab_bench.py [rows] [n] [sections]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from waveforms_amd import _engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**7
nsec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sos = butter(2 * nsec, 0.1, output='sos')
secs = [(s[:3], s[3:]) for s in sos]
x = torch.randn((rows, n), dtype=torch.float64, device='cuda')
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
whole = _engine.IirPlan(secs, n, rows, np.float64)
D = whole.state_dim
zi = torch.zeros((rows, D), dtype=torch.float64, device='cuda')
zf = torch.empty_like(zi)

def bench(fn, R=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / R

def run_whole():
    whole.apply(x.data_ptr(), n, y.data_ptr(), n, zi.data_ptr(), zf.data_ptr(), 0.0, stream)
run_whole(); torch.cuda.synchronize()
ref = y.clone(); zref = zf.clone()
print(f'whole: {bench(run_whole):.3f} ms')
for slab in (2**17, 2**18, 2**19, 2**20, 2**21):
    plans, offs = {}, []
    o = 0
    while o < n:
        m = min(slab, n - o)
        if m not in plans:
            plans[m] = _engine.IirPlan(secs, m, rows, np.float64)
        offs.append((o, m)); o += m
    st = [torch.zeros((rows, D), dtype=torch.float64, device='cuda') for _ in range(2)]
    def run_slabs():
        st[0].zero_()
        for i, (o, m) in enumerate(offs):
            plans[m].apply(x.data_ptr() + 8 * o, n, y.data_ptr() + 8 * o, n, st[i & 1].data_ptr(), st[(i + 1) & 1].data_ptr(), 0.0, stream)
    run_slabs(); torch.cuda.synchronize()
    err = float((y - ref).abs().max())
    print(f'slab {slab:8d} ({rows * slab * 8 / 2**20:.0f} MB in): {bench(run_slabs):.3f} ms  launches {3 * len(offs)}  max diff {err:.1e}')
    for p in plans.values(): p.close()
