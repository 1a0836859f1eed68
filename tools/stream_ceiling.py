"""Practical HBM ceilings on this box for the shapes the stages stream: write-only (fill),
read-only (sum), copy (read+write), all on 256 x 1e7 fp64 (20.48 GB) like the headline batch."""
import torch
n = 256 * 10**7
x = torch.empty(n, dtype=torch.float64, device='cuda').normal_()
y = torch.empty_like(x)
def t(f, bytes_, name, R=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / R
    print(f'{name}: {ms:.3f} ms  {bytes_ / ms * 1e-9:.2f} TB/s')
t(lambda: y.fill_(1.5), n * 8, 'fill (write 20.48 GB)')
t(lambda: y.copy_(x), 2 * n * 8, 'copy (read+write 40.96 GB)')
t(lambda: torch.mul(x, 2.0, out=y), 2 * n * 8, 'scale (read+write 40.96 GB)')
t(lambda: x.sum(), n * 8, 'sum (read 20.48 GB)')
