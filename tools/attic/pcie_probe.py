"""Host<->device copy rates for an 80 MB NumPy array: pageable vs registered-in-place vs pinned."""
import time, numpy as np, torch
n = 10**7
a = np.random.rand(n)
d = torch.empty(n, dtype=torch.float64, device='cuda')
rt = torch.cuda.cudart()
def t(f, R=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R * 1e3
ta = torch.from_numpy(a)
print('pageable H2D %.2f ms' % t(lambda: d.copy_(ta)))
out = np.empty(n); to = torch.from_numpy(out)
print('pageable D2H %.2f ms' % t(lambda: to.copy_(d)))
def reg_copy():
    rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    d.copy_(ta, non_blocking=True); torch.cuda.synchronize()
    rt.cudaHostUnregister(a.ctypes.data)
print('register+H2D+unregister %.2f ms' % t(reg_copy, 5))
def reg_copy_back():
    rt.cudaHostRegister(out.ctypes.data, out.nbytes, 0)
    to.copy_(d, non_blocking=True); torch.cuda.synchronize()
    rt.cudaHostUnregister(out.ctypes.data)
print('register+D2H+unregister %.2f ms' % t(reg_copy_back, 5))
p = torch.empty(n, dtype=torch.float64).pin_memory()
print('pinned H2D %.2f ms   pinned D2H %.2f ms' % (t(lambda: d.copy_(p, non_blocking=True)), t(lambda: p.copy_(d, non_blocking=True))))
t0 = time.perf_counter(); q = torch.empty(n, dtype=torch.float64).pin_memory(); print('pin_memory alloc 80 MB %.2f ms' % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); np.copyto(out, p.numpy()); print('host memcpy pinned->pageable 80 MB %.2f ms' % ((time.perf_counter() - t0) * 1e3))
