"""The library is called from several Python threads at once (SURVEY.md §8(b) "Threading":
distinct plans per thread, ctypes releases the GIL during calls).  Every thread runs the
drop-in path (plan -> launch -> copy back -> destroy, which takes and returns blocks of the
device block cache) and a device-resident batch launch on its own stream, and checks its
results against the C oracle; errors are thread-local (wfk_last_error)."""
import threading

import numpy as np
import pytest

import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl
from waveforms_amd._sampling import BatchSampler

pytestmark = pytest.mark.gpu


def _worker(k, errs):
    try:
        import torch
        rng = np.random.default_rng(50 + k)
        stream = torch.cuda.Stream()
        for it in range(12):
            npulse = int(rng.integers(2, 9))
            n = int(rng.integers(3000, 60000))
            w = wl.sum_channel(wf, npulse, 7000 + 100 * k + it)
            t = np.linspace(0.0, npulse * wl.SPAN, n)
            got = w(t)                                            # tlist mode, host round trip
            ref = c_oracle.eval_tlist(_flatten.flatten([w]), t)[0]
            assert np.max(np.abs(got - ref)) <= 1e-9, ('tlist', k, it)
            grid = ('linspace', 0.0, npulse * wl.SPAN, n, False)
            chans = [wl.sum_channel(wf, npulse, 9000 + 10 * k + c) for c in range(3)]
            bs = BatchSampler(chans, grid)
            out = torch.empty((3, n), dtype=torch.float64, device='cuda')
            with torch.cuda.stream(stream):
                bs.launch_torch(out)
            stream.synchronize()
            refb = c_oracle.eval_grid(_flatten.flatten(chans), _flatten.grid_from_desc(grid))
            assert np.max(np.abs(out.cpu().numpy() - refb)) <= 1e-9, ('grid', k, it)
            bs.close()
        # an error raised in this thread is reported to this thread
        with pytest.raises(Exception):
            _engine.Plan(_flatten.flatten([w]), grid=_flatten.grid_linspace(0.0, -1.0, 10, False))
    except BaseException as e:   # noqa: BLE001 - collected and re-raised by the test
        errs.append((k, repr(e)))


def test_four_threads_share_the_library():
    errs = []
    threads = [threading.Thread(target=_worker, args=(k, errs)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errs, errs


def test_repeated_small_calls_reuse_their_plan():
    """the per-thread plan cache of the drop-in calls (_sampling._cached_grid_plan): same values call after call,
    a changed tree or grid is a new plan, trees with Python callables are never cached, other threads have their own"""
    import threading
    import numpy as np
    import waveforms_amd as wf
    from oracle import np_oracle
    from waveforms_amd import _sampling, workloads as wl
    w = wl.sum_channel(wf, 12, 5)
    t = np.linspace(0, 12 * wl.SPAN, 5001)
    ref = np_oracle.call(w, t)
    _sampling._tls.__dict__.pop('plans', None)
    a = w(t)
    n0 = len(_sampling._tls.plans)
    b = w(t)
    assert len(_sampling._tls.plans) == n0 == 1 and np.array_equal(a, b) and np.max(np.abs(a - ref)) <= 1e-9
    w2 = w * 0.5                                   # a new tree: a new plan, the old values stay right
    assert np.max(np.abs(w2(t) - 0.5 * ref)) <= 1e-9 and len(_sampling._tls.plans) == 2
    assert np.max(np.abs(w(t[:-1].copy()) - ref[:-1])) <= 1e-9 and len(_sampling._tls.plans) == 3    # another grid
    w.max = 0.25                                   # clip changed on the same object
    assert np.max(np.abs(w(t) - np.clip(ref, -np.inf, 0.25))) <= 1e-9
    w.max = np.inf
    f = wf.function(np.tanh, start=-1, stop=1) * wf.cos(3.0)
    n1 = len(_sampling._tls.plans)
    f(np.linspace(-2, 2, 2001))
    assert len(_sampling._tls.plans) == n1         # Python callable: not cached
    for _ in range(40):                            # LRU bound
        (w * np.random.rand())(t)
    assert len(_sampling._tls.plans) <= _sampling._PLAN_CACHE_SIZE
    out = []
    th = threading.Thread(target=lambda: out.append((w(t), len(_sampling._tls.plans))))
    th.start(); th.join()
    assert np.array_equal(out[0][0], a) and out[0][1] == 1
