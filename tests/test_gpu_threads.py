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
