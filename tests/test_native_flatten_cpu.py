"""The C walk of the tuple IR (waveforms_amd/_cflatten, csrc/wfk_pyflatten.c) produces byte for byte the arrays of the
Python walk for every tree of the suite, and leaves everything outside its hot case to the Python walk."""
import numpy as np
import pytest

import cases
import waveforms_amd as wf
from waveforms_amd import _flatten, workloads as wl


def _both(channels, **kw):
    assert _flatten._cflatten is not None, 'waveforms_amd/_cflatten.so is not built (make -C waveforms_amd/csrc)'
    fast = _flatten.flatten(channels, **kw)
    saved, _flatten._cflatten = _flatten._cflatten, None
    try:
        slow = _flatten.flatten(channels, **kw)
    finally:
        _flatten._cflatten = saved
    return fast, slow


def _same(fast, slow):
    for f in ('n_channels', 'n_members', 'n_pieces', 'n_terms', 'n_factors', 'n_pool'):
        assert getattr(fast.struct, f) == getattr(slow.struct, f), f
    for k, b in slow.arrays.items():
        a = fast.arrays[k]
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), k
    assert fast.complex_amp == slow.complex_amp and fast.host_complex == slow.host_complex


@pytest.mark.parametrize('name', sorted(cases.CASES))
def test_native_walk_equals_python_walk(name):
    build, _grid = cases.CASES[name]
    _same(*_both([build(wf)]))


def test_batches_vstacks_and_awg_trains():
    chans = [wl.sum_channel(wf, 9, 3), wl.vstack_channel(wf, 5, 7) >> 3e-9, wl.awg_channel(wf, 1, 4000, 2e9),
             wf.zero(), 0.25 + wl.sum_channel(wf, 2, 1) * (1 + 0.5j)]
    chans[0].max, chans[0].min = 0.5, -0.25
    _same(*_both(chans))
    _same(*_both([]))


def test_what_the_native_walk_hands_back():
    """callables, variable-arity primitives, complex powers, list-built trees, an own function library: the C walk
    returns None for the call and the Python walk produces the program"""
    assert _flatten._cflatten is not None
    argc = {4: 1, 2: 1, 1: 0}
    ok = wf.gaussian(2.0) * wf.cos(3.0)
    assert _flatten._cflatten.flatten_members([(ok.bounds, ok.seq)], argc) is not None
    assert _flatten._cflatten.flatten_members([(ok.bounds, ok.seq)], {4: 1}) is None            # GAUSSIAN not plain here
    assert _flatten._cflatten.flatten_members([(list(ok.bounds), ok.seq)], argc) is None         # a list, not a tuple
    assert _flatten._cflatten.flatten_members([(ok.bounds[:-1] + (5.0, ), ok.seq)], argc) is None   # last bound finite
    cp = wf.cos(3.0) ** (1 + 1j)
    assert _flatten._cflatten.flatten_members([(cp.bounds, cp.seq)], argc) is None               # complex power
    # (a complex power on a built-in primitive evaluates it through the device: covered by tests/test_gpu_powers.py)
    for w in (wf.samplingPoints(0, 1, np.linspace(0, 1, 9)), wf.function(np.tanh, start=-1, stop=1) * wf.cos(2.0)):
        t = np.linspace(-2, 2, 101)
        _same(*_both([w], axis=t))
