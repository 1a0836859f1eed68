"""Big grid batches are compiled as channel blocks on host threads and concatenated (wfk_compile_blocks): the plan must
be the one a single-threaded compile produces -- same tier, same piece indices, same op counts (host-only plans here;
bit-identical samples: tests/test_gpu_block_compile.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import waveforms_amd as wf
from waveforms_amd import _engine, _flatten, workloads as wl


def _info(plan):
    info = _engine.wfk_plan_info()
    _engine.lib().wfk_plan_get_info(plan._h, C.byref(info))
    return {f[0]: getattr(info, f[0]) for f in info._fields_}


def _plan(prog, grid, threads):
    os.environ['WFK_COMPILE_THREADS'] = str(threads)
    try:
        return _engine.Plan(prog, grid=grid)
    finally:
        del os.environ['WFK_COMPILE_THREADS']


@pytest.mark.parametrize('kind', ['awg', 'awg_tables', 'lean', 'mixed_shapes'])
def test_blocks_equal_the_single_compile(kind):
    if kind == 'awg':        # short tier: 40 rows x 300 pulses at 2 GS/s
        chans = [wl.awg_channel(wf, c, 18000, 2e9, c % 3 == 0) for c in range(40)]
        grid = _flatten.grid_from_desc(wl.awg_grid(18000, 2e9))
    elif kind == 'awg_tables':   # short tier with pool tables (samplingPoints envelopes, sampled flat-top edges): they move with their block
        chans = [wl.awg_interp_channel(wf, c, 18000, 2e9) if c % 2 else wl.awg_shape_channel(wf, 'flat_top', c, 18000, 2e9) for c in range(40)]
        grid = _flatten.grid_from_desc(wl.awg_grid(18000, 2e9))
    elif kind == 'lean':     # lean tier: 36 rows x 240 pulses on a fine grid
        chans = [wl.sum_channel(wf, 240, 50 + c) for c in range(36)]
        grid = _flatten.grid_from_desc(('linspace', 0.0, 240 * wl.SPAN, 2_000_000, False))
    else:                    # one row with an erf edge among lean rows: not a shape the block path takes -> one piece
        chans = [wl.sum_channel(wf, 240, 50 + c) for c in range(35)] + \
                [(wf.square(400e-9, edge=50e-9) >> 1e-6) * wf.cos(2 * np.pi * 50e6) + wl.sum_channel(wf, 240, 3)]
        grid = _flatten.grid_from_desc(('linspace', 0.0, 240 * wl.SPAN, 2_000_000, False))
    prog = _flatten.flatten(chans)
    assert prog.struct.n_pieces >= 8192
    one, many = _plan(prog, grid, 1), _plan(prog, grid, 4)
    a, b = _info(one), _info(many)
    a.pop('param_doubles'), b.pop('param_doubles')        # (the blocks keep their own padding between them)
    assert a == b, (a, b)
    assert one.kernel_name() == many.kernel_name() and one.kernel_name(np.float32) == many.kernel_name(np.float32)
    for m in (0, 1, len(chans) // 2, len(chans) - 1):
        assert np.array_equal(one.member_index(m), many.member_index(m))
        assert bool(_engine.lib().wfk_plan_channel_is_complex(one._h, m)) == bool(_engine.lib().wfk_plan_channel_is_complex(many._h, m))


def test_invalid_programs_are_refused_before_any_block_runs():
    chans = [wl.awg_channel(wf, c, 18000, 2e9) for c in range(40)]
    prog = _flatten.flatten(chans)
    grid = _flatten.grid_from_desc(wl.awg_grid(18000, 2e9))
    bad = prog.arrays['fc_type'].copy()
    bad[prog.struct.n_factors - 1] = 99                            # an id without a device form, in the LAST block
    prog.arrays['fc_type'], prog.struct.fc_type = bad, bad.ctypes.data
    with pytest.raises(NotImplementedError):
        _plan(prog, grid, 4)
