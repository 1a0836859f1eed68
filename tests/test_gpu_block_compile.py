"""Channel blocks compiled on host threads (wfk_compile_blocks) give the SAME samples, bit for bit, as the plan compiled in
one piece, and those agree with the oracle."""
import os

import numpy as np
import pytest

import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten, workloads as wl

pytestmark = pytest.mark.gpu


def _run(prog, grid, threads, dtype=np.float64):
    os.environ['WFK_COMPILE_THREADS'] = str(threads)
    try:
        plan = _engine.Plan(prog, grid=grid)
    finally:
        del os.environ['WFK_COMPILE_THREADS']
    try:
        return plan.run_host(dtype), plan.kernel_name(dtype)
    finally:
        plan.close()


@pytest.mark.parametrize('kind', ['awg', 'awg_interp', 'flat_top', 'lean'])
def test_block_compiled_plans_sample_bit_identically(kind):
    if kind == 'awg':
        chans = [wl.awg_channel(wf, c, 18000, 2e9, c % 3 == 0) for c in range(40)]
        grid = _flatten.grid_from_desc(wl.awg_grid(18000, 2e9))
    elif kind == 'awg_interp':     # table envelopes: the blocks' pools are concatenated, the ops' table offsets move with them
        chans = [wl.awg_interp_channel(wf, c, 18000, 2e9) for c in range(40)]
        grid = _flatten.grid_from_desc(wl.awg_grid(18000, 2e9))
    elif kind == 'flat_top':       # sampled erf edges: a block shares edge tables among ITS channels only, so ...
        chans = [wl.awg_shape_channel(wf, 'flat_top', c, 18000, 2e9) for c in range(40)]
        grid = _flatten.grid_from_desc(wl.awg_grid(18000, 2e9))
    else:
        chans = [wl.sum_channel(wf, 240, 50 + c) for c in range(36)]
        grid = _flatten.grid_from_desc(('linspace', 0.0, 240 * wl.SPAN, 300_000, False))
    prog = _flatten.flatten(chans)
    assert prog.struct.n_pieces >= 8192
    one, name1 = _run(prog, grid, 1)
    for threads in (3, 4, 16):
        many, name = _run(prog, grid, threads)
        assert name == name1, threads
        if kind == 'flat_top':     # ... which edge's samples a shared table holds differs: equal to the sharing bound (2e-11 in v)
            assert np.max(np.abs(one - many)) <= 5e-11, threads
        else:
            assert np.array_equal(one, many), threads
    f1, _ = _run(prog, grid, 1, np.float32)
    f4, _ = _run(prog, grid, 5, np.float32)
    assert np.array_equal(f1, f4) or (kind == 'flat_top' and np.max(np.abs(f1 - f4)) <= 1e-6)
    ref = c_oracle.eval_grid(_flatten.flatten(chans[:3] + chans[-2:]), grid)
    got = np.concatenate([one[:3], one[-2:]])
    assert np.max(np.abs(got - ref)) <= 1e-9
