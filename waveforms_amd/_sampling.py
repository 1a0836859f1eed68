"""Sampling entry points: the Python side of the drop-in boundary.

`call_waveform` / `call_vstack` / `sample_waveform` keep the reference semantics of
`Waveform.__call__` (waveforms/waveform.py:529-563), `WaveVStack.__call__`
(:679-693) and `Waveform.sample` / `_sample_iter` (:173-257) -- argument meaning,
output dtype rule, `out`/`accumulate`/`frag` behaviour, exception types -- while all
evaluation happens in the HIP library.  `sample_batch` is the additive multi-channel
API (no reference counterpart): many channels, one launch, output left in HBM.
"""
from __future__ import annotations

import collections
import os
import threading

import numpy as np

from . import _engine, _flatten
from ._ir import ZERO


def _as_time_array(x):
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError('x must be a scalar or a 1-D sorted array')
    return np.ascontiguousarray(x, dtype=np.float64)


def _live_pieces(plan, member, seq):
    """[(start, stop, expr)] of the evaluated pieces of one member: start < stop
    and expr != ZERO, exactly the reference's condition (_waveform.pyx:160-161)."""
    out, start = [], 0
    for stop, expr in zip(plan.member_index(member), seq):
        stop = int(stop)
        if start < stop and expr != ZERO:
            out.append((start, stop, expr))
        start = stop
    return out


def _has_complex_amp(expr):
    return any(isinstance(a, complex) for a in expr[1])


def _result_dtype(live):
    # dtype detection of calc_parts (_waveform.pyx:164-166)
    return np.complex128 if any(_has_complex_amp(e) for _, _, e in live) else np.float64


def _is_complex(w, plan):
    """The reference's dtype rule for a Waveform (_waveform.pyx:164-166): complex128 when an evaluated
    piece is complex -- a complex amplitude, a complex-valued callable or a complex power."""
    return (_result_dtype(_live_pieces(plan, 0, w.seq)) is np.complex128) or plan.prog.host_complex


def _finish(w, plan, frag, out, accumulate):
    live = _live_pieces(plan, 0, w.seq)
    dtype = np.complex128 if (_result_dtype(live) is np.complex128 or plan.prog.host_complex) else np.float64
    if not frag and out is not None and not accumulate and isinstance(out, np.ndarray) and out.ndim == 1 \
            and out.dtype == dtype and out.flags.c_contiguous and out.flags.writeable and len(out) >= plan.n \
            and _engine.all_finite(out):
        # `out *= 0; out[:n] += res` of the reference (waveform.py:548-563) without a temporary and without two
        # more passes over 80 MB: the samples are copied straight into the caller's array.  (`out *= 0` keeps a
        # NaN / inf that sits in `out`: an array holding one takes the reference's own two passes below.)
        plan.run_host_into(out[:plan.n])
        if len(out) > plan.n:
            out[plan.n:] *= 0
        return out
    res = plan.run_host(dtype)[0]
    if not frag:
        if out is None:
            return res
        if not accumulate:
            out *= 0
        out[:len(res)] += res
        return out
    parts = []
    for a, b, expr in live:
        if all(len(f) == 0 for f, _ in expr[0]):
            part = res[a]            # constant piece: the reference yields a scalar
        else:
            part = res[a:b].copy()
        parts.append((a, b, part))
    if out is None:
        return parts
    if accumulate:
        raise NotImplementedError  # as the reference (_merge_parts, waveform.py:516-522)
    out.clear()
    out.extend(parts)
    return out


# ---- plans of recent grid calls, per thread ---------------------------------------------------------------------
# A drop-in call on a small grid is host work: flatten 300 us + plan creation 160 us + 45 us of launch and copy for a
# 100-pulse channel on 1e4 points.  Scripts call the SAME waveform on the SAME grid again and again (a shot loop, a
# scan that changes another channel): the plan of such a call is kept -- keyed by the identity of the waveform's
# immutable (bounds, seq) tuples, which the entry keeps alive, and by the grid's numbers -- and the repeat costs the
# launch alone.  Per thread (a plan owns scratch for its launches), 16 entries, never for trees with Python callables
# (their values are the callable's business at every call), never for a Waveform built on MUTABLE sequences (lists
# handed to the constructor: identity says nothing about content), never for the chunk grids of `_sample_iter`
# (each is used once).  The registry's generation counter is part of the key.
_PLAN_CACHE_SIZE = 16
_PLAN_CACHE_MAX_N = 1 << 20
_tls = threading.local()


def _tree_key(w):
    from .waveform import WaveVStack
    if getattr(w, 'function_lib', None) is not None:
        return None, None
    if isinstance(w, WaveVStack):
        members = tuple(w.wlist)
        if not all(type(m) is tuple and type(m[0]) is tuple and type(m[1]) is tuple for m in members):
            return None, None
        return (('v', w.offset, w.shift) + tuple((id(b), id(s)) for b, s in members)), members
    if type(w.bounds) is not tuple or type(w.seq) is not tuple:
        return None, None
    return ('w', id(w.bounds), id(w.seq), w.min, w.max), (w.bounds, w.seq)


def _cached_grid_plan(w, grid, function_lib, no_cache=False):
    """-> (plan, owned): owned plans are the caller's to close, cached ones stay open"""
    key, keep = (None, None) if (function_lib is not None or no_cache) else _tree_key(w)
    if int(grid.n) > _PLAN_CACHE_MAX_N:
        key = None                              # (a cached plan keeps its result buffer on the device: small calls only)
    if key is None:
        return _engine.Plan(_flatten.flatten([w], grid, function_lib), grid=grid), True
    cache = getattr(_tls, 'plans', None)
    if cache is None:
        cache = _tls.plans = collections.OrderedDict()
    from . import waveform as _wmod
    key = key + (grid.t0, grid.step, int(grid.n), int(grid.has_last), grid.last, int(grid.i0),
                 _wmod._registry_generation)
    hit = cache.get(key)
    if hit is not None and hit[0]._h:          # (a plan someone closed behind the cache's back is rebuilt)
        cache.move_to_end(key)
        return hit[0], False
    prog = _flatten.flatten([w], grid, None)
    plan = _engine.Plan(prog, grid=grid)
    if _flatten.SAMPLED in prog.arrays['fc_type'][:prog.struct.n_factors]:
        return plan, True                      # a registered Python callable: evaluated afresh at every call
    cache[key] = (plan, keep)
    while len(cache) > _PLAN_CACHE_SIZE:
        _, (old, _keep) = cache.popitem(last=False)
        old.close()
    return plan, False


def _plan_for_axis(w, t, function_lib, grid=False):
    """x is almost always np.linspace / np.arange output: when it is bit-identical to the grid
    formula (checked element by element in the library) the plan is compiled in grid mode --
    fused ops, no upload of x -- and the device regenerates exactly the caller's times.  Any
    other sorted x stays in tlist mode."""
    if grid is False:                      # (not looked at yet by the caller)
        grid = _engine.detect_grid(t)
    if grid is not None:
        return _cached_grid_plan(w, grid, function_lib)
    return _engine.Plan(_flatten.flatten([w], t, function_lib), t=t), True


def call_waveform(w, x, frag=False, out=None, accumulate=False, function_lib=None):
    if isinstance(x, (int, float, complex)):
        return call_waveform(w, np.array([x]), function_lib=function_lib)[0]
    t = _as_time_array(x)
    grid = _engine.detect_grid(t)
    if _RUNS_ON and grid is None and not frag and len(t) >= 2 * _RUN_MIN:
        runs = _engine.detect_grid_runs(t, _RUN_MIN)
        if runs is not None:
            return _call_runs(w, t, runs, out, accumulate, function_lib)
    plan, owned = _plan_for_axis(w, t, function_lib, grid)
    try:
        return _finish(w, plan, frag, out, accumulate)
    finally:
        if owned:
            plan.close()


_RUN_MIN = 4096      # shortest run worth a plan of its own
# Run-by-run grid sampling of an x made of several grids is OFF unless WFK_GRID_RUNS=1: since the time-list tier
# evaluates fused groups pointwise it is the faster way (1e7 points in four runs: 11.7 ms run by run -- 8 ms of
# it the host's run detection -- against 4.2 ms as ONE time list, tools/big_call_latency.py); the run path is
# kept for callers who want the grid tiers' values (and as the place the library's run detector is exercised).
_RUNS_ON = os.environ.get('WFK_GRID_RUNS') == '1'


def _call_runs(w, t, runs, out, accumulate, function_lib):
    """x = several NumPy grids back to back (windows of one sequence, chunks at two rates; every run verified
    element by element by the library): each run is sampled in grid mode -- fused ops, no upload of x -- into
    its part of one result.  Same values as one call per run (reference waveform.py:529-563)."""
    from .waveform import WaveVStack
    ends = [a for a, _ in runs[1:]] + [len(t)]
    plans = [_engine.Plan(_flatten.flatten([w], g, function_lib), grid=g) for _, g in runs]
    try:
        cplx = (not isinstance(w, WaveVStack)) and any(_is_complex(w, p) for p in plans)
        dtype = np.complex128 if cplx else np.float64
        res = _engine.pinned_empty((len(t), ), dtype)
        for (a, _), b, p in zip(runs, ends, plans):
            p.run_host_into(res[a:b])
    finally:
        for p in plans:
            p.close()
    if out is None or isinstance(w, WaveVStack):
        return res
    if not accumulate:
        out *= 0
    out[:len(res)] += res
    return out


def call_vstack(w, x, function_lib=None):
    if function_lib is None and w.function_lib is not None:
        function_lib = w.function_lib
    if isinstance(x, (int, float, complex)):
        return call_vstack(w, np.array([x]), function_lib)[0]
    t = _as_time_array(x)
    grid = _engine.detect_grid(t)
    if _RUNS_ON and grid is None and len(t) >= 2 * _RUN_MIN:
        runs = _engine.detect_grid_runs(t, _RUN_MIN)
        if runs is not None:
            return _call_runs(w, t, runs, None, False, function_lib)
    plan, owned = _plan_for_axis(w, t, function_lib, grid)
    try:
        return plan.run_host(np.float64)[0]
    finally:
        if owned:
            plan.close()


def _sos_sections(sos):
    sos = np.asarray(sos, dtype=np.float64).reshape(-1, 6)
    return [(row[:3], row[3:]) for row in sos]


def _rotated(prog):
    """The program with every amplitude multiplied by -1j: its real part is the imaginary part of `prog`."""
    arrays = dict(prog.arrays)
    arrays['tm_amp_re'] = np.ascontiguousarray(prog.arrays['tm_amp_im'])
    arrays['tm_amp_im'] = np.ascontiguousarray(-prog.arrays['tm_amp_re'])
    s = prog.struct
    counts = {k: getattr(s, k) for k in ('n_channels', 'n_members', 'n_pieces', 'n_terms', 'n_factors', 'n_pool')}
    return _flatten.Program(arrays, counts, True, prog.host_complex)


def _sample_filtered(w, plan, sos, initial, zi):
    """sampler -> SOS IIR cascade, both on the device; one download.
    == sosfilt(sos, sig - initial, zi) + initial (reference waveform.py:193-203,244-251).
    A complex-valued waveform goes through as two real rows -- its real and its imaginary part, the
    second sampled from the program with the amplitudes turned by -1j: the sections are real, so the
    filter acts on the two parts independently and the result is exact."""
    from .waveform import WaveVStack
    cplx = (not isinstance(w, WaveVStack)) and _is_complex(w, plan)
    if zi is not None and np.iscomplexobj(zi) and np.any(np.imag(zi) != 0):
        cplx = True          # a chunk of real samples behind complex ones: the carried state is complex
    n = plan.n
    rows = 2 if cplx else 1
    # one sampler -> IIR chain per real row (wfk_chain_iir_*): the wave that owns a chunk of the IIR scan evaluates its
    # input itself when the tree is fully fused (the unfiltered samples never exist in memory), otherwise sampler and
    # filter run back to back on the device buffer
    sections = _sos_sections(sos)
    chains = [_engine.ChainIirPlan(plan.prog, plan.grid, sections)]
    D = chains[0].state_dim
    buf = _engine.DeviceBuffer(max(n, 1) * 8 * rows)
    dzi = _engine.DeviceBuffer(max(D, 1) * 8 * rows)
    dzf = _engine.DeviceBuffer(max(D, 1) * 8 * rows)
    try:
        z0 = np.zeros(D, dtype=np.complex128) if zi is None else np.asarray(zi, dtype=np.complex128).reshape(-1)
        init = complex(initial or 0.0)
        if cplx:
            chains.append(_engine.ChainIirPlan(_rotated(plan.prog), plan.grid, sections))
        dzi.upload(np.ascontiguousarray(np.concatenate([z0.real, z0.imag])[:D * rows]))
        for attempt in range(2):
            ok = True
            for r, chain in enumerate(chains):
                off, zoff = r * max(n, 1) * 8, r * max(D, 1) * 8
                ok = chain.launch(buf.ptr + off, max(n, 1), dzi.ptr + zoff, dzf.ptr + zoff,
                                  init.imag if r else init.real) and ok
            ok = all([chain.status() for chain in chains]) and ok
            if ok:
                break
            # a single-pass look-back timed out (its outputs hold NaN): the chains have switched to the
            # three-launch form behind the plain sampler; launch again
            if attempt == 1:
                raise _engine.EngineError('IIR stage failed twice')
        _engine.sync()
        if n:
            sig = buf.download((rows, max(n, 1)), np.float64)
            sig = sig[0] + 1j * sig[1] if cplx else sig[0]
        else:
            sig = np.zeros(0, dtype=np.complex128 if cplx else np.float64)
        zf = dzf.download((rows, max(D, 1)), np.float64)[:, :D]
        zf = (zf[0] + 1j * zf[1] if cplx else zf[0]).reshape(-1, 2)
    finally:
        buf.close()
        dzi.close()
        dzf.close()
        for chain in chains:
            chain.close()
    return sig, zf


def _sample_on_grid(w, grid, out, function_lib, filters=None, zi=None, no_cache=False):
    from .waveform import WaveVStack
    plan, owned = _cached_grid_plan(w, grid, function_lib, no_cache)
    try:
        if filters is not None:
            sos, initial = filters
            return _sample_filtered(w, plan, sos, initial, zi)
        if isinstance(w, WaveVStack):
            return plan.run_host(np.float64)[0]
        return _finish(w, plan, False, out, False)
    finally:
        if owned:
            plan.close()


def sample_waveform(w, sample_rate=None, out=None, chunk_size=None, function_lib=None,
                    filters=None):
    if sample_rate is None:
        sample_rate = w.sample_rate
    if w.start is None or w.stop is None or sample_rate is None:
        raise ValueError(
            f'Waveform is not initialized. {w.start=}, {w.stop=}, {sample_rate=}')
    if filters is None:
        filters = w.filters
    if chunk_size is None:
        grid = _flatten.grid_arange(w.start, w.stop, 1 / sample_rate)
        if filters is not None:
            # NB the reference passes out= to __call__ and then returns the FILTERED
            # array, a new object (waveform.py:191-204)
            return _sample_on_grid(w, grid, None, function_lib, filters)[0]
        return _sample_on_grid(w, grid, out, function_lib)
    return _sample_iter(w, sample_rate, chunk_size, out, function_lib, filters)


def _sample_iter(w, sample_rate, chunk_size, out, function_lib, filters):
    # chunk grid: np.linspace(start, stop, size, endpoint=False) per chunk; IIR state is
    # carried from chunk to chunk (reference: waveforms/waveform.py:209-257)
    start, start_n = float(w.start), 0
    length = chunk_size / sample_rate
    zi = None
    while start < w.stop:
        if start + length > w.stop:
            length = w.stop - start
            stop = float(w.stop)
            size = round((stop - start) * sample_rate)
        else:
            stop = start + length
            size = chunk_size
        grid = _flatten.grid_linspace(start, stop, size, endpoint=False)
        if filters is None:
            yield _sample_on_grid(w, grid, None if out is None else out[start_n:],
                                  function_lib, no_cache=True)
        else:
            if size == 0:
                # a last chunk of zero samples (start one ulp short of stop): the reference hands
                # it to scipy.signal.sosfilt, which rejects an empty signal (waveform.py:249)
                raise ValueError('cannot reshape array of size 0 into shape (0)')
            sig, zi = _sample_on_grid(w, grid, None, function_lib, filters, zi, no_cache=True)
            if out is not None:
                out[start_n:start_n + size] = sig
            yield sig
        start = stop
        start_n += chunk_size


# ---------------------------------------------------------------------------
# batched multi-channel API (additive)
# ---------------------------------------------------------------------------
class BatchSampler:
    """Many channels (Waveform / WaveVStack objects) sampled on ONE uniform grid
    by a single kernel launch.  Build once, launch many times.

        bs = BatchSampler(channels, ('linspace', 0.0, 3e-6, 10**7, False))
        bs.launch(out_ptr, ch_stride, dtype)        # async, output stays in HBM
        arr = bs.to_host(np.float32)                # or: run + copy back
    """

    def __init__(self, channels, grid, function_lib=None, tile=1):
        """`tile` > 1 repeats the channel list that many times (rows c, c + len(channels), ... are copies:
        every copy owns its rows of every device table) -- synthetic batches of thousands of rows without
        building thousands of expression trees."""
        if not isinstance(grid, _flatten.wfk_grid):
            grid = _flatten.grid_from_desc(grid)
        self.grid = grid
        self.prog = _flatten.tile_program(_flatten.flatten(list(channels), grid, function_lib), tile)
        self.plan = _engine.Plan(self.prog, grid=grid)
        self.n = self.plan.n
        self.n_channels = self.plan.n_channels

    def launch(self, out_ptr: int, ch_stride: int | None = None, dtype=np.float64,
               accumulate=False, stream: int = 0):
        kind = _engine._KIND_OF[np.dtype(dtype)]
        self.plan.launch(out_ptr, self.n if ch_stride is None else ch_stride, kind,
                         accumulate, stream)

    def launch_torch(self, out, accumulate=False):
        """Launch into a CUDA/HIP torch tensor of shape (n_channels, >= n) on the
        current torch stream; returns `out`."""
        import torch
        if not out.is_cuda or out.dim() != 2 or out.shape[0] != self.n_channels \
                or out.shape[1] < self.n or out.stride(1) != 1:
            raise ValueError('out must be a (n_channels, >=n) row-contiguous device tensor')
        dtype = {torch.float64: np.float64, torch.float32: np.float32,
                 torch.complex128: np.complex128, torch.complex64: np.complex64}[out.dtype]
        stream = torch.cuda.current_stream(out.device).cuda_stream
        self.launch(out.data_ptr(), out.stride(0), dtype, accumulate, stream)
        return out

    def to_host(self, dtype=np.float64):
        return self.plan.run_host(dtype)

    def close(self):
        self.plan.close()


def sample_batch(channels, grid, dtype=np.float64, out=None):
    """Sample `channels` on `grid`.  With a device tensor `out` the result stays in
    HBM (returned as `out`); otherwise a (n_channels, n) NumPy array is returned."""
    bs = BatchSampler(channels, grid)
    try:
        if out is not None:
            return bs.launch_torch(out)
        return bs.to_host(dtype)
    finally:
        if out is None:
            bs.close()
