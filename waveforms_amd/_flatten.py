"""Host-side flattener: expression trees -> the `wfk_program` struct-of-arrays of
include/wfk.h.

Depth-first in the order of the reference's own flattener `Waveform._tolist`
(waveforms/waveform.py:259-276): channel -> member -> piece -> term -> factor.
Nothing numeric happens here; piece/sample index ranges, vstack merging, fast-path
selection and table generation are done inside the library at plan creation.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _ir

_ARGC = {  # primitive id -> number of scalar args (None = variable)
    _ir.LINEAR: 0, _ir.GAUSSIAN: 1, _ir.ERF: 1, _ir.COS: 1, _ir.SINC: 1,
    _ir.EXP: 1, _ir.INTERP: None, _ir.LINEARCHIRP: 4, _ir.EXPONENTIALCHIRP: 3,
    _ir.HYPERBOLICCHIRP: 3, _ir.COSH: 1, _ir.SINH: 1, _ir.DRAG: 6,
    _ir.MOLLIFIER: 2, _ir.D_GAUSSIAN: 2, _ir.DRAG_SIN: None, _ir.DRAG_SINX: None,
}


class wfk_program(C.Structure):
    _fields_ = [
        ('n_channels', C.c_int32), ('n_members', C.c_int32),
        ('n_pieces', C.c_int32), ('n_terms', C.c_int32),
        ('n_factors', C.c_int32), ('n_pool', C.c_int64),
        ('ch_member_off', C.c_void_p), ('ch_offset', C.c_void_p),
        ('ch_tshift', C.c_void_p), ('ch_clip_lo', C.c_void_p),
        ('ch_clip_hi', C.c_void_p), ('mb_piece_off', C.c_void_p),
        ('pc_bound', C.c_void_p), ('pc_term_off', C.c_void_p),
        ('tm_amp_re', C.c_void_p), ('tm_amp_im', C.c_void_p),
        ('tm_factor_off', C.c_void_p), ('fc_type', C.c_void_p),
        ('fc_power', C.c_void_p), ('fc_shift', C.c_void_p),
        ('fc_arg_off', C.c_void_p), ('pool', C.c_void_p),
    ]


class wfk_grid(C.Structure):
    _fields_ = [('t0', C.c_double), ('step', C.c_double), ('n', C.c_int64),
                ('has_last', C.c_int32), ('last', C.c_double), ('i0', C.c_int64)]


class Program:
    """Owns the NumPy arrays behind a `wfk_program` and exposes `.struct`."""

    def __init__(self, arrays: dict, counts: dict, complex_amp: bool, host_complex: bool = False):
        self.arrays = arrays
        self.complex_amp = complex_amp
        # a Python callable returned complex values, or a factor carries a complex power, in a piece
        # that holds samples: the reference's result is complex128 then (calc_parts looks at the
        # evaluated part's dtype, _waveform.pyx:164-166) even if every amplitude is real
        self.host_complex = host_complex
        s = wfk_program()
        for k, v in counts.items():
            setattr(s, k, v)
        for k, a in arrays.items():
            setattr(s, k, a.ctypes.data)
        self.struct = s
        self.n_channels = counts['n_channels']
        self.n_members = counts['n_members']

    def member_range(self, channel):
        off = self.arrays['ch_member_off']
        return int(off[channel]), int(off[channel + 1])

    def member_bounds(self, member):
        off = self.arrays['mb_piece_off']
        return self.arrays['pc_bound'][off[member]:off[member + 1]]


# fixed-arity primitives whose args go to the pool as they are (the hot case of flatten())
_SIMPLE_ARGC = {k: v for k, v in _ARGC.items() if v is not None and k != _ir.DRAG}


# a complex power in any spelling: Python complex, np.complex128 (a subclass of it) and np.complex64 (not one)
_COMPLEX = (complex, np.complexfloating)

SAMPLED = 1000   # include/wfk.h WFK_SAMPLED: a factor evaluated by the caller, (i0, values...)


def _factor_args(factor):
    type_id, *args, _shift = factor
    if type_id not in _ARGC:
        raise KeyError(type_id)       # as function_lib[func_id] of the reference (_waveform.pyx:131)
    want = _ARGC[type_id]
    if type_id in (_ir.DRAG_SIN, _ir.DRAG_SINX):
        if len(args) != (7 if type_id == _ir.DRAG_SIN else 8):
            raise ValueError(f'{_ir.PRIMITIVE_NAMES[type_id]} factor has the wrong arity')
        from .multy_drag import device_args
        return device_args(type_id, args)
    if type_id == _ir.INTERP:
        if len(args) != 3:
            raise ValueError('INTERP factor must be (7, start, stop, points, shift)')
        start, stop, pts = args
        pts = [float(p) for p in pts]
        if len(pts) < 1:
            raise ValueError('INTERP needs at least one point')
        return [float(start), float(stop), *pts]
    if len(args) != want:
        raise ValueError(
            f'{_ir.PRIMITIVE_NAMES[type_id]} factor takes {want} args, got {len(args)}')
    if type_id == _ir.DRAG:
        args = list(args)
        args[4] = math.nan if args[4] is None else args[4]
    return [float(a) for a in args]


def channel_members(w):
    """(members, offset, tshift, clip_lo, clip_hi) of a Waveform / WaveVStack."""
    from .waveform import WaveVStack
    if isinstance(w, WaveVStack):
        off = w.offset
        return (w.wlist, float(off.real) if isinstance(off, complex) else float(off),
                float(w.shift), -math.inf, math.inf)
    return [(w.bounds, w.seq)], 0.0, 0.0, float(w.min), float(w.max)


def grid_values(grid: 'wfk_grid') -> np.ndarray:
    """The NumPy array a `wfk_grid` stands for: t[i] = fl(fl(i*step) + t0), last element
    overridden when has_last (the linspace / arange element formulas)."""
    t = np.arange(int(grid.i0), int(grid.i0) + int(grid.n), dtype=np.float64) * grid.step + grid.t0
    if grid.has_last and grid.n > 0:
        t[-1] = grid.last
    return t


class _HostFactors:
    """Python-callable primitives (function(), registerBaseFunc, function_lib= overrides).

    A callable cannot run on the device, so the HOST side of the product calls it where the
    reference does -- `function_lib[id](x[start:stop] - shift, *args)` once per distinct factor
    of a piece (_apply / _calc's cache, _waveform.pyx:130-147), with start/stop from
    np.searchsorted on the (channel-shifted) time axis (:156) -- and hands the values to the
    device as a WFK_SAMPLED table factor.  Built-in ids never come through here."""

    def __init__(self, axis):
        self._axis = axis
        self._x = None

    def x(self):
        if self._x is None:
            if self._axis is None:
                raise ValueError('a Python-callable primitive needs the time axis to be evaluated on')
            self._x = (grid_values(self._axis) if isinstance(self._axis, wfk_grid)
                       else np.ascontiguousarray(self._axis, dtype=np.float64))
        return self._x

    def member_axis(self, tshift, bounds):
        """(x - tshift, piece edges) of one member: what calc_parts sees (waveform.py:683-691)."""
        x = self.x()
        xs = x - tshift if tshift != 0 else x
        return xs, np.searchsorted(xs, bounds)

    @staticmethod
    def evaluate(fn, factor, xs, start, stop):
        """-> float64 or complex128 values of the factor over the piece's samples (a complex-valued
        callable is legal: `_apply` is dtype-agnostic, _waveform.pyx:130-131)."""
        _tid, *args, shift = factor
        if stop <= start:
            return np.zeros(0)
        v = np.asarray(fn(xs[start:stop] - shift, *args))
        dt = np.complex128 if np.iscomplexobj(v) else np.float64
        return np.ascontiguousarray(np.broadcast_to(v.astype(dt, copy=False), (stop - start, )))


try:                                   # the native walk of the hot case (csrc/wfk_pyflatten.c, built by the Makefile)
    from . import _cflatten
except ImportError:                    # pragma: no cover  (host-side serialisation only: the Python walk below does the same)
    _cflatten = None
_native_argc = (None, None)            # (registry generation, {type id: argc}) of the default function library


def _flatten_native(channels):
    """The whole call through the C walk when every channel uses the registry's device primitives as they are;
    None when anything needs the Python walk (callables, remapped ids, variable-arity primitives, complex powers)."""
    global _native_argc
    from . import waveform as _w
    if _native_argc[0] != _w._registry_generation:
        _native_argc = (_w._registry_generation,
                        {tid: _SIMPLE_ARGC[tid] for tid, fn in _w._baseFunc.items()
                         if isinstance(fn, _w.BuiltinPrimitive) and fn.type_id == tid and tid in _SIMPLE_ARGC})
    members, ch_member_off, ch = [], [0], []
    for w in channels:
        if getattr(w, 'function_lib', None) is not None:
            return None
        mem, offset, tshift, lo, hi = channel_members(w)
        members.extend(mem)
        ch_member_off.append(len(members))
        ch.append((offset, tshift, lo, hi))
    r = _cflatten.flatten_members(members, _native_argc[1])
    if r is None:
        return None
    mb, bound, pt, are, aim, tf, ft, fp, fs, fa, pool, any_complex = r

    def arr(b, dt):
        a = np.frombuffer(b, dtype=dt)
        return a if a.size else np.zeros(1, dtype=dt)      # keep a valid pointer
    cols = np.asarray(ch, dtype=np.float64).reshape(-1, 4)
    arrays = dict(
        ch_member_off=np.asarray(ch_member_off, dtype=np.int32), ch_offset=np.ascontiguousarray(cols[:, 0]) if len(ch) else np.zeros(1),
        ch_tshift=np.ascontiguousarray(cols[:, 1]) if len(ch) else np.zeros(1),
        ch_clip_lo=np.ascontiguousarray(cols[:, 2]) if len(ch) else np.zeros(1),
        ch_clip_hi=np.ascontiguousarray(cols[:, 3]) if len(ch) else np.zeros(1),
        mb_piece_off=arr(mb, np.int32), pc_bound=arr(bound, np.float64), pc_term_off=arr(pt, np.int32),
        tm_amp_re=arr(are, np.float64), tm_amp_im=arr(aim, np.float64), tm_factor_off=arr(tf, np.int32),
        fc_type=arr(ft, np.int32), fc_power=arr(fp, np.float64), fc_shift=arr(fs, np.float64),
        fc_arg_off=arr(fa, np.int64), pool=arr(pool, np.float64))
    counts = dict(n_channels=len(ch), n_members=len(members), n_pieces=len(bound) // 8, n_terms=len(are) // 8,
                  n_factors=len(ft) // 4, n_pool=len(pool) // 8)
    return Program(arrays, counts, bool(any_complex), False)


def flatten(channels, axis=None, function_lib=None) -> Program:
    """channels -> wfk_program.  `axis` (a wfk_grid or the sorted time array) and
    `function_lib` matter only for primitives that are Python callables: those are evaluated
    here, on the exact sample times (see _HostFactors)."""
    from .waveform import BuiltinPrimitive, _baseFunc
    if _cflatten is not None and function_lib is None:
        channels = list(channels)
        prog = _flatten_native(channels)
        if prog is not None:
            return prog
    ch_member_off = [0]
    ch_offset, ch_tshift, ch_lo, ch_hi = [], [], [], []
    mb_piece_off = [0]
    pc_bound, pc_term_off = [], [0]
    amp_re, amp_im, tm_factor_off = [], [], [0]
    fc_type, fc_power, fc_shift, fc_arg_off = [], [], [], [0]
    pool, pool_parts, pool_len = [], [], 0     # scalars collect in `pool`; big tables go in as arrays
    any_complex = False
    host_complex = False
    host = None

    for w in channels:
        members, offset, tshift, lo, hi = channel_members(w)
        ch_offset.append(offset)
        ch_tshift.append(tshift)
        ch_lo.append(lo)
        ch_hi.append(hi)
        # the library of this channel: explicit argument, else the WaveVStack's own, else the
        # registry (reference: waveform.py:539-540, 685-689).  `native`: ids whose entry is the
        # device primitive of that very id -- the hot case, no lookup per factor.
        lib = function_lib if function_lib is not None else getattr(w, 'function_lib', None)
        if lib is None:
            lib = _baseFunc
        native = {tid for tid, fn in lib.items()
                  if isinstance(fn, BuiltinPrimitive) and fn.type_id == tid}
        for bounds, seq in members:
            if len(bounds) != len(seq):
                raise ValueError('bounds/seq mismatch')
            if not bounds or bounds[-1] != math.inf:
                # The reference's calc_parts (_waveform.pyx:155-169) walks whatever bounds it is given and leaves
                # the samples behind the last one untouched (zero): trees whose last bound is finite, or that
                # have no piece at all -- `mask()` of a waveform that never returns to zero builds such ones,
                # waveform.py:456-482 -- evaluate to 0 there.  Same thing, said explicitly: a zero piece to +inf.
                bounds, seq = tuple(bounds) + (math.inf, ), tuple(seq) + (_ir.ZERO, )
            xs = edges = None
            for ip, (b, (terms, amps)) in enumerate(zip(bounds, seq)):
                pc_bound.append(float(b))
                cache = {}            # distinct factor -> values, per piece (_calc's lru_cache)
                for (factors, powers), amp in zip(terms, amps):
                    if isinstance(amp, complex):
                        any_complex = True
                    # Factors whose VALUES are complex -- a complex-valued Python callable, or any factor
                    # raised to a complex power (`value**n`, _waveform.pyx:143-146) -- cannot be device
                    # factors (those are real).  The host evaluates them where the reference does, applies
                    # the power with NumPy as the reference does, and the term is expanded over (re, im) of
                    # each such factor: amp (re + i im) rest = amp re rest + (i amp) im rest.
                    # Hot path: every factor a native device primitive with a real power -> appended as it is.
                    nf0, np0 = len(fc_type), len(pool)
                    for f, n in zip(factors, powers):
                        tid = f[0]
                        if tid not in native or isinstance(n, _COMPLEX):
                            break
                        fc_power.append(n)
                        fc_shift.append(f[-1])
                        fc_type.append(tid)
                        argc = _SIMPLE_ARGC.get(tid)
                        if argc is not None and len(f) == argc + 2:
                            if argc:
                                pool.extend(f[1:-1])    # converted to float64 in one go below
                        else:
                            pool.extend(_factor_args(f))
                        fc_arg_off.append(pool_len + len(pool))
                    else:
                        amp_re.append(float(amp.real))
                        amp_im.append(float(amp.imag))
                        tm_factor_off.append(len(fc_type))
                        continue
                    del fc_type[nf0:], fc_power[nf0:], fc_shift[nf0:], fc_arg_off[nf0 + 1:], pool[np0:]
                    cplx = []         # [(start, re values, im values)]
                    staged = []       # the term's other factors: (type, power, shift, args | (start, values))
                    for f, n in zip(factors, powers):
                        tid = f[0]
                        if tid in native and not isinstance(n, _COMPLEX):
                            argc = _SIMPLE_ARGC.get(tid)
                            staged.append((tid, n, f[-1], f[1:-1] if (argc is not None and len(f) == argc + 2)
                                           else _factor_args(f)))
                            continue
                        fn = lib[tid]     # KeyError for an unknown id, like the reference
                        if isinstance(fn, BuiltinPrimitive) and not isinstance(n, _COMPLEX):
                            # the id is mapped onto ANOTHER device primitive
                            staged.append((fn.type_id, n, f[-1], _factor_args((fn.type_id, ) + tuple(f[1:]))))
                            continue
                        if host is None:
                            host = _HostFactors(axis)
                        if xs is None:
                            xs, edges = host.member_axis(tshift, bounds)
                        start = int(edges[ip - 1]) if ip > 0 else 0
                        stop = int(edges[ip])
                        if f not in cache:
                            cache[f] = host.evaluate(fn, f, xs, start, stop)
                        vals = cache[f]
                        if isinstance(n, _COMPLEX) or np.iscomplexobj(vals):
                            v = vals if n == 1 else vals**n
                            v = v.astype(np.complex128, copy=False)
                            cplx.append((start, np.ascontiguousarray(v.real), np.ascontiguousarray(v.imag)))
                            if stop > start:
                                host_complex = True
                        else:
                            staged.append((SAMPLED, n, f[-1], (start, vals)))
                    if cplx and (math.isfinite(lo) or math.isfinite(hi)):
                        pass      # (clip of a complex piece: np.clip's lexicographic rule, done by the kernels)
                    for choice in range(1 << len(cplx)):
                        a_c = complex(amp)
                        tabs = []
                        for k, (start, re, im) in enumerate(cplx):
                            if (choice >> k) & 1:
                                a_c = a_c * 1j
                                tabs.append((start, im))
                            else:
                                tabs.append((start, re))
                        if cplx:
                            any_complex = True
                        amp_re.append(float(a_c.real))
                        amp_im.append(float(a_c.imag))
                        for tid, n, shift, args in staged + [(SAMPLED, 1, 0.0, t) for t in tabs]:
                            fc_power.append(n)
                            fc_shift.append(shift)
                            fc_type.append(tid)
                            if tid == SAMPLED:
                                start, vals = args
                                pool.append(float(start))
                                if len(vals):
                                    pool_parts.append(np.asarray(pool, dtype=np.float64))
                                    pool_parts.append(vals)
                                    pool_len += len(pool) + len(vals)
                                    pool = []
                            elif args:
                                pool.extend(args)    # converted to float64 in one go below
                            fc_arg_off.append(pool_len + len(pool))
                        tm_factor_off.append(len(fc_type))
                pc_term_off.append(len(amp_re))
            mb_piece_off.append(len(pc_bound))
        ch_member_off.append(len(mb_piece_off) - 1)
    if pool_parts:
        pool_parts.append(np.asarray(pool, dtype=np.float64))
        pool = np.concatenate(pool_parts)

    def i32(x):
        return np.ascontiguousarray(x, dtype=np.int32)

    def f64(x):
        a = np.ascontiguousarray(x, dtype=np.float64)
        return a if a.size else np.zeros(1)  # keep a valid pointer

    arrays = dict(
        ch_member_off=i32(ch_member_off), ch_offset=f64(ch_offset),
        ch_tshift=f64(ch_tshift), ch_clip_lo=f64(ch_lo), ch_clip_hi=f64(ch_hi),
        mb_piece_off=i32(mb_piece_off), pc_bound=f64(pc_bound),
        pc_term_off=i32(pc_term_off), tm_amp_re=f64(amp_re),
        tm_amp_im=f64(amp_im), tm_factor_off=i32(tm_factor_off),
        fc_type=i32(fc_type) if fc_type else np.zeros(1, np.int32),
        fc_power=f64(fc_power), fc_shift=f64(fc_shift),
        fc_arg_off=np.ascontiguousarray(fc_arg_off, dtype=np.int64),
        pool=f64(pool))
    counts = dict(n_channels=len(ch_offset), n_members=len(mb_piece_off) - 1,
                  n_pieces=len(pc_bound), n_terms=len(amp_re),
                  n_factors=len(fc_type), n_pool=len(pool))
    return Program(arrays, counts, any_complex, host_complex)


def tile_program(prog: Program, reps: int) -> Program:
    """`prog` with its channel list repeated `reps` times (every copy owns its rows of every
    table: the device tables of a big synthetic batch are as large as those of distinct channels)."""
    if reps <= 1:
        return prog
    a, s = prog.arrays, prog.struct

    def offs(name, total):
        o = a[name]
        body = o[1:].astype(np.int64)
        return np.ascontiguousarray(np.concatenate(
            [o[:1].astype(np.int64)] + [body + r * total for r in range(reps)]), dtype=o.dtype)

    def rep(name, count):
        return np.ascontiguousarray(np.tile(a[name][:count], reps)) if count else a[name]
    arrays = dict(
        ch_member_off=offs('ch_member_off', s.n_members), ch_offset=rep('ch_offset', s.n_channels),
        ch_tshift=rep('ch_tshift', s.n_channels), ch_clip_lo=rep('ch_clip_lo', s.n_channels),
        ch_clip_hi=rep('ch_clip_hi', s.n_channels), mb_piece_off=offs('mb_piece_off', s.n_pieces),
        pc_bound=rep('pc_bound', s.n_pieces), pc_term_off=offs('pc_term_off', s.n_terms),
        tm_amp_re=rep('tm_amp_re', s.n_terms), tm_amp_im=rep('tm_amp_im', s.n_terms),
        tm_factor_off=offs('tm_factor_off', s.n_factors), fc_type=rep('fc_type', s.n_factors),
        fc_power=rep('fc_power', s.n_factors), fc_shift=rep('fc_shift', s.n_factors),
        fc_arg_off=offs('fc_arg_off', s.n_pool), pool=rep('pool', s.n_pool))
    counts = dict(n_channels=s.n_channels * reps, n_members=s.n_members * reps,
                  n_pieces=s.n_pieces * reps, n_terms=s.n_terms * reps,
                  n_factors=s.n_factors * reps, n_pool=s.n_pool * reps)
    if max(counts['n_pieces'], counts['n_terms'], counts['n_factors']) >= 2**31:
        raise ValueError('tiled program exceeds the 32-bit table indices')
    return Program(arrays, counts, prog.complex_amp, prog.host_complex)


def grid_linspace(a, b, n, endpoint=True) -> wfk_grid:
    """np.linspace(a, b, n, endpoint): step = (b-a)/div; t[i] = fl(fl(i*step)+a),
    last element overridden by b when endpoint."""
    a, b, n = float(a), float(b), int(n)
    div = (n - 1) if endpoint else n
    step = (b - a) / div if div > 0 else math.nan
    if n > 0 and div > 0 and step == 0.0 and b != a:
        raise NotImplementedError('linspace with underflowing step')
    if n == 1:
        step = 0.0  # only t[0] = a is ever produced
    has_last = 1 if (endpoint and n > 1) else 0
    return wfk_grid(a, step, n, has_last, b)


def grid_arange(start, stop, step) -> wfk_grid:
    """np.arange(start, stop, step) for floats: n = ceil((stop-start)/step),
    delta = fl(fl(start+step) - start), t[i] = fl(start + fl(i*delta))."""
    start, stop, step = float(start), float(stop), float(step)
    n = max(0, int(math.ceil((stop - start) / step)))
    delta = (start + step) - start
    return wfk_grid(start, delta, n, 0, 0.0)


def grid_slice(grid: wfk_grid, a: int, b: int) -> wfk_grid:
    """Samples [a, b) of `grid` as a grid of their own (wfk_grid.i0): times, piece indices and sample values are
    those of the same samples of the whole grid -- what a rank of a time-sharded job, or one chunk of a chunked
    job, samples."""
    a, b = int(a), int(b)
    if not 0 <= a <= b <= int(grid.n):
        raise ValueError('slice outside the grid')
    last_in = bool(grid.has_last) and b == int(grid.n) and b > a
    return wfk_grid(grid.t0, grid.step, b - a, 1 if last_in else 0, grid.last if last_in else 0.0, int(grid.i0) + a)


def grid_from_desc(desc) -> wfk_grid:
    if desc[0] == 'linspace':
        return grid_linspace(*desc[1:])
    if desc[0] == 'arange':
        return grid_arange(*desc[1:])
    raise ValueError(desc[0])
