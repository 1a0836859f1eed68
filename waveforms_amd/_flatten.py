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
                ('has_last', C.c_int32), ('last', C.c_double)]


class Program:
    """Owns the NumPy arrays behind a `wfk_program` and exposes `.struct`."""

    def __init__(self, arrays: dict, counts: dict, complex_amp: bool):
        self.arrays = arrays
        self.complex_amp = complex_amp
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


def _factor_args(factor):
    type_id, *args, _shift = factor
    if type_id not in _ARGC:
        raise NotImplementedError(
            f'primitive id {type_id} has no device implementation '
            f'(only the built-in ids 1..17 run on the GPU; Python callables '
            f'registered with registerBaseFunc/function() cannot)')
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


def flatten(channels) -> Program:
    ch_member_off = [0]
    ch_offset, ch_tshift, ch_lo, ch_hi = [], [], [], []
    mb_piece_off = [0]
    pc_bound, pc_term_off = [], [0]
    amp_re, amp_im, tm_factor_off = [], [], [0]
    fc_type, fc_power, fc_shift, fc_arg_off, pool = [], [], [], [0], []
    any_complex = False

    for w in channels:
        members, offset, tshift, lo, hi = channel_members(w)
        ch_offset.append(offset)
        ch_tshift.append(tshift)
        ch_lo.append(lo)
        ch_hi.append(hi)
        for bounds, seq in members:
            if len(bounds) != len(seq) or not bounds or bounds[-1] != math.inf:
                raise ValueError('bounds/seq mismatch or last bound is not +inf')
            for b, (terms, amps) in zip(bounds, seq):
                pc_bound.append(float(b))
                for (factors, powers), amp in zip(terms, amps):
                    if isinstance(amp, complex):
                        any_complex = True
                        if math.isfinite(lo) or math.isfinite(hi):
                            raise NotImplementedError(
                                'clip (min/max) of a complex-valued waveform')
                    amp_re.append(float(amp.real))
                    amp_im.append(float(amp.imag))
                    for f, n in zip(factors, powers):
                        if isinstance(n, complex):
                            raise NotImplementedError('complex power')
                        fc_type.append(f[0])
                        fc_power.append(n)
                        fc_shift.append(f[-1])
                        argc = _SIMPLE_ARGC.get(f[0])
                        if argc is not None and len(f) == argc + 2:
                            if argc:
                                pool.extend(f[1:-1])    # converted to float64 in one go below
                        else:
                            pool.extend(_factor_args(f))
                        fc_arg_off.append(len(pool))
                    tm_factor_off.append(len(fc_type))
                pc_term_off.append(len(amp_re))
            mb_piece_off.append(len(pc_bound))
        ch_member_off.append(len(mb_piece_off) - 1)

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
    return Program(arrays, counts, any_complex)


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


def grid_from_desc(desc) -> wfk_grid:
    if desc[0] == 'linspace':
        return grid_linspace(*desc[1:])
    if desc[0] == 'arange':
        return grid_arange(*desc[1:])
    raise ValueError(desc[0])
