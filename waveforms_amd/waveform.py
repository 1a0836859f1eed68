"""`Waveform` / `WaveVStack` object model and pulse constructors.

Public names, signatures, attribute names and the flat-list / tree encodings are
those of the reference (waveforms/waveform.py:125-844, 886-896, 1055-1527) so that
existing pulse scripts run unchanged.  What differs is *what sampling does*:
`Waveform.__call__`, `Waveform.sample` and `WaveVStack.__call__` flatten the tree
(`_flatten.py`) and hand it through the ctypes C-ABI (`_engine.py`,
include/wfk.h) to the HIP sampler in `csrc/`.  There is no NumPy evaluation path
in this package: if `libwfk_hip.so` or a GPU is missing, sampling raises.
"""
from __future__ import annotations

import math
from typing import Iterable

import numpy as np
from numpy import inf, pi

from . import _ir
from ._ir import (COS, COSH, D_GAUSSIAN, DRAG, ERF, EXP, EXPONENTIALCHIRP,
                  GAUSSIAN, HALF, HYPERBOLICCHIRP, INTERP, LINEAR, LINEARCHIRP,
                  MOLLIFIER, NDIGITS, ONE, SINC, SINH, ZERO, const_expr,
                  primitive)

# --------------------------------------------------------------------------
# primitive registry (reference: _waveform.pyx:10-13, 264-287, 374-388)
# --------------------------------------------------------------------------
class BuiltinPrimitive:
    """Registry entry of a primitive that has a device implementation (ids 1..17).

    It marks the id as "evaluated by the HIP kernels"; calling it evaluates the primitive
    on the device too (`_baseFunc[GAUSSIAN](t, sigma)` inside a user's own callable works),
    so there is no NumPy evaluation of built-ins anywhere in this package."""

    __slots__ = ('type_id', )

    def __init__(self, type_id):
        self.type_id = type_id

    def __call__(self, t, *args):
        scalar = np.ndim(t) == 0
        x = np.atleast_1d(np.asarray(t, dtype=np.float64))
        # one piece up to +inf: np.searchsorted(x, [inf]) == len(x) whatever the order of x
        y = Waveform(seq=(primitive(self.type_id, *args), ))(x.ravel()).reshape(x.shape)
        return y[0] if scalar else y

    def __repr__(self):
        return f'<device primitive {_ir.PRIMITIVE_NAMES.get(self.type_id, self.type_id)}>'

    def __reduce__(self):
        return (BuiltinPrimitive, (self.type_id, ))


#: id -> evaluator, the default `function_lib` (reference: `_baseFunc`).  Built-in ids map to
#: BuiltinPrimitive markers; ids from registerBaseFunc()/function() map to the user's callable,
#: which the HOST calls once per distinct factor and piece on the exact sample times
#: (`function_lib[id](x[start:stop] - shift, *args)`, reference _waveform.pyx:130-131); the
#: device then reads the values as a per-sample table factor (_flatten.py, WFK_SAMPLED).
_baseFunc: dict[int, object] = {i: BuiltinPrimitive(i) for i in range(1, _ir.FIRST_USER_TYPE)}
_next_type_id = _ir.FIRST_USER_TYPE
#: bumped whenever `_baseFunc` changes (keys of cached plans include it: `_sampling._cached_grid_plan`)
_registry_generation = 0


def registerBaseFunc(func) -> int:
    """Register a Python callable `func(t_shifted_array, *args) -> array` as a primitive and
    return its id (reference: _waveform.pyx:264-271)."""
    global _next_type_id, _registry_generation
    type_id = _next_type_id
    _next_type_id += 1
    _baseFunc[type_id] = func
    _registry_generation += 1
    return type_id


def packBaseFunc():
    """reference: _waveform.pyx:274-275"""
    import pickle
    return pickle.dumps(_baseFunc)


def updateBaseFunc(buf):
    """reference: _waveform.pyx:278-279"""
    import pickle
    global _registry_generation
    _baseFunc.update(pickle.loads(buf))
    _registry_generation += 1


def registerDerivative(type_id, rule):
    """Register d/dt rule `rule(shift, *args) -> expr` (reference: _waveform.pyx:282-283)."""
    _ir.DERIVATIVE_RULES[type_id] = rule


def _rnd(x):
    return round(x, NDIGITS)


class Waveform:
    """Piecewise symbolic function of time: `bounds[i-1] <= t < bounds[i]` selects
    `seq[i]` (reference: waveforms/waveform.py:125-138)."""

    __slots__ = ('bounds', 'seq', 'max', 'min', 'start', 'stop', 'sample_rate',
                 'filters', 'label')

    def __init__(self, bounds=(+inf, ), seq=(ZERO, ), min=-inf, max=inf):
        self.bounds = bounds
        self.seq = seq
        self.max = max
        self.min = min
        self.start = None
        self.stop = None
        self.sample_rate = None
        self.filters = None
        self.label = None

    # ---- support interval ------------------------------------------------
    @staticmethod
    def _begin(bounds, seq):
        for i, e in enumerate(seq):
            if e != ZERO:
                return -inf if i == 0 else bounds[i - 1]
        return inf

    @staticmethod
    def _end(bounds, seq):
        for i in range(len(seq) - 1, -1, -1):
            if seq[i] != ZERO:
                return inf if i == len(seq) - 1 else bounds[i]
        return -inf

    @property
    def begin(self):
        b = self._begin(self.bounds, self.seq)
        return b if self.start is None else max(self.start, b)

    @property
    def end(self):
        e = self._end(self.bounds, self.seq)
        return e if self.stop is None else min(self.stop, e)

    # ---- sampling: the hot path -------------------------------------------
    def __call__(self, x, frag=False, out=None, accumulate=False,
                 function_lib=None):
        """Sample at the sorted times `x` on the GPU
        (reference: waveforms/waveform.py:529-563)."""
        from . import _sampling
        return _sampling.call_waveform(self, x, frag, out, accumulate,
                                       function_lib)

    def sample(self, sample_rate=None, out=None, chunk_size=None,
               function_lib=None, filters=None):
        """Sample on the grid np.arange(start, stop, 1/sample_rate)
        (reference: waveforms/waveform.py:173-257)."""
        from . import _sampling
        return _sampling.sample_waveform(self, sample_rate, out, chunk_size,
                                         function_lib, filters)

    # ---- flat encodings (reference: waveforms/waveform.py:259-382) ---------
    @staticmethod
    def _tolist(bounds, seq, ret=None):
        ret = [] if ret is None else ret
        ret.append(len(bounds))
        for expr, b in zip(seq, bounds):
            terms, amps = expr
            ret.append(b)
            ret.append(len(amps))
            for (factors, powers), amp in zip(terms, amps):
                ret.append(amp)
                ret.append(len(powers))
                for f, n in zip(factors, powers):
                    ret.append(n)
                    ret.append(len(f))
                    ret.extend(f)
        return ret

    @staticmethod
    def _fromlist(l, pos=0):
        def take(k):
            nonlocal pos
            chunk = tuple(l[pos:pos + k])
            if len(chunk) != k:
                raise ValueError('Invalid waveform format')
            pos += k
            return chunk

        (npieces, ) = take(1)
        bounds, seq = [], []
        for _ in range(npieces):
            b, nterms = take(2)
            bounds.append(b)
            terms, amps = [], []
            for _ in range(nterms):
                amp, nfac = take(2)
                amps.append(amp)
                factors, powers = [], []
                for _ in range(nfac):
                    n, flen = take(2)
                    powers.append(n)
                    factors.append(take(flen))
                terms.append((tuple(factors), tuple(powers)))
            seq.append((tuple(terms), tuple(amps)))
        return tuple(bounds), tuple(seq), pos

    def _header_filters(self, l):
        if self.filters is None:
            l.append(None)
        else:
            sos, initial = self.filters
            flat = list(np.asarray(sos).reshape(-1))
            l.append(len(flat))
            l.extend(flat)
            l.append(initial)

    def tolist(self):
        l = [self.max, self.min, self.start, self.stop, self.sample_rate]
        self._header_filters(l)
        return self._tolist(self.bounds, self.seq, l)

    @staticmethod
    def _read_filters(l, pos, nsos):
        if nsos is None:
            return None, pos
        sos = np.array(l[pos:pos + nsos]).reshape(-1, 6)
        return (sos, l[pos + nsos]), pos + nsos + 1

    @classmethod
    def fromlist(cls, l):
        w = cls()
        w.max, w.min, w.start, w.stop, w.sample_rate, nsos = l[:6]
        w.filters, pos = cls._read_filters(l, 6, nsos)
        w.bounds, w.seq, _ = cls._fromlist(l, pos)
        return w

    def totree(self):
        header = (self.max, self.min, self.start, self.stop, self.sample_rate,
                  self.filters)
        body = tuple(
            (b, tuple((amp, tuple((n, f) for f, n in zip(factors, powers)))
                      for (factors, powers), amp in zip(*expr)))
            for expr, b in zip(self.seq, self.bounds))
        return header, body

    @staticmethod
    def fromtree(tree):
        w = Waveform()
        header, body = tree
        w.max, w.min, w.start, w.stop, w.sample_rate, w.filters = header
        bounds, seq = [], []
        for b, pieces in body:
            bounds.append(b)
            terms = tuple((tuple(f for _, f in fl), tuple(n for n, _ in fl))
                          for _, fl in pieces)
            seq.append((terms, tuple(amp for amp, _ in pieces)))
        w.bounds, w.seq = tuple(bounds), tuple(seq)
        return w

    # ---- symbolic normal form (reference: waveforms/waveform.py:384-400, 418-482) ---
    def simplify(self, eps=1e-15):
        seq = [_ir.simplify(self.seq[0], eps)]
        bounds = [self.bounds[0]]
        for expr, b in zip(self.seq[1:], self.bounds[1:]):
            expr = _ir.simplify(expr, eps)
            if expr == seq[-1]:
                seq.pop()
                bounds.pop()
            seq.append(expr)
            bounds.append(b)
        return Waveform(tuple(bounds), tuple(seq))

    def filter(self, low=0, high=inf, eps=1e-15):
        return Waveform(self.bounds,
                        tuple(_ir.band_filter(e, low, high, eps) for e in self.seq))

    @property
    def marker(self):
        w = self.simplify()
        return Waveform(w.bounds, tuple(ZERO if e == ZERO else ONE for e in w.seq))

    def mask(self, edge: float = 0):
        """0/1 window over the support, widened by `edge` at the rising side of a run of non-zero
        pieces and narrowed by `edge` at the first zero piece after it (reference:
        waveforms/waveform.py:455-482, whose conventions -- the FIRST non-zero piece of a run sets the
        upper bound of the window, the FIRST zero piece after it the next one, piece 0 is only looked
        at when it is zero -- are kept; pinned by tests/golden/logic.json)."""
        from itertools import groupby
        m = self.marker
        live = [e != ZERO for e in m.seq]
        out = [] if live[0] else [(m.bounds[0] - edge, ZERO)]
        # runs of equal liveness over pieces 1..: only the first piece of a run acts.  The walk
        # starts "outside", so a leading run of zero pieces (or of live pieces after a live piece 0
        # -- the reference's state starts outside there too) is handled by the same two rules.
        outside = True
        for is_live, run in groupby(range(1, len(live)), key=live.__getitem__):
            first = next(run)
            if is_live and outside:
                out.append((m.bounds[first] + edge, ONE))
                outside = False
            elif not is_live and not outside:
                b = m.bounds[first] - edge
                if b <= out[-1][0]:
                    out[-1] = (b, out[-1][1])    # the narrowed window closes before it opened: pull its bound in
                else:
                    out.append((b, ZERO))
                outside = True
        return Waveform(tuple(b for b, _ in out), tuple(e for _, e in out))

    def __or__(self, other):
        if isinstance(other, (int, float, complex)):
            other = const(other)
        return self._comb(other, lambda a, b: ONE if (a != ZERO or b != ZERO) else ZERO)

    def __ior__(self, other):
        return self | other

    def __and__(self, other):
        if isinstance(other, (int, float, complex)):
            other = const(other)
        return self._comb(other, lambda a, b: ONE if (a != ZERO and b != ZERO) else ZERO)

    def __iand__(self, other):
        return self & other

    # ---- algebra (reference: waveforms/waveform.py:402-515) ----------------
    def _comb(self, other, oper):
        return Waveform(*_ir.combine_pieces(self.bounds, self.seq,
                                            other.bounds, other.seq, oper))

    def __pow__(self, n):
        return Waveform(self.bounds, tuple(_ir.power(e, n) for e in self.seq))

    def __add__(self, other):
        if isinstance(other, Waveform):
            return self._comb(other, _ir.add)
        return self + const(other)

    def __radd__(self, v):
        return const(v) + self

    def _scaled(self, v, wave_first):
        """self * const(v) (wave_first) or const(v) * self: the one-piece constant operand of
        combine_pieces, without building it (same products, same fusion of equal neighbours)."""
        cw = const(v)
        if len(cw.seq) != 1 or self.bounds[-1] != inf:
            return self._comb(cw, _ir.mul) if wave_first else cw._comb(self, _ir.mul)
        ce = cw.seq[0]
        bounds, seq = [], []
        mul = _ir.mul
        for b, e in zip(self.bounds, self.seq):
            e = mul(e, ce) if wave_first else mul(ce, e)
            if seq and e == seq[-1]:
                bounds[-1] = b
            else:
                bounds.append(b)
                seq.append(e)
        return Waveform(tuple(bounds), tuple(seq))

    def __mul__(self, other):
        if isinstance(other, Waveform):
            return self._comb(other, _ir.mul)
        return self._scaled(other, True)

    def __rmul__(self, v):
        return self._scaled(v, False)

    def __truediv__(self, other):
        if isinstance(other, Waveform):
            raise TypeError('division by waveform')
        return self * const(1 / other)

    def __neg__(self):
        return -1 * self

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, v):
        return v + (-self)

    def __rshift__(self, time):
        return Waveform(tuple(_rnd(b + time) for b in self.bounds),
                        tuple(_ir.shift(e, time) for e in self.seq))

    def __lshift__(self, time):
        return self >> (-time)

    def __hash__(self):
        return hash((self.max, self.min, self.start, self.stop,
                     self.sample_rate, self.bounds, self.seq))

    def __eq__(self, o):
        """Equality of the simplified trees (reference: waveforms/waveform.py:569-579)."""
        if isinstance(o, (int, float, complex)):
            return self == const(o)
        if isinstance(o, Waveform):
            a, b = self.simplify(), o.simplify()
            return a.seq == b.seq and a.bounds == b.bounds and (
                a.max, a.min, a.start, a.stop) == (b.max, b.min, b.start, b.stop)
        return False


class WaveVStack(Waveform):
    """Lazy sum of many piecewise waveforms into ONE output channel
    (reference: waveforms/waveform.py:638-844)."""

    def __init__(self, wlist: list[Waveform] = []):
        self.wlist = [(w.bounds, w.seq) for w in wlist]
        self.start = None
        self.stop = None
        self.sample_rate = None
        self.offset = 0
        self.shift = 0
        self.filters = None
        self.label = None
        self.function_lib = None

    @property
    def begin(self):
        b = min((self._begin(*m) for m in self.wlist), default=-inf)
        return b if self.start is None else max(self.start, b)

    @property
    def end(self):
        e = max((self._end(*m) for m in self.wlist), default=inf)
        return e if self.stop is None else min(self.stop, e)

    def __call__(self, x, frag=False, out=None, function_lib=None):
        """real(offset + sum of members at x - shift), on the GPU
        (reference: waveforms/waveform.py:679-693)."""
        assert frag is False, 'WaveVStack does not support frag mode'
        from . import _sampling
        return _sampling.call_vstack(self, x, function_lib)

    def simplify(self, eps=1e-15):
        """Collapse the stack into one simplified Waveform
        (reference: waveforms/waveform.py:734-749)."""
        if not self.wlist:
            return zero()
        wav = Waveform(*_ir.wave_sum(self.wlist))
        if self.offset != 0:
            wav += self.offset
        if self.shift != 0:
            wav >>= self.shift
        wav = wav.simplify(eps)
        wav.start, wav.stop, wav.sample_rate = self.start, self.stop, self.sample_rate
        wav.filters, wav.label = self.filters, self.label
        return wav

    def tolist(self):
        l = [self.start, self.stop, self.offset, self.shift, self.sample_rate]
        self._header_filters(l)
        l.append(len(self.wlist))
        for bounds, seq in self.wlist:
            self._tolist(bounds, seq, l)
        return l

    @classmethod
    def fromlist(cls, l):
        w = cls()
        w.start, w.stop, w.offset, w.shift, w.sample_rate, nsos = l[:6]
        w.filters, pos = cls._read_filters(l, 6, nsos)
        count = l[pos]
        pos += 1
        for _ in range(count):
            bounds, seq, pos = cls._fromlist(l, pos)
            w.wlist.append((bounds, seq))
        return w

    def _clone_meta(self, ret):
        ret.filters = self.filters
        ret.label = self.label
        return ret

    @staticmethod
    def _rshift(wlist, time):
        if time == 0:
            return wlist
        return [(tuple(_rnd(b + time) for b in bounds),
                 tuple(_ir.shift(e, time) for e in seq))
                for bounds, seq in wlist]

    def __rshift__(self, time):
        ret = WaveVStack()
        ret.wlist = self.wlist
        ret.sample_rate, ret.start, ret.stop = (self.sample_rate, self.start,
                                                self.stop)
        ret.shift = self.shift + time
        ret.offset = self.offset
        return self._clone_meta(ret)

    def __add__(self, other):
        # A sum keeps its members (reference waveform.py:771-785).  The result is a NEW stack with no pending shift and the
        # offset the reference gives it: another stack's members join (both sides brought to absolute time first when
        # their pending shifts differ) and the offsets add; a Waveform joins as one more member, moved back by this
        # stack's pending shift; a number becomes the offset.
        out = WaveVStack()
        if isinstance(other, WaveVStack):
            aligned = other.shift == self.shift
            mine = self.wlist if aligned else self._rshift(self.wlist, self.shift)
            theirs = other.wlist if aligned else self._rshift(other.wlist, other.shift)
            out.wlist = [*mine, *theirs]
            out.offset = self.offset + other.offset
        elif isinstance(other, Waveform):
            moved = other << self.shift
            out.wlist = [*self.wlist, (moved.bounds, moved.seq)]
        else:
            out.wlist = list(self.wlist)
            out.offset = out.offset + other
        return self._clone_meta(out)

    def __radd__(self, v):
        return self + v

    def __mul__(self, other):
        if isinstance(other, Waveform):
            other = other.simplify() << self.shift
            ret = WaveVStack([Waveform(*m) * other for m in self.wlist])
            if self.offset != 0:
                w = other * self.offset
                ret.wlist.append((w.bounds, w.seq))
        else:
            ret = WaveVStack([Waveform(*m) * other for m in self.wlist])
            ret.offset = self.offset * other
        return self._clone_meta(ret)

    def __rmul__(self, v):
        return self * v

    def __eq__(self, other):
        if self.wlist:
            return False
        return zero() == other

    __hash__ = None

    def __getstate__(self):
        # the function library travels pickled, when it can be (reference: waveform.py:823-833
        # uses dill; plain pickle here -- dill is not a dependency of this package)
        function_lib = self.function_lib
        if function_lib:
            try:
                import pickle
                function_lib = pickle.dumps(function_lib)
            except Exception:
                function_lib = None
        return (self.wlist, self.start, self.stop, self.sample_rate,
                self.offset, self.shift, self.filters, self.label, function_lib or None)

    def __setstate__(self, state):
        (self.wlist, self.start, self.stop, self.sample_rate, self.offset,
         self.shift, self.filters, self.label, function_lib) = state
        if function_lib:
            try:
                import pickle
                function_lib = pickle.loads(function_lib)
            except Exception:
                function_lib = None
        self.function_lib = function_lib or None


# --------------------------------------------------------------------------
# constructors (reference: waveforms/waveform.py:886-896, 1055-1527)
# --------------------------------------------------------------------------
def zero():
    # (a fresh object per call: callers set .start / .stop / .sample_rate / .min / .max on what they get)
    return Waveform()


def one():
    return Waveform(seq=(ONE, ))


def const(c):
    return Waveform(seq=(const_expr(1.0 * c), ))


def D(wav: Waveform, d: int = 1) -> Waveform:
    """d-th symbolic time derivative."""
    assert d >= 0 and isinstance(d, int), "d must be a non-negative integer"
    for _ in range(d):
        wav = Waveform(bounds=wav.bounds,
                       seq=tuple(_ir.derivative(e) for e in wav.seq))
    return wav


def convolve(a, b):
    """Placeholder in the reference as well (waveform.py:1074-1075: `pass`)."""
    return None


def sign():
    return Waveform(bounds=(0, +inf), seq=(const_expr(-1), ONE))


def _window(lo, hi, inner):
    return Waveform(bounds=(lo, hi, +inf), seq=(ZERO, inner, ZERO))


def step(edge, type='erf'):
    """Unit step with an "erf", "cos" or "linear" edge of width `edge`."""
    if edge == 0:
        return Waveform(bounds=(0, +inf), seq=(ZERO, ONE))
    if type == 'cos':
        rise = _ir.add(HALF, _ir.mul(HALF,
                                     primitive(COS, pi / edge, shift=0.5 * edge)))
        lo, hi = _rnd(-edge / 2), _rnd(edge / 2)
    elif type == 'linear':
        rise = _ir.add(HALF, _ir.mul(const_expr(1 / edge), primitive(LINEAR)))
        lo, hi = _rnd(-edge / 2), _rnd(edge / 2)
    else:
        rise = ((((), ()), (((ERF, edge / 5, 0), ), (1, ))), (0.5, 0.5))
        lo, hi = -_rnd(edge), _rnd(edge)
    return Waveform(bounds=(lo, hi, +inf), seq=(ZERO, rise, ONE))


def square(width: float, edge: float = 0, type: str = 'erf') -> Waveform:
    if width <= 0:
        return zero()
    if edge == 0:
        return _window(_rnd(-0.5 * width), _rnd(0.5 * width), ONE)
    return ((step(edge, type=type) << width / 2) -
            (step(edge, type=type) >> width / 2))


def gaussian(width: float, plateau: float = 0.0, d: int | None = None):
    """Gaussian with `width` = 2 x FWHM, truncated at +-0.75 width."""
    if width <= 0 and plateau <= 0.0:
        return zero()
    sigma = width / 3.3302184446307908  # width / (4 sqrt(ln 2))
    if d is None:
        def bell(s):
            return primitive(GAUSSIAN, sigma, shift=s)
    else:
        def bell(s):
            return primitive(D_GAUSSIAN, sigma, d, shift=s)
    if _rnd(0.5 * plateau) <= 0.0:
        return _window(_rnd(-0.75 * width), _rnd(0.75 * width), bell(0))
    return Waveform(bounds=(_rnd(-0.75 * width - 0.5 * plateau),
                            _rnd(-0.5 * plateau), _rnd(0.5 * plateau),
                            _rnd(0.75 * width + 0.5 * plateau), +inf),
                    seq=(ZERO, bell(-0.5 * plateau), ONE, bell(0.5 * plateau),
                         ZERO))


def cos(w: float, phi: float = 0) -> Waveform:
    if w == 0:
        return const(np.cos(phi))
    if w < 0:
        w, phi = -w, -phi
    return Waveform(seq=(primitive(COS, w, shift=-phi / w), ))


def sin(w: float, phi: float = 0) -> Waveform:
    if w == 0:
        return const(np.sin(phi))
    if w < 0:
        w, phi = -w, -phi + pi
    return Waveform(seq=(primitive(COS, w, shift=(pi / 2 - phi) / w), ))


def exp(alpha) -> Waveform:
    if isinstance(alpha, complex):
        carrier = cos(alpha.imag) + 1j * sin(alpha.imag)
        return carrier if alpha.real == 0 else exp(alpha.real) * carrier
    return Waveform(seq=(primitive(EXP, alpha), ))


def sinc(bw: float) -> Waveform:
    if bw <= 0:
        return zero()
    width = 100 / bw
    return _window(_rnd(-0.5 * width), _rnd(0.5 * width), primitive(SINC, bw))


def cosPulse(width: float, plateau: float = 0.0) -> Waveform:
    if _rnd(0.5 * plateau) > 0:
        return square(plateau + 0.5 * width, edge=0.5 * width, type='cos')
    if width <= 0:
        return zero()
    bump = ((((), ()), (((COS, 6.283185307179586 / width, 0), ), (1, ))),
            (0.5, 0.5))
    return _window(_rnd(-0.5 * width), _rnd(0.5 * width), bump)


def hanning(width: float, plateau: float = 0.0) -> Waveform:
    return cosPulse(width, plateau=plateau)


def cosh(w: float) -> Waveform:
    return Waveform(seq=(primitive(COSH, w), ))


def sinh(w: float) -> Waveform:
    return Waveform(seq=(primitive(SINH, w), ))


def coshPulse(width: float, eps: float = 1.0, plateau: float = 0.0):
    """(cosh(eps/2) - cosh(eps t / width)) / (cosh(eps/2) - 1) on |t| < width/2,
    optionally split around a unit plateau."""
    if width <= 0 and plateau <= 0:
        return zero()
    w = eps / width
    A = np.cosh(eps / 2)
    amps = (A / (A - 1), -1 / (A - 1))

    def edge(s):
        return ((((), ()), (((COSH, w, s), ), (1, ))), amps)

    if plateau == 0.0 or _rnd(-0.5 * plateau) == _rnd(0.5 * plateau):
        return _window(_rnd(-0.5 * width), _rnd(0.5 * width), edge(0))
    return Waveform(bounds=(_rnd(-0.5 * width - 0.5 * plateau),
                            _rnd(-0.5 * plateau), _rnd(0.5 * plateau),
                            _rnd(0.5 * width + 0.5 * plateau), +inf),
                    seq=(ZERO, edge(-0.5 * plateau), ONE, edge(0.5 * plateau),
                         ZERO))


def general_cosine(duration: float, *arg: float) -> Waveform:
    coef = np.asarray(arg)
    coef /= coef[::2].sum()
    wav = zero()
    for i, a in enumerate(coef, start=1):
        wav += a / 2 * (1 - (-1)**i * cos(i * 2 * pi / duration))
    return wav * square(duration)


def slepian(duration: float, *arg: float) -> Waveform:
    return general_cosine(duration, *arg)


def mollifier(width: float, plateau: float = 0.0, d: int = 0) -> Waveform:
    """exp(1/((t/r)^2 - 1) + 1) on |t| < r = width/2 (d-th derivative)."""
    assert d >= 0 and isinstance(d, int), "d must be a non-negative integer"
    assert width > 0, "width must be positive"
    r = width / 2
    if plateau <= 0:
        return _window(-0.5 * width, 0.5 * width, primitive(MOLLIFIER, r, d))
    return Waveform(bounds=(-0.5 * width - 0.5 * plateau, -0.5 * plateau,
                            0.5 * plateau, 0.5 * width + 0.5 * plateau, inf),
                    seq=(ZERO, primitive(MOLLIFIER, r, d,
                                         shift=-0.5 * plateau), ONE,
                         primitive(MOLLIFIER, r, d, shift=0.5 * plateau), ZERO))


def poly(a):
    """a[0] + a[1] t + a[2] t^2 + ...  The amplitude tuple is *all* of `a`,
    including zero coefficients whose terms were skipped -- the reference's
    alignment quirk (waveforms/waveform.py:1333)."""
    terms = []
    if a[0] != 0:
        terms.append(((), ()))
    for n, c in enumerate(a[1:], start=1):
        if c != 0:
            terms.append((((LINEAR, 0), ), (n, )))
    return Waveform(seq=((tuple(terms), tuple(a)), ))


def t():
    return Waveform(seq=((((LINEAR, 0), ), (1, )), (1, )))


def drag(freq: float, width: float, plateau: float = 0, delta: float = 0,
         block_freq: float | None = None, phase: float = 0, t0: float = 0):
    phase += pi * delta * (width + plateau)
    if plateau <= 0:
        return _window(_rnd(t0), _rnd(t0 + width),
                       primitive(DRAG, t0, freq, width, delta, block_freq,
                                 phase))
    w = 2 * pi * (freq + delta)
    hold = primitive(COS, w, shift=(phase + 2 * pi * delta * t0) / w)
    if width <= 0:
        return _window(_rnd(t0), _rnd(t0 + plateau), hold)
    return Waveform(
        seq=(ZERO, primitive(DRAG, t0, freq, width, delta, block_freq, phase),
             hold,
             primitive(DRAG, t0 + plateau, freq, width, delta, block_freq,
                       phase - 2 * pi * delta * plateau), ZERO),
        bounds=(_rnd(t0), _rnd(t0 + width / 2), _rnd(t0 + width / 2 + plateau),
                _rnd(t0 + width + plateau), +inf))


def chirp(f0: float, f1: float, T: float, phi0: float = 0,
          type: str = 'linear') -> Waveform:
    if f0 == f1:
        return sin(f0, phi0)
    if T <= 0:
        raise ValueError('T must be positive')
    if type == 'linear':
        body = primitive(LINEARCHIRP, f0, f1, T, phi0)
    elif type in ('exp', 'exponential', 'geometric'):
        if f0 == 0:
            raise ValueError('f0 must be non-zero')
        body = primitive(EXPONENTIALCHIRP, f0, np.log(f1 / f0) / T, phi0)
    elif type in ('hyperbolic', 'hyp'):
        if f0 * f1 == 0:
            return const(np.sin(phi0))
        body = primitive(HYPERBOLICCHIRP, f0, (f0 - f1) / (f1 * T), phi0)
    else:
        raise ValueError(f'unknown type {type}')
    return _window(0, _rnd(T), body)


def interp(x, y) -> Waveform:
    """Piecewise-linear waveform through the points (x, y)
    (reference: waveforms/waveform.py:1425-1440)."""
    seq, bounds = [ZERO], [x[0]]
    for x1, x2, y1, y2 in zip(x[:-1], x[1:], y[:-1], y[1:]):
        if x2 == x1:
            continue
        seq.append(_ir.add(_ir.mul(const_expr((y2 - y1) / (x2 - x1)),
                                   primitive(LINEAR, shift=x1)), const_expr(y1)))
        bounds.append(x2)
    bounds.append(inf)
    seq.append(ZERO)
    return Waveform(seq=tuple(seq), bounds=tuple(_rnd(b) for b in bounds)).simplify()


def cut(wav: Waveform, start=None, stop=None, head=None, tail=None, min=None,
        max=None) -> Waveform:
    # reference waveform.py:1443-1466: pin the level at one end (`head` at `start`; failing that `tail` at `stop`), gate
    # the result to [start, stop), set the clip bounds -- always on a new Waveform (`+ 0` when nothing is pinned)
    def level(t):
        return wav(np.array([1.0 * t]))[0]

    if start is not None and head is not None:
        lift = head - level(start)
    elif stop is not None and tail is not None:
        lift = tail - level(stop)
    else:
        lift = 0
    out = wav + lift
    for edge, gate in ((start, step(0)), (stop, 1 - step(0))):
        if edge is not None:
            out = out * (gate >> edge)
    if min is not None:
        out.min = min
    if max is not None:
        out.max = max
    return out


def function(fun, *args, start=None, stop=None):
    wav = Waveform(seq=(primitive(registerBaseFunc(fun), *args), ))
    if start is not None:
        wav = wav * (step(0) >> start)
    if stop is not None:
        wav = wav * ((1 - step(0)) >> stop)
    return wav


def samplingPoints(start, stop, points):
    return _window(_rnd(start), _rnd(stop),
                   primitive(INTERP, start, stop, tuple(points)))


def mixing(I: Waveform, Q: Waveform | None = None, *, phase: float = 0.0,
           freq: float = 0.0, ratioIQ: float = 1.0, phaseDiff: float = 0.0,
           block_freq: float | None = None,
           DRAGScaling: float | None = None) -> tuple[Waveform, Waveform]:
    """SSB (freq != 0) or envelope mixing of an I/Q pair, with optional DRAG
    correction built from the symbolic derivative
    (reference: waveforms/waveform.py:1487-1527)."""
    w = 2 * pi * freq
    if Q is None:
        # (the reference multiplies a zero waveform through and adds it: products with ZERO are ZERO
        #  and adding the zero waveform changes neither bounds nor pieces -- skipped here)
        if freq != 0.0:
            Iout = I * cos(w, -phase)
            Qout = -I * sin(w, -phase + phaseDiff)
        else:
            Iout = I * np.cos(-phase)
            Qout = -I * np.sin(-phase)
    elif freq != 0.0:
        Iout = I * cos(w, -phase) + Q * sin(w, -phase)
        Qout = -I * sin(w, -phase + phaseDiff) + Q * cos(w, -phase + phaseDiff)
    else:
        Iout = I * np.cos(-phase) + Q * np.sin(-phase)
        Qout = -I * np.sin(-phase) + Q * np.cos(-phase)

    if block_freq is not None and block_freq != freq:
        a = block_freq / (block_freq - freq)
        b = 1 / (block_freq - freq)
        Iout, Qout = (a * Iout + b / (2 * pi) * D(Qout),
                      a * Qout - b / (2 * pi) * D(Iout))
    elif DRAGScaling is not None and DRAGScaling != 0:
        Iout, Qout = ((1 - w * DRAGScaling) * Iout - DRAGScaling * D(Qout),
                      (1 - w * DRAGScaling) * Qout + DRAGScaling * D(Iout))
    return Iout, ratioIQ * Qout


__all__ = [
    'D', 'Waveform', 'WaveVStack', 'chirp', 'const', 'cos', 'cosh', 'coshPulse',
    'cosPulse', 'cut', 'drag', 'exp', 'function', 'gaussian', 'general_cosine',
    'hanning', 'interp', 'mixing', 'mollifier', 'one', 'poly', 'registerBaseFunc',
    'registerDerivative', 'samplingPoints', 'sign', 'sin', 'sinc', 'sinh',
    'slepian', 'square', 'step', 't', 'zero'
]
