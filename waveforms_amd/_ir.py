"""Symbolic tuple IR for piecewise sum-of-products pulse expressions (host side).

This is the front-end half of the hot path's *input*: the nested-tuple expression
tree that `Waveform.__call__` hands to the sampler.  It never touches samples; all
of it is O(#terms) Python.  The data model is the reference's public one
(reference: waveforms/_waveform.pyx:15,29-48 and SURVEY.md Appendix A):

    expr   := (terms, amps)          parallel tuples;   ZERO = ((), ())
    term   := (factors, powers)      parallel tuples;   constant term = ((), ())
    factor := (type_id, *args, shift)

Terms of an expr and factors of a term are kept in *windowed insertion order*
(see `_merge_into`), which coincides with sorted order for canonical inputs; the
flat lists produced from these tuples are pinned element-for-element against the
reference by tests/test_frontend_golden.py.
"""
from __future__ import annotations

import itertools
import math
from bisect import bisect_left

import numpy as np

NDIGITS = 15

# Primitive type ids: identical numbering to the reference registry
# (reference: waveforms/_waveform.pyx:374-388; multy_drag.py registers 16 and 17).
LINEAR = 1
GAUSSIAN = 2
ERF = 3
COS = 4
SINC = 5
EXP = 6
INTERP = 7
LINEARCHIRP = 8
EXPONENTIALCHIRP = 9
HYPERBOLICCHIRP = 10
COSH = 11
SINH = 12
DRAG = 13
MOLLIFIER = 14
D_GAUSSIAN = 15
DRAG_SIN = 16   # reserved (multy_drag)
DRAG_SINX = 17  # reserved
FIRST_USER_TYPE = 18

PRIMITIVE_NAMES = {
    LINEAR: "LINEAR", GAUSSIAN: "GAUSSIAN", ERF: "ERF", COS: "COS", SINC: "SINC",
    EXP: "EXP", INTERP: "INTERP", LINEARCHIRP: "LINEARCHIRP",
    EXPONENTIALCHIRP: "EXPONENTIALCHIRP", HYPERBOLICCHIRP: "HYPERBOLICCHIRP",
    COSH: "COSH", SINH: "SINH", DRAG: "DRAG", MOLLIFIER: "MOLLIFIER",
    D_GAUSSIAN: "D_GAUSSIAN", DRAG_SIN: "DRAG_SIN", DRAG_SINX: "DRAG_SINX",
}

ZERO = ((), ())
_UNIT_TERM = ((), ())  # the term with no factors: multiplies to 1


def const_expr(c):
    """Expression for the number `c` (reference: _waveform.pyx:29-32)."""
    if c == 0:
        return ZERO
    return ((_UNIT_TERM, ), (c, ))


ONE = const_expr(1.0)
_UNIT_KEYS = (_UNIT_TERM, )
HALF = const_expr(1 / 2)


def is_const(expr) -> bool:
    """True for ZERO and for a single constant term (reference: _waveform.pyx:43-44)."""
    return expr == ZERO or expr[0] == (_UNIT_TERM, )


def primitive(type_id, *args, shift=0):
    """1.0 * type_id(t - shift, *args) as an expression (reference: _waveform.pyx:47-48)."""
    return ((((type_id, *args, shift), ), (1, )), ), (1.0, )


def _merge_into(keys: list, vals: list, key, val, lo: int, hi: int):
    """Insert (key, val) into the parallel lists inside the window [lo, hi).

    Leftmost position in the window whose key is >= `key` is the slot.  An equal
    key merges values (entry removed if the sum is exactly 0); otherwise a new
    entry is inserted.  Returns the slot and the new window end.  The *caller*
    narrows the window to start at the returned slot, which is what makes the
    order "windowed" rather than globally sorted when inputs arrive out of order
    (reference behaviour: _waveform.pyx:51-65).
    """
    slot = bisect_left(keys, key, lo, hi)
    if slot < hi and keys[slot] == key:
        merged = val + vals[slot]
        if merged == 0:
            del keys[slot]
            del vals[slot]
            return slot, hi - 1
        vals[slot] = merged
        return slot, hi
    keys.insert(slot, key)
    vals.insert(slot, val)
    return slot, hi + 1


def add(x, y):
    """x + y on (keys, values) pairs: used for expr+expr (terms/amps) and, by
    `mul`, for term*term (factors/powers) (reference: _waveform.pyx:82-88)."""
    kx, ky = x[0], y[0]
    if type(kx) is tuple and type(ky) is tuple:     # (malformed operands take the long way and
        if not ky:                                  #  fail exactly where the reference fails)
            return x                                # nothing to merge in
        if not kx:
            # merging into nothing reproduces y whenever y's keys already ascend (then every
            # insertion lands at the end of the window): the case of "zero piece + pulse"
            n = len(ky)
            if n == 1 or all(ky[i] < ky[i + 1] for i in range(n - 1)):
                return y
    keys, vals = list(x[0]), list(x[1])
    lo, hi = 0, len(keys)
    for k, v in zip(ky, y[1]):
        lo, hi = _merge_into(keys, vals, k, v, lo, hi)
    return tuple(keys), tuple(vals)


def _ascending(keys) -> bool:
    n = len(keys)
    if n < 2:
        return True
    try:
        return all(keys[i] < keys[i + 1] for i in range(n - 1))
    except TypeError:               # (malformed operands: take the long way, fail where the reference fails)
        return False


def _scaled(keys, amps, c, amps_first):
    """(keys, amps * c) without the zero products; amps_first: the product is amp * c, else c * amp
    (the operand order of the reference's loop, which matters for complex / float rounding)."""
    out_k, out_v = [], []
    for k, a in zip(keys, amps):
        v = a * c if amps_first else c * a
        if v == 0:
            continue
        out_k.append(k)
        out_v.append(v)
    if len(out_k) == len(keys):
        return keys, tuple(out_v)
    return tuple(out_k), tuple(out_v)


def mul(x, y):
    """Product of two expressions, distributing over terms
    (reference: _waveform.pyx:68-79)."""
    kx, ky = x[0], y[0]
    if type(kx) is tuple and type(ky) is tuple:
        if not kx or not ky:
            return ZERO             # a zero operand: no term survives
        # A constant operand (one unit term) scales the other's amplitudes: every product term is the
        # other operand's own term object (add() with the empty term returns it), and with ascending
        # keys every insertion lands at the end of the window -- the loop below would rebuild the same
        # tuples.  Zero products drop out, as there.
        if kx == _UNIT_KEYS and _ascending(ky):
            return _scaled(ky, y[1], x[1][0], False)
        if ky == _UNIT_KEYS and _ascending(kx):
            return _scaled(kx, x[1], y[1][0], True)
    keys, vals = [], []
    lo = hi = 0
    for (tx, ty), (ax, ay) in zip(itertools.product(x[0], y[0]),
                                  itertools.product(x[1], y[1])):
        amp = ax * ay
        if amp == 0:
            continue
        lo, hi = _merge_into(keys, vals, add(tx, ty), amp, lo, hi)
    return tuple(keys), tuple(vals)


def shift(expr, dt):
    """expr(t - dt): every factor's trailing shift grows by dt; constants are
    returned unchanged (reference: _waveform.pyx:91-102)."""
    if is_const(expr):
        return expr
    moved = []
    for factors, powers in expr[0]:
        moved.append((tuple((*f[:-1], f[-1] + dt) for f in factors), powers))
    return tuple(moved), expr[1]


def power(expr, n):
    """expr ** n (reference: _waveform.pyx:105-127)."""
    if expr == ZERO:
        return ZERO
    if n == 0:
        return ONE
    if is_const(expr):
        return const_expr(expr[1][0]**n)
    if len(expr[0]) == 1:
        (factors, powers), amp = expr[0][0], expr[1][0]
        return (((factors, tuple(n * p for p in powers)), ), (amp**n, ))
    assert isinstance(n, int) and n > 0
    out = ONE
    for _ in range(n):
        out = mul(out, expr)
    return out


def combine_pieces(b1, s1, b2, s2, oper):
    """Pointwise `oper` of two piecewise expressions; adjacent equal pieces are
    fused (reference: _waveform.pyx:216-235)."""
    bounds, seq = [], []
    n1, n2 = len(b1), len(b2)
    # one operand is a single piece up to +inf (a constant, a carrier): the other's bounds survive
    if n2 == 1 and b2[0] == math.inf and n1 >= 1 and b1[-1] == math.inf:
        e2 = s2[0]
        for b, e1 in zip(b1, s1):
            e = oper(e1, e2)
            if seq and e == seq[-1]:
                bounds[-1] = b
            else:
                bounds.append(b)
                seq.append(e)
        return tuple(bounds), tuple(seq)
    if n1 == 1 and b1[0] == math.inf and n2 >= 1 and b2[-1] == math.inf:
        e1 = s1[0]
        for b, e2 in zip(b2, s2):
            e = oper(e1, e2)
            if seq and e == seq[-1]:
                bounds[-1] = b
            else:
                bounds.append(b)
                seq.append(e)
        return tuple(bounds), tuple(seq)
    i = j = 0
    while i < n1 or j < n2:
        e = oper(s1[i], s2[j])
        b = min(b1[i], b2[j])
        if seq and e == seq[-1]:
            bounds[-1] = b
        else:
            bounds.append(b)
            seq.append(e)
        if b == b1[i]:
            i += 1
        if b == b2[j]:
            j += 1
    return tuple(bounds), tuple(seq)


def wave_sum(waves):
    """Sum a list of (bounds, seq) piecewise expressions into one
    (reference: _waveform.pyx:172-213)."""
    if not waves:
        return ((+math.inf, ), (ZERO, ))
    bounds, seq = waves[0]
    if len(waves) == 1:
        return bounds, seq
    bounds, seq = list(bounds), list(seq)
    for ob, os_ in waves[1:]:
        if len(ob) == 1:
            seq = [add(s, os_[0]) for s in seq]
        elif len(bounds) == 1:
            head = seq[0]
            bounds = list(ob)
            seq = [add(head, s) for s in os_]
        else:
            lo = 0
            for b, s in zip(ob, os_):
                i = bisect_left(bounds, b, lo=lo)
                if bounds[i] > b:
                    bounds.insert(i, b)
                    seq.insert(i, s if i == 0 else add(s, seq[i]))
                    last = i - 1
                else:
                    last = i
                for k in range(lo + 1, last + 1):
                    seq[k] = add(seq[k], s)
                lo = i
    i = 0
    while i < len(bounds) - 1:
        if seq[i] == seq[i + 1]:
            del seq[i]
            del bounds[i]
        else:
            i += 1
    return tuple(bounds), tuple(seq)


# --------------------------------------------------------------------------
# Symbolic derivative (used by D() and mixing(DRAGScaling=...)).
# Per-primitive rules: reference _waveform.pyx:391-463.
# --------------------------------------------------------------------------

def _single(factors, powers, amp):
    return (((tuple(factors), tuple(powers)), ), (amp, ))


def _rule_LINEAR(s, *a):
    return ONE


def _rule_GAUSSIAN(s, sigma):
    return _single([(LINEAR, s), (GAUSSIAN, sigma, s)], [1, 1], -2 / sigma**2)


def _rule_ERF(s, sigma):
    return _single([(GAUSSIAN, sigma, s)], [1], 2 / sigma / np.sqrt(math.pi))


def _rule_COS(s, w):
    return _single([(COS, w, s - math.pi / w / 2)], [1], w)


def _rule_SINC(s, *a):
    # d/dt sinc is expressed by the reference with a 4-field COS factor that no
    # evaluator accepts (reference: _waveform.pyx:410-413); kept for tuple parity.
    return (((((LINEAR, s), (COS, *a, s)), (-1, 1)),
             (((LINEAR, s), (COS, a[0], a[1] - math.pi / 2, s)), (-2, 1))),
            (1, -1 / a[0]))


def _rule_EXP(s, alpha):
    return _single([(EXP, alpha, s)], [1], alpha)


def _rule_INTERP(s, start, stop, points):
    grad = tuple(np.gradient(np.asarray(points)))
    return _single([(INTERP, start, stop, grad, s)], [1],
                   (len(points) - 1) / (stop - start))


def _rule_COSH(s, w):
    return _single([(SINH, w, s)], [1], w)


def _rule_SINH(s, w):
    return _single([(COSH, w, s)], [1], w)


def _rule_LINEARCHIRP(s, f0, f1, T, phi0):
    quad = (LINEARCHIRP, f0, f1, T, phi0 + math.pi / 2, s)
    terms = ((((quad, ), (1, ))), (((LINEAR, s), quad), (1, 1)))
    amps = (2 * math.pi * f0, 2 * math.pi * (f1 - f0) / T)
    if f0 == 0:
        return terms[1:], amps[1:]
    return terms, amps


def _rule_EXPONENTIALCHIRP(s, f0, alpha, phi0):
    return _single([(EXP, alpha, s),
                    (EXPONENTIALCHIRP, f0, alpha, phi0 + math.pi / 2, s)],
                   [1, 1], 2 * math.pi * f0)


def _rule_HYPERBOLICCHIRP(s, f0, k, phi0):
    return _single([(LINEAR, s - 1 / k),
                    (HYPERBOLICCHIRP, f0, k, phi0 + math.pi / 2, s)], [-1, 1],
                   2 * math.pi * f0)


def _rule_MOLLIFIER(s, r, d):
    return _single([(MOLLIFIER, r, d + 1, s)], [1], 1)


def _rule_D_GAUSSIAN(s, sigma, n):
    return _single([(D_GAUSSIAN, sigma, n + 1, s)], [1], 1)


DERIVATIVE_RULES = {
    LINEAR: _rule_LINEAR, GAUSSIAN: _rule_GAUSSIAN, ERF: _rule_ERF,
    COS: _rule_COS, SINC: _rule_SINC, EXP: _rule_EXP, INTERP: _rule_INTERP,
    COSH: _rule_COSH, SINH: _rule_SINH, LINEARCHIRP: _rule_LINEARCHIRP,
    EXPONENTIALCHIRP: _rule_EXPONENTIALCHIRP,
    HYPERBOLICCHIRP: _rule_HYPERBOLICCHIRP, MOLLIFIER: _rule_MOLLIFIER,
    D_GAUSSIAN: _rule_D_GAUSSIAN,
}


def _d_factor(factor):
    type_id, *args, s = factor
    return DERIVATIVE_RULES[type_id](s, *args)


def derivative(expr):
    """d/dt of an expression: sum rule, product rule, power rule, then the
    per-primitive table (reference: _waveform.pyx:243-261)."""
    if is_const(expr):
        return ZERO
    terms, amps = expr
    if len(amps) > 1:
        return add(derivative((terms[:1], amps[:1])),
                   derivative((terms[1:], amps[1:])))
    (factors, powers), amp = terms[0], amps[0]
    if len(factors) > 1:
        head = (((factors[:1], powers[:1]), ), (amp, ))
        tail = (((factors[1:], powers[1:]), ), (1, ))
        return add(mul(head, derivative(tail)), mul(derivative(head), tail))
    f, n = factors[0], powers[0]
    if n == 1:
        return mul(_d_factor(f), const_expr(amp))
    lowered = ((((f, ), (n - 1, )), ), (n * amp, ))
    return mul(lowered, derivative(((((f, ), (1, )), ), (1, ))))


# --------------------------------------------------------------------------
# Symbolic normal form: simplify / filter.
# Same results, term for term, as the reference (waveforms/_waveform.pyx:483-654):
# products of cosines become sums (product-to-sum), EXP factors merge, Gaussian powers
# fold into the width, and terms with the same (non-cosine part, frequency) are merged
# into one cosine per real / imaginary amplitude.
# --------------------------------------------------------------------------
def _binom(n, k):
    return math.comb(n, k) if 0 <= k <= n else 0


def _cos_pow(factor, n):
    """cos(w (t - s))**n as a sum of cosines of multiples of w."""
    _, w, s = factor
    out = ZERO
    for k in range(n // 2 + 1):
        if n == 2 * k:
            out = add(out, const_expr(_binom(n, k) / 2**n))
        else:
            out = add(out, ((((((COS, (n - 2 * k) * w, s), ), (1, )), ),
                             (_binom(n, k) / 2**(n - 1), ))))
    return out


def _cos_times_cos(x, y, v):
    """v cos(a) cos(b) = v/2 cos(a+b) + v/2 cos(a-b), lower frequency first."""
    _, w1, t1 = x
    _, w2, t2 = y
    if w2 > w1:
        w1, w2, t1, t2 = w2, w1, t2, t1
    hi = (COS, w1 + w2, (w1 * t1 + w2 * t2) / (w1 + w2))
    if w1 == w2:
        c = v * np.cos(w1 * t1 - w2 * t2) / 2
        if c == 0:
            return (((hi, ), (1, )), ), (0.5 * v, )
        return (((), ()), ((hi, ), (1, ))), (c, 0.5 * v)
    lo = (COS, w1 - w2, (w1 * t1 - w2 * t2) / (w1 - w2))
    if lo[1] > hi[1]:
        lo, hi = hi, lo
    return (((lo, ), (1, )), ((hi, ), (1, ))), (0.5 * v, 0.5 * v)


def _trig_product(x, y):
    """Product of two expressions whose terms hold at most one COS factor each."""
    if is_const(x) or is_const(y):
        return mul(x, y)
    out = ZERO
    for (t1, t2), (v1, v2) in zip(itertools.product(x[0], y[0]),
                                  itertools.product(x[1], y[1])):
        v = v1 * v2
        rest = ONE
        trig = []
        for f, n in zip(itertools.chain(t1[0], t2[0]), itertools.chain(t1[1], t2[1])):
            if f[0] == COS:
                trig.append(f)
            else:
                rest = mul(rest, ((((f, ), (n, )), ), (1, )))
        if len(trig) == 1:
            term = mul(rest, ((((trig[0], ), (1, )), ), (v, )))
        elif len(trig) == 2:
            term = mul(rest, _cos_times_cos(trig[0], trig[1], v))
        else:
            term = mul(rest, const_expr(v))
        out = add(out, term)
    return out


def _reduce_term(term, v):
    """One term -> sum of terms with at most one COS factor and at most one EXP."""
    trig = ONE
    alpha = shift_ = 0
    keep_f, keep_n = [], []
    for f, n in zip(*term):
        if f[0] == COS:
            trig = _trig_product(trig, _cos_pow(f, n))
        elif f[0] == EXP:
            x = alpha * shift_ + n * f[1] * f[-1]
            alpha += n * f[1]
            shift_ = 0 if alpha == 0 else x / alpha
        elif f[0] == GAUSSIAN and n != 1:
            keep_f.append((f[0], f[1] / np.sqrt(n), f[2]))
            keep_n.append(1)
        else:
            keep_f.append(f)
            keep_n.append(n)
    out = (((tuple(keep_f), tuple(keep_n)), ), (v, ))
    if alpha != 0:
        out = mul(out, primitive(EXP, alpha, shift=shift_))
    return mul(out, trig)


def _split_carrier(term):
    freq = shift_ = 0
    rest_f, rest_n = [], []
    for f, n in zip(*term):
        if f[0] == COS:
            if freq != 0:
                raise ValueError("run _exp_trig_Reduce first")
            freq, shift_ = f[1], f[-1]
        else:
            rest_f.append(f)
            rest_n.append(n)
    return freq, shift_, (tuple(rest_f), tuple(rest_n))


def simplify(expr, eps):
    merged = {}
    v = 0
    for t, v in zip(*expr):
        for t, v in zip(*_reduce_term(t, v)):
            freq, sh, t = _split_carrier(t)
            v_r, v_i, sh_r, sh_i = v.real, v.imag, sh, sh
            if (t, freq) in merged:
                p_r, ps_r, p_i, ps_i = merged[(t, freq)]
                if freq == 0:
                    v_r, v_i = v.real + p_r, v.imag + p_i
                else:
                    a = p_r * np.cos(freq * ps_r) + v_r * np.cos(freq * sh_r)
                    b = p_r * np.sin(freq * ps_r) + v_r * np.sin(freq * sh_r)
                    sh_r, v_r = np.arctan2(b, a) / freq, np.sqrt(a**2 + b**2)
                    a = p_i * np.cos(freq * ps_i) + v_i * np.cos(freq * sh_i)
                    b = p_i * np.sin(freq * ps_i) + v_i * np.sin(freq * sh_i)
                    sh_i, v_i = np.arctan2(b, a) / freq, np.sqrt(a**2 + b**2)
            merged[(t, freq)] = v_r, sh_r, v_i, sh_i
    out = ZERO
    for (t, freq), (v_r, sh_r, v_i, sh_i) in merged.items():
        # NB `v` is the amplitude of the LAST reduced term, not of this entry: the
        # reference's stale loop variable (_waveform.pyx:615),
        # kept so that simplified trees are identical.
        if freq == 0 and abs(v) >= eps:
            out = add(out, ((t, ), (v_r if v_i == 0 else v_r + 1j * v_i, )))
            continue
        if abs(v_i) < eps and abs(v_r) < eps:
            continue
        if abs(v_i) < eps:
            carrier = (((((COS, freq, sh_r), ), (1, )), ), (v_r, ))
        elif abs(v_r) < eps:
            carrier = (((((COS, freq, sh_i), ), (1, )), ), (v_i * 1j, ))
        else:
            carrier = (((((COS, freq, sh_r), ), (1, )), (((COS, freq, sh_i), ), (1, ))),
                       (v_r, v_i * 1j))
        out = add(out, mul(((t, ), (1, )), carrier))
    return out


def band_filter(expr, low, high, eps):
    """Keep the terms whose carrier frequency lies in [low, high)
    (reference: _waveform.pyx:638-654)."""
    expr = simplify(expr, eps)
    out = ZERO
    for t, v in zip(*expr):
        for f, n in zip(*t):
            if f[0] == COS:
                if low <= f[1] < high:
                    out = add(out, ((t, ), (v, )))
                break
        else:
            if low <= 0:
                out = add(out, ((t, ), (v, )))
    return out
