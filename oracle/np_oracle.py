"""NumPy restatement of the reference evaluator (TEST INFRASTRUCTURE ONLY).

Same pass structure as the reference's CPU path -- one NumPy ufunc pass per
operation, a per-piece factor cache, np.clip per piece, `out[a:b] += part` --
so that (i) it is a second, independent checker for the HIP path on the GPU box,
where the reference itself does not exist, and (ii) timing it gives the
"reference-like CPU path" that BASELINE.md §3 names as the denominator of the
speed-up target.  Follows:
    calc_parts / _calc / _apply      waveforms/_waveform.pyx:130-169
    built-in primitives              waveforms/_waveform.pyx:290-371
    Waveform.__call__ / _fill_parts  waveforms/waveform.py:524-563
    WaveVStack.__call__              waveforms/waveform.py:679-693
    Waveform.sample                  waveforms/waveform.py:173-207
    predistort (FIR branch)          waveforms/distortion.py:329-337
Pinned against tests/golden (made by running the real reference) in
tests/test_oracle_golden.py.  Never imported by waveforms_amd.
"""
import numpy as np
import scipy.special as special
from scipy.signal import fftconvolve

_ZERO = ((), ())


def _drag(t, t0, freq, width, delta, block_freq, phase):
    o = np.pi / width
    ox = np.sin(o * (t - t0))**2
    wt = 2 * np.pi * (freq + delta) * t - (2 * np.pi * delta * t0 + phase)
    if block_freq is None or block_freq - delta == 0:
        return ox * np.cos(wt)
    b = 1 / np.pi / 2 / (block_freq - delta)
    oy = -b * o * np.sin(2 * o * (t - t0))
    return ox * np.cos(wt) + oy * np.sin(wt)


def _mollifier(t, r, d):
    x = t / r
    q = np.abs(x)**2 - 1
    if d == 0:
        return np.where(q >= 0, 0, np.exp(1 / q + 1))
    p = np.poly1d([-2, 0])
    for n in range(1, d):
        p = (np.poly1d([1, 0, -2, 0, 1]) * p.deriv() +
             np.poly1d([-4 * n, 0, 4 * n - 2, 0]) * p)
    return np.where(q >= 0, 0, np.exp(1 / q + 1) / (-q)**(2 * d)) * p(x) / r**d


PRIMITIVES = {
    1: lambda t: t,
    2: lambda t, s: np.exp(-(t / s)**2),
    3: lambda t, s: special.erf(t / s),
    4: lambda t, w: np.cos(w * t),
    5: lambda t, bw: np.sinc(bw * t),
    6: lambda t, a: np.exp(a * t),
    7: lambda t, a, b, pts: np.interp(t, np.linspace(a, b, len(pts)), pts),
    8: lambda t, f0, f1, T, p: np.sin(p + 2 * np.pi * ((f1 - f0) / (2 * T) * t**2 + f0 * t)),
    9: lambda t, f0, al, p: np.sin(p + 2 * np.pi * f0 * (np.exp(al * t) - 1) / al),
    10: lambda t, f0, k, p: np.sin(p + 2 * np.pi * f0 / k * np.log(1 + k * t)),
    11: lambda t, w: np.cosh(w * t),
    12: lambda t, w: np.sinh(w * t),
    13: _drag,
    14: _mollifier,
    15: lambda t, s, n: ((-1)**n / s**n * special.hermite(n)(t / s) *
                         np.exp(-(t / s)**2)),
}


def eval_expr(expr, x, lib=None):
    # lib: function_lib of the reference (_apply, _waveform.pyx:130-131): id -> callable(t, *args)
    lib = PRIMITIVES if lib is None else lib
    cache = {}
    total = 0
    for (factors, powers), amp in zip(*expr):
        prod = 1
        for f, n in zip(factors, powers):
            if f not in cache:
                type_id, *args, shift = f
                cache[f] = lib[type_id](x - shift, *args)
            prod = prod * (cache[f] if n == 1 else cache[f]**n)
        total = total + amp * prod
    return total


def pieces(bounds, seq, x, lo=-np.inf, hi=np.inf, lib=None):
    edges = np.searchsorted(x, bounds)
    parts, dtype, a = [], float, 0
    for i, b in enumerate(edges):
        if a < b and seq[i] != _ZERO:
            part = np.clip(eval_expr(seq[i], x[a:b], lib), lo, hi)
            if isinstance(part, complex) or (isinstance(part, np.ndarray) and
                                             isinstance(part[0], complex)):
                dtype = complex
            parts.append((a, b, part))
        a = b
    return parts, dtype


def call_waveform(w, x, lib=None):
    parts, dtype = pieces(w.bounds, w.seq, x, w.min, w.max, lib)
    out = np.zeros_like(x, dtype=dtype)
    for a, b, part in parts:
        out[a:b] += part
    return out


def call_vstack(w, x, lib=None):
    out = np.full_like(x, w.offset, dtype=np.complex128)
    if w.shift != 0:
        x = x - w.shift
    for bounds, seq in w.wlist:
        for a, b, part in pieces(bounds, seq, x, lib=lib)[0]:
            out[a:b] += part
    return out.real


def call(w, x, lib=None):
    return call_vstack(w, x, lib) if hasattr(w, 'wlist') else call_waveform(w, x, lib)


def sample(w):
    return call(w, np.arange(w.start, w.stop, 1 / w.sample_rate))


def predistort_fir(sig, ker):
    n = len(sig)
    padded = np.hstack((np.zeros_like(sig), sig, np.zeros_like(sig)))
    a = n + len(ker) // 2
    return fftconvolve(padded, ker, mode='full')[a:a + n]


# ---- IIR stages --------------------------------------------
def sample_filtered(w, chunk_size=None):
    """Waveform.sample with filters=(sos, initial): scipy.signal.sosfilt on the sampled
    signal, state carried across chunks (reference: waveform.py:193-203, 209-257)."""
    from scipy.signal import sosfilt
    sos, initial = w.filters
    sos = np.array(sos)
    if chunk_size is None:
        sig = sample(w)
        return sosfilt(sos, sig - initial) + initial if initial else sosfilt(sos, sig)
    out, start = [], float(w.start)
    length = chunk_size / w.sample_rate
    zi = np.zeros((sos.shape[0], 2))
    while start < w.stop:
        if start + length > w.stop:
            length, stop = w.stop - start, float(w.stop)
            size = round((stop - start) * w.sample_rate)
        else:
            stop, size = start + length, chunk_size
        sig = call(w, np.linspace(start, stop, size, endpoint=False))
        if initial:
            sig = sig - initial
        sig, zi = sosfilt(sos, sig, zi=zi)
        out.append(sig + initial if initial else sig)
        start = stop
    return np.concatenate(out)


def predistort(sig, filters=None, ker=None, initial=0.0, zi=None):
    """predistort with both branches (reference: distortion.py:289-337) -> (out, zf)."""
    from scipy.signal import lfilter, lfiltic
    zf = None
    if filters is not None:
        b, a = np.poly1d([1.0]), np.poly1d([1.0])
        for b_, a_ in filters:
            b, a = b * np.poly1d(b_), a * np.poly1d(a_)
        b, a = b.coeffs, a.coeffs
        if zi is None:
            zi = lfiltic(b, a, np.full(len(a) - 1, initial), np.full(len(b) - 1, initial))
        sig, zf = lfilter(b, a, sig, zi=zi)
    if ker is not None:
        sig = predistort_fir(sig, ker)
    return sig, zf


# ---- multi-notch DRAG primitives, ids 16/17 (reference: waveforms/multy_drag.py) ----
def _md_setup(width, delta, block_freq):
    bs = []
    if isinstance(block_freq, float):
        block_freq = (block_freq, )
    if block_freq is not None:
        bs = list(1 / np.pi / 2 / (np.array(block_freq) - delta))
    m = max((len(bs) + 2) >> 1 << 1, 2)
    # B[n] = sum over n-subsets of prod(b) * J^n, J = [[0, 1], [-1, 0]]
    B = np.zeros((len(bs) + 1, 2, 2))
    B[0] = np.eye(2)
    for b in bs:
        B[1:] = B[1:] + B[:-1] @ np.array([[0, b], [-b, 0]])
    o = np.pi / width
    A = np.zeros((len(bs) + 1, m + 1))      # derivatives of sin^m in the (s^p, s^p c) basis
    A[0, m] = 1
    for i in range(1, len(bs) + 1):
        if i % 2:
            A[i][:-1] = A[i - 1][1:] * np.arange(1, m + 1) * o
        else:
            A[i][:-2] = A[i - 2][2:] * np.arange(1, m) * np.arange(2, m + 1)
            A[i] = (A[i] - A[i - 2] * np.arange(m + 1)**2) * o**2
    return bs, m, B, o, A


def _md_envelope(t, t0, width, plateau, m, o, A):
    rise, fall = t <= t0 + width / 2, t >= t0 + plateau + width / 2
    mid = (t > t0 + width / 2) * (t < t0 + plateau + width / 2)
    tau = np.where(fall, t - t0 - plateau, t - t0)
    s = np.where(mid, 0.0, np.sin(o * tau))
    c = np.where(mid, 0.0, np.cos(o * tau))
    basis = s**np.arange(m + 1).reshape(-1, 1)
    basis[1::2] = basis[1::2] * c
    d = A @ basis                            # d[n] = n-th derivative of the envelope
    d[0][mid] = 1
    return d, mid


def _md_crest(m, A, B):
    top = np.ones(m + 1)
    top[1::2] = 0
    top = A @ top
    return np.sqrt(np.sum(np.abs(np.einsum('ijk,ki->j', B, np.array([top, 0 * top])))**2))


def _md_tab_poly(f, x):
    import math
    M = len(f)
    rhs = np.copy(f)
    rhs[0] -= 1
    C = np.array([[x**(M + l - n) * math.factorial(M + l) / math.factorial(M + l - n)
                   for l in range(M)] for n in range(M)])
    from scipy.linalg import inv
    return np.poly1d([*np.flip(inv(C) @ rhs), *np.zeros(M - 1), 1])


def _drag_sin(t, t0, freq, width, delta, block_freq, phase, plateau=0):
    bs, m, B, o, A = _md_setup(width, delta, block_freq)
    d, _ = _md_envelope(t, t0, width, plateau, m, o, A)
    om = np.einsum('ijk,kim->jm', B, np.array([d, 0 * d])) / _md_crest(m, A, B)
    wt = 2 * np.pi * (freq + delta) * t - (2 * np.pi * delta * t0 + phase)
    return om[0] * np.cos(wt) + om[1] * np.sin(wt)


def _drag_sinx(t, t0, freq, width, delta, block_freq, phase, plateau=0, tab=0.618):
    bs, m, B, o, A = _md_setup(width, delta, block_freq)
    d, _ = _md_envelope(t, t0, width, plateau, m, o, A)

    def at(x):
        v = np.sin(o * x)**np.arange(m + 1)
        v[1::2] = v[1::2] * np.cos(o * x)
        return A @ v

    left = _md_tab_poly(at((1 - tab) * width / 2), -tab * width / 2)
    right = _md_tab_poly(at((1 + tab) * width / 2), tab * width / 2)
    lm = (t >= t0 + width / 2 - tab * width / 2) * (t <= t0 + width / 2)
    rm = (t >= t0 + plateau + width / 2) * (t <= t0 + plateau + width / 2 + tab * width / 2)
    for n in range(len(bs) + 1):
        d[n][lm] = np.polyder(left, m=n)(t[lm] - t0 - width / 2)
        d[n][rm] = np.polyder(right, m=n)(t[rm] - t0 - plateau - width / 2)
    om = np.einsum('ijk,kim->jm', B, np.array([d, 0 * d]))
    wt = 2 * np.pi * (freq + delta) * t - (2 * np.pi * delta * t0 + phase)
    return om[0] * np.cos(wt) + om[1] * np.sin(wt)


PRIMITIVES[16] = _drag_sin
PRIMITIVES[17] = _drag_sinx
