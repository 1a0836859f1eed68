"""Multi-notch DRAG pulses `drag_sin` / `drag_sinx`.

Constructors and primitive ids (16, 17) follow the reference
(waveforms/multy_drag.py:180-232).  The reference evaluates these primitives per sample
with `np.piecewise` + `einsum` over small matrices; all of that structure depends only on
the pulse parameters, so here it is *compiled on the host* into coefficient tables
(`device_args`) and the device evaluates, per sample, one sincos for the envelope, two
short Horner sums, and one sincos for the carrier:

    Omega_j(t) = sum_n B_j[n] * d^n/dt^n sin^m(o (t - t0))          (rising / falling edge)
               = sum_p P_j[p] * s^p * (c if p odd),  s = sin(o tau), c = cos(o tau)
    value      = Omega_x cos(wt) + Omega_y sin(wt)

with B_j[n] = e_n(b) * (J^n)[j][0] (e_n: elementary symmetric polynomial of the notch
coefficients b_i = 1 / (2 pi (block_freq_i - delta)), J = [[0, 1], [-1, 0]]) -- the
closed form of the reference's `B_series_mat` recursion.
"""
from __future__ import annotations

import math

import numpy as np
from numpy import inf, pi

from ._ir import DRAG_SIN, DRAG_SINX, NDIGITS, ZERO, primitive
from .waveform import Waveform


def _as_tuple(block_freq):
    if block_freq is None:
        return None
    if isinstance(block_freq, (int, float)):
        return (float(block_freq), )
    return tuple(block_freq)


def drag_sin(freq, width, plateau=0, delta=0, block_freq=None, phase=0, t0=0):
    """reference: waveforms/multy_drag.py:180-190"""
    phase += pi * delta * (width + plateau)
    if isinstance(block_freq, float):
        block_freq = (block_freq, )
    return Waveform(seq=(ZERO, primitive(DRAG_SIN, t0, freq, width, delta, block_freq,
                                         phase, plateau), ZERO),
                    bounds=(round(t0, NDIGITS), round(t0 + width + plateau, NDIGITS), +inf))


def drag_sinx(freq, width, plateau=0, delta=0, block_freq=None, phase=0, t0=0, tab=0.618):
    """reference: waveforms/multy_drag.py:215-232"""
    phase += pi * delta * (width + plateau)
    if isinstance(block_freq, float):
        block_freq = (block_freq, )
    return Waveform(seq=(ZERO, primitive(DRAG_SINX, t0, freq, width, delta, block_freq,
                                         phase, plateau, tab), ZERO),
                    bounds=(round(t0, NDIGITS), round(t0 + width + plateau, NDIGITS), +inf))


# --------------------------------------------------------------------------
# host-side compilation of the envelope structure
# --------------------------------------------------------------------------
def _notch_weights(block_freq, delta):
    """(m, Bx[n], By[n]) for n = 0..N: B_j[n] = e_n(b) * (J^n)[j][0]."""
    bs = []
    if block_freq is not None:
        bs = list(1 / np.pi / 2 / (np.array(block_freq, dtype=float) - delta))
    m = max((len(bs) + 2) >> 1 << 1, 2)
    e = np.poly1d([1.0])
    for b in bs:                       # prod (x + b): coefficient of x^(N-n) is e_n
        e = e * np.poly1d([1.0, b])
    en = e.coeffs                      # e_0 .. e_N
    jx = [1.0, 0.0, -1.0, 0.0]
    jy = [0.0, -1.0, 0.0, 1.0]
    bx = np.array([en[n] * jx[n % 4] for n in range(len(bs) + 1)])
    by = np.array([en[n] * jy[n % 4] for n in range(len(bs) + 1)])
    return m, bx, by


def _sin_power_derivatives(m, order, o):
    """D[n][p]: d^n/dt^n sin^m(o t) = sum_p D[n][p] s^p (times c for odd p).
    Even orders stay polynomials in s:  (s^p)'' = o^2 (p (p-1) s^(p-2) - p^2 s^p);
    an odd order is the derivative of the even one below it: (s^p)' = o p s^(p-1) c."""
    D = np.zeros((order + 1, m + 1))
    D[0, m] = 1.0
    p = np.arange(m + 1)
    for n in range(1, order + 1):
        if n % 2:
            D[n, :-1] = o * p[1:] * D[n - 1, 1:]
        else:
            D[n, :-2] = p[2:] * (p[2:] - 1) * D[n - 2, 2:]
            D[n] -= p**2 * D[n - 2]
            D[n] *= o * o
    return D


def _basis_at(m, o, tau):
    """[s^p * (c if p odd)] at one point."""
    s, c = np.sin(o * tau), np.cos(o * tau)
    v = s**np.arange(m + 1)
    v[1::2] *= c
    return v


def _tab_polynomial(target, x):
    """Polynomial 1 + sum_l q_l tau^(M+l), l < M, whose derivatives of order 0..M-1 at
    tau = x equal `target` (M = len(target)); coefficients highest degree first
    (reference: waveforms/multy_drag.py:77-90)."""
    M = len(target)
    rhs = np.array(target, dtype=float)
    rhs[0] -= 1.0
    C = np.zeros((M, M))
    for n in range(M):
        for l in range(M):
            C[n, l] = x**(M + l - n) * math.factorial(M + l) / math.factorial(M + l - n)
    q = np.linalg.solve(C, rhs)
    return np.poly1d([*q[::-1], *np.zeros(M - 1), 1.0])


def device_args(type_id, args):
    """Reference factor args -> the compiled argument list of include/wfk.h ids 16/17:
      [t0, freq, width, delta, phase, plateau, tab_half_width, m, dq,
       Px[0..m], Py[0..m], Cx, Cy,  then for id 17: QLx, QLy, QRx, QRy (dq+1 each)]"""
    if type_id == DRAG_SIN:
        t0, freq, width, delta, block_freq, phase, plateau = args
        tab = None
    else:
        t0, freq, width, delta, block_freq, phase, plateau, tab = args
    block_freq = _as_tuple(block_freq)
    m, bx, by = _notch_weights(block_freq, delta)
    N = len(bx) - 1
    o = np.pi / width
    D = _sin_power_derivatives(m, N, o)
    flat = np.ones(m + 1)
    flat[1::2] = 0
    top = D @ flat                         # derivatives at the crest (s = 1, c = 0)
    norm = 1.0
    if type_id == DRAG_SIN:
        norm = math.sqrt((bx @ top)**2 + (by @ top)**2)
    px, py = (bx @ D) / norm, (by @ D) / norm
    # plateau: the reference leaves D[n][0] in the higher orders and forces order 0 to 1
    mid = D[:, 0].copy()
    mid[0] = 1.0
    cx, cy = (bx @ mid) / norm, (by @ mid) / norm
    out = [t0, freq, width, delta, phase, plateau, 0.0, float(m), -1.0, *px, *py, cx, cy]
    if type_id == DRAG_SINX:
        half = tab * width / 2
        out[6] = half
        left = _tab_polynomial(D @ _basis_at(m, o, (1 - tab) * width / 2), -half)
        right = _tab_polynomial(D @ _basis_at(m, o, (1 + tab) * width / 2), half)
        polys = []
        for P in (left, right):
            ders = [np.polyder(P, n) if n else P for n in range(N + 1)]
            for w in (bx, by):
                acc = np.poly1d([0.0])
                for n in range(N + 1):
                    acc = acc + w[n] * ders[n]
                polys.append(acc)
        dq = max(len(p.coeffs) for p in polys) - 1
        out[8] = float(dq)
        for p in polys:
            c = np.zeros(dq + 1)
            c[dq + 1 - len(p.coeffs):] = p.coeffs
            out.extend(c)
    return [float(v) for v in out]
