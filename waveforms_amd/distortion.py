"""Pre-distortion stages on the GPU: `predistort(sig, filters=..., ker=...)`.

Signature and semantics follow the reference's `predistort`
(waveforms/distortion.py:289-337):
  * FIR branch (`ker=`): out[i] = sum_k ker[k] * sig[i + len(ker)//2 - k], zero padded --
    what the reference computes with one giant `scipy.signal.fftconvolve`;
  * IIR branch (`filters=`): the combined transfer function run as
    `scipy.signal.lfilter(b, a, sig, zi=lfiltic(...))` -- here a
    block-parallel linear-recurrence scan (csrc/wfk_iir.hip).
Filter *design* (polynomial products, zpk conversions: O(order) host work, never per
sample) uses NumPy/SciPy exactly as the reference does.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import _engine


class FirStage:
    """Device-resident FIR stage for `batch` rows of `n` samples (build once, apply
    many times).  `apply_torch(x, y)`: x, y (batch, >= n) device tensors.  `ker` of shape (batch, K):
    one kernel per row."""

    def __init__(self, ker, n: int, batch: int = 1, dtype=np.float64):
        self.plan = _engine.FirPlan(ker, n, batch, dtype)
        self.n, self.batch, self.dtype = int(n), int(batch), np.dtype(dtype)

    def apply(self, in_ptr, in_stride, out_ptr, out_stride, stream=0):
        self.plan.apply(in_ptr, in_stride, out_ptr, out_stride, stream)

    def apply_torch(self, x, y):
        import torch
        want = torch.float64 if self.dtype == np.float64 else torch.float32
        for t in (x, y):
            if (not t.is_cuda or t.dtype != want or t.dim() != 2 or
                    t.shape[0] != self.batch or t.shape[1] < self.n or t.stride(1) != 1):
                raise ValueError('expected (batch, >=n) row-contiguous device tensors '
                                 'of the plan dtype')
        if x.data_ptr() == y.data_ptr():
            raise ValueError('FIR is out of place')
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self.apply(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), stream)
        return y

    def close(self):
        self.plan.close()


class SampledFir:
    """`predistort(wav(t), ker=ker)` for many channels on one uniform grid, device-resident:
    the sampler runs INSIDE the FIR transform when the channels are fully fused (`fused`; on fine grids
    as stride-256 chains, at AWG sample rates the short tier's way: `plan.kernel_name()`), so
    the unfiltered samples never touch HBM (reference chain: waveform.py:529-563 ->
    distortion.py:329-337).  Build once, launch many times.

        sf = SampledFir(channels, ('linspace', 0.0, 3e-6, 10**7, False), ker)   # ker (K,) or (n_channels, K)
        sf.launch_torch(out)          # (n_channels, >= n) device tensor of the plan dtype
    """

    def __init__(self, channels, grid, ker, dtype=np.float64, function_lib=None, tile=1):
        """`tile` > 1 repeats the channel list that many times (synthetic batches, as BatchSampler's)"""
        from . import _flatten
        if not isinstance(grid, _flatten.wfk_grid):
            grid = _flatten.grid_from_desc(grid)
        self.prog = _flatten.tile_program(_flatten.flatten(list(channels), grid, function_lib), tile)
        self.plan = _engine.ChainPlan(self.prog, grid, ker, dtype)
        self.n, self.n_channels, self.dtype = self.plan.n, self.plan.n_channels, np.dtype(dtype)
        self.fused, self.why_not = self.plan.fused, self.plan.why_not

    def launch(self, out_ptr, out_stride=None, stream=0):
        self.plan.launch(out_ptr, self.n if out_stride is None else out_stride, stream)

    def launch_torch(self, out):
        import torch
        want = torch.float64 if self.dtype == np.float64 else torch.float32
        if (not out.is_cuda or out.dtype != want or out.dim() != 2 or out.shape[0] != self.n_channels
                or out.shape[1] < self.n or out.stride(1) != 1):
            raise ValueError('out must be a (n_channels, >=n) row-contiguous device tensor of the plan dtype')
        self.launch(out.data_ptr(), out.stride(0), torch.cuda.current_stream(out.device).cuda_stream)
        return out

    def to_host(self):
        buf = _engine.DeviceBuffer(max(self.n_channels * self.n, 1) * self.dtype.itemsize)
        try:
            self.launch(buf.ptr)
            _engine.sync()
            return buf.download((self.n_channels, self.n), self.dtype)
        finally:
            buf.close()

    def close(self):
        self.plan.close()


class SampledIir:
    """`sosfilt(sos, wav(t) - initial, zi) + initial` (Waveform.sample(filters=), reference waveform.py:190-203,
    244-251) -- or any cascade of (b, a) sections, optionally followed by the FIR of `predistort(., filters, ker)`
    (distortion.py:298-337) -- for many channels on one uniform grid, device-resident.  When the channels are fully
    fused and the cascade's first pass carries <= 4 state values, the wave that owns a chunk of the IIR scan evaluates
    its input itself (`fused`, `plan.kernel_name()` == 'iir_sampled<...>'): the unfiltered samples never touch HBM.

        si = SampledIir(channels, ('linspace', 0.0, 3e-6, 10**7, False), sos)        # sos (n_sections, 6) or [(b, a), ...]
        si.launch_torch(out, initial=0.0)     # (n_channels, >= n) device tensor of the plan dtype; -> final state or None
    """

    def __init__(self, channels, grid, sections, ker=None, dtype=np.float64, function_lib=None, tile=1):
        """`tile` > 1 repeats the channel list that many times (synthetic batches, as BatchSampler's)"""
        from . import _flatten
        if not isinstance(grid, _flatten.wfk_grid):
            grid = _flatten.grid_from_desc(grid)
        try:
            sec = np.asarray(sections, dtype=np.float64)
        except (ValueError, TypeError):
            sec = None
        if sec is not None and sec.ndim == 2 and sec.shape[1] == 6:
            sections = [(r[:3], r[3:]) for r in sec]              # an SOS matrix (scipy.signal.sosfilt)
        self.prog = _flatten.tile_program(_flatten.flatten(list(channels), grid, function_lib), tile)
        self.plan = _engine.ChainIirPlan(self.prog, grid, sections, ker, dtype)
        self.n, self.n_channels, self.dtype = self.plan.n, self.plan.n_channels, np.dtype(dtype)
        self.state_dim = self.plan.state_dim
        self.why_not = self.plan.why_not

    @property
    def fused(self):
        return self.plan.fused

    def launch(self, out_ptr, out_stride=None, zi_ptr=None, zf_ptr=None, initial=0.0, stream=0):
        """-> False if the library refused the launch after an earlier look-back timeout (launch again)"""
        return self.plan.launch(out_ptr, self.n if out_stride is None else out_stride, zi_ptr, zf_ptr, initial, stream)

    def launch_torch(self, out, initial=0.0, zi=None, zf=None):
        """zi / zf: optional (n_channels, state_dim) float64 device tensors (initial state in / final state out)"""
        import torch
        want = torch.float64 if self.dtype == np.float64 else torch.float32
        if (not out.is_cuda or out.dtype != want or out.dim() != 2 or out.shape[0] != self.n_channels
                or out.shape[1] < self.n or out.stride(1) != 1):
            raise ValueError('out must be a (n_channels, >=n) row-contiguous device tensor of the plan dtype')
        for z in (zi, zf):
            if z is not None and (not z.is_cuda or z.dtype != torch.float64 or not z.is_contiguous()
                                  or tuple(z.shape) != (self.n_channels, self.state_dim)):
                raise ValueError('zi / zf must be contiguous (n_channels, state_dim) float64 device tensors')
        s = torch.cuda.current_stream(out.device).cuda_stream
        for attempt in range(2):
            ok = self.launch(out.data_ptr(), out.stride(0), None if zi is None else zi.data_ptr(),
                             None if zf is None else zf.data_ptr(), initial, s)
            if ok:
                return out
        raise _engine.EngineError('IIR chain refused twice')

    def to_host(self, initial=0.0, zi=None, return_zf=False):
        """NumPy result (n_channels, n); a look-back timeout (outputs NaN) is retried once in the unfused form"""
        D = max(self.state_dim, 1)
        buf = _engine.DeviceBuffer(max(self.n_channels * self.n, 1) * self.dtype.itemsize)
        dzi = _engine.DeviceBuffer(self.n_channels * D * 8) if zi is not None else None
        dzf = _engine.DeviceBuffer(self.n_channels * D * 8) if return_zf else None
        try:
            if zi is not None:
                z = np.broadcast_to(np.asarray(zi, dtype=np.float64).reshape(-1, self.state_dim)
                                    if np.ndim(zi) > 1 else np.asarray(zi, dtype=np.float64),
                                    (self.n_channels, self.state_dim))
                dzi.upload(np.ascontiguousarray(z))
            for attempt in range(2):
                ok = self.launch(buf.ptr, None, None if dzi is None else dzi.ptr, None if dzf is None else dzf.ptr, initial)
                if self.plan.status() and ok:
                    break
                if attempt == 1:
                    raise _engine.EngineError('IIR chain failed twice')
            out = buf.download((self.n_channels, self.n), self.dtype)
            if return_zf:
                return out, dzf.download((self.n_channels, D), np.float64)[:, :self.state_dim]
            return out
        finally:
            buf.close()
            if dzi is not None:
                dzi.close()
            if dzf is not None:
                dzf.close()

    def close(self):
        self.plan.close()


def fir_host(sig: np.ndarray, ker: np.ndarray) -> np.ndarray:
    """NumPy in, NumPy out: upload, overlap-save FIR on the device, download.  Complex signals and / or
    kernels (the reference's `fftconvolve` takes both, distortion.py:329-337) run as real convolutions of
    their parts: (sr + i si) * (kr + i ki) = sr*kr - si*ki + i (sr*ki + si*kr)."""
    sig, ker = np.asarray(sig), np.asarray(ker)
    if np.iscomplexobj(sig) or np.iscomplexobj(ker):
        shape = np.shape(sig)
        rows = np.atleast_2d(sig)
        parts = [np.ascontiguousarray(rows.real, dtype=np.float64)]
        if np.iscomplexobj(sig):
            parts.append(np.ascontiguousarray(rows.imag, dtype=np.float64))
        stacked = np.concatenate(parts, axis=0)              # the parts as extra rows: one launch per kernel part
        nb = rows.shape[0]
        a = _fir_host_real(stacked, np.ascontiguousarray(ker.real, dtype=np.float64))
        out = a[:nb].astype(np.complex128)
        if len(parts) == 2:
            out += 1j * a[nb:]
        if np.iscomplexobj(ker):
            b = _fir_host_real(stacked, np.ascontiguousarray(ker.imag, dtype=np.float64))
            out += 1j * b[:nb]
            if len(parts) == 2:
                out -= b[nb:]
        return out.reshape(shape)
    return _fir_host_real(sig, ker)


def _fir_host_real(sig: np.ndarray, ker: np.ndarray) -> np.ndarray:
    sig2 = np.ascontiguousarray(np.atleast_2d(sig), dtype=np.float64)
    batch, n = sig2.shape
    if n == 0:
        return np.zeros_like(np.asarray(sig, dtype=np.float64))
    stage = FirStage(ker, n, batch, np.float64)
    din = _engine.DeviceBuffer(sig2.nbytes)
    dout = _engine.DeviceBuffer(sig2.nbytes)
    try:
        din.upload(sig2)
        stage.apply(din.ptr, n, dout.ptr, n)
        _engine.sync()
        out = dout.download(sig2.shape, np.float64)
    finally:
        din.close()
        dout.close()
        stage.close()
    return out.reshape(np.shape(sig))


def combine_filters(filters):
    """Product of the (b, a) transfer functions (reference: distortion.py:226-244)."""
    b, a = np.poly1d([1.0]), np.poly1d([1.0])
    for b_, a_ in filters:
        b = b * np.poly1d(b_)
        a = a * np.poly1d(a_)
    return b.coeffs, a.coeffs


def exp_decay_filter(amp, tau, sample_rate, inv=False, output='ba'):
    """Multi-exponential step-response filter: u(t) -> u(t)(1 - sum A_i exp(-t/tau_i))
    (reference: distortion.py:100-185).  Returns (b, a), sos or (z, p, k)."""
    from scipy.signal import zpk2sos, zpk2tf
    if isinstance(amp, (int, float, complex)):
        amp, tau = [amp], [tau]
    num, den = np.poly1d([0.0]), np.poly1d([1.0])
    for i, (A, t) in enumerate(zip(amp, tau)):
        den = den * np.poly1d([1, -1 / t])
        term = np.poly1d([-A, 0.0])
        for j, t_ in enumerate(tau):
            if j != i:
                term = term * np.poly1d([1, -1 / t_])
        num = num + term
    num = num + den
    z = np.exp(-num.roots / sample_rate)
    p = np.exp(-1 / (np.asarray(tau) * sample_rate))
    if inv:
        z, p = p, z
    p = p[np.abs(p) < 1]
    k = (np.prod(1 - p) / np.prod(1 - z)).real
    if output == 'sos':
        return zpk2sos(z, p, k)
    if output == 'ba':
        return zpk2tf(z, p, k)
    if output == 'zpk':
        return z, p, k
    raise ValueError(f"Invalid output type: {output}")


def iir_host(sig, sections, zi=None, initial=0.0, ker=None):
    """NumPy in/out: upload, IIR scan (+ optional FIR) on the device, download.
    Returns (y, zf) with zf the final filter state (scipy layout).  A complex signal, initial value or
    state (scipy's lfilter / sosfilt take them; reference distortion.py:298-321) runs as two real passes:
    the coefficients are real, so real and imaginary parts filter independently."""
    cplx = np.iscomplexobj(sig) or np.iscomplexobj(initial) or (zi is not None and np.iscomplexobj(zi))
    if cplx or (ker is not None and np.iscomplexobj(ker)):
        if any(np.iscomplexobj(np.asarray(c)) for sec in sections for c in sec):
            raise NotImplementedError('IIR sections with complex coefficients')
        sig = np.asarray(sig)
        kr = None if (ker is None or np.iscomplexobj(ker)) else ker
        zr = None if zi is None else np.real(zi)
        yr, zfr = iir_host(np.real(sig), sections, zr, float(np.real(initial)), kr)
        if cplx:
            zim = None if zi is None else np.imag(zi)
            yi, zfi = iir_host(np.imag(sig), sections, zim, float(np.imag(initial)), kr)
            y, zf = yr + 1j * yi, zfr + 1j * zfi
        else:
            y, zf = yr, zfr
        if ker is not None and kr is None:
            y = fir_host(y, ker)
        return y, zf
    sig2 = np.ascontiguousarray(np.atleast_2d(sig), dtype=np.float64)
    batch, n = sig2.shape
    plan = _engine.IirPlan(sections, n, batch, np.float64)
    D = plan.state_dim
    bufs = []

    def dev(nbytes):
        b = _engine.DeviceBuffer(max(nbytes, 8))
        bufs.append(b)
        return b

    fir = None
    try:
        x, y = dev(sig2.nbytes), dev(sig2.nbytes)
        x.upload(sig2)
        dzi = None
        if zi is not None:
            z = np.ascontiguousarray(np.broadcast_to(np.asarray(zi, dtype=np.float64).reshape(-1),
                                                     (batch, D)))
            dzi = dev(z.nbytes)
            dzi.upload(z)
        dzf = dev(batch * D * 8)
        ok = plan.apply(x.ptr, n, y.ptr, n, dzi.ptr if dzi else None, dzf.ptr, initial)
        if not (plan.status() and ok):   # a single-pass look-back timed out (stalled predecessor chunk): every part
            ok = plan.apply(x.ptr, n, y.ptr, n, dzi.ptr if dzi else None, dzf.ptr, initial)  # of the plan has switched
            if not (plan.status() and ok):                                                   # form; x is intact
                raise _engine.EngineError('IIR stage failed twice')
        res = y
        if ker is not None and n > 0:
            fir = FirStage(ker, n, batch, np.float64)
            fir.apply(y.ptr, n, x.ptr, n)
            res = x
        _engine.sync()
        out = res.download(sig2.shape, np.float64) if n else sig2.copy()
        zf = dzf.download((batch, D), np.float64) if n else np.zeros((batch, D))
    finally:
        for b in bufs:
            b.close()
        plan.close()
        if fir is not None:
            fir.close()
    return out.reshape(np.shape(sig)), (zf[0] if np.ndim(sig) == 1 else zf)


class CascadeState(np.ndarray):
    """The `zf` that `predistort(..., return_zf=True)` hands out for a combined filter order > 16: the
    direct-form-II-transposed states of the caller's sections back to back (NOT scipy.signal.lfilter's state
    of the combined polynomials, which has the same length).  Only such an array is accepted back as `zi`
    on that path, so an lfilter-format state cannot be misread silently."""


def predistort(sig, filters=None, ker=None, initial=0.0, initial_x=None,
               initial_y=None, zi=None, return_zf=False):
    """reference: waveforms/distortion.py:289-337, both branches on the GPU.

    Combined filter order <= 16 (every exp-decay correction with up to 16 time constants): the reference's
    semantics to the letter -- `zi` / the returned `zf` are scipy.signal.lfilter's state of the combined
    (b, a), `initial` / `initial_x` / `initial_y` seed it through lfiltic.

    Combined order > 16: a direct form of that order is not usable in double precision (the reference's own
    lfilter returns 1e125 / NaN once poles crowd z = 1), so the caller's sections run as a cascade:
      * `initial=c` starts every section in its steady state for the level that reaches it (c times the DC
        gains before it).  For sections of unit DC gain -- every exp_decay_filter -- this IS the reference's
        `initial_x = initial_y = c`; tests/golden/iir.npz holds reference runs of order 17 and 20 where its
        direct form is still accurate, and the cascade agrees with them to 1e-9.
      * `return_zf` hands out a `CascadeState`; `zi` must be one (ValueError otherwise).
      * `initial_x` / `initial_y` histories cannot be expressed: ValueError."""
    sig = np.asarray(sig)
    zf = None
    if filters is not None:
        from scipy.signal import lfiltic, tf2zpk
        b, a = combine_filters(filters)
        _, p, _ = tf2zpk(b, a)
        if not np.all(np.abs(p) < 1):
            warnings.warn('Warning: filter is unstable')
        if max(len(b), len(a)) - 1 > 16:
            if initial_x is not None or initial_y is not None:
                raise ValueError('predistort: a combined filter of order > 16 has no usable direct form in double '
                                 'precision; initial_x / initial_y histories cannot be honoured -- use initial=')
            if zi is not None and not isinstance(zi, CascadeState):
                raise ValueError('predistort: for a combined filter order > 16 `zi` must be the CascadeState that '
                                 'return_zf=True handed out (the sections\' states back to back); an lfilter-format '
                                 'state of the combined polynomials has the same length but another meaning')
            return _predistort_high_order(sig, filters, zi, ker, return_zf, complex(initial) if zi is None else None)
        if zi is None:
            ix = (np.full((len(b) - 1, ), initial) if initial_x is None else
                  np.asarray(initial_x)[:len(b) - 1])
            iy = (np.full((len(a) - 1, ), initial) if initial_y is None else
                  np.asarray(initial_y)[:len(a) - 1])
            zi = lfiltic(b, a, iy, ix)
        sections = [(b, a)]
        sig, zf = iir_host(sig, sections, zi=zi, ker=ker)
        return (sig, zf) if return_zf else sig
    if ker is None:
        return (sig, zf) if return_zf else sig
    out = fir_host(sig, np.asarray(ker))
    return (out, zf) if return_zf else out


def _predistort_high_order(sig, filters, zi, ker, return_zf, steady):
    """predistort(filters=...) for a combined order > 16 (reference distortion.py:298-321: one lfilter call on
    the product polynomials), which has no single device section -- and no usable direct form at all: with
    twenty poles next to z = 1 the free response of a direct-form state is the difference of terms 1e50
    times its size, the reference's own double-precision lfilter returns 1e125 or NaN on such sections
    (measured; tests/test_gpu_iir.py), and a state vector of doubles cannot even carry the information.
    The same LTI system is the cascade of the caller's own sections, run on the device:
      * `initial=c` ("the line sat at c for ever", steady != None): every section starts in ITS steady state
        for the constant level that reaches it, c times the DC gains before it -- exact, well conditioned;
      * zi / return_zf: the state is the CASCADE's (the sections' direct-form-II-transposed states back to
        back, sum of the section orders = the combined order values): what `return_zf` hands out, `zi`
        takes back, and a signal processed in pieces equals the signal processed whole."""
    secs = [(np.atleast_1d(np.asarray(b_, dtype=np.float64)), np.atleast_1d(np.asarray(a_, dtype=np.float64)))
            for b_, a_ in filters]
    orders = [max(len(b_), len(a_)) - 1 for b_, a_ in secs]
    zsec = None
    if steady is not None:
        if steady != 0:
            # (a complex level -- scipy takes a complex `initial` -- seeds real and imaginary parts alike:
            #  the sections are real and lfiltic is linear in its histories)
            from scipy.signal import lfiltic
            level, parts = (steady if steady.imag != 0 else steady.real), []
            for (b_, a_), m in zip(secs, orders):
                gain = b_.sum() / a_.sum()
                z = lfiltic(b_, a_, np.full(len(a_) - 1, level * gain), np.full(len(b_) - 1, level))
                parts.append(np.concatenate([z, np.zeros(m - len(z))]))
                level *= gain
            zsec = np.concatenate(parts)
    elif zi is not None:
        zsec = np.asarray(zi).reshape(-1)
        if len(zsec) != sum(orders):
            raise ValueError(f'predistort: zi must hold the {sum(orders)} cascade state values')
    out, zf = iir_host(sig, secs, zi=zsec, ker=ker)
    return (out, np.asarray(zf).view(CascadeState)) if return_zf else out


def distort(points, params, sample_rate, initial=0.0):
    """reference: waveforms/distortion.py:340-346."""
    filters = []
    for amp, tau in np.asarray(params).reshape(-1, 2):
        filters.append(exp_decay_filter(amp, abs(tau), sample_rate))
    return predistort(points, filters, initial=initial)


# --------------------------------------------------------------------------
# FFT-domain operations
# --------------------------------------------------------------------------
def reflection_filter(f, A, tau):
    """Transfer function of a single reflection (reference: distortion.py:188-205)."""
    return (1 - A) / (1 - A * np.exp(-2j * np.pi * f * tau))


def transfer_host(sig, H):
    """irfft(rfft(sig) * H) on the device; H: the n//2+1 non-negative-frequency bins."""
    sig = np.ascontiguousarray(sig, dtype=np.float64)
    n = len(sig)
    Hc = np.ascontiguousarray(H, dtype=np.complex128)
    assert Hc.shape == (n // 2 + 1, )
    plan = _engine.SpectralPlan(n, 1, np.float64)
    x, y, h = (_engine.DeviceBuffer(n * 8), _engine.DeviceBuffer(n * 8),
               _engine.DeviceBuffer(Hc.nbytes))
    try:
        x.upload(sig)
        h.upload(Hc)
        plan.apply(x.ptr, y.ptr, h.ptr)
        _engine.sync()
        return y.download((n, ), np.float64)
    finally:
        for b in (x, y, h):
            b.close()
        plan.close()


def reflection(sig, A, tau, sample_rate):
    """ifft(fft(sig) * H(f)).real with H the reflection filter
    (reference: distortion.py:208-210); whole-signal FFT on the device."""
    freq = np.fft.rfftfreq(len(sig), 1 / sample_rate)
    return transfer_host(sig, reflection_filter(freq, A, tau))


def correct_reflection(sig, A, tau, sample_rate=None):
    """Inverse of `reflection` (reference: distortion.py:213-223); symbolic for a Waveform."""
    from .waveform import Waveform
    if isinstance(sig, Waveform):
        return 1 / (1 - A) * sig - A / (1 - A) * (sig >> tau)
    if sample_rate is None:
        raise ValueError('sample_rate is not given')
    freq = np.fft.rfftfreq(len(sig), 1 / sample_rate)
    return transfer_host(sig, 1 / reflection_filter(freq, A, tau))


def shift(signal, delay, dt):
    """Delay a sampled signal by `delay` (3-tap fractional interpolation on the device +
    integer shift; reference: distortion.py:12-39)."""
    signal = np.asarray(signal)          # (dtype kept as upstream does: complex signals stay complex)
    points = int(delay // dt)
    delta = delay / dt - points
    if delta > 0:
        signal = fir_host(signal, np.array([0, 1 - delta, delta]))
    if points == 0:
        return signal
    ret = np.zeros_like(signal)
    if points < 0:
        ret[:points] = signal[-points:]
    else:
        ret[points:] = signal[:-points]
    return ret


def zDistortKernel(dt, params):
    """FIR kernel of a Z-line distortion model (filter DESIGN: one small FFT on the host,
    reference: distortion.py:52-60)."""
    t = 3 * np.asarray(params)[:, 0].max()
    omega = 2 * np.pi * np.fft.fftfreq(int(t / dt) + 1, dt)
    H = 1
    for tau, A in params:
        H = H + (1j * A * omega * tau) / (1j * omega * tau + 1)
    return np.fft.ifftshift(np.fft.ifft(1 / H)).real


def high_pass_filter(tau, sample_rate):
    """reference: distortion.py:63-70"""
    k = 2.0 * tau * sample_rate
    return [k / (1 + k), -k / (1 + k)], [1.0, (1 - k) / (1 + k)]


# --------------------------------------------------------------------------
# filter-design helpers (host, O(order) or one short FFT: design time, not the data path)
# --------------------------------------------------------------------------
def extractKernel(sig_in, sig_out, sample_rate, bw=None, skip=0):
    """Deconvolution kernel that maps `sig_out` back onto `sig_in` (reference:
    distortion.py:42-48): centred inverse FFT of the spectral ratio, optionally smoothed
    with a +-3 sigma Gaussian of `2*sample_rate/bw` points, `skip` samples cut at each end."""
    ratio = np.fft.fft(sig_in) / np.fft.fft(sig_out)
    ker = np.fft.ifftshift(np.fft.ifft(ratio)).real
    if bw is not None and bw < 0.5 * sample_rate:
        g = np.exp(-0.5 * np.linspace(-3.0, 3.0, int(2 * sample_rate / bw))**2)
        ker = np.convolve(ker, g / g.sum(), mode='same')
    return ker[int(skip):len(ker) - int(skip)]


def exp_decay_filter_old(amp, tau, sample_rate):
    """First-order section of H(w) = A / (1 - 1j/(w tau)) in the reference's earlier
    discretisation (distortion.py:73-99) -> (b, a)."""
    decay = np.exp(-1 / (abs(sample_rate * tau) * (1 + amp)))      # = 1 - alpha
    alpha = 1 - decay
    if amp >= 0:
        k = amp / (1 + amp - alpha)
        a0, a1 = 1 - k + k * alpha, -(1 - k) * (1 - alpha)
    else:
        k = -amp / (1 + amp) / (1 - alpha)
        a0, a1 = 1 + k - k * alpha, -(1 + k) * (1 - alpha)
    return [1 / a0, -(1 - alpha) / a0], [1, a1 / a0]


def factor_filter(b, a):
    """Split (b, a) into first-order sections, one per pole/zero pair, the gain spread
    evenly over them (reference: distortion.py:247-265).  Note np.poly1d indexing: `b[0]`
    is the CONSTANT coefficient, as in the reference."""
    from itertools import zip_longest
    bp, ap = np.poly1d(b), np.poly1d(a)
    poles, zeros = ap.roots, bp.roots
    gain = (bp[0] / ap[0])**(1 / max(len(zeros), len(poles)))
    return [([gain, -gain * z], [1, -p]) for p, z in zip_longest(poles, zeros, fillvalue=0)]


def stable_filter(exp_decay_filters, sample_rate):
    """True when every pole of the combined exp-decay cascade lies inside the unit circle
    (reference: distortion.py:268-286; it unpacks exp_decay_filter's (b, a) as (a, b) and
    swaps them back when combining, which is kept)."""
    from scipy.signal import tf2zpk
    pairs = []
    for amp, tau in exp_decay_filters:
        first, second = exp_decay_filter(amp, tau, sample_rate)
        pairs.append((second, first))
    b, a = combine_filters(pairs)
    _, poles, _ = tf2zpk(b, a)
    return bool(np.all(np.abs(poles) < 1))
