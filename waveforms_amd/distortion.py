"""FIR stage of the hot path: `predistort(sig, ker=...)` on the GPU.

Signature and semantics follow the reference's `predistort`
(waveforms/distortion.py:289-337).  Only the FIR branch (`ker=`) is on the device
path this round: out[i] = sum_k ker[k] * sig[i + len(ker)//2 - k] with zero padding,
what the reference computes with one giant `scipy.signal.fftconvolve`.  The IIR
branch (`filters=`) is SURVEY.md §8(f) N1 and raises NotImplementedError.
"""
from __future__ import annotations

import numpy as np

from . import _engine


class FirStage:
    """Device-resident FIR stage for `batch` rows of `n` samples (build once, apply
    many times).  `apply_torch(x, y)`: x, y (batch, >= n) device tensors."""

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


def fir_host(sig: np.ndarray, ker: np.ndarray) -> np.ndarray:
    """NumPy in, NumPy out: upload, rocFFT overlap-save on the device, download."""
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


def predistort(sig, filters=None, ker=None, initial=0.0, initial_x=None,
               initial_y=None, zi=None, return_zf=False):
    """reference: waveforms/distortion.py:289-337 (FIR branch on the GPU)."""
    if filters is not None:
        raise NotImplementedError(
            'IIR pre-distortion (filters=) is not implemented on the device yet '
            '(SURVEY.md §8(f) N1); only the FIR branch ker= is')
    sig = np.asarray(sig)
    if ker is None:
        return sig
    if return_zf:
        raise NotImplementedError('return_zf requires the IIR branch')
    return fir_host(sig, np.asarray(ker, dtype=np.float64))
