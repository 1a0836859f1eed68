"""ctypes binding of libwfk_hip.so (include/wfk.h).

This is the only door from Python to the sampler.  There is no fallback: if the
shared library is missing it cannot be imported, and if no GPU is visible every
launch raises.  ctypes releases the GIL during calls.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

import numpy as np

from ._flatten import Program, wfk_grid, wfk_program

_HERE = os.path.dirname(os.path.abspath(__file__))
# WFK_LIB: developer knob for A/B builds of the same library (tools/*_bench.py)
LIB_PATH = os.environ.get('WFK_LIB') or os.path.join(_HERE, 'csrc', 'libwfk_hip.so')

OUT_F64, OUT_F32, OUT_C128, OUT_C64 = 0, 1, 2, 3
ACCUMULATE = 1
_KIND_OF = {np.dtype(np.float64): OUT_F64, np.dtype(np.float32): OUT_F32,
            np.dtype(np.complex128): OUT_C128, np.dtype(np.complex64): OUT_C64}
_DTYPE_OF = {v: k for k, v in _KIND_OF.items()}

E_UNSUP = -2
E_TIMEOUT = -5


class wfk_plan_info(C.Structure):
    _fields_ = [('n_channels', C.c_int32), ('n', C.c_int64), ('tile', C.c_int32),
                ('n_tiles', C.c_int64), ('n_pieces', C.c_int32),
                ('param_doubles', C.c_int64), ('n_fast', C.c_int32),
                ('n_direct', C.c_int32), ('n_fused', C.c_int32),
                ('n_generic', C.c_int32)]


class EngineError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libwfk_hip.so (built by waveforms_amd/csrc/Makefile or
    __graft_entry__.build()).  Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f'{LIB_PATH} not found: build it with `make -C waveforms_amd/csrc` '
                f'(hipcc --offload-arch=gfx950). waveforms_amd has no CPU sampling path.')
        # PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 / librocfft under
        # the same SONAMEs as /opt/rocm.  Whichever is loaded first serves the whole
        # process, and a system runtime loaded before torch leaves torch without a GPU.
        # So when torch is installed, let it load its runtime first.
        if importlib.util.find_spec('torch') is not None:
            import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        l.wfk_last_error.restype = C.c_char_p
        P, I64, I32, VP = C.POINTER, C.c_int64, C.c_int32, C.c_void_p
        l.wfk_plan_create_grid.argtypes = [P(wfk_program), P(wfk_grid), P(VP)]
        l.wfk_plan_create_tlist.argtypes = [P(wfk_program), VP, I64, P(VP)]
        l.wfk_grid_detect.argtypes = [VP, I64, P(wfk_grid)]
        l.wfk_grid_detect_runs.argtypes = [VP, I64, I64, I32, VP, P(wfk_grid)]
        l.wfk_plan_destroy.argtypes = [VP]
        l.wfk_plan_get_info.argtypes = [VP, P(wfk_plan_info)]
        l.wfk_plan_member_index.argtypes = [VP, I32, VP, I32]
        l.wfk_plan_channel_is_complex.argtypes = [VP, I32]
        l.wfk_plan_kernel_name.argtypes = [VP, C.c_int]
        l.wfk_plan_kernel_name.restype = C.c_char_p
        l.wfk_plan_table_bytes.argtypes = [VP]
        l.wfk_plan_table_bytes.restype = C.c_int64
        l.wfk_plan_launch.argtypes = [VP, VP, I64, C.c_int, C.c_uint32, VP]
        l.wfk_plan_run_host.argtypes = [VP, VP, I64, C.c_int]
        l.wfk_fir_plan_create.argtypes = [VP, I32, I64, I32, C.c_int, P(VP)]
        l.wfk_fir_plan_create_rows.argtypes = [VP, I32, I64, I32, C.c_int, P(VP)]
        l.wfk_chain_plan_create_rows.argtypes = [P(wfk_program), P(wfk_grid), VP, I32, C.c_int, P(VP)]
        l.wfk_fir_apply.argtypes = [VP, VP, I64, VP, I64, VP]
        l.wfk_fir_plan_destroy.argtypes = [VP]
        l.wfk_chain_plan_create.argtypes = [P(wfk_program), P(wfk_grid), VP, I32, C.c_int, P(VP)]
        l.wfk_chain_is_fused.argtypes = [VP]
        l.wfk_chain_unfused_reason.argtypes = [VP]
        l.wfk_chain_unfused_reason.restype = C.c_char_p
        l.wfk_chain_kernel_name.argtypes = [VP]
        l.wfk_chain_kernel_name.restype = C.c_char_p
        l.wfk_chain_table_bytes.argtypes = [VP]
        l.wfk_chain_table_bytes.restype = I64
        l.wfk_chain_launch.argtypes = [VP, VP, I64, VP]
        l.wfk_chain_plan_destroy.argtypes = [VP]
        l.wfk_chain_iir_plan_create.argtypes = [P(wfk_program), P(wfk_grid), I32, VP, VP, VP, VP, I32, I32, C.c_int, P(VP)]
        l.wfk_chain_iir_is_fused.argtypes = [VP]
        l.wfk_chain_iir_unfused_reason.argtypes = [VP]
        l.wfk_chain_iir_unfused_reason.restype = C.c_char_p
        l.wfk_chain_iir_kernel_name.argtypes = [VP]
        l.wfk_chain_iir_kernel_name.restype = C.c_char_p
        l.wfk_chain_iir_table_bytes.argtypes = [VP]
        l.wfk_chain_iir_table_bytes.restype = I64
        l.wfk_chain_iir_state_dim.argtypes = [VP]
        l.wfk_chain_iir_launch.argtypes = [VP, VP, I64, VP, VP, C.c_double, VP]
        l.wfk_chain_iir_status.argtypes = [VP, VP]
        l.wfk_chain_iir_plan_destroy.argtypes = [VP]
        l.wfk_iir_plan_create.argtypes = [I32, VP, VP, VP, I64, I32, C.c_int, P(VP)]
        l.wfk_iir_state_dim.argtypes = [VP]
        l.wfk_iir_apply.argtypes = [VP, VP, I64, VP, I64, VP, VP, C.c_double, VP]
        l.wfk_iir_plan_destroy.argtypes = [VP]
        l.wfk_iir_status.argtypes = [VP, VP]
        l.wfk_spectral_plan_create.argtypes = [I64, I32, C.c_int, P(VP)]
        l.wfk_spectral_apply.argtypes = [VP, VP, VP, VP, VP]
        l.wfk_spectral_plan_destroy.argtypes = [VP]
        l.wfk_host_alloc.argtypes = [P(VP), C.c_size_t]
        l.wfk_host_free.argtypes = [VP]
        l.wfk_host_all_finite.argtypes = [VP, I64]
        l.wfk_malloc.argtypes = [P(VP), C.c_size_t]
        l.wfk_free.argtypes = [VP]
        l.wfk_memcpy_h2d.argtypes = [VP, VP, C.c_size_t]
        l.wfk_memcpy_d2h.argtypes = [VP, VP, C.c_size_t]
        l.wfk_memset.argtypes = [VP, C.c_int, C.c_size_t]
        l.wfk_stream_sync.argtypes = [VP]
        l.wfk_device_count.argtypes = [P(C.c_int)]
        l.wfk_set_device.argtypes = [C.c_int]
        if l.wfk_abi_version() != 2:
            raise EngineError('libwfk_hip.so ABI version mismatch')
        _lib = l
    return _lib


def check(rc):
    if rc < 0:
        msg = lib().wfk_last_error().decode('utf-8', 'replace')
        if rc == E_UNSUP:
            raise NotImplementedError(msg)
        raise EngineError(f'wfk error {rc}: {msg}')
    return rc


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().wfk_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def set_device(ordinal: int):
    check(lib().wfk_set_device(ordinal))


def detect_grid(t: np.ndarray):
    """-> wfk_grid if the float64 array `t` is bit-identical to a np.linspace / np.arange grid
    (exact element-wise check inside the library), else None."""
    if os.environ.get('WFK_NO_GRID_DETECT') == '1' or t.dtype != np.float64 or not t.flags.c_contiguous:
        return None
    g = wfk_grid()
    return g if lib().wfk_grid_detect(t.ctypes.data, len(t), C.byref(g)) == 1 else None


def detect_grid_runs(t: np.ndarray, min_len: int = 4096, max_runs: int = 256):
    """-> [(start, wfk_grid), ...] if the float64 array `t` is several NumPy grids back to back (each run
    verified element by element, at least `min_len` samples long), else None."""
    if os.environ.get('WFK_NO_GRID_DETECT') == '1' or t.dtype != np.float64 or not t.flags.c_contiguous:
        return None
    starts = np.zeros(max_runs, dtype=np.int64)
    grids = (wfk_grid * max_runs)()
    k = lib().wfk_grid_detect_runs(t.ctypes.data, len(t), min_len, max_runs, starts.ctypes.data, grids)
    if k <= 0:
        return None
    out = []
    for i in range(k):
        g = wfk_grid()
        C.memmove(C.byref(g), C.byref(grids[i]), C.sizeof(wfk_grid))
        out.append((int(starts[i]), g))
    return out


class Plan:
    """A compiled program bound to a time axis (grid or explicit t)."""

    def __init__(self, prog: Program, grid: wfk_grid | None = None, t=None):
        self.prog = prog
        self._h = C.c_void_p()
        if grid is not None:
            self.grid = grid
            check(lib().wfk_plan_create_grid(C.byref(prog.struct), C.byref(grid),
                                             C.byref(self._h)))
        else:
            t = np.ascontiguousarray(t, dtype=np.float64)
            check(lib().wfk_plan_create_tlist(C.byref(prog.struct), t.ctypes.data,
                                              len(t), C.byref(self._h)))
        info = wfk_plan_info()
        check(lib().wfk_plan_get_info(self._h, C.byref(info)))
        self.info = info
        self.n = int(info.n)
        self.n_channels = int(info.n_channels)

    def close(self):
        if self._h and _lib is not None:
            _lib.wfk_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def member_index(self, member: int) -> np.ndarray:
        """np.searchsorted(x - shift, bounds) of one member, computed by the library."""
        nb = len(self.prog.member_bounds(member))
        idx = np.empty(nb, dtype=np.int64)
        check(lib().wfk_plan_member_index(self._h, member, idx.ctypes.data, nb))
        return idx

    def kernel_name(self, dtype=np.float64) -> str:
        """Symbol of the kernel a launch with this output dtype runs (as rocprofv3 shows it)."""
        return lib().wfk_plan_kernel_name(self._h, _KIND_OF[np.dtype(dtype)]).decode()

    def table_bytes(self) -> int:
        """Bytes of device tables a launch reads next to the output stream."""
        return int(lib().wfk_plan_table_bytes(self._h))

    def launch(self, out_ptr: int, ch_stride: int, kind: int, accumulate=False,
               stream: int = 0):
        """Asynchronous launch into device memory at `out_ptr`."""
        check(lib().wfk_plan_launch(self._h, out_ptr, ch_stride, kind,
                                    ACCUMULATE if accumulate else 0, stream))

    def run_host(self, dtype=np.float64) -> np.ndarray:
        """Launch, copy back, synchronise -> (n_channels, n) NumPy array.  Big results live in a
        page-locked block from the library's cache (`pinned_empty`): no page faults, DMA straight into
        it, and for a single channel the copy overlaps the kernel part by part."""
        dtype = np.dtype(dtype)
        out = pinned_empty((self.n_channels, self.n), dtype)
        if out.size:
            check(lib().wfk_plan_run_host(self._h, out.ctypes.data, self.n,
                                          _KIND_OF[dtype]))
        return out

    def run_host_into(self, out: np.ndarray):
        """Launch and copy straight into the caller's C-contiguous array of n_channels * n elements
        (`Waveform.__call__(x, out=...)`): no temporary, no second pass over the data."""
        if out.size != self.n_channels * self.n or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError('out must be a writeable C-contiguous array of n_channels * n elements')
        if out.size:
            check(lib().wfk_plan_run_host(self._h, out.ctypes.data, self.n, _KIND_OF[out.dtype]))
        return out


PINNED_MIN_BYTES = 4 << 20


def all_finite(a: np.ndarray) -> bool:
    """no NaN / inf in a C-contiguous float64 / complex128 array (threaded scan inside the library)"""
    n = a.size * (2 if a.dtype == np.complex128 else 1)
    return bool(lib().wfk_host_all_finite(a.ctypes.data, n))


def pinned_empty(shape, dtype) -> np.ndarray:
    """np.empty, but results of >= 4 MB are backed by a page-locked block from the library's cache
    (wfk_host_alloc); the block goes back to the cache when the array and all its views are gone.
    Falls back to an ordinary array when pinned memory is not to be had."""
    import weakref
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    if nbytes < PINNED_MIN_BYTES:
        return np.empty(shape, dtype=dtype)
    p = C.c_void_p()
    if lib().wfk_host_alloc(C.byref(p), nbytes) != 0 or not p.value:
        return np.empty(shape, dtype=dtype)
    buf = (C.c_byte * nbytes).from_address(p.value)
    weakref.finalize(buf, lib().wfk_host_free, p.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class FirPlan:
    """out[i] = sum_k ker[k] * sig[i + K//2 - k] (zero padded) for `batch` rows.  A 2-D `ker` of shape
    (batch, K) gives every row its own kernel (one predistortion kernel per AWG line)."""

    def __init__(self, ker, n: int, batch: int = 1, dtype=np.float64):
        ker = np.ascontiguousarray(ker, dtype=np.float64)
        self.n, self.batch, self.dtype = int(n), int(batch), np.dtype(dtype)
        self._h = C.c_void_p()
        if ker.ndim == 2:
            if ker.shape[0] != self.batch:
                raise ValueError('per-row kernels: ker must have shape (batch, K)')
            check(lib().wfk_fir_plan_create_rows(ker.ctypes.data, ker.shape[1], self.n, self.batch,
                                                 _KIND_OF[self.dtype], C.byref(self._h)))
            return
        check(lib().wfk_fir_plan_create(ker.ctypes.data, len(ker), self.n, self.batch,
                                        _KIND_OF[self.dtype], C.byref(self._h)))

    def apply(self, in_ptr: int, in_stride: int, out_ptr: int, out_stride: int,
              stream: int = 0):
        check(lib().wfk_fir_apply(self._h, in_ptr, in_stride, out_ptr, out_stride,
                                  stream))

    def close(self):
        if self._h and _lib is not None:
            _lib.wfk_fir_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class ChainPlan:
    """sampler -> FIR for every channel of `prog` on `grid` (predistort(wav(t), ker=ker)); fused
    into ONE kernel when the program is fully fused (lean plan on a fine grid, or a pure short-tier
    plan at AWG sample rates) and K <= 1537 (`fused`, `why_not`, `kernel_name()`)."""

    def __init__(self, prog: Program, grid: wfk_grid, ker, dtype=np.float64):
        ker = np.ascontiguousarray(ker, dtype=np.float64)
        self.prog, self.grid, self.dtype = prog, grid, np.dtype(dtype)
        self.n, self.n_channels = int(grid.n), prog.n_channels
        self._h = C.c_void_p()
        if ker.ndim == 2:          # one kernel per channel
            if ker.shape[0] != prog.n_channels:
                raise ValueError('per-channel kernels: ker must have shape (n_channels, K)')
            check(lib().wfk_chain_plan_create_rows(C.byref(prog.struct), C.byref(grid), ker.ctypes.data, ker.shape[1],
                                                   _KIND_OF[self.dtype], C.byref(self._h)))
        else:
            check(lib().wfk_chain_plan_create(C.byref(prog.struct), C.byref(grid), ker.ctypes.data, len(ker),
                                              _KIND_OF[self.dtype], C.byref(self._h)))
        self.fused = bool(lib().wfk_chain_is_fused(self._h))
        self.why_not = lib().wfk_chain_unfused_reason(self._h).decode()

    def launch(self, out_ptr: int, out_stride: int, stream: int = 0):
        check(lib().wfk_chain_launch(self._h, out_ptr, out_stride, stream))

    def kernel_name(self) -> str:
        """'fir_sampled<T,HOPB>' (fine grids), 'fir_short<T,HOPB>' (AWG rates) or the unfused pair"""
        return lib().wfk_chain_kernel_name(self._h).decode()

    def table_bytes(self) -> int:
        return int(lib().wfk_chain_table_bytes(self._h))

    def close(self):
        if self._h and _lib is not None:
            _lib.wfk_chain_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def _pack_sections(sections):
    """list of (b, a) -> (orders int32, b flat, a flat), every section padded to max(len(b), len(a))"""
    orders, bs, as_ = [], [], []
    for b, a in sections:
        b = np.atleast_1d(np.asarray(b, dtype=np.float64))
        a = np.atleast_1d(np.asarray(a, dtype=np.float64))
        m = max(len(b), len(a))
        bs.append(np.concatenate([b, np.zeros(m - len(b))]))
        as_.append(np.concatenate([a, np.zeros(m - len(a))]))
        orders.append(m - 1)
    return (np.asarray(orders, dtype=np.int32), np.ascontiguousarray(np.concatenate(bs)),
            np.ascontiguousarray(np.concatenate(as_)))


class ChainIirPlan:
    """sampler -> IIR cascade (-> FIR) for every channel of `prog` on `grid`, device-resident:
    `sosfilt(sos, wav(t) - initial, zi) + initial` of Waveform.sample(filters=) (reference waveform.py:190-203,
    244-251) and `predistort(wav(t), filters, ker)` (distortion.py:298-337).  When the program is fully fused and the
    (first pass of the) cascade has a state dimension <= 4, the wave that owns a chunk of the IIR scan EVALUATES its
    input (`iir_sampled<...>`): the unfiltered samples never touch HBM (`fused`, `why_not`, `kernel_name()`).
    `ker` (K,) or (n_channels, K): an FIR stage behind the cascade."""

    def __init__(self, prog: Program, grid: wfk_grid, sections, ker=None, dtype=np.float64):
        orders, bflat, aflat = _pack_sections(sections)
        self.prog, self.grid, self.dtype = prog, grid, np.dtype(dtype)
        self.n, self.n_channels = int(grid.n), prog.n_channels
        self._h = C.c_void_p()
        kp, K, rows = None, 0, 0
        if ker is not None:
            ker = np.ascontiguousarray(ker, dtype=np.float64)
            if ker.ndim == 2 and ker.shape[0] != prog.n_channels:
                raise ValueError('per-channel kernels: ker must have shape (n_channels, K)')
            kp, K, rows = ker.ctypes.data, ker.shape[-1], int(ker.ndim == 2)
        check(lib().wfk_chain_iir_plan_create(C.byref(prog.struct), C.byref(grid), len(orders), orders.ctypes.data,
                                              bflat.ctypes.data, aflat.ctypes.data, kp, K, rows,
                                              _KIND_OF[self.dtype], C.byref(self._h)))
        self.state_dim = int(sum(orders))
        self.why_not = lib().wfk_chain_iir_unfused_reason(self._h).decode()

    @property
    def fused(self) -> bool:
        return bool(lib().wfk_chain_iir_is_fused(self._h))

    def launch(self, out_ptr: int, out_stride: int, zi_ptr=None, zf_ptr=None, initial=0.0, stream: int = 0) -> bool:
        """-> False when the library refused the launch with WFK_ETIMEOUT (an earlier launch of this plan ran into a
        look-back timeout; the plan has switched to the unfused three-launch form: launch again)"""
        rc = lib().wfk_chain_iir_launch(self._h, out_ptr, out_stride, zi_ptr, zf_ptr, float(initial), stream)
        if rc == E_TIMEOUT:
            return False
        check(rc)
        return True

    def status(self, stream=0) -> bool:
        """Synchronise `stream`; False if a launch since the last check timed out in a look-back (outputs hold NaN)"""
        rc = lib().wfk_chain_iir_status(self._h, stream)
        if rc == E_TIMEOUT:
            return False
        check(rc)
        return True

    def kernel_name(self) -> str:
        return lib().wfk_chain_iir_kernel_name(self._h).decode()

    def table_bytes(self) -> int:
        return int(lib().wfk_chain_iir_table_bytes(self._h))

    def close(self):
        if self._h and _lib is not None:
            _lib.wfk_chain_iir_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class IirPlan:
    """Cascade of direct-form-II-transposed sections along each of `batch` rows.

    sections: list of (b, a) coefficient sequences; a section's order is
    max(len(b), len(a)) - 1 (scipy.signal.lfilter convention).  An SOS matrix is the
    list [(row[:3], row[3:]) for row in sos] (scipy.signal.sosfilt)."""

    def __init__(self, sections, n: int, batch: int = 1, dtype=np.float64):
        orders, bflat, aflat = _pack_sections(sections)
        self.n, self.batch, self.dtype = int(n), int(batch), np.dtype(dtype)
        self._h = C.c_void_p()
        check(lib().wfk_iir_plan_create(len(orders), orders.ctypes.data, bflat.ctypes.data,
                                        aflat.ctypes.data, self.n, self.batch,
                                        _KIND_OF[self.dtype], C.byref(self._h)))
        self.state_dim = int(sum(orders))

    def apply(self, in_ptr, in_stride, out_ptr, out_stride, zi_ptr=None, zf_ptr=None,
              initial=0.0, stream=0):
        """-> False when the library refused the launch with WFK_ETIMEOUT: an EARLIER launch of this plan
        (for instance the other row of a complex waveform) ran into a look-back timeout; nothing was
        launched, the plan has switched to the three-launch form -- treat it like `status() == False`."""
        rc = lib().wfk_iir_apply(self._h, in_ptr, in_stride, out_ptr, out_stride, zi_ptr,
                                 zf_ptr, float(initial), stream)
        if rc == E_TIMEOUT:
            return False
        check(rc)
        return True

    def status(self, stream=0) -> bool:
        """Synchronise `stream`; False if a single-pass launch since the last check ran into a look-back
        timeout (its outputs hold NaN; the plan has switched to the three-launch form: apply again)."""
        rc = lib().wfk_iir_status(self._h, stream)
        if rc == E_TIMEOUT:
            return False
        check(rc)
        return True

    def close(self):
        if self._h and _lib is not None:
            _lib.wfk_iir_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class SpectralPlan:
    """out = irfft(rfft(x) * H) along each of `batch` contiguous rows of n samples."""

    def __init__(self, n: int, batch: int = 1, dtype=np.float64):
        self.n, self.batch, self.dtype = int(n), int(batch), np.dtype(dtype)
        self._h = C.c_void_p()
        check(lib().wfk_spectral_plan_create(self.n, self.batch, _KIND_OF[self.dtype],
                                             C.byref(self._h)))

    def apply(self, in_ptr, out_ptr, H_ptr, stream=0):
        check(lib().wfk_spectral_apply(self._h, in_ptr, out_ptr, H_ptr, stream))

    def close(self):
        if self._h and _lib is not None:
            _lib.wfk_spectral_plan_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class DeviceBuffer:
    """Raw device allocation through the C-ABI (for callers without torch)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().wfk_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        check(lib().wfk_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes))

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(lib().wfk_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def zero(self):
        check(lib().wfk_memset(self.ptr, 0, self.nbytes))

    def close(self):
        if getattr(self, 'ptr', None) and _lib is not None:
            _lib.wfk_free(self.ptr)
            self.ptr = None

    __del__ = close


def sync(stream: int = 0):
    check(lib().wfk_stream_sync(stream))
