"""ctypes wrapper of oracle/_build/libwfk_oracle.so (TEST INFRASTRUCTURE ONLY).

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the waveforms_amd product path."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libwfk_oracle.so')


def build(force=False):
    if force or not os.path.exists(LIB) or (
            os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, 'wfk_oracle.c'))):
        subprocess.run(['make', '-s', '-C', HERE], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.wfk_oracle_fir.restype = None
        _lib.wfk_oracle_grid.restype = None
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def eval_grid(prog, grid, want_complex=False):
    """prog: waveforms_amd._flatten.Program; grid: wfk_grid -> (n_channels, n) array."""
    n = int(grid.n)
    re = np.empty((prog.n_channels, max(n, 1)))
    im = np.empty_like(re) if want_complex else None
    rc = lib().wfk_oracle_eval_grid(C.byref(prog.struct), C.byref(grid), _ptr(re),
                                    _ptr(im) if want_complex else None,
                                    C.c_int64(re.shape[1]))
    if rc:
        raise RuntimeError(f'oracle rc={rc}')
    re = re[:, :n]
    return re + 1j * im[:, :n] if want_complex else re


def eval_tlist(prog, t, want_complex=False):
    t = np.ascontiguousarray(t, dtype=np.float64)
    n = len(t)
    re = np.empty((prog.n_channels, max(n, 1)))
    im = np.empty_like(re) if want_complex else None
    rc = lib().wfk_oracle_eval_tlist(C.byref(prog.struct), _ptr(t), C.c_int64(n),
                                     _ptr(re), _ptr(im) if want_complex else None,
                                     C.c_int64(re.shape[1]))
    if rc:
        raise RuntimeError(f'oracle rc={rc}')
    re = re[:, :n]
    return re + 1j * im[:, :n] if want_complex else re


def member_index(prog, member, grid=None, t=None):
    nb = len(prog.member_bounds(member))
    idx = np.empty(nb, dtype=np.int64)
    if t is not None:
        t = np.ascontiguousarray(t, dtype=np.float64)
    lib().wfk_oracle_member_index(C.byref(prog.struct),
                                  C.byref(grid) if grid is not None else None,
                                  _ptr(t) if t is not None else None,
                                  C.c_int64(len(t) if t is not None else 0),
                                  C.c_int32(member), _ptr(idx))
    return idx


def grid_values(grid):
    t = np.empty(int(grid.n))
    lib().wfk_oracle_grid(C.byref(grid), _ptr(t))
    return t


def fir(sig, ker):
    sig = np.ascontiguousarray(sig, dtype=np.float64)
    ker = np.ascontiguousarray(ker, dtype=np.float64)
    out = np.empty_like(sig)
    lib().wfk_oracle_fir(_ptr(sig), C.c_int64(len(sig)), _ptr(ker),
                         C.c_int32(len(ker)), _ptr(out))
    return out
