"""Helper of the user-callable tests: the function library the ORACLE should use for a
product-side `function_lib` (built-in markers -> the oracle's own NumPy primitives)."""
from oracle import np_oracle
from waveforms_amd.waveform import BuiltinPrimitive, _baseFunc


def oracle_lib(w, lib):
    if lib is None:
        lib = getattr(w, 'function_lib', None) or _baseFunc
    out = {}
    for tid, fn in lib.items():
        out[tid] = np_oracle.PRIMITIVES[fn.type_id] if isinstance(fn, BuiltinPrimitive) else fn
    return out
