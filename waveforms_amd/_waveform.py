"""Names of the reference's extension module `waveforms._waveform` (a Cython module there),
for scripts that import from it directly: the primitive ids, the registry and the expression
algebra.  Evaluation itself (`calc_parts`) is not a Python function here -- it is the HIP
sampler behind `Waveform.__call__` (include/wfk.h)."""
from ._ir import (COS, COSH, D_GAUSSIAN, DRAG, ERF, EXP, EXPONENTIALCHIRP, GAUSSIAN,  # noqa: F401
                  HYPERBOLICCHIRP, INTERP, LINEAR, LINEARCHIRP, MOLLIFIER, NDIGITS, SINC, SINH,
                  add, is_const, mul, shift, wave_sum)
from ._ir import HALF as _half, ONE as _one, ZERO as _zero  # noqa: F401
from ._ir import const_expr as _const, derivative as _D, power as pow, primitive as basic_wave  # noqa: F401
from .waveform import (_baseFunc, packBaseFunc, registerBaseFunc, registerDerivative,  # noqa: F401
                       updateBaseFunc)
