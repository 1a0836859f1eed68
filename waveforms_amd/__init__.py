"""waveforms_amd -- MI355X-native sampling engine behind the feihoo87/waveforms API.

Public names follow the reference's export list (waveforms/__init__.py:3-11).
Sampling (`Waveform.__call__`, `Waveform.sample`, `WaveVStack.__call__`) and
`distortion.predistort(ker=...)` run as hand-written HIP kernels for gfx950
behind the C-ABI of include/wfk.h; see DESIGN.md.
"""
from numpy import e, pi

from .waveform import (D, Waveform, WaveVStack, chirp, const, cos, cosh,
                       coshPulse, cosPulse, cut, drag, exp, function, gaussian,
                       general_cosine, hanning, interp, mixing, mollifier, one, poly,
                       registerBaseFunc, registerDerivative, samplingPoints,
                       sign, sin, sinc, sinh, slepian, square, step, t, zero)

from . import _waveform
from .multy_drag import drag_sin, drag_sinx
from .waveform_parser import wave_eval

__version__ = "0.1.0"
