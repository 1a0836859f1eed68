"""Reference tests/test_multi_drag.py through the drop-in API: the multi-notch DRAG pulses
have a spectral null at every block frequency (property test on 1 000 001 samples)."""
import numpy as np
import pytest

from waveforms_amd import drag_sin, drag_sinx

pytestmark = pytest.mark.gpu


def _null_index(wav, ttt, freq, bq):
    freq_list = (freq + np.linspace(-0.02e6, 0.02e6, 21) + bq).reshape([1, -1])
    ff = np.exp(-2j * np.pi * freq_list * (ttt.reshape([-1, 1])))
    return int(np.argmin(np.abs(wav(ttt) @ ff)))


@pytest.mark.parametrize('ctor', [drag_sin, drag_sinx])
def test_spectral_nulls(ctor):
    t0, freq, width, plateau = 0e-9, 5e9, 22.22e-9, 0
    np.random.seed(1234)
    delta = np.random.random() * 9.5e6 - 19e6
    block_freq = tuple(np.concatenate(
        (np.random.random([np.random.randint(4) + 1]) * 100e6 + 20e6,
         -np.random.random([np.random.randint(4) + 1]) * 100e6 - 20e6)))
    extra = (np.random.random() * 0.8 + 0.2, ) if ctor is drag_sinx else ()
    ttt = np.linspace(t0 - (width + plateau) * 10, t0 + (width + plateau) * 11, 1000001)

    def pulse(bf):
        I = ctor(freq, width, plateau, delta, bf, 0, t0, *extra)
        Q = ctor(freq, width, plateau, delta, bf, -np.pi / 2, t0, *extra)
        return I - 1j * Q

    wav = pulse(block_freq)
    for bq in block_freq:
        assert _null_index(wav, ttt, freq, bq) == 10
    single = np.random.random() * 100e6 + 20e6
    assert _null_index(pulse(single), ttt, freq, single) == 10
