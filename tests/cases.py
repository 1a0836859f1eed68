"""Pulse scripts shared by the golden-vector generator (run against the imported
reference, oracle/make_golden.py) and by the parity tests (run against
waveforms_amd).  Each case: name -> (builder(ns) -> Waveform | WaveVStack, grid).

Grid descriptors: ('linspace', a, b, n, endpoint) | ('arange', a, b, step).
Cases mirror what the reference's own tests sample (tests/test_waveform.py,
tests/test_wavevstack.py in the reference) plus edge cases for every built-in
primitive, clip, complex amplitudes, powers and piece-boundary alignment.
"""
import numpy as np
from numpy import pi

from waveforms_amd import workloads as wl

LIN = ('linspace', -10.0, 10.0, 1001, True)

# THE fp32 bound: a float launch is within this fraction of the waveform's peak of the fp64 oracle, in every
# test and every soak (tools/*_soak.py).  north_star's contract is 1e-3; the kernels compute in double and
# round once on the store, so what remains is float rounding of the result (6e-8) plus the float phasor /
# envelope recurrences of the fused tiers (worst observed over the soaks: 3.03e-5, chain_soak seed 29703).
FP32_TOL = 5e-5
# float launches of scripts moved 10 us .. 10 ms from t = 0 (tools/fuzz_soak.py far / awgfar): the lean kernel's float phasor /
# envelope recurrences on terms that cancel reach 1.6e-4 of peak in 1 of 3000-6000 scripts (worst observed); contract 1e-3
FP32_FAR_TOL = 2e-4

# THE fp64 bounds, one per tier, as fractions of max(1, peak) of the fp64 oracle (north_star's contract: < 1e-9 absolute
# on O(1) waveforms).  Tests and soaks (tools/*_soak.py) hold THESE numbers; a test may assert something tighter for a
# case it knows (the README pulses sit at 1e-12), never something looser.
#   grid tiers (lean, short, general, the FIR / IIR chains' samplers): a fused group is admitted only while
#       |rate| x (rounding of NumPy's grid time) <= 2.5e-10, closing multipliers add up to 1e-10 (worst soak: 3.5e-10)
#   time lists, fused groups evaluated pointwise: the same 2.5e-10 admission on 2.4e-16 |t| rate, + the inline
#       sincos / exp (2 ulp) -- worst of 16 000 soak scripts near t = 0 1.9e-11; the scripts moved 10 us .. 10 ms out
#       (tools/fuzz_soak.py far / awgfar, round 5) reach 7.4e-10 on grid and time-list launches alike: the contract
#   time lists, libm tier (one device-libm call per factor on the caller's own times): 1e-11
#   IIR stages (blocked scan against SciPy's sequential recurrence, any of the three execution forms): 1e-10 for cascades
#       of first- and second-order sections (what sample(filters=sos) and exp_decay_filter give); ONE section of order
#       3 / 4 in direct form with clustered poles -- butter(4, 0.022) as a single (b, a) -- is ill-conditioned in any
#       evaluation order: 5e-10 (iirchain_soak seed 12713: the single-pass scan 1.1e-10 from a long-double recursion, SciPy
#       1.7e-11; sections whose transition powers grow past 1e3 take the three-launch form since: 2.4e-11)
FP64_GRID_TOL = 1e-9
FP64_TLIST_FUSED_TOL = 1e-9
FP64_TLIST_LIBM_TOL = 1e-11
FP64_IIR_TOL = 1e-10
FP64_IIR_ORDER34_TOL = 5e-10
#   FIR stage alone (LDS-FFT / rocFFT overlap-save against the time-domain definition, random kernels up to 7000 taps): 1e-11
#   IIR stage alone on RANDOM stable sections of order <= 9 (real poles up to 0.999, tools/stage_soak.py): the contract, 1e-9
FP64_FIR_TOL = 1e-11
FP64_IIR_RANDOM_TOL = 1e-9


def _ref_tolist_case(ns):          # reference tests/test_waveform.py:38-48
    p = ns.gaussian(10) >> 5
    p += ns.gaussian(10) >> 50
    return p * ns.cos(200)


def _vstack4(ns):                  # reference tests/test_wavevstack.py:10-26
    return ns.WaveVStack([ns.cos(1), ns.sin(2), ns.gaussian(3),
                          ns.poly([1, -1 / 2, 1 / 6, -1 / 12])])


def _vstack_ops(ns):               # reference tests/test_wavevstack.py:46-88
    w = _vstack4(ns)
    return ((w * ns.sin(2) + 3) >> 0.6) + ns.sin(2)


def _vstack_shift(ns):
    return _vstack4(ns) << 1.4


def _complex_amp(ns):              # reference tests/test_waveform.py:89-93
    return 1j * (ns.cos(9) >> 1) + 1 * (ns.cos(9) >> 2) - 1j * (ns.cos(9) >> 3)


def _exp_complex(ns):              # reference tests/test_waveform.py:96-105
    return 2 * (ns.exp(1.01 + 22j)**2 << 1) * ns.exp(1.01 + 22j)


def _clip(ns):
    w = ns.cut(3 * ns.gaussian(4) * ns.cos(5), start=-2.5, stop=2.0, min=-0.7,
               max=1.1)
    return w


def _readme_x(ns):
    return wl.readme_xy(ns)[0]


def _readme_y(ns):
    return wl.readme_xy(ns)[1]


def _mix_block(ns):
    I, Q = ns.mixing(ns.gaussian(30e-9) >> 40e-9, ns.cosPulse(30e-9) >> 40e-9,
                     freq=37e6, phase=0.4, phaseDiff=0.05, ratioIQ=0.9,
                     block_freq=-150e6)
    return I + 0.5 * Q


def _mix_env(ns):
    I, Q = ns.mixing(ns.square(50e-9, edge=10e-9) >> 40e-9, phase=1.1,
                     DRAGScaling=3e-9)
    return I - Q


CASES = {
    # --- reference-test mirrors -------------------------------------------
    'cos1': (lambda ns: ns.cos(1), LIN),
    'sin1': (lambda ns: ns.sin(1), LIN),
    'gauss2': (lambda ns: ns.gaussian(2), LIN),
    'gauss2_shift3': (lambda ns: ns.gaussian(2) >> 3, LIN),
    'poly': (lambda ns: ns.poly([1, -1 / 2, 1 / 6, -1 / 12]), LIN),
    'square_shift': (lambda ns: ns.square(20e-9) >> 40e-9,
                     ('linspace', 0.0, 2000e-9, 8000, True)),
    'ref_tolist': (_ref_tolist_case, ('linspace', -5.0, 60.0, 2601, True)),
    'op_add': (lambda ns: ns.cos(1) + ns.sin(2), LIN),
    'op_sub': (lambda ns: ns.cos(1) - ns.sin(2), LIN),
    'op_mul': (lambda ns: ns.cos(1) * ns.sin(2), LIN),
    'op_div': (lambda ns: ns.cos(1) / 2, LIN),
    'trig3': (lambda ns: ns.cos(1) * ns.sin(2) * ns.cos(3, 4), LIN),
    'complex_amp': (_complex_amp, ('linspace', -2.0, 2.0, 1001, True)),
    'exp_complex': (_exp_complex, ('linspace', -2.0, 2.0, 1001, True)),
    'chirp_lin': (lambda ns: ns.chirp(1, 2, 10, 4, 'linear'),
                  ('linspace', 0.0, 10.0, 1000, False)),
    'chirp_exp': (lambda ns: ns.chirp(1, 2, 10, 4, 'exponential'),
                  ('linspace', 0.0, 10.0, 1000, False)),
    'chirp_hyp': (lambda ns: ns.chirp(1, 2, 10, 4, 'hyperbolic'),
                  ('linspace', 0.0, 10.0, 1000, False)),
    'vstack4': (_vstack4, LIN),
    'vstack_ops': (_vstack_ops, LIN),
    'vstack_shift': (_vstack_shift, LIN),
    'vstack_empty': (lambda ns: ns.WaveVStack([]) + 1.5, LIN),
    # --- primitives & constructors ------------------------------------------
    'step_erf': (lambda ns: ns.step(2.0) >> 1, LIN),
    'step_cos': (lambda ns: ns.step(2.0, 'cos') >> 1, LIN),
    'step_lin': (lambda ns: ns.step(2.0, 'linear') >> 1, LIN),
    'step0': (lambda ns: ns.step(0) >> 0.5, LIN),
    'sign': (lambda ns: ns.sign(), LIN),
    'square_erf': (lambda ns: ns.square(8, edge=2), LIN),
    'square_cos': (lambda ns: ns.square(8, edge=2, type='cos'), LIN),
    'square_lin': (lambda ns: ns.square(8, edge=2, type='linear'), LIN),
    'gauss_plateau': (lambda ns: ns.gaussian(4, plateau=6), LIN),
    'dgauss2': (lambda ns: ns.gaussian(6, d=2), LIN),
    'dgauss3_plateau': (lambda ns: ns.gaussian(4, plateau=3, d=3), LIN),
    'cospulse': (lambda ns: ns.cosPulse(7) >> 1.5, LIN),
    'cospulse_plateau': (lambda ns: ns.cosPulse(4, plateau=5), LIN),
    'coshpulse': (lambda ns: ns.coshPulse(9, eps=2.5), LIN),
    'coshpulse_plateau': (lambda ns: ns.coshPulse(6, eps=1.0, plateau=4), LIN),
    'sinc': (lambda ns: ns.sinc(0.8) >> 0.3,
             ('linspace', -70.0, 70.0, 2801, True)),
    'exp_real': (lambda ns: ns.exp(-0.3) * ns.step(0), LIN),
    'cosh_sinh': (lambda ns: 0.01 * ns.cosh(0.4) - 0.02 * ns.sinh(0.35), LIN),
    'mollifier0': (lambda ns: ns.mollifier(12.0), LIN),
    'mollifier1': (lambda ns: ns.mollifier(12.0, d=1), LIN),
    'mollifier2_plateau': (lambda ns: ns.mollifier(8.0, plateau=4.0, d=2), LIN),
    'general_cosine': (lambda ns: ns.general_cosine(16.0, 1.0, 0.3, 0.1), LIN),
    't_lin': (lambda ns: 0.5 * ns.poly([1, 0.5]) + 1, LIN),
    'drag_plain': (lambda ns: ns.drag(0.4, 12.0, t0=-6.0), LIN),
    'drag_block': (lambda ns: ns.drag(0.4, 12.0, delta=0.05, block_freq=0.9,
                                      phase=0.3, t0=-7.0), LIN),
    'drag_plateau': (lambda ns: ns.drag(0.4, 6.0, plateau=5.0, delta=0.02,
                                        block_freq=-0.5, t0=-8.0), LIN),
    'sampling_points': (lambda ns: ns.samplingPoints(
        -4.0, 6.0, [0.0, 1.0, -0.5, 2.0, 0.25, -1.0, 0.0]), LIN),
    # --- multi-notch DRAG primitives (ids 16/17, reference multy_drag.py) ------
    'mdrag_sin': (lambda ns: ns.drag_sin(0.4, 14.0, 0, 0.013, (0.9, 0.55, -0.7), 0.3, -7.0),
                  LIN),
    'mdrag_sin_plateau': (lambda ns: ns.drag_sin(0.4, 8.0, 5.0, -0.02, (0.8, -0.6, 1.3, -1.1, 0.45),
                                                 -0.4, -6.5), LIN),
    'mdrag_sin_single': (lambda ns: ns.drag_sin(0.3, 12.0, 0, 0.0, 0.75, 0.0, -6.0), LIN),
    'mdrag_sin_none': (lambda ns: ns.drag_sin(0.3, 12.0, 0, 0.01, None, 0.2, -6.0), LIN),
    'mdrag_sinx': (lambda ns: ns.drag_sinx(0.4, 14.0, 0, 0.013, (0.9, 0.55, -0.7), 0.3, -7.0,
                                           0.45), LIN),
    'mdrag_sinx_plateau': (lambda ns: ns.drag_sinx(0.4, 8.0, 5.0, -0.02, (0.8, -0.6, 1.3, -1.1),
                                                   -0.4, -6.5, 0.7), LIN),
    # --- algebra -------------------------------------------------------------
    'pow2': (lambda ns: (ns.cos(1.3) + 0.5)**2, LIN),
    'pow3_term': (lambda ns: (ns.gaussian(9) * ns.cos(2))**3, LIN),
    'pow_frac': (lambda ns: (ns.gaussian(9) >> 1)**0.5, LIN),
    'pow_neg': (lambda ns: (ns.cosh(0.2))**-2, LIN),
    'deriv2': (lambda ns: ns.D(ns.gaussian(8) * ns.cos(3), 2), LIN),
    'deriv_erf': (lambda ns: ns.D(ns.square(8, edge=2)), LIN),
    'deriv_misc': (lambda ns: ns.D(ns.exp(-0.2) * ns.sinh(0.1) +
                                   ns.chirp(0.1, 0.3, 10, 1.0)), LIN),
    'clip': (_clip, LIN),
    'clip_min_pos': (lambda ns: ns.cut(ns.gaussian(6), min=0.2, max=0.8), LIN),
    # --- mixing / DRAG -------------------------------------------------------
    'readme_x': (_readme_x, wl.readme_grid()),
    'readme_y': (_readme_y, wl.readme_grid()),
    'mix_block': (_mix_block, ('linspace', 0.0, 100e-9, 4001, True)),
    'mix_env': (_mix_env, ('linspace', -20e-9, 100e-9, 3001, False)),
    'c2_small': (lambda ns: wl.sum_channel(ns, 7, 3),
                 ('linspace', 0.0, 7 * wl.SPAN, 5000, False)),
    'c3_small': (lambda ns: wl.vstack_channel(ns, 5, 101),
                 ('linspace', 0.0, 5 * wl.SPAN, 4000, False)),
    'c2_duty30': (lambda ns: wl.sum_channel(ns, 5, 4, 100e-9),
                  ('arange', 0.0, 500e-9, 0.125e-9)),
    # --- grid edge cases -----------------------------------------------------
    'single_point': (lambda ns: ns.gaussian(2) * ns.cos(3),
                     ('linspace', 0.25, 0.25, 1, True)),
    'all_outside': (lambda ns: ns.gaussian(2) >> 100, LIN),
    # coarse grids: the Gaussian recurrence runs with a large per-step ratio and a
    # seed far outside the piece (H ~ 1), or must fall back to libm (H ~ 1.9)
    'coarse_gauss_rec': (lambda ns: (ns.gaussian(3.3302184446307908) * ns.cos(2.0)) >> 7.3,
                         ('linspace', -50.0, 50.0, 6400, True)),
    'coarse_gauss_direct': (lambda ns: (ns.gaussian(3.3302184446307908) * ns.cos(2.0)) >> 7.3,
                            ('linspace', -50.0, 50.0, 3400, True)),
    'coarse_exp': (lambda ns: ns.exp(-0.9) * ns.square(60) * ns.cos(0.3),
                   ('linspace', -50.0, 50.0, 2500, True)),
    'tiny_pieces': (lambda ns: wl.sum_channel(ns, 40, 9, 2e-9),
                    ('linspace', 0.0, 100e-9, 777, True)),
}


def sos_cases():
    """Waveform.sample() (np.arange grid) cases: name -> (builder, start, stop, rate)."""
    return {
        'sample_cos': (lambda ns: ns.cos(1), -10, 10.02, 50),
        'sample_readme': (_readme_x, -1e-6, 9e-6, 1e9),
        'sample_vstack': (_vstack4, -10, 10.02, 50),
        'sample_odd': (lambda ns: ns.gaussian(3) * ns.cos(7) >> 0.1, -4.0, 4.3,
                       123.456),
    }


def iir_cases():
    """sample(filters=(sos, initial)) cases: name -> (builder, start, stop, rate,
    butter order, cutoff, initial).  First two mirror the reference's test_filters
    (tests/test_waveform.py:169-194, tests/test_wavevstack.py:113-137)."""
    return {
        'iir_step': (lambda ns: ns.step(0), -1, 1, 1000, 3, 4.0, 0),
        'iir_vstack': (lambda ns: ns.WaveVStack([ns.step(0) << 0.5, -ns.step(0)]),
                       -1, 1, 1000, 3, 4.0, 0),
        'iir_initial': (lambda ns: 0.3 + ns.square(0.8, edge=0.1) * ns.cos(40.0),
                        -1, 1.5, 4000, 4, 90.0, 0.3),
        'iir_long': (lambda ns: ns.square(30.0) * ns.cos(3.0) >> 40,
                     0, 80, 1500, 2, 7.0, 0),
        # complex-valued waveform through the (real) sections: scipy.signal.sosfilt takes complex input
        'iir_cplx': (lambda ns: (1 + 0.5j) * ns.square(0.8, edge=0.1) * ns.cos(40.0) + 0.2j * (ns.gaussian(0.5) >> 0.6),
                     -1, 1.5, 4000, 4, 90.0, 0.3),
        # fully fused trees on rows long enough for the single-pass scan: the sampler runs INSIDE the IIR pass
        # (wfk_chain_iir_*, iir_sampled: rows of >= 32 768 samples); 42 000 samples, chunked in 36 000s (the first chunk
        # fuses, the last does not)
        'iir_fused2': (lambda ns: 0.1 + wl.sum_channel(ns, 14, 31), 0, 420e-9, 1e11, 4, 2e9, 0.1),
        'iir_fused1': (lambda ns: wl.vstack_channel(ns, 14, 32), 0, 420e-9, 1e11, 2, 5e9, 0),
    }


def iir_chunk(name):
    """chunk_size of the chunked sample() of an iir_case"""
    return 36000 if name.startswith('iir_fused') else 300


def predistort_cases():
    """predistort(sig, filters=[exp_decay_filter(A, tau, 1e9)...], ker=...) cases:
    (n, [(A, tau)...], initial, fir_taps)."""
    return [
        (5000, [(0.02, 150e-9)], 0.0, 0),
        (60000, [(0.02, 150e-9), (-0.01, 2.5e-6)], 0.25, 0),
        (30000, [(0.03, 40e-9), (0.01, 900e-9), (-0.005, 20e-6)], -0.1, 33),
        (9, [(0.05, 10e-9)], 1.0, 0),
    ]


def predistort_inputs(i):
    n, _, initial, k = predistort_cases()[i]
    rng = np.random.default_rng(700 + i)
    sig = rng.normal(size=n) + initial
    ker = rng.normal(size=k) if k else None
    return sig, ker


def predistort_high_cases():
    """predistort(filters=) of combined order 17 and 20 where the reference's direct-form lfilter is still
    finite and accurate: poles spread over 0.01..0.6 (time constants of 0.2-2 samples; the generator prints
    |reference - scipy cascade| <= 2e-11 for these.  With poles up to 0.8 the reference is already 3e-9 off
    the cascade, up to 0.95 it is 1e-3 off, with 40-sample time constants it returns 1e70):
    (n, [(A, tau)...], initial).  Bounds the product's cascade form against the reference by a number."""
    out = []
    for order, seed, initial in ((17, 0, 0.0), (20, 1, 0.25), (18, 2, -0.4 + 0.3j)):
        rng = np.random.default_rng(900 + seed)
        poles = np.linspace(0.01, 0.6, order) * rng.uniform(0.98, 1.02, order)
        taus = -1.0 / np.log(poles) / 1e9
        amps = rng.uniform(-0.02, 0.02, order)
        out.append((3000, list(zip(amps, taus)), initial))
    return out


def predistort_high_input(i):
    n, _, initial = predistort_high_cases()[i]
    rng = np.random.default_rng(950 + i)
    sig = np.concatenate([np.zeros(50), np.ones(n - 50)]) * 0.5 + 0.05 * rng.normal(size=n) + initial
    return sig


def predistort_cplx_cases():
    """predistort on COMPLEX inputs (scipy's lfilter / fftconvolve take them, reference distortion.py:298-337):
    (n, [(A, tau)...] | None, initial, fir_taps, real signal?, complex kernel?, explicit complex zi?)."""
    return [
        (4000, [(0.02, 150e-9)], 0.0, 0, False, False, False),
        (3000, None, 0.0, 33, False, False, False),                 # FIR branch, complex signal
        (3000, None, 0.0, 33, True, True, False),                   # FIR branch, REAL signal, complex kernel
        (5000, [(0.03, 40e-9), (0.01, 900e-9)], 0.3, 17, False, False, False),
        (2500, [(0.02, 150e-9)], 0.0, 9, False, True, True),        # complex signal, kernel and state
    ]


def predistort_cplx_inputs(i):
    n, params, initial, k, real_sig, cker, czi = predistort_cplx_cases()[i]
    rng = np.random.default_rng(800 + i)
    sig = rng.normal(size=n) + initial
    if not real_sig:
        sig = sig + 1j * rng.normal(size=n)
    ker = None
    if k:
        ker = rng.normal(size=k) / k
        if cker:
            ker = ker + 1j * rng.normal(size=k) / k
    zi = None
    if czi:
        zi = rng.normal(size=len(params)) * 0.01 + 1j * rng.normal(size=len(params)) * 0.01
    return sig, ker, zi


def spectral_cases():
    """(n, A, tau, sample_rate) for reflection / correct_reflection / shift."""
    return [(4096, 0.1, 12.5e-9, 1e9), (10007, -0.2, 31e-9, 2e9), (30000, 0.05, 3e-9, 1e9)]


def spectral_input(i):
    n = spectral_cases()[i][0]
    rng = np.random.default_rng(900 + i)
    return np.cumsum(rng.normal(size=n)) / 30 + rng.normal(size=n) * 0.01


# ---- edge inputs for Waveform.__call__ / WaveVStack.__call__ (tests/test_gpu_edges.py) ----
def _edge_pulse(ns):
    return (ns.gaussian(2.0) >> 1.0) * ns.cos(9.0) + 0.25 * ns.square(3.0, edge=0.4)


def _edge_aligned(ns):
    return (ns.square(2.0) >> 1.0) * ns.cos(5) + (ns.square(1.0) >> 3.5) * ns.gaussian(1.0)


def _edge_tiny_pieces(ns):
    w = ns.zero()
    for k in range(300):
        w = w + ((ns.square(0.01) >> (0.01 * k + 0.005)) * (0.1 + 0.01 * k)) * ns.cos(1.0 + k)
    return w


def _edge_complex(ns):
    return (ns.gaussian(1.5) >> 0.5) * ns.exp(2j * pi * 1.3) * (0.5 - 0.25j)


def _edge_clip(ns):
    w = 3 * _edge_pulse(ns)
    w.min, w.max = -0.5, 0.75
    return w


def _edge_scaled(scale):
    def build(ns):
        return (ns.gaussian(20 * scale) >> (60 * scale)) * ns.cos(2 * pi * 0.21 / scale, 0.3) \
            + 0.5 * ((ns.cosPulse(30 * scale) >> (150 * scale)) * ns.sin(2 * pi * 0.05 / scale))
    return build


def _edge_xs():
    rng = np.random.default_rng(2024)
    rnd = np.sort(rng.uniform(-6, 6, 3001))
    dup = np.sort(np.concatenate([rnd[::7], rnd[::7], [-0.5, -0.5, 2.5, 2.5, 1.0]]))
    return [np.array([]), np.array([0.7]), np.array([1e9]), np.array([-1e9]),
            np.linspace(50, 60, 1001), np.linspace(-90, -80, 77),
            np.linspace(-6, 6, 5001), rnd, dup,
            np.linspace(-6, 6, 2), np.linspace(-6, 6, 63), np.linspace(-6, 6, 65),
            np.linspace(-6, 6, 1025), np.linspace(-6, 6, 16 * 1024 + 1)]


def edge_cases():
    """name -> (builder(ns), [x arrays]); x arrays are sorted but not always uniform."""
    xs = _edge_xs()
    out = {
        'pulse': (_edge_pulse, xs),
        'vstack': (lambda ns: ns.WaveVStack([_edge_pulse(ns), ns.sin(3) * ns.square(2)]) * 0.5 + 0.125, xs),
        'aligned': (_edge_aligned, [np.linspace(0, 8, 8 * 64 + 1), np.linspace(0, 8, 9),
                                    np.array([0.0, 2.0, 2.0, 3.0, 4.0, 4.0])]),
        'tiny_pieces': (_edge_tiny_pieces, [np.linspace(0, 3, 301), np.linspace(0, 3, 30001)]),
        'complex': (_edge_complex, xs[:6] + [np.linspace(-3, 3, 65), np.linspace(-3, 3, 1025)]),
        'clip': (_edge_clip, xs[6:9]),
        'carrier_only': (lambda ns: ns.cos(3.0, 0.2) * 0.5, xs[6:9]),
        'step_exp': (lambda ns: ns.step(0.0) * ns.exp(-0.5), xs[6:9]),
        'neg_half': (lambda ns: (1 - ns.step(0.0)) * ns.sin(2.0), xs[6:9]),
    }
    for scale in (1e-9, 1e-6, 1e3):
        out[f'scaled_{scale:g}'] = (_edge_scaled(scale), [np.linspace(0, 200 * scale, 20001)])
    return out


# ---- filter-design helpers of distortion.py (tests/test_symbolic_cpu.py) ----
def extract_cases():
    """(n, sample_rate, bw, skip) for extractKernel."""
    return [(2048, 1e9, None, 0), (4096, 2e9, 0.4e9, 16), (3001, 1e9, 0.7e9, 5)]


def extract_input(i):
    n = extract_cases()[i][0]
    rng = np.random.default_rng(1200 + i)
    a = np.cumsum(rng.normal(size=n)) / 20 + rng.normal(size=n)
    k = np.exp(-np.arange(24) / 5.0)
    b = np.convolve(a, k / k.sum(), mode='full')[:n] + 0.3 * a
    return a, b


def decay_old_cases():
    return [(0.05, 200e-9, 1e9), (-0.03, 1.5e-6, 2e9), (0.4, 30e-9, 1e9), (0.0, 1e-6, 1e9)]


def factor_cases():
    return [([1.0, -0.5], [1.0, -0.9]), ([0.2, 0.1, -0.05], [1.0, -1.5, 0.56]),
            ([2.0, -1.0], [1.0, -1.2, 0.35])]


def stable_cases():
    return [([(0.05, 200e-9)], 1e9), ([(0.05, 200e-9), (-0.02, 2e-6)], 1e9),
            ([(-0.9, 50e-9), (0.3, 10e-9)], 2e9), ([(5.0, 1e-9)], 1e9)]


# ---- random pulse scripts (tests/test_gpu_fuzz.py, tests/test_oracle_golden.py) ----
# `wf` is the namespace the script is built in: this package, or the real reference when
# oracle/make_golden.py generates tests/golden/fuzz.npz.  The rng draws do not depend on it.
def random_pulse(wf, rng, scale):
    kind = rng.integers(0, 9)
    w = scale * rng.uniform(0.5, 6.0)
    if kind == 0:
        p = wf.gaussian(w)
    elif kind == 1:
        p = wf.cosPulse(w, plateau=scale * rng.uniform(0, 2) * (rng.random() < 0.4))
    elif kind == 2:
        p = wf.square(w, edge=w * rng.uniform(0.05, 0.3) * (rng.random() < 0.6),
                      type=str(rng.choice(['erf', 'cos', 'linear'])))
    elif kind == 3:
        p = wf.gaussian(w, plateau=scale * rng.uniform(0.1, 2))
    elif kind == 4:
        p = wf.drag(rng.uniform(-2, 2) / scale, w, delta=rng.uniform(-0.1, 0.1) / scale,
                    block_freq=None if rng.random() < 0.3 else rng.uniform(1, 3) / scale,
                    phase=rng.uniform(0, 6), t0=-w / 2)
    elif kind == 5:
        p = wf.coshPulse(w, eps=rng.uniform(0.5, 3))
    elif kind == 6:
        p = wf.gaussian(w) * wf.poly([rng.uniform(-1, 1), rng.uniform(-1, 1) / scale,
                                      rng.uniform(-1, 1) / scale**2])
    elif kind == 7:
        p = wf.D(wf.gaussian(w)) * (scale * 0.3)
    else:
        p = wf.mollifier(w)
    if rng.random() < 0.8:
        f = rng.uniform(-3, 3) / scale
        no_d = kind == 4    # the DRAG primitive has no derivative rule (nor upstream)
        I, Q = wf.mixing(p, freq=f, phase=rng.uniform(0, 6),
                         DRAGScaling=None if (no_d or rng.random() < 0.4)
                         else rng.uniform(-0.05, 0.05) * scale,
                         block_freq=None if (no_d or rng.random() < 0.8)
                         else f + rng.uniform(0.5, 2) / scale)
        p = I if rng.random() < 0.5 else Q
    if rng.random() < 0.3:
        p = p * wf.cos(rng.uniform(0.5, 4) / scale, rng.uniform(0, 6))
    return rng.uniform(0.1, 2.0) * (p >> (scale * rng.uniform(-8, 8)))


def random_channel(wf, rng):
    scale = 10.0**rng.uniform(-9, 0)
    n = int(rng.integers(1, 7))
    pulses = [random_pulse(wf, rng, scale) for _ in range(n)]
    if rng.random() < 0.4:
        ch = wf.WaveVStack(pulses)
        if rng.random() < 0.5:
            ch = (ch + rng.uniform(-0.5, 0.5)) >> (scale * rng.uniform(-1, 1))
    else:
        ch = pulses[0]
        for p in pulses[1:]:
            ch = ch + p
        if rng.random() < 0.2:
            ch = wf.cut(ch, min=-0.4, max=0.6)
    npts = int(rng.integers(1, 60000))
    a = scale * rng.uniform(-14, -6)
    b = a + scale * rng.uniform(2, 30)
    grid = (('linspace', a, b, npts, bool(rng.random() < 0.5)) if rng.random() < 0.7
            else ('arange', a, b, (b - a) / npts))
    return ch, grid


def random_awg_channel(wf, rng):
    """A pulse train on an AWG-rate grid (1-5 GS/s np.arange, the regime of Waveform.sample): 3-60 random
    pulses of 8-60 ns (random_pulse: every fusable shape, erf edges, DRAG primitives, mollifiers, ...)
    placed one after the other with random gaps, as a sum, a WaveVStack, clipped or complex."""
    rate = float(rng.choice([1e9, 1.2e9, 2e9, 2.4e9, 3.2e9, 5e9]))
    scale = float(rng.uniform(4e-9, 12e-9))
    n = int(rng.integers(3, 60))
    pulses, t = [], 0.0
    for _ in range(n):
        p = random_pulse(wf, rng, scale)
        t += scale * float(rng.uniform(2.0, 14.0))
        pulses.append(p >> t)
    mode = rng.random()
    if mode < 0.3:
        ch = wf.WaveVStack(pulses)
        if rng.random() < 0.5:
            ch = (ch + rng.uniform(-0.5, 0.5)) >> (scale * rng.uniform(-1, 1))
    else:
        ch = pulses[0]
        for k, p in enumerate(pulses[1:]):
            ch = ch + (p * (1j if (mode > 0.9 and k % 3 == 0) else 1))
        if 0.3 <= mode < 0.45:
            ch = wf.cut(ch, min=-0.4, max=0.6)
    a = -scale * float(rng.uniform(0, 10))
    b = t + scale * float(rng.uniform(0, 12))
    return ch, ('arange', a, b, 1.0 / rate)


FUZZ_GOLD = 120      # seeds evaluated by the real reference (oracle/make_golden.py -> fuzz.npz)


def fuzz_golden_case(wf, seed):
    """Random script `seed` on a reduced grid (<= 1500 points: keeps the fixture small)."""
    rng = np.random.default_rng(10_000 + seed)
    ch, grid = random_channel(wf, rng)
    if grid[0] == 'linspace':
        grid = ('linspace', grid[1], grid[2], min(grid[3], 1500), grid[4])
    else:
        n = max(1, min(1500, int(np.ceil((grid[2] - grid[1]) / grid[3]))))
        grid = ('arange', grid[1], grid[2], (grid[2] - grid[1]) / n)
    return ch, grid


def far_from_origin_case(wf, seed):
    """Random script and grid moved 1e2..1e7 grid spans away from t = 0 (tools/fuzz_soak_big.py
    'offset'): the regime in which NumPy's grid rounding is visible through fast carriers."""
    rng = np.random.default_rng(77_000 + seed)
    ch, grid = random_channel(wf, rng)
    nch = int(rng.integers(1, 4))
    chans = [ch] + [random_channel(wf, rng)[0] for _ in range(nch - 1)]
    rng.integers(100000, 3000000)
    off = (grid[2] - grid[1]) * 10.0**rng.uniform(2, 7) * (1 if rng.random() < 0.5 else -1)
    chans = [c >> off for c in chans]
    npts = int(rng.integers(1000, 300000))
    return chans, ('linspace', grid[1] + off, grid[2] + off, npts, bool(rng.random() < 0.5))


FAR_GOLD = 40        # far-from-origin scripts evaluated by the real reference (fuzz.npz: far<seed>.<ch>)


def far_golden_case(wf, seed):
    chans, grid = far_from_origin_case(wf, seed)
    return chans, (grid[0], grid[1], grid[2], min(grid[3], 1500), grid[4])


# ---------------------------------------------------------------------------------------
# Python-callable primitives: function(), registerBaseFunc, function_lib= (reference
# waveform.py:1470-1478, 178/535/679; _waveform.pyx:130-131, 264-271).  Module-level
# callables (picklable).  Each case: name -> builder(ns) -> (waveform, function_lib | None, x)
# ---------------------------------------------------------------------------------------
def uf_tanh(t, k, a):
    return a * np.tanh(k * t)


def uf_const(t):
    return 0.75                      # a scalar: broadcast by the reference's arithmetic


def uf_bump(t, w):
    return 1.0 / (1.0 + (t / w)**2)  # Lorentzian


def uf_triangle(t, w):
    return 2.0 / np.pi * np.arcsin(np.sin(w * t + np.pi / 2))   # triangle wave in place of cos


def uf_table(t, pts):
    return np.interp(t, np.linspace(-40e-9, 40e-9, len(pts)), pts)   # tuple argument


def _base_lib(ns):
    return dict(ns._waveform._baseFunc)


def _user_x(n=4001, a=-120e-9, b=160e-9):
    return np.linspace(a, b, n)


def _u_tanh(ns):
    w = ns.function(uf_tanh, 3e7, 0.8, start=-50e-9, stop=50e-9) * ns.cos(2 * pi * 50e6)
    return w, None, _user_x()


def _u_power_const(ns):
    w = ns.function(uf_tanh, 5e7, 1.0, start=-60e-9, stop=60e-9)**2 + \
        (ns.function(uf_const, start=10e-9, stop=90e-9) >> 20e-9) * 0.5
    return w, None, _user_x()


def _u_shared_factor(ns):
    f = ns.function(uf_bump, 15e-9, start=-70e-9, stop=70e-9) >> 12e-9
    I, Q = ns.mixing(f, freq=80e6, phase=0.3, DRAGScaling=2e-10 * 0)     # f in two terms of one piece
    return I + 0.25 * (ns.gaussian(30e-9) >> 60e-9) * ns.function(uf_tanh, 1e8, 1.0), None, _user_x()


def _u_vstack(ns):
    ws = [ns.function(uf_bump, 10e-9, start=-40e-9, stop=40e-9) * ns.cos(2 * pi * 30e6),
          ns.gaussian(40e-9) >> 30e-9,
          ns.function(uf_table, tuple(np.sin(np.arange(17) * 0.7)), start=-40e-9, stop=40e-9) >> 50e-9]
    w = (ns.WaveVStack(ws) + 0.125) >> 7e-9
    return w, None, _user_x()


def _u_lib_override(ns):
    lib = _base_lib(ns)
    lib[2] = uf_bump                 # GAUSSIAN(t, std_sq2) -> Lorentzian of the same argument
    w = (ns.gaussian(40e-9) >> 10e-9) * ns.cos(2 * pi * 40e6) + 0.2 * ns.sin(2 * pi * 10e6)
    return w, lib, _user_x()


def _u_lib_vstack_attr(ns):
    lib = _base_lib(ns)
    lib[4] = uf_triangle             # COS(t, w) -> triangle wave
    w = ns.WaveVStack([ns.gaussian(50e-9) * ns.cos(2 * pi * 25e6), ns.square(60e-9) >> 70e-9])
    w.function_lib = lib
    return w, None, _user_x()


def _u_lib_remap_builtin(ns):
    lib = _base_lib(ns)
    lib[11] = lib[12]                # COSH evaluated as SINH (a built-in under another id)
    w = ns.cosh(2e7) * (ns.square(100e-9) >> 20e-9)
    return w, lib, _user_x()


def _u_complex_amp(ns):
    w = (1 + 2j) * ns.function(uf_tanh, 4e7, 1.0, start=-30e-9, stop=80e-9) + \
        0.5j * (ns.gaussian(20e-9) >> 100e-9)
    return w, None, _user_x()


def _u_registered_id(ns):
    tid = ns.registerBaseFunc(uf_bump)
    expr = ns._waveform.basic_wave(tid, 8e-9, shift=25e-9)
    w = ns.Waveform(bounds=(round(-20e-9, 15), round(90e-9, 15), np.inf),
                    seq=(ns._waveform._zero, expr, ns._waveform._zero)) * 1.5
    return w, None, _user_x()


def uf_cexp(t, w, tau):
    return np.exp(1j * w * t - (t / tau)**2)          # complex-valued callable


def _u_complex_fn(ns):
    f = ns.function(uf_cexp, 2 * pi * 30e6, 40e-9, start=-60e-9, stop=90e-9)
    w = 0.8 * f * ns.cos(2 * pi * 11e6) + (0.3 - 0.2j) * (f >> 20e-9)**2 + 0.1 * (ns.gaussian(30e-9) >> 100e-9)
    return w, None, _user_x()


def _u_complex_pow(ns):
    # built-in factors raised to complex powers: value**n with NumPy's complex power (_waveform.pyx:143-146)
    g = ns.gaussian(60e-9) >> 20e-9
    w = 0.7 * g**(1.5 + 0.8j) + (ns.square(50e-9) >> 110e-9) * (ns.cos(2 * pi * 20e6)**(2 + 1j)) * 0.25
    return w, None, _user_x()


def _u_clip_complex(ns):
    # np.clip on a complex part: lexicographic against the real bounds (_waveform.pyx:162)
    w = (1 + 2j) * (ns.gaussian(50e-9) >> 10e-9) * ns.cos(2 * pi * 40e6) + (-1.5 + 0.5j) * (ns.gaussian(30e-9) >> 90e-9) \
        + 0.4 * (ns.square(40e-9) >> -70e-9)
    w.min, w.max = -0.5, 0.4
    return w, None, _user_x()


USER_CASES = {
    'u_complex_fn': _u_complex_fn, 'u_complex_pow': _u_complex_pow, 'u_clip_complex': _u_clip_complex,
    'u_tanh': _u_tanh, 'u_power_const': _u_power_const, 'u_shared_factor': _u_shared_factor,
    'u_vstack': _u_vstack, 'u_lib_override': _u_lib_override,
    'u_lib_vstack_attr': _u_lib_vstack_attr, 'u_lib_remap_builtin': _u_lib_remap_builtin,
    'u_complex_amp': _u_complex_amp, 'u_registered_id': _u_registered_id,
}


def user_sample_case(ns):
    """Waveform.sample(function_lib=...) on its own arange grid, plain and chunked."""
    w = ns.square(200e-9, edge=40e-9, type='erf') * ns.cos(2 * pi * 15e6) + \
        ns.function(uf_bump, 30e-9, start=-150e-9, stop=150e-9)
    lib = _base_lib(ns)              # (after function(): an explicit library REPLACES the registry)
    lib[3] = uf_tanh_edge            # ERF(t, std_sq2) -> tanh edge
    w.start, w.stop, w.sample_rate = -300e-9, 300e-9, 2e9
    return w, lib


def uf_tanh_edge(t, s):
    return np.tanh(t / s)


# ---- oversampled grids: what the late round-2 fusions evaluate (erf edges as closing ops,
# exponential envelopes).  Evaluated by the real reference into tests/golden/late.npz
# (oracle/make_golden.py late): strided subset + the samples around every piece edge + sums.
def _late_tones(ns, seed, nt, lo=-300e6, hi=300e6, cplx=False):
    rng = np.random.default_rng(seed)
    out = None
    for _ in range(nt):
        t = rng.uniform(0.05, 0.3) * ns.cos(2 * pi * rng.uniform(lo, hi), rng.uniform(0, 6))
        if cplx:
            t = t * complex(rng.uniform(-1, 1), rng.uniform(-1, 1))
        out = t if out is None else out + t
    return out


def _late_flat_tops(ns):
    w = ns.zero()
    for k in range(4):
        w = w + 0.7 * (ns.square(30e-9, edge=4e-9) >> (20e-9 + k * 50e-9))
    return w


def _late_readout(ns, nt, seed, cplx=False):
    w = ns.zero()
    tones = _late_tones(ns, seed, nt, cplx=cplx)
    for k in range(4):
        w = w + ((ns.square(30e-9, edge=4e-9) >> ((k + 0.5) * 55e-9 + 1.3e-9)) * tones)
    return w


def _late_mixing(ns):
    env = ns.square(40e-9, edge=6e-9) >> 60e-9
    I, Q = ns.mixing(env, freq=137e6, phase=0.4, DRAGScaling=3e-10)
    return I + 0.3 * Q


def _late_flat_gauss_stack(ns):
    a = (ns.square(30e-9, edge=5e-9) * ns.gaussian(60e-9) * _late_tones(ns, 3, 2)) >> 50e-9
    b = (ns.gaussian(20e-9) >> 36e-9) * ns.cos(2 * pi * 80e6)
    return ns.WaveVStack([a, b]) + 0.05


def _late_cosh(ns):
    p = ns.coshPulse(60e-9, eps=3.0, plateau=25e-9) >> 110e-9
    I, _ = ns.mixing(p, freq=140e6, phase=0.7, DRAGScaling=2e-10)
    return I + 0.5 * (ns.coshPulse(40e-9, eps=1.0) >> 40e-9)


def _late_exp(ns):
    decay = (ns.square(80e-9) >> 50e-9) * (ns.exp(-1 / 30e-9) >> 10e-9)
    g = (ns.gaussian(40e-9) >> 150e-9) * (ns.exp(2e7) >> 150e-9) * ns.cos(2 * pi * 210e6)
    s = (ns.square(40e-9) >> 190e-9) * (ns.sinh(3e7) >> 190e-9) * (ns.exp(-1e7) >> 180e-9)
    return decay - 2 * g + s


def _late_far(ns):
    t0 = 1.0e-3
    w = ns.zero()
    tones = _late_tones(ns, 21, 3, 250e6, 350e6)
    for k in range(3):
        w = w + ((ns.square(30e-9, edge=4e-9) >> (t0 + (k + 0.5) * 60e-9)) * tones)
    return w


LATE_GRID = ('linspace', 0.0, 220e-9, 800_003, False)
LATE_CASES = {
    'flat_tops': (_late_flat_tops, LATE_GRID),
    'readout3': (lambda ns: _late_readout(ns, 3, 8), LATE_GRID),
    'readout10': (lambda ns: _late_readout(ns, 10, 15), LATE_GRID),
    'readout_cplx': (lambda ns: _late_readout(ns, 3, 9, cplx=True), LATE_GRID),
    'mixing_flat_drag': (_late_mixing, LATE_GRID),
    'flat_gauss_stack': (_late_flat_gauss_stack, LATE_GRID),
    'cosh': (_late_cosh, LATE_GRID),
    'exp': (_late_exp, LATE_GRID),
    'far_flat': (_late_far, ('linspace', 1.0e-3, 1.0e-3 + 180e-9, 600_000, False)),
}


# ---- AWG-rate grids (round 3): what Waveform.sample() is called with (1-5 GS/s, pulses of tens of
# samples) -- the short-piece tier (wfk_short.hip).  Reference vectors: tests/golden/awg.npz.
def _awg_readme(ns, rate):
    """The README sequence stretched over 10 us: cosPulse + mixing(DRAGScaling): five-term pieces."""
    pulse = ns.cosPulse(20e-9)
    x = ns.zero()
    for k in range(200):
        I, Q = ns.mixing((0.3 + 0.003 * k) * pulse >> (25e-9 + 50e-9 * k), freq=-20e6 + 1e5 * k,
                         phase=0.01 * k, DRAGScaling=0.2)
        x = x + (I if k % 2 else Q)
    return x


def _awg_mixed(ns, rate):
    """Gaussian pulses of several widths, plateaus (constant pieces), an exponential decay, a cubic
    polynomial under a Gaussian, gaps between them."""
    w = ns.zero()
    t = 0.0
    rng = np.random.default_rng(77)
    for k in range(150):
        width = float(rng.choice([12e-9, 20e-9, 32e-9, 41e-9]))
        kind = k % 5
        t += 0.75 * width + float(rng.choice([0.0, 0.0, 7e-9, 31e-9]))
        if kind == 0:
            p = ns.gaussian(width) * ns.cos(2 * pi * float(rng.uniform(20e6, 300e6)))
        elif kind == 1:
            I, Q = ns.mixing(ns.gaussian(width), freq=float(rng.uniform(-250e6, 250e6)),
                             phase=float(rng.uniform(0, 6)), DRAGScaling=float(rng.uniform(1e-10, 4e-10)))
            p = I - 0.5 * Q
        elif kind == 2:
            p = 0.4 * ns.square(1.5 * width) * ns.cos(2 * pi * float(rng.uniform(20e6, 120e6)))
        elif kind == 3:
            p = ns.square(1.5 * width) * (ns.exp(-1 / (0.7 * width)) >> (-0.75 * width))
        else:
            p = ns.gaussian(width) * ns.poly([0.2, 3e7, -2e15, 5e22])
        w = w + float(rng.uniform(0.2, 1.0)) * (p >> t)
        t += 0.75 * width
    return w


def _awg_cplx(ns, rate):
    w = ns.zero()
    for k in range(120):
        I, Q = ns.mixing(0.6 * ns.gaussian(24e-9) >> (20e-9 + 40e-9 * k), freq=35e6 + 3e6 * k, phase=0.1 * k,
                         DRAGScaling=2e-10)
        w = w + (I + 1j * Q)
    return w


def _awg_vstack(ns, rate):
    ws = []
    for k in range(90):
        I, _ = ns.mixing((0.2 + 0.005 * k) * ns.gaussian(30e-9) >> (30e-9 + 45e-9 * k), freq=-80e6 + 2e6 * k,
                         phase=0.3 * k, DRAGScaling=1.5e-10)
        ws.append(I)
    return (ns.WaveVStack(ws) >> 3.3e-9) + 0.125


def _awg_clip(ns, rate):
    w = ns.zero()
    for k in range(150):
        w = w + ((0.1 + 0.01 * k) * ns.gaussian(20e-9) * ns.cos(2 * pi * 150e6) >> (16e-9 + 31e-9 * k))
    w.min, w.max = -0.45, 0.7
    return w


def _awg_chan(c, duty30=False):
    def build(ns, rate):
        from waveforms_amd import workloads as wl
        return wl.awg_channel(ns, c, 100000 if rate == 2e9 else 20000, rate, duty30)
    return build


def _awg_grid(n, rate):
    return ('arange', 0.0, n / rate, 1.0 / rate)


def awg_c4_subset(n):
    """sample indices of the awg_c4.npz vectors: every 3rd + 33 around every multiple of 1024 + both ends"""
    idx = set(range(0, n, 3)) | set(range(min(n, 80))) | set(range(max(0, n - 80), n))
    for m in range(0, n, 1024):
        idx |= set(range(max(0, m - 16), min(n, m + 17)))
    return np.array(sorted(idx), dtype=np.int64)


AWG_CASES = {   # name -> (build(ns, rate), rate, n)
    'b2b_2g': (_awg_chan(0), 2e9, 100000),
    'duty30_2g': (_awg_chan(1, True), 2e9, 100000),
    'b2b_1g': (_awg_chan(2), 1e9, 20000),
    'b2b_2p4g': (_awg_chan(3), 2.4e9, 20000),
    'b2b_5g': (_awg_chan(4), 5e9, 20000),
    'readme_1g': (_awg_readme, 1e9, 10000),
    'readme_2p4g': (_awg_readme, 2.4e9, 24000),
    'mixed_2g': (_awg_mixed, 2e9, 20000),
    'cplx_2g': (_awg_cplx, 2e9, 10000),
    'vstack_2g': (_awg_vstack, 2e9, 8400),
    'clip_2g': (_awg_clip, 2e9, 10000),
}


# ---------------------------------------------------------------------------------------------------
# N4 on the device: the text front-end, simplify, filter, marker / mask / | / &, interp
# and the CLI -- trees that come out of the symbolic layer, SAMPLED.  oracle/make_golden.py builds the twin
# of every case with the real reference's Python API and stores what it samples to (tests/golden/n4.npz);
# tests/test_gpu_frontend.py runs the product's own front-end through the HIP path against those vectors.
# ---------------------------------------------------------------------------------------------------
_PG = ('linspace', -130.0, 40.0, 6801, True)

PARSER_CASES = {   # name -> (text for wave_eval, twin(ns) built through the Python API, grid)
    # reference tests/test_waveform.py:141-166 (test_parser)
    'p_one': ("one()", lambda ns: ns.one(), LIN),
    'p_zero': ("zero()", lambda ns: ns.zero(), LIN),
    'p_pi': ("pi", lambda ns: ns.const(pi), LIN),
    'p_e': ("e", lambda ns: ns.const(np.e), LIN),
    'p_ref_w2': ("(gaussian(10) << 100) + square(20, edge=5, type='linear') * cos(2*pi*23.1)",
                 lambda ns: (ns.gaussian(10) << 100) + ns.square(20, edge=5, type='linear') * ns.cos(2 * pi * 23.1), _PG),
    'p_ref_w3': ("((gaussian(10) << 50) + ((square(20, 5, type='linear') * cos(2*pi*23.1)) >> 50)) << 50",
                 lambda ns: (ns.gaussian(10) << 100) + ns.square(20, edge=5, type='linear') * ns.cos(2 * pi * 23.1), _PG),
    'p_ref_w4': ("(gaussian(10) << 100) + square(20, 5, 'linear') * cos(2*pi*23.1)",
                 lambda ns: (ns.gaussian(10) << 100) + ns.square(20, edge=5, type='linear') * ns.cos(2 * pi * 23.1), _PG),
    'p_ref_poly': ("poly([1, -1/2, 1/6, -1/12])", lambda ns: ns.poly([1, -1 / 2, 1 / 6, -1 / 12]), LIN),
    'p_ref_poly_tuple': ("poly((1, -1/2, 1/6, -1/12))", lambda ns: ns.poly([1, -1 / 2, 1 / 6, -1 / 12]), LIN),
    # the rest of the grammar: precedence quirks, keyword arguments, complex numbers, every constructor family
    'p_unary_loosest': ("-cos(2) + gaussian(3)", lambda ns: -(ns.cos(2) + ns.gaussian(3)), LIN),
    'p_pow_left': ("(cos(1.5) + 2) ** 2 ** 2", lambda ns: ((ns.cos(1.5) + 2)**2)**2, LIN),
    'p_caret': ("gaussian(6) ^ 3 / 4", lambda ns: ns.gaussian(6)**3 / 4, LIN),
    'p_shift_loosest': ("gaussian(4) * cos(3) >> 1 + 1.5", lambda ns: (ns.gaussian(4) * ns.cos(3)) >> 2.5, LIN),
    'p_complex': ("(1 + 2j) * cos(2, 0.3) * gaussian(5) + 0.5j * sin(1)",
                  lambda ns: (1 + 2j) * ns.cos(2, 0.3) * ns.gaussian(5) + 0.5j * ns.sin(1), LIN),
    'p_kwargs': ("gaussian(width=4, plateau=2) * 0.7 + cosPulse(5, plateau=1.5) >> 2",
                 lambda ns: (ns.gaussian(4, plateau=2) * 0.7 + ns.cosPulse(5, plateau=1.5)) >> 2, LIN),
    'p_drag': ("drag(0.4, 12.0, delta=0.05, block_freq=0.9, phase=0.3, t0=-7.0)",
               lambda ns: ns.drag(0.4, 12.0, delta=0.05, block_freq=0.9, phase=0.3, t0=-7.0), LIN),
    'p_chirp': ("chirp(1, 2, 10, 4, 'exponential') * square(9) >> 4.5",
                lambda ns: (ns.chirp(1, 2, 10, 4, 'exponential') * ns.square(9)) >> 4.5, LIN),
    'p_interp': ("interp([-4, -1, 0, 2.5, 6], [0, 1, -0.5, 2, 0]) * cos(3)",
                 lambda ns: ns.interp([-4, -1, 0, 2.5, 6], [0, 1, -0.5, 2, 0]) * ns.cos(3), LIN),
    'p_sampling_points': ("samplingPoints(-4.0, 6.0, [0.0, 1.0, -0.5, 2.0, 0.25, -1.0, 0.0])",
                          lambda ns: ns.samplingPoints(-4.0, 6.0, [0.0, 1.0, -0.5, 2.0, 0.25, -1.0, 0.0]), LIN),
    'p_mollifier_sinc': ("mollifier(12.0, d=1) + 0.1 * sinc(0.8) * square(18)",
                         lambda ns: ns.mollifier(12.0, d=1) + 0.1 * ns.sinc(0.8) * ns.square(18), LIN),
    'p_cut': ("cut(gaussian(6) * cos(4), start=-2, stop=3, max=0.5)",
              lambda ns: ns.cut(ns.gaussian(6) * ns.cos(4), start=-2, stop=3, max=0.5), LIN),
    'p_deriv': ("D(gaussian(8) * cos(3), 2) / 10", lambda ns: ns.D(ns.gaussian(8) * ns.cos(3), 2) / 10, LIN),
    'p_cosh': ("0.01 * cosh(0.4) - 0.02 * sinh(0.35) + coshPulse(9, eps=2.5)",
               lambda ns: 0.01 * ns.cosh(0.4) - 0.02 * ns.sinh(0.35) + ns.coshPulse(9, eps=2.5), LIN),
    'p_exp_trig': ("2 * (exp(1.01 + 22j) ** 2 << 1) * exp(1.01 + 22j)",
                   lambda ns: 2 * (ns.exp(1.01 + 22j)**2 << 1) * ns.exp(1.01 + 22j), ('linspace', -2.0, 2.0, 1001, True)),
    'p_general_cosine': ("general_cosine(16.0, 1.0, 0.3, 0.1) + hanning(7) >> 1",
                         lambda ns: (ns.general_cosine(16.0, 1.0, 0.3, 0.1) + ns.hanning(7)) >> 1, LIN),
    'p_mdrag': ("drag_sin(0.4, 14.0, 0, 0.013, (0.9, 0.55, -0.7), 0.3, -7.0)",
                lambda ns: ns.drag_sin(0.4, 14.0, 0, 0.013, (0.9, 0.55, -0.7), 0.3, -7.0), LIN),
    'p_pulse_train': ("(gaussian(20e-9) >> 30e-9) * cos(2*pi*150e6, 0.4) + (square(40e-9, edge=5e-9) >> 100e-9) * cos(2*pi*80e6)",
                      lambda ns: (ns.gaussian(20e-9) >> 30e-9) * ns.cos(2 * pi * 150e6, 0.4)
                      + (ns.square(40e-9, edge=5e-9) >> 100e-9) * ns.cos(2 * pi * 80e6),
                      ('linspace', 0.0, 160e-9, 4001, True)),
}


def _tones(ns, freqs, amps=None, env=None):
    w = ns.zero()
    for k, f in enumerate(freqs):
        a = 1.0 if amps is None else amps[k]
        w = w + a * ns.cos(f, 0.1 * k)
    return w if env is None else w * env


FILTER_CASES = {   # name -> (build(ns), low, high, grid): w.filter(low, high), reference _waveform.pyx:638-654
    'f_lowpass': (lambda ns: _tones(ns, [2.0, 9.0, 30.0]) + 1, 0, 5, LIN),
    'f_highpass': (lambda ns: _tones(ns, [2.0, 9.0, 30.0]) + 1, 5, np.inf, LIN),
    'f_band': (lambda ns: _tones(ns, [2.0, 9.0, 30.0], [0.5, 1.5, -0.7]), 5, 20, LIN),
    'f_band_env': (lambda ns: _tones(ns, [2.0, 9.0, 30.0], [0.5, 1.5, -0.7], ns.gaussian(8)) + ns.gaussian(3), 5, 20, LIN),
    'f_pieces': (lambda ns: (ns.square(6) << 4) * _tones(ns, [1.0, 12.0]) + (ns.square(6) >> 4) * _tones(ns, [3.0, 25.0]), 2, 20, LIN),
    'f_products': (lambda ns: ns.cos(3) * ns.cos(11) * ns.gaussian(9) + ns.sin(1.5) * ns.sin(2.5), 6, 12, LIN),   # products reduce to sums first
    'f_complex': (lambda ns: (1 + 0.5j) * ns.cos(7) + 2j * ns.cos(15, 0.3) + 0.25, 0, 10, LIN),
    'f_nothing_left': (lambda ns: _tones(ns, [2.0, 9.0]), 100, 200, LIN),
    'f_mixing': (lambda ns: ns.mixing(ns.gaussian(6), freq=8.0, phase=0.2, DRAGScaling=0.01)[0] + ns.cos(1.0) * ns.square(12), 4, 16, LIN),
}

INTERP_CASES = {   # name -> (build(ns), grid): interp() -- piecewise LINEAR segments (reference waveform.py:1425-1440)
    'i_plain': (lambda ns: ns.interp([0.0, 1.0, 3.0, 4.5], [0.0, 2.0, -1.0, 0.5]), LIN),
    'i_repeated_x': (lambda ns: ns.interp([-5.0, -2.0, -2.0, 1.0, 6.0], [1.0, 1.0, -1.0, 0.0, 3.0]), LIN),
    'i_many': (lambda ns: ns.interp(np.linspace(-9, 9, 73), np.sin(np.linspace(-9, 9, 73))**3), LIN),
    'i_mod': (lambda ns: (ns.interp([-6.0, -1.0, 2.0, 7.0], [0.0, 1.0, 1.0, 0.0]) * ns.cos(5, 0.7)) >> 0.5, LIN),
    'i_awg': (lambda ns: ns.interp(np.arange(0, 41) * 5e-9, np.hanning(41)) * ns.cos(2 * pi * 120e6),
              ('arange', -10e-9, 230e-9, 0.5e-9)),
}

CLI_CASES = {   # name -> (argv of `python -m waveforms_amd sample` minus EXPR OUT, EXPR, twin(ns), start, stop, rate, amplitude)
    # ('>>' binds loosest of the binary operators: "a * b + c >> s" shifts the whole sum)
    'cli_awg': (['-S', '2e9', '-b', '1e-6'], "(gaussian(100e-9) >> 300e-9) * cos(2*pi*50e6) + square(200e-9, edge=20e-9) >> 600e-9",
                lambda ns: ((ns.gaussian(100e-9) >> 300e-9) * ns.cos(2 * pi * 50e6) + ns.square(200e-9, edge=20e-9)) >> 600e-9,
                0, 1e-6, 2e9, 1),
    'cli_defaults': ([], "cos(2*pi*440) * gaussian(0.5) >> 0.5", lambda ns: (ns.cos(2 * pi * 440) * ns.gaussian(0.5)) >> 0.5, 0, 1, 44100, 1),
    'cli_duration': (['-S', '1000', '-a', '2', '-l', '5', '-A', '3'], "sin(7) * square(3) >> 4.5",
                     lambda ns: (ns.sin(7) * ns.square(3)) >> 4.5, 2, 7, 1000, 3),
}

def out_nonfinite_cases():
    """wav(x, out=buf) with NaN / inf sitting in `buf`: the reference zeroes with `out *= 0` (waveform.py:551),
    which keeps them.  name -> (build(ns), x, buf)"""
    x = np.linspace(-10, 10, 2001)
    real = np.linspace(3, 4, 2001)
    real[[3, 700, 1999]] = [np.nan, np.inf, -np.inf]
    longer = np.concatenate([real, [1.0, np.nan, 2.0, -np.inf]])           # len(out) > len(x)
    cplx = (np.linspace(0, 1, 2001) + 1j * np.linspace(1, 2, 2001))
    cplx[5] = np.nan
    cplx[900] = complex(np.inf, 1.0)
    cplx[1500] = complex(2.0, -np.inf)
    return {
        'real': (lambda ns: ns.gaussian(6) * ns.cos(3) + 0.25, x, real),
        'longer': (lambda ns: ns.gaussian(6) * ns.cos(3) + 0.25, x, longer),
        'cplx': (lambda ns: (1 + 2j) * ns.gaussian(6) * ns.cos(3), x, cplx),
        'finite': (lambda ns: ns.gaussian(6) * ns.cos(3) + 0.25, x, np.linspace(3, 4, 2001)),
    }


def n4_logic_names():
    """plain-Waveform cases of CASES that tests/golden/logic.json holds marker / mask / | / & lists for"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'logic.json')) as f:
        gold = json.load(f)
    return sorted(n for n, v in gold.items() if 'error' not in v)
