"""Synthetic workloads of BASELINE.json's configs C1..C5.

Every builder takes a namespace `ns` exposing the reference's public names
(`cosPulse`, `gaussian`, `mixing`, `zero`, `WaveVStack`, ...).  With
`ns = waveforms_amd` they build this package's trees; the golden-vector script
(oracle/make_golden.py) passes the imported reference instead, so both sides
run the *same* pulse script.
"""
from __future__ import annotations

import numpy as np

W = 20e-9          # gaussian(W): support +-0.75 W
SPAN = 1.5 * W     # contiguous pulses => 100 % duty


def readme_xy(ns):
    """C1: the README 3-pulse cosPulse + mixing sequence (README.md:28-53)."""
    pulse = ns.cosPulse(20e-9)
    x_wav, y_wav = ns.zero(), ns.zero()
    I, Q = ns.mixing(0.5 * pulse, freq=-20e6, DRAGScaling=0.2)
    x_wav += I
    y_wav += Q
    I, Q = ns.mixing(pulse >> 1e-6, freq=-20e6, phase=np.pi / 2,
                     DRAGScaling=0.2)
    x_wav += I
    y_wav += Q
    I, Q = ns.mixing((0.5 * pulse) >> 2e-6, freq=-20e6, DRAGScaling=0.2)
    x_wav += I
    y_wav += Q
    return x_wav, y_wav


def readme_grid():
    return ('linspace', -1e-6, 9e-6, 10001, True)


def pulses(ns, nseg: int, seed: int, spacing: float = SPAN):
    """`nseg` gaussian+DRAG pulses, SSB-mixed; parameters drawn in the order
    A, f, phi from default_rng(seed) (SURVEY.md §8(d) shared pulse builder)."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(nseg):
        A = rng.uniform(0.1, 1)
        f = rng.uniform(-200e6, 200e6)
        phi = rng.uniform(0, 2 * np.pi)
        I, _ = ns.mixing(A * ns.gaussian(W) >> ((k + 0.5) * spacing), freq=f,
                         phase=phi, DRAGScaling=1e-10)
        out.append(I)
    return out


def sum_channel(ns, nseg: int, seed: int, spacing: float = SPAN):
    """One `Waveform` channel: the pulses summed symbolically (C2, C4, C5)."""
    # pairwise tree sum keeps the host-side merge O(P log P) instead of O(P^2)
    ws = pulses(ns, nseg, seed, spacing)
    while len(ws) > 1:
        nxt = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)]
        if len(ws) % 2:
            nxt.append(ws[-1])
        ws = nxt
    return ws[0]


def vstack_channel(ns, nseg: int, seed: int):
    """One `WaveVStack` channel (C3)."""
    return ns.WaveVStack(pulses(ns, nseg, seed))


def c2_channel(ns, duty30: bool = False):
    return sum_channel(ns, 100, 0, 100e-9 if duty30 else SPAN)


def c2_grid(n=10**7, duty30: bool = False):
    span = 100e-9 if duty30 else SPAN
    return ('linspace', 0.0, 100 * span, n, False)


def c2_drag_channel(ns):
    """C2 variant built from the DRAG primitive (type 13, reference waveform.py:1347-1379):
    100 contiguous `drag()` pulses with random carrier, detuning, notch and phase."""
    rng = np.random.default_rng(0)
    width = SPAN
    ws = []
    for k in range(100):
        A = rng.uniform(0.1, 1)
        f = rng.uniform(-200e6, 200e6)
        phi = rng.uniform(0, 2 * np.pi)
        ws.append(A * ns.drag(f, width, delta=rng.uniform(-5e6, 5e6),
                              block_freq=f - 250e6, phase=phi, t0=k * width))
    while len(ws) > 1:
        nxt = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)]
        if len(ws) % 2:
            nxt.append(ws[-1])
        ws = nxt
    return ws[0]


def c3_channels(ns, nch=256):
    return [vstack_channel(ns, 20, 100 + c) for c in range(nch)]


def c3_grid(n=10**6):
    return ('linspace', 0.0, 20 * SPAN, n, False)


def c4_channels(ns, nch=256, nseg=100):
    return [sum_channel(ns, nseg, 1000 + c) for c in range(nch)]


def c4_kernel(K=1024):
    ker = np.random.default_rng(1).normal(size=K)
    ker /= abs(ker).sum()
    return ker


def awg_channel(ns, c, n_pts=100000, rate=2e9, duty30=False, seed0=7000):
    """One channel at an AWG sample rate: `mixing(A * gaussian(20 ns), freq, phase, DRAGScaling)`
    pulses back to back (spacing SPAN = 30 ns, 100 % duty) or 100 ns apart (30 % duty) over the
    n_pts / rate seconds that `awg_grid` samples -- what `Waveform.sample()` is called with
    (reference waveforms/waveform.py:173-207; README.md:50-53 samples at 1 GS/s).  At 2 GS/s a
    pulse is 60 samples."""
    spacing = 100e-9 if duty30 else SPAN
    nseg = int(n_pts / rate / spacing)
    return sum_channel(ns, nseg, seed0 + c, spacing)


def awg_interp_channel(ns, c, n_pts=100000, rate=2e9, nshapes=8, knots=301):
    """One channel of numerically optimised pulses at an AWG sample rate: every pulse is a `samplingPoints`
    envelope (301 knots over 30 ns, one of `nshapes` shapes of the channel's gate set) under its own carrier
    (`mixing(A * env, freq, phase)`), back to back -- what an optimal-control pulse library looks like to
    `Waveform.sample()` (reference waveform.py:173-207, INTERP: _waveform.pyx:309-311)."""
    rng = np.random.default_rng(8000 + c)
    x = np.linspace(0.0, 1.0, knots)
    shapes = []
    for _ in range(nshapes):
        y = np.hanning(knots).copy()
        for h in (2, 3, 5):
            y *= 1.0 + 0.15 * rng.normal() * np.sin(2 * np.pi * h * x + rng.uniform(0, 6))
        shapes.append(tuple(y))
    nseg = int(n_pts / rate / SPAN)
    ws = []
    for k in range(nseg):
        env = ns.samplingPoints(-SPAN / 2, SPAN / 2, shapes[int(rng.integers(nshapes))])
        I, _ = ns.mixing(rng.uniform(0.1, 1) * env >> ((k + 0.5) * SPAN), freq=rng.uniform(-200e6, 200e6),
                         phase=rng.uniform(0, 2 * np.pi))
        ws.append(I)
    return _tree_sum(ws)


def awg_shape_channel(ns, shape, c, n_pts=100000, rate=2e9):
    """One channel of 60-sample pulses of another shape at an AWG sample rate, back to back (bench.py `also.awg_shapes`,
    tools/awg_shapes_bench.py): 'flat_top' (erf edges under a carrier), 'linear_chirp', 'exp_chirp' (device libm: the
    pointwise tier), 'ten_tones' (a Gaussian under ten carriers)."""
    rng = np.random.default_rng(900 + c)
    nseg = int(n_pts / rate / SPAN)

    def pulse():
        if shape == 'flat_top':
            return ns.square(0.6 * SPAN, edge=0.1 * SPAN) * ns.cos(2 * np.pi * rng.uniform(-2e8, 2e8), rng.uniform(0, 6))
        if shape == 'linear_chirp':
            return ns.chirp(rng.uniform(5e7, 1e8), rng.uniform(1.5e8, 3e8), SPAN) * ns.cosPulse(SPAN)
        if shape == 'exp_chirp':
            return ns.chirp(rng.uniform(5e7, 1e8), rng.uniform(1.5e8, 3e8), SPAN, type='exponential') * ns.cosPulse(SPAN)
        if shape == 'ten_tones':
            tones = None
            for _ in range(10):
                tone = rng.uniform(0.05, 0.2) * ns.cos(2 * np.pi * rng.uniform(-3e8, 3e8), rng.uniform(0, 6))
                tones = tone if tones is None else tones + tone
            return ns.gaussian(W) * tones
        raise ValueError(shape)
    return _tree_sum([rng.uniform(0.2, 1) * pulse() >> ((k + 0.5) * SPAN) for k in range(nseg)])


def awg_grid(n_pts=100000, rate=2e9):
    """np.arange(0, n_pts / rate, 1 / rate): the grid of Waveform.sample (waveform.py:190)."""
    return ('arange', 0.0, n_pts / rate, 1.0 / rate)


def make_grid(desc):
    """Materialise a grid descriptor with NumPy (host reference grid)."""
    kind = desc[0]
    if kind == 'linspace':
        _, a, b, n, endpoint = desc
        return np.linspace(a, b, n, endpoint=endpoint)
    if kind == 'arange':
        _, a, b, step = desc
        return np.arange(a, b, step)
    raise ValueError(kind)


# ---- tiers beside the BASELINE configs (bench.py `also.tlist / direct / multitone`) -------------------
def _tree_sum(ws):
    while len(ws) > 1:
        nxt = [ws[i] + ws[i + 1] for i in range(0, len(ws) - 1, 2)]
        if len(ws) % 2:
            nxt.append(ws[-1])
        ws = nxt
    return ws[0]


def jittered_times(n=2 * 10**6, seed=0):
    """A sorted NON-uniform time axis over the C2 span: the C2 grid of n points with every sample moved by
    N(0, 0.3 dt) -- what `Waveform.__call__(x)` gets when x is not np.linspace / np.arange output (measured
    timestamps, a warped axis).  Time-list plans read it from HBM: 16 B/sample algorithmic."""
    g = make_grid(c2_grid(n))
    rng = np.random.default_rng(seed)
    return np.sort(g + rng.normal(size=n) * (g[1] - g[0]) * 0.3)


def multitone_channel(ns, c, ntones=10, nseg=100):
    """Frequency-multiplexed drive: every gaussian(W) pulse carries `ntones` tones (a ntones-qubit bus)."""
    rng = np.random.default_rng(3000 + c)
    ws = []
    for k in range(nseg):
        tones = None
        for _ in range(ntones):
            tone = rng.uniform(0.05, 0.2) * ns.cos(2 * np.pi * rng.uniform(-300e6, 300e6), rng.uniform(0, 6))
            tones = tone if tones is None else tones + tone
        ws.append((ns.gaussian(W) >> ((k + 0.5) * SPAN)) * tones)
    return _tree_sum(ws)


DIRECT_T = 3e-6      # time span of the direct-tier shapes (100 pulses of 30 ns)


def direct_channel(ns, shape, c=0):
    """Primitives without a recurrence form, 100 pulses back to back over DIRECT_T:
    'sinc' (25 overlapping unbounded sinc pulses), 'mollifier', 'interp' (samplingPoints envelopes of
    1000 knots -- numerically optimised pulse shapes -- under a carrier)."""
    rng = np.random.default_rng(4000 + c)
    if shape == 'sinc':
        return _tree_sum([ns.sinc(4 / W) >> ((k + 0.5) * SPAN * 4) for k in range(25)])
    if shape == 'mollifier':
        return _tree_sum([rng.uniform(0.5, 1) * ns.mollifier(W) >> ((k + 0.5) * SPAN) for k in range(100)])
    if shape == 'interp':
        ws = []
        for k in range(100):
            env = ns.samplingPoints(-SPAN / 2, SPAN / 2, np.hanning(1000) * rng.uniform(0.5, 1))
            I, _ = ns.mixing(env >> ((k + 0.5) * SPAN), freq=rng.uniform(-200e6, 200e6), phase=rng.uniform(0, 6))
            ws.append(I)
        return _tree_sum(ws)
    raise ValueError(shape)
