"""IIR stages through the C-ABI: Waveform.sample(filters=(sos,
initial)) [scipy.signal.sosfilt semantics] and predistort(filters=...) [lfilter +
lfiltic], against golden outputs of the real reference (tests/golden/iir.npz), the
oracle, and the reference's own test_filters closed form."""
import numpy as np
import pytest
from scipy.signal import butter, lfilter, lfiltic, sosfilt, tf2sos

import cases
from cases import FP32_TOL
import golden_io
import waveforms_amd as wf
from oracle import np_oracle
from waveforms_amd import _engine, distortion

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['default', 'three-launch'])
def _iir_form(request):
    """every test runs with the library's own choice (single pass where the shape allows: one or two
    biquads on rows of >= 8192 samples) and with the three-launch block scan forced"""
    import os
    if request.param == 'three-launch':
        os.environ['WFK_IIR_ONEPASS'] = '0'
    yield
    os.environ.pop('WFK_IIR_ONEPASS', None)
IIR = golden_io.npz('iir.npz')


def _lfilter_longdouble(b, a, x, initial):
    """The lfilter recurrence (direct form II transposed, zi from lfiltic) in np.longdouble."""
    ld = np.longdouble
    zi = lfiltic(b, a, np.full(len(a) - 1, initial), np.full(len(b) - 1, initial))
    m = max(len(a), len(b)) - 1
    bb = np.zeros(m + 1, dtype=ld)
    aa = np.zeros(m + 1, dtype=ld)
    bb[:len(b)] = np.asarray(b, dtype=ld) / ld(a[0])
    aa[:len(a)] = np.asarray(a, dtype=ld) / ld(a[0])
    z = np.asarray(zi, dtype=ld).copy()
    y = np.empty(len(x), dtype=ld)
    for t in range(len(x)):
        xx = ld(x[t])
        yy = bb[0] * xx + z[0]
        for j in range(m - 1):
            z[j] = bb[j + 1] * xx - aa[j + 1] * yy + z[j + 1]
        z[m - 1] = bb[m] * xx - aa[m] * yy
        y[t] = yy
    return y.astype(np.float64)


def _configure(name):
    build, start, stop, rate, order, fc, initial = cases.iir_cases()[name]
    w = build(wf)
    w.start, w.stop, w.sample_rate = start, stop, rate
    b, a = butter(order, fc, 'lowpass', fs=rate)
    w.filters = (tf2sos(b, a), initial)
    return w


@pytest.mark.parametrize('name', sorted(cases.iir_cases()))
def test_sample_with_filters(name):
    w = _configure(name)
    want = IIR[name + '.full']
    got = w.sample()
    assert got.shape == want.shape and got.dtype == want.dtype      # (complex128 for a complex waveform)
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.abs(want).max())
    assert np.max(np.abs(got - np_oracle.sample_filtered(w))) <= 1e-10
    # chunked: state carried chunk to chunk, also written into `out`
    wantc = IIR[name + '.chunked']
    cs = cases.iir_chunk(name)
    out = np.zeros(len(wantc) + cs, dtype=want.dtype)
    gotc = np.concatenate(list(w.sample(chunk_size=cs, out=out)))
    assert gotc.shape == wantc.shape
    assert np.max(np.abs(gotc - wantc)) <= 1e-10 * max(1.0, np.abs(wantc).max())
    assert np.array_equal(out[:len(gotc)], gotc)
    # survives the flat-list round trip (reference test_filters)
    w2 = type(w).fromlist(w.tolist())
    assert np.allclose(w2.sample(), want, atol=1e-10)


def test_reference_test_filters():
    # reference tests/test_waveform.py:169-194 and tests/test_wavevstack.py:113-137
    sample_rate = 1000
    b, a = butter(3, 4.0, 'lowpass', fs=sample_rate)
    zi = lfiltic(b, a, [0])
    t = np.linspace(-1, 1, 2000, endpoint=False)
    wav = wf.step(0)
    wav.sample_rate, wav.start, wav.stop = sample_rate, -1, 1
    wav.filters = (tf2sos(b, a), 0)
    points = lfilter(b, a, np.heaviside(t, 1), zi=zi)[0]
    assert np.allclose(wav.sample(), points)
    assert np.allclose(wf.Waveform.fromtree(wav.totree()).sample(), points)
    vs = wf.WaveVStack([wf.step(0) << 0.5, -wf.step(0)])
    vs.sample_rate, vs.start, vs.stop = sample_rate, -1, 1
    vs.filters = (tf2sos(b, a), 0)
    pts = lfilter(b, a, np.heaviside(t + 0.5, 1) - np.heaviside(t, 1), zi=zi)[0]
    assert np.allclose(vs.sample(), pts, atol=1e-6)
    assert np.allclose(wf.WaveVStack.fromlist(vs.tolist()).sample(), pts, atol=1e-6)


@pytest.mark.parametrize('i', range(len(cases.predistort_cplx_cases())))
def test_predistort_on_complex_inputs(i):
    """scipy's lfilter / fftconvolve take complex signals, kernels and states, so the reference's predistort
    does (distortion.py:298-337); here the real and imaginary parts run as real passes on the device"""
    n, params, initial, k, real_sig, cker, czi = cases.predistort_cplx_cases()[i]
    sig, ker, zi = cases.predistort_cplx_inputs(i)
    filters = None if params is None else [distortion.exp_decay_filter(A, tau, 1e9) for A, tau in params]
    want = IIR[f'pdc{i}.out']
    if filters is None:
        got, zf = distortion.predistort(sig, None, ker=ker), None
    else:
        got, zf = distortion.predistort(sig, filters, ker=ker, initial=initial, zi=zi, return_zf=True)
    assert got.dtype == np.complex128 and got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.abs(want).max())
    if zf is not None:
        wzf = IIR[f'pdc{i}.zf']
        assert np.iscomplexobj(zf) and np.max(np.abs(zf - wzf)) <= 1e-10 * max(1.0, np.abs(wzf).max())
    ora, _ = np_oracle.predistort(sig, filters, ker, initial, zi)
    assert np.max(np.abs(got - ora)) <= 1e-10 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('i', range(len(cases.predistort_cases())))
def test_predistort_filters(i):
    n, params, initial, k = cases.predistort_cases()[i]
    sig, ker = cases.predistort_inputs(i)
    filters = [distortion.exp_decay_filter(A, tau, 1e9) for A, tau in params]
    b, a = distortion.combine_filters(filters)
    assert np.allclose(np.concatenate([b, a]), IIR[f'pd{i}.ba'], rtol=1e-12, atol=0)
    want, wzf = IIR[f'pd{i}.out'], IIR[f'pd{i}.zf']
    got, zf = distortion.predistort(sig, filters, ker=ker, initial=initial, return_zf=True)
    scale = max(1.0, np.abs(want).max())
    assert got.shape == want.shape
    # Clustered poles (0.99995, 0.9989, 0.975 in case 2) make the direct-form filter
    # ill-conditioned: the reference's own double-precision result is only good to
    # `self_err` against an extended-precision evaluation of the same recurrence.  The
    # bound is therefore 1e-9 * scale or a small multiple of that self error.
    exact = _lfilter_longdouble(b, a, sig, initial)
    ref_iir, _ = np_oracle.predistort(sig, filters, None, initial)
    self_err = float(np.max(np.abs(ref_iir - exact)))
    gain = float(np.abs(ker).sum()) if ker is not None else 1.0
    tol = max(1e-9 * scale, 10 * self_err * gain)
    assert np.max(np.abs(got - want)) <= tol, (np.max(np.abs(got - want)), tol)
    assert np.max(np.abs(zf - wzf)) <= max(1e-9, 10 * self_err) * max(1.0, np.abs(wzf).max())
    ora, ozf = np_oracle.predistort(sig, filters, ker, initial)
    assert np.max(np.abs(got - ora)) <= tol
    d = distortion.distort(sig, np.asarray(params).reshape(-1), 1e9, initial)
    assert np.max(np.abs(d - IIR[f'pd{i}.distort'])) <= max(1e-9 * scale, 10 * self_err)
    # and the IIR stage alone stays within a small multiple of the reference's own
    # distance from the extended-precision result
    mine, _ = distortion.predistort(sig, filters, initial=initial, return_zf=True)
    assert np.max(np.abs(mine - exact)) <= max(1e-10 * scale, 10 * self_err)


def test_iir_batch_shapes_and_properties():
    """Raw plan interface: batch of rows, every compiled shape (biquad cascades, single
    sections of order 1..6, a mixed-order cascade on the generic path), fp32 I/O,
    zi/zf, linearity, block-boundary sizes."""
    import torch
    rng = np.random.default_rng(11)
    shapes = []
    for order in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        b, a = butter(order, 0.05 + 0.02 * order)
        shapes.append([(b, a)])
    for order in (2, 4, 6, 8, 10, 12, 16):     # 5+ biquads: consecutive passes of <= 4 each
        sos = butter(order, 0.08, output='sos')
        shapes.append([(r[:3], r[3:]) for r in sos])
    b1, a1 = butter(1, 0.1)
    b3, a3 = butter(3, 0.2)
    shapes.append([(b1, a1), (b3, a3)])          # mixed orders: cut into runs of equal order
    firsts = [(np.array([1.0 + 0.01 * k, -0.9 - 0.01 * k]), np.array([1.0, -0.95 + 0.02 * k])) for k in range(6)]
    shapes.append(firsts)                         # six first-order sections: runs of 4 + 2
    sos2 = butter(4, 0.1, output='sos')
    shapes.append(firsts[:3] + [(r[:3], r[3:]) for r in sos2])   # orders 1,1,1,2,2
    shapes.append([(b3, a3), (b3, a3)])          # two order-3 sections: one pass each
    for n in (1, 63, 2048, 2049, 150_001):
        for sec in shapes:
            batch = 3
            x = rng.normal(size=(batch, n))
            D = sum(max(len(b), len(a)) - 1 for b, a in sec)
            zi = rng.normal(size=(batch, D)) * 0.1
            want = np.empty_like(x)
            wzf = np.empty((batch, D))
            for r in range(batch):
                y, off = x[r], 0
                for b, a in sec:
                    m = max(len(b), len(a)) - 1
                    y, z = lfilter(b, a, y, zi=zi[r, off:off + m])
                    wzf[r, off:off + m] = z
                    off += m
                want[r] = y
            plan = _engine.IirPlan(sec, n, batch, np.float64)
            xd = torch.from_numpy(x).cuda()
            yd = torch.empty_like(xd)
            zid = torch.from_numpy(zi).cuda()
            zfd = torch.empty_like(zid)
            plan.apply(xd.data_ptr(), n, yd.data_ptr(), n, zid.data_ptr(), zfd.data_ptr())
            torch.cuda.synchronize()
            scale = max(1.0, np.abs(want).max())
            assert np.max(np.abs(yd.cpu().numpy() - want)) <= 1e-10 * scale, (n, len(sec))
            assert np.max(np.abs(zfd.cpu().numpy() - wzf)) <= 1e-10 * scale
    # fp32 I/O (state stays double) + `initial` + in-place
    n, sec = 100_000, [(r[:3], r[3:]) for r in butter(4, 0.03, output='sos')]
    x = rng.normal(size=(2, n)) + 0.7
    want = np.stack([sosfilt(butter(4, 0.03, output='sos'), r - 0.7) + 0.7 for r in x])
    plan = _engine.IirPlan(sec, n, 2, np.float32)
    xd = torch.from_numpy(x).float().cuda()
    plan.apply(xd.data_ptr(), n, xd.data_ptr(), n, None, None, 0.7)
    torch.cuda.synchronize()
    assert np.max(np.abs(xd.cpu().numpy().astype(np.float64) - want)) <= 1e-5


def test_long_cascade_fp32_initial_in_place():
    """8 biquads = two passes of four; fp32 I/O, DC `initial`, in place, no zi/zf."""
    import torch
    rng = np.random.default_rng(12)
    n = 70_001
    sos = butter(16, 0.1, output='sos')
    x = rng.normal(size=(2, n)) + 0.3
    want = np.stack([sosfilt(sos, r - 0.3) + 0.3 for r in x])
    plan = _engine.IirPlan([(r[:3], r[3:]) for r in sos], n, 2, np.float32)
    assert plan.state_dim == 16
    xd = torch.from_numpy(x).float().cuda()
    plan.apply(xd.data_ptr(), n, xd.data_ptr(), n, None, None, 0.3)
    torch.cuda.synchronize()
    assert np.max(np.abs(xd.cpu().numpy().astype(np.float64) - want)) <= FP32_TOL


def test_order_zero_and_empty_filter_lists():
    # scipy.signal.lfilter accepts a bare gain and predistort(filters=[]) combines to b = a = [1]
    # (reference distortion.py:298-321): the device runs them as a scale
    rng = np.random.default_rng(3)
    x = rng.normal(size=5000)
    assert np.allclose(distortion.predistort(x, filters=[]), lfilter([1.0], [1.0], x), atol=0, rtol=0)
    got = distortion.predistort(x, filters=[([0.5], [2.0])])
    assert np.max(np.abs(got - lfilter([0.5], [2.0], x))) <= 1e-15
    assert np.array_equal(distortion.distort(x, [], 1e9), x)


def test_long_cascades_split_into_passes():
    # 12 biquads (order-24 Butterworth as SOS) and a mixed-order cascade of total order 21:
    # scipy has no limit on the number of sections; the device runs consecutive passes
    rng = np.random.default_rng(4)
    x = rng.normal(size=30000)
    sos = butter(24, 0.2, output='sos')
    zi = rng.normal(size=(len(sos), 2)) * 1e-3
    n = len(x)
    plan = _engine.IirPlan([(r[:3], r[3:]) for r in sos], n, 1, np.float64)
    dx, dy = _engine.DeviceBuffer(n * 8), _engine.DeviceBuffer(n * 8)
    dzi, dzf = _engine.DeviceBuffer(zi.nbytes), _engine.DeviceBuffer(zi.nbytes)
    dx.upload(x)
    dzi.upload(zi)
    plan.apply(dx.ptr, n, dy.ptr, n, dzi.ptr, dzf.ptr)
    _engine.sync()
    want, zf = sosfilt(sos, x, zi=zi[:, None, :].reshape(len(sos), 2))
    got = dy.download((n, ), np.float64)
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.abs(want).max())
    assert np.max(np.abs(dzf.download(zi.shape, np.float64) - zf)) <= 1e-10 * max(1.0, np.abs(zf).max())
    for b_ in (dx, dy, dzi, dzf):
        b_.close()
    plan.close()
    # mixed orders 5 + 0 + 7 + 2 + 7 = 21 states, through predistort-like sections
    secs = [butter(5, 0.3), ([0.7], [1.0]), butter(7, 0.25), butter(2, 0.4), butter(7, 0.35)]
    got, _ = distortion.iir_host(x, secs)
    want = x
    for b, a in secs:
        want = lfilter(b, a, want)
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('i', range(len(cases.predistort_high_cases())))
def test_predistort_order_17_to_20_against_the_reference(i):
    """The product runs a combined order > 16 as the cascade of the caller's sections (DESIGN: deviations);
    where the reference's direct-form lfilter is still accurate (poles 0.01..0.6) the two agree: the
    deviation is bounded by 1e-9 against reference-generated vectors (iir.npz pdh*), real and complex
    `initial` included.  A plain ndarray as `zi` is refused on this path (it would be misread)."""
    n, params, initial = cases.predistort_high_cases()[i]
    sig = cases.predistort_high_input(i)
    filters = [distortion.exp_decay_filter(A, tau, 1e9) for A, tau in params]
    want = IIR[f'pdh{i}.out']
    got, zf = distortion.predistort(sig, filters, initial=initial, return_zf=True)
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.abs(want).max())
    assert isinstance(zf, distortion.CascadeState) and zf.shape == (len(params), )
    with pytest.raises(ValueError, match='CascadeState'):
        distortion.predistort(sig, filters, zi=np.asarray(zf))
    # two pieces == the whole, through the state type
    y1, z1 = distortion.predistort(sig[:1234], filters, initial=initial, return_zf=True)
    y2 = distortion.predistort(sig[1234:], filters, zi=z1)
    assert np.max(np.abs(np.concatenate([y1, y2]) - want)) <= 1e-9 * max(1.0, np.abs(want).max())


def test_predistort_combined_order_above_16():
    # 20 first-order exp-decay sections: the reference multiplies them into one order-20
    # polynomial pair and calls lfilter; zero initial state -> the cascade is the same system
    rng = np.random.default_rng(5)
    x = np.concatenate([np.zeros(100), np.ones(4000)])
    filters = [distortion.exp_decay_filter(a_, t_, 1e9)
               for a_, t_ in zip(rng.uniform(-0.02, 0.02, 20), rng.uniform(20e-9, 900e-9, 20))]
    got = distortion.predistort(x, filters)
    want = x
    for b, a in filters:
        want = lfilter(b, a, want)
    assert np.max(np.abs(got - want)) <= 1e-9
    # Non-zero initial state and return_zf at this order.  (The reference itself is of no use as a
    # yardstick here: its order-20 direct-form lfilter overflows to 1e125 / NaN in double precision on
    # such sections; the truth is the cascade.)  initial = c means "the line sat at c for ever": the
    # exp-decay sections have unit DC gain, so the answer is cascade(x - c) from rest, + c.
    c = 0.3
    got_i = distortion.predistort(x + c, filters, initial=c)
    assert np.max(np.abs(got_i - (want + c))) <= 1e-9
    # zf handed back as zi continues the signal: two halves == one run (the state is the cascade's:
    # a direct-form state of this order cannot carry the information in doubles)
    xs = rng.normal(size=6000)
    whole = xs
    for b, a in filters:
        whole = lfilter(b, a, whole)
    y1, zf = distortion.predistort(xs[:2500], filters, return_zf=True)
    assert zf.shape == (20, )
    y2, zf2 = distortion.predistort(xs[2500:], filters, zi=zf, return_zf=True)
    assert np.max(np.abs(np.concatenate([y1, y2]) - whole)) <= 1e-9
    # ... also when the pieces are shorter than the order, and with the FIR stage behind it
    ys, z = [], None
    for k in range(0, 60, 7):
        y_, z = distortion.predistort(xs[k:k + 7], filters, zi=z, return_zf=True)
        ys.append(y_)
    assert np.max(np.abs(np.concatenate(ys) - whole[:63])) <= 1e-9
    ker = rng.normal(size=9)
    yk = distortion.predistort(xs, filters, ker=ker, initial=0.0)
    assert np.max(np.abs(yk - np.convolve(whole, ker, 'full')[4:4 + len(xs)])) <= 1e-9


@pytest.mark.parametrize('nsec,n,rows', [(2, 100003, 3), (1, 65536, 2), (2, 8192 * 4, 5), (2, 1_000_001, 7),
                                          (2, 3_000_017, 1), (1, 2_000_003, 2)])   # (few long rows: several look-back windows)
def test_single_pass_chained_scan(nsec, n, rows):
    """One or two biquads on long rows as ONE kernel (chained scan with decoupled look-back: x is
    read once; the default for these shapes).  Against scipy.signal.sosfilt with random initial state,
    final state, DC offset, in place, float32, and the three-launch form of the same plan
    (WFK_IIR_ONEPASS=0)."""
    import os
    rng = np.random.default_rng(n + nsec)
    sos = butter(2 * nsec, 0.07, output='sos')
    secs = [(r[:3], r[3:]) for r in sos]
    x = rng.normal(size=(rows, n))
    zi = rng.normal(size=(rows, nsec, 2)) * 0.1
    want = np.empty_like(x)
    zfw = np.empty_like(zi)
    for r in range(rows):
        want[r], zfw[r] = sosfilt(sos, x[r] - 0.25, zi=zi[r])
    want += 0.25

    def run(dtype, inplace=False):
        plan = _engine.IirPlan(secs, n, rows, dtype)
        es = np.dtype(dtype).itemsize
        dx, dy = _engine.DeviceBuffer(rows * n * es), _engine.DeviceBuffer(rows * n * es)
        dzi, dzf = _engine.DeviceBuffer(zi.nbytes), _engine.DeviceBuffer(zi.nbytes)
        dx.upload(x.astype(dtype))
        dzi.upload(zi)
        for _ in range(2):                      # twice: flags of the first launch must not leak into the second
            if inplace:
                dx.upload(x.astype(dtype))
            plan.apply(dx.ptr, n, dx.ptr if inplace else dy.ptr, n, dzi.ptr, dzf.ptr, 0.25)
            _engine.sync()
        got = (dx if inplace else dy).download((rows, n), dtype)
        zf = dzf.download(zi.shape, np.float64)
        for b in (dx, dy, dzi, dzf):
            b.close()
        plan.close()
        return got, zf

    got, zf = run(np.float64)
    got_ip, _ = run(np.float64, inplace=True)
    g32, _ = run(np.float32)
    pk = max(1.0, np.abs(want).max())
    assert np.max(np.abs(got - want)) <= 1e-11 * pk
    assert np.max(np.abs(zf - zfw)) <= 1e-11 * max(1.0, np.abs(zfw).max())
    assert np.array_equal(got_ip, got)
    assert np.max(np.abs(g32 - want)) <= FP32_TOL * pk
    os.environ['WFK_IIR_ONEPASS'] = '0'        # the three-launch form of the same plan
    try:
        three, zf3 = run(np.float64)
    finally:
        del os.environ['WFK_IIR_ONEPASS']
    assert np.max(np.abs(three - got)) <= 1e-12 * pk and np.max(np.abs(zf3 - zf)) <= 1e-12 * max(1.0, np.abs(zf).max())


@pytest.mark.parametrize('nsec,n,rows', [(1, 70001, 3), (2, 200003, 2), (3, 65536, 5), (4, 1_000_001, 4), (4, 3_000_017, 1)])
def test_first_order_cascades(nsec, n, rows):
    """Cascades of up to four FIRST-order sections -- the exponential corrections of a flux-line
    predistortion (reference distortion.py:298-321: lfilter per (b, a)) -- take the single-pass form
    too (state dimension <= 4).  Against scipy.signal.lfilter applied section by section, with
    initial state, final state, DC offset, float32; both forms through the module fixture."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(31 * nsec + n)
    secs = []
    for k in range(nsec):
        tau = rng.uniform(20, 4000)                       # samples
        amp = rng.uniform(-0.08, 0.08)
        a1 = -np.exp(-1.0 / tau)
        b0 = 1.0 + amp
        b1 = a1 * (1.0 + amp * (1 - np.exp(-1.0 / tau)) * 0 ) - amp * 0 + a1 * 0
        b1 = a1 - amp * (1 + a1) * 0.3
        secs.append((np.array([b0, b1]), np.array([1.0, a1])))
    x = rng.normal(size=(rows, n))
    zi = rng.normal(size=(rows, nsec, 1)) * 0.1
    want = np.empty_like(x)
    zfw = np.empty_like(zi)
    for r in range(rows):
        y = x[r] - 0.25
        for k, (b, a) in enumerate(secs):
            y, zf_k = lfilter(b, a, y, zi=zi[r, k])
            zfw[r, k] = zf_k
        want[r] = y + 0.25

    def run(dtype):
        plan = _engine.IirPlan(secs, n, rows, dtype)
        assert plan.state_dim == nsec
        es = np.dtype(dtype).itemsize
        dx, dy = _engine.DeviceBuffer(rows * n * es), _engine.DeviceBuffer(rows * n * es)
        dzi, dzf = _engine.DeviceBuffer(zi.nbytes), _engine.DeviceBuffer(zi.nbytes)
        dx.upload(x.astype(dtype))
        dzi.upload(zi)
        for _ in range(2):
            plan.apply(dx.ptr, n, dy.ptr, n, dzi.ptr, dzf.ptr, 0.25)
            _engine.sync()
        got = dy.download((rows, n), dtype)
        zf = dzf.download(zi.shape, np.float64)
        for b_ in (dx, dy, dzi, dzf):
            b_.close()
        plan.close()
        return got, zf

    got, zf = run(np.float64)
    pk = max(1.0, np.abs(want).max())
    assert np.max(np.abs(got - want)) <= 1e-11 * pk
    assert np.max(np.abs(zf - zfw)) <= 1e-11 * max(1.0, np.abs(zfw).max())
    g32, _ = run(np.float32)
    assert np.max(np.abs(g32 - want)) <= FP32_TOL * pk


@pytest.mark.parametrize('rows,n,first', [(64, 200_003, False), (130, 150_001, True)])
def test_single_pass_persistent_waves(rows, n, first):
    """From 64 rows on the single pass runs as a bounded set of persistent waves per row walking the
    ticket counter (more chunks per row than waves per row): sosfilt / lfilter per row, in place,
    twice in a row (epochs), against the three-launch form."""
    import os
    from scipy.signal import lfilter
    rng = np.random.default_rng(rows)
    if first:
        secs = [(np.array([1.0 + 0.01 * k, -0.9 - 0.01 * k]), np.array([1.0, -0.95 + 0.02 * k])) for k in range(3)]
    else:
        secs = [(r[:3], r[3:]) for r in butter(4, 0.07, output='sos')]
    x = rng.normal(size=(rows, n))
    want = np.empty_like(x)
    for r in range(rows):
        y = x[r] - 0.1
        for b, a in secs:
            y = lfilter(b, a, y)
        want[r] = y + 0.1

    def run():
        plan = _engine.IirPlan(secs, n, rows, np.float64)
        dx = _engine.DeviceBuffer(rows * n * 8)
        for _ in range(2):
            dx.upload(x)
            plan.apply(dx.ptr, n, dx.ptr, n, None, None, 0.1)     # in place
            _engine.sync()
        got = dx.download((rows, n), np.float64)
        dx.close()
        plan.close()
        return got

    got = run()
    pk = max(1.0, np.abs(want).max())
    assert np.max(np.abs(got - want)) <= 1e-11 * pk
    os.environ['WFK_IIR_ONEPASS'] = '0'
    try:
        assert np.max(np.abs(run() - got)) <= 1e-12 * pk
    finally:
        os.environ.pop('WFK_IIR_ONEPASS', None)


def test_lookback_timeout_is_an_error_not_silent_nans():
    """A look-back that runs out of polls (stalled / preempted predecessor chunk) must reach the caller:
    forced here with WFK_IIR_SPIN=0 (no poll at all).  The launch's outputs hold NaN, wfk_iir_status says
    so, the plan switches to the three-launch form, and the second launch is correct; the Python stages
    (predistort, sample(filters=)) recover by themselves."""
    import os
    if os.environ.get('WFK_IIR_ONEPASS') == '0':
        pytest.skip('three-launch form: nothing to time out')
    rng = np.random.default_rng(3)
    n, rows = 300_000, 3
    sos = butter(4, 0.05, output='sos')
    secs = [(r[:3], r[3:]) for r in sos]
    x = rng.normal(size=(rows, n))
    want = np.stack([sosfilt(sos, r) for r in x])
    plan = _engine.IirPlan(secs, n, rows, np.float64)
    dx, dy = _engine.DeviceBuffer(x.nbytes), _engine.DeviceBuffer(x.nbytes)
    dx.upload(x)
    os.environ['WFK_IIR_SPIN'] = '0'
    try:
        plan.apply(dx.ptr, n, dy.ptr, n)
        assert plan.status() is False                       # reported ...
        assert np.isnan(dy.download((rows, n), np.float64)).any()   # ... and nothing plausible was written
        assert plan.status() is True                        # (the word is cleared by the check)
        plan.apply(dx.ptr, n, dy.ptr, n)                    # three-launch form now
        assert plan.status() is True
        assert np.max(np.abs(dy.download((rows, n), np.float64) - want)) <= 1e-10
        # a caller that never asks: the NEXT apply of a faulted plan fails loudly
        plan2 = _engine.IirPlan(secs, n, rows, np.float64)
        plan2.apply(dx.ptr, n, dy.ptr, n)
        _engine.sync()
        assert plan2.apply(dx.ptr, n, dy.ptr, n) is False   # WFK_ETIMEOUT: refused, nothing launched
        assert b'look-back timed out' in _engine.lib().wfk_last_error()
        assert plan2.apply(dx.ptr, n, dy.ptr, n) is True
        assert plan2.status() is True
        assert np.max(np.abs(dy.download((rows, n), np.float64) - want)) <= 1e-10
        plan2.close()
        # the host-level stages retry by themselves
        y, _ = distortion.iir_host(x[0], secs)
        assert np.max(np.abs(y - want[0])) <= 1e-10
        # ... also when the SECOND row of a complex waveform is the one that meets the first row's fault
        # (its apply is refused with WFK_ETIMEOUT before status() is asked), and on a cut cascade
        # where every part can stall
        import waveforms_amd as wf
        w = (wf.gaussian(2e-6) >> 5e-6) * wf.cos(2 * np.pi * 3e6) * (1 + 0.5j)
        w.start, w.stop, w.sample_rate = 0.0, 1e-4, 2e9
        got = w.sample(filters=(sos, 0.1))
        ref = w.sample()
        want_c = sosfilt(sos, ref.real - 0.1) + 0.1 + 1j * sosfilt(sos, ref.imag)
        assert np.max(np.abs(got - want_c)) <= 1e-10
        sos8 = butter(8, 0.1, output='sos')
        y8, _ = distortion.iir_host(x[0], [(r[:3], r[3:]) for r in sos8])
        assert np.max(np.abs(y8 - sosfilt(sos8, x[0]))) <= 1e-9
    finally:
        del os.environ['WFK_IIR_SPIN']
        plan.close()
        dx.close()
        dy.close()
