"""Long differential fuzz of the sample() API (not part of the suite): random scripts with random
start/stop/sample_rate, chunk sizes and SOS filters, `Waveform.sample` / `WaveVStack.sample`
(arange grid, chunked linspace grids with carried filter state, both on the device) against the
NumPy restatement of the reference's sample/_sample_iter.  usage: sample_api_soak.py [count]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from cases import FP64_GRID_TOL
from scipy import signal
import cases
import waveforms_amd as wf
from oracle import np_oracle

def chunk_want(w, chunk_size):
    """the reference's _sample_iter grid walk (waveform.py:209-257), no filter"""
    out, start = [], float(w.start)
    length = chunk_size / w.sample_rate
    while start < w.stop:
        if start + length > w.stop:
            length, stop = w.stop - start, float(w.stop)
            size = round((stop - start) * w.sample_rate)
        else:
            stop, size = start + length, chunk_size
        out.append(np_oracle.call(w, np.linspace(start, stop, size, endpoint=False)))
        start = stop
    return np.concatenate(out) if out else np.zeros(0)


count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad, t0, skipped = [], time.time(), 0
for it in range(count):
    rng = np.random.default_rng(31_000 + it)
    ch, grid = cases.random_channel(wf, rng)
    a, b = grid[1], grid[2]
    npts = int(rng.integers(1, 30000))
    ch.start, ch.stop, ch.sample_rate = a, b, npts / (b - a)
    mode = int(rng.integers(0, 4))
    try:
        want_exc = None
        if mode >= 2:
            pass
        if mode == 0:
            got, want = ch.sample(), np_oracle.sample(ch)
        elif mode == 1:
            cs = int(rng.integers(1, npts + 10))
            got = ch.sample(chunk_size=cs)
            got = np.concatenate(list(got)) if not isinstance(got, np.ndarray) else got
            want = chunk_want(ch, cs)
        else:
            sos = signal.butter(int(rng.integers(1, 5)), rng.uniform(0.01, 0.6), output='sos')
            initial = float(rng.choice([0.0, 0.25]))
            ch.filters = (sos, initial)
            cs = None if mode == 2 else int(rng.integers(1, npts + 10))
            try:
                want = np_oracle.sample_filtered(ch, cs)
            except ValueError as ex:          # the reference raises (empty last chunk): so must we
                try:
                    got = ch.sample(chunk_size=cs)
                    got = np.concatenate(list(got)) if not isinstance(got, np.ndarray) else got
                    bad.append((it, mode, 'no ValueError', repr(ex))); print('FAIL', bad[-1], flush=True)
                except ValueError:
                    pass
                continue
            got = ch.sample(chunk_size=cs)
            got = np.concatenate(list(got)) if not isinstance(got, np.ndarray) else got
        if want is None:
            continue
        got, want = np.real(np.asarray(got)), np.real(np.asarray(want))
        if got.shape != want.shape:
            bad.append((it, mode, 'shape', got.shape, want.shape)); print('FAIL', bad[-1], flush=True); continue
        pk = max(1.0, float(np.abs(want).max())) if want.size else 1.0
        e = float(np.max(np.abs(got - want), initial=0.0)) / pk
        if not e <= FP64_GRID_TOL:
            bad.append((it, mode, e)); print('FAIL', bad[-1], flush=True)
    except NotImplementedError:
        skipped += 1
    except Exception as ex:
        bad.append((it, mode, repr(ex))); print('ERROR', bad[-1], flush=True)
    if it % 100 == 99:
        print(f'{it + 1} rounds, {len(bad)} failures, {time.time() - t0:.0f} s', flush=True)
print('done', count, 'rounds;', skipped, 'skipped (NotImplementedError);', len(bad), 'failures', bad[:10])
