"""Waveform.__call__ options on arbitrary sorted x (GPU box): non-uniform / duplicated / off-support
times, scalar x, frag=True part lists, out= and accumulate=, int / float32 x, against the NumPy
restatement of calc_parts/_fill_parts.  usage: python tools/call_api_soak.py [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from cases import FP64_GRID_TOL
import cases
import waveforms_amd as wf
from oracle import np_oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
bad = []
for it in range(count):
    rng = np.random.default_rng(91_000 + it)
    try:
        ch, grid = cases.random_channel(wf, rng)
        a, b = grid[1], grid[2]
        n = int(rng.integers(0, 5000))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            x = np.sort(rng.uniform(a, b, size=n))
        elif kind == 1:
            x = np.sort(rng.choice(np.linspace(a, b, max(2, n // 3 + 1)), size=n))      # duplicates
        elif kind == 2:
            x = np.sort(rng.uniform(a - (b - a), b + (b - a), size=n))                  # beyond support
        elif kind == 3:
            x = np.sort(np.concatenate([rng.uniform(a, b, size=n // 2),
                                        np.repeat(rng.uniform(a, b), n - n // 2)]))
        else:
            x = np.linspace(a, b, max(n, 1)).astype(np.float32).astype(np.float64)      # float32-exact times
            x = np.sort(x)
        vst = isinstance(ch, wf.WaveVStack)
        want = np_oracle.call(ch, x)
        got = ch(x)
        pk = max(1.0, float(np.abs(want).max())) if want.size else 1.0
        tol = FP64_GRID_TOL * pk
        if got.shape != want.shape or got.dtype != want.dtype or np.max(np.abs(got - want), initial=0.0) > tol:
            bad.append((it, 'call', got.shape, want.shape, got.dtype, want.dtype)); print('FAIL', bad[-1], flush=True); continue
        if len(x):
            i = int(rng.integers(0, len(x)))
            s = ch(float(x[i]))
            if abs(s - want[i]) > tol:
                bad.append((it, 'scalar', s, want[i])); print('FAIL', bad[-1], flush=True)
        if not vst:
            parts = ch(x, frag=True)
            wparts, _ = np_oracle.pieces(ch.bounds, ch.seq, x, ch.min, ch.max)
            ok = len(parts) == len(wparts)
            for (a1, b1, p1), (a2, b2, p2) in zip(parts, wparts):
                ok = ok and a1 == a2 and b1 == b2 and np.ndim(p1) == np.ndim(p2) and \
                    np.max(np.abs(np.asarray(p1) - np.asarray(p2)), initial=0.0) <= tol
            if not ok:
                bad.append((it, 'frag', len(parts), len(wparts))); print('FAIL', bad[-1], flush=True)
            if want.dtype == np.float64:
                out = rng.normal(size=len(x))
                keep = out.copy()
                r = ch(x, out=out, accumulate=True)
                if r is not out or np.max(np.abs(out - (keep + want)), initial=0.0) > tol:
                    bad.append((it, 'accumulate')); print('FAIL', bad[-1], flush=True)
                r = ch(x, out=out)
                if r is not out or np.max(np.abs(out - want), initial=0.0) > tol:
                    bad.append((it, 'out')); print('FAIL', bad[-1], flush=True)
    except NotImplementedError:
        pass
    except Exception as ex:
        bad.append((it, repr(ex))); print('ERROR', bad[-1], flush=True)
print('done', count, 'rounds;', len(bad), 'failures', bad[:8])
