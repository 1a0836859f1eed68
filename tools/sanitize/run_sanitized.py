#!/usr/bin/env python3
"""ASan + UBSan run of the host-side native code (CPU only).

    python tools/sanitize/run_sanitized.py [--scripts N]

Builds  waveforms_amd/csrc/wfk_compile.cpp  (the 700-line index/table compiler that walks
caller-supplied offset arrays) and  oracle/wfk_oracle.c  with
`-fsanitize=address,undefined -fno-sanitize-recover=undefined` into tools/sanitize/_build/,
then re-runs itself under LD_PRELOAD=libasan and drives both with
  * N random scripts (tests/cases.random_channel: every primitive, vstacks, clips, grids over
    nine decades) in grid AND tlist mode,
  * the far-from-origin scripts, the edge grids (n = 0, 1, 2, endpoint on/off) and
  * malformed programs (broken offsets / bounds / arg counts / ids / NaN clip), which must be
    REJECTED with an error code, not walked.
Exit code 0 and no sanitizer report = clean.  GPU sanitizers are not available on this pool;
the device code is covered by the parity tests instead."""
import argparse
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD = os.path.join(HERE, '_build')
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g', '-O1']


def build():
    os.makedirs(BUILD, exist_ok=True)
    inc = ['-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'waveforms_amd', 'csrc')]
    subprocess.run(['g++', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', *SAN, *inc,
                    os.path.join(ROOT, 'waveforms_amd', 'csrc', 'wfk_compile.cpp'),
                    os.path.join(HERE, 'wfk_san_shim.cpp'),
                    '-o', os.path.join(BUILD, 'libwfk_compile_san.so')], check=True)
    subprocess.run(['gcc', '-std=c11', '-D_GNU_SOURCE', '-fPIC', '-shared', '-ffp-contract=off', *SAN,
                    *inc, os.path.join(ROOT, 'oracle', 'wfk_oracle.c'),
                    '-o', os.path.join(BUILD, 'libwfk_oracle_san.so'), '-lm'], check=True)


def preload():
    libs = [subprocess.run(['gcc', '-print-file-name=' + n], capture_output=True, text=True,
                           check=True).stdout.strip() for n in ('libasan.so', 'libubsan.so')]
    return ':'.join(libs)


def drive(n_scripts):
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
    import cases
    import waveforms_amd as wf
    from waveforms_amd import _flatten
    comp = C.CDLL(os.path.join(BUILD, 'libwfk_compile_san.so'))
    orc = C.CDLL(os.path.join(BUILD, 'libwfk_oracle_san.so'))
    err = C.create_string_buffer(512)

    def compile_(prog, grid=None, t=None):
        d = C.c_double(0)
        if t is not None:
            t = np.ascontiguousarray(t, dtype=np.float64)
        return comp.wfk_san_compile(C.byref(prog.struct), C.byref(grid) if grid is not None else None,
                                    t.ctypes.data_as(C.c_void_p) if t is not None else None,
                                    C.c_int64(len(t) if t is not None else 0), C.byref(d), err, 512)

    def oracle_grid(prog, grid):
        n = int(grid.n)
        re = np.empty((prog.n_channels, max(n, 1)))
        im = np.empty_like(re)
        return orc.wfk_oracle_eval_grid(C.byref(prog.struct), C.byref(grid), re.ctypes.data_as(C.c_void_p),
                                        im.ctypes.data_as(C.c_void_p), C.c_int64(re.shape[1]))

    def oracle_tlist(prog, t):
        t = np.ascontiguousarray(t, dtype=np.float64)
        re = np.empty((prog.n_channels, max(len(t), 1)))
        return orc.wfk_oracle_eval_tlist(C.byref(prog.struct), t.ctypes.data_as(C.c_void_p),
                                         C.c_int64(len(t)), re.ctypes.data_as(C.c_void_p), None,
                                         C.c_int64(re.shape[1]))

    done = 0
    rng = np.random.default_rng(2026)
    for i in range(n_scripts):
        ch, gd = cases.random_channel(wf, rng)
        prog = _flatten.flatten([ch])
        grid = _flatten.grid_from_desc(gd)
        rc = compile_(prog, grid=grid)
        assert rc in (0, -2), (i, rc, err.value)
        if int(grid.n) <= 20000:
            assert oracle_grid(prog, grid) == 0
        # tlist mode on a (jittered, sorted) copy of the grid
        n = min(int(grid.n), 4000)
        t = np.sort(np.linspace(grid.t0, grid.t0 + grid.step * max(n - 1, 0), n) +
                    rng.normal(size=n) * grid.step * 0.3) if n else np.zeros(0)
        rc = compile_(prog, t=t)
        assert rc in (0, -2), (i, rc, err.value)
        assert oracle_tlist(prog, t) == 0
        done += 1
    for seed in range(min(40, n_scripts)):
        chans, gd = cases.far_from_origin_case(wf, seed)
        prog = _flatten.flatten(chans)
        assert compile_(prog, grid=_flatten.grid_from_desc(gd)) in (0, -2)
        done += 1
    # oversampled grids: erf edges as closing multiplier ops (twins, several tones, overlapping edges),
    # exponential envelopes (EXP / COSH / SINH, alone and under Gaussians), multi-tone pieces at the
    # lean kernel's op limit -- the host paths added late in round 2
    def tones(nt):
        out = None
        for _ in range(nt):
            t = rng.uniform(0.05, 0.3) * wf.cos(2 * np.pi * rng.uniform(-3e8, 3e8), rng.uniform(0, 6))
            out = t if out is None else out + t
        return out
    W = 40e-9
    fine = [wf.square(W, edge=5e-9) >> 60e-9,
            (wf.square(W, edge=5e-9) >> 60e-9) * tones(10),
            (wf.square(W, edge=5e-9) >> 60e-9) * tones(14) * (0.3 - 0.4j),
            (wf.square(6e-9, edge=5e-9) >> 60e-9) * tones(3),
            wf.mixing(wf.square(W, edge=6e-9) >> 60e-9, freq=1.3e8, phase=0.2, DRAGScaling=2e-10)[0],
            (wf.square(W, edge=5e-9) * wf.gaussian(2 * W) * tones(2)) >> 60e-9,
            wf.coshPulse(W, eps=3.0, plateau=10e-9) >> 60e-9,
            (wf.square(W) >> 60e-9) * (wf.exp(-3e7) >> 20e-9) * tones(2),
            (wf.gaussian(W) >> 60e-9) * (wf.exp(4e7) >> 60e-9) * (wf.sinh(1e7) >> 50e-9),
            (wf.square(W) >> 60e-9) * wf.exp(1e11),
            (wf.gaussian(W) >> 60e-9) * tones(16)]
    for gd in (('linspace', 0.0, 120e-9, 400001, False), ('linspace', 1e-3, 1e-3 + 120e-9, 300000, True),
               ('linspace', 0.0, 120e-9, 2001, False)):
        g = _flatten.grid_from_desc(gd)
        for k in range(0, len(fine), 3):
            prog = _flatten.flatten([c >> gd[1] for c in fine[k:k + 3]])
            assert compile_(prog, grid=g) == 0, err.value
            done += 1
    # AWG-rate grids: the short-piece tier's unit / slot / record tables (tests/cases.py AWG_CASES and
    # random pulse trains at 1-5 GS/s, ragged lengths)
    for name, (build, rate, n) in cases.AWG_CASES.items():
        for nn in (n, n // 3 + 1, 17):
            assert compile_(_flatten.flatten([build(wf, rate)]), grid=_flatten.grid_from_desc(cases._awg_grid(nn, rate))) == 0, (name, err.value)
            done += 1
    from waveforms_amd import workloads as wl
    for seed in range(12):
        rate = float(rng.choice([1e9, 2e9, 2.4e9, 3.2e9, 5e9]))
        chans = [wl.awg_channel(wf, 100 + seed * 3 + c, 9000, rate, duty30=bool(c % 2)) for c in range(3)]
        g = _flatten.grid_arange(float(rng.uniform(-50e-9, 50e-9)), 9000 / rate, 1 / rate)
        assert compile_(_flatten.flatten(chans), grid=g) == 0, err.value
        done += 1
    # multi-channel programs (chunk tables across channels) and the edge grids
    chans = [cases.random_channel(wf, rng)[0] for _ in range(9)]
    prog = _flatten.flatten(chans)
    for gd in (('linspace', 0.0, 1e-6, 0, True), ('linspace', 0.0, 1e-6, 1, True),
               ('linspace', 0.0, 1e-6, 2, False), ('linspace', -3e-6, 5e-6, 70001, True),
               ('arange', -1e-6, 2e-6, 1e-9), ('arange', 0.0, 0.0, 1e-9)):
        g = _flatten.grid_from_desc(gd)
        assert compile_(prog, grid=g) == 0, err.value
        assert oracle_grid(prog, g) == 0
        done += 1
    assert compile_(prog, t=np.zeros(0)) == 0
    # malformed programs: must be rejected (negative code), never walked out of bounds
    good_grid = _flatten.grid_linspace(0.0, 1.0, 100)

    def broken(edit, want_any_error=True):
        p = _flatten.flatten([wf.gaussian(0.5) * wf.cos(3.0) >> 0.5, wf.square(0.3) * 2 + wf.sinc(4.0)])
        edit(p)
        rc = compile_(p, grid=good_grid)
        assert rc < 0, ('accepted a malformed program', rc)

    def e1(p): p.arrays['pc_term_off'][1] = 99
    def e2(p): p.arrays['pc_bound'][-1] = 1.0
    def e3(p): p.arrays['pc_bound'][0], p.arrays['pc_bound'][1] = 5.0, -5.0
    def e4(p): p.arrays['fc_arg_off'][-1] += 1
    def e5(p): p.arrays['fc_type'][0] = 99
    def e6(p): p.arrays['ch_clip_lo'][0] = float('nan')
    def e7(p): p.arrays['tm_factor_off'][-1] = 10**6
    def e8(p): p.arrays['ch_member_off'][1] = 7
    def e9(p): p.arrays['mb_piece_off'][1] = 0
    def e10(p): p.arrays['fc_arg_off'][1] = -3
    def e11(p): p.struct.n_terms = -1
    def e12(p): p.struct.n_pool = 10**7
    for e in (e1, e2, e3, e4, e5, e6, e7, e8, e9, e10, e11, e12):
        broken(e)
        done += 1
    assert compile_(_flatten.flatten([wf.gaussian(0.5)]), grid=_flatten.wfk_grid(0.0, -1.0, 10, 0, 0.0)) < 0
    # wfk_grid_detect walks a caller array (threaded above 2^20 elements, bisection on a probe set)
    comp.wfk_grid_detect.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_flatten.wfk_grid)]
    g = _flatten.wfk_grid()
    for t in (np.linspace(-1e-6, 9e-6, 10001), np.linspace(0, 3e-6, 1_300_003, endpoint=False),
              np.arange(-1e-6, 2e-6, 1e-9), np.linspace(-7.3e-3, -7.3e-3 + 1e-5, 70001, endpoint=False),
              np.sort(rng.uniform(0, 1, 5000)), np.zeros(17), np.linspace(0, 1, 16), np.linspace(0, 1, 15)):
        t = np.ascontiguousarray(t)
        rc = comp.wfk_grid_detect(t.ctypes.data, len(t), C.byref(g))
        assert rc in (0, 1)
        if rc == 1:
            assert np.array_equal(_flatten.grid_values(g), t)
        bad_t = t.copy()
        if len(bad_t) > 20:
            bad_t[len(bad_t) // 2] = np.nextafter(bad_t[len(bad_t) // 2], np.inf)
            assert comp.wfk_grid_detect(bad_t.ctypes.data, len(bad_t), C.byref(g)) == 0 or not np.all(np.diff(t) > 0)
        done += 1
    print(f'sanitized run clean: {done} programs through wfk_compile (ASan+UBSan), oracle C alongside')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scripts', type=int, default=300)
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--canary', action='store_true', help='(child) overflow a heap buffer on purpose')
    a = ap.parse_args()
    if a.child and a.canary:
        C.CDLL(os.path.join(BUILD, 'libwfk_compile_san.so')).wfk_san_canary(4)
        return 0
    if a.child:
        drive(a.scripts)
        return 0
    build()
    env = dict(os.environ, LD_PRELOAD=preload(),
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=97',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1:exitcode=98')
    # the harness must be live: the canary overflow has to be caught (UBSan's object-size check
    # fires first, exit code 98; ASan's heap-buffer-overflow, 97, if that check is compiled out)
    c = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--canary'], env=env,
                       capture_output=True, text=True)
    if c.returncode not in (97, 98) or not ('heap-buffer-overflow' in c.stderr or 'runtime error' in c.stderr):
        print('sanitizer canary was NOT caught: the harness is not live', c.returncode, file=sys.stderr)
        return 96
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--scripts', str(a.scripts)],
                       env=env)
    return r.returncode


if __name__ == '__main__':
    sys.exit(main())
