"""Per-channel error of far-from-origin carriers vs the C oracle (diagnostic)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import waveforms_amd as wf
from oracle import c_oracle
from waveforms_amd import _engine, _flatten

for t_center, f in [(0.99e-3, -347e6), (4e-3, 300e6), (16e-3, 300e6), (16e-3, -300e6), (64e-3, 300e6)]:
    rng = np.random.default_rng(0)
    A, ph = rng.uniform(0.3, 1), rng.uniform(0, 6.28)
    chans = {
        'I': wf.mixing(A * wf.gaussian(200e-9) >> t_center, freq=f, phase=ph, DRAGScaling=1e-10)[0],
        'I_nodrag': wf.mixing(A * wf.gaussian(200e-9) >> t_center, freq=f, phase=ph)[0],
        'I_nophase': wf.mixing(A * wf.gaussian(200e-9) >> t_center, freq=f)[0],
        'cw_sq': wf.cos(2 * np.pi * f) * (wf.square(1e-6) >> t_center),
        'cw_shifted': (wf.cos(2 * np.pi * f) * wf.square(1e-6)) >> t_center,
    }
    g = _flatten.grid_linspace(t_center - 1e-6, t_center + 1e-6, 40001, False)
    for name, w in chans.items():
        prog = _flatten.flatten([w])
        plan = _engine.Plan(prog, grid=g)
        got = plan.run_host(np.float64)[0]
        ora = c_oracle.eval_grid(prog, g)[0]
        err = np.abs(got - ora)
        k = int(np.argmax(err))
        print('%8.2e %9.2e %-10s %-40s err %.2e at %d  ulp(th)=%.1e' % (
            t_center, f, name, plan.kernel_name(), err.max(), k, np.spacing(2 * np.pi * abs(f) * t_center)))
