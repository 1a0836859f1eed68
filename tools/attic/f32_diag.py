import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import cases, golden_io
import waveforms_amd as wf
from waveforms_amd import _engine, _flatten
S = golden_io.npz('samples.npz')
for name in sys.argv[1:] or ['c2_duty30', 'c2_small']:
    build, grid = cases.CASES[name]
    want = S[name + '.y']
    g = _flatten.grid_from_desc(grid)
    plan = _engine.Plan(_flatten.flatten([build(wf)]), grid=g)
    got = plan.run_host(np.float32)[0].astype(np.float64)
    e = np.abs(got - want)
    k = int(np.argmax(e))
    print(name, grid, plan.kernel_name(np.float32), 'max err %.2e at %d (want %.4f got %.4f) peak %.2f' % (e.max(), k, want[k], got[k], np.abs(want).max()))
    idx = np.nonzero(e > 2e-5)[0]
    print('  bad samples', len(idx), idx[:10], idx[-5:])
