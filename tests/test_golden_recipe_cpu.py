"""The committed fixture recipe stays runnable: `oracle/make_golden.py --check` regenerates every fixture
group from the live reference into a scratch directory and compares it bit for bit with tests/golden/.
Only where the reference exists (the build container); the GPU box has the fixtures, not the reference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir('/root/reference/waveforms'), reason='reference sources not on this machine')
def test_make_golden_check_reproduces_every_fixture():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'make_golden.py'), '--check'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'check: 0 file(s) differ' in r.stdout
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(ROOT, 'oracle', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    written = sorted(f for _fn, files in mg.FIXTURES.values() for f in files)
    assert written == sorted(os.listdir(os.path.join(ROOT, 'tests', 'golden'))), 'a fixture has no generator'
