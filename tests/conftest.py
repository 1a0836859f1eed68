import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def has_gpu():
    """True when the HIP library sees a device (this is what every `gpu` test needs)."""
    try:
        from waveforms_amd import _engine
        return _engine.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # a plain `pytest tests` on a box without a GPU skips the device tests instead of failing
    # every one of them with EngineError; `-m gpu` on the GPU box is unaffected
    gpu_items = [it for it in items if it.get_closest_marker('gpu')]
    if gpu_items and not has_gpu():
        skip = pytest.mark.skip(reason='no MI355X / HIP device visible')
        for it in gpu_items:
            it.add_marker(skip)
