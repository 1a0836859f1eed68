"""The plain-C consumer of include/wfk.h (tests/c_abi/abi_smoke.c) ON the GPU box: plan -> wfk_plan_run_host ->
sample-by-sample comparison with libm inside the C program (no Python, no torch between the caller and the ABI)."""
import pytest

from test_abi_cpu import run_c_consumer

pytestmark = pytest.mark.gpu


def test_plain_c_consumer_samples_on_the_device(tmp_path):
    out = run_c_consumer(tmp_path)
    assert 'sampled on the device, parity ok' in out, out
