"""Host-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only;
SURVEY.md §5).  tools/sanitize/run_sanitized.py builds wfk_compile.cpp and oracle/wfk_oracle.c
with -fsanitize=address,undefined and drives them with random scripts, edge grids and malformed
programs; a deliberate overflow proves the harness is live."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_compiler_and_oracle_clean_under_asan_ubsan():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'sanitize', 'run_sanitized.py'),
                        '--scripts', '80'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    assert 'sanitized run clean' in r.stdout
    assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr
