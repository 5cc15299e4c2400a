"""-m gpu: a few fuzz cases and the one-shot pipeline once more in guard mode (VT_DEBUG_GUARD=1, a child process: the switch is read
when the library makes its first allocation).  Every device buffer then sits between two fields of NaN that are checked when
the buffer is released: an out-of-bounds read shows up as NaN in a result, an out-of-bounds write aborts with a message."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_selected_cases_in_guard_mode():
    env = dict(os.environ, VT_DEBUG_GUARD='1')
    nodes = ['tests/test_gpu_fuzz.py::test_random_cases_match_oracle[%d]' % s for s in (0, 3, 7, 11)]
    nodes += ['tests/test_gpu_parity.py::test_oneshot_pipeline_ragged_chunks', 'tests/test_gpu_parity.py::test_marching_staging_modes']
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-p', 'no:cacheprovider'] + nodes, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=850)
    tail = r.stdout.decode(errors='replace')[-3000:]
    assert r.returncode == 0, tail
    assert 'vt guard' not in tail
