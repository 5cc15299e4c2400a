"""-m gpu: round 1's kernel families as independent implementations.

The product library carries the plane-quad marching kernel, the lane-block / box / packed-footprint kernels and the direct kernel.
The plain and plane-pair marching kernels (4, 5) and the axis-0-separable box kernel (3) of round 1 are compiled only into the test
build (`make LEGACY=1 OUTDIR=../lib_legacy`, built by `__graft_entry__.build()`).  Here the parity tests that force those families
(VT_NO_QUAD, VT_NO_ZPAIR, VT_NO_MARCH) run against that build in a child process (`VT_LIB` points the ctypes shim at it): three more
implementations of the same arithmetic, written before the current kernels, held to the same oracle at the same tolerances."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGACY_LIB = os.path.join(ROOT, 'voltools_amd', 'lib_legacy', 'libvoltools_hip.so')


def test_legacy_kernel_families_against_the_oracle():
    if not os.path.exists(LEGACY_LIB):
        pytest.skip('test build not present (python -c "import __graft_entry__ as g; g.build()")')
    env = dict(os.environ, VT_LIB=LEGACY_LIB)
    select = ('test_default_dispatch_uses_tiled_kernel_on_large_volumes or test_marching_staging_modes or '
              'test_marching_schedule_does_not_change_results or test_golden_reference_margin12 or test_secondary_copy_failure or '
              '(test_tiled_and_direct_match_oracle and (rot_inplane45 or shift_frac or rot_axis1_shift or rot_axis2) and (linear or filt_bspline-))')
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_parity.py'), '-x', '-q', '-m', 'gpu', '-k', select,
           '-p', 'no:cacheprovider']
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = res.stdout[-3000:]
    assert res.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail
    # the child really ran on the test build
    probe = subprocess.run([sys.executable, '-c', 'from voltools_amd import _native; print(int(_native.has_legacy_kernels()))'],
                           cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True, timeout=300)
    assert probe.stdout.strip().endswith('1'), probe.stdout
