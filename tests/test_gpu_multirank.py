"""-m gpu: two, three and four ranks sharing the one GPU of the test box, over gloo (RCCL needs a GPU per rank).  Every rank runs
the real HIP slab path -- halo exchange at construction, vt_volume_create_slab, marching kernels, fused projection and its
all-reduce -- and checks its planes against the oracle on the whole volume (tools/slab2_check.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


# (at most 6 processes may have the card open on the test box: this process, the launcher and 4 ranks; 8 ranks run on CPU over gloo in
# tests/test_distributed.py)
@pytest.mark.parametrize('world', [2, 3, 4])
def test_slab_ranks_share_one_gpu(world):
    env = dict(os.environ, OMP_NUM_THREADS='4', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'tools', 'slab2_check.py')]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    lines = [ln for ln in res.stdout.splitlines() if 'slab err' in ln]
    assert res.returncode == 0, res.stdout[-3000:]
    assert lines and 'FAIL' not in res.stdout
