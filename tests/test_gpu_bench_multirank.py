"""-m gpu: the multi-GPU measurement entry point (BASELINE config #5) executed end to end on the one-GPU test box.

`python bench.py --gpus N` launches its own ranks (torch.distributed.run) BEFORE the parent touches the GPU; with
BENCH_ONE_GPU=1 every rank uses GPU 0 and the process group is gloo (RCCL wants one GPU per rank), so this runs the same
code path the driver runs on an 8-GPU node -- slab construction, halo exchange, barrier + max-over-ranks timing, the one JSON
line -- except for the transport.  The test process itself never initialises the GPU: bench.py is always a child."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_args, extra_env=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + extra_args
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


COMMON = ['--size', '256', '--interp', 'bspline', '--steps', '5', '--warmup', '2', '--prewarm-ms', '0', '--no-cpu-baseline']


@pytest.mark.parametrize('world,strong', [(2, False), (2, True), (3, False), (4, True)])
def test_bench_self_launch_ranks_share_one_gpu(world, strong):
    args = ['--gpus', str(world)] + COMMON + (['--strong'] if strong else [])
    r = run_bench(args, {'BENCH_ONE_GPU': '1'})
    assert r['n_gpus'] == world and r['steps'] == 5 and r['warmup'] == 2
    assert r['value'] > 0 and r['ms_per_step'] > 0
    assert r['scaling'] == ('strong' if strong else 'weak')
    planes = 256 // world if strong else 256
    assert f'{world} axis-0 slabs of {planes}x256x256' in r['config']['workload']
    assert r['unit'] == 'Mvoxels/s' and r['dtype'] == 'f32' and r['vs_baseline'] is None
    # whole-job aggregate: all ranks' voxels per step over the max-over-ranks time
    vox = planes * 256 * 256 * world
    assert abs(r['value'] - vox / (r['ms_per_step'] * 1e-3) / 1e6) <= 0.02 * r['value']
    # the only communication of the path: the halo planes, once (bspline: 2 planes from each neighbour)
    h = r['halo_exchange']
    assert h['halo_planes'] == 2 and h['halo_ms'] > 0
    assert h['exchanged_bytes'] == 2 * 256 * 256 * 4                    # rank 0 has one neighbour
    assert r['roofline']['bound'] == 'hbm' and 0 < r['roofline']['frac'] < 1


def test_bench_multirank_line_carries_the_cpu_baseline():
    """North star: "throughput at 1/2/4/8 GPUs reported next to the CPU baseline timed on the same box's host cores" -- the N > 1 line
    has `cpu_baseline` too (rank 0 times a bounded sample of its slab's workload; the 15 s scipy pass over the whole volume is N = 1 only)."""
    args = ['--gpus', '2', '--size', '256', '--interp', 'bspline', '--steps', '5', '--warmup', '2', '--prewarm-ms', '0', '--cpu-seconds', '1']
    r = run_bench(args, {'BENCH_ONE_GPU': '1'})
    cb = r['cpu_baseline']
    assert r['n_gpus'] == 2 and cb['value'] > 0 and cb['kind'] == 'port' and cb['cores'] >= 1 and cb['unit'] == 'Mvoxels/s'
    assert 'scipy_1thread' in cb and 'scipy_1thread_same_workload' not in cb


def test_bench_device_generated_slabs():
    """Slabs of >= 768^3 (BASELINE config #5 uses 1024^3 per GPU) are generated on the device and handed to SlabVolume as torch tensors:
    two ranks of 768^3 on the one GPU, weak scaling."""
    r = run_bench(['--gpus', '2', '--size', '768', '--interp', 'bspline', '--steps', '3', '--warmup', '1', '--prewarm-ms', '0', '--no-cpu-baseline'],
                  {'BENCH_ONE_GPU': '1'}, timeout=900)
    assert r['n_gpus'] == 2 and r['scaling'] == 'weak' and r['value'] > 0
    assert '2 axis-0 slabs of 768x768x768' in r['config']['workload']
    assert r['halo_exchange']['exchanged_bytes'] == 2 * 768 * 768 * 4 and r['config']['kernel'] == 8


def test_bench_forced_slab_path_matches_plain_path():
    """One rank through SlabVolume (a 1-rank RCCL group, the N > 1 code path) against the plain StaticVolume path: same kernel,
    same launch geometry; kernel time within 15 % (measured: within 2 %; identical handles of different processes can differ by 5-6 %,
    profiles/r03_placement_probe.txt)."""
    args = ['--gpus', '1', '--size', '512', '--interp', 'bspline', '--steps', '60', '--warmup', '5', '--prewarm-ms', '100',
            '--no-cpu-baseline', '--no-extra-1024']
    plain = run_bench(args)
    slab = run_bench(args, {'BENCH_FORCE_SLAB': '1', 'MASTER_PORT': '29577'})
    assert 'halo_exchange' in slab and slab['halo_exchange']['exchanged_bytes'] == 0
    assert plain['config']['kernel'] == slab['config']['kernel'] and plain['config']['tile'] == slab['config']['tile']
    a, b = plain['roofline']['kernel_ms'], slab['roofline']['kernel_ms']
    print(f'plain {a} ms, forced slab {b} ms')
    assert abs(a - b) <= 0.15 * a
