"""Two (or more) ranks sharing ONE GPU over gloo: the real HIP slab kernels + the real halo exchange + the projection
all-reduce, checked against the oracle on the whole volume.  (RCCL needs one GPU per rank; this is the closest thing a
single-GPU box can run.)   torchrun --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/slab2_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import voltools_amd as vt
from voltools_amd.distributed import SlabVolume, slab_bounds
from oracle import oracle

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
counts = [40 + 8 * r for r in range(world)]
G, H, W = sum(counts), 72, 80
vol = np.random.RandomState(5).random_sample((G, H, W)).astype(np.float32)
g0, g1 = slab_bounds(counts)[rank]
c = np.divide(np.subtract((G, H, W), 1), 2, dtype=np.float32)
ok = True
for interp, tol in (('linear', 1e-6), ('bspline', 1e-6), ('filt_bspline', 3e-6)):      # 16 warm-up planes: same bar as a whole volume
    sv = SlabVolume(vol[g0:g1], interpolation=interp, device='gpu:0', reach=2)   # the second matrix shifts by 1.25 planes
    for m in (vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(0.0, 1.5, -2.0), center=c),
              vt.utils.transform_matrix(rotation=(0, 45, 0), translation=(1.25, 0, 0), center=c)):
        want = oracle.affine(vol, m, interp)
        got = sv.affine(m)
        err = float(np.abs(got - want[g0:g1]).max())
        proj = sv.projection(m).cpu().numpy()
        perr = float(np.abs(proj - want.astype(np.float64).sum(axis=0)).max())
        good = err <= tol and perr <= tol * G
        ok = ok and good
        print(f'rank {rank}/{world} {interp:13s} planes [{g0},{g1}) window {sv.window} exchanged {sv.exchanged_bytes} B: '
              f'slab err {err:.2e}  projection err {perr:.2e}  kernel {sv.info().last_kernel}  {"ok" if good else "FAIL"}', flush=True)
    sv.close()
t = torch.tensor([0 if ok else 1])
dist.all_reduce(t)
dist.destroy_process_group()
sys.exit(int(t.item()))
