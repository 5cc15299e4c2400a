"""Headline benchmark: Mvoxels/s of the resident (StaticVolume) transform hot path on MI355X.

Metric (BASELINE.json): "Mvoxels/s + achieved HBM GB/s, 512^3 f32 filt_bspline, 1/2/4/8 GPU".
A step = one StaticVolume.rotate() of a resident, prefiltered 512^3 float32 volume into a device-resident
output buffer (BASELINE config #3; the README's `static_vol_affine_out` column, volume.py:61-91), i.e. one
launch of the cubic transform kernel.  The one-time prefilter is reported separately (volume.py:48-50 runs
it once at construction).  Rotations follow the README sweep `rotate((0, i, 0))` (README.md:25-27) so every
step uses a different angle; the roofline object is quoted on the same launches.

N > 1 (torchrun, one rank per GPU): the global volume is (N*512) x 512 x 512, slab-partitioned along axis 0;
halos are exchanged once at upload over RCCL, steps need no communication (weak scaling).

Output: ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # default: the README sweep, `for i in range(180): rotate((0, i, 0))` (README.md:25-27) -- one full set of angles; the
    # per-angle time varies by +-10 % (footprint shape), so a shorter sweep depends on where it starts
    ap.add_argument('--steps', type=int, default=180)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--interp', default='filt_bspline')
    ap.add_argument('--strong', action='store_true', help='strong scaling: ONE size^3 volume cut into axis-0 slabs over the '
                    'ranks (SURVEY 8d config 5); default is weak scaling, one size^3 slab per rank')
    ap.add_argument('--prewarm-ms', type=float, default=250.0, help='untimed launches for at least this long before the timed steps')
    ap.add_argument('--no-extra-1024', action='store_true', help='skip the 1024^3 trilinear entry of `extra`')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target wall time of the CPU baseline sample')
    return ap.parse_args()


def cpu_baseline(vol, interp, target_s, scipy_same_workload=True):
    """The CPU oracle (a port of the reference GPU-path semantics, OpenMP) timed on a bounded sample of the
    same workload: blocks of output planes of the same sweep over this rank's volume (rotations about its own centre).
    Reported, never the target.  `scipy_same_workload`: also time the reference's real CPU path (scipy, one thread) ONCE
    on the whole headline volume (512^3 filt_bspline: ~15 s)."""
    from oracle import oracle
    import voltools_amd as vt
    threads = oracle.num_threads()
    nd, n = int(vol.shape[0]), int(vol.shape[1])
    src = oracle.prefilter(vol) if interp.startswith('filt') else vol
    kind = {'filt_bspline': 'bspline', 'filt_bspline_simple': 'bspline_simple'}.get(interp, interp)
    centre = np.divide(np.subtract(vol.shape, 1), 2, dtype=np.float32)
    planes = max(1, nd // 2)
    done, dt, k = 0, 0.0, 0
    t_begin = time.perf_counter()
    while dt < target_s and k < 180:
        m64 = np.asarray(vt.utils.transform_matrix(rotation=(0, float(k), 0), rotation_units='deg', rotation_order='rzxz', center=centre),
                         dtype=np.float64)
        t0 = time.perf_counter()
        oracle.affine_ex(src, m64, kind, (planes, n, n), out_plane0=nd // 4)
        dt += time.perf_counter() - t0
        done += planes * n * n
        k += 1
        if time.perf_counter() - t_begin > 3 * target_s:
            break
    res = {'value': round(done / dt / 1e6, 2), 'unit': 'Mvoxels/s', 'cores': threads, 'kind': 'port',
           'sample': f'{k} sweep angles x {planes} of {nd} output planes of the same {nd}x{n}x{n} {interp} transform (prefilter not '
                     f'timed), oracle/vt_oracle.c with {threads} OpenMP threads, {dt:.1f} s of CPU work'}
    # the reference's actual CPU path (scipy, single-threaded) on BASELINE config #1's size: config #1 itself (200^3 'linear',
    # 45 deg rzxz) and the same volume with the headline interpolation
    try:
        small = np.random.RandomState(0).random_sample((200, 200, 200)).astype(np.float32)
        for key, ip in (('scipy_1thread_config1', 'linear'), ('scipy_1thread', interp)):
            t0 = time.perf_counter()
            vt.transform(small, rotation=(0, 45, 0), rotation_units='deg', rotation_order='rzxz', interpolation=ip, device='cpu')
            dt = time.perf_counter() - t0
            res[key] = {'value': round(200 ** 3 / dt / 1e6, 2), 'unit': 'Mvoxels/s', 'cores': 1,
                        'sample': f'200^3 {ip}, 45 deg rzxz, via voltools_amd device="cpu" (scipy.ndimage.affine_transform, the '
                                  f'reference CPU path' + (', BASELINE config #1' if ip == 'linear' else '') + f'), {dt:.2f} s'}
        if scipy_same_workload and vol.size <= 512 ** 3:
            # the SAME workload through the reference's CPU path: one sweep step of the whole volume (spline_filter included, as
            # transforms.py:126-134 asks scipy for it on every call)
            t0 = time.perf_counter()
            vt.transform(vol, rotation=(0, 45, 0), rotation_units='deg', rotation_order='rzxz', interpolation=interp, device='cpu')
            dt = time.perf_counter() - t0
            res['scipy_1thread_same_workload'] = {'value': round(vol.size / dt / 1e6, 2), 'unit': 'Mvoxels/s', 'cores': 1,
                                                  'sample': f'one step (45 deg rzxz) of the whole {nd}x{n}x{n} {interp} volume via device="cpu" '
                                                            f'(scipy.ndimage.affine_transform incl. its spline prefilter), {dt:.1f} s'}
    except Exception as e:  # pragma: no cover
        res['scipy_1thread'] = {'error': str(e)}
    return res


def measured_traffic(kernel_prefix, case):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes of the SAME command
    (profiles/rNN_<case>_summary.json, written by tools/profile_round.sh: separate --pmc passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  The newest round's file wins."""
    import glob
    best = None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_{case}_summary.json')))
    if case == 'bench':
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r01_summary.json'))) + files      # round 1's name for it
    for f in files:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        tot, cnt = 0.0, 0
        extra = {}
        for name, m in d.get('pmc_mean_per_launch', {}).items():
            if name.startswith(kernel_prefix) and 'hbm_traffic_bytes' in m:
                tot += m['hbm_traffic_bytes'] * m.get('launches_sampled', 1)
                cnt += m.get('launches_sampled', 1)
                # the cubic kernels also report LDS activity and the L2 hit rate (same PMC passes)
                if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m:
                    extra['l2_hit_rate'] = round(m['TCC_HIT_sum'] / max(1.0, m['TCC_HIT_sum'] + m['TCC_MISS_sum']), 3)
                if 'SQ_LDS_IDX_ACTIVE' in m:
                    extra['lds_active_cycles'] = round(m['SQ_LDS_IDX_ACTIVE'])
                    extra['lds_bank_conflict_cycles'] = round(m.get('SQ_LDS_BANK_CONFLICT', 0.0))
        if cnt:
            best = {'bytes': tot / cnt, 'source': os.path.basename(f), 'extra': extra}
    return best


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one per GPU, through torch.distributed.run --
    BEFORE anything in this process touches the GPU (the parent never does) -- and exit with their code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    import voltools_amd as vt
    from voltools_amd import _native

    if not torch.cuda.is_available() or _native.device_count() < 1:
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    # rehearsal on a one-GPU box: BENCH_ONE_GPU=1 puts every rank on GPU 0 and uses gloo (RCCL wants a GPU per rank)
    one_gpu = os.environ.get('BENCH_ONE_GPU') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = f'gpu:{local_rank}'
    n = args.size
    interp = args.interp

    # synthetic data: uniform [0,1) float32, seeded per rank
    strong = args.strong and world > 1
    nd = n // world if strong else n               # planes of this rank's slab
    if strong and n % world:
        raise SystemExit('--strong needs size divisible by the number of ranks')
    if n >= 768:
        # config #5's slabs (1024^3 each) are generated on the device: 12 GB of host random numbers per rank would be most of the run
        gen = torch.Generator(device=f'cuda:{local_rank}')
        gen.manual_seed(rank)
        vol = torch.rand((nd, n, n), dtype=torch.float32, device=f'cuda:{local_rank}', generator=gen)
    else:
        vol = np.random.RandomState(rank).random_sample((nd, n, n)).astype(np.float32)
    use_slab = world > 1 or os.environ.get('BENCH_FORCE_SLAB') == '1'
    if use_slab and dist is None:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank))
    if use_slab:
        from voltools_amd.distributed import SlabVolume
        sv = SlabVolume(vol, interpolation=interp, device=dev, group=dist.group.WORLD)
    else:
        sv = vt.StaticVolume(vol, interpolation=interp, device=dev)
    out = vt.empty((nd, n, n), device=dev)
    gshape = (nd * world, n, n)
    centre = np.divide(np.subtract(gshape, 1), 2, dtype=np.float32)
    # The timed steps span the README sweep whatever their number: `--steps 180` is `for i in range(180): rotate((0, i, 0))` itself
    # (README.md:25-27), fewer steps take every (180 / steps)-th angle -- 20 steps are 0, 9, ..., 171 degrees, not 5..24: the per-angle
    # time varies by +-10 % with the footprint's shape, and a short run must not depend on where it starts.  Warm-up launches use the
    # first angles of the same list.
    def sweep_angle(i, count):
        return float(i % 180) if count >= 180 else float(round(i * 180.0 / count) % 180)
    timed = [vt.utils.transform_matrix(rotation=(0, sweep_angle(i, args.steps), 0), rotation_units='deg', rotation_order='rzxz', center=centre)
             for i in range(args.steps)]
    mats = [timed[i % len(timed)] for i in range(args.warmup)] + timed

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        sv.affine(mats[i], output=out)
    # Time-based pre-warm, outside `warmup` and `steps`: a 20-step run of 0.25 ms launches is over before the clocks have
    # ramped (round 1: the driver's 20-step run read 11 % below the 180-step profile of the same kernel).
    prewarm_ms = 0.0
    if args.prewarm_ms > 0:
        t_pw = time.perf_counter()
        k = 0
        while True:
            for _ in range(32):
                sv.affine(mats[k % len(mats)], output=out)
                k += 1
            sv.synchronize()
            prewarm_ms = (time.perf_counter() - t_pw) * 1e3
            if prewarm_ms >= args.prewarm_ms:
                break
    barrier()
    sv.timer_start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sv.affine(mats[args.warmup + i], output=out)
    kernel_ms_total = sv.timer_stop()          # HIP events on the stream the kernels run on
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-angle distribution (after the timed region): the sweep's angles differ by +-10 % (footprint shape).  One event pair
    # brackets `rep` back-to-back launches of the SAME angle, so the event pair's own cost and the launch gap (~17 us together,
    # 8 % of a 0.2 ms launch: round 2's single-launch samples read above the mean of the timed region) are amortised.
    rep = 8
    step_ms = []
    for i in range(min(args.steps, 180)):
        sv.timer_start()
        for _ in range(rep):
            sv.affine(mats[args.warmup + i], output=out)
        step_ms.append(sv.timer_stop() / rep)
    step_ms = np.asarray(step_ms)

    info = sv.info()
    vox_per_step = nd * n * n * world
    value = vox_per_step * args.steps / elapsed / 1e6
    kernel_ms = kernel_ms_total / args.steps
    algo_bytes = 8.0 * nd * n * n                    # 4 B compulsory source read + 4 B store per output voxel (per rank's launch)
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    # (the README sweep has an integer axis-0 offset: on the plane-quad kernel its launches run the one-tap-plane instantiations,
    #  KIND 3 trilinear / KIND 4 cubic on the z-convolved copy; VT_QUAD_ZFIR=0 keeps the four-plane cubic kernel, KIND 1 / 2)
    quad = int(info.last_kernel) == 8
    zfir = quad and os.environ.get('VT_QUAD_ZFIR', '1') != '0'
    kind_code = 3 if (interp == 'linear' and quad) else 0 if interp == 'linear' else 4 if zfir else 1 if interp in ('bspline', 'filt_bspline') else 2
    kname = {1: f'vt::affine_direct<{kind_code}>', 2: f'vt::affine_tiled<{kind_code}', 3: f'vt::affine_tiled_zsep<{kind_code}',
             4: f'vt::affine_march_zsep<{kind_code}', 5: f'vt::affine_march_zpair<{kind_code}', 8: f'vt::affine_march4<{kind_code}'}.get(int(info.last_kernel), 'vt::affine')
    case = {(512, 'filt_bspline'): 'bench', (1024, 'filt_bspline'): 'sweep1024', (512, 'linear'): 'linear512', (1024, 'linear'): 'linear1024'}.get((n, interp))
    traffic = measured_traffic('void ' + kname, case) if (case and world == 1) else None
    result = {
        'metric': f'Mvoxels/s, {n}^3 f32 {interp} StaticVolume transform (resident source, device output)',
        'value': round(value, 1), 'unit': 'Mvoxels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{n}^3 float32 {interp}, StaticVolume resident in HBM, rotate((0,i,0)) rzxz sweep, '
                               f'output= device buffer' + (f', {world} axis-0 slabs of {nd}x{n}x{n}' if world > 1 else ''),
                   'tile': list(info.last_tile), 'lds_bytes': int(info.last_lds_bytes), 'kernel': int(info.last_kernel),
                   'prefilter_ms_once': round(float(info.prefilter_ms), 3), 'prewarm_ms': round(prewarm_ms, 1),
                   # what the handle built lazily and keeps (the z-convolved plane-quad copies of the orientations the sweep uses, the
                   # in-plane transposed copy): GPU time of those builds, once per handle, and everything resident now
                   'zfir_copy_ms_once': round(float(getattr(info, 'copies_ms', 0.0)), 3), 'resident_bytes': int(info.resident_bytes),
                   'angles': 'i' if args.steps >= 180 else f'round(i * 180 / {args.steps})'},
        'roofline': {'bound': 'hbm', 'kernel': kname, 'achieved': round(achieved, 1), 'peak': 8000.0,
                     'unit': 'GB/s', 'frac': round(achieved / 8000.0, 4),
                     'traffic': (round(traffic['bytes']) if traffic else None),
                     'traffic_source': (traffic['source'] if traffic else None),
                     'counters': (traffic['extra'] if traffic else None),
                     'kernel_ms': round(kernel_ms, 4), 'algorithmic_bytes_per_launch': algo_bytes,
                     'per_angle_ms': {'min': round(float(step_ms.min()), 4), 'median': round(float(np.median(step_ms)), 4),
                                      'max': round(float(step_ms.max()), 4), 'mean': round(float(step_ms.mean()), 4),
                                      'method': f'{rep} back-to-back launches of one angle per HIP event pair, after the timed region'}},
    }
    if world > 1:
        # a first multi-GPU run should diagnose itself: every rank's own kernel time and halo time, not only the maximum
        mine = torch.tensor([kernel_ms, float(getattr(sv, 'halo_ms', 0.0))], dtype=torch.float64, device='cuda')
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per = [[float(x) for x in t_.tolist()] for t_ in allr]
        result['per_rank'] = {'kernel_ms': [round(a[0], 4) for a in per], 'halo_ms': [round(a[1], 3) for a in per],
                              'kernel_ms_min': round(min(a[0] for a in per), 4), 'kernel_ms_max': round(max(a[0] for a in per), 4)}
    if use_slab:
        # the path's only communication happened once, at construction (halo planes, point to point)
        hm = torch.tensor([float(sv.halo_ms)], dtype=torch.float64, device='cuda')
        if world > 1:
            dist.all_reduce(hm, op=dist.ReduceOp.MAX)
        result['halo_exchange'] = {'halo_ms': round(float(hm.item()), 3), 'exchanged_bytes': int(sv.exchanged_bytes),
                                   'sent_bytes': int(sv.sent_bytes), 'window_planes': [int(sv.window[0]), int(sv.window[1])],
                                   'halo_planes': int(sv.halo), 'when': 'once, at SlabVolume construction (rank 0\'s bytes; max over ranks ms)',
                                   'backend': 'gloo (BENCH_ONE_GPU rehearsal)' if one_gpu else 'nccl (RCCL)'}

    if rank == 0 and world == 1:
        # extra (not the headline): the linear kernel on the same volume, and the prefilter's own roofline
        extra = {}
        svl = vt.StaticVolume(vol, interpolation='linear', device=dev)
        for i in range(args.steps):                  # one untimed pass: the lazily built resident copies (transposed, plane-quad) exist afterwards
            svl.affine(mats[args.warmup + i], output=out)
        svl.synchronize()
        svl.timer_start()
        for i in range(args.steps):
            svl.affine(mats[args.warmup + i], output=out)
        ms = svl.timer_stop() / args.steps
        extra['linear'] = {'kernel_ms': round(ms, 4), 'Mvoxels_per_s': round(n ** 3 / ms / 1e3, 1),
                           'achieved_GBps': round(algo_bytes / ms / 1e6, 1), 'frac_of_8TBps': round(algo_bytes / ms / 1e6 / 8000.0, 4),
                           'tile': list(svl.info().last_tile)}
        svl.close()
        if n == 512 and not args.no_extra_1024:
            # the north star's other number: trilinear transform of a 1024^3 volume (BASELINE.json north_star: ">= 70 % of peak HBM
            # bandwidth"), same sweep; the volume is generated on the device (torch: plumbing) so this costs ~2 s
            try:
                g = torch.Generator(device=f'cuda:{local_rank}')
                g.manual_seed(1024)
                big = torch.rand((1024, 1024, 1024), dtype=torch.float32, device=f'cuda:{local_rank}', generator=g)
                svb = vt.StaticVolume(big, interpolation='linear', device=dev)
                outb = vt.empty((1024, 1024, 1024), device=dev)
                cb = np.divide(np.subtract((1024, 1024, 1024), 1), 2, dtype=np.float32)
                mb = [vt.utils.transform_matrix(rotation=(0, float(a), 0), rotation_units='deg', rotation_order='rzxz', center=cb)
                      for a in range(0, 180, 6)]
                for m_ in mb:                         # one untimed pass (builds the transposed / plane-quad copies the sweep uses)
                    svb.affine(m_, output=outb)
                svb.synchronize()
                svb.timer_start()
                for m_ in mb:
                    svb.affine(m_, output=outb)
                msb = svb.timer_stop() / len(mb)
                extra['linear_1024'] = {'kernel_ms': round(msb, 4), 'Mvoxels_per_s': round(1024 ** 3 / msb / 1e3, 1),
                                        'achieved_GBps': round(8.0 * 1024 ** 3 / msb / 1e6, 1),
                                        'frac_of_8TBps': round(8.0 * 1024 ** 3 / msb / 1e6 / 8000.0, 4),
                                        'angles': '0..174 step 6 (30 launches)', 'tile': list(svb.info().last_tile), 'kernel': int(svb.info().last_kernel)}
                svb.close()
                # BASELINE config #4: the same volume, `filt_bspline`, same 30 angles (its 180-step sweep at a sixth of the launches)
                svc = vt.StaticVolume(big, interpolation='filt_bspline', device=dev)
                for m_ in mb:
                    svc.affine(m_, output=outb)
                svc.synchronize()
                svc.timer_start()
                for m_ in mb:
                    svc.affine(m_, output=outb)
                msc = svc.timer_stop() / len(mb)
                ic = svc.info()
                extra['filt_bspline_1024'] = {'kernel_ms': round(msc, 4), 'Mvoxels_per_s': round(1024 ** 3 / msc / 1e3, 1),
                                              'achieved_GBps': round(8.0 * 1024 ** 3 / msc / 1e6, 1),
                                              'frac_of_8TBps': round(8.0 * 1024 ** 3 / msc / 1e6 / 8000.0, 4),
                                              'angles': '0..174 step 6 (30 launches)', 'tile': list(ic.last_tile), 'kernel': int(ic.last_kernel),
                                              'prefilter_ms_once': round(float(ic.prefilter_ms), 3), 'resident_bytes': int(ic.resident_bytes),
                                              'copies_ms_once': round(float(getattr(ic, 'copies_ms', 0.0)), 3)}
                svc.close()
                outb.free()
                del big
                torch.cuda.empty_cache()
            except Exception as e:  # pragma: no cover  (e.g. a smaller device)
                extra.setdefault('linear_1024', {'error': str(e)})
                extra.setdefault('filt_bspline_1024', {'error': str(e)})
        if n == 512:
            # the reference's own benchmark protocol (tests/benchmark.py:52-54): random `sxyz` rotations about size/2 on the resident
            # volume, device output -- the general-matrix kernels (lane blocks for cubic, packed footprints for trilinear)
            rs_g = np.random.RandomState(1)
            gm = [vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2))
                  for r in rs_g.uniform(-180, 180, (100, 3))][:40]
            axes = {'protocol': '16 angles (10..160 degrees) about array axis 1 / 2 through the centre, StaticVolume, device output'}
            gen = {'protocol': '40 of the 100 random sxyz rotations of tests/benchmark.py (RandomState(1)), StaticVolume, device output, second pass over the list'}
            for ip_g, handle in ((interp, sv), ('linear', None)):
                h = handle if handle is not None else vt.StaticVolume(vol, interpolation=ip_g, device=dev)
                for m_ in gm:                       # one untimed pass: the lazily built resident copies (axis-permuted ones included) exist afterwards
                    h.affine(m_, output=out)
                h.synchronize()
                h.timer_start()
                for m_ in gm:
                    h.affine(m_, output=out)
                msg = h.timer_stop() / len(gm)
                gen[ip_g] = {'ms': round(msg, 4), 'frac_of_8TBps': round(algo_bytes / msg / 1e6 / 8000.0, 4), 'kernel': int(h.info().last_kernel)}
                # rotations about the other two array axes (sxyz (0, a, 0) / (0, 0, a) about the centre): the plane-quad kernel on the
                # axis-0 <-> 1 exchanged copy, the row kernel on the plain one
                c_ax = np.divide(np.subtract((n, n, n), 1), 2, dtype=np.float32)
                for ax_name, rot_ax in (('axis1', lambda a: (0, a, 0)), ('axis2', lambda a: (0, 0, a))):
                    am = [vt.utils.transform_matrix(rotation=rot_ax(float(a)), rotation_order='sxyz', center=c_ax) for a in range(10, 170, 10)]
                    for m_ in am[:2]:
                        h.affine(m_, output=out)
                    h.synchronize()
                    h.timer_start()
                    for m_ in am:
                        h.affine(m_, output=out)
                    ms_ax = h.timer_stop() / len(am)
                    axes.setdefault(ax_name, {})[ip_g] = {'ms': round(ms_ax, 4), 'frac_of_8TBps': round(algo_bytes / ms_ax / 1e6 / 8000.0, 4),
                                                          'kernel': int(h.info().last_kernel)}
                if handle is None:
                    h.close()
            extra['general_rotations'] = gen
            extra['single_axis_rotations'] = axes
        pf_ms = float(info.prefilter_ms)
        if pf_ms > 0:
            # the first prefilter of a process also pays for loading its kernels; a second resident volume shows the
            # steady-state cost (what every later StaticVolume of this process pays)
            # (three more volumes, the fastest counts: the first one after the 1024^3 extra above gets device buffers fresh from
            # hipMalloc, whose first touch is part of the kernels' time -- 0.80 instead of 0.57 ms)
            pf2 = 1e30
            for _ in range(3):
                sv2 = vt.StaticVolume(vol, interpolation=interp, device=dev)
                pf2 = min(pf2, float(sv2.info().prefilter_ms))
                sv2.close()
            extra['prefilter'] = {'ms_once': round(pf_ms, 3), 'ms_warm': round(pf2, 3),
                                  'achieved_GBps': round(24.0 * n ** 3 / pf2 / 1e6, 1), 'frac_of_8TBps': round(24.0 * n ** 3 / pf2 / 1e6 / 8000.0, 4),
                                  'algorithmic_bytes': 24.0 * n ** 3, 'kernels': 'prefilter_xy<16,10> (X+Y fused) + prefilter_block<16,18> (Z)'}
        result['extra'] = extra
    sv.close()
    out.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not args.no_cpu_baseline:
        # "throughput at 1/2/4/8 GPUs reported next to the CPU baseline timed on the same box's host cores" (north star): rank 0 times a
        # bounded sample of ITS slab's workload -- after the process group is gone, so that no rank sits in a collective while the host
        # works, and at N > 1 on a bounded crop of the slab (at most 256 planes cross PCIe and are prefiltered on the host, whatever the
        # slab's size); the 15 s single-thread scipy pass over the whole headline volume runs at N = 1 only
        if world > 1 and nd > 256:
            host_vol = (vol[:256] if isinstance(vol, np.ndarray) else vol[:256].cpu().numpy())
        else:
            host_vol = vol if isinstance(vol, np.ndarray) else vol.cpu().numpy()
        result['cpu_baseline'] = cpu_baseline(np.ascontiguousarray(host_vol), interp, args.cpu_seconds if world == 1 else min(args.cpu_seconds, 8.0),
                                              scipy_same_workload=(world == 1))
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
