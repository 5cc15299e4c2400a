"""LDS floats per output voxel of candidate LDS layouts for general 3-D rotations (the reference's protocol: 100 random `sxyz`
rotations, tests/benchmark.py:52-54), on the CPU.  Layouts:
  bbox     axis-aligned bounding box of the tile's source footprint (affine_tiled / affine_block)
  shear1   rows (z, y) of the bounding box, row start x0(z, y) linear in (z, y), 16-byte granular (round 4 verdict's proposal)
  shear2   triangular shear: y origin linear in z, x origin linear in (z, y) (box = bounding box of S.A.[0,T-1]^3, S unit lower triangular)
  exact    the footprint itself with 16-byte-aligned row spans (what packed row spans approach)
python3 tools/footprint_survey.py [halo]     halo 0 = trilinear (2 taps per axis), 1 = cubic (4 taps)"""
import sys, itertools
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from voltools_amd.utils import transform_matrix

halo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
taps = 2 + 2 * halo
rs = np.random.RandomState(1)
rs.random_sample((8, 8, 8))
n = 512
rots = rs.uniform(-180, 180, (100, 3))
mats = [np.asarray(transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n, n, n), 2)), dtype=np.float64)[:3, :3] for r in rots]

def exact_rows(A, T, align=4):
    """floats of the union over sub-voxel phases ~ one random phase: rows spans of the point set, aligned"""
    d, h, w = np.meshgrid(np.arange(T[0]), np.arange(T[1]), np.arange(T[2]), indexing='ij')
    p = np.stack([d.ravel(), h.ravel(), w.ravel()], 0).astype(np.float64)
    s = A @ p + np.array([[0.37], [0.61], [0.13]])
    i = np.floor(s).astype(np.int64)
    spans = {}
    for dz in range(-halo, taps - halo):
        for dy in range(-halo, taps - halo):
            key = (i[0] + dz) * 100000 + (i[1] + dy)
            lo = i[2] - halo; hi = i[2] + taps - halo - 1
            order = np.argsort(key, kind='stable')
            k = key[order]; l = lo[order]; u = hi[order]
            b = np.flatnonzero(np.r_[True, k[1:] != k[:-1]])
            mn = np.minimum.reduceat(l, b); mx = np.maximum.reduceat(u, b)
            for kk, a_, c_ in zip(k[b], mn, mx):
                if kk in spans:
                    spans[kk] = (min(spans[kk][0], a_), max(spans[kk][1], c_))
                else:
                    spans[kk] = (a_, c_)
    tot = 0
    for a_, c_ in spans.values():
        a2 = (a_ // align) * align
        tot += ((c_ - a2) // align + 1) * align
    return tot, len(spans)

def bbox(A, T):
    ext = np.abs(A) @ (np.array(T) - 1.0)
    L = np.floor(ext) + 1 + taps        # conservative: any phase
    L[2] = np.ceil((L[2] + 3) / 4) * 4
    return float(np.prod(L))

def shear1(A, T):
    # rows = all (z, y) of the bbox; x' = x - a*y - b*z with (a, b) free: the x' extent over the parallelepiped is min over (a, b) of
    # sum_c |A2c - a*A1c - b*A0c| * (T_c - 1): an L1 fit; candidates: zero two of the three terms
    Tm = np.array(T) - 1.0
    ext = np.abs(A) @ Tm
    best = np.inf
    for c0, c1 in itertools.combinations(range(3), 2):
        M = np.array([[A[1, c0], A[0, c0]], [A[1, c1], A[0, c1]]])
        if abs(np.linalg.det(M)) < 1e-9: continue
        ab = np.linalg.solve(M, np.array([A[2, c0], A[2, c1]]))
        r = A[2] - ab[0] * A[1] - ab[1] * A[0]
        e = np.abs(r) @ Tm + (abs(ab[0]) + abs(ab[1])) * (taps - 1)   # the stencil's own y/z extent shifts the row start too
        best = min(best, e)
    best = min(best, ext[2])
    Lx = np.ceil((np.floor(best) + 1 + taps + 3 + 3) / 4) * 4      # +3 alignment of the origin, +3 rounding of the per-row shear to 4
    return float((np.floor(ext[0]) + 1 + taps) * (np.floor(ext[1]) + 1 + taps) * Lx)

def shear2(A, T):
    Tm = np.array(T) - 1.0
    ext0 = np.abs(A[0]) @ Tm
    best = np.inf
    for cy in range(3):                       # y' = y - g z zeroes column cy of row 1
        if abs(A[0, cy]) < 1e-9: continue
        g = A[1, cy] / A[0, cy]
        r1 = A[1] - g * A[0]
        e1 = np.abs(r1) @ Tm + abs(g) * (taps - 1)
        # x' = x - a y' - b z zeroes two columns of row 2
        B = np.stack([A[0], r1], 0)
        for c0, c1 in itertools.combinations(range(3), 2):
            M = np.array([[B[1, c0], B[0, c0]], [B[1, c1], B[0, c1]]])
            if abs(np.linalg.det(M)) < 1e-9: continue
            ab = np.linalg.solve(M, np.array([A[2, c0], A[2, c1]]))
            r2 = A[2] - ab[0] * B[1] - ab[1] * B[0]
            e2 = np.abs(r2) @ Tm + (abs(ab[0]) + abs(ab[1])) * (taps - 1)
            Lx = np.ceil((np.floor(e2) + 1 + taps + 3 + 3) / 4) * 4
            v = (np.floor(ext0) + 1 + taps) * (np.floor(e1) + 2 + taps) * Lx
            best = min(best, v)
    return float(best)

def best_perm(A, fn, T, perms):
    return min(fn(A[list(p)], T) for p in perms)

# source-axis orders with the contiguous axis last; the product keeps the copies [z][y][x], [x][y][z], [z][x][y] (+ their slow-axis swaps as LDS row order is free)
perms_all = list(itertools.permutations(range(3)))
tiles = [(16, 16, 16), (8, 16, 16), (8, 8, 16), (8, 8, 32), (8, 16, 32), (4, 8, 32), (16, 8, 32), (4, 4, 64), (8, 8, 64)]
print(f'halo {halo} ({taps} taps per axis); mean / max LDS floats per output voxel over the 100 rotations, best source-axis order per matrix')
print(f'{"tile":>12} {"voxels":>6} | {"bbox":>11} | {"shear1":>11} | {"shear2":>11} | {"exact":>11} rows')
for T in tiles:
    nv = T[0] * T[1] * T[2]
    res = {k: [] for k in ('bbox', 'shear1', 'shear2', 'exact', 'rows')}
    for A in mats:
        res['bbox'].append(best_perm(A, bbox, T, perms_all) / nv)
        res['shear1'].append(best_perm(A, shear1, T, perms_all) / nv)
        res['shear2'].append(best_perm(A, shear2, T, perms_all) / nv)
        ex = [exact_rows(A[list(p)], T) for p in perms_all[:1]] if len(res['exact']) >= 20 else [exact_rows(A[list(p)], T) for p in perms_all]
        e = min(ex)
        res['exact'].append(e[0] / nv); res['rows'].append(e[1])
    f = lambda k: f'{np.mean(res[k]):5.2f}/{np.max(res[k]):5.2f}'
    print(f'{str(T):>12} {nv:6d} | {f("bbox")} | {f("shear1")} | {f("shear2")} | {f("exact")} {np.mean(res["rows"]):.0f}', flush=True)
