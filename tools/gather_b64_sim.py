"""Bank model of the cubic general-rotation gather with 8-byte reads (ds_read_b64: 2 groups of 32 lanes, 64 banks of 4 B, a lane
takes an aligned bank pair; equal addresses broadcast): LDS cycles per read = max distinct pair addresses on one bank pair.
All 48 reads of a voxel (16 tap rows x 3 pairs) are the same 32 addresses shifted by a constant, so one read per tile position
is the whole story.  Compares lane->voxel shapes of a 32-lane group and LDS row / plane strides (floats, multiples of 4: rows are
whole 16-B vectors) over random rotations; strides are picked per matrix on a few tile positions and scored on others.
CPU only.  python3 tools/gather_b64_sim.py"""
import numpy as np

rs = np.random.RandomState(0)


def rand_rot():
    q = rs.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def block(shape):
    d, h, w = shape
    return np.array([[i, j, k] for i in range(d) for j in range(h) for k in range(w)], dtype=float)


def degree(A, lanes, RS, PS, bases):
    tot = 0
    for base in bases:
        fl = np.floor(base + lanes @ A.T).astype(int)
        pair = (fl[:, 0] * PS + fl[:, 1] * RS + ((fl[:, 2] - 1) & ~1)) // 2
        tot += np.bincount(np.unique(pair) % 32, minlength=32).max()
    return tot / len(bases)


def bases(n, seed):
    r = np.random.RandomState(seed)
    return [np.array([40.0, 40.0, 40.0]) + r.uniform(0, 1, 3) for _ in range(n)]


if __name__ == '__main__':
    rots = [rand_rot() for _ in range(40)]
    fit, held = bases(6, 1), bases(40, 2)
    shapes = [(1, 2, 16), (1, 1, 32), (2, 4, 4), (1, 4, 8), (2, 2, 8), (4, 4, 2), (4, 8, 1), (2, 16, 1), (8, 4, 1), (1, 8, 4), (2, 8, 2), (4, 2, 4), (8, 2, 2), (2, 2, 8)]
    cands = [(RS, pm) for RS in (32, 36, 40, 44) for pm in range(0, 64, 4)]
    Ly = 28
    for shape in shapes:
        lanes = block(shape)
        dense = np.mean([degree(A, lanes, 32, 32 * Ly, held) for A in rots])
        tab = np.array([[degree(A, lanes, RS, RS * Ly + pm, fit) for (RS, pm) in cands] for A in rots])     # (rot, cand) on the fit positions
        g = int(np.argmin(tab.mean(axis=0)))
        fixed = np.mean([degree(A, lanes, cands[g][0], cands[g][0] * Ly + cands[g][1], held) for A in rots])
        per = np.mean([degree(A, lanes, cands[i][0], cands[i][0] * Ly + cands[i][1], held) for A, i in zip(rots, tab.argmin(axis=1))])
        per36 = []
        for A, row in zip(rots, tab):
            idx = [i for i, c in enumerate(cands) if c[0] == 36]
            i = idx[int(np.argmin(row[idx]))]
            per36.append(degree(A, lanes, 36, 36 * Ly + cands[i][1], held))
        print(f'lanes {shape}: RS=32 dense {dense:.2f}   one fixed stride pair {fixed:.2f} (RS {cands[g][0]}, PS%64 {(cands[g][0] * Ly + cands[g][1]) % 64})'
              f'   per-matrix strides {per:.2f}   per-matrix PS with RS=36 {np.mean(per36):.2f}', flush=True)
