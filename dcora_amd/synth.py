"""Synthetic SE(3) lattice pose graph of SURVEY.md section 8(d) / BASELINE.json config 5: nx x ny x nz lattice visited in
boustrophedon (snake) order for odometry, loop closures between lattice-adjacent non-consecutive poses kept with
probability p; unit spacing, random rotation per pose; translation noise sigma_t (information 1/sigma_t^2 I), rotation
noise sigma_r rad isotropic (information 1/sigma_r^2 I) -- the information matrices of data/smallGrid3D.g2o for the
defaults.  Deterministic for a given seed (numpy PCG64)."""
import numpy as np

from . import Dataset


def _rand_rot(rng, m):
    q = rng.standard_normal((m, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((m, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _exp_so3(w):
    th = np.linalg.norm(w, axis=1)
    k = w / np.maximum(th, 1e-300)[:, None]
    K = np.zeros((w.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, None, None], np.cos(th)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def lattice_se3(nx=50, ny=50, nz=40, p_lc=0.4, sigma_t=0.1, sigma_r=0.2, seed=20250310):
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    # snake order: index -> lattice coordinate
    coords = np.empty((n, 3), dtype=np.int64)
    idx = 0
    order = {}
    for z in range(nz):
        ys = range(ny) if z % 2 == 0 else range(ny - 1, -1, -1)
        for yi, y in enumerate(ys):
            fwd = (yi % 2 == 0) if z % 2 == 0 else (yi % 2 == 0)
            xs = range(nx) if fwd else range(nx - 1, -1, -1)
            for x in xs:
                coords[idx] = (x, y, z)
                idx += 1
    key = coords[:, 0] + nx * (coords[:, 1] + ny * coords[:, 2])
    where = np.empty(n, dtype=np.int64)
    where[key] = np.arange(n)
    Rgt = _rand_rot(rng, n)
    tgt = coords.astype(np.float64)
    # candidate edges: odometry + lattice neighbours (+x, +y, +z)
    src = [np.arange(n - 1)]
    dst = [np.arange(1, n)]
    for axis, lim in ((0, nx), (1, ny), (2, nz)):
        ok = coords[:, axis] + 1 < lim
        a = np.nonzero(ok)[0]
        c2 = coords[a].copy()
        c2[:, axis] += 1
        b = where[c2[:, 0] + nx * (c2[:, 1] + ny * c2[:, 2])]
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        keep = (hi - lo) > 1
        keep &= rng.random(len(a)) < p_lc
        src.append(lo[keep])
        dst.append(hi[keep])
    i = np.concatenate(src)
    j = np.concatenate(dst)
    m = len(i)
    Rij = np.transpose(Rgt[i], (0, 2, 1)) @ Rgt[j]
    tij = np.einsum("mab,mb->ma", np.transpose(Rgt[i], (0, 2, 1)), tgt[j] - tgt[i])
    Rij = Rij @ _exp_so3(sigma_r * rng.standard_normal((m, 3)))
    tij = tij + sigma_t * rng.standard_normal((m, 3))
    ids = np.zeros((m, 4), np.int32)
    ids[:, 1], ids[:, 3] = i, j
    vals = np.zeros((m, 15))
    vals[:, :9] = np.transpose(Rij, (0, 2, 1)).reshape(m, 9)  # column-major R
    vals[:, 9:12] = tij
    tau = 1.0 / sigma_t ** 2            # 3 / trace(inv(100 I)) = 100
    kappa = 1.0 / (2 * sigma_r ** 2)    # 3 / (2 trace(inv(25 I))) = 12.5
    vals[:, 12], vals[:, 13], vals[:, 14] = kappa, tau, 1.0
    return Dataset(3, n, ids, vals)


def lattice_se2(nx=96, ny=96, p_lc=0.5, sigma_t=0.1, sigma_r=0.1, seed=20250310):
    """the planar twin of lattice_se3: an nx x ny lattice of SE(2) poses visited row by row in snake order (odometry), loop
    closures between lattice-adjacent non-consecutive poses kept with probability p_lc; unit spacing, random heading"""
    rng = np.random.default_rng(seed)
    n = nx * ny
    coords = np.empty((n, 2), dtype=np.int64)
    idx = 0
    for y in range(ny):
        xs = range(nx) if y % 2 == 0 else range(nx - 1, -1, -1)
        for x in xs:
            coords[idx] = (x, y)
            idx += 1
    where = np.empty(n, dtype=np.int64)
    where[coords[:, 0] + nx * coords[:, 1]] = np.arange(n)
    th = rng.uniform(-np.pi, np.pi, n)
    rot = lambda a: np.stack([np.stack([np.cos(a), -np.sin(a)], -1), np.stack([np.sin(a), np.cos(a)], -1)], -2)
    Rgt, tgt = rot(th), coords.astype(np.float64)
    src, dst = [np.arange(n - 1)], [np.arange(1, n)]
    for axis, lim in ((0, nx), (1, ny)):
        a = np.nonzero(coords[:, axis] + 1 < lim)[0]
        c2 = coords[a].copy()
        c2[:, axis] += 1
        b = where[c2[:, 0] + nx * c2[:, 1]]
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        keep = ((hi - lo) > 1) & (rng.random(len(a)) < p_lc)
        src.append(lo[keep])
        dst.append(hi[keep])
    i, j = np.concatenate(src), np.concatenate(dst)
    m = len(i)
    Rij = np.transpose(Rgt[i], (0, 2, 1)) @ Rgt[j] @ rot(sigma_r * rng.standard_normal(m))
    tij = np.einsum("mab,mb->ma", np.transpose(Rgt[i], (0, 2, 1)), tgt[j] - tgt[i]) + sigma_t * rng.standard_normal((m, 2))
    ids = np.zeros((m, 4), np.int32)
    ids[:, 1], ids[:, 3] = i, j
    vals = np.zeros((m, 9))
    vals[:, :4] = np.transpose(Rij, (0, 2, 1)).reshape(m, 4)  # column-major R
    vals[:, 4:6] = tij
    vals[:, 6], vals[:, 7], vals[:, 8] = 1.0 / sigma_r ** 2, 1.0 / sigma_t ** 2, 1.0
    return Dataset(2, n, ids, vals)
