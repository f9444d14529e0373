"""Independent numpy reader of g2o files + dense edge-wise PGO cost, used to
cross-check the oracle's and the product's readers / Q builders
(format: ref src/DCORA_utils.cpp:179-375)."""
import gzip
import numpy as np


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def quat_R(qx, qy, qz, qw):
    x, y, z, w = qx, qy, qz, qw
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot2(th):
    return np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])


def read_g2o(path):
    """returns dict(d, n, edges=[(i,j,R,t,kappa,tau)], vertices={id: (R,t)})"""
    edges, verts, d = [], {}, 0
    with _open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "VERTEX_SE2":
                d = d or 2
                i = int(tok[1]); x, y, th = map(float, tok[2:5])
                verts[i] = (rot2(th), np.array([x, y]))
            elif tok[0] == "VERTEX_SE3:QUAT":
                d = d or 3
                i = int(tok[1]); v = list(map(float, tok[2:9]))
                verts[i] = (quat_R(*v[3:7]), np.array(v[0:3]))
            elif tok[0] == "EDGE_SE2":
                d = d or 2
                i, j = int(tok[1]), int(tok[2]); v = list(map(float, tok[3:12]))
                I11, I12, I13, I22, I23, I33 = v[3:9]
                tau = 2.0 / np.trace(np.linalg.inv(np.array([[I11, I12], [I12, I22]])))
                edges.append((i, j, rot2(v[2]), np.array(v[0:2]), I33, tau))
            elif tok[0] == "EDGE_SE3:QUAT":
                d = d or 3
                i, j = int(tok[1]), int(tok[2]); v = list(map(float, tok[3:]))
                t = np.array(v[0:3]); R = quat_R(*v[3:7]); I = v[7:28]
                Tc = np.array([[I[0], I[1], I[2]], [I[1], I[6], I[7]], [I[2], I[7], I[11]]])
                Rc = np.array([[I[15], I[16], I[17]], [I[16], I[18], I[19]], [I[17], I[19], I[20]]])
                tau = 3.0 / np.trace(np.linalg.inv(Tc))
                kappa = 3.0 / (2.0 * np.trace(np.linalg.inv(Rc)))
                edges.append((i, j, R, t, kappa, tau))
            else:
                raise ValueError(tok[0])
    n = max(max(e[0], e[1]) for e in edges) + 1
    return dict(d=d, n=n, edges=edges, vertices=verts)


def edgewise_cost(g, X):
    """f(X) = 1/2 sum_e kappa |Y_j - Y_i R|^2 + tau |p_j - p_i - Y_i t|^2 (ref: DCORA_utils.cpp:2095-2101)"""
    d = g["d"]; dh = d + 1; f = 0.0
    for (i, j, R, t, kappa, tau) in g["edges"]:
        Yi, pi = X[:, i * dh:i * dh + d], X[:, i * dh + d]
        Yj, pj = X[:, j * dh:j * dh + d], X[:, j * dh + d]
        f += 0.5 * kappa * np.sum((Yj - Yi @ R) ** 2) + 0.5 * tau * np.sum((pj - pi - Yi @ t) ** 2)
    return f


def dense_Q(g):
    """dense connection Laplacian from per-edge blocks (independent of the incidence-matrix route)"""
    d = g["d"]; dh = d + 1; n = g["n"]
    Q = np.zeros((dh * n, dh * n))
    for (i, j, R, t, kappa, tau) in g["edges"]:
        T = np.eye(dh); T[:d, :d] = R; T[:d, d] = t
        Om = np.diag([kappa] * d + [tau])
        si, sj = slice(i * dh, i * dh + dh), slice(j * dh, j * dh + dh)
        Q[si, si] += T @ Om @ T.T
        Q[sj, sj] += Om
        Q[si, sj] += -T @ Om
        Q[sj, si] += -Om @ T.T
    return Q


def ground_truth_X(g):
    d = g["d"]; dh = d + 1; n = g["n"]
    X = np.zeros((d, dh * n))
    for i in range(n):
        R, t = g["vertices"][i]
        X[:, i * dh:i * dh + d] = R
        X[:, i * dh + d] = t
    return X
