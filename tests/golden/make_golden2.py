"""Generates tests/golden/vectors2.npz: golden vectors of the pieces either side of the hot path (initialisation,
rounding, robust estimation, range-aided data feed).  Expected values come from the CPU oracle (oracle/), each
cross-checked against an independent computation before it is written (the script aborts on any disagreement).
Re-run only when the oracle changes on purpose:  python tests/golden/make_golden2.py
"""
import gzip
import os
import shutil
import sys
import tempfile

import numpy as np
from scipy.stats import chi2

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import common  # noqa: E402
import g2o_np  # noqa: E402
from oracle import orc  # noqa: E402

out = {}

# ---- chordal initialisation + rounding on smallGrid3D ----
name = "smallGrid3D"
ds = common.oracle_dataset(name)
g = g2o_np.read_g2o(common.data_path(name))
T = orc.chordal_initialization(ds)
for i in range(ds.n):
    R = T[:, 4 * i:4 * i + 3]
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-10) and np.linalg.det(R) > 0
assert np.allclose(T[:, :3], np.eye(3)) and np.allclose(T[:, 3], 0)
out["chordal/T"] = T
out["chordal/cost"] = g2o_np.edgewise_cost(g, T)
rng = np.random.default_rng(99)
X = orc.project_to_manifold(5, 3, ds.n, rng.uniform(-1, 1, (5, 4 * ds.n)))
Tr = orc.align_lifted_trajectory_to_frame(X, X[:, 8:12], 3, ds.n, True)
# independent: R0^T X, SVD projection, translation shift
R0, p0 = X[:, 8:11], X[:, 11]
for i in (0, 7, 124):
    M = R0.T @ X[:, 4 * i:4 * i + 3]
    U, _, Vt = np.linalg.svd(M)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        U[:, -1] *= -1
    assert np.allclose(Tr[:, 4 * i:4 * i + 3], U @ Vt, atol=1e-10)
    assert np.allclose(Tr[:, 4 * i + 3], R0.T @ (X[:, 4 * i + 3] - p0), atol=1e-12)
out["round/X"] = X
out["round/anchor_pose"] = np.array(2)
out["round/T"] = Tr

# ---- robust estimation ----
r = np.array([0.1, 0.5, 2.0, 3.0, 5.0, 7.5, 20.0])
out["robust/r"] = r
for nm in ("L2", "L1", "Huber", "TLS", "GM"):
    out["robust/w_" + nm] = orc.robust_weights(r, cost_type=nm)
out["robust/w_GNC_mu0.5_updates3"] = orc.robust_weights(r, 3, cost_type="GNC_TLS", GNCInitMu=0.5)
for q, dof in ((0.9, 6), (0.95, 4), (0.99, 3)):
    assert abs(orc.chi2inv(q, dof) - chi2.ppf(q, dof)) < 1e-10
out["robust/chi2inv_0.9_6"] = orc.chi2inv(0.9, 6)
out["robust/threshold_0.9"] = orc.error_threshold_at_quantile(0.9, 3)

# ---- range-aided data feed: Q and the odometry start of the two noiseless fixtures ----
for nm in ("range_aided_slam_test_2d", "range_aided_slam_test_3d"):
    fd, tmp = tempfile.mkstemp(suffix=".pyfg")
    with os.fdopen(fd, "wb") as o, gzip.open(os.path.join(common.DATA, nm + ".pyfg.gz"), "rb") as src:
        shutil.copyfileobj(src, o)
    ra = orc.RADataset(tmp)
    os.unlink(tmp)
    Qd = ra.Q.to_scipy().toarray()
    assert np.allclose(Qd, Qd.T, atol=1e-12)
    assert np.abs(ra.gt @ Qd).max() < 1e-7 * np.abs(Qd).max()   # noiseless: the ground truth has zero gradient
    assert np.linalg.eigvalsh(Qd)[0] > -1e-8 * np.abs(Qd).max()
    out[nm + "/dims"] = np.array([ra.d, ra.n, ra.l, ra.b])
    out[nm + "/Q_dense"] = Qd
    out[nm + "/X_odom"] = ra.X_odom
    out[nm + "/gt"] = ra.gt

np.savez_compressed(os.path.join(HERE, "vectors2.npz"), **out)
print("wrote", len(out), "arrays")
