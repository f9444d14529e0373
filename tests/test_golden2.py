"""tests/golden/vectors2.npz (made by tests/golden/make_golden2.py): the oracle and the product against committed
vectors of the pieces either side of the hot path -- chordal initialisation, rounding, robust weights, chi-square
thresholds, the range-aided Q and the CORA odometry start."""
import os

import numpy as np
import pytest

import common

GOLD = np.load(os.path.join(common.HERE, "golden", "vectors2.npz"))
RA = ("range_aided_slam_test_2d", "range_aided_slam_test_3d")


def test_oracle_reproduces_golden2(built):
    from oracle import orc
    ds = common.oracle_dataset("smallGrid3D")
    assert np.abs(orc.chordal_initialization(ds) - GOLD["chordal/T"]).max() < 1e-10
    X, a = GOLD["round/X"], int(GOLD["round/anchor_pose"])
    T = orc.align_lifted_trajectory_to_frame(X, X[:, 4 * a:4 * a + 4], 3, ds.n, True)
    assert np.abs(T - GOLD["round/T"]).max() < 1e-12
    r = GOLD["robust/r"]
    for nm in ("L2", "L1", "Huber", "TLS", "GM"):
        assert np.array_equal(orc.robust_weights(r, cost_type=nm), GOLD["robust/w_" + nm])
    assert np.allclose(orc.robust_weights(r, 3, cost_type="GNC_TLS", GNCInitMu=0.5),
                       GOLD["robust/w_GNC_mu0.5_updates3"], rtol=1e-15)
    assert abs(orc.chi2inv(0.9, 6) - float(GOLD["robust/chi2inv_0.9_6"])) < 1e-12


def test_product_host_functions_match_golden2(built):
    """the pieces of the product that need no device: chordal initialisation, readers / Q builder, robust weights"""
    import dcora_amd as da
    import dcora_amd.robust as hip
    ds = common.product_dataset("smallGrid3D")
    assert np.abs(da.chordal_initialization(ds) - GOLD["chordal/T"]).max() < 1e-10
    r = GOLD["robust/r"]
    for nm in ("L2", "L1", "Huber", "TLS", "GM"):
        assert np.array_equal(hip.robust_weights(r, hip.RobustCostParameters(nm)), GOLD["robust/w_" + nm])
    p = hip.RobustCostParameters("GNC_TLS", GNCInitMu=0.5)
    assert np.allclose(hip.robust_weights(r, p, 3), GOLD["robust/w_GNC_mu0.5_updates3"], rtol=1e-15)
    assert abs(hip.chi2inv(0.9, 6) - float(GOLD["robust/chi2inv_0.9_6"])) < 1e-12
    assert abs(hip.computeErrorThresholdAtQuantile(0.9, 3) - float(GOLD["robust/threshold_0.9"])) < 1e-12
    for nm in RA:
        ra = da.RADataset(os.path.join(common.DATA, nm + ".pyfg.gz"))
        assert [ra.d, ra.n, ra.l, ra.b] == list(GOLD[nm + "/dims"])
        Qg = GOLD[nm + "/Q_dense"]
        assert np.abs(ra.Q.to_scipy().toarray() - Qg).max() < 1e-10 * np.abs(Qg).max()
        assert np.array_equal(ra.X_odom, GOLD[nm + "/X_odom"])
        assert np.array_equal(ra.gt, GOLD[nm + "/gt"])


@pytest.mark.gpu
def test_hip_matches_golden2(built):
    import dcora_amd as da
    ds = common.product_dataset("smallGrid3D")
    X, a = GOLD["round/X"], int(GOLD["round/anchor_pose"])
    T = da.align_lifted_trajectory_to_frame(X, X[:, 4 * a:4 * a + 4], 3, ds.n, True)
    assert np.abs(T - GOLD["round/T"]).max() < 1e-10
    # the chordal start evaluated on the device: 2 f = sum of weighted residuals
    Q = da.build_Q_pgo(ds)
    P = da.QuadraticProblem(3, ds.d, ds.n, Q, reg=-1.0)
    assert abs(P.f(GOLD["chordal/T"]) - float(GOLD["chordal/cost"])) < 1e-9 * float(GOLD["chordal/cost"])
    for nm in RA:
        d, n, l, b = (int(x) for x in GOLD[nm + "/dims"])
        ra = da.RADataset(os.path.join(common.DATA, nm + ".pyfg.gz"))
        Pr = da.QuadraticProblem(d, d, n, ra.Q, reg=-1.0, l=l, b=b)
        # noiseless fixture: zero cost and zero gradient at the ground truth, on the device
        assert abs(Pr.f(GOLD[nm + "/gt"])) < 1e-10 and Pr.RieGradNorm(GOLD[nm + "/gt"]) < 1e-6
