"""Robust estimation (ref include/DCORA/DCORA_robust.h, src/DCORA_robust.cpp, src/DCORA_solver.cpp).  The cases
are the reference's own tests restated: tests/testRobust.cpp:24-42 (trivial rotation averaging), :44-75 (10 inliers
among 50), :77-102 (trivial pose averaging), :228-309 (testRobustPGO: one inlier and one outlier loop closure),
tests/testUtils.cpp:136-152 (chi2inv).  CPU: oracle + the product's host code (RobustCost, averaging, chi2inv need
no device).  GPU: per-measurement residuals, solvePGO, solveRobustPGO."""
import numpy as np
import pytest
from scipy.stats import chi2

import common


def angular2ChordalSO3(rad):
    return 2 * np.sqrt(2) * np.sin(rad / 2)  # ref src/DCORA_utils.cpp:2108


def rand_rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def backends():
    import dcora_amd.robust as hip
    from oracle import orc
    return [("oracle", orc.robust_single_rotation_averaging, orc.robust_single_pose_averaging),
            ("product", lambda Rs, kappa, threshold: hip.robustSingleRotationAveraging(Rs, kappa, threshold),
             lambda Rs, ts, kappa, tau, threshold: hip.robustSinglePoseAveraging(Rs, ts, kappa, tau, threshold))]


def test_chi2inv_and_error_threshold(built):
    import dcora_amd.robust as hip
    from oracle import orc
    for q, dof in [(0.95, 4), (0.9, 6), (0.5, 1), (0.999, 6), (0.01, 3)]:
        assert abs(orc.chi2inv(q, dof) - chi2.ppf(q, dof)) < 1e-10
        assert abs(hip.chi2inv(q, dof) - chi2.ppf(q, dof)) < 1e-10
    # tests/testUtils.cpp:136-152: the 0.95 quantile at 4 dof covers 95 % of the samples
    thr = hip.chi2inv(0.95, 4)
    x = np.random.default_rng(1).chisquare(4, 100000)
    assert abs(np.mean(x < thr) - 0.95) < 0.01
    assert abs(hip.computeErrorThresholdAtQuantile(0.9, 3) - np.sqrt(chi2.ppf(0.9, 6))) < 1e-10
    assert hip.computeErrorThresholdAtQuantile(1.0, 3) == 1e5 and orc.error_threshold_at_quantile(1.0, 3) == 1e5
    with pytest.raises(Exception):
        hip.computeErrorThresholdAtQuantile(0.9, 2)  # "only supports 3D problem" (CHECK in the reference)


def test_robust_cost_weights(built):
    """ref src/DCORA_robust.cpp:56-100, known values of every weight function; GNC mu update :116-136"""
    import dcora_amd.robust as hip
    from oracle import orc
    r = np.array([0.5, 2.0, 5.0, 20.0])
    cases = {"L2": np.ones(4), "L1": 1 / r, "Huber": np.array([1, 1, 3 / 5, 3 / 20]), "TLS": np.array([1, 1, 1, 0.0]),
             "GM": 1 / (1 + r * r) ** 2}
    for name, want in cases.items():
        assert np.allclose(orc.robust_weights(r, cost_type=name), want, rtol=1e-15)
        assert np.allclose(hip.robust_weights(r, hip.RobustCostParameters(name)), want, rtol=1e-15)
    # GNC-TLS, eq. (14) of Yang et al.: 0 above (mu+1)/mu barc^2, 1 below mu/(mu+1) barc^2, sqrt(...) - mu between
    for updates in (0, 1, 5):
        mu, barc = 0.5 * 1.4 ** updates, 5.0
        want = []
        for ri in r:
            rs, b2 = ri * ri, barc * barc
            want.append(0.0 if rs >= (mu + 1) / mu * b2 else 1.0 if rs <= mu / (mu + 1) * b2
                        else np.sqrt(b2 * mu * (mu + 1) / rs) - mu)
        p = hip.RobustCostParameters("GNC_TLS", GNCInitMu=0.5)
        assert np.allclose(hip.robust_weights(r, p, updates), want, rtol=1e-14)
        assert np.allclose(orc.robust_weights(r, updates, cost_type="GNC_TLS", GNCInitMu=0.5), want, rtol=1e-14)
    # update() stops growing mu after GNCMaxNumIters
    p = hip.RobustCostParameters("GNC_TLS", GNCInitMu=0.5, GNCMaxNumIters=3)
    assert np.array_equal(hip.robust_weights(r, p, 3), hip.robust_weights(r, p, 10))


def test_robust_single_rotation_averaging_trivial(built):
    rng = np.random.default_rng(0)
    for name, rot_avg, _ in backends():
        for trial in range(50):
            Rt = rand_rot(rng)
            R, inl = rot_avg([Rt], np.ones(1), angular2ChordalSO3(0.5))
            assert np.linalg.norm(R - Rt) <= 1e-8 and list(inl) == [0], name


def test_robust_single_rotation_averaging(built):
    rng = np.random.default_rng(1)
    tol, cbar = angular2ChordalSO3(0.02), angular2ChordalSO3(0.3)
    for name, rot_avg, _ in backends():
        for trial in range(50):
            Rt = rand_rot(rng)
            Rs = [Rt.copy() for _ in range(10)]
            while len(Rs) < 50:
                Rr = rand_rot(rng)
                if np.linalg.norm(Rr - Rt) > 1.2 * cbar:
                    Rs.append(Rr)
            R, inl = rot_avg(Rs, np.ones(50), cbar)
            assert abs(np.linalg.det(R) - 1) < 1e-9 and np.allclose(R.T @ R, np.eye(3), atol=1e-9)
            assert np.linalg.norm(R - Rt) <= tol, name
            assert list(inl) == list(range(10)), name


def test_robust_single_pose_averaging_trivial(built):
    import dcora_amd.robust as hip
    rng = np.random.default_rng(2)
    barc = hip.computeErrorThresholdAtQuantile(0.9, 3)
    for name, _, pose_avg in backends():
        for trial in range(50):
            Rt, tt = rand_rot(rng), np.zeros(3)
            R, t, inl = pose_avg([Rt], [tt], 10000 * np.ones(1), 100 * np.ones(1), barc)
            assert np.linalg.norm(R - Rt) <= 1e-8 and np.linalg.norm(t - tt) <= 1e-8 and list(inl) == [0], name


def test_robust_pose_averaging_rejects_outliers(built):
    """the case the reference keeps commented out (dpgo issue #2): checked here with well separated outliers"""
    import dcora_amd.robust as hip
    rng = np.random.default_rng(3)
    barc = hip.computeErrorThresholdAtQuantile(0.9, 3)
    for name, _, pose_avg in backends():
        Rt, tt = rand_rot(rng), np.zeros(3)
        Rs, ts = [Rt.copy() for _ in range(10)], [tt.copy() for _ in range(10)]
        while len(Rs) < 30:
            Rr, tr = rand_rot(rng), rng.uniform(-1, 1, 3)
            if np.sqrt(10000 * np.sum((Rt - Rr) ** 2) + 100 * np.sum((tt - tr) ** 2)) > 20 * barc:
                Rs.append(Rr)
                ts.append(tr)
        R, t, inl = pose_avg(Rs, ts, 10000 * np.ones(30), 100 * np.ones(30), barc)
        assert list(inl) == list(range(10)), name
        assert np.linalg.norm(R - Rt) <= angular2ChordalSO3(0.02) and np.linalg.norm(t - tt) <= 1e-2


# ------------------------------------------------------------------------------------------------------------------
def robust_pgo_case(seed):
    """tests/testRobust.cpp:228-298: 4 poses, odometry (fixed weights), one inlier and one outlier loop closure"""
    import dcora_amd as da
    rng = np.random.default_rng(seed)
    d, n, kappa, tau = 3, 4, 10000.0, 100.0
    Rs = [rand_rot(rng) for _ in range(n)]
    ts = [i * np.ones(3) for i in range(n)]

    def rel(i, j):
        return Rs[i].T @ Rs[j], Rs[i].T @ (ts[j] - ts[i])

    edges = [(i, i + 1) + rel(i, i + 1) + (True,) for i in range(n - 1)]
    edges.append((0, 3) + rel(0, 3) + (False,))
    edges.append((1, 3, rand_rot(rng), np.zeros(3), False))
    ids = np.zeros((len(edges), 4), np.int32)
    vals = np.zeros((len(edges), 15))
    for e, (i, j, R, t, fx) in enumerate(edges):
        ids[e, 1], ids[e, 3] = i, j
        vals[e, :9] = R.T.reshape(-1)  # column-major
        vals[e, 9:12] = t
        vals[e, 12:] = kappa, tau, 1.0
    fixed = np.array([int(e[4]) for e in edges], np.int32)
    # odometryInitialization (ref src/DCORA_solver.cpp:270-302)
    T0 = np.zeros((3, 4 * n))
    T0[:, :3] = np.eye(3)
    for i in range(1, n):
        Rm, tm = edges[i - 1][2], edges[i - 1][3]
        T0[:, 4 * i:4 * i + 3] = T0[:, 4 * (i - 1):4 * (i - 1) + 3] @ Rm
        T0[:, 4 * i + 3] = T0[:, 4 * (i - 1) + 3] + T0[:, 4 * (i - 1):4 * (i - 1) + 3] @ tm
    return da.Dataset(d, n, ids, vals), fixed, T0


def test_oracle_robust_pgo_classifies_inlier_and_outlier(built):
    from oracle import orc
    for seed in range(5):
        ds, fixed, T0 = robust_pgo_case(seed)
        dso = orc.Dataset(ds.d, ds.n, ds.ids, ds.vals)
        T, w = orc.solve_robust_pgo(dso, fixed, T0=T0, robust=dict(cost_type="GNC_TLS", GNCBarc=7.0),
                                    gradnorm_tol=1e-1, RTR_iterations=50)
        assert abs(w[3] - 1) < 1e-6 and abs(w[4]) < 1e-6 and np.all(w[:3] == 1)
        # with the outlier switched off the odometry-consistent trajectory has (almost) zero residual everywhere else
        err = orc.measurement_errors(dso, T)
        assert np.all(err[:4] < 1e-6)


@pytest.mark.gpu
def test_measurement_errors_match_oracle(built):
    import dcora_amd as da
    import dcora_amd.robust as hip
    from oracle import orc
    ds, dso = common.product_dataset("smallGrid3D"), common.oracle_dataset("smallGrid3D")
    T = da.chordal_initialization(ds)
    eo = orc.measurement_errors(dso, T)
    assert np.allclose(hip.measurement_errors(ds, T), eo, rtol=1e-12, atol=1e-12)
    # lifted poses give the same residuals (Agent::computeMeasurementResidual works on the lifted iterate)
    lift = np.linalg.qr(np.random.default_rng(0).standard_normal((5, 3)))[0]
    assert np.allclose(hip.measurement_errors(ds, lift @ T), eo, rtol=1e-10, atol=1e-10)
    # cost = 1/2 sum w (residual)^2
    Q = da.build_Q_pgo(ds)
    P = da.QuadraticProblem(3, ds.d, ds.n, Q, reg=-1.0)
    assert abs(P.f(T) - 0.5 * np.sum(ds.vals[:, -1] * eo)) < 1e-8 * abs(P.f(T))


@pytest.mark.gpu
def test_solve_pgo_matches_oracle(built):
    import dcora_amd as da
    import dcora_amd.robust as hip
    from oracle import orc
    ds, dso = common.product_dataset("smallGrid3D"), common.oracle_dataset("smallGrid3D")
    prm = da.ROptParameters(RTR_iterations=50, gradnorm_tol=1e-3)
    T, res = hip.solvePGO(ds, prm)
    To = orc.solve_pgo(dso, RTR_iterations=50, gradnorm_tol=1e-3)
    assert res["gradNormOpt"] < 1e-3
    assert common.rel(T, To) < 1e-6
    Q = da.build_Q_pgo(ds)
    P = da.QuadraticProblem(3, ds.d, ds.n, Q, reg=-1.0)
    assert abs(res["fOpt"] - P.f(To)) < 1e-8 * abs(res["fOpt"])


@pytest.mark.gpu
def test_solve_robust_pgo_classifies_inlier_and_outlier(built):
    """tests/testRobust.cpp:228-309"""
    import dcora_amd as da
    import dcora_amd.robust as hip
    from oracle import orc
    prm = da.ROptParameters(gradnorm_tol=1e-1, RTR_iterations=50)
    rob = hip.RobustCostParameters("GNC_TLS", GNCBarc=7.0)
    for seed in range(5):
        ds, fixed, T0 = robust_pgo_case(seed)
        T, w = hip.solveRobustPGO(ds, prm, rob, fixedWeight=fixed, T0=T0)
        assert abs(w[3] - 1) < 1e-6 and abs(w[4]) < 1e-6 and np.all(w[:3] == 1)
        dso = orc.Dataset(ds.d, ds.n, ds.ids, ds.vals)
        To, wo = orc.solve_robust_pgo(dso, fixed, T0=T0, robust=dict(cost_type="GNC_TLS", GNCBarc=7.0),
                                      gradnorm_tol=1e-1, RTR_iterations=50)
        assert np.allclose(w, wo, atol=1e-6)
        # same trajectory up to the solver tolerance (gradnorm_tol 1e-1 on kappa = 1e4: ~1e-5 in the poses)
        assert np.abs(T - To).max() < 1e-3
