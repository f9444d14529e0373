"""Rounding / solution recovery: alignLiftedTrajectoryToFrame (ref src/DCORA_utils.cpp:2262-2289),
Agent::getStatesInLocalFrame (ref src/Agent.cpp:950-1003), projectSolutionRASLAM (ref src/DCORA_utils.cpp:1984-2031).

CPU: the oracle against the reference's own checks (tests/testUtils.cpp:245-262 feasibility of the projected
solution; tests/testAgent.cpp:125-148 ground truth in the frame of pose 0 on the noiseless fixtures).
GPU: the device kernels against the oracle."""
import os

import numpy as np
import pytest

import common
import g2o_np


def _stiefel(rng, r, d):
    return np.linalg.qr(rng.standard_normal((r, d)))[0]


def _rot_col(d, l, b, i):
    """column of rotation block i: this ABI uses the SE ordering when l = b = 0 (include/dcora_hip.h), the RA
    ordering otherwise"""
    return (d + 1) * i if (l == 0 and b == 0) else d * i


def _feasible_ra(rng, d, n, l, b):
    k = (d + 1) * n + l + b
    X = rng.standard_normal((d, k))
    for i in range(n):
        Q = np.linalg.qr(rng.standard_normal((d, d)))[0]
        if np.linalg.det(Q) < 0:
            Q[:, -1] *= -1
        c = _rot_col(d, l, b, i)
        X[:, c:c + d] = Q
    if l:
        X[:, d * n:d * n + l] /= np.linalg.norm(X[:, d * n:d * n + l], axis=0, keepdims=True)
    return X


def _check_feasible(P, d, n, l, b=1):
    for i in range(n):
        c = _rot_col(d, l, b, i)
        R = P[:, c:c + d]
        assert np.allclose(R.T @ R, np.eye(d), atol=1e-10) and np.linalg.det(R) > 0.999
    if l:
        assert np.allclose(np.linalg.norm(P[:, d * n:d * n + l], axis=0), 1.0, atol=1e-12)


@pytest.mark.parametrize("name", ["pose_graph_optimization_test_2d", "pose_graph_optimization_test_3d"])
def test_oracle_alignment_recovers_ground_truth_in_the_frame_of_pose_0(built, name):
    from oracle import orc
    g = g2o_np.read_g2o(common.data_path(name))
    X = g2o_np.ground_truth_X(g)
    d, n, dh = g["d"], g["n"], g["d"] + 1
    XL = _stiefel(np.random.default_rng(0), 5, d) @ X
    for glob in (True, False):
        T = orc.align_lifted_trajectory_to_frame(XL, XL[:, :dh], d, n, glob)
        R0, t0 = X[:, :d], X[:, d]
        for i in range(n):
            assert np.allclose(T[:, dh * i:dh * i + d], R0.T @ X[:, dh * i:dh * i + d], atol=1e-9)
            assert np.allclose(T[:, dh * i + d], R0.T @ (X[:, dh * i + d] - t0), atol=1e-9)


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("l,b", [(0, 0), (6, 0), (0, 7), (6, 7)])
def test_oracle_projected_solution_is_feasible_and_keeps_a_rank_d_solution(built, d, l, b):
    """tests/testUtils.cpp:245-262 (feasibility on a random lifted point) + exactness on a lifted rank-d point"""
    from oracle import orc
    rng = np.random.default_rng(7)
    r, n = 5, 10
    k = (d + 1) * n + l + b
    M = rng.uniform(-1, 1, (r, k))
    X = orc.project_to_manifold(r, d, n, M, l=l, b=b)
    P = orc.project_solution_raslam(X, r, d, n, l, b)
    assert P.shape == (d, k)
    _check_feasible(P, d, n, l, b)
    Xd = _feasible_ra(rng, d, n, l, b)
    P = orc.project_solution_raslam(_stiefel(rng, r, d) @ Xd, r, d, n, l, b)
    assert np.allclose(P.T @ P, Xd.T @ Xd, atol=1e-9)  # equal up to a global rotation
    _check_feasible(P, d, n, l, b)


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,r", [("smallGrid3D", 5), ("pose_graph_optimization_test_2d", 4)])
def test_device_alignment_matches_oracle(built, name, r):
    import dcora_amd as da
    from oracle import orc
    ds = common.product_dataset(name)
    d, n, dh = ds.d, ds.n, ds.d + 1
    X = common.random_point(r, d, n, 21, orc.project_to_manifold)
    anchor = X[:, 3 * dh:4 * dh]  # any lifted pose may serve as the global anchor
    for glob in (True, False):
        To = orc.align_lifted_trajectory_to_frame(X, anchor, d, n, glob)
        Tp = da.align_lifted_trajectory_to_frame(X, anchor, d, n, glob)
        if not glob:
            # local frame: origin = pose 0 of the trajectory (ref src/Agent.cpp:963-980)
            To = orc.align_lifted_trajectory_to_frame(X, X[:, :dh], d, n, False)
            Tp = da.align_lifted_trajectory_to_frame(X, None, d, n, False)
        assert np.abs(Tp - To).max() < 1e-10
        for i in range(n):
            R = Tp[:, dh * i:dh * i + d]
            assert np.allclose(R.T @ R, np.eye(d), atol=1e-10) and np.linalg.det(R) > 0.999


@pytest.mark.gpu
def test_device_rounding_of_the_certified_solution_of_smallGrid3D(built):
    """solve -> certify -> round: the rounded trajectory is feasible and its cost is within the usual rounding gap"""
    import dcora_amd as da
    from oracle import orc
    ds = common.product_dataset("smallGrid3D")
    g = g2o_np.read_g2o(common.data_path("smallGrid3D"))
    r, d, n = 5, ds.d, ds.n
    T0 = da.chordal_initialization(ds)
    X0 = np.zeros((r, 4 * n))
    X0[:d] = T0
    s = da.RbcdSession(ds, num_robots=5, r=r)
    s.set_X(X0)
    out = s.run(max_iters=500, rgrad_tol=1e-3)
    X = s.get_X()
    T = da.align_lifted_trajectory_to_frame(X, X[:, :4], d, n, True)
    assert np.abs(T - orc.align_lifted_trajectory_to_frame(X, X[:, :4], d, n, True)).max() < 1e-9
    f_sdp, f_round = 0.5 * out["cost"][-1], g2o_np.edgewise_cost(g, T)
    assert f_round >= f_sdp - 1e-6 * abs(f_sdp)     # the relaxation is a lower bound
    assert f_round <= 1.001 * f_sdp                  # exact recovery when the solution has rank d
    assert np.allclose(T[:, :3], np.eye(3), atol=1e-9) and np.allclose(T[:, 3], 0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["range_aided_slam_test_2d", "range_aided_slam_test_3d"])
def test_device_ra_states_and_projection_match_oracle(built, name):
    import dcora_amd as da
    from oracle import orc
    ra = da.RADataset(os.path.join(common.DATA, name + ".pyfg.gz"))
    d, n, l, b, r = ra.d, ra.n, ra.l, ra.b, 5
    rng = np.random.default_rng(3)
    # (a) lifted ground truth: states in the local frame = ground truth in the frame of pose 0
    XL = _stiefel(rng, r, d) @ ra.gt
    Tp, Sp, Lp = da.ra_states_in_local_frame(XL, r, d, n, l, b)
    To, So, Lo = orc.ra_states_in_local_frame(XL, r, d, n, l, b)
    assert np.abs(Tp - To).max() < 1e-10 and np.abs(Sp - So).max() < 1e-10 and np.abs(Lp - Lo).max() < 1e-10
    R0, t0 = ra.gt[:, :d], ra.gt[:, d * n + l]
    for i in range(n):
        assert np.allclose(Tp[:, (d + 1) * i:(d + 1) * i + d], R0.T @ ra.gt[:, d * i:d * i + d], atol=1e-9)
        assert np.allclose(Tp[:, (d + 1) * i + d], R0.T @ (ra.gt[:, d * n + l + i] - t0), atol=1e-9)
    assert np.allclose(Sp, R0.T @ ra.gt[:, d * n:d * n + l], atol=1e-9)
    assert np.allclose(Lp, R0.T @ (ra.gt[:, d * n + l + n:] - t0[:, None]), atol=1e-9)
    # (b) projectSolutionRASLAM: same Gram matrix as the oracle (both are defined up to a global rotation)
    X = orc.project_to_manifold(r, d, n, rng.uniform(-1, 1, (r, ra.k)), l=l, b=b)
    Pp = da.project_solution_raslam(X, r, d, n, l, b)
    Po = orc.project_solution_raslam(X, r, d, n, l, b)
    _check_feasible(Pp, d, n, l)
    assert np.allclose(Pp.T @ Pp, Po.T @ Po, atol=1e-8)
    Pp = da.project_solution_raslam(XL, r, d, n, l, b)
    assert np.allclose(Pp.T @ Pp, ra.gt.T @ ra.gt, atol=1e-8)


@pytest.mark.parametrize("d", [2, 3])
def test_trajectory_log_format(built, d, tmp_path):
    """Logger::logTrajectory (ref src/Logger.cpp:107-145): header, one line per pose, 9 decimals, x y z qx qy qz qw
    with Eigen's rotation-to-quaternion branch rule; planar poses are embedded in 3D"""
    import dcora_amd as da
    import g2o_np
    rng = np.random.default_rng(3)
    n = 40
    T = np.zeros((d, (d + 1) * n))
    Rs = []
    for i in range(n):
        Q = np.linalg.qr(rng.standard_normal((d, d)))[0]
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        if i == 0:
            Q = np.eye(d)
        if i == 1 and d == 3:  # half turn: trace < 0 branch
            Q = np.diag([1.0, -1.0, -1.0])
        Rs.append(Q)
        T[:, i * (d + 1):i * (d + 1) + d] = Q
        T[:, i * (d + 1) + d] = 10 * rng.standard_normal(d)
    p = tmp_path / "trajectory.txt"
    da.log_trajectory(p, T, d, n)
    lines = p.read_text().splitlines()
    assert lines[0] == "# pose_index x y z qx qy qz qw" and len(lines) == n + 1
    for i, line in enumerate(lines[1:]):
        tok = line.split()
        assert int(tok[0]) == i and all(len(x.split(".")[1]) == 9 for x in tok[1:])
        v = np.array(list(map(float, tok[1:])))
        t3 = np.zeros(3)
        t3[:d] = T[:, i * (d + 1) + d]
        assert np.abs(v[:3] - t3).max() < 1e-9
        R3 = np.eye(3)
        R3[:d, :d] = Rs[i]
        assert abs(np.linalg.norm(v[3:]) - 1) < 1e-8
        assert np.abs(g2o_np.quat_R(*v[3:]) - R3).max() < 1e-8
        tr = np.trace(R3)
        if tr > 0:
            assert v[6] > 0  # w = sqrt(trace + 1) / 2
        else:
            a = int(np.argmax(np.diag(R3)))
            assert v[3 + a] > 0
    with pytest.raises(da.DcoraError):
        da.log_trajectory(tmp_path / "no_such_dir" / "t.txt", T, d, n)
