"""Device-resident multi-robot range-aided SLAM session (dcora_ra_rbcd_*: the Agents on RangeAidedSLAMGraphs and the loop
body of examples/MultiRobotExample_RASLAM.cpp) against the same loop driven from numpy over the CPU oracle's local
solver, projection and gradient: greedy block sequence, cost and gradient-norm traces, final iterate.  Also the
reference's own statement on the noiseless fixtures (tests/testAgent.cpp:290-456): the ground truth is a fixed point."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import common
from test_raslam import ra_path, ra_plain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import dcora_amd as da
    from oracle import orc
    if da.device_count() < 1:
        pytest.fail("no GPU visible: the product has no CPU fallback")
    return da, orc


from cora_flow import oracle_ra_rbcd_loop as _oracle_loop  # noqa: E402


def _lifted_start(da, orc, ra, r, seed, noise):
    rng = np.random.default_rng(seed)
    lift = np.linalg.qr(rng.standard_normal((r, ra.d)))[0]
    M = lift @ ra.gt + noise * rng.standard_normal((r, ra.k))
    return orc.project_to_manifold(r, ra.d, ra.n, M, l=ra.l, b=ra.b)


@pytest.mark.parametrize("name,r,accel,iters,restart_interval", [
    ("range_aided_slam_test_3d", 4, True, 14, 5),
    ("range_aided_slam_test_2d", 3, True, 10, 30),
    ("range_aided_slam_test_3d", 3, False, 8, 30),
    ("tiers", 3, True, 6, 4),
])
def test_ra_session_trace_matches_the_oracle_loop(env, name, r, accel, iters, restart_interval):
    da, orc = env
    ra = da.RADataset(ra_path(name))
    noise = 0.05 if name != "tiers" else 0.0
    if name == "tiers":  # the CORA driver's odometry start lifted to rank r
        X0 = np.zeros((r, ra.k))
        X0[:ra.d] = ra.X_odom
    else:
        X0 = _lifted_start(da, orc, ra, r, 5, noise)
    opt = dict(RTR_iterations=3, RTR_tCG_iterations=50, gradnorm_tol=1e-2)
    s = da.RaRbcdSession(ra, r, acceleration=accel, restart_interval=restart_interval)
    assert s.robots == ra.robots and s.R == len(ra.robots)
    s.set_X(X0)
    out = s.run(max_iters=iters, rgrad_tol=0.0)
    Xo, tr = _oracle_loop(da, orc, ra, X0, r, iters, accel, restart_interval, opt)
    assert np.array_equal(out["selected"], tr[:, 0].astype(int))
    if name == "tiers":
        # cond(Q) ~ 1e6: the trust-region ratio of a truncated local solve (3 x 50 tCG) sits at its noise floor, the
        # two arithmetic orders take slightly different steps; the loop itself (blocks, restarts) is the same
        assert np.allclose(out["cost"], tr[:, 1], rtol=5e-3)
    else:
        assert np.allclose(out["cost"], tr[:, 1], rtol=1e-7, atol=1e-12)
        assert np.allclose(out["gradnorm"], tr[:, 2], rtol=1e-4, atol=1e-9)
        assert common.rel(s.get_X(), Xo) < 1e-6
    assert out["cost"][-1] < out["cost"][0]


@pytest.mark.parametrize("inner,tol", [(3, 1e-10), (10, 1e-9)])
def test_tiers_trace_is_tight_while_the_tcg_runs_are_short(env, inner, tol):
    """the 5e-3 of the test above is conjugate gradients on cond(Q) ~ 1e6, not the loop: with the local solves capped
    at 3 or 10 tCG iterations per trust-region step (same blocks, same restarts, same everything else) the session and
    the oracle loop agree to 1e-12 over the same six iterations; runs of 50 iterations amplify the last-bit
    differences of two summation orders to 1e-3 (measured: tools/dbg_tiers_trace.py)"""
    da, orc = env
    ra = da.RADataset(ra_path("tiers"))
    r = 3
    X0 = np.zeros((r, ra.k))
    X0[:ra.d] = ra.X_odom
    opt = dict(RTR_iterations=3, RTR_tCG_iterations=inner, gradnorm_tol=1e-2)
    s = da.RaRbcdSession(ra, r, acceleration=True, restart_interval=4, params=da.ROptParameters(**opt))
    s.set_X(X0)
    out = s.run(max_iters=6, rgrad_tol=0.0)
    Xo, tr = _oracle_loop(da, orc, ra, X0, r, 6, True, 4, opt)
    assert np.array_equal(out["selected"], tr[:, 0].astype(int))
    assert np.allclose(out["cost"], tr[:, 1], rtol=tol, atol=0.0), np.abs(out["cost"] - tr[:, 1]) / np.abs(tr[:, 1])
    assert np.allclose(out["gradnorm"], tr[:, 2], rtol=1e-6)
    assert common.rel(s.get_X(), Xo) < 1e-7


@pytest.mark.parametrize("name", ["range_aided_slam_test_2d", "range_aided_slam_test_3d"])
def test_ra_session_keeps_the_ground_truth_and_returns_to_it(env, name):
    """ref tests/testAgent.cpp:290-456 with the example's local parameters (RTR 200 x 200, tol 1e-4)"""
    da, orc = env
    ra = da.RADataset(ra_path(name))
    d = ra.d
    prm = da.ROptParameters(RTR_iterations=200, RTR_tCG_iterations=200, gradnorm_tol=1e-4)
    s = da.RaRbcdSession(ra, d, params=prm)
    s.set_X(ra.gt)
    c2, gn, bn, nxt = s.evaluate()
    assert abs(c2) < 1e-12 and gn < 1e-6
    for sel in range(s.R):
        s.iterate(sel)
        assert np.abs(s.get_X() - ra.gt).max() < 1e-9  # OPTIMIZATION_TOL, ref tests/testAgent.cpp:20
    r = d + 1
    s2 = da.RaRbcdSession(ra, r, params=prm)
    s2.set_X(_lifted_start(da, orc, ra, r, 4, 0.05))
    out = s2.run(max_iters=400, rgrad_tol=1e-4)
    # two agents exchanging through a handful of ranges: block-coordinate descent closes in slowly; it must get far
    # down from the perturbed start and keep descending towards the noiseless optimum (cost 0)
    assert out["cost"][-1] < 1e-4 * out["cost"][0] and out["cost"][-1] < 1e-4
    assert out["cost"][-1] <= out["cost"][len(out["cost"]) // 2]


def test_ra_session_refuses_map_owned_variables(env):
    """landmarks without a robot letter belong to the passive map agent (ref src/Agent.cpp:541): a session that
    optimises every variable cannot host them"""
    da, orc = env
    ra = da.RADataset(ra_path("pyfg_se3_test_data"))
    with pytest.raises(da.DcoraError, match="map agent"):
        da.RaRbcdSession(ra, 4)


@pytest.mark.parametrize("name", ["range_aided_slam_test_2d", "range_aided_slam_test_3d"])
def test_multi_robot_raslam_driver_reaches_the_certified_optimum(env, name):
    """examples/MultiRobotExample_RASLAM.cpp from a random start at rank d on the noiseless fixtures: whatever
    critical points the levels pass through, the driver must end certified at (numerically) cost 0"""
    da, orc = env
    from dcora_amd import driver
    ra = da.RADataset(ra_path(name))
    d = ra.d
    rng = np.random.default_rng(12)
    X0 = orc.project_to_manifold(d, d, ra.n, rng.standard_normal((d, ra.k)), l=ra.l, b=ra.b)
    out = driver.multi_robot_raslam_example(ra, X0, max_iters=300, rgrad_tol=1e-3, r_max=d + 5)
    assert out["certified"], out["levels"]
    assert out["levels"][-1]["cost_2f"] < 1e-3  # the RBCD loop stops at |rgrad| < 1e-3
    # the oracle certifies the device's final point too
    r = out["rank"]
    So = orc.dual_certificate(r, d, ra.n, out["X"], orc.CSR.from_scipy(ra.Q.to_scipy()), l=ra.l, b=ra.b)
    assert orc.fast_verification(So, 1e-3, block=1)[0]
