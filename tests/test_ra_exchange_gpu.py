"""The multi-robot range-aided SLAM session with ONE PROCESS PER RANK (partitioning by robot symbol for pyfg
files, ref src/DCORA_utils.cpp:1370-1512): several processes run the RBCD++ loop of
examples/MultiRobotExample_RASLAM.cpp through dcora_exchange_create_ra -- an agent's public poses, unit spheres and
landmarks travel between the ranks like public poses do -- and must reproduce the one-process session: same blocks, costs
to rounding (the evaluation sums per agent instead of centrally), the same iterates."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest
import scipy.sparse.linalg as spla

import common
from test_raslam import ra_path

pytestmark = pytest.mark.gpu

WORKER = os.path.join(common.HERE, "ra_exchange_worker.py")


def run_ranks(tmp_path, world, name, r, iters, X0, accel, restart, transport=None, certify=None):
    np.save(os.path.join(tmp_path, "X0.npy"), X0)
    job = "ra%s" % uuid.uuid4().hex[:12]
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if transport:
        env["DCORA_EXCHANGE"] = transport
    else:
        env.pop("DCORA_EXCHANGE", None)
    procs = [subprocess.Popen([sys.executable, WORKER, str(k), str(world), job, name, str(r), str(iters), str(tmp_path),
                               str(int(accel)), str(restart)] + ([] if certify is None else [repr(certify)]), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for k in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for k, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (k, outs[k][-3000:])
    return [np.load(os.path.join(tmp_path, "rank%d.npz" % k)) for k in range(world)]


@pytest.mark.parametrize("name,r,world,iters,accel,restart,transport", [
    ("range_aided_slam_test_3d", 4, 2, 14, True, 5, None),      # two robots, one per rank; restart rounds inside
    ("range_aided_slam_test_2d", 3, 2, 10, False, 30, "staged"),
    ("tiers", 3, 2, 5, True, 4, None),                           # four robots, two per rank; a landmark every pose ranges to
    ("tiers", 3, 4, 4, True, 30, None),                          # one robot per rank
])
def test_ra_ranks_reproduce_the_one_process_session(tmp_path, name, r, world, iters, accel, restart, transport):
    import dcora_amd as da
    ra = da.RADataset(ra_path(name))
    if name == "tiers":
        X0 = np.zeros((r, ra.k))
        X0[:ra.d] = ra.X_odom
    else:
        rng = np.random.default_rng(5)
        lift = np.linalg.qr(rng.standard_normal((r, ra.d)))[0]
        X0 = da.manifold_project(r, ra.d, ra.n, lift @ ra.gt + 0.05 * rng.standard_normal((r, ra.k)), l=ra.l, b=ra.b)
    s = da.RaRbcdSession(ra, r, acceleration=accel, restart_interval=restart)
    s.set_X(X0)
    out = s.run(max_iters=iters, rgrad_tol=0.0)
    X = s.get_X()
    s.close()
    res = run_ranks(str(tmp_path), world, name, r, iters, X0, accel, restart, transport)
    for k, o in enumerate(res):
        assert int(o["mode"]) == (2 if transport == "staged" else 1)
        assert np.array_equal(o["selected"], out["selected"]), (k, o["selected"], out["selected"])
        assert np.allclose(o["cost"], out["cost"], rtol=1e-9, atol=1e-12), (k, o["cost"], out["cost"])
        assert np.allclose(o["gradnorm"], out["gradnorm"], rtol=1e-6, atol=1e-9)
        assert np.array_equal(o["X"], X), "rank %d: iterates differ from the one-process session (max %g)" % (
            k, np.max(np.abs(o["X"] - X)))
        assert np.array_equal(o["cost"], res[0]["cost"])
    assert sum(int(o["posts"]) for o in res) > 0


@pytest.mark.parametrize("name,r,world,iters", [
    ("range_aided_slam_test_3d", 4, 2, 6),   # two robots, one per rank
    ("range_aided_slam_test_2d", 3, 2, 6),
    ("tiers", 3, 2, 4),                      # four robots, two per rank; lambda_max ~ 2e6: the row-block runs cannot converge
    ("tiers", 3, 4, 4),
])
def test_ra_certification_across_ranks_matches_one_gpu(tmp_path, name, r, world, iters):
    """dcora_exchange_certify on a range-aided session (a rank's rows = its agents' variables, scattered over the global
    ordering; Lambda on rotation blocks AND unit spheres) against fastVerification of the same iterate on one GPU: same
    verdict, lambda_min, theta, the eigenvector up to its sign, identical bits on every rank.  On tiers.pyfg the landmark
    every pose ranges to puts the largest eigenvalue at 2e6: the row-block Lanczos runs end without converging and the
    eigenpair comes from rank 0's shift-and-invert run (distributed = 0), as on one GPU."""
    import dcora_amd as da
    ra = da.RADataset(ra_path(name))
    eta = 1e-3
    if name == "tiers":
        X0 = np.zeros((r, ra.k))
        X0[:ra.d] = ra.X_odom
    else:
        X0 = da.manifold_project(r, ra.d, ra.n, np.random.default_rng(7).standard_normal((r, ra.k)), l=ra.l, b=ra.b)
    res = run_ranks(str(tmp_path), world, name, r, iters, X0, True, 30, None, certify=eta)
    X = res[0]["X"]
    S = da.dual_certificate(r, ra.d, ra.n, X, ra.Q, l=ra.l, b=ra.b)
    psd, theta, v, lmin = da.fast_verification(S, eta, block=1)
    A = S.to_scipy()
    assert not psd
    for k, o in enumerate(res):
        assert not bool(o["cert_ok"])
        if name != "tiers":
            assert bool(o["cert_distributed"]), "rank %d: the row-block Lanczos runs did not converge" % k
        assert int(o["cert_matvecs"]) > 0
        lam, th, vv = float(o["cert_lambda"]), float(o["cert_theta"]), o["cert_v"]
        assert abs(lam - lmin) <= 1e-6 * max(1.0, abs(lmin)), (lam, lmin)
        assert abs(np.linalg.norm(vv) - 1) < 1e-9
        assert abs(th - vv @ (A @ vv)) <= 1e-9 * max(1.0, abs(th))          # theta is the curvature along v
        assert abs(th - theta) <= 1e-5 * max(1.0, abs(theta)), (th, theta)
        assert min(np.linalg.norm(vv - v), np.linalg.norm(vv + v)) < 1e-3
        # (S + eta I) v = lambda v to the tolerance of the run, which is relative to the largest eigenvalue
        assert np.linalg.norm(A @ vv + eta * vv - lam * vv) < 1e-3 + 1e-4 * spla.norm(A, 1)
        assert np.array_equal(vv, res[0]["cert_v"]) and lam == float(res[0]["cert_lambda"])
