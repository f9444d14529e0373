"""Partitioned sparse inverse of the preconditioner (dcora_amd/csrc/sparse_precond.h), the large-block form of
(Q + reg I)^-1 (ref src/Graph.cpp:1901-1917, applied in src/QuadraticProblem.cpp:70-84).

CPU: the host builder's schedule replayed on the host equals a plain sparse Cholesky solve.
GPU: the device replay equals the oracle's CHOLMOD-style solve, alone and inside RTR / RBCD."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

import common


def _selftest(A, block, r):
    from dcora_amd import capi
    L = C.CDLL(capi.LIB_PATH)
    A = sp.csr_matrix(A)
    A.sort_indices()
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    err, info = C.c_double(), np.zeros(4)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.dcora_debug_partinv_selftest(C.c_int(A.shape[0]), vp(rp), vp(ci), vp(v), C.c_int(block), C.c_int(r),
                                        C.byref(err), vp(info))
    return rc, err.value, info


@pytest.mark.parametrize("name", ["tinyGrid3D", "smallGrid3D", "sphere2500", "pose_graph_optimization_test_2d"])
def test_schedule_replayed_on_the_host_solves_the_system(built, name):
    import dcora_amd as da
    ds = common.product_dataset(name)
    Q = da.build_Q_pgo(ds).to_scipy()
    A = Q + 0.1 * sp.identity(Q.shape[0])
    rc, err, info = _selftest(A, ds.d + 1, 5)
    assert rc == 0 and err < 1e-12
    # forward + backward levels, the top level's two steps merged into one: odd, and shallow (~ log2(n) levels)
    assert info[0] % 2 == 1 and info[0] <= 2 * (2 + np.ceil(np.log2(max(ds.n, 2))))


def test_schedule_on_the_range_aided_layout_and_scalar_blocks(built):
    import dcora_amd as da
    ra = da.RADataset(os.path.join(common.DATA, "single_drone.pyfg.gz"))
    Q = ra.Q.to_scipy()
    A = Q + 0.5 * sp.identity(Q.shape[0])
    rc, err, info = _selftest(A, 1, 3)
    assert rc == 0 and err < 1e-11


@pytest.mark.parametrize("name,cap", [("smallGrid3D", 1 << 20), ("smallGrid3D", 3000), ("sphere2500", 50000),
                                      ("pose_graph_optimization_test_2d", 4096)])
def test_weights_streamed_in_chunks_equal_the_weights_written_in_one_piece(built, name, cap):
    """the product never holds the stored weights on the host: write_weights hands them to a WeightSink in ascending
    chunks (DeviceWeightSink forwards each to the device while the next is formed).  Here a host sink with a small
    chunk collects them: every weight arrives exactly once, in order, bit-equal to the one-piece build, and padding
    between fills arrives as zeros although the chunk memory is recycled dirty."""
    import dcora_amd as da
    from dcora_amd import capi
    ds = common.product_dataset(name)
    A = sp.csr_matrix(da.build_Q_pgo(ds).to_scipy() + 0.1 * sp.identity((ds.d + 1) * ds.n))
    A.sort_indices()
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    out = np.zeros(4)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L = C.CDLL(capi.LIB_PATH)
    rc = L.dcora_debug_partinv_stream_check(C.c_int(A.shape[0]), vp(rp), vp(ci), vp(v), C.c_int(ds.d + 1),
                                            C.c_longlong(cap), vp(out))
    assert rc == 0
    assert out[0] > 0 and out[2] == 0, out
    assert out[3] <= cap and out[1] >= np.ceil(out[0] / cap)


def test_indefinite_matrix_is_reported(built):
    A = sp.diags([1.0, -1.0, 2.0]).tocsr()
    rc, err, info = _selftest(A, 1, 1)
    assert rc != 0


# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def sparse_env(built):
    old = os.environ.get("DCORA_PRECOND")
    os.environ["DCORA_PRECOND"] = "sparse"
    yield
    if old is None:
        del os.environ["DCORA_PRECOND"]
    else:
        os.environ["DCORA_PRECOND"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("name,r", [("smallGrid3D", 5), ("sphere2500", 5), ("sphere2500", 7),
                                    ("pose_graph_optimization_test_2d", 3),
                                    # every instantiation of k_sp_mtile: one column per lane (r <= 4; r = 2 runs on the
                                    # 2-D graph), a column pair per lane (r = 5 .. 8), two pairs (r > 8)
                                    ("pose_graph_optimization_test_2d", 2), ("sphere2500", 3), ("sphere2500", 4),
                                    ("sphere2500", 6), ("sphere2500", 8), ("sphere2500", 9), ("smallGrid3D", 12)])
def test_device_replay_matches_oracle_preconditioner(sparse_env, name, r):
    import dcora_amd as da
    from oracle import orc
    ds, dso = common.product_dataset(name), common.oracle_dataset(name)
    Q, Qo = da.build_Q_pgo(ds), orc.build_Q_pgo(dso)
    P = da.QuadraticProblem(r, ds.d, ds.n, Q, reg=0.1)
    Po = orc.Problem(r, ds.d, ds.n, Qo)
    assert P.precond_info()["kind"] == "sparse"
    X = common.random_point(r, ds.d, ds.n, 11, orc.project_to_manifold)
    V = common.random_tangent(r, ds.d, ds.n, 12)
    assert common.rel(P.PreCondition(X, V), Po.precondition(X, V)) < 1e-10
    P.close()


@pytest.mark.gpu
def test_rtr_with_sparse_preconditioner_matches_oracle(sparse_env):
    import dcora_amd as da
    from oracle import orc
    name, r = "smallGrid3D", 5
    ds, dso = common.product_dataset(name), common.oracle_dataset(name)
    Q, Qo = da.build_Q_pgo(ds), orc.build_Q_pgo(dso)
    P = da.QuadraticProblem(r, ds.d, ds.n, Q, reg=0.1)
    Po = orc.Problem(r, ds.d, ds.n, Qo)
    X0 = common.random_point(r, ds.d, ds.n, 5, orc.project_to_manifold)
    opt = da.QuadraticOptimizer(P)
    X = opt.optimize(X0)
    res = opt.getOptResult()
    Xo, reso = Po.optimize(X0)
    assert res["outer_iterations"] == reso["outer_iters"], (res, reso)
    assert res["inner_iterations"] == reso["inner_iters"]
    assert abs(res["fOpt"] - reso["fOpt"]) <= 1e-9 * abs(reso["fOpt"])
    assert common.rel(X, Xo) < 1e-7
    P.close()


@pytest.mark.gpu
def test_rbcd_with_sparse_preconditioner_matches_oracle(sparse_env):
    import dcora_amd as da
    from oracle import orc
    ds, dso = common.product_dataset("smallGrid3D"), common.oracle_dataset("smallGrid3D")
    r, iters = 5, 40
    X0 = common.random_point(r, ds.d, ds.n, 1, orc.project_to_manifold)
    tr = orc.run_rbcd(dso, X0, num_robots=5, r_min=r, max_iters=iters, staircase=0, rgrad_tol=1e-12)
    s = da.RbcdSession(ds, num_robots=5, r=r)
    s.set_X(X0)
    out = s.run(max_iters=iters, rgrad_tol=1e-12)
    assert np.array_equal(out["selected"], tr["selected"])
    assert np.allclose(out["cost"], tr["cost"], rtol=1e-7)
    assert common.rel(s.get_X(), tr["X"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name,radius,seed", [("smallGrid3D", 100.0, 5), ("smallGrid3D", 1e4, 7), ("tinyGrid3D", 1e4, 7)])
def test_rtr_with_rejected_steps_reuses_z0_and_matches_oracle(sparse_env, name, radius, seed):
    """from a random point several of ten RTR iterations are rejected (ref src/QuadraticOptimizer.cpp:234-280, ROPTLIB's
    acceptance rule rho > 0.1): the iterate and its gradient stay, and the device solver starts the next iteration from
    the kept z0 = P grad instead of another application of the preconditioner -- same outer, tCG
    and accepted counts as the oracle, which applies the preconditioner again, and the same optimum"""
    import dcora_amd as da
    from oracle import orc
    r = 5
    ds, dso = common.product_dataset(name), common.oracle_dataset(name)
    P = da.QuadraticProblem(r, ds.d, ds.n, da.build_Q_pgo(ds), reg=0.1)
    Po = orc.Problem(r, ds.d, ds.n, orc.build_Q_pgo(dso))
    assert P.precond_info()["kind"] == "sparse"
    X0 = common.random_point(r, ds.d, ds.n, seed, orc.project_to_manifold)
    opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=10, RTR_initial_radius=radius, gradnorm_tol=1e-6))
    X = opt.optimize(X0)
    res = opt.getOptResult()
    Xo, reso = Po.optimize(X0, RTR_iterations=10, RTR_initial_radius=radius, gradnorm_tol=1e-6)
    assert reso["accepted"] < reso["outer_iters"]  # the case holds rejected steps
    assert res["outer_iterations"] == reso["outer_iters"], (res, reso)
    assert res["inner_iterations"] == reso["inner_iters"], (res, reso)
    assert res["accepted_steps"] == reso["accepted"], (res, reso)
    assert abs(res["fOpt"] - reso["fOpt"]) <= 1e-9 * abs(reso["fOpt"])
    assert common.rel(X, Xo) < 1e-7
    P.close()
