"""Two live forms of one computation, side by side (the choice is read once per process, so one child process per form):
the generic layout's Hessian product in one launch (k_spmm_dir_fix) against the two-launch form through the library's
test hook, the solver loop on top of it repeating itself bit for bit, the stored weights of the sparse preconditioner
formed on the device against the host fill (DCORA_SP_FILL=host), and the certificate's Lanczos cycles kept on the device
against the per-step form (DCORA_LANCZOS=sync)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

RA_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import dcora_amd as da
from test_raslam import ra_path
out = {}
for name, r in (("tiers", 3), ("range_aided_slam_test_3d", 4)):
    ra = da.RADataset(ra_path(name))
    if name == "tiers":
        X0 = np.zeros((r, ra.k)); X0[:ra.d] = ra.X_odom
    else:
        rng = np.random.default_rng(5)
        lift = np.linalg.qr(rng.standard_normal((r, ra.d)))[0]
        X0 = da.manifold_project(r, ra.d, ra.n, lift @ ra.gt + 0.05 * rng.standard_normal((r, ra.k)), l=ra.l, b=ra.b)
    for rep in range(3):
        P = da.QuadraticProblem(r, ra.d, ra.n, ra.Q, l=ra.l, b=ra.b)
        opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=6, RTR_tCG_iterations=40, gradnorm_tol=1e-12))
        out["%s_X%d" % (name, rep)] = opt.optimize(X0)
        res = opt.getOptResult()
        out["%s_f%d" % (name, rep)] = np.array([res["fOpt"], res["gradNormOpt"], res["inner_iterations"]])
        P.close()
np.savez(sys.argv[2], **out)
"""


def test_generic_solver_loop_repeats_bitwise(built, tmp_path):
    """run to run the generic-layout RTR repeats itself bit for bit (tiers has a long row: its term of
    <delta, H delta> must not move around the partial array with the arrival order of the workgroups that share the
    row -- it did in a first version of k_spmm_dir_fix, and the iterates drifted)"""
    e = dict(os.environ)
    out = os.path.join(str(tmp_path), "loop.npz")
    res = subprocess.run([sys.executable, "-c", RA_CHILD, os.path.dirname(common.HERE), out], env=e,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    o = np.load(out)
    for name in ("tiers", "range_aided_slam_test_3d"):
        for rep in (1, 2):
            assert np.array_equal(o[name + "_X0"], o[name + "_X%d" % rep]), (name, rep)
            assert np.array_equal(o[name + "_f0"], o[name + "_f%d" % rep]), (name, rep)


def _ra_case(name, r):
    import dcora_amd as da
    from test_raslam import ra_path
    ra = da.RADataset(ra_path(name))
    rng = np.random.default_rng(11)
    X = da.manifold_project(r, ra.d, ra.n, rng.standard_normal((r, ra.k)), l=ra.l, b=ra.b)
    return da.QuadraticProblem(r, ra.d, ra.n, ra.Q, l=ra.l, b=ra.b), X, rng.standard_normal((r, ra.k))


def _pgo_case(name, r):
    import dcora_amd as da
    ds = common.product_dataset(name)
    rng = np.random.default_rng(12)
    k = (ds.d + 1) * ds.n
    X = da.manifold_project(r, ds.d, ds.n, rng.standard_normal((r, k)))
    return da.QuadraticProblem(r, ds.d, ds.n, da.build_Q_pgo(ds), G=rng.standard_normal((r, k)), reg=0.1), X, \
        rng.standard_normal((r, k))


@pytest.mark.parametrize("kind,name,r", [
    ("ra", "tiers", 2), ("ra", "tiers", 3), ("ra", "tiers", 7),          # d = 2, a landmark every pose ranges to
    ("ra", "range_aided_slam_test_3d", 3), ("ra", "range_aided_slam_test_3d", 4), ("ra", "range_aided_slam_test_3d", 9),
    ("ra", "range_aided_slam_test_2d", 2), ("ra", "range_aided_slam_test_2d", 5),
    ("pgo", "sphere2500", 3), ("pgo", "sphere2500", 5), ("pgo", "sphere2500", 16),   # pose layout, d = 3
    ("pgo", "pose_graph_optimization_test_2d", 2), ("pgo", "pose_graph_optimization_test_2d", 6),
])
def test_hessian_in_one_launch_against_the_two_launch_form(built, kind, name, r):
    """k_spmm_dir_fix (delta Q, EucHvToHv and the partial sums of <delta, H delta> in one launch -- the Hessian product
    of the generic-layout tCG) against k_spmm + k_hessfix on the same point and direction: the same terms in the same
    order per entry (rounding of a different FMA contraction at most), the inner product to the rounding of another
    order of the partial sums"""
    P, X, V = (_ra_case if kind == "ra" else _pgo_case)(name, r)
    ref = P.HessVec(X, V)
    one, dots = P.HessVecSolverForm(X, V)
    P.close()
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(one - ref)) <= 1e-13 * scale, np.max(np.abs(one - ref)) / scale
    exact = float(np.sum(V * ref))
    mag = float(np.sum(np.abs(V * ref)))
    assert abs(dots[0] - exact) <= 1e-13 * mag and abs(dots[1] - exact) <= 1e-13 * mag, (dots, exact)


WEIGHTS_CHILD = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import common
import dcora_amd as da
from dcora_amd import capi, synth
L = C.CDLL(capi.LIB_PATH)
out = {}
for name, ds in (("lattice", synth.lattice_se3(14, 12, 10, seed=9)), ("sphere", common.product_dataset("sphere2500"))):
    Q = da.build_Q_pgo(ds)
    P = da.QuadraticProblem(5, ds.d, ds.n, Q, reg=0.1)
    d = np.zeros(4)
    assert L.dcora_debug_sparse_weights_digest(P.h, d.ctypes.data_as(C.c_void_p)) == 0
    out[name] = d
    rng = np.random.default_rng(2)
    X = da.manifold_project(5, ds.d, ds.n, rng.standard_normal((5, (ds.d + 1) * ds.n)))
    out[name + "_z"] = P.PreCondition(X, P.RieGrad(X))
    P.close()
np.savez(sys.argv[2], **out)
"""


def test_weights_formed_on_the_device_equal_the_weights_formed_on_the_host(built, tmp_path):
    """the stored weights of the sparse preconditioner formed by k_fill_weights from device-resident sources (default),
    and by the host's threads streamed in chunks (DCORA_SP_FILL=host): same count, same place for every weight (position-weighted sum), values to the rounding of M = D^-T D^-1 computed
    by the device's tile products instead of the host's loops; the preconditioned vectors agree to 1e-12"""
    runs = {}
    for tag, env in (("device", {}), ("streamed", {"DCORA_SP_FILL": "host"})):
        e = dict(os.environ)
        e["DCORA_PRECOND"] = "sparse"
        e.update(env)
        out = os.path.join(str(tmp_path), tag + ".npz")
        res = subprocess.run([sys.executable, "-c", WEIGHTS_CHILD, os.path.dirname(common.HERE), out], env=e,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        runs[tag] = np.load(out)
    for name in ("lattice", "sphere"):
        a, b = runs["device"][name], runs["streamed"][name]
        assert a[0] == b[0] and a[0] > 1000, (name, a, b)
        assert np.allclose(a[1:], b[1:], rtol=1e-11, atol=1e-11 * a[2]), (name, a, b)
        assert common.rel(runs["device"][name + "_z"], runs["streamed"][name + "_z"]) < 1e-12, name


LANCZOS_CHILD = r"""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import common
import dcora_amd as da
out = {}
ds = common.product_dataset("sphere2500")
Q = da.build_Q_pgo(ds)
rng = np.random.default_rng(4)
X = da.manifold_project(5, ds.d, ds.n, rng.standard_normal((5, 4 * ds.n)))
S = da.dual_certificate(5, ds.d, ds.n, X, Q)                      # far from a critical point: indefinite
ok, lam, v, mv = da.min_eig(S, tol=1e-6)
out["sphere"] = np.array([ok, lam, mv]); out["sphere_v"] = v
n = 400                                                          # a matrix the Krylov space dies on at once
Z = da.Csr.from_scipy(sp.csr_matrix((np.zeros(n), (np.arange(n), np.arange(n))), shape=(n, n)))
ok, lam, v, mv = da.min_eig(Z, tol=1e-6)
out["zero"] = np.array([ok, lam, mv]); out["zero_v"] = v
I2 = da.Csr.from_scipy((2.0 * sp.identity(n)).tocsr())           # every vector an eigenvector: the remainder of the
ok, lam, v, mv = da.min_eig(I2, tol=1e-6)                        # first step is rounding noise, not a direction
out["two_i"] = np.array([ok, lam, mv])
D = da.Csr.from_scipy(sp.diags(np.r_[-2.0, np.ones(200), 3.0 * np.ones(199)]).tocsr())   # three distinct eigenvalues
ok, lam, v, mv = da.min_eig(D, tol=1e-8)
out["diag"] = np.array([ok, lam, mv]); out["diag_v"] = v
np.savez(sys.argv[2], **out)
"""


def test_lanczos_cycles_on_the_device_against_the_per_step_form(built, tmp_path):
    """the certificate's eigensolver keeps a restart cycle's coefficients on the device (default) or reads them back
    after every step (DCORA_LANCZOS=sync, the form used across ranks): same eigenvalue, same eigenvector up to sign on
    an indefinite certificate of sphere2500; a zero matrix (the next vector's norm vanishes at the first step: the fast
    form flags it and redoes the step on the slow path), 2 I (the remainder is rounding noise: a relative test must call
    the Krylov space exhausted -- with the absolute 1e-300 of rounds 1-2 the noise was normalised into the basis and
    the run ended at -1.4e8) and a matrix with three distinct eigenvalues end where they must"""
    runs = {}
    for tag, env in (("device", {}), ("sync", {"DCORA_LANCZOS": "sync"})):
        e = dict(os.environ)
        e.update(env)
        out = os.path.join(str(tmp_path), tag + ".npz")
        res = subprocess.run([sys.executable, "-c", LANCZOS_CHILD, os.path.dirname(common.HERE), out], env=e,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        runs[tag] = np.load(out)
    a, b = runs["device"], runs["sync"]
    assert a["sphere"][0] == 1 and b["sphere"][0] == 1 and a["sphere"][1] < 0
    assert abs(a["sphere"][1] - b["sphere"][1]) <= 1e-8 * abs(b["sphere"][1]), (a["sphere"], b["sphere"])
    assert abs(abs(float(a["sphere_v"] @ b["sphere_v"])) - 1.0) < 1e-6
    for r_ in (a, b):
        assert r_["zero"][0] == 1 and abs(r_["zero"][1]) < 1e-12, r_["zero"]
        assert abs(np.linalg.norm(r_["zero_v"]) - 1.0) < 1e-9
        assert r_["two_i"][0] == 1 and abs(r_["two_i"][1] - 2.0) < 1e-12 and r_["two_i"][2] < 200, r_["two_i"]
        assert r_["diag"][0] == 1 and abs(r_["diag"][1] + 2.0) < 1e-7, r_["diag"]
        assert abs(abs(r_["diag_v"][0]) - 1.0) < 1e-6


def _block_problem(da, name, R, b, r, seed=3):
    """the local problem of agent b of a contiguous R-way split at a random point (with its linear term)"""
    import bench
    ds = common.product_dataset(name)
    nb, ids, vals = bench.agent_block(ds, R, b)
    rng = np.random.default_rng(seed)
    k = (ds.d + 1) * nb
    G = 0.3 * rng.standard_normal((r, k))
    X = da.manifold_project(r, ds.d, nb, rng.uniform(-1, 1, (r, k)))
    return ds.d, nb, da.build_Q_pgo(ds, n=nb, agent=b, ids=ids, vals=vals), G, X


@pytest.mark.parametrize("name,R,r", [("sphere2500", 5, 5), ("sphere2500", 5, 4), ("sphere2500", 5, 6),
                                      ("sphere2500", 8, 5), ("smallGrid3D", 1, 5), ("torus3D", 10, 5)])
def test_one_launch_tcg_run_is_bitwise_the_launches(built, name, R, r):
    """The dense tCG run as ONE launch (k_tcg_run: PC-first, [A, PC] per iteration and the retraction inside one kernel,
    grid-wide steps instead of kernel boundaries) against the launches per iteration (DCORA_SOLVER_TCG=launch): the same
    arithmetic term for term, so the same iterates BIT FOR BIT, the same iteration counts and exit reasons -- with the
    default parameters (runs ended by the trust region), with long runs (residual rule, iteration cap) and at the
    first / last agent of a split (a last workgroup of one pose where n is odd)."""
    import dcora_amd as da
    for b in sorted({0, R - 1}):
        d, nb, Q, G, X = _block_problem(da, name, R, b, r)
        for prm in (da.ROptParameters(), da.ROptParameters(RTR_iterations=8, RTR_tCG_iterations=60, gradnorm_tol=1e-9),
                    da.ROptParameters(RTR_iterations=4, RTR_tCG_iterations=3)):
            got = {}
            for form in ("launch", None):
                if form:
                    os.environ["DCORA_SOLVER_TCG"] = form
                try:
                    P = da.QuadraticProblem(r, d, nb, Q, G=G)
                finally:
                    os.environ.pop("DCORA_SOLVER_TCG", None)
                want = "two launches" if form else "one launch per run"
                assert P.solver_info()["tcg"] == want, (name, R, r, b, P.solver_info())
                opt = da.QuadraticOptimizer(P, prm)
                Xs = opt.optimize(X)
                res = opt.getOptResult()
                got[form] = (Xs, res)
                assert P.solver_info()["tcg"] == want   # (the run form did not give up)
                P.close()
            (Xa, ra), (Xb, rb) = got["launch"], got[None]
            for key in ("outer_iterations", "inner_iterations", "fOpt", "gradNormOpt", "fInit", "tCGStatus"):
                assert ra[key] == rb[key], (name, R, r, b, key, ra[key], rb[key])
            assert ra["inner_iterations"] > 0
            assert np.array_equal(Xa, Xb), np.abs(Xa - Xb).max()


def test_one_launch_tcg_run_that_gives_up_falls_back_to_the_launches(built):
    """a run whose grid is not co-resident (test hook: workgroup 0 leaves before the first grid step) gives up within
    milliseconds, everything queued behind it is a no-op, the host repeats the RTR iteration on the launches and drops
    the form for this problem: same result bit for bit, whichever iteration of the solve the fault hits"""
    import dcora_amd as da
    from dcora_amd import capi
    d, nb, Q, G, X = _block_problem(da, "sphere2500", 5, 2, 5)
    prm = da.ROptParameters(RTR_iterations=6, RTR_tCG_iterations=40, gradnorm_tol=1e-9)
    os.environ["DCORA_SOLVER_TCG"] = "launch"
    try:
        P = da.QuadraticProblem(5, d, nb, Q, G=G)
    finally:
        os.environ.pop("DCORA_SOLVER_TCG", None)
    opt = da.QuadraticOptimizer(P, prm)
    Xref, ref = opt.optimize(X), opt.getOptResult()
    P.close()
    assert ref["outer_iterations"] >= 4
    for healthy_runs in (0, 1, 3):
        P = da.QuadraticProblem(5, d, nb, Q, G=G)
        assert P.solver_info()["tcg"] == "one launch per run"
        if healthy_runs:  # let that many runs of the solve pass first: a warm-up solve of `healthy_runs` RTR iterations
            warm = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=healthy_runs, RTR_tCG_iterations=40,
                                                               gradnorm_tol=1e-9))
            warm.optimize(X)
        assert capi.lib().dcora_debug_tcg_run_fault(1) == 0
        opt = da.QuadraticOptimizer(P, prm)
        Xs, res = opt.optimize(X), opt.getOptResult()
        assert capi.lib().dcora_debug_tcg_run_fault(0) == 0
        assert P.solver_info()["tcg"] == "two launches"   # dropped for good
        for key in ("outer_iterations", "inner_iterations", "fOpt", "gradNormOpt", "tCGStatus"):
            assert ref[key] == res[key], (healthy_runs, key, ref[key], res[key])
        assert np.array_equal(Xref, Xs)
        Xs2 = da.QuadraticOptimizer(P, prm).optimize(X)   # and the problem keeps working on the launches
        assert np.array_equal(Xref, Xs2)
        P.close()
    # The LAST iteration of a solve: the host has enqueued everything and would leave; it must stay until that run has
    # ended, or the iteration is lost without a trace (round 5: seen as a flaky bit difference between ranks sharing a
    # GPU, where other ranks' waiting kernels keep the grid from being co-resident).  RTR_iterations = 3 as in RBCD.
    prm3 = da.ROptParameters(RTR_iterations=3, RTR_tCG_iterations=40, gradnorm_tol=1e-9)
    os.environ["DCORA_SOLVER_TCG"] = "launch"
    try:
        P = da.QuadraticProblem(5, d, nb, Q, G=G)
    finally:
        os.environ.pop("DCORA_SOLVER_TCG", None)
    opt = da.QuadraticOptimizer(P, prm3)
    Xref3, ref3 = opt.optimize(X), opt.getOptResult()
    P.close()
    assert ref3["outer_iterations"] == 3
    for skip in (2, 1):
        P = da.QuadraticProblem(5, d, nb, Q, G=G)
        assert capi.lib().dcora_debug_tcg_run_fault_at(skip, 1) == 0
        opt = da.QuadraticOptimizer(P, prm3)
        Xs, res = opt.optimize(X), opt.getOptResult()
        assert capi.lib().dcora_debug_tcg_run_fault(0) == 0
        assert P.solver_info()["tcg"] == "two launches"
        for key in ("outer_iterations", "inner_iterations", "fOpt", "gradNormOpt", "tCGStatus"):
            assert ref3[key] == res[key], (skip, key, ref3[key], res[key])
        assert np.array_equal(Xref3, Xs), skip
        P.close()
