"""Forms of two hot kernels live side by side (the choice is read once per process): the block Q-apply without LDS
(k_spmm_bsr2, the default) against the LDS-staged one (DCORA_BSR_KERNEL=v1), which must agree BITWISE (same
summation order), the form with 16-byte gathers (k_spmm_bsr3, DCORA_BSR_KERNEL=v3) against both to rounding, and the
entry-per-lane level kernel of the sparse preconditioner (k_sp_level2) against the (entry, value)-per-lane one
(DCORA_SP_KERNEL=v1), which sum in a different order and must agree to rounding.  One child process per form.
The generic layout's Hessian product in one launch (k_spmm_dir_fix) is checked against the two-launch form through the
library's test hook, and the solver loop on top of either form must repeat itself bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import common
import dcora_amd as da
from dcora_amd import synth
out = {}
cases = [("lattice", synth.lattice_se3(9, 8, 7, seed=5)), ("sphere", common.product_dataset("sphere2500")),
         ("grid2d", common.product_dataset("pose_graph_optimization_test_2d"))]
for name, ds in cases:
    Q = da.build_Q_pgo(ds)
    k = (ds.d + 1) * ds.n
    for r in sorted({ds.d, 5, 8}):
        rng = np.random.default_rng(3)
        X = rng.standard_normal((r, k))
        Gm = rng.standard_normal((r, k))
        P = da.QuadraticProblem(r, ds.d, ds.n, Q, G=Gm, reg=0.1)
        out["%s_%d_info" % (name, r)] = np.array([P.qapply_info()["kernel"] == "k_spmm", P.precond_info()["kind"] == "sparse"])
        out["%s_%d_f" % (name, r)] = np.array([P.f(X)])
        out["%s_%d_g" % (name, r)] = P.EucGrad(X)
        Xm = da.manifold_project(r, ds.d, ds.n, X)
        V = P.RieGrad(Xm)
        out["%s_%d_z" % (name, r)] = P.PreCondition(Xm, V)
        P.close()
np.savez(sys.argv[2], **out)
'''


def _run(tmp_path, tag, env):
    e = dict(os.environ)
    e.update({"DCORA_QAPPLY": "bsr", "DCORA_PRECOND": "sparse"})
    e.update(env)
    out = os.path.join(str(tmp_path), tag + ".npz")
    res = subprocess.run([sys.executable, "-c", CHILD, os.path.dirname(common.HERE), out], env=e, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return np.load(out)


def test_both_forms_of_the_block_qapply_and_of_the_level_kernel_agree(built, tmp_path):
    if os.environ.get("DCORA_SOLVER_V1"):
        pytest.skip("DCORA_SOLVER_V1 switches the block Q-apply off")
    new = _run(tmp_path, "new", {})
    old = _run(tmp_path, "old", {"DCORA_BSR_KERNEL": "v1"})
    v3 = _run(tmp_path, "v3", {"DCORA_BSR_KERNEL": "v3"})  # 16-byte gathers, column pairs summed per pose (another order)
    assert set(new.files) == set(old.files) == set(v3.files)
    for key in new.files:
        if key.endswith("_info"):
            assert not new[key][0] and new[key][1], key   # the block Q-apply and the sparse preconditioner ran
        elif key.endswith("_z"):
            assert common.rel(new[key], old[key]) < 1e-12, key
            assert common.rel(v3[key], old[key]) < 1e-12, key
        else:
            assert np.array_equal(new[key], old[key]), key
            assert common.rel(v3[key], old[key]) < 1e-13, key



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


@pytest.mark.parametrize("form", ["one", "two"])
def test_generic_solver_loop_repeats_bitwise(built, tmp_path, form):
    """run to run the generic-layout RTR repeats itself bit for bit in both forms of its Hessian product (tiers has a
    long row: its term of <delta, H delta> must not move around the partial array with the arrival order of the
    workgroups that share the row -- it did in a first version of k_spmm_dir_fix, and the iterates drifted)"""
    e = dict(os.environ)
    if form == "two":
        e["DCORA_HESS_FUSE"] = "0"
    out = os.path.join(str(tmp_path), form + ".npz")
    res = subprocess.run([sys.executable, "-c", RA_CHILD, os.path.dirname(common.HERE), out], env=e,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    o = np.load(out)
    for name in ("tiers", "range_aided_slam_test_3d"):
        for rep in (1, 2):
            assert np.array_equal(o[name + "_X0"], o[name + "_X%d" % rep]), (form, name, rep)
            assert np.array_equal(o[name + "_f0"], o[name + "_f%d" % rep]), (form, name, rep)


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


PACING_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import dcora_amd as da
from dcora_amd import synth
from test_raslam import ra_path
out = {}
ds = synth.lattice_se3(12, 10, 9, seed=3)
r = 5
rng = np.random.default_rng(7)
X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, 4 * ds.n)))
s = da.RbcdSession(ds, num_robots=3, r=r)          # fused path, sparse preconditioner forced by the environment
s.set_X(X0)
o = s.run(max_iters=12, rgrad_tol=0.0)
out["pgo_X"] = s.get_X(); out["pgo_cost"] = o["cost"]; out["pgo_kind"] = np.array([s.precond_kinds()[0] == "sparse"]) if hasattr(s, "precond_kinds") else np.array([True])
s.close()
ra = da.RADataset(ra_path("tiers"))                 # generic path (range-aided layout)
X0 = np.zeros((3, ra.k)); X0[:ra.d] = ra.X_odom
P = da.QuadraticProblem(3, ra.d, ra.n, ra.Q, l=ra.l, b=ra.b)
opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=8, RTR_tCG_iterations=30, gradnorm_tol=1e-12))
out["ra_X"] = opt.optimize(X0)
res = opt.getOptResult()
out["ra_it"] = np.array([res["outer_iterations"], res["inner_iterations"]])
P.close()
np.savez(sys.argv[2], **out)
"""


def test_verdict_pacing_enqueues_the_same_work_as_the_lookahead(built, tmp_path):
    """the solver host enqueues the sparse replay behind the step-length kernel's verdict (default) or two whole tCG
    iterations ahead (DCORA_SP_PACING=lookahead): the launches the first form skips are gated no-ops of the second,
    so iterates and iteration counts must be bit-identical -- fused pose-graph path and generic (range-aided) path"""
    runs = {}
    for tag, env in (("verdict", {}), ("lookahead", {"DCORA_SP_PACING": "lookahead"})):
        e = dict(os.environ)
        e["DCORA_PRECOND"] = "sparse"
        e.update(env)
        out = os.path.join(str(tmp_path), tag + ".npz")
        res = subprocess.run([sys.executable, "-c", PACING_CHILD, os.path.dirname(common.HERE), out], env=e,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        runs[tag] = np.load(out)
    a, b = runs["verdict"], runs["lookahead"]
    for key in ("pgo_X", "pgo_cost", "ra_X", "ra_it"):
        assert np.array_equal(a[key], b[key]), key
    assert a["pgo_cost"][-1] < a["pgo_cost"][0]


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
    by the host's threads streamed in chunks (DCORA_SP_FILL=host) and by the host in one piece (DCORA_SP_WEIGHTS=host):
    same count, same place for every weight (position-weighted sum), values to the rounding of M = D^-T D^-1 computed
    by the device's tile products instead of the host's loops; the preconditioned vectors agree to 1e-12"""
    runs = {}
    for tag, env in (("device", {}), ("streamed", {"DCORA_SP_FILL": "host"}), ("one_piece", {"DCORA_SP_WEIGHTS": "host"})):
        e = dict(os.environ)
        e["DCORA_PRECOND"] = "sparse"
        e.update(env)
        out = os.path.join(str(tmp_path), tag + ".npz")
        res = subprocess.run([sys.executable, "-c", WEIGHTS_CHILD, os.path.dirname(common.HERE), out], env=e,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        runs[tag] = np.load(out)
    for name in ("lattice", "sphere"):
        a, b, c = runs["device"][name], runs["streamed"][name], runs["one_piece"][name]
        assert np.array_equal(b, c), (name, b, c)                      # the two host fills: the same arithmetic
        assert a[0] == b[0] and a[0] > 1000, (name, a, b)
        assert np.allclose(a[1:], b[1:], rtol=1e-11, atol=1e-11 * a[2]), (name, a, b)
        for tag in ("streamed", "one_piece"):
            assert common.rel(runs["device"][name + "_z"], runs[tag][name + "_z"]) < 1e-12, (name, tag)


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
