"""dcora_amd -- MI355X-native implementation of DCORA's per-agent Riemannian local solver, certification and
RBCD loop.  This package is a thin Python view (for tests and bench.py) of the C ABI in include/dcora_hip.h;
the product is libdcora_hip.so (HIP kernels for gfx950 + C++ host code).  Class and method names mirror the
reference's C++ API (QuadraticProblem / QuadraticOptimizer / ROptParameters, ref include/DCORA/*.h).
"""
import ctypes as C
import gzip
import os
import shutil
import tempfile

import numpy as np

from . import capi
from .capi import DcoraError, Dims, ROptParams, ROptResult, RbcdOptions, F, unF, check

__all__ = ["QuadraticProblem", "QuadraticOptimizer", "ROptParameters", "Dataset", "Csr", "RbcdSession", "RaRbcdSession", "DcoraError",
           "build_Q_pgo", "dual_certificate", "is_psd", "min_eig", "fast_verification", "manifold_project",
           "device_count"]


def device_count():
    return capi.lib().dcora_device_count()


class Csr:
    """row-major CSR, int32 indices (ref include/DCORA/DCORA_types.h:36)"""

    def __init__(self, n, rp, ci, v):
        self.n = int(n)
        self.rp = np.ascontiguousarray(rp, dtype=np.int32)
        self.ci = np.ascontiguousarray(ci, dtype=np.int32)
        self.v = np.ascontiguousarray(v, dtype=np.float64)

    @property
    def nnz(self):
        return int(self.rp[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.v, self.ci, self.rp), shape=(self.n, self.n))

    @staticmethod
    def from_scipy(A):
        A = A.tocsr()
        A.sort_indices()
        return Csr(A.shape[0], A.indptr, A.indices, A.data)


class Dataset:
    """measurement list: ids m x 4 (r1,p1,r2,p2), vals m x (d*d+d+3) (R col-major, t, kappa, tau, weight)"""

    def __init__(self, d, n, ids, vals):
        self.d, self.n = int(d), int(n)
        self.ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1, 4)
        self.vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(self.ids.shape[0], self.d * self.d + self.d + 3)

    @property
    def m(self):
        return self.ids.shape[0]

    @staticmethod
    def load_g2o(path):
        """read_g2o_file (ref src/DCORA_utils.cpp:179-375); accepts .g2o and .g2o.gz"""
        L = capi.lib()
        tmp = None
        if str(path).endswith(".gz"):
            fd, tmp = tempfile.mkstemp(suffix=".g2o")
            with os.fdopen(fd, "wb") as out, gzip.open(path, "rb") as src:
                shutil.copyfileobj(src, out)
            path = tmp
        try:
            h = C.c_void_p()
            check(L.dcora_dataset_load_g2o(str(path).encode(), C.byref(h)))
        finally:
            if tmp:
                os.unlink(tmp)
        d, n, m = C.c_int(), C.c_int(), C.c_int()
        check(L.dcora_dataset_info(h, C.byref(d), C.byref(n), C.byref(m)))
        ids = np.zeros((m.value, 4), np.int32)
        vals = np.zeros((m.value, d.value * d.value + d.value + 3), np.float64)
        check(L.dcora_dataset_copy(h, ids, vals))
        L.dcora_dataset_destroy(h)
        return Dataset(d.value, n.value, ids, vals)

    def handle(self):
        h = C.c_void_p()
        check(capi.lib().dcora_dataset_create(self.d, self.n, self.m, self.ids, self.vals, C.byref(h)))
        return h


def chordal_initialization(ds, device=None):
    """chordalInitialization (ref src/DCORA_solver.cpp:218-268); returns T, d x (d+1) n.  device: solve the two SPD
    systems on that GPU (large graphs) instead of on the host"""
    h = ds.handle()
    out = np.zeros(ds.d * (ds.d + 1) * ds.n)
    try:
        if device is None:
            check(capi.lib().dcora_dataset_chordal_init(h, out))
        else:
            check(capi.lib().dcora_dataset_chordal_init_device(h, int(device), out))
    finally:
        capi.lib().dcora_dataset_destroy(h)
    return unF(out, ds.d, (ds.d + 1) * ds.n)


def build_Q_pgo(ds, n=None, agent=0, ids=None, vals=None):
    """Graph::constructQuadraticCostTermPGO (ref src/Graph.cpp:579-683)"""
    ids = ds.ids if ids is None else np.ascontiguousarray(ids, np.int32)
    vals = ds.vals if vals is None else np.ascontiguousarray(vals, np.float64)
    h = C.c_void_p()
    check(capi.lib().dcora_graph_build_Q_pgo(ds.d, ds.n if n is None else n, agent, ids.shape[0], ids, vals,
                                             C.byref(h)))
    return Csr(*capi.take_csr(h))


class RADataset:
    """centralised range-aided SLAM problem read from a .pyfg file (ref src/DCORA_utils.cpp:437-1167, 1169-1365;
    Q: ref src/Graph.cpp:824-1188).  X is r x k in the RA ordering [Y1..Yn | s1..sl | p1..pn | L1..Lb]."""

    def handle(self):
        """a fresh dcora_radataset_t of the file (the caller destroys it)"""
        L = capi.lib()
        path, tmp = self.path, None
        if str(path).endswith(".gz"):
            fd, tmp = tempfile.mkstemp(suffix=".pyfg")
            with os.fdopen(fd, "wb") as out, gzip.open(path, "rb") as src:
                shutil.copyfileobj(src, out)
            path = tmp
        try:
            h = C.c_void_p()
            check(L.dcora_radataset_load_pyfg(str(path).encode(), C.byref(h)))
        finally:
            if tmp:
                os.unlink(tmp)
        return h

    def __init__(self, path, init_seed=20250310):
        L = capi.lib()
        self.path = path
        h = self.handle()
        info = np.zeros(7, np.int32)
        check(L.dcora_radataset_info(h, info))
        self.d, self.n, self.l, self.b = (int(x) for x in info[:4])
        self.num_pose_pose, self.num_pose_landmark, self.num_range = (int(x) for x in info[4:])
        self.k = (self.d + 1) * self.n + self.l + self.b
        gt = np.zeros(self.d * self.k)
        check(L.dcora_radataset_ground_truth(h, gt))
        self.gt = unF(gt, self.d, self.k)
        q = C.c_void_p()
        check(L.dcora_radataset_build_Q(h, C.byref(q)))
        self.Q = Csr(*capi.take_csr(q))
        # start point of the centralised CORA driver (ref examples/SingleRobotExample_RASLAM.cpp:92-150), d x k
        x0 = np.zeros(self.d * self.k)
        check(L.dcora_radataset_odometry_init(h, init_seed, x0))
        self.X_odom = unF(x0, self.d, self.k)
        # ownership of the merged variables and every robot's columns in its own RA ordering
        # (ref src/DCORA_utils.cpp:1370-1512, src/Graph.cpp:584-616, 1092-1097)
        self.pose_robot = np.zeros(max(self.n, 1), np.int32)
        self.sphere_robot = np.zeros(max(self.l, 1), np.int32)
        self.landmark_robot = np.zeros(max(self.b, 1), np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(L.dcora_radataset_ownership(h, vp(self.pose_robot), vp(self.sphere_robot), vp(self.landmark_robot)))
        self.pose_robot, self.sphere_robot = self.pose_robot[:self.n], self.sphere_robot[:self.l]
        self.landmark_robot = self.landmark_robot[:self.b]
        self.robots = sorted(set(self.pose_robot.tolist()))
        self.agent_columns = {}
        for rb in sorted(set(self.robots) | set(self.landmark_robot.tolist())):
            dims3, own, ka = np.zeros(3, np.int32), np.zeros(self.k, np.int32), C.c_int()
            check(L.dcora_radataset_agent_columns(h, rb, dims3, vp(own), C.byref(ka)))
            self.agent_columns[rb] = (tuple(int(x) for x in dims3), own[:ka.value].copy())
        L.dcora_radataset_destroy(h)

    def agent_blocks(self, robot, Q=None):
        """(n_a, l_a, b_a), own columns, Q_aa (Csr, agent ordering), coupling C (scipy k_a x k, global columns):
        the agent's local problem is 1/2 <Q_aa, X_a^T X_a> + <X_a, X_global C^T>"""
        import scipy.sparse as sp
        Q = self.Q if Q is None else Q
        dims3, own = self.agent_columns[robot]
        own = np.ascontiguousarray(own, np.int32)
        qa, cc = C.c_void_p(), C.c_void_p()
        check(capi.lib().dcora_graph_extract_agent_blocks(Q.n, Q.rp, Q.ci, Q.v, own.size, own, C.byref(qa),
                                                          C.byref(cc)))
        Qaa = Csr(*capi.take_csr(qa))
        ka, rp, ci, v = capi.take_csr(cc)
        return dims3, own, Qaa, sp.csr_matrix((v, ci, rp), shape=(ka, Q.n))


def precond_regularization(Q, device=0):
    """Graph::computePreconditionerRegularization (ref src/Graph.cpp:1921-1960)"""
    reg = C.c_double()
    check(capi.lib().dcora_graph_precond_regularization(Q.n, Q.rp, Q.ci, Q.v, device, C.byref(reg)))
    return reg.value


class ROptParameters:
    """ref include/DCORA/DCORA_types.h:152-168"""
    RTR, RGD = 0, 1

    def __init__(self, **kw):
        self.c = ROptParams()
        capi.lib().dcora_ropt_params_default(C.byref(self.c))
        for k, v in kw.items():
            setattr(self.c, k, v)

    def __getattr__(self, k):
        return getattr(self.__dict__["c"], k)


class QuadraticProblem:
    """ref include/DCORA/QuadraticProblem.h: f, RieGrad, RieGradNorm, Retract, PreCondition, escapeSaddle"""

    def __init__(self, r, d, n, Q, G=None, reg=0.1, l=0, b=0, device=0, layout=0):
        """layout: 0 = SE ordering when l = b = 0, RA otherwise; capi.LAYOUT_RA keeps the RA ordering of a range-aided
        graph that holds neither ranges nor landmarks (ref src/Graph.cpp:68-75)"""
        self.r, self.d, self.n, self.l, self.b = r, d, n, l, b
        self.k = (d + 1) * n + l + b
        if Q.n != self.k:
            raise ValueError("Q is %d x %d, expected %d" % (Q.n, Q.n, self.k))
        dims = Dims(r, d, n, l, b, layout)
        g = None
        if G is not None:
            self._G = F(G)
            g = self._G.ctypes.data_as(C.c_void_p)
        self.h = C.c_void_p()
        check(capi.lib().dcora_problem_create(C.byref(dims), Q.rp, Q.ci, Q.v, g, float(reg), device, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            capi.lib().dcora_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def relaxation_rank(self):
        return self.r

    def problem_dimension(self):
        return self.k

    def _out(self):
        return np.zeros(self.r * self.k)

    def set_linear_term(self, G):
        g = None
        if G is not None:
            self._G = F(G)
            g = self._G.ctypes.data_as(C.c_void_p)
        check(capi.lib().dcora_problem_set_linear_term(self.h, g))

    def f(self, Y):
        out = C.c_double()
        check(capi.lib().dcora_problem_cost(self.h, F(Y), C.byref(out)))
        return out.value

    def EucGrad(self, Y):
        out = self._out()
        check(capi.lib().dcora_problem_eucgrad(self.h, F(Y), out))
        return unF(out, self.r, self.k)

    def RieGrad(self, Y):
        out = self._out()
        nrm = C.c_double()
        check(capi.lib().dcora_problem_riegrad(self.h, F(Y), out.ctypes.data_as(C.c_void_p), C.byref(nrm)))
        return unF(out, self.r, self.k)

    def RieGradNorm(self, Y):
        nrm = C.c_double()
        check(capi.lib().dcora_problem_riegrad(self.h, F(Y), None, C.byref(nrm)))
        return nrm.value

    def HessVec(self, Y, V):
        out = self._out()
        check(capi.lib().dcora_problem_hessvec(self.h, F(Y), F(V), out))
        return unF(out, self.r, self.k)

    def HessVecSolverForm(self, Y, V):
        """test hook: H[V] as the generic solver loop forms it (one launch) and {<V, H V> from that kernel's partial
        sums, the same from the two-launch form}"""
        out = self._out()
        dots = np.zeros(2)
        check(capi.lib().dcora_debug_hessvec_solver_form(self.h, F(Y), F(V), out, dots))
        return unF(out, self.r, self.k), dots

    def PreCondition(self, Y, V):
        out = self._out()
        check(capi.lib().dcora_problem_precondition(self.h, F(Y), F(V), out))
        return unF(out, self.r, self.k)

    def Retract(self, Y, V):
        out = self._out()
        check(capi.lib().dcora_problem_retract(self.h, F(Y), F(V), out))
        return unF(out, self.r, self.k)

    def projectToTangentSpace(self, Y, V):
        out = self._out()
        check(capi.lib().dcora_problem_tangent_project(self.h, F(Y), F(V), out))
        return unF(out, self.r, self.k)

    def escapeSaddle(self, Xopt, theta, v, gradient_tolerance=1e-6, preconditioned_gradient_tolerance=1e-6,
                     isSecondOrder=False):
        out = self._out()
        ok = C.c_int()
        check(capi.lib().dcora_problem_escape_saddle(self.h, F(Xopt), float(theta),
                                                     np.ascontiguousarray(v, np.float64), gradient_tolerance,
                                                     preconditioned_gradient_tolerance, int(isSecondOrder), out,
                                                     C.byref(ok)))
        return (unF(out, self.r, self.k) if ok.value else None)

    def time_qapply(self, reps=100):
        ms, by = C.c_double(), C.c_double()
        check(capi.lib().dcora_problem_time_qapply(self.h, reps, C.byref(ms), C.byref(by)))
        return ms.value, by.value


def _time_precond(self, reps=100):
    ms, by = C.c_double(), C.c_double()
    check(capi.lib().dcora_problem_time_precond(self.h, reps, C.byref(ms), C.byref(by)))
    return ms.value, by.value


QuadraticProblem.time_precond = _time_precond


def _precond_info(self):
    """how (Q + reg I)^-1 is held on the device: dense inverse or partitioned sparse inverse"""
    info = np.zeros(5)
    check(capi.lib().dcora_problem_precond_info(self.h, info))
    return {"kind": {0: "none", 1: "dense", 2: "sparse"}[int(info[0])], "launches": int(info[1]),
            "nnzL": int(info[2]), "setup_ms": float(info[3]), "stored_doubles_per_apply": float(info[4])}


QuadraticProblem.precond_info = _precond_info


def _qapply_info(self):
    info = np.zeros(4)
    check(capi.lib().dcora_problem_qapply_info(self.h, info))
    return {"kernel": "k_spmm_bsrq" if info[0] else "k_spmm", "nnz": int(info[1]), "blocks": int(info[2]),
            "stored_matrix_bytes": float(info[3])}


QuadraticProblem.qapply_info = _qapply_info


def _solver_info(self):
    """how the tCG iteration runs: 'three launches', 'two launches' or 'one launch per run' (k_tcg_run)"""
    info = np.zeros(2)
    check(capi.lib().dcora_problem_solver_info(self.h, info))
    return {"tcg": {0: "three launches", 1: "two launches", 2: "one launch per run"}[int(info[0])]}


QuadraticProblem.solver_info = _solver_info


def time_qapply_rotating(problems, reps=60):
    """average launch time of the Q-apply over several problems in turn on one stream (HBM-cold when their bytes
    exceed the Infinity Cache)"""
    arr = (C.c_void_p * len(problems))(*[getattr(p.h, 'value', p.h) for p in problems])
    ms = C.c_double()
    check(capi.lib().dcora_problem_time_qapply_rotating(arr, len(problems), reps, C.byref(ms)))
    return ms.value


def precond_cache_info():
    v = np.zeros(4)
    check(capi.lib().dcora_precond_cache_info(v))
    return {"hits": int(v[0]), "misses": int(v[1]), "entries": int(v[2]), "device_bytes": float(v[3])}


def precond_cache_clear():
    check(capi.lib().dcora_precond_cache_clear())


class QuadraticOptimizer:
    """ref include/DCORA/QuadraticOptimizer.h: optimize(Y), getOptResult()"""

    def __init__(self, problem, params=None):
        self.problem = problem
        self.params = params or ROptParameters()
        self.result = ROptResult()

    def optimize(self, Y):
        out = np.zeros(self.problem.r * self.problem.k)
        check(capi.lib().dcora_optimizer_optimize(self.problem.h, C.byref(self.params.c), F(Y), out,
                                                  C.byref(self.result)))
        return unF(out, self.problem.r, self.problem.k)

    def getOptResult(self):
        return self.result.as_dict()


def manifold_project(r, d, n, M, l=0, b=0, device=0, layout=0):
    """projectToSEMatrix / projectToRAMatrix (ref src/DCORA_utils.cpp:2201-2220)"""
    dims = Dims(r, d, n, l, b, layout)
    k = (d + 1) * n + l + b
    out = np.zeros(r * k)
    check(capi.lib().dcora_manifold_project(C.byref(dims), F(M), out, device))
    return unF(out, r, k)


def dual_certificate(r, d, n, X, Q, l=0, b=0, device=0, layout=0):
    """constructDualCertificateMatrixPGO / RASLAM (ref src/DCORA_utils.cpp:1898-1982)"""
    dims = Dims(r, d, n, l, b, layout)
    h = C.c_void_p()
    check(capi.lib().dcora_cert_dual_matrix(C.byref(dims), F(X), Q.rp, Q.ci, Q.v, device, C.byref(h)))
    return Csr(*capi.take_csr(h))


def is_psd(S, block=1):
    out = C.c_int()
    check(capi.lib().dcora_cert_is_psd(S.n, S.rp, S.ci, S.v, block, C.byref(out)))
    return bool(out.value)


def cert_prepare(Q, d, n, l=0, b=0, block=1, device=0, layout=0):
    """analysis of the PSD test of the dual certificate S = Q - Lambda ahead of time (dcora_cert_prepare): its pattern
    is known from Q's; meant for another host thread while the agents iterate"""
    dims = Dims(1, d, n, l, b, layout)
    check(capi.lib().dcora_cert_prepare(C.byref(dims), Q.rp, Q.ci, block, device))


def is_psd_device(S, block=1, device=0, info=False):
    """the PSD test with the numeric factorisation on the device (dcora_cert_is_psd_device)"""
    out = C.c_int()
    i6 = np.zeros(8)
    check(capi.lib().dcora_cert_is_psd_device(S.n, S.rp, S.ci, S.v, block, device, C.byref(out), i6))
    if info:
        return bool(out.value), {"symbolic_ms": i6[0], "numeric_ms": i6[1], "arena_bytes": i6[2], "flops": i6[3],
                                 "levels": int(i6[4]), "launches": int(i6[5]), "logdet": i6[6],
                                 "lookup_ms": i6[7]}
    return bool(out.value)


def chol_host_selftest(S, block=1):
    """(is_pd, max |P A P^T - L L^T|, info) of the multifrontal schedule executed on the host (validation only)"""
    out, res = C.c_int(), C.c_double()
    i4 = np.zeros(4)
    check(capi.lib().dcora_chol_host_selftest(S.n, S.rp, S.ci, S.v, block, C.byref(out), C.byref(res), i4))
    return bool(out.value), res.value, {"pieces": int(i4[0]), "levels": int(i4[1]), "arena": int(i4[2]),
                                        "flops": i4[3]}


def stream_triad_gbps(n=1 << 28, reps=20, device=0):
    """measured HBM stream figure of the box: a = b + s c over three arrays of n doubles (6.4 GB by default, far
    beyond the 256 MiB Infinity Cache)"""
    L = capi.lib()
    L.dcora_debug_stream_triad.restype = C.c_int
    L.dcora_debug_stream_triad.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    out = C.c_double()
    check(L.dcora_debug_stream_triad(device, n, reps, C.byref(out)))
    return out.value


def chol_cache_clear():
    check(capi.lib().dcora_chol_cache_clear())


def min_eig(S, max_iterations=1000, tol=1e-3, ncv=20, seed=12345, device=0):
    lam, mv = C.c_double(), C.c_long()
    v = np.zeros(S.n)
    st = capi.lib().dcora_cert_min_eig(S.n, S.rp, S.ci, S.v, max_iterations, tol, ncv, seed, device, C.byref(lam), v,
                                       C.byref(mv))
    if st not in (0, 5):
        check(st)
    return st == 0, lam.value, v, mv.value


def fast_verification(S, eta, block=1, device=0):
    psd, th, lm = C.c_int(), C.c_double(), C.c_double()
    v = np.zeros(S.n)
    st = capi.lib().dcora_cert_fast_verification(S.n, S.rp, S.ci, S.v, float(eta), block, device, C.byref(psd),
                                                 C.byref(th), v, C.byref(lm))
    if st not in (0, 5):
        check(st)
    return bool(psd.value), th.value, v, lm.value


def lambda_min_certified(S, eta, block=1, max_iterations=120):
    """certified lower bound of lambda_min(S) after the certificate S + eta I >= 0 was accepted: the Lanczos estimate on
    (S + eta I)^-1 minus its Ritz residual, verified by a Cholesky factorisation of S - bound I (-eta if that fails)"""
    lam, it = C.c_double(), C.c_int()
    check(capi.lib().dcora_cert_lambda_min_certified(S.n, S.rp, S.ci, S.v, eta, block, max_iterations, C.byref(lam),
                                                     C.byref(it)))
    return lam.value, it.value


def suboptimality_gap(r, d, n, X, psd, eta, lambda_min=None, l=0, b=0, lambda_bound=None):
    """bound on f(X) - f* that goes with a certificate (dcora_cert_suboptimality_gap, an addition of this library): after
    fast_verification(S, eta) returned psd (S + eta I >= 0) or, otherwise, its lambda_min output (of S + eta I);
    lambda_bound overrides (e.g. min(lambda_min_certified(S, eta), 0))"""
    lam = lambda_bound if lambda_bound is not None else (-eta if psd else float(lambda_min) - eta)
    gap, neff = C.c_double(), C.c_double()
    dims = Dims(r, d, n, l, b)
    check(capi.lib().dcora_cert_suboptimality_gap(C.byref(dims), F(X), lam, C.byref(gap), C.byref(neff)))
    return gap.value, neff.value


class RbcdSession:
    """Agents + synchronous RBCD++ driver on the device (ref examples/MultiRobotExample.cpp:121-307)"""

    def __init__(self, ds, num_robots=5, r=5, acceleration=True, restart_interval=30, params=None, rank=0,
                 world_size=1, device=0, stream=None):
        self.ds, self.R, self.r = ds, num_robots, r
        self.k = (ds.d + 1) * ds.n
        o = RbcdOptions()
        capi.lib().dcora_rbcd_options_default(C.byref(o))
        o.num_robots, o.r, o.acceleration, o.restart_interval = num_robots, r, int(acceleration), restart_interval
        if params is not None:
            o.local = params.c
        o.rank, o.world_size, o.device = rank, world_size, device
        o.stream = stream  # raw hipStream_t (int) or None
        dsh = ds.handle()
        self.h = C.c_void_p()
        try:
            check(capi.lib().dcora_rbcd_create(dsh, C.byref(o), C.byref(self.h)))
        finally:
            capi.lib().dcora_dataset_destroy(dsh)

    def close(self):
        if getattr(self, "h", None):
            capi.lib().dcora_rbcd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_X(self, X):
        check(capi.lib().dcora_rbcd_set_X(self.h, F(X)))

    def get_X(self):
        out = np.zeros(self.r * self.k)
        check(capi.lib().dcora_rbcd_get_X(self.h, out))
        return unF(out, self.r, self.k)

    def iterate(self, selected):
        c2, gn, nxt = C.c_double(), C.c_double(), C.c_int()
        bn = np.zeros(self.R)
        check(capi.lib().dcora_rbcd_iterate(self.h, selected, C.byref(c2), C.byref(gn),
                                            bn.ctypes.data_as(C.c_void_p), C.byref(nxt)))
        return c2.value, gn.value, bn, nxt.value

    def iterate_set(self, agents, allow_adjacent=False):
        """the agents of the set update at the same time from one snapshot of their neighbours' states"""
        a = np.ascontiguousarray(agents, dtype=np.int32)
        check(capi.lib().dcora_rbcd_iterate_set(self.h, a, a.size, int(allow_adjacent)))

    def set_acceleration(self, on):
        check(capi.lib().dcora_rbcd_set_acceleration(self.h, int(bool(on))))

    def colours(self):
        col, nc = np.zeros(self.R, np.int32), C.c_int()
        check(capi.lib().dcora_rbcd_agent_colours(self.h, col, C.byref(nc)))
        return col, nc.value

    def evaluate(self):
        c2, gn, nxt = C.c_double(), C.c_double(), C.c_int()
        bn = np.zeros(self.R)
        check(capi.lib().dcora_rbcd_evaluate(self.h, C.byref(c2), C.byref(gn), bn.ctypes.data_as(C.c_void_p),
                                             C.byref(nxt)))
        return c2.value, gn.value, bn, nxt.value

    def run(self, max_iters=1000, rgrad_tol=0.1):
        it = C.c_int()
        cost, gn = np.zeros(max_iters), np.zeros(max_iters)
        sel = np.zeros(max_iters, np.int32)
        check(capi.lib().dcora_rbcd_run(self.h, max_iters, rgrad_tol, C.byref(it), cost.ctypes.data_as(C.c_void_p),
                                        gn.ctypes.data_as(C.c_void_p), sel.ctypes.data_as(C.c_void_p)))
        n = it.value
        return dict(iters=n, cost=cost[:n], gradnorm=gn[:n], selected=sel[:n])

    def last_result(self):
        r = ROptResult()
        check(capi.lib().dcora_rbcd_last_result(self.h, C.byref(r)))
        return r.as_dict()

    # ---- multi-process pieces ----
    def X_device_ptr(self):
        p = C.c_void_p()
        check(capi.lib().dcora_rbcd_X_device_ptr(self.h, C.byref(p)))
        return p.value

    def public_count(self, agent):
        c = C.c_int()
        check(capi.lib().dcora_rbcd_public_count(self.h, agent, C.byref(c)))
        return c.value

    def public_indices(self, agent):
        idx = np.zeros(max(self.public_count(agent), 1), np.int32)
        check(capi.lib().dcora_rbcd_public_indices(self.h, agent, idx))
        return idx[:self.public_count(agent)]

    def pack_public_dev(self, agent, ptr):
        check(capi.lib().dcora_rbcd_pack_public_dev(self.h, agent, C.c_void_p(ptr)))

    def unpack_public_dev(self, agent, ptr):
        check(capi.lib().dcora_rbcd_unpack_public_dev(self.h, agent, C.c_void_p(ptr)))

    def phase_nonselected(self, selected):
        check(capi.lib().dcora_rbcd_phase_nonselected(self.h, selected))

    def phase_selected(self, selected):
        check(capi.lib().dcora_rbcd_phase_selected(self.h, selected))

    def phase_evaluate_dev(self, ptr):
        check(capi.lib().dcora_rbcd_phase_evaluate_dev(self.h, C.c_void_p(ptr)))

    def synchronize(self):
        check(capi.lib().dcora_rbcd_synchronize(self.h))

    def profile_tcg_runs(self, enable=True):
        """HIP events around every one-launch tCG run (k_tcg_run) of the agents while enabled (a measurement hook)"""
        check(capi.lib().dcora_rbcd_profile_tcg_runs(self.h, int(enable)))

    def profile_tcg_read(self):
        out = np.zeros(2)
        check(capi.lib().dcora_rbcd_profile_tcg_read(self.h, out))
        return {"launches": int(out[0]), "total_us": float(out[1])}


class Exchange:
    """neighbour exchange of public poses between the ranks of one node (dcora_exchange_*): the session must have been
    created with rank / world_size; every rank creates the exchange under the same job name"""
    IPC, STAGED = 1, 2

    def __init__(self, session, job_name):
        self.s = session
        self.h = C.c_void_p()
        create = capi.lib().dcora_exchange_create_ra if isinstance(session, RaRbcdSession) else capi.lib().dcora_exchange_create
        check(create(session.h, job_name.encode(), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            capi.lib().dcora_exchange_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        v = np.zeros(10)
        check(capi.lib().dcora_exchange_info(self.h, v))
        return dict(transport={1: "ipc peer stores", 2: "shared host segment"}.get(int(v[0]), "?"), mode=int(v[0]),
                    halo_finegrained=bool(v[8]), wait="device (the scatter kernel polls the flag)" if v[9] else "host spin",
                    peers=int(v[1]), posts=int(v[2]), waits=int(v[3]), bytes_posted=float(v[4]), post_s=float(v[5]),
                    wait_s=float(v[6]), eval_wait_s=float(v[7]))

    def link_report(self):
        """what the start-up link check of dcora_exchange_create did: rounds run, whether the device-side wait / the IPC
        transport were given up, microseconds of the last round"""
        v = np.zeros(4)
        check(capi.lib().dcora_exchange_link_report(self.h, v))
        return dict(rounds=int(v[0]), gave_up_device_wait=bool(v[1]), gave_up_ipc=bool(v[2]), last_round_us=float(v[3]))

    def all_ready(self, ready):
        """AND over the ranks of `ready` (Agent::shouldTerminate's team condition, ref src/Agent.cpp:1137-1153)"""
        out = C.c_int()
        check(capi.lib().dcora_exchange_all_ready(self.h, int(bool(ready)), C.byref(out)))
        return bool(out.value)

    def certify(self, Q, eta, k):
        """fastVerification of the current iterate across the ranks (dcora_exchange_certify); Q: the global Csr on rank 0,
        None elsewhere; k = (d + 1) n.  -> (certified, theta, lambda_min of S + eta I, v, matvecs, distributed)"""
        cert, dist = C.c_int(), C.c_int()
        th, lm, mv = C.c_double(), C.c_double(), C.c_longlong()
        v = np.zeros(k)
        if Q is None:
            check(capi.lib().dcora_exchange_certify(self.h, k, None, None, None, eta, C.byref(cert), C.byref(th),
                                                    C.byref(lm), v, C.byref(mv), C.byref(dist)))
        else:
            check(capi.lib().dcora_exchange_certify(self.h, k, Q.rp.ctypes.data, Q.ci.ctypes.data, Q.v.ctypes.data, eta,
                                                    C.byref(cert), C.byref(th), C.byref(lm), v, C.byref(mv),
                                                    C.byref(dist)))
        return bool(cert.value), th.value, lm.value, v, mv.value, bool(dist.value)

    def post(self, agents):
        a = np.ascontiguousarray(agents, dtype=np.int32)
        check(capi.lib().dcora_exchange_post(self.h, a, a.size))

    def wait(self, agents):
        a = np.ascontiguousarray(agents, dtype=np.int32)
        check(capi.lib().dcora_exchange_wait(self.h, a, a.size))

    def _eval(self, fn, *first):
        c2, gn, nxt = C.c_double(), C.c_double(), C.c_int()
        bn = np.zeros(self.s.R)
        check(fn(self.h, *first, C.byref(c2), C.byref(gn), bn.ctypes.data_as(C.c_void_p), C.byref(nxt)))
        return c2.value, gn.value, bn, nxt.value

    def evaluate(self):
        return self._eval(capi.lib().dcora_exchange_evaluate)

    def iterate(self, selected):
        return self._eval(capi.lib().dcora_exchange_rbcd_iterate, int(selected))

    def tick(self, agents, allow_adjacent=False):
        a = np.ascontiguousarray(agents, dtype=np.int32)
        check(capi.lib().dcora_exchange_rbcd_tick(self.h, a, a.size, int(allow_adjacent)))

    def set_X(self, X):
        check(capi.lib().dcora_exchange_set_X(self.h, F(X)))

    def gather_X(self):
        out = np.zeros(self.s.r * self.s.k)
        check(capi.lib().dcora_exchange_gather_X(self.h, out))
        return unF(out, self.s.r, self.s.k)

    def barrier(self):
        check(capi.lib().dcora_exchange_barrier(self.h))


class RaRbcdSession:
    """the agents of a multi-robot range-aided SLAM problem + the synchronous RBCD++ driver on the device
    (ref examples/MultiRobotExample_RASLAM.cpp); X is the merged r x k matrix in the RA ordering"""

    def __init__(self, ra, r, acceleration=True, restart_interval=30, params=None, device=0, rank=0, world_size=1):
        self.ra, self.r, self.k = ra, r, ra.k
        o = RbcdOptions()
        capi.lib().dcora_rbcd_options_default(C.byref(o))
        o.r, o.acceleration, o.restart_interval, o.device = r, int(acceleration), restart_interval, device
        o.rank, o.world_size = rank, world_size
        if params is not None:
            o.local = params.c
        dsh = ra.handle()
        self.h = C.c_void_p()
        try:
            check(capi.lib().dcora_ra_rbcd_create(dsh, C.byref(o), C.byref(self.h)))
        finally:
            capi.lib().dcora_radataset_destroy(dsh)
        n = C.c_int()
        check(capi.lib().dcora_ra_rbcd_info(self.h, C.byref(n), None))
        self.R = n.value
        rb = np.zeros(self.R, np.int32)
        check(capi.lib().dcora_ra_rbcd_info(self.h, C.byref(n), rb.ctypes.data_as(C.c_void_p)))
        self.robots = rb.tolist()

    def close(self):
        if getattr(self, "h", None):
            capi.lib().dcora_ra_rbcd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_X(self, X):
        check(capi.lib().dcora_ra_rbcd_set_X(self.h, F(X)))

    def get_X(self):
        out = np.zeros(self.r * self.k)
        check(capi.lib().dcora_ra_rbcd_get_X(self.h, out))
        return unF(out, self.r, self.k)

    def _eval(self, fn, *first):
        c2, gn, nxt = C.c_double(), C.c_double(), C.c_int()
        bn = np.zeros(self.R)
        check(fn(self.h, *first, C.byref(c2), C.byref(gn), bn.ctypes.data_as(C.c_void_p), C.byref(nxt)))
        return c2.value, gn.value, bn, nxt.value

    def iterate(self, selected):
        return self._eval(capi.lib().dcora_ra_rbcd_iterate, selected)

    def evaluate(self):
        return self._eval(capi.lib().dcora_ra_rbcd_evaluate)

    def run(self, max_iters=1000, rgrad_tol=0.1):
        it = C.c_int()
        cost, gn = np.zeros(max_iters), np.zeros(max_iters)
        sel = np.zeros(max_iters, np.int32)
        check(capi.lib().dcora_ra_rbcd_run(self.h, max_iters, rgrad_tol, C.byref(it),
                                           cost.ctypes.data_as(C.c_void_p), gn.ctypes.data_as(C.c_void_p),
                                           sel.ctypes.data_as(C.c_void_p)))
        n = it.value
        return dict(iters=n, cost=cost[:n], gradnorm=gn[:n], selected=sel[:n])

    def last_result(self):
        r = ROptResult()
        check(capi.lib().dcora_ra_rbcd_last_result(self.h, C.byref(r)))
        return r.as_dict()


def align_lifted_trajectory_to_frame(X, anchor, d, n, global_alignment=True, device=0):
    """alignLiftedTrajectoryToFrame (ref src/DCORA_utils.cpp:2262-2289): X r x (d+1) n (SE ordering), anchor
    r x (d+1) -> d x (d+1) n with rotation blocks in SO(d)"""
    X = np.asarray(X, dtype=np.float64)
    dims = Dims(X.shape[0], d, n, 0, 0)
    out = np.zeros(d * (d + 1) * n)
    a = None if anchor is None else F(anchor)
    check(capi.lib().dcora_round_align_trajectory(C.byref(dims), F(X), None if a is None else a.ctypes.data_as(C.c_void_p),
                                                  int(global_alignment), out, None, None, device))
    return unF(out, d, (d + 1) * n)


def ra_states_in_local_frame(X, r, d, n, l, b, device=0):
    """Agent::getStatesInLocalFrame (ref src/Agent.cpp:950-1003): X r x k (RA ordering) -> (trajectory d x (d+1) n in
    the SE ordering, unit spheres d x l, landmarks d x b), all in the frame of pose 0"""
    dims = Dims(r, d, n, l, b, capi.LAYOUT_RA)
    T, S, Lm = np.zeros(d * (d + 1) * n), np.zeros(max(d * l, 1)), np.zeros(max(d * b, 1))
    check(capi.lib().dcora_round_align_trajectory(C.byref(dims), F(X), None, 0, T, S.ctypes.data_as(C.c_void_p),
                                                  Lm.ctypes.data_as(C.c_void_p), device))
    return unF(T, d, (d + 1) * n), unF(S[:d * l], d, l), unF(Lm[:d * b], d, b)


def project_solution_raslam(X, r, d, n, l, b, device=0):
    """projectSolutionRASLAM (ref src/DCORA_utils.cpp:1984-2031): r x k -> d x k"""
    dims = Dims(r, d, n, l, b, capi.LAYOUT_RA)
    k = (d + 1) * n + l + b
    out = np.zeros(d * k)
    check(capi.lib().dcora_round_project_solution_raslam(C.byref(dims), F(X), out, device))
    return unF(out, d, k)


def log_trajectory(path, T, d, n):
    """Logger::logTrajectory (ref src/Logger.cpp:107-145): T is d x (d+1) n"""
    check(capi.lib().dcora_log_trajectory(str(path).encode(), d, n, F(np.asarray(T, dtype=np.float64))))


def _agent_iterate(self, agent, do_optimization=True):
    """Agent::iterate(doOptimization) of one agent (agents advance in lockstep)"""
    check(capi.lib().dcora_rbcd_agent_iterate(self.h, agent, int(bool(do_optimization))))


def _agent_info(self, agent):
    npz, first, it = C.c_int(), C.c_int(), C.c_int()
    check(capi.lib().dcora_rbcd_agent_info(self.h, agent, C.byref(npz), C.byref(first), C.byref(it)))
    return dict(num_poses=npz.value, first_pose=first.value, iteration_number=it.value)


def _agent_get_X(self, agent):
    cols = (self.ds.d + 1) * _agent_info(self, agent)["num_poses"]
    out = np.zeros(self.r * cols)
    check(capi.lib().dcora_rbcd_agent_get_X(self.h, agent, out))
    return unF(out, self.r, cols)


def _agent_set_X(self, agent, X):
    check(capi.lib().dcora_rbcd_agent_set_X(self.h, agent, F(X)))


RbcdSession.agent_iterate = _agent_iterate
RbcdSession.agent_get_X = _agent_get_X
RbcdSession.agent_set_X = _agent_set_X
RbcdSession.agent_info = _agent_info
