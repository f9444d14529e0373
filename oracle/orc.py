"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/liboracle.so (the CPU restatement of the DCORA
hot path, see oracle/oracle.hpp).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product
(dcora_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "-j8"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_g2o_load.restype = C.c_void_p
        L.orc_g2o_load.argtypes = [C.c_char_p]
        L.orc_ds_create.restype = C.c_void_p
        L.orc_ds_create.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _dp]
        L.orc_ds_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        L.orc_ds_copy.argtypes = [C.c_void_p, _ip, _dp]
        L.orc_ds_free.argtypes = [C.c_void_p]
        L.orc_csr_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_csr_copy.argtypes = [C.c_void_p, _ip, _ip, _dp]
        L.orc_csr_free.argtypes = [C.c_void_p]
        L.orc_build_Q_pgo.restype = C.c_void_p
        L.orc_build_Q_pgo.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _ip, _dp]
        L.orc_build_G_pgo.restype = C.c_int
        L.orc_build_G_pgo.argtypes = [C.c_int] * 5 + [_ip, _dp, C.c_int, _ip, _dp, _dp]
        L.orc_problem_create.restype = C.c_void_p
        L.orc_problem_create.argtypes = [C.c_int] * 6 + [_ip, _ip, _dp, C.c_void_p, C.c_double]
        L.orc_problem_free.argtypes = [C.c_void_p]
        L.orc_problem_nnzL.restype = C.c_long
        L.orc_problem_nnzL.argtypes = [C.c_void_p]
        L.orc_f.restype = C.c_double
        L.orc_f.argtypes = [C.c_void_p, _dp]
        L.orc_egrad.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_rgrad.restype = C.c_double
        L.orc_rgrad.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_hess.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_precondition.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_precon_solve.argtypes = [C.c_void_p, _dp, _dp]
        for name in ("orc_tangent_project", "orc_retract"):
            getattr(L, name).argtypes = [C.c_int] * 5 + [_dp, _dp, _dp]
        L.orc_project_to_manifold.argtypes = [C.c_int] * 5 + [_dp, _dp]
        L.orc_project_to_rotation_group.argtypes = [C.c_int, _dp, _dp]
        L.orc_optimize.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.orc_dual_certificate.restype = C.c_void_p
        L.orc_dual_certificate.argtypes = [C.c_int] * 5 + [_dp, C.c_int, _ip, _ip, _dp]
        L.orc_is_psd.restype = C.c_int
        L.orc_is_psd.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int]
        L.orc_min_eig.restype = C.c_int
        L.orc_min_eig.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, C.c_double, C.c_int, C.c_ulonglong,
                                  C.POINTER(C.c_double), _dp, C.POINTER(C.c_long)]
        L.orc_lanczos_lm.restype = C.c_int
        L.orc_lanczos_lm.argtypes = [C.c_int, _ip, _ip, _dp, C.c_double, C.c_int, C.c_int, C.c_double,
                                     C.c_ulonglong, C.POINTER(C.c_double), _dp, C.POINTER(C.c_long)]
        L.orc_fast_verification.restype = C.c_int
        L.orc_fast_verification.argtypes = [C.c_int, _ip, _ip, _dp, C.c_double, C.c_int,
                                            C.POINTER(C.c_double), _dp, C.POINTER(C.c_double)]
        L.orc_escape_saddle.restype = C.c_int
        L.orc_escape_saddle.argtypes = [C.c_void_p, _dp, C.c_double, _dp, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_align_lifted_trajectory.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.orc_project_solution_raslam.argtypes = [C.c_int] * 5 + [_dp, _dp]
        L.orc_ra_states_in_local_frame.argtypes = [C.c_int] * 5 + [_dp, _dp, _dp, _dp]
        L.orc_chordal_init.restype = C.c_int
        L.orc_chordal_init.argtypes = [C.c_void_p, _dp]
        L.orc_pyfg_load.restype = C.c_void_p
        L.orc_pyfg_load.argtypes = [C.c_char_p]
        L.orc_ra_info.argtypes = [C.c_void_p, _ip]
        L.orc_ra_copy.argtypes = [C.c_void_p, _ip, _dp, _ip, _dp, _ip, _dp, _dp]
        L.orc_ra_odometry_init.argtypes = [C.c_void_p, C.c_ulonglong, _dp]
        L.orc_build_Q_ra.restype = C.c_void_p
        L.orc_build_Q_ra.argtypes = [C.c_void_p]
        L.orc_ra_free.argtypes = [C.c_void_p]
        L.orc_run_rbcd.restype = C.c_void_p
        L.orc_run_rbcd.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.orc_run_coloured.restype = C.c_void_p
        L.orc_run_coloured.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.orc_trace_info.argtypes = [C.c_void_p, _dp]
        L.orc_trace_copy.argtypes = [C.c_void_p, _dp, _dp, _ip, _ip, C.c_void_p]
        L.orc_trace_free.argtypes = [C.c_void_p]
        L.orc_trace_seconds.argtypes = [C.c_void_p, _dp]
        _lib = L
    return _lib


def F(a):
    """column-major (Fortran) float64 copy flattened in memory order"""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).reshape(-1)


def unF(flat, rows, cols):
    return np.asarray(flat).reshape(cols, rows).T.copy()


class CSR:
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
        return CSR(A.shape[0], A.indptr, A.indices, A.data)


def _take_csr(h):
    L = lib()
    n, nnz = C.c_int(), C.c_int()
    L.orc_csr_info(h, C.byref(n), C.byref(nnz))
    rp = np.zeros(n.value + 1, np.int32)
    ci = np.zeros(nnz.value, np.int32)
    v = np.zeros(nnz.value, np.float64)
    L.orc_csr_copy(h, rp, ci, v)
    L.orc_csr_free(h)
    return CSR(n.value, rp, ci, v)


class Dataset:
    """ids: m x 4 int32 (r1,p1,r2,p2); vals: m x (d*d+d+3) (R col-major, t, kappa, tau, weight)"""

    def __init__(self, d, n, ids, vals):
        self.d, self.n = int(d), int(n)
        self.ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.vals = np.ascontiguousarray(vals, dtype=np.float64)

    @property
    def m(self):
        return self.ids.shape[0]


def read_g2o(path):
    L = lib()
    h = L.orc_g2o_load(path.encode())
    if not h:
        raise IOError(path)
    d, n, m = C.c_int(), C.c_int(), C.c_int()
    L.orc_ds_info(h, C.byref(d), C.byref(n), C.byref(m))
    ids = np.zeros((m.value, 4), np.int32)
    vals = np.zeros((m.value, d.value * d.value + d.value + 3), np.float64)
    L.orc_ds_copy(h, ids, vals)
    L.orc_ds_free(h)
    return Dataset(d.value, n.value, ids, vals)


def build_Q_pgo(ds, n=None, agent=0, ids=None, vals=None):
    L = lib()
    ids = ds.ids if ids is None else np.ascontiguousarray(ids, np.int32)
    vals = ds.vals if vals is None else np.ascontiguousarray(vals, np.float64)
    h = L.orc_build_Q_pgo(ds.d, ds.n if n is None else n, agent, ids.shape[0], ids, vals)
    return _take_csr(h)


def build_G_pgo(r, d, n, agent, ids, vals, keys, poses):
    """poses: list of r x (d+1) arrays"""
    L = lib()
    ids = np.ascontiguousarray(ids, np.int32)
    vals = np.ascontiguousarray(vals, np.float64)
    keys = np.ascontiguousarray(keys, np.int32).reshape(-1, 2)
    flat = np.concatenate([F(p) for p in poses]) if len(poses) else np.zeros(0)
    G = np.zeros(r * (d + 1) * n)
    ok = L.orc_build_G_pgo(r, d, n, agent, ids.shape[0], ids, vals, keys.shape[0], keys, flat, G)
    return (unF(G, r, (d + 1) * n) if ok else None)


class Problem:
    def __init__(self, r, d, n, Q, G=None, reg=0.1, l=0, b=0):
        L = lib()
        self.r, self.d, self.n, self.l, self.b = r, d, n, l, b
        self.k = (d + 1) * n + l + b
        assert Q.n == self.k
        self._G = None if G is None else F(G)
        gp = None if G is None else self._G.ctypes.data_as(C.c_void_p)
        self.h = L.orc_problem_create(r, d, n, l, b, self.k, Q.rp, Q.ci, Q.v, gp, float(reg))

    def __del__(self):
        try:
            lib().orc_problem_free(self.h)
        except Exception:
            pass

    def f(self, X):
        return lib().orc_f(self.h, F(X))

    def egrad(self, X):
        out = np.zeros(self.r * self.k)
        lib().orc_egrad(self.h, F(X), out)
        return unF(out, self.r, self.k)

    def rgrad(self, X):
        out = np.zeros(self.r * self.k)
        lib().orc_rgrad(self.h, F(X), out)
        return unF(out, self.r, self.k)

    def hess(self, X, V):
        out = np.zeros(self.r * self.k)
        lib().orc_hess(self.h, F(X), F(V), out)
        return unF(out, self.r, self.k)

    def precondition(self, X, V):
        out = np.zeros(self.r * self.k)
        lib().orc_precondition(self.h, F(X), F(V), out)
        return unF(out, self.r, self.k)

    def precon_solve(self, V):
        out = np.zeros(self.r * self.k)
        lib().orc_precon_solve(self.h, F(V), out)
        return unF(out, self.r, self.k)

    def nnzL(self):
        return lib().orc_problem_nnzL(self.h)

    def optimize(self, X0, method=0, gradnorm_tol=1e-2, RGD_stepsize=1e-3, RGD_use_precond=1,
                 RTR_iterations=3, RTR_tCG_iterations=50, RTR_initial_radius=100.0):
        prm = np.array([method, gradnorm_tol, RGD_stepsize, RGD_use_precond, RTR_iterations,
                        RTR_tCG_iterations, RTR_initial_radius], dtype=np.float64)
        out = np.zeros(self.r * self.k)
        res = np.zeros(10)
        lib().orc_optimize(self.h, prm, F(X0), out, res)
        keys = ["success", "fInit", "gradNormInit", "fOpt", "gradNormOpt", "elapsedMs", "tcg_status",
                "outer_iters", "inner_iters", "accepted"]
        return unF(out, self.r, self.k), dict(zip(keys, res.tolist()))

    def escape_saddle(self, Xopt, theta, v, gtol=1e-6, pgtol=1e-6, second_order=False):
        out = np.zeros(self.r * self.k)
        ok = lib().orc_escape_saddle(self.h, F(Xopt), float(theta), np.ascontiguousarray(v, np.float64), gtol,
                                     pgtol, int(second_order), out)
        return (unF(out, self.r, self.k) if ok else None)


def tangent_project(r, d, n, X, V, l=0, b=0):
    k = (d + 1) * n + l + b
    out = np.zeros(r * k)
    lib().orc_tangent_project(r, d, n, l, b, F(X), F(V), out)
    return unF(out, r, k)


def retract(r, d, n, X, V, l=0, b=0):
    k = (d + 1) * n + l + b
    out = np.zeros(r * k)
    lib().orc_retract(r, d, n, l, b, F(X), F(V), out)
    return unF(out, r, k)


def project_to_manifold(r, d, n, M, l=0, b=0):
    k = (d + 1) * n + l + b
    out = np.zeros(r * k)
    lib().orc_project_to_manifold(r, d, n, l, b, F(M), out)
    return unF(out, r, k)


def project_to_rotation_group(M):
    d = M.shape[0]
    out = np.zeros(d * d)
    lib().orc_project_to_rotation_group(d, F(M), out)
    return unF(out, d, d)


def dual_certificate(r, d, n, X, Q, l=0, b=0):
    h = lib().orc_dual_certificate(r, d, n, l, b, F(X), Q.n, Q.rp, Q.ci, Q.v)
    return _take_csr(h)


def is_psd(S, block=1):
    return bool(lib().orc_is_psd(S.n, S.rp, S.ci, S.v, block))


def min_eig(S, maxit=1000, tol=1e-3, ncv=20, seed=12345):
    lam, mv = C.c_double(), C.c_long()
    v = np.zeros(S.n)
    ok = lib().orc_min_eig(S.n, S.rp, S.ci, S.v, maxit, tol, ncv, seed, C.byref(lam), v, C.byref(mv))
    return bool(ok), lam.value, v, mv.value


def lanczos_lm(S, shift=0.0, ncv=20, maxit=1000, tol=1e-4, seed=12345):
    lam, mv = C.c_double(), C.c_long()
    v = np.zeros(S.n)
    ok = lib().orc_lanczos_lm(S.n, S.rp, S.ci, S.v, shift, ncv, maxit, tol, seed, C.byref(lam), v, C.byref(mv))
    return bool(ok), lam.value, v, mv.value


def fast_verification(S, eta, block=1):
    th, lm = C.c_double(), C.c_double()
    v = np.zeros(S.n)
    ok = lib().orc_fast_verification(S.n, S.rp, S.ci, S.v, eta, block, C.byref(th), v, C.byref(lm))
    return bool(ok), th.value, v, lm.value


def run_rbcd(ds, X0, num_robots=5, r_min=5, r_max=100, max_iters=1000, min_eig_tol=1e-3, rgrad_tol=0.1,
             acceleration=1, staircase=1, verbose=0, method=0, gradnorm_tol=1e-2, RGD_stepsize=1e-3,
             RGD_use_precond=1, RTR_iterations=3, RTR_tCG_iterations=50, RTR_initial_radius=100.0, threads=1):
    L = lib()
    h = L.orc_ds_create(ds.d, ds.n, ds.m, ds.ids, ds.vals)
    opts = np.array([num_robots, r_min, r_max, max_iters, min_eig_tol, rgrad_tol, acceleration, staircase, verbose,
                     method, gradnorm_tol, RGD_stepsize, RGD_use_precond, RTR_iterations, RTR_tCG_iterations,
                     RTR_initial_radius, threads], dtype=np.float64)
    X0 = np.asarray(X0, dtype=np.float64)
    t = L.orc_run_rbcd(h, opts, F(X0), X0.shape[0])
    info = np.zeros(8)
    L.orc_trace_info(t, info)
    it = int(info[0])
    cost, gn = np.zeros(it), np.zeros(it)
    sel, rk = np.zeros(it, np.int32), np.zeros(it, np.int32)
    rfin = int(info[1])
    k = (ds.d + 1) * ds.n
    Xf = np.zeros(rfin * k)
    L.orc_trace_copy(t, cost, gn, sel, rk, Xf.ctypes.data_as(C.c_void_p))
    secs = np.zeros(it)
    L.orc_trace_seconds(t, secs)
    L.orc_trace_free(t)
    L.orc_ds_free(h)
    return dict(total_iters=it, final_rank=rfin, certified=int(info[2]), theta=info[3], lambda_min=info[4],
                rbcd_seconds=info[5], cert_seconds=info[6], setup_seconds=info[7], cost=cost, gradnorm=gn,
                selected=sel, rank=rk, X=unF(Xf, rfin, k), seconds=secs)


def run_coloured(ds, X0, num_robots=5, r=5, sweeps=10, threads=1, method=0, gradnorm_tol=1e-2, RGD_stepsize=1e-3,
                 RGD_use_precond=1, RTR_iterations=3, RTR_tCG_iterations=50, RTR_initial_radius=100.0):
    """coloured simultaneous updates (agents of one colour at once, non-accelerated; ref src/Agent.cpp:650-678 as ticks)
    with `threads` host threads: per-sweep central cost, loop seconds, set-up seconds"""
    L = lib()
    h = L.orc_ds_create(ds.d, ds.n, ds.m, ds.ids, ds.vals)
    opts = np.array([num_robots, r, 100, sweeps, 1e-3, 0.0, 0, 0, 0, method, gradnorm_tol, RGD_stepsize,
                     RGD_use_precond, RTR_iterations, RTR_tCG_iterations, RTR_initial_radius, threads], dtype=np.float64)
    X0 = np.asarray(X0, dtype=np.float64)
    t = L.orc_run_coloured(h, opts, F(X0), X0.shape[0])
    info = np.zeros(8)
    L.orc_trace_info(t, info)
    it = int(info[0])
    cost, gn = np.zeros(it), np.zeros(it)
    sel, rk = np.zeros(it, np.int32), np.zeros(it, np.int32)
    k = (ds.d + 1) * ds.n
    Xf = np.zeros(r * k)
    L.orc_trace_copy(t, cost, gn, sel, rk, Xf.ctypes.data_as(C.c_void_p))
    L.orc_trace_free(t)
    L.orc_ds_free(h)
    return dict(sweeps=it, colours=int(sel[0]) if it else 0, loop_seconds=info[5], setup_seconds=info[7], cost=cost,
                gradnorm=gn, X=unF(Xf, r, k))


class RADataset:
    """centralised range-aided SLAM dataset (global indices, RA ordering of the ground truth)"""

    def __init__(self, path, init_seed=20250310):
        L = lib()
        h = L.orc_pyfg_load(str(path).encode())
        if not h:
            raise IOError(path)
        info = np.zeros(7, np.int32)
        L.orc_ra_info(h, info)
        self.d, self.n, self.l, self.b = (int(x) for x in info[:4])
        mpp, mpl, mr = (int(x) for x in info[4:])
        d = self.d
        self.k = (d + 1) * self.n + self.l + self.b
        self.pp_ids = np.zeros((mpp, 2), np.int32)
        self.pp_vals = np.zeros((mpp, d * d + d + 3))
        self.pl_ids = np.zeros((max(mpl, 1), 2), np.int32)
        self.pl_vals = np.zeros((max(mpl, 1), d + 2))
        self.r_ids = np.zeros((max(mr, 1), 5), np.int32)
        self.r_vals = np.zeros((max(mr, 1), 3))
        gt = np.zeros(d * self.k)
        L.orc_ra_copy(h, self.pp_ids, self.pp_vals, self.pl_ids, self.pl_vals, self.r_ids, self.r_vals, gt)
        self.pl_ids, self.pl_vals = self.pl_ids[:mpl], self.pl_vals[:mpl]
        self.r_ids, self.r_vals = self.r_ids[:mr], self.r_vals[:mr]
        self.gt = unF(gt, d, self.k)
        self.Q = _take_csr(L.orc_build_Q_ra(h))
        x0 = np.zeros(d * self.k)
        L.orc_ra_odometry_init(h, init_seed, x0)
        self.X_odom = unF(x0, d, self.k)  # ref examples/SingleRobotExample_RASLAM.cpp:92-150
        L.orc_ra_free(h)


def chordal_initialization(ds):
    """ref src/DCORA_solver.cpp:218-268; returns d x (d+1) n"""
    L = lib()
    h = L.orc_ds_create(ds.d, ds.n, ds.m, ds.ids, ds.vals)
    out = np.zeros(ds.d * (ds.d + 1) * ds.n)
    ok = L.orc_chordal_init(h, out)
    L.orc_ds_free(h)
    return unF(out, ds.d, (ds.d + 1) * ds.n) if ok else None


def align_lifted_trajectory_to_frame(X, anchor, d, n, global_alignment=True):
    """ref src/DCORA_utils.cpp:2262-2289 (global) / src/Agent.cpp:963-980 (local); returns d x (d+1) n"""
    r = X.shape[0]
    out = np.zeros(d * (d + 1) * n)
    lib().orc_align_lifted_trajectory(r, d, n, F(X), F(anchor), int(global_alignment), out)
    return unF(out, d, (d + 1) * n)


def project_solution_raslam(X, r, d, n, l, b):
    """ref src/DCORA_utils.cpp:1984-2031; returns d x k"""
    k = (d + 1) * n + l + b
    out = np.zeros(d * k)
    lib().orc_project_solution_raslam(r, d, n, l, b, F(X), out)
    return unF(out, d, k)


def ra_states_in_local_frame(X, r, d, n, l, b):
    """ref src/Agent.cpp:950-1003; returns (trajectory d x (d+1) n, unit spheres d x l, landmarks d x b)"""
    T, S, Lm = np.zeros(d * (d + 1) * n), np.zeros(max(d * l, 1)), np.zeros(max(d * b, 1))
    lib().orc_ra_states_in_local_frame(r, d, n, l, b, F(X), T, S, Lm)
    return unF(T, d, (d + 1) * n), unF(S[:d * l], d, l), unF(Lm[:d * b], d, b)


# ---- robust estimation (oracle_robust.cpp) ----------------------------------------------------------------------------
ROBUST_TYPES = {"L2": 0, "L1": 1, "TLS": 2, "Huber": 3, "GM": 4, "GNC_TLS": 5}


def _robust_prm(cost_type="L2", GNCMaxNumIters=20, GNCBarc=5.0, GNCMuStep=1.4, GNCInitMu=1e-4, HuberThreshold=3.0,
                TLSThreshold=10.0):
    """ref include/DCORA/DCORA_robust.h:50-60 (defaults)"""
    return np.array([ROBUST_TYPES[cost_type], GNCMaxNumIters, GNCBarc, GNCMuStep, GNCInitMu, HuberThreshold,
                     TLSThreshold], dtype=np.float64)


def _bind_robust():
    L = lib()
    if getattr(L, "_robust_bound", False):
        return L
    L.orc_robust_weights.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp]
    L.orc_chi2inv.restype = C.c_double
    L.orc_chi2inv.argtypes = [C.c_double, C.c_int]
    L.orc_error_threshold_at_quantile.restype = C.c_double
    L.orc_error_threshold_at_quantile.argtypes = [C.c_double, C.c_int]
    L.orc_robust_rotation_averaging.argtypes = [C.c_int, C.c_int, _dp, C.c_void_p, C.c_double, _dp, _ip]
    L.orc_robust_pose_averaging.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_void_p, C.c_void_p, C.c_double, _dp, _dp,
                                            _ip]
    L.orc_measurement_errors.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.orc_solve_pgo.argtypes = [C.c_void_p, _dp, C.c_void_p, _dp]
    L.orc_solve_robust_pgo.argtypes = [C.c_void_p, _dp, _dp, _ip, C.c_void_p, _dp, _dp]
    L._robust_bound = True
    return L


def robust_weights(r, updates=0, **prm):
    """RobustCost::weight after `updates` calls of update() (ref src/DCORA_robust.cpp:56-137)"""
    r = np.ascontiguousarray(r, np.float64)
    w = np.zeros_like(r)
    _bind_robust().orc_robust_weights(_robust_prm(**prm), updates, r.size, r, w)
    return w


def chi2inv(q, dof):
    return _bind_robust().orc_chi2inv(q, dof)


def error_threshold_at_quantile(q, dim=3):
    return _bind_robust().orc_error_threshold_at_quantile(q, dim)


def _vp(a):
    return None if a is None else np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)


def robust_single_rotation_averaging(Rs, kappa=None, threshold=1.0):
    """Rs: list of d x d; returns (Ropt, inlier indices) (ref src/DCORA_solver.cpp:76-141)"""
    d, n = Rs[0].shape[0], len(Rs)
    flat = np.concatenate([F(R) for R in Rs])
    Ropt, inl = np.zeros(d * d), np.zeros(n, np.int32)
    k = None if kappa is None else np.ascontiguousarray(kappa, np.float64)
    _bind_robust().orc_robust_rotation_averaging(d, n, flat, _vp(k), threshold, Ropt, inl)
    return unF(Ropt, d, d), np.nonzero(inl)[0]


def robust_single_pose_averaging(Rs, ts, kappa=None, tau=None, threshold=1.0):
    """ref src/DCORA_solver.cpp:143-216"""
    d, n = Rs[0].shape[0], len(Rs)
    flat = np.concatenate([F(R) for R in Rs])
    tflat = np.ascontiguousarray(np.concatenate([np.asarray(t, np.float64).reshape(-1) for t in ts]))
    Ropt, topt, inl = np.zeros(d * d), np.zeros(d), np.zeros(n, np.int32)
    k = None if kappa is None else np.ascontiguousarray(kappa, np.float64)
    ta = None if tau is None else np.ascontiguousarray(tau, np.float64)
    _bind_robust().orc_robust_pose_averaging(d, n, flat, tflat, _vp(k), _vp(ta), threshold, Ropt, topt, inl)
    return unF(Ropt, d, d), topt, np.nonzero(inl)[0]


def measurement_errors(ds, T):
    """computeMeasurementError of every edge (ref src/DCORA_utils.cpp:2095-2101)"""
    L = _bind_robust()
    h = L.orc_ds_create(ds.d, ds.n, ds.m, ds.ids, ds.vals)
    out = np.zeros(ds.m)
    L.orc_measurement_errors(h, int(np.asarray(T).shape[0]), F(T), out)
    L.orc_ds_free(h)
    return out


def _opt(gradnorm_tol=1e-2, RTR_iterations=3, RTR_tCG_iterations=50, RTR_initial_radius=100.0):
    return np.array([gradnorm_tol, RTR_iterations, RTR_tCG_iterations, RTR_initial_radius], dtype=np.float64)


def solve_pgo(ds, T0=None, **opt):
    """solvePGO (ref src/DCORA_solver.cpp:304-328)"""
    L = _bind_robust()
    h = L.orc_ds_create(ds.d, ds.n, ds.m, ds.ids, ds.vals)
    out = np.zeros(ds.d * (ds.d + 1) * ds.n)
    t0 = None if T0 is None else F(T0)
    L.orc_solve_pgo(h, _opt(**opt), _vp(t0), out)
    L.orc_ds_free(h)
    return unF(out, ds.d, (ds.d + 1) * ds.n)


def solve_robust_pgo(ds, fixed, T0=None, robust=None, **opt):
    """solveRobustPGO (ref src/DCORA_solver.cpp:330-409); returns (T, weights)"""
    L = _bind_robust()
    h = L.orc_ds_create(ds.d, ds.n, ds.m, ds.ids, ds.vals)
    out, w = np.zeros(ds.d * (ds.d + 1) * ds.n), np.zeros(ds.m)
    t0 = None if T0 is None else F(T0)
    rp = _robust_prm(**(robust or {"cost_type": "GNC_TLS"}))
    L.orc_solve_robust_pgo(h, _opt(**opt), rp, np.ascontiguousarray(fixed, np.int32), _vp(t0), out, w)
    L.orc_ds_free(h)
    return unF(out, ds.d, (ds.d + 1) * ds.n), w


# ---- cross-robot frame alignment: numpy restatement (ref src/Agent.cpp:460-520, 694-833) ----
def _pose4(T):
    d = T.shape[0]
    M = np.eye(d + 1)
    M[:d] = T
    return M


def neighbor_transform(incoming, Rm, tm, T_w2_f2, T_w1_f1):
    """Agent::computeNeighborTransform (:694-727) in homogeneous matrices"""
    d = Rm.shape[0]
    dT = np.eye(d + 1)
    dT[:d, :d], dT[:d, d] = Rm, tm
    f1f2 = np.linalg.inv(dT) if incoming else dT
    T_w2_f1 = _pose4(T_w2_f2) @ np.linalg.inv(f1f2)
    return (T_w2_f1 @ np.linalg.inv(_pose4(T_w1_f1)))[:d]


def robust_neighbor_transform(cands, two_stage=False, min_inliers=2):
    """computeRobustNeighborTransform (:782-833) / ...TwoStage (:729-780) on the oracle's GNC averaging"""
    d = cands[0].shape[0]
    Rs, ts = [T[:, :d] for T in cands], [T[:, d] for T in cands]
    m = len(cands)
    if two_stage:
        R, inl = robust_single_rotation_averaging(Rs, kappa=np.ones(m), threshold=2 * np.sqrt(2) * np.sin(0.25))
        if len(inl) < min_inliers:
            return None, len(inl)
        t = np.mean([ts[i] for i in inl], axis=0)
    else:
        R, t, inl = robust_single_pose_averaging(Rs, ts, kappa=1.82 * np.ones(m), tau=0.01 * np.ones(m),
                                                 threshold=error_threshold_at_quantile(0.9, 3))
        if len(inl) < min_inliers:
            return None, len(inl)
    return np.hstack([R, np.asarray(t).reshape(d, 1)]), len(inl)


def initialize_in_global_frame(T_world_robot, T_local, YLift, n, l=0, b=0):
    """Agent::initializeInGlobalFrame (:460-520) with alignTrajectoryToFrame / alignUnitSpheresToFrame /
    alignLandmarksToFrame (src/DCORA_utils.cpp:2222-2260)"""
    d = T_world_robot.shape[0]
    R, t = T_world_robot[:, :d], T_world_robot[:, d:d + 1]
    G = R @ T_local
    if l == 0 and b == 0:
        G[:, d::d + 1] += t
    else:
        G[:, d * n + l:] += t
    return YLift @ G
