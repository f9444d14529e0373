"""The centralised CORA driver of the reference (examples/SingleRobotExample_RASLAM.cpp:48-283) written once over a
small backend adapter, so the same flow runs on the product (dcora_amd, GPU) and on the CPU oracle:

    X0 = odometry start at rank d
    for r = d, d+1, ...:  RTR(200 x 200, tol 1e-4) at rank r  ->  S = Q - Lambda(X)  ->  fastVerification(S, 1e-4)
        certified:  projectSolutionRASLAM -> refine at rank d -> done
        else:       escapeSaddle (second-order step) into rank r + 1

Used by bench.py's config-4 side measurement and by tests/test_cora.py; the backend over the CPU oracle lives in
oracle/flows.py (test infrastructure)."""
import time

import numpy as np
import scipy.sparse as sp

PARAMS = dict(RTR_iterations=200, RTR_tCG_iterations=200, gradnorm_tol=1e-4)
MIN_EIG_TOL = 1e-4


class ProductBackend:
    name = "hip"

    def __init__(self, ra):
        import dcora_amd as da
        self.da, self.ra, self.Q = da, ra, ra.Q
        self.reg = da.precond_regularization(ra.Q)
        self._prep = None

    def problem(self, r):
        ra = self.ra
        return self.da.QuadraticProblem(r, ra.d, ra.n, self.Q, reg=self.reg, l=ra.l, b=ra.b)

    def optimize(self, P, X):
        if self._prep is None:
            # the PSD test of the certificate (S has Q's pattern) is analysed on another host thread during the first solve
            import threading
            ra = self.ra
            self._prep = threading.Thread(target=self.da.cert_prepare, args=(self.Q, ra.d, ra.n),
                                          kwargs=dict(l=ra.l, b=ra.b, block=1), daemon=True)
            self._prep.start()
        opt = self.da.QuadraticOptimizer(P, self.da.ROptParameters(**PARAMS))
        Xo = opt.optimize(X)
        res = opt.getOptResult()
        return Xo, res["fOpt"], res["gradNormOpt"], res["outer_iterations"], res["inner_iterations"]

    def certificate(self, r, X):
        ra = self.ra
        S = self.da.dual_certificate(r, ra.d, ra.n, X, self.Q, l=ra.l, b=ra.b)
        if self._prep is not None:
            self._prep.join()
        psd, theta, v, lmin = self.da.fast_verification(S, MIN_EIG_TOL, block=1)
        return psd, theta, v

    def escape(self, Pnext, X, theta, v):
        return Pnext.escapeSaddle(X, theta, v, 1e-4, 1e-4, isSecondOrder=True)

    def project(self, X, r):
        ra = self.ra
        return self.da.project_solution_raslam(X, r, ra.d, ra.n, ra.l, ra.b)

    def close(self, P):
        P.close()


def cora(backend, X0, d, r_max=20, log=None):
    """returns dict(levels=[{r, f, gradnorm, outer, inner, psd, theta, ms}], certified, r_final, X, X_rounded,
    f_rounded, ms_total) -- the clock covers the loop of the driver (problem construction included, as in the
    reference, where every level builds its Graph / QuadraticProblem afresh)"""
    t_start = time.perf_counter()
    X = np.array(X0, dtype=np.float64)
    levels, certified, Xr, fr = [], False, None, None
    r = d
    while r < r_max:
        t0 = time.perf_counter()
        P = backend.problem(r)
        Xopt, f, gn, outer, inner = backend.optimize(P, X)
        psd, theta, v = backend.certificate(r, Xopt)
        lv = dict(r=r, f=f, gradnorm=gn, outer=outer, inner=inner, psd=bool(psd), theta=theta)
        if psd:
            certified = True
            Xp = Xopt if r == d else backend.project(Xopt, r)
            Pd = P if r == d else backend.problem(d)
            Xr, fr, gr, o2, i2 = backend.optimize(Pd, Xp)
            lv.update(refine_outer=o2, refine_inner=i2)
            if Pd is not P:
                backend.close(Pd)
            backend.close(P)
            lv["ms"] = 1e3 * (time.perf_counter() - t0)
            levels.append(lv)
            X = Xopt
            break
        Pn = backend.problem(r + 1)
        Xn = backend.escape(Pn, Xopt, theta, v)
        backend.close(Pn)
        backend.close(P)
        lv["ms"] = 1e3 * (time.perf_counter() - t0)
        levels.append(lv)
        if log:
            log(lv)
        if Xn is None:
            X = Xopt
            break
        X = Xn
        r += 1
    return dict(levels=levels, certified=certified, r_final=levels[-1]["r"], X=X, X_rounded=Xr, f_rounded=fr,
                ms_total=1e3 * (time.perf_counter() - t_start))
