"""The synchronous multi-robot driver of the reference (examples/MultiRobotExample.cpp:121-372) over the C ABI:

    for r = r_min, r_min + 1, ...:
        agents at rank r, X = current point                       (:172-217)
        RBCD++ until |rgrad| < tol or max_iters                   (:223-307)   dcora_rbcd_run
        S = Q - Lambda(X);  fastVerification(S, min_eig_tol)      (:320-334)   dcora_cert_*
        certified: done;  else escapeSaddle into rank r + 1       (:352-366)   dcora_problem_escape_saddle

Everything numerical runs on the device; this file is the control flow of the example program."""
import threading
import time

import numpy as np

from . import (QuadraticProblem, RbcdSession, build_Q_pgo, cert_prepare, dual_certificate, fast_verification,
               lambda_min_certified, suboptimality_gap)


def multi_robot_example(ds, X0, num_robots=5, r_min=5, r_max=100, max_iters=1000, rgrad_tol=0.1, min_eig_tol=1e-3,
                        gradient_tolerance=1e-6, preconditioned_gradient_tolerance=1e-6, acceleration=True,
                        params=None, device=0, refine_gap=False):
    """X0: r_min x (d+1) n start point.  Returns a dict: X (final rank x k), rank, certified, theta, per-level
    records (rank, iterations, cost 2f, gradnorm, seconds of RBCD / certification / escape) and the traces."""
    d, n = ds.d, ds.n
    k = (d + 1) * n
    Q = build_Q_pgo(ds)
    X = np.asarray(X0, dtype=np.float64)
    if X.shape != (r_min, k):
        raise ValueError("X0 must be r_min x (d+1) n")
    levels, cost, gradnorm, selected, rank = [], [], [], [], []
    certified, theta, total = False, 0.0, 0
    r = r_min
    prep = None
    while r < r_max:
        ts = time.perf_counter()
        if prep is None:
            # the certificate S = Q - Lambda has Q's pattern: its PSD test is analysed (ordering, fronts, device image)
            # on another host thread while the agents iterate -- inside the clock of the first level, like the agents'
            # own set-up; the reference analyses inside isSparseSymmetricMatrixPSD, after the loop
            prep = threading.Thread(target=cert_prepare, args=(Q, d, n), kwargs=dict(block=d + 1, device=device),
                                    daemon=True)
            prep.start()
        s = RbcdSession(ds, num_robots=num_robots, r=r, acceleration=acceleration, params=params, device=device)
        s.set_X(X)
        t0 = time.perf_counter()
        out = s.run(max_iters=max_iters, rgrad_tol=rgrad_tol)
        t1 = time.perf_counter()
        Xopt = s.get_X()
        total += out["iters"]
        cost.append(out["cost"])
        gradnorm.append(out["gradnorm"])
        selected.append(out["selected"])
        rank.append(np.full(out["iters"], r, np.int32))
        S = dual_certificate(r, d, n, Xopt, Q, device=device)
        prep.join()
        psd, theta, v, lmin = fast_verification(S, min_eig_tol, block=d + 1, device=device)
        t2 = time.perf_counter()
        s.close()  # (the agents live through the certification, as in the reference's driver)
        lev = {"rank": r, "iterations": int(out["iters"]), "cost_2f": float(out["cost"][-1]),
               "gradnorm": float(out["gradnorm"][-1]), "setup_s": t0 - ts, "rbcd_s": t1 - t0,
               "certification_s": t2 - t1,
               "certified": bool(psd), "theta": float(theta)}
        # bound on f(X) - f* implied by the certificate (this library's addition, include/dcora_hip.h)
        lev["suboptimality_gap_f"], lev["n_eff"] = suboptimality_gap(r, d, n, Xopt, psd, min_eig_tol, lmin)
        if psd and refine_gap:  # a verified lower bound of lambda_min(S) instead of the -eta the test guarantees
            tg = time.perf_counter()
            lam, _ = lambda_min_certified(S, min_eig_tol, block=d + 1)
            lev["lambda_min_S"] = lam
            lev["suboptimality_gap_f_refined"] = suboptimality_gap(r, d, n, Xopt, psd, min_eig_tol,
                                                                   lambda_bound=min(lam, 0.0))[0]
            lev["gap_refinement_s"] = time.perf_counter() - tg
        levels.append(lev)
        X = Xopt
        if psd:
            certified = True
            break
        if theta >= -min_eig_tol / 2:  # :333-335: the eigenvalue computation did not reach the precision to escape
            raise RuntimeError("escape direction computation did not converge to the desired precision")
        Pn = QuadraticProblem(r + 1, d, n, Q, device=device)
        Xn = Pn.escapeSaddle(Xopt, theta, v, gradient_tolerance, preconditioned_gradient_tolerance)
        Pn.close()
        lev["escape_s"] = time.perf_counter() - t2
        if Xn is None:  # :367-370: no descent found along the escape direction
            lev["escaped"] = False
            break
        lev["escaped"] = True
        X = Xn
        r += 1
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
    return {"X": X, "rank": X.shape[0], "certified": certified, "theta": float(theta), "total_iters": int(total),
            "suboptimality_gap_f": levels[-1]["suboptimality_gap_f"] if levels else None, "levels": levels, "cost": cat(cost, float), "gradnorm": cat(gradnorm, float),
            "selected": cat(selected, np.int32), "rank_trace": cat(rank, np.int32)}


def loop_closure_mask(ds, num_robots):
    """measurements the robust outer loop reweights: everything but odometry, i.e. but consecutive poses of one
    robot under the driver's contiguous partition (ref src/Graph.cpp activeLoopClosures / odometry split)"""
    per = ds.n // num_robots
    rob = np.minimum(ds.ids[:, [1, 3]] // per, num_robots - 1)
    return ~((rob[:, 0] == rob[:, 1]) & (ds.ids[:, 3] == ds.ids[:, 1] + 1))


def multi_robot_gnc_example(ds, X0, num_robots=5, r=5, robust=None, num_weight_updates=10, inner_iters=30,
                            rgrad_tol=0.1, max_final_iters=1000, acceleration=True, params=None, fixed=None,
                            device=0):
    """The agents' robust outer loop (ref src/Agent.cpp:1280-1441) around the RBCD session, for all agents at once:

        initializeRobustOptimization: weight 1 on every loop closure that is not fixed, RobustCost reset   (:1332-1346)
        repeat robustOptNumWeightUpdates times:
            RBCD++ for at most robustOptInnerIters iterations (or until |rgrad| < tol)                    (:1280-1330)
            updateMeasurementWeights: residual = sqrt(computeMeasurementError) on the lifted iterate,
                weight = RobustCost::weight(residual); data matrices rebuilt; RobustCost::update;
                warm start (robustOptNumResets = 0); acceleration re-initialised                          (:1397-1441)
        RBCD++ to convergence with the final weights

    ds.vals[:, -1] (the weights) is updated in place.  Returns X, weights, per-round records."""
    from . import robust as rb
    robust = robust or rb.RobustCostParameters("GNC_TLS")
    lc = loop_closure_mask(ds, num_robots)
    if fixed is not None:
        lc &= ~np.asarray(fixed, bool)
    w = ds.vals[:, -1]
    w[lc] = 1.0
    X = np.asarray(X0, dtype=np.float64)
    rounds = []

    def rbcd(X, iters):
        s = RbcdSession(ds, num_robots=num_robots, r=r, acceleration=acceleration, params=params, device=device)
        s.set_X(X)
        out = s.run(max_iters=iters, rgrad_tol=rgrad_tol)
        Xn = s.get_X()
        s.close()
        return Xn, out

    for u in range(num_weight_updates):
        X, out = rbcd(X, inner_iters)
        e = rb.measurement_errors(ds, X, device=device)
        w[lc] = rb.robust_weights(np.sqrt(e[lc]), robust, num_updates=u)
        rounds.append({"iterations": int(out["iters"]), "cost_2f": float(out["cost"][-1]),
                       "accepted": int(np.sum(w[lc] > 1 - 1e-8)), "rejected": int(np.sum(w[lc] < 1e-8))})
    X, out = rbcd(X, max_final_iters)
    return {"X": X, "weights": w.copy(), "loop_closures": lc, "rounds": rounds,
            "final": {"iterations": int(out["iters"]), "cost_2f": float(out["cost"][-1]),
                      "gradnorm": float(out["gradnorm"][-1])}}


def multi_robot_raslam_example(ra, X0, r_min=None, r_max=100, max_iters=1000, rgrad_tol=0.1, min_eig_tol=1e-3,
                               gradient_tolerance=1e-4, preconditioned_gradient_tolerance=1e-4, acceleration=True,
                               params=None, device=0):
    """The multi-robot range-aided SLAM driver (ref examples/MultiRobotExample_RASLAM.cpp) over the C ABI: per level
    the agents' RBCD++ (dcora_ra_rbcd_*), the merged problem's dual certificate and fastVerification, escapeSaddle of
    the central problem into the next rank.  Defaults of the example: r_min = d, local RTR 200 x 200 at 1e-4,
    |rgrad| < 0.1, eigenvalue tolerance 1e-3.  X0: r_min x k (RA ordering)."""
    from . import RaRbcdSession, ROptParameters, precond_regularization
    d, n, l, b, k = ra.d, ra.n, ra.l, ra.b, ra.k
    r_min = d if r_min is None else r_min
    if params is None:
        params = ROptParameters(RTR_iterations=200, RTR_tCG_iterations=200, gradnorm_tol=1e-4)
    X = np.asarray(X0, dtype=np.float64)
    if X.shape != (r_min, k):
        raise ValueError("X0 must be r_min x k")
    reg = None
    levels, certified, theta, total = [], False, 0.0, 0
    r = r_min
    while r < r_max:
        ts = time.perf_counter()
        s = RaRbcdSession(ra, r, acceleration=acceleration, params=params, device=device)
        s.set_X(X)
        t0 = time.perf_counter()
        out = s.run(max_iters=max_iters, rgrad_tol=rgrad_tol)
        t1 = time.perf_counter()
        Xopt = s.get_X()
        s.close()
        total += out["iters"]
        S = dual_certificate(r, d, n, Xopt, ra.Q, l=l, b=b, device=device)
        psd, theta, v, lmin = fast_verification(S, min_eig_tol, block=1, device=device)
        t2 = time.perf_counter()
        lev = {"rank": r, "iterations": int(out["iters"]), "cost_2f": float(out["cost"][-1]),
               "gradnorm": float(out["gradnorm"][-1]), "setup_s": t0 - ts, "rbcd_s": t1 - t0,
               "certification_s": t2 - t1,
               "certified": bool(psd), "theta": float(theta)}
        levels.append(lev)
        X = Xopt
        if psd:
            certified = True
            break
        if theta >= -min_eig_tol / 2:
            raise RuntimeError("escape direction computation did not converge to the desired precision")
        if reg is None:
            reg = precond_regularization(ra.Q, device=device)
        Pn = QuadraticProblem(r + 1, d, n, ra.Q, reg=reg, l=l, b=b, device=device)
        Xn = Pn.escapeSaddle(Xopt, theta, v, gradient_tolerance, preconditioned_gradient_tolerance)
        Pn.close()
        lev["escape_s"] = time.perf_counter() - t2
        lev["escaped"] = Xn is not None
        if Xn is None:
            break
        X = Xn
        r += 1
    return {"X": X, "rank": X.shape[0], "certified": certified, "theta": float(theta), "total_iters": int(total),
            "levels": levels}
