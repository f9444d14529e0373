"""Robust estimation entry points of the C ABI (include/dcora_hip.h, "Robust estimation"): RobustCost weights,
chi-square quantile, robust single rotation / pose averaging, per-measurement residuals on the device, solvePGO and
solveRobustPGO (ref include/DCORA/DCORA_robust.h, src/DCORA_robust.cpp, src/DCORA_solver.cpp)."""
import ctypes as C

import numpy as np

from . import capi
from .capi import F, RobustParams, ROptResult, check, unF

TYPES = {"L2": 0, "L1": 1, "TLS": 2, "Huber": 3, "GM": 4, "GNC_TLS": 5}


class RobustCostParameters:
    """ref include/DCORA/DCORA_robust.h:25-60 (same defaults)"""

    def __init__(self, costType="L2", **kw):
        self.c = RobustParams()
        capi.lib().dcora_robust_params_default(C.byref(self.c))
        self.c.cost_type = TYPES[costType]
        for k, v in kw.items():
            setattr(self.c, k, v)


def robust_weights(r, params, num_updates=0):
    r = np.ascontiguousarray(r, np.float64)
    w = np.zeros_like(r)
    check(capi.lib().dcora_robust_weights(C.byref(params.c), num_updates, r.size, r, w))
    return w


def chi2inv(quantile, dof):
    out = C.c_double()
    check(capi.lib().dcora_chi2inv(quantile, dof, C.byref(out)))
    return out.value


def computeErrorThresholdAtQuantile(quantile, dimension=3):
    out = C.c_double()
    check(capi.lib().dcora_robust_error_threshold_at_quantile(quantile, dimension, C.byref(out)))
    return out.value


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def robustSingleRotationAveraging(Rs, kappa=None, errorThreshold=1.0):
    """returns (ROpt, inlier indices)"""
    d, n = Rs[0].shape[0], len(Rs)
    flat = np.concatenate([F(R) for R in Rs])
    k = None if kappa is None else np.ascontiguousarray(kappa, np.float64)
    Ropt, inl = np.zeros(d * d), np.zeros(n, np.int32)
    check(capi.lib().dcora_robust_single_rotation_averaging(d, n, flat, _vp(k), errorThreshold, Ropt, inl))
    return unF(Ropt, d, d), np.nonzero(inl)[0]


def robustSinglePoseAveraging(Rs, ts, kappa=None, tau=None, errorThreshold=1.0):
    """returns (ROpt, tOpt, inlier indices)"""
    d, n = Rs[0].shape[0], len(Rs)
    flat = np.concatenate([F(R) for R in Rs])
    tflat = np.ascontiguousarray(np.concatenate([np.asarray(t, np.float64).reshape(-1) for t in ts]))
    k = None if kappa is None else np.ascontiguousarray(kappa, np.float64)
    ta = None if tau is None else np.ascontiguousarray(tau, np.float64)
    Ropt, topt, inl = np.zeros(d * d), np.zeros(d), np.zeros(n, np.int32)
    check(capi.lib().dcora_robust_single_pose_averaging(d, n, flat, tflat, _vp(k), _vp(ta), errorThreshold, Ropt, topt,
                                                        inl))
    return unF(Ropt, d, d), topt, np.nonzero(inl)[0]


def measurement_errors(ds, X, device=0):
    """computeMeasurementError of every edge on the device; X is r x (d+1) n (lifted or not)"""
    X = np.asarray(X, dtype=np.float64)
    h = ds.handle()
    out = np.zeros(ds.m)
    try:
        check(capi.lib().dcora_measurement_errors(h, X.shape[0], F(X), out, device))
    finally:
        capi.lib().dcora_dataset_destroy(h)
    return out


def solvePGO(ds, params, T0=None, device=0):
    """solvePGO (ref src/DCORA_solver.cpp:304-328); returns (T, result dict)"""
    h = ds.handle()
    out = np.zeros(ds.d * (ds.d + 1) * ds.n)
    res = ROptResult()
    t0 = None if T0 is None else F(T0)
    try:
        check(capi.lib().dcora_solve_pgo(h, C.byref(params.c), _vp(t0), out, C.byref(res), device))
    finally:
        capi.lib().dcora_dataset_destroy(h)
    return unF(out, ds.d, (ds.d + 1) * ds.n), res.as_dict()


def solveRobustPGO(ds, params, robust, fixedWeight=None, T0=None, device=0):
    """solveRobustPGO (ref src/DCORA_solver.cpp:330-409); returns (T, final weights)"""
    h = ds.handle()
    out, w = np.zeros(ds.d * (ds.d + 1) * ds.n), np.zeros(ds.m)
    t0 = None if T0 is None else F(T0)
    fx = None if fixedWeight is None else np.ascontiguousarray(fixedWeight, np.int32)
    try:
        check(capi.lib().dcora_solve_robust_pgo(h, C.byref(params.c), C.byref(robust.c), _vp(fx), _vp(t0), out,
                                                _vp(w), device))
    finally:
        capi.lib().dcora_dataset_destroy(h)
    return unF(out, ds.d, (ds.d + 1) * ds.n), w
