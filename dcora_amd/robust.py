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


# ---- cross-robot frame alignment (ref src/Agent.cpp:460-520, 694-833); poses are d x (d+1) arrays [R t] ----
def computeNeighborTransforms(incoming, meas_R, meas_t, nbr_poses, my_poses):
    """Agent::computeNeighborTransform for every inter-robot loop closure: list of T_world2_world1"""
    from .capi import Dims  # noqa: F401
    m, d = len(meas_R), meas_R[0].shape[0]
    inc = np.ascontiguousarray(incoming, np.int32)
    out = np.zeros(m * d * (d + 1))
    cat = lambda Ms: np.concatenate([F(M) for M in Ms])
    tflat = np.ascontiguousarray(np.concatenate([np.asarray(t, np.float64).reshape(-1) for t in meas_t]))
    check(capi.lib().dcora_agent_neighbor_transforms(d, m, inc, cat(meas_R), tflat, cat(nbr_poses), cat(my_poses), out))
    return [unF(out[i * d * (d + 1):(i + 1) * d * (d + 1)], d, d + 1) for i in range(m)]


def computeRobustNeighborTransform(candidates, two_stage=False, robustInitMinInliers=2):
    """Agent::computeRobustNeighborTransform[TwoStage]: (T_world_robot or None, number of inliers)"""
    m, d = len(candidates), candidates[0].shape[0]
    flat = np.concatenate([F(T) for T in candidates])
    T, nin, ok = np.zeros(d * (d + 1)), C.c_int(), C.c_int()
    check(capi.lib().dcora_agent_robust_neighbor_transform(d, m, flat, int(two_stage), robustInitMinInliers, T,
                                                           C.byref(nin), C.byref(ok)))
    return (unF(T, d, d + 1) if ok.value else None), nin.value


def initializeInGlobalFrame(T_world_robot, T_local, YLift, n, l=0, b=0):
    """Agent::initializeInGlobalFrame: the lifted start point r x k of an agent whose local estimate is T_local"""
    from .capi import Dims
    d, r = T_world_robot.shape[0], YLift.shape[0]
    dims = Dims(r, d, n, l, b)
    out = np.zeros(r * T_local.shape[1])
    check(capi.lib().dcora_agent_initialize_in_global_frame(C.byref(dims), F(T_world_robot), F(T_local), F(YLift), out))
    return unF(out, r, T_local.shape[1])


def fixedStiefelVariable(r, d):
    """the shared lifting matrix YLift (ref src/DCORA_utils.cpp:2053-2056)"""
    out = np.zeros(r * d)
    check(capi.lib().dcora_fixed_stiefel_variable(r, d, out))
    return unF(out, r, d)
