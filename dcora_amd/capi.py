"""ctypes binding of dcora_amd/lib/libdcora_hip.so (C ABI declared in include/dcora_hip.h).

The library is the product: hand-written HIP kernels for gfx950 plus the C++ host code that paces them.
There is no CPU fallback -- every compute entry point returns DCORA_ERR_NO_DEVICE when no GPU is usable and
this module raises DcoraError for any non-zero status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdcora_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dcora_hip.h")


class DcoraError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("dcora status %d: %s" % (status, msg))
        self.status = status


class Dims(C.Structure):
    _fields_ = [("r", C.c_int), ("d", C.c_int), ("n", C.c_int), ("l", C.c_int), ("b", C.c_int), ("layout", C.c_int)]


LAYOUT_AUTO, LAYOUT_SE, LAYOUT_RA = 0, 1, 2


class ROptParams(C.Structure):
    _fields_ = [("method", C.c_int), ("verbose", C.c_int), ("gradnorm_tol", C.c_double),
                ("RGD_stepsize", C.c_double), ("RGD_use_preconditioner", C.c_int), ("RTR_iterations", C.c_int),
                ("RTR_tCG_iterations", C.c_int), ("RTR_initial_radius", C.c_double)]


class ROptResult(C.Structure):
    _fields_ = [("success", C.c_int), ("fInit", C.c_double), ("gradNormInit", C.c_double), ("fOpt", C.c_double),
                ("gradNormOpt", C.c_double), ("elapsedMs", C.c_double), ("tCGStatus", C.c_int),
                ("outer_iterations", C.c_int), ("inner_iterations", C.c_int), ("accepted_steps", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RobustParams(C.Structure):
    _fields_ = [("cost_type", C.c_int), ("GNCMaxNumIters", C.c_int), ("GNCBarc", C.c_double),
                ("GNCMuStep", C.c_double), ("GNCInitMu", C.c_double), ("HuberThreshold", C.c_double),
                ("TLSThreshold", C.c_double)]


class RbcdOptions(C.Structure):
    _fields_ = [("num_robots", C.c_int), ("r", C.c_int), ("acceleration", C.c_int), ("restart_interval", C.c_int),
                ("local", ROptParams), ("rank", C.c_int), ("world_size", C.c_int), ("device", C.c_int),
                ("stream", C.c_void_p)]


def build(force=False):
    """compile the library for gfx950 (hipcc cross-compiles without a GPU)"""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_vp = C.c_void_p
_PI, _PD = C.POINTER(C.c_int), C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/dcora_hip.h declares
SIGNATURES = {
    "dcora_status_string": (C.c_char_p, [C.c_int]),
    "dcora_last_error": (C.c_char_p, []),
    "dcora_device_count": (C.c_int, []),
    "dcora_ropt_params_default": (None, [C.POINTER(ROptParams)]),
    "dcora_problem_create": (C.c_int, [C.POINTER(Dims), _ip, _ip, _dp, _vp, C.c_double, C.c_int, C.POINTER(_vp)]),
    "dcora_problem_destroy": (C.c_int, [_vp]),
    "dcora_problem_set_linear_term": (C.c_int, [_vp, _vp]),
    "dcora_problem_cost": (C.c_int, [_vp, _dp, _PD]),
    "dcora_problem_eucgrad": (C.c_int, [_vp, _dp, _dp]),
    "dcora_problem_riegrad": (C.c_int, [_vp, _dp, _vp, _PD]),
    "dcora_problem_hessvec": (C.c_int, [_vp, _dp, _dp, _dp]),
    "dcora_debug_hessvec_solver_form": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "dcora_problem_precondition": (C.c_int, [_vp, _dp, _dp, _dp]),
    "dcora_problem_retract": (C.c_int, [_vp, _dp, _dp, _dp]),
    "dcora_problem_tangent_project": (C.c_int, [_vp, _dp, _dp, _dp]),
    "dcora_problem_escape_saddle": (C.c_int, [_vp, _dp, C.c_double, _dp, C.c_double, C.c_double, C.c_int, _dp, _PI]),
    "dcora_manifold_project": (C.c_int, [C.POINTER(Dims), _dp, _dp, C.c_int]),
    "dcora_optimizer_optimize": (C.c_int, [_vp, C.POINTER(ROptParams), _dp, _dp, C.POINTER(ROptResult)]),
    "dcora_csr_info": (C.c_int, [_vp, _PI, _PI]),
    "dcora_csr_copy": (C.c_int, [_vp, _ip, _ip, _dp]),
    "dcora_csr_destroy": (C.c_int, [_vp]),
    "dcora_cert_dual_matrix": (C.c_int, [C.POINTER(Dims), _dp, _ip, _ip, _dp, C.c_int, C.POINTER(_vp)]),
    "dcora_cert_is_psd": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, _PI]),
    "dcora_cert_is_psd_device": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, C.c_int, _PI, _dp]),
    "dcora_chol_host_selftest": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, _PI, _PD, _dp]),
    "dcora_chol_cache_clear": (C.c_int, []),
    "dcora_cert_min_eig": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, C.c_double, C.c_int, C.c_ulonglong, C.c_int,
                                     _PD, _dp, C.POINTER(C.c_long)]),
    "dcora_cert_fast_verification": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_double, C.c_int, C.c_int, _PI, _PD, _dp,
                                               _PD]),
    "dcora_cert_lambda_min_certified": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_double, C.c_int, C.c_int, _PD, _PI]),
    "dcora_cert_suboptimality_gap": (C.c_int, [C.POINTER(Dims), _dp, C.c_double, _PD, _PD]),
    "dcora_dataset_load_g2o": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "dcora_dataset_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _dp, C.POINTER(_vp)]),
    "dcora_dataset_info": (C.c_int, [_vp, _PI, _PI, _PI]),
    "dcora_dataset_copy": (C.c_int, [_vp, _ip, _dp]),
    "dcora_dataset_destroy": (C.c_int, [_vp]),
    "dcora_dataset_chordal_init": (C.c_int, [_vp, _dp]),
    "dcora_dataset_chordal_init_device": (C.c_int, [_vp, C.c_int, _dp]),
    "dcora_graph_build_Q_pgo": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _ip, _dp, C.POINTER(_vp)]),
    "dcora_radataset_load_pyfg": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "dcora_radataset_info": (C.c_int, [_vp, _ip]),
    "dcora_radataset_create": (C.c_int, [C.c_int] * 5 + [_vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, C.POINTER(_vp)]),
    "dcora_radataset_copy": (C.c_int, [_vp] * 7),
    "dcora_radataset_ground_truth": (C.c_int, [_vp, _dp]),
    "dcora_radataset_build_Q": (C.c_int, [_vp, C.POINTER(_vp)]),
    "dcora_radataset_odometry_init": (C.c_int, [_vp, C.c_ulonglong, _dp]),
    "dcora_radataset_ownership": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dcora_radataset_agent_columns": (C.c_int, [_vp, C.c_int, _ip, _vp, _PI]),
    "dcora_graph_extract_agent_blocks": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, _ip, C.POINTER(_vp),
                                                   C.POINTER(_vp)]),
    "dcora_radataset_destroy": (C.c_int, [_vp]),
    "dcora_graph_precond_regularization": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, _PD]),
    "dcora_rbcd_options_default": (None, [C.POINTER(RbcdOptions)]),
    "dcora_rbcd_create": (C.c_int, [_vp, C.POINTER(RbcdOptions), C.POINTER(_vp)]),
    "dcora_rbcd_destroy": (C.c_int, [_vp]),
    "dcora_rbcd_set_X": (C.c_int, [_vp, _dp]),
    "dcora_rbcd_get_X": (C.c_int, [_vp, _dp]),
    "dcora_rbcd_iterate": (C.c_int, [_vp, C.c_int, _PD, _PD, _vp, _PI]),
    "dcora_rbcd_run": (C.c_int, [_vp, C.c_int, C.c_double, _PI, _vp, _vp, _vp]),
    "dcora_rbcd_iterate_set": (C.c_int, [_vp, _ip, C.c_int, C.c_int]),
    "dcora_rbcd_set_acceleration": (C.c_int, [_vp, C.c_int]),
    "dcora_rbcd_agent_colours": (C.c_int, [_vp, _ip, _PI]),
    "dcora_rbcd_evaluate": (C.c_int, [_vp, _PD, _PD, _vp, _PI]),
    "dcora_rbcd_agent_iterate": (C.c_int, [_vp, C.c_int, C.c_int]),
    "dcora_rbcd_agent_get_X": (C.c_int, [_vp, C.c_int, _dp]),
    "dcora_rbcd_agent_set_X": (C.c_int, [_vp, C.c_int, _dp]),
    "dcora_rbcd_agent_update_neighbor": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, _dp, C.c_int]),
    "dcora_rbcd_agent_last_skipped": (C.c_int, [_vp, C.c_int, _PI]),
    "dcora_rbcd_agent_info": (C.c_int, [_vp, C.c_int, _PI, _PI, _PI]),
    "dcora_rbcd_last_result": (C.c_int, [_vp, C.POINTER(ROptResult)]),
    "dcora_rbcd_profile_tcg_runs": (C.c_int, [_vp, C.c_int]),
    "dcora_rbcd_profile_tcg_read": (C.c_int, [_vp, _dp]),
    "dcora_rbcd_X_device_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "dcora_rbcd_public_count": (C.c_int, [_vp, C.c_int, _PI]),
    "dcora_rbcd_public_indices": (C.c_int, [_vp, C.c_int, _ip]),
    "dcora_rbcd_pack_public_dev": (C.c_int, [_vp, C.c_int, _vp]),
    "dcora_rbcd_unpack_public_dev": (C.c_int, [_vp, C.c_int, _vp]),
    "dcora_rbcd_phase_nonselected": (C.c_int, [_vp, C.c_int]),
    "dcora_rbcd_phase_selected": (C.c_int, [_vp, C.c_int]),
    "dcora_rbcd_phase_evaluate_dev": (C.c_int, [_vp, _vp]),
    "dcora_rbcd_synchronize": (C.c_int, [_vp]),
    "dcora_exchange_create": (C.c_int, [_vp, C.c_char_p, C.POINTER(_vp)]),
    "dcora_exchange_destroy": (C.c_int, [_vp]),
    "dcora_exchange_create_ra": (C.c_int, [_vp, C.c_char_p, C.POINTER(_vp)]),
    "dcora_exchange_info": (C.c_int, [_vp, _dp]),
    "dcora_exchange_link_report": (C.c_int, [_vp, _dp]),
    "dcora_exchange_post": (C.c_int, [_vp, _ip, C.c_int]),
    "dcora_exchange_wait": (C.c_int, [_vp, _ip, C.c_int]),
    "dcora_exchange_evaluate": (C.c_int, [_vp, _PD, _PD, _vp, _PI]),
    "dcora_exchange_rbcd_iterate": (C.c_int, [_vp, C.c_int, _PD, _PD, _vp, _PI]),
    "dcora_exchange_rbcd_tick": (C.c_int, [_vp, _ip, C.c_int, C.c_int]),
    "dcora_exchange_set_X": (C.c_int, [_vp, _dp]),
    "dcora_exchange_gather_X": (C.c_int, [_vp, _dp]),
    "dcora_exchange_barrier": (C.c_int, [_vp]),
    "dcora_exchange_all_ready": (C.c_int, [_vp, C.c_int, _PI]),
    "dcora_exchange_certify": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_double, _PI, _PD, _PD, _dp,
                                         C.POINTER(C.c_longlong), _PI]),
    "dcora_exchange_host_selftest": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _PD]),
    "dcora_debug_exchange_leave_stale": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "dcora_debug_exchange_probe_fault": (C.c_int, [C.c_int]),
    "dcora_cert_prepare": (C.c_int, [C.POINTER(Dims), _ip, _ip, C.c_int, C.c_int]),
    "dcora_debug_tcg_run_fault": (C.c_int, [C.c_int]),
    "dcora_debug_tcg_run_fault_at": (C.c_int, [C.c_int, C.c_int]),
    "dcora_problem_solver_info": (C.c_int, [_vp, _dp]),
    "dcora_ra_rbcd_create": (C.c_int, [_vp, C.POINTER(RbcdOptions), C.POINTER(_vp)]),
    "dcora_ra_rbcd_destroy": (C.c_int, [_vp]),
    "dcora_ra_rbcd_info": (C.c_int, [_vp, _PI, _vp]),
    "dcora_ra_rbcd_set_X": (C.c_int, [_vp, _dp]),
    "dcora_ra_rbcd_get_X": (C.c_int, [_vp, _dp]),
    "dcora_ra_rbcd_iterate": (C.c_int, [_vp, C.c_int, _PD, _PD, _vp, _PI]),
    "dcora_ra_rbcd_evaluate": (C.c_int, [_vp, _PD, _PD, _vp, _PI]),
    "dcora_ra_rbcd_run": (C.c_int, [_vp, C.c_int, C.c_double, _PI, _vp, _vp, _vp]),
    "dcora_ra_rbcd_last_result": (C.c_int, [_vp, C.POINTER(ROptResult)]),
    "dcora_problem_time_qapply": (C.c_int, [_vp, C.c_int, _PD, _PD]),
    "dcora_problem_time_qapply_rotating": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _PD]),
    "dcora_problem_qapply_info": (C.c_int, [_vp, _dp]),
    "dcora_problem_time_precond": (C.c_int, [_vp, C.c_int, _PD, _PD]),
    "dcora_problem_precond_info": (C.c_int, [_vp, _dp]),
    "dcora_precond_cache_info": (C.c_int, [_dp]),
    "dcora_precond_cache_clear": (C.c_int, []),
    "dcora_robust_params_default": (None, [C.POINTER(RobustParams)]),
    "dcora_robust_weights": (C.c_int, [C.POINTER(RobustParams), C.c_int, C.c_int, _dp, _dp]),
    "dcora_chi2inv": (C.c_int, [C.c_double, C.c_int, _PD]),
    "dcora_robust_error_threshold_at_quantile": (C.c_int, [C.c_double, C.c_int, _PD]),
    "dcora_robust_single_rotation_averaging": (C.c_int, [C.c_int, C.c_int, _dp, _vp, C.c_double, _dp, _ip]),
    "dcora_robust_single_pose_averaging": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _vp, _vp, C.c_double, _dp, _dp, _ip]),
    "dcora_agent_neighbor_transforms": (C.c_int, [C.c_int, C.c_int, _ip, _dp, _dp, _dp, _dp, _dp]),
    "dcora_agent_robust_neighbor_transform": (C.c_int, [C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, _PI, _PI]),
    "dcora_log_trajectory": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _dp]),
    "dcora_fixed_stiefel_variable": (C.c_int, [C.c_int, C.c_int, _dp]),
    "dcora_agent_initialize_in_global_frame": (C.c_int, [C.POINTER(Dims), _dp, _dp, _dp, _dp]),
    "dcora_measurement_errors": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int]),
    "dcora_solve_pgo": (C.c_int, [_vp, C.POINTER(ROptParams), _vp, _dp, C.POINTER(ROptResult), C.c_int]),
    "dcora_solve_robust_pgo": (C.c_int, [_vp, C.POINTER(ROptParams), C.POINTER(RobustParams), _vp, _vp, _dp, _vp,
                                         C.c_int]),
    "dcora_round_align_trajectory": (C.c_int, [C.POINTER(Dims), _dp, _vp, C.c_int, _dp, _vp, _vp, C.c_int]),
    "dcora_round_project_solution_raslam": (C.c_int, [C.POINTER(Dims), _dp, _dp, C.c_int]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DcoraError(-1, "libdcora_hip.so is not built (run __graft_entry__.build()); there is no fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise DcoraError(status, lib().dcora_last_error().decode() or lib().dcora_status_string(status).decode())


def F(a):
    """column-major float64 buffer of a 2-D array"""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).reshape(-1)


def unF(flat, rows, cols):
    return np.asarray(flat).reshape(cols, rows).T.copy()


def take_csr(h):
    L = lib()
    n, nnz = C.c_int(), C.c_int()
    check(L.dcora_csr_info(h, C.byref(n), C.byref(nnz)))
    rp = np.zeros(n.value + 1, np.int32)
    ci = np.zeros(max(nnz.value, 1), np.int32)
    v = np.zeros(max(nnz.value, 1), np.float64)
    check(L.dcora_csr_copy(h, rp, ci, v))
    L.dcora_csr_destroy(h)
    return n.value, rp, ci[:nnz.value].copy(), v[:nnz.value].copy()
