"""ctypes binding of include/ezpz_amd.h.  The library is required: there is no Python/CPU fallback."""
import ctypes as C
import os

import numpy as np

from . import build as _build

CONSTRAINT_DTYPE = np.dtype(
    [("kind", "<u2"), ("tag", "u1"), ("flags", "u1"), ("priority", "<u4"), ("ids", "<u4", (8,)), ("param", "<f8"),
     ("weight", "<f8")]
)
assert CONSTRAINT_DTYPE.itemsize == 56
STATUS_DTYPE = np.dtype(
    [("iterations", "<u4"), ("converged", "<u4"), ("n_unsatisfied", "<u4"), ("n_warnings", "<u4"),
     ("final_residual_inf", "<f8"), ("final_lambda", "<f8")]
)
assert STATUS_DTYPE.itemsize == 32


class CConfig(C.Structure):
    _fields_ = [("max_iterations", C.c_uint64), ("residual_tolerance", C.c_double), ("step_tolerance", C.c_double),
                ("initial_lambda", C.c_double)]


class CWarning(C.Structure):
    _fields_ = [("about_constraint", C.c_int32), ("content", C.c_int32)]


class COutcome(C.Structure):
    _fields_ = [("error", C.c_int32), ("err_constraint_id", C.c_int32), ("err_variable", C.c_int64),
                ("iterations", C.c_uint64), ("converged", C.c_int32), ("priority_solved", C.c_uint32),
                ("n_unsatisfied", C.c_uint64), ("n_warnings", C.c_uint64), ("num_vars", C.c_uint64),
                ("num_eqs", C.c_uint64), ("final_lambda", C.c_double), ("final_residual_inf", C.c_double)]


class CSystemInfo(C.Structure):
    _fields_ = [("n_constraints", C.c_uint64), ("n_vars", C.c_uint64), ("n_rows", C.c_uint64), ("nnz_j", C.c_uint64),
                ("nnz_a", C.c_uint64), ("nnz_l", C.c_uint64), ("n_levels", C.c_uint64), ("n_components", C.c_uint64),
                ("program_bytes", C.c_uint64), ("workspace_bytes", C.c_uint64), ("team_size", C.c_uint32),
                ("workspace_in_lds", C.c_uint32), ("team_mode", C.c_uint32), ("n_partitions", C.c_uint32),
                ("program_in_lds", C.c_uint32), ("grid_workgroups", C.c_uint32), ("front_workgroups", C.c_uint32),
                ("front_max_batch", C.c_uint32)]


class CLaunchPolicy(C.Structure):
    _fields_ = [("compute_units", C.c_uint32), ("lanes_min_systems_small", C.c_uint64), ("lanes_min_systems_large", C.c_uint64),
                ("lanes_large_from_vars", C.c_uint32), ("jit_lane_min_batch", C.c_uint64), ("jit_comp_min_batch", C.c_uint64),
                ("jit_comp_min_values", C.c_uint64), ("jit_after_launches", C.c_uint32), ("lane_max_vars", C.c_uint32),
                ("lane_max_constraints", C.c_uint32), ("comp_min_components", C.c_uint32), ("comp_max_component_vars", C.c_uint32),
                ("comp_max_component_constraints", C.c_uint32), ("comp_max_classes", C.c_uint32), ("rec_min_vars_one_solve", C.c_uint32),
                ("rec_min_vars_batch", C.c_uint32), ("rec_one_wavefront_max_vars", C.c_uint32), ("rec_max_components", C.c_uint32),
                ("rec_wide_one_solve_max_vars", C.c_uint32), ("sub_team_max_width", C.c_uint32), ("dense8_max_vars", C.c_uint32),
                ("zero_copy_max_bytes", C.c_uint64), ("h2h_piece_min_bytes", C.c_uint64), ("h2h_piece_max_bytes", C.c_uint64),
                ("h2h_pieces_per_call", C.c_uint32), ("one_call_host_mask_max_constraints", C.c_uint32),
                ("one_call_host_log_max_entries", C.c_uint32), ("front_min_vars_one_solve", C.c_uint32),
                ("front_min_vars_batch", C.c_uint32), ("front_vars_per_workgroup", C.c_uint32), ("front_max_workgroups", C.c_uint32),
                ("front_small_call_wgs_per_round", C.c_uint32)]


# every symbol include/ezpz_amd.h declares
EXPORTS = [
    "ezpz_default_config", "ezpz_device_count", "ezpz_error_string", "ezpz_system_create", "ezpz_system_destroy",
    "ezpz_system_info", "ezpz_system_solve_batch_device", "ezpz_system_solve_batch", "ezpz_solve_inner", "ezpz_solve",
    "ezpz_problem_parse", "ezpz_problem_destroy", "ezpz_problem_num_constraints", "ezpz_problem_num_vars",
    "ezpz_problem_constraints", "ezpz_problem_guesses", "ezpz_problem_num_labels", "ezpz_problem_label",
    "ezpz_analyze", "ezpz_system_eval_batch", "ezpz_system_jacobian_pattern", "ezpz_cache_clear",
    "ezpz_solve_batch",
    "ezpz_current_device",
    "ezpz_solve_analysis",
    "ezpz_system_freedom_batch",
    "ezpz_system_freedom_batch_device",
    "ezpz_resolve_sides",
    "ezpz_system_specialize",
    "ezpz_host_register",
    "ezpz_host_unregister",
    "ezpz_specialized_source",
    "ezpz_multi_create", "ezpz_multi_destroy", "ezpz_multi_device_count", "ezpz_multi_device", "ezpz_multi_shard",
    "ezpz_multi_specialize", "ezpz_multi_solve_batch", "ezpz_system_solve_batch_multi",
    "ezpz_debug_call_trace", "ezpz_launch_policy", "ezpz_debug_front_plan", "ezpz_debug_set_stamps", "ezpz_debug_jit_compilations", "ezpz_debug_freedom_exits",
    "ezpz_mixed_create", "ezpz_mixed_destroy", "ezpz_mixed_total_values", "ezpz_mixed_offsets", "ezpz_mixed_solve_device",
    "ezpz_mixed_solve", "ezpz_system_solve_batch_mixed", "ezpz_multi_solve_batch_mixed",
]

_lib = None


def lib():
    """Loads (building first if the sources are newer) libezpz_amd.so.  Raises if it cannot be built/loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("EZPZ_AMD_LIB") or _build.LIB  # EZPZ_AMD_LIB: A/B-test another build of the library
    if path == _build.LIB and os.environ.get("EZPZ_AMD_NO_BUILD") != "1":
        path = _build.build()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m ezpz_amd.build` (needs hipcc)")
    L = C.CDLL(path)
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    L.ezpz_default_config.restype = None
    L.ezpz_default_config.argtypes = [C.POINTER(CConfig)]
    L.ezpz_device_count.restype = C.c_int
    L.ezpz_error_string.restype = C.c_char_p
    L.ezpz_error_string.argtypes = [C.c_int]
    L.ezpz_system_create.restype = C.c_int
    L.ezpz_system_create.argtypes = [vp, sz, sz, C.c_int, u32, C.POINTER(vp), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int64)]
    L.ezpz_system_destroy.restype = None
    L.ezpz_system_destroy.argtypes = [vp]
    L.ezpz_system_info.restype = C.c_int
    L.ezpz_system_info.argtypes = [vp, C.POINTER(CSystemInfo)]
    L.ezpz_solve_batch.restype = C.c_int
    L.ezpz_solve_batch.argtypes = [vp, sz, sz, vp, sz, C.POINTER(CConfig), vp, vp, vp, vp, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int64)]
    L.ezpz_system_freedom_batch.restype = C.c_int
    L.ezpz_system_freedom_batch.argtypes = [vp, vp, sz, vp, vp]
    L.ezpz_system_freedom_batch_device.restype = C.c_int
    L.ezpz_system_freedom_batch_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    L.ezpz_current_device.restype = C.c_int
    L.ezpz_current_device.argtypes = []
    L.ezpz_host_register.restype = C.c_int
    L.ezpz_host_register.argtypes = [vp, sz]
    L.ezpz_host_unregister.restype = C.c_int
    L.ezpz_host_unregister.argtypes = [vp]
    L.ezpz_system_specialize.restype = C.c_int
    L.ezpz_system_specialize.argtypes = [vp, C.c_int]
    L.ezpz_specialized_source.restype = C.c_long
    L.ezpz_specialized_source.argtypes = [vp, sz, sz, C.c_int, C.c_char_p, sz]
    L.ezpz_multi_create.restype = C.c_int
    L.ezpz_multi_create.argtypes = [vp, sz, sz, C.c_uint64, u32, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.ezpz_multi_destroy.restype = None
    L.ezpz_multi_destroy.argtypes = [vp]
    L.ezpz_multi_device_count.restype = C.c_int
    L.ezpz_multi_device_count.argtypes = [vp]
    L.ezpz_multi_device.restype = C.c_int
    L.ezpz_multi_device.argtypes = [vp, C.c_int]
    L.ezpz_multi_shard.restype = None
    L.ezpz_multi_shard.argtypes = [vp, sz, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.ezpz_multi_specialize.restype = C.c_int
    L.ezpz_multi_specialize.argtypes = [vp, C.c_int]
    L.ezpz_multi_solve_batch.restype = C.c_int
    L.ezpz_multi_solve_batch.argtypes = [vp, vp, sz, C.POINTER(CConfig), vp, vp, vp]
    L.ezpz_system_solve_batch_multi.restype = C.c_int
    L.ezpz_system_solve_batch_multi.argtypes = [vp, sz, sz, C.c_uint64, vp, sz, C.POINTER(CConfig), vp, vp]
    L.ezpz_resolve_sides.restype = C.c_int
    L.ezpz_resolve_sides.argtypes = [vp, sz, vp, sz]
    L.ezpz_mixed_create.restype = C.c_int
    L.ezpz_mixed_create.argtypes = [vp, sz, vp, sz, C.POINTER(vp)]
    L.ezpz_mixed_destroy.restype = None
    L.ezpz_mixed_destroy.argtypes = [vp]
    L.ezpz_mixed_total_values.restype = sz
    L.ezpz_mixed_total_values.argtypes = [vp]
    L.ezpz_mixed_offsets.restype = None
    L.ezpz_mixed_offsets.argtypes = [vp, vp]
    L.ezpz_mixed_solve_device.restype = C.c_int
    L.ezpz_mixed_solve_device.argtypes = [vp, vp, C.POINTER(CConfig), vp, vp, vp]
    L.ezpz_mixed_solve.restype = C.c_int
    L.ezpz_mixed_solve.argtypes = [vp, vp, C.POINTER(CConfig), vp, vp]
    L.ezpz_system_solve_batch_mixed.restype = C.c_int
    L.ezpz_system_solve_batch_mixed.argtypes = [vp, sz, vp, vp, sz, C.POINTER(CConfig), vp, vp]
    L.ezpz_multi_solve_batch_mixed.restype = C.c_int
    L.ezpz_multi_solve_batch_mixed.argtypes = [vp, sz, vp, vp, sz, C.POINTER(CConfig), vp, vp]
    L.ezpz_launch_policy.restype = C.c_int
    L.ezpz_launch_policy.argtypes = [C.c_int, C.POINTER(CLaunchPolicy)]
    L.ezpz_debug_call_trace.restype = sz
    L.ezpz_debug_call_trace.argtypes = [vp, sz]
    L.ezpz_debug_set_stamps.restype = None
    L.ezpz_debug_set_stamps.argtypes = [vp]
    L.ezpz_debug_freedom_exits.restype = None
    L.ezpz_debug_freedom_exits.argtypes = [vp]
    L.ezpz_debug_jit_compilations.restype = C.c_uint64
    L.ezpz_debug_jit_compilations.argtypes = []
    L.ezpz_debug_front_plan.restype = C.c_long
    L.ezpz_debug_front_plan.argtypes = [vp, sz, sz, u32, u32, C.c_uint64, vp, sz, vp]
    L.ezpz_cache_clear.restype = None
    L.ezpz_cache_clear.argtypes = []
    L.ezpz_analyze.restype = C.c_int
    L.ezpz_analyze.argtypes = [vp, sz, sz, C.POINTER(CSystemInfo), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.ezpz_system_eval_batch.restype = C.c_int
    L.ezpz_system_eval_batch.argtypes = [vp, vp, sz, vp, vp, vp]
    L.ezpz_system_jacobian_pattern.restype = C.c_int
    L.ezpz_system_jacobian_pattern.argtypes = [vp, vp, vp]
    L.ezpz_system_solve_batch_device.restype = C.c_int
    L.ezpz_system_solve_batch_device.argtypes = [vp, vp, sz, C.POINTER(CConfig), vp, vp, vp, vp, u32, vp]
    L.ezpz_system_solve_batch.restype = C.c_int
    L.ezpz_system_solve_batch.argtypes = [vp, vp, sz, C.POINTER(CConfig), vp, vp, vp, vp, u32]
    L.ezpz_solve_inner.restype = C.c_int
    L.ezpz_solve_inner.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(CConfig), vp, vp, vp, sz, C.POINTER(COutcome)]
    L.ezpz_solve.restype = C.c_int
    L.ezpz_solve.argtypes = [vp, sz, vp, vp, sz, C.POINTER(CConfig), vp, vp, vp, sz, C.POINTER(COutcome)]
    L.ezpz_solve_analysis.restype = C.c_int
    L.ezpz_solve_analysis.argtypes = list(L.ezpz_solve.argtypes) + [vp, C.POINTER(C.c_uint64)]
    L.ezpz_problem_parse.restype = C.c_int
    L.ezpz_problem_parse.argtypes = [C.c_char_p, sz, C.POINTER(vp), C.c_char_p, sz]
    L.ezpz_problem_destroy.restype = None
    L.ezpz_problem_destroy.argtypes = [vp]
    L.ezpz_problem_num_constraints.restype = sz
    L.ezpz_problem_num_constraints.argtypes = [vp]
    L.ezpz_problem_num_vars.restype = sz
    L.ezpz_problem_num_vars.argtypes = [vp]
    L.ezpz_problem_constraints.restype = vp
    L.ezpz_problem_constraints.argtypes = [vp]
    L.ezpz_problem_guesses.restype = vp
    L.ezpz_problem_guesses.argtypes = [vp]
    L.ezpz_problem_num_labels.restype = sz
    L.ezpz_problem_num_labels.argtypes = [vp, C.c_int]
    L.ezpz_problem_label.restype = C.c_char_p
    L.ezpz_problem_label.argtypes = [vp, C.c_int, sz]
    _lib = L
    return L
