"""ctypes binding of ``libfiksi_amd.so`` (C ABI: ``include/fiksi_amd.h``, ``include/fiksi_amd_builder.h``).

The shared library is built in-tree by ``__graft_entry__.build()`` (``make -C fiksi_amd/csrc``). There is no
fallback: if it is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FIKSI_AMD_LIBRARY: another build of the same C ABI — the host-only sanitizer build (`make -C fiksi_amd/csrc asan`,
# tests/test_host_sanitizers.py); every device entry point of that one reports FX_ERR_NO_DEVICE
LIB_PATH = os.environ.get("FIKSI_AMD_LIBRARY") or os.path.join(_HERE, "libfiksi_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C fiksi_amd/csrc`). fiksi_amd has no pure-Python or CPU fallback."
    )



def _share_hip_runtime_with_torch() -> str:
    """One process can hold only one HIP/ROCr runtime. PyTorch-ROCm wheels bundle their own
    (``torch/lib/libamdhip64.so``, SONAME ``libamdhip64.so.7`` like the system one), and this library is
    linked against ``libamdhip64.so.7``: if torch is imported first the loader hands us torch's copy and
    all is well, but if we come first the system copy is loaded, a later ``import torch`` brings its own
    second runtime, and whichever initialises the GPU second finds no device. So when torch is installed
    its runtime is loaded here, ahead of ours, and both sides share it in either import order.
    ``FIKSI_AMD_HIP_RUNTIME=system`` opts out (processes that never import torch)."""
    import sys

    if os.environ.get("FIKSI_AMD_HIP_RUNTIME", "auto") == "system":
        return "system"
    if "torch" in sys.modules:
        return "torch (already imported)"
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return "system"
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if not os.path.exists(path):
            return "system"
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        return "torch (preloaded)"
    except Exception:  # a broken torch install must not break this library
        return "system"


HIP_RUNTIME = _share_hip_runtime_with_torch()
lib = C.CDLL(LIB_PATH)

u8p, u16p, u32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64))
f64p = C.POINTER(C.c_double)


class FxBatch(C.Structure):
    _fields_ = [
        ("n_systems", C.c_uint32),
        ("var_off", C.c_void_p),
        ("expr_off", C.c_void_p),
        ("vars", C.c_void_p),
        ("var_fixed", C.c_void_p),
        ("expr_tag", C.c_void_p),
        ("expr_idx", C.c_void_p),
        ("expr_param", C.c_void_p),
        ("var_comp", C.c_void_p),
        ("expr_comp", C.c_void_p),
    ]


class FxLmOpts(C.Structure):
    _fields_ = [
        ("lambda0", C.c_double), ("sse_tol", C.c_double), ("step_tol", C.c_double), ("ftol", C.c_double),
        ("accept_factor", C.c_double), ("reject_factor", C.c_double), ("singular_factor", C.c_double),
        ("lambda_min", C.c_double), ("max_outer", C.c_uint32), ("max_trials", C.c_uint32),
        ("solver", C.c_uint32), ("precision", C.c_uint32),
    ]


class FxSolvingOpts(C.Structure):
    _fields_ = [
        ("optimizer", C.c_uint32), ("decomposer", C.c_uint32), ("perturb", C.c_uint32), ("plan_budget", C.c_uint32),
        ("lm", FxLmOpts),
    ]


class FxResult(C.Structure):
    _fields_ = [
        ("accepted", C.c_uint32), ("trials", C.c_uint32), ("exit", C.c_uint32), ("ncomp", C.c_uint32),
        ("scale", C.c_double), ("sse0", C.c_double), ("sse", C.c_double), ("sse_unscaled", C.c_double),
    ]


# every entry point the headers declare: (name, restype, argtypes)
_vp = C.c_void_p
SIGNATURES = [
    ("fx_abi_version", C.c_int, []),
    ("fx_last_error", C.c_char_p, []),
    ("fx_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("fx_ctx_create", C.c_int, [C.POINTER(_vp), C.c_int]),
    ("fx_ctx_destroy", None, [_vp]),
    ("fx_ctx_set_routing", C.c_int, [_vp, C.c_int, C.c_uint32]),
    ("fx_ctx_set_presort", C.c_int, [_vp, C.c_int, C.c_uint32]),
    ("fx_ctx_set_hold_passes", C.c_int, [_vp, C.c_uint32]),
    ("fx_ctx_set_one_structure_builds", C.c_int, [_vp, C.c_int]),
    ("fx_ctx_set_ladder", C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, C.c_int]),
    ("fx_ctx_set_wide_routing", C.c_int, [_vp, C.c_int]),
    ("fx_ctx_set_sparse_fronts", C.c_int, [_vp, C.c_int, C.c_uint32]),
    ("fx_ctx_set_host_threads", C.c_int, [_vp, C.c_uint32]),
    ("fx_ctx_set_batch_hints", C.c_int, [_vp, C.c_uint32]),
    ("fx_host_register", C.c_int, [_vp, _vp, C.c_size_t]),
    ("fx_host_unregister", C.c_int, [_vp, _vp]),
    ("fx_ctx_synchronize", C.c_int, [_vp]),
    ("fx_ctx_device_name", C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    ("fx_lm_opts_default", None, [C.POINTER(FxLmOpts)]),
    ("fx_lm_opts_default_f32", None, [C.POINTER(FxLmOpts)]),
    ("fx_solving_opts_default", None, [C.POINTER(FxSolvingOpts)]),
    ("fx_batch_validate", C.c_int, [C.POINTER(FxBatch)]),
    ("fx_jacobian_structure", C.c_int, [C.POINTER(FxBatch), u64p, _vp, _vp]),
    ("fx_batch_upload", C.c_int, [_vp, C.POINTER(FxBatch), C.POINTER(_vp)]),
    ("fx_batch_free", None, [_vp, _vp]),
    ("fx_batch_set_vars", C.c_int, [_vp, _vp, _vp]),
    ("fx_batch_set_params", C.c_int, [_vp, _vp, _vp]),
    ("fx_batch_schedule_by_last_solve", C.c_int, [_vp, _vp, C.c_int]),
    ("fx_batch_get_vars", C.c_int, [_vp, _vp, _vp]),
    ("fx_batch_get_results", C.c_int, [_vp, _vp, _vp]),
    ("fx_batch_nnz", C.c_uint64, [_vp]),
    ("fx_system_solve_device", C.c_int, [_vp, _vp, C.POINTER(FxSolvingOpts)]),
    ("fx_lm_solve_device", C.c_int, [_vp, _vp, C.POINTER(FxLmOpts)]),
    ("fx_eval_residual_jacobian_device", C.c_int, [_vp, _vp, C.c_int]),
    ("fx_eval_residual_device", C.c_int, [_vp, _vp, C.c_int]),
    ("fx_batch_get_residuals", C.c_int, [_vp, _vp, _vp]),
    ("fx_batch_get_jacobian_values", C.c_int, [_vp, _vp, _vp]),
    ("fx_debug_phase_cycles", C.c_int, [_vp, _vp, C.POINTER(FxSolvingOpts), u64p]),
    ("fx_debug_solve_route", C.c_int, [_vp, _vp, C.POINTER(FxSolvingOpts), C.POINTER(C.c_int)]),
    ("fx_debug_grouped_build", C.c_int, [_vp, _vp, C.POINTER(FxSolvingOpts), C.POINTER(C.c_int)]),
    ("fx_timer_begin", C.c_int, [_vp]),
    ("fx_timer_end", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("fx_system_solve_batch", C.c_int, [_vp, C.POINTER(FxBatch), C.POINTER(FxSolvingOpts), _vp]),
    ("fx_lm_solve_batch", C.c_int, [_vp, C.POINTER(FxBatch), C.POINTER(FxLmOpts), _vp]),
    ("fx_system_solve_batch_multi", C.c_int, [C.POINTER(_vp), C.c_uint32, C.POINTER(FxBatch), C.POINTER(FxSolvingOpts), _vp, _vp]),
    ("fx_eval_residual_jacobian", C.c_int, [_vp, C.POINTER(FxBatch), _vp, _vp]),
    ("fx_constraint_residuals", C.c_int, [_vp, C.POINTER(FxBatch), _vp]),
    ("fx_system_prepare_batch", C.c_int, [_vp, C.POINTER(FxBatch), C.c_uint32, _vp, _vp, _vp]),
    ("fx_cluster_solve_batch", C.c_int, [_vp, C.POINTER(FxBatch), C.POINTER(FxLmOpts), _vp]),
    ("fx_pose_transform_points", C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, _vp, C.c_uint32]),
    ("fx_unscale_vars", C.c_int, [_vp, C.c_double, _vp, _vp, _vp, C.c_uint32]),
    ("fx_unscale_vars_strided", C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    ("fx_analyze_batch", C.c_int, [_vp, C.POINTER(FxBatch), _vp]),
    ("fx_eval_residual_dense_jacobian", C.c_int, [_vp, C.POINTER(FxBatch), _vp, _vp, _vp, _vp]),
    ("fx_single_pass_blocks", C.c_int, [C.POINTER(FxBatch), C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("fx_atan2_cr_batch", None, [C.c_uint64, _vp, _vp, _vp]),
    ("fx_qr_symbolic", C.c_int, [C.c_int32, C.c_int32, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, C.c_int32]),
    # builder
    ("fxs_system_new", C.c_int, [C.POINTER(_vp)]),
    ("fxs_system_free", None, [_vp]),
    ("fxs_system_id", C.c_uint32, [_vp]),
    ("fxs_num_elements", C.c_uint32, [_vp]),
    ("fxs_num_constraints", C.c_uint32, [_vp]),
    ("fxs_num_variables", C.c_uint32, [_vp]),
    ("fxs_num_expressions", C.c_uint32, [_vp]),
    ("fxs_length_create", C.c_int64, [_vp, C.c_double]),
    ("fxs_point_create", C.c_int64, [_vp, C.c_double, C.c_double]),
    ("fxs_line_create", C.c_int64, [_vp, C.c_uint32, C.c_uint32]),
    ("fxs_circle_create", C.c_int64, [_vp, C.c_uint32, C.c_uint32]),
    ("fxs_element_tag_of", C.c_int, [_vp, C.c_uint32]),
    ("fxs_element_fix", C.c_int, [_vp, C.c_uint32]),
    ("fxs_element_unfix", C.c_int, [_vp, C.c_uint32]),
    ("fxs_element_get_value", C.c_int, [_vp, C.c_uint32, f64p]),
    ("fxs_point_update_value", C.c_int, [_vp, C.c_uint32, C.c_double, C.c_double]),
    ("fxs_length_update_value", C.c_int, [_vp, C.c_uint32, C.c_double]),
    ("fxs_constraint_create", C.c_int64, [_vp, C.c_int, u32p, C.c_uint32, C.c_double]),
    ("fxs_constraint_tag_of", C.c_int, [_vp, C.c_uint32]),
    ("fxs_constraint_valency", C.c_int, [C.c_int]),
    ("fxs_constraint_update_parameter", C.c_int, [_vp, C.c_uint32, C.c_double]),
    ("fxs_components", C.c_int, [_vp, u32p, _vp, _vp]),
    ("fxs_export_graph", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("fxs_recursive_plan", C.c_int, [_vp, C.c_uint64, _vp, C.c_uint32, u32p, u32p]),
    ("fxs_flatten", C.c_int, [C.POINTER(_vp), C.c_uint32, C.POINTER(_vp)]),
    ("fxs_flat_batch", C.POINTER(FxBatch), [_vp]),
    ("fxs_flat_free", None, [_vp]),
    ("fxs_flat_scatter", C.c_int, [_vp, C.POINTER(_vp), C.c_uint32]),
    ("fxs_system_solve", C.c_int, [_vp, _vp, C.POINTER(FxSolvingOpts), C.POINTER(FxResult)]),
    ("fxs_systems_solve", C.c_int, [C.POINTER(_vp), C.c_uint32, _vp, C.POINTER(FxSolvingOpts), _vp]),
    ("fxs_system_constraint_residuals", C.c_int, [_vp, _vp, _vp]),
    ("fxs_system_analyze", C.c_int, [_vp, _vp, _vp, u32p]),
]

for _name, _res, _args in SIGNATURES:
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class FiksiError(RuntimeError):
    """A C-ABI call returned a negative fx_status."""

    def __init__(self, code: int, where: str):
        msg = lib.fx_last_error()
        super().__init__(f"{where} failed with fx_status {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code: int, where: str) -> int:
    if code < 0:
        raise FiksiError(int(code), where)
    return int(code)
