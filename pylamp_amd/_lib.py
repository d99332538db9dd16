"""ctypes binding of libpylamp_hip.so (C ABI: include/pylamp_hip.h).

The library is built in-tree by __graft_entry__.build() / pylamp_amd/csrc/Makefile.  If it
is missing or cannot be loaded every call fails loudly — there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpylamp_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class SolveStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("rel_residual", C.c_double),
                ("solve_ms", C.c_double), ("operator_applies", C.c_int), ("precond_applies", C.c_int),
                ("used_direct", C.c_int), ("reserved_", C.c_int), ("error_estimate", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class StepConfig(C.Structure):
    _fields_ = [("do_heatdiff", C.c_int), ("do_subgrid_heatdiff", C.c_int), ("tdep_rho", C.c_int),
                ("tdep_eta", C.c_int), ("etamin", C.c_double), ("etamax", C.c_double), ("tref", C.c_double),
                ("tstep_adv_max", C.c_double), ("tstep_adv_min", C.c_double), ("tstep_dif_max", C.c_double),
                ("tstep_dif_min", C.c_double), ("tstep_modifier", C.c_double), ("bcstokes", C.c_int * 4),
                ("bcheat", C.c_int * 4), ("bcheatvals", C.c_double * 4), ("stokes_rtol", C.c_double),
                ("heat_rtol", C.c_double), ("stokes_maxit", C.c_int), ("heat_maxit", C.c_int),
                ("length", C.c_double * 2), ("tracdens", C.c_int), ("tracdens_min", C.c_int),
                ("inject_seed", C.c_uint64), ("surface_stabilization", C.c_int), ("surfstab_theta", C.c_double),
                ("surfstab_tstep", C.c_double), ("inject_unique_ids", C.c_int), ("tracs_fence_disabled", C.c_int)]


class StepReport(C.Structure):
    _fields_ = [("tstep", C.c_double), ("limiter", C.c_int), ("tstep_heat", C.c_double),
                ("tstep_stokes", C.c_double), ("stokes", SolveStats), ("heat", SolveStats),
                ("ms_props", C.c_double), ("ms_scatter", C.c_double), ("ms_stokes", C.c_double),
                ("ms_heat", C.c_double), ("ms_gather", C.c_double), ("ms_advect", C.c_double),
                ("ms_sort", C.c_double), ("ms_total", C.c_double), ("ntrac", C.c_int64), ("ninjected", C.c_int64),
                ("stokes_resolves", C.c_int), ("nremoved", C.c_int64)]


# name -> (restype, argtypes); every symbol declared in include/pylamp_hip.h
SIGNATURES = {
    "pl_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]),
    "pl_destroy": (None, [C.c_void_p]),
    "pl_last_error": (C.c_char_p, [C.c_void_p]),
    "pl_sync": (C.c_int, [C.c_void_p]),
    "pl_abi_layout": (C.c_int, [C.POINTER(C.c_size_t)]),
    "pl3_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    "pl3_destroy": (None, [C.c_void_p]),
    "pl3_set_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pl3_local_block": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "pl3_comm_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "pl3_last_error": (C.c_char_p, [C.c_void_p]),
    "pl3_stokes_set_coeffs": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "pl3_stokes_set_wall_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "pl3_stokes_get_scaling": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "pl3_stokes_apply": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "pl3_stokes_rhs": (C.c_int, [C.c_void_p, c_double_p]),
    "pl3_stokes_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_double, C.c_int, C.POINTER(SolveStats)]),
    "pl3_stokes_apply_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p]),
    "pl3_stokes_mg_info": (C.c_int, [C.c_void_p, c_int_p, c_double_p, C.c_int]),
    "pl3_heat_set_coeffs": (C.c_int, [C.c_void_p] + [c_double_p] * 10 + [c_int_p, c_double_p, C.c_double]),
    "pl3_heat_apply": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "pl3_heat_rhs": (C.c_int, [C.c_void_p, c_double_p]),
    "pl3_heat_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_double, C.c_int, C.POINTER(SolveStats)]),
    "pl3_get_solution": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "pl_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, c_int_p, C.POINTER(C.c_size_t)]),
    "pl_set_comm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pl_set_comm_2d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pl_local_rows": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "pl_local_block": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "pl_local_group_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "pl_local_group_destroy": (None, [C.c_void_p]),
    "pl_local_group_abort": (None, [C.c_void_p]),
    "pl_local_group_enter": (None, [C.c_void_p]),
    "pl_local_group_leave": (None, [C.c_void_p, C.c_void_p]),
    "pl_set_comm_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pl_comm_info": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p]),
    "pl_comm_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "pl_comm_times": (C.c_int, [C.c_void_p, c_double_p, C.c_int]),
    "pl_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pl_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pl_dev_add": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int64]),
    "pl_timer_start": (C.c_int, [C.c_void_p]),
    "pl_timer_stop_ms": (C.c_int, [C.c_void_p, c_double_p]),
    "pl_stokes_set_coeffs": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_int_p, C.c_int,
                                       C.c_double, C.c_double]),
    "pl_stokes_get_scaling": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "pl_stokes_apply": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "pl_stokes_rhs": (C.c_int, [C.c_void_p, c_double_p]),
    "pl_stokes_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_double, C.c_int,
                                  C.POINTER(SolveStats)]),
    "pl_stokes_apply_bench": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "pl_stokes_apply_scaled_bench": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "pl_stream_triad_bench": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, c_double_p]),
    "pl_stokes_precond_apply": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "pl_stokes_mg_info": (C.c_int, [C.c_void_p, c_int_p, c_double_p, C.c_int]),
    "pl_stokes_mg_precision": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "pl_stokes_set_mg_precision": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong]),
    "pl_stokes_sweep_bench": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "pl_heat_set_coeffs": (C.c_int, [C.c_void_p] + [c_double_p] * 8 + [c_int_p, c_double_p, C.c_double]),
    "pl_heat_apply": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "pl_heat_rhs": (C.c_int, [C.c_void_p, c_double_p]),
    "pl_heat_solve": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_double, C.c_int, C.POINTER(SolveStats)]),
    "pl_heat_apply_bench": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "pl_trac2grid": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p, C.c_int64, C.c_int, c_int_p,
                               C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(c_double_p)]),
    "pl_trac2grid_rect": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p, C.c_int64, C.c_int, c_int_p,
                                    c_double_p, c_double_p, C.POINTER(c_double_p)]),
    "pl_mic_set_search": (C.c_int, [C.c_void_p, C.c_int]),
    "pl_grid2trac": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, C.c_int, C.POINTER(c_double_p), C.c_int,
                               C.c_int, c_double_p, c_double_p, C.c_int, C.c_double, C.c_int, c_double_p,
                               C.c_int64, C.POINTER(C.c_int64)]),
    "pl_rk4": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, C.c_int, C.c_int, c_double_p, c_double_p,
                         c_double_p, c_double_p, C.c_double, c_double_p, c_double_p]),
    "pl_tracers_upload": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p]),
    "pl_tracers_download": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p]),
    "pl_tracers_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "pl_tracers_census": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), c_int_p, c_int_p]),
    "pl_tracers_layout": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "pl_step": (C.c_int, [C.c_void_p, C.POINTER(StepConfig), C.c_int, C.POINTER(StepReport)]),
    "pl_resident_scatter": (C.c_int, [C.c_void_p, C.POINTER(StepConfig), C.c_int]),
    "pl_resident_temp_to_tracers": (C.c_int, [C.c_void_p, C.POINTER(StepConfig), C.c_int, c_double_p, C.c_double]),
    "pl_resident_rk4": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_double, C.c_int, c_double_p]),
    "pl_get_field": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p]),
    "pl_get_tracer_velocity": (C.c_int, [C.c_void_p, C.c_int64, c_double_p]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Exception("pylamp_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)" % LIB_PATH)
    # torch ships its own libamdhip64.so.7; importing it first makes the dynamic loader bind
    # this library to the SAME HIP runtime (one runtime per process) when torch is in use.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError here == missing export
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def f64(a):
    """C-contiguous float64 view/copy."""
    return np.ascontiguousarray(a, dtype=np.float64)


def check(ctx_handle, rc):
    if rc != 0:
        msg = load().pl_last_error(ctx_handle)
        raise Exception(msg.decode() if msg else "libpylamp_hip error %d" % rc)
