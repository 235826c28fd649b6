"""ctypes binding of libpathfit.so (include/pathfit.h).

There is no CPU fallback: if the HIP library or a GPU is missing, every entry
point fails loudly (``PathfitError``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PF_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libpathfit.so")   # PF_LIB: A/B builds
_LIB = None


class PathfitError(RuntimeError):
    pass


class ScoreParams(C.Structure):
    _fields_ = [("variant", C.c_int32), ("restrict_policy", C.c_int32), ("w_turn", C.c_double),
                ("w_safe", C.c_double), ("min_safe", C.c_double), ("diag_pen", C.c_double)]


class MaacoParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("alpha", "beta", "rho", "Q", "a_turn_coef", "wh_max", "wh_min",
                                          "k_h_adaptive", "q0_initial", "C0_initial_pheromone")] + \
               [("num_iterations", C.c_int32), ("start", C.c_int32), ("target", C.c_int32)]


class MpaParams(C.Structure):
    _fields_ = [("P_const", C.c_double), ("levy_beta", C.c_double), ("levy_sigma", C.c_double),
                ("FADs_rate", C.c_double), ("num_predators", C.c_int32), ("start", C.c_int32),
                ("target", C.c_int32), ("allow_diag", C.c_int32), ("restrict_corner", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("pops", "pushes", "nbr_examined", "path_cells", "steps", "candidates",
                                         "decrease_keys", "overflow_agents", "pruned_rebuilds", "settled_searches",
                                         "sequential_searches")]


# every symbol include/pathfit.h declares: (name, restype, argtypes)
_vp, _i32, _i64, _u64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
SYMBOLS = {
    "pf_version": (C.c_char_p, []),
    "pf_device_count": (C.c_int, []),
    "pf_create": (C.c_int, [_vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "pf_destroy": (None, [_vp]),
    "pf_update_grid": (C.c_int, [_vp, _vp]),
    "pf_last_error": (C.c_char_p, [_vp]),
    "pf_stream": (_vp, [_vp]),
    "pf_sync": (C.c_int, [_vp]),
    "pf_dev_alloc": (C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    "pf_dev_free": (C.c_int, [_vp, _vp]),
    "pf_h2d": (C.c_int, [_vp, _vp, _vp, _i64]),
    "pf_d2h": (C.c_int, [_vp, _vp, _vp, _i64]),
    "pf_d2d": (C.c_int, [_vp, _vp, _vp, _i64]),
    "pf_memset": (C.c_int, [_vp, _vp, _i32, _i64]),
    "pf_get_counters": (C.c_int, [_vp, C.POINTER(Counters)]),
    "pf_last_kernel_ms": (C.c_float, [_vp]),
    "pf_astar_batch": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "pf_score_batch": (C.c_int, [_vp, C.POINTER(ScoreParams), _i32, _i32, _vp, _vp, _vp]),
    "pf_decode_batch": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp,
                                  C.POINTER(ScoreParams), _vp]),
    "pf_pso_update": (C.c_int, [_vp, _i32, _i32, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _u64, _u64, _u64]),
    "pf_pso_pbest": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pf_pso_update_keep": (C.c_int, [_vp, _i32, _i32, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    "pf_pso_commit": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pf_maaco_setup": (C.c_int, [_vp, C.POINTER(MaacoParams)]),
    "pf_maaco_walk_batch": (C.c_int, [_vp, _i32, _u64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pf_maaco_evaporate": (C.c_int, [_vp]),
    "pf_maaco_deposit": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "pf_maaco_clip": (C.c_int, [_vp, _dbl]),
    "pf_maaco_update": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _dbl]),
    "pf_maaco_iterate": (C.c_int, [_vp, _i32, _u64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _vp]),
    "pf_maaco_best_path": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32)]),
    "pf_maaco_get_pheromone": (C.c_int, [_vp, _vp]),
    "pf_maaco_set_pheromone": (C.c_int, [_vp, _vp]),
    "pf_maaco_tau_dev": (_vp, [_vp]),
    "pf_maaco_best_scan": (C.c_int, [_i32, _vp, _vp, _i32, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_i32)]),
    "pf_ga_select": (C.c_int, [_u64, _i32, _i32, _i32, _vp, _vp]),
    "pf_ga_random_chromosomes": (C.c_int, [_u64, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "pf_ga_breed": (C.c_int, [_u64, _i32, _i32, _i32, _dbl, _dbl, _vp, _i32, _i32, _vp, _vp]),
    "pf_mpa_setup": (C.c_int, [_vp, C.POINTER(MpaParams), C.POINTER(ScoreParams)]),
    "pf_mpa_phase_batch": (C.c_int, [_vp, _i32, _dbl, _i32, _u64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                     _vp, _vp, _vp, _vp, _vp]),
    "pf_mpa_fads_batch": (C.c_int, [_vp, _dbl, _i32, _u64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pf_mpa_iter_batch": (C.c_int, [_vp, _i32, _dbl, _i32, _u64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pf_mpa_rebuild_batch": (C.c_int, [_vp, _i32, _u64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp]),
    "pf_set_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "pf_selftest_sqrt": (C.c_int, [_vp, _i32, _vp, _vp]),
    "pf_selftest_rng": (C.c_int, [_vp, _u64, _u64, _u64, _u64, _vp, _vp, _vp]),
    "pf_ga_select_dev": (C.c_int, [_vp, _u64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pf_ga_breed_dev": (C.c_int, [_vp, _u64, _i32, _i32, _i32, _dbl, _dbl, _vp, _vp, _i32, _i32, _vp]),
    "pf_ga_assemble_dev": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pf_sort_order_by_key": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp]),
    "pf_gather_col": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp]),
    "pf_sorted_head": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "pf_vec_add_f64": (C.c_int, [_vp, _i32, _vp, _vp, _dbl, _vp]),
    "pf_mpa_elite_buf": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "pf_mpa_pick_elite": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32]),
    "pf_mpa_local_view": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "pf_maaco_deposit_begin": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "pf_maaco_deposit_cells": (C.c_int, [_vp, _i32, _i32]),
    "pf_maaco_best_dev": (C.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "pf_comm_unique_id": (C.c_int, [_vp]),
    "pf_comm_init": (C.c_int, [_vp, _i32, _i32, _vp]),
    "pf_comm_destroy": (C.c_int, [_vp]),
    "pf_comm_rank": (C.c_int, [_vp]),
    "pf_comm_world": (C.c_int, [_vp]),
    "pf_comm_all_gather": (C.c_int, [_vp, _vp, _vp, _i64]),
    "pf_comm_broadcast": (C.c_int, [_vp, _vp, _i64, _i32]),
    "pf_comm_all_reduce_f64": (C.c_int, [_vp, _vp, _i64, _i32]),
    "pf_comm_send": (C.c_int, [_vp, _vp, _i64, _i32]),
    "pf_comm_recv": (C.c_int, [_vp, _vp, _i64, _i32]),
    "pf_comm_sendrecv": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i32]),
    "pf_pso_pbest_paths": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pf_pso_scan": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _dbl, _i32, C.POINTER(_i32), C.POINTER(_dbl), C.POINTER(_i32)]),
    "pf_d2h_counts": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "pf_span_begin": (C.c_int, [_vp]),
    "pf_span_end": (C.c_int, [_vp]),
    "pf_span_total": (C.c_int, [_vp, C.POINTER(_dbl), C.POINTER(_i64), _i32]),
    "pf_selftest_mpa_targets": (C.c_int, [_vp, _u64, _i32, _i32, _dbl, _dbl, _dbl, _vp, _vp, _vp, C.POINTER(_i64)]),
    "pf_mpa_doubts_resolved": (C.c_longlong, [_vp]),
    "pf_mpa_memory": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}


def so_path():
    return _SO


def lib():
    """Load libpathfit.so; raise PathfitError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise PathfitError(
                f"{_SO} not found: build it with `python maaco-path-planing_amd/build.py` "
                "(there is no CPU fallback for the population-fitness hot path)")
        L = C.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)      # AttributeError here == the library misses a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
