"""ctypes binding of libecne_hip.so (C ABI declared in include/ecne.h).

The library is the product: if it is missing this module raises — there is no Python or CPU
implementation of the propagation rules to fall back to.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, os.environ.get("ECNE_LIB", "libecne_hip.so"))      # (ECNE_LIB: a developer build beside the product, e.g. the -DECNE_JITTER soak library)


class Info(C.Structure):
    _fields_ = [("field_size", C.c_uint32), ("n_wires", C.c_uint32), ("n_pub_out", C.c_uint32),
                ("n_pub_in", C.c_uint32), ("n_prv_in", C.c_uint32), ("n_constraints", C.c_uint32),
                ("n_labels", C.c_uint64), ("nnz", C.c_uint64 * 3), ("n_vars", C.c_int64)]


class SystemInfo(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_rows_main", C.c_int64), ("n_vars", C.c_int64),
                ("n_specials", C.c_int64), ("n_known", C.c_int64), ("n_targets", C.c_int64),
                ("nnz", C.c_uint64 * 3)]


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("secp_solve", C.c_int32), ("debug", C.c_int32),
                ("queue_mode", C.c_int32), ("stream", C.c_void_p)]


class Summary(C.Structure):
    _fields_ = [("status", C.c_int32), ("function_good", C.c_int32),
                ("unique_nontrivial", C.c_int64), ("n_nontrivial", C.c_int64),
                ("unique_targets", C.c_int64), ("n_targets", C.c_int64),
                ("successful_steps", C.c_int64), ("outer_iterations", C.c_int64),
                ("pops", C.c_int64), ("num_unique", C.c_int64),
                ("rule_hits", C.c_int64 * 16), ("n_rows", C.c_int64), ("n_vars", C.c_int64), ("pop_nnz", C.c_int64),
                ("device_ms", C.c_double), ("classify_ms", C.c_double), ("queue_ms", C.c_double * 8), ("multi_ms", C.c_double * 8), ("phase_ms", C.c_double * 8),
                ("sched", C.c_int64 * 16), ("team", C.c_int64 * 4)]


# every symbol include/ecne.h declares
EXPORTS = [
    "ecne_r1cs_load", "ecne_r1cs_info", "ecne_r1cs_csr", "ecne_r1cs_io", "ecne_r1cs_free",
    "ecne_system_from_r1cs", "ecne_abstract", "ecne_system_info_get", "ecne_system_special",
    "ecne_system_rows", "ecne_system_free", "ecne_solve", "ecne_solve_batch", "ecne_result_summary", "ecne_result_summaries", "ecne_results_free", "ecne_result_states",
    "ecne_result_bad_rows", "ecne_result_free", "ecne_classify", "ecne_fp_selftest", "ecne_fp_sqrt",
    "ecne_device_count", "ecne_strerror", "ecne_version",
    "ecne_set_host_threads", "ecne_system_set_io", "ecne_system_clear_specials", "ecne_system_add_special", "ecne_system_io", "ecne_system_report_order", "ecne_abstract_stats", "ecne_fp_solve_quadratic", "ecne_set_frontend", "ecne_frontend_stats", "ecne_system_dict_rows", "ecne_debug_static_array", "ecne_system_set_secp_solve", "ecne_result_digest", "ecne_set_split", "ecne_system_split_info", "ecne_warmup",
]

_L = None


def lib():
    global _L
    if _L is not None:
        return _L
    if not os.path.exists(SO):
        raise ImportError(
            "libecne_hip.so is not built (%s). Build it with `python -m ecneproject_amd.build` "
            "(hipcc --offload-arch=gfx950). ecneproject_amd has no CPU fallback." % SO)
    L = C.CDLL(SO)
    vp, i64p = C.c_void_p, C.POINTER(C.c_int64)
    L.ecne_r1cs_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.ecne_r1cs_info.argtypes = [vp, C.POINTER(Info)]
    L.ecne_r1cs_csr.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32)),
                                C.POINTER(C.POINTER(C.c_uint64))]
    L.ecne_r1cs_io.argtypes = [vp, C.POINTER(i64p), C.POINTER(C.c_size_t), C.POINTER(i64p), C.POINTER(C.c_size_t)]
    L.ecne_r1cs_free.argtypes = [vp]
    L.ecne_r1cs_free.restype = None
    L.ecne_system_from_r1cs.argtypes = [vp, C.POINTER(vp)]
    L.ecne_abstract.argtypes = [vp, vp, C.c_char_p]
    L.ecne_system_info_get.argtypes = [vp, C.POINTER(SystemInfo)]
    L.ecne_system_special.argtypes = [vp, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(i64p), C.POINTER(C.c_size_t),
                                      C.POINTER(i64p), C.POINTER(C.c_size_t)]
    L.ecne_system_rows.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)),
                                   C.POINTER(C.POINTER(C.c_uint64))]
    L.ecne_system_free.argtypes = [vp]
    L.ecne_system_free.restype = None
    L.ecne_solve.argtypes = [vp, C.POINTER(Opts), C.POINTER(vp)]
    L.ecne_solve_batch.argtypes = [C.POINTER(vp), C.c_size_t, C.POINTER(Opts), C.POINTER(vp)]
    L.ecne_result_summary.argtypes = [vp, C.POINTER(Summary)]
    L.ecne_result_states.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint64)),
                                     C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_int32)),
                                     C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint64))]
    L.ecne_result_bad_rows.argtypes = [vp, C.POINTER(i64p), C.POINTER(C.c_size_t)]
    L.ecne_result_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ecne_result_free.argtypes = [vp]
    L.ecne_result_free.restype = None
    L.ecne_result_summaries.argtypes = [C.POINTER(vp), C.c_size_t, C.POINTER(Summary)]
    L.ecne_results_free.argtypes = [C.POINTER(vp), C.c_size_t]
    L.ecne_results_free.restype = None
    L.ecne_classify.argtypes = [vp, C.POINTER(Opts), vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.ecne_fp_selftest.argtypes = [C.c_int, C.c_int, C.c_size_t, vp, vp, vp]
    L.ecne_fp_sqrt.argtypes = [vp, vp]
    L.ecne_device_count.argtypes = []
    L.ecne_strerror.argtypes = [C.c_int]
    L.ecne_strerror.restype = C.c_char_p
    L.ecne_version.argtypes = []
    L.ecne_version.restype = C.c_char_p
    L.ecne_set_host_threads.argtypes = [C.c_int]
    L.ecne_system_set_io.argtypes = [vp, i64p, C.c_size_t, i64p, C.c_size_t]
    L.ecne_system_clear_specials.argtypes = [vp]
    L.ecne_system_add_special.argtypes = [vp, C.c_char_p, i64p, C.c_size_t, i64p, C.c_size_t]
    L.ecne_system_report_order.argtypes = [vp, C.c_int64, C.POINTER(i64p), C.POINTER(C.c_size_t)]
    L.ecne_abstract_stats.argtypes = [C.POINTER(C.c_double)]
    L.ecne_set_frontend.argtypes = [C.c_int]
    L.ecne_set_split.argtypes = [C.c_int]
    L.ecne_system_split_info.argtypes = [vp, C.POINTER(C.c_double)]
    L.ecne_system_set_secp_solve.argtypes = [vp, C.c_int]
    L.ecne_system_dict_rows.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    L.ecne_debug_static_array.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ecne_frontend_stats.argtypes = [C.POINTER(C.c_double)]
    L.ecne_fp_solve_quadratic.argtypes = [vp, vp, vp, C.c_int, vp, C.POINTER(C.c_int)]
    L.ecne_system_io.argtypes = [vp, C.POINTER(i64p), C.POINTER(C.c_size_t), C.POINTER(i64p), C.POINTER(C.c_size_t)]
    _L = L
    return L
