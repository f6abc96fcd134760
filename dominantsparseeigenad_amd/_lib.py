"""ctypes binding of libdsea.so (C ABI declared in include/dsea.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C dominantsparseeigenad_amd/csrc``.
There is no fallback: if the shared object is missing, ``load()`` raises and every GPU entry point of
the package fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSEA_LIB", os.path.join(_HERE, "csrc", "libdsea.so"))

_lib = None

CG_RR, CG_DAD, CG_RRNEW, CG_ALPHA, CG_BETA, CG_RESNORM, CG_DONE, CG_ITERS = range(8)
CG_STATE_LEN = 8
ERR_ARG = -1
ERR_NOT_CONVERGED = -5
ERR_BREAKDOWN = -7
ERR_TIMEOUT = -8
ERR_COMM = -9
ERR_PREMISE = -10
ERR_SECOND_PASS = -11
ERR_UNSUPPORTED = -6
COMM_ID_BYTES = 128
TUNE_TFIM_TILE_LOG2, TUNE_CSR_GROUP, TUNE_SELL_UNROLL, TUNE_SELL_XCD_MAP, TUNE_SELL_NT = 1, 2, 3, 4, 5
TUNE_SELL_MAX_WIDTH = 6
SDDMM_ACCUMULATE, SDDMM_SYMMETRIC = 1, 2
POP_OVERLAP, POP_PAIRWISE, POP_NO_EXCHANGE, POP_CG_REFERENCE, POP_CG_ONE_REDUCTION = 1, 2, 4, 8, 16
# caller-supplied collectives of dsea_comm_create_callbacks (device pointers + the stream the data was produced on)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_void_p)
ALLTOALL_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p)
SENDRECV_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p)


class DseaError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/dsea.h one to one
_SIGNATURES = {
    "dsea_version": (c_int, []),
    "dsea_error_string": (c_char_p, [c_int]),
    "dsea_last_hip_error": (c_int, []),
    "dsea_op_set_tuning": (c_int, [c_void_p, c_int, c_int]),
    "dsea_ws_bytes": (c_int, [c_int64, c_int, POINTER(c_size_t)]),
    "dsea_ws_create": (c_int, [c_void_p, c_size_t, c_int64, c_int, POINTER(c_void_p)]),
    "dsea_ws_destroy": (c_int, [c_void_p]),
    "dsea_ws_set_rows_per_lane": (c_int, [c_void_p, c_int]),
    "dsea_ws_set_split": (c_int, [c_void_p, c_int]),
    "dsea_ws_set_persist": (c_int, [c_void_p, c_int]),
    "dsea_cg_last_form": (c_int, [c_void_p, POINTER(c_int)]),
    "dsea_ws_set_lanczos_persist": (c_int, [c_void_p, c_int]),
    "dsea_ws_set_reorth_passes": (c_int, [c_void_p, c_int]),
    "dsea_ws_set_fault_injection": (c_int, [c_void_p, c_int]),
    "dsea_ws_set_shadow": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_double]),
    "dsea_lanczos_lp_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), c_void_p]),
    "dsea_ws_set_partial_reorth": (c_int, [c_void_p, c_int, c_double]),
    "dsea_lanczos_partial_step": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
    "dsea_lanczos_reorth_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double), c_void_p]),
    "dsea_profile_begin": (c_int, [c_void_p, c_int]),
    "dsea_profile_end": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double)]),
    "dsea_op_create_tfim": (c_int, [c_int, c_int, c_int64, c_void_p, c_double, c_double, POINTER(c_void_p)]),
    "dsea_op_create_csr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_op_create_sell": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_op_create_sell16": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_op_create_sell16p2": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_op_create_sell16v8": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_op_create_stencil3": (c_int, [c_int64, c_double, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_op_create_dense": (c_int, [c_int64, c_void_p, c_int64, c_int, POINTER(c_void_p)]),
    "dsea_op_symdense_work_bytes": (c_size_t, [c_int64]),
    "dsea_op_create_symdense": (c_int, [c_int64, c_void_p, c_int, c_int64, c_void_p, POINTER(c_void_p)]),
    "dsea_op_transfer_work_bytes": (c_size_t, [c_int, c_int]),
    "dsea_op_create_transfer": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_op_update_vals": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "dsea_op_sddmm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p]),
    "dsea_op_set_slab": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "dsea_op_destroy": (c_int, [c_void_p]),
    "dsea_op_dim": (c_int, [c_void_p, POINTER(c_int64)]),
    "dsea_spmv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dsea_dot": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dsea_shift_dot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_axpy": (c_int, [c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_nrm2sq": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dsea_probe_stream": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_scale_store": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_lanczos_rdots": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "dsea_lanczos_axpy_norm": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "dsea_lanczos_store": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                   c_void_p]),
    "dsea_lanczos_callable_alpha": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dsea_lanczos_callable_step": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "dsea_ritz_combine": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "dsea_project_out": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_cg_init": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_cg_init_check": (c_int, [c_void_p, c_void_p, c_double, c_void_p]),
    "dsea_cg_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_cg_check": (c_int, [c_void_p, c_void_p, c_double, c_void_p]),
    "dsea_cg_direction": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_cg_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int64, c_int64,
                             c_void_p]),
    "dsea_lanczos_form_r": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "dsea_hypercube_flipsum": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "dsea_plz_dots": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    "dsea_plz_correct": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dsea_plz_correct_matvec": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "dsea_axpy_multi_dot": (c_int, [c_void_p, c_double, c_void_p, POINTER(c_void_p), c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dsea_plz_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                c_void_p, c_int64, c_void_p]),
    "dsea_comm_unique_id": (c_int, [c_void_p]),
    "dsea_comm_init_rank": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "dsea_comm_adopt": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "dsea_comm_create_callbacks": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_comm_destroy": (c_int, [c_void_p]),
    "dsea_comm_allreduce": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_comm_alltoall": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_comm_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dsea_pop_tfim_scratch_doubles": (c_size_t, [c_int, c_int]),
    "dsea_pop_create_tfim": (c_int, [c_int, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p, c_int, c_double,
                                     POINTER(c_void_p)]),
    "dsea_pop_create_stencil3": (c_int, [c_int64, c_double, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_pop_create_csr": (c_int, [c_void_p, c_void_p, POINTER(c_void_p)]),
    "dsea_pop_sddmm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p]),
    "dsea_pop_destroy": (c_int, [c_void_p]),
    "dsea_pop_set_flags": (c_int, [c_void_p, c_int]),
    "dsea_pop_matvec": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dsea_pop_dot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "dsea_pop_lanczos_run": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p]),
    "dsea_pop_lanczos_status": (c_int, [c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "dsea_pop_cg_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int64, c_int,
                                POINTER(c_int64), POINTER(c_double), c_void_p]),
    "dsea_lanczos_run": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                 c_void_p]),
    "dsea_lanczos_run_basisfree": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "dsea_arnoldi_extend": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int,
                                    c_void_p]),
    "dsea_arnoldi_second_passes": (c_int, [c_void_p, POINTER(c_int64), c_void_p]),
    "dsea_ws_set_arnoldi_optimistic": (c_int, [c_void_p, c_int]),
    "dsea_arnoldi_status": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), c_void_p]),
    "dsea_arnoldi_status_enqueue": (c_int, [c_void_p, c_void_p, c_void_p]),
    "dsea_arnoldi_clear_record": (c_int, [c_void_p, c_void_p]),
    "dsea_arnoldi_orth": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int,
                                  c_void_p]),
    "dsea_gmres_work_doubles": (c_size_t, [c_int]),
    "dsea_gmres_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_double,
                                 c_void_p, c_void_p]),
    "dsea_gmres_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                c_void_p, c_double, c_void_p, c_void_p]),
    "dsea_gmres_end": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dsea_gmres_cycle": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                 c_double, c_void_p, c_int, c_void_p]),
    "dsea_lanczos_status": (c_int, [c_void_p, POINTER(c_int), c_void_p]),
    "dsea_cg_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int64,
                            c_int, POINTER(c_int64), POINTER(c_double), c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load():
    """Return the ctypes handle of libdsea.so, loading it on first use.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DseaError(
            "libdsea.so not found at %s -- the HIP extension is not built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C dominantsparseeigenad_amd/csrc`.  There is no CPU fallback for CUDA tensors." % LIB_PATH
        )
    # PyTorch-ROCm bundles its own HIP runtime (libamdhip64).  It must be the one already in the process when
    # libdsea.so is mapped, otherwise the library binds to a second runtime instance (the system one) that
    # shares no device context / streams with torch (observed: hipErrorNoDevice on the first call).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, what="dsea call", allow=()):
    if status == 0 or status in allow:
        return status
    lib = load()
    msg = lib.dsea_error_string(status).decode()
    extra = ""
    if status == -4:
        extra = " (hipError %d)" % lib.dsea_last_hip_error()
    raise DseaError("%s failed: %s%s" % (what, msg, extra))
