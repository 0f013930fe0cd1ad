"""ctypes binding of libqpn_hip.so (include/qpn_hip.h).  No CPU fallback exists: if the HIP
library is missing or does not export the ABI, importing/using the engine raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QPN_HIP_LIB selects another build of the SAME library (A/B of kernel variants; the Julia shim honours it too)
LIB_PATH = os.environ.get("QPN_HIP_LIB") or os.path.join(_HERE, "libqpn_hip.so")

# every symbol include/qpn_hip.h declares
ABI_SYMBOLS = (
    "qpn_abi_version", "qpn_ctx_create", "qpn_ctx_destroy", "qpn_ctx_set_stream",
    "qpn_ctx_use_own_stream",
    "qpn_ctx_synchronize", "qpn_ctx_last_error", "qpn_strerror", "qpn_avi_default_opts",
    "qpn_solve_avi_batch", "qpn_solve_mcp_csc", "qpn_check_avi_batch", "qpn_comp_indices",
    "qpn_assemble_nodes", "qpn_solve_nodes", "qpn_solve_nodes_into", "qpn_order_nodes_by_pivots",
    "qpn_set_node_order", "qpn_verify_nodes",
    "qpn_shared_alloc", "qpn_shared_open", "qpn_shared_close", "qpn_shared_free", "qpn_set_primal_mirrors",
    "qpn_sweep_status", "qpn_ctx_set_auto_schedule", "qpn_ctx_set_option",
    "qpn_nodes_upload", "qpn_nodes_update", "qpn_nodes_set_schedule", "qpn_nodes_free", "qpn_nodes_info", "qpn_solve_nodes_h",
    "qpn_verify_nodes_h", "qpn_pool_size", "qpn_assemble_pools", "qpn_local_pieces", "qpn_recipes_from_masks",
    "qpn_recipes_batch", "qpn_reduced_pieces",
)

MEM_HOST, MEM_DEVICE = 0, 1
AVI_FLAG_COLD_START = 1
MAX_MIRRORS = 7
IPC_HANDLE_BYTES = 64
SHARED_FINE_GRAINED = 1
SWEEP_BOX_BYTES = 512
OPT_MID_ROUTE = 1
OPT_BIG_ROUTE = 2
OPT_SYM_ROUTE = 3


class LibraryMissing(RuntimeError):
    pass


class PoolShape(C.Structure):
    _fields_ = [("players", C.c_int32), ("nd", C.c_int32), ("p", C.c_int32), ("n_i", C.c_void_p), ("m_i", C.c_void_p),
                ("dpos", C.c_void_p)]


POOL_REDUCED, POOL_REFERENCE = 0, 1


class AviOpts(C.Structure):
    _fields_ = [("check_tol", C.c_double), ("piv_tol", C.c_double), ("feas_tol", C.c_double),
                ("comp_tol", C.c_double), ("max_pivots", C.c_int32), ("flags", C.c_int32)]


_lib = None


def load_library():
    """Load libqpn_hip.so and declare the prototypes.  Raises LibraryMissing -- never falls
    back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for s in ABI_SYMBOLS:
        if not hasattr(lib, s):
            raise LibraryMissing(f"{LIB_PATH} does not export {s}")
    dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    lib.qpn_abi_version.restype = C.c_int
    lib.qpn_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.qpn_ctx_destroy.argtypes = [vp]
    lib.qpn_ctx_set_stream.argtypes = [vp, vp]
    lib.qpn_ctx_use_own_stream.argtypes = [vp]
    lib.qpn_ctx_synchronize.argtypes = [vp]
    lib.qpn_ctx_last_error.argtypes = [vp]
    lib.qpn_ctx_last_error.restype = C.c_char_p
    lib.qpn_strerror.argtypes = [C.c_int]
    lib.qpn_strerror.restype = C.c_char_p
    lib.qpn_avi_default_opts.argtypes = [C.POINTER(AviOpts)]
    lib.qpn_avi_default_opts.restype = None
    # pointers are passed as raw addresses (c_void_p) so host numpy and device torch buffers
    # go through the same call
    lib.qpn_solve_avi_batch.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int64, vp, vp, vp, vp,
                                        C.c_int64, vp, vp, vp, vp, vp, C.POINTER(AviOpts), C.c_int]
    lib.qpn_solve_mcp_csc.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                      C.POINTER(AviOpts)]
    lib.qpn_check_avi_batch.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int64, vp, vp, vp, vp,
                                        C.c_int64, vp, C.c_double, vp, vp, C.c_int]
    lib.qpn_comp_indices.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_double, C.c_int32, vp, C.c_int]
    lib.qpn_assemble_nodes.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp,
                                       vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int]
    lib.qpn_solve_nodes.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp,
                                    C.c_int64, vp, vp, vp, vp, vp, C.POINTER(AviOpts), C.c_int]
    lib.qpn_solve_nodes_into.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp,
                                         vp, C.c_int64, vp, vp, vp, vp, vp, C.POINTER(AviOpts), C.c_int, vp, C.c_int64]
    lib.qpn_order_nodes_by_pivots.argtypes = [vp, vp, C.c_int32, C.c_int]
    lib.qpn_set_node_order.argtypes = [vp, vp, C.c_int32, C.c_int]
    lib.qpn_ctx_set_auto_schedule.argtypes = [vp, C.c_int32]
    lib.qpn_ctx_set_option.argtypes = [vp, C.c_int32, C.c_int32]
    lib.qpn_verify_nodes.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp,
                                     vp, vp, vp, vp, vp, C.c_int64, C.c_double, vp, vp, vp, C.c_int]
    lib.qpn_shared_alloc.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(vp), vp]
    lib.qpn_shared_open.argtypes = [vp, vp, C.POINTER(vp)]
    lib.qpn_shared_close.argtypes = [vp, vp]
    lib.qpn_shared_free.argtypes = [vp, vp]
    lib.qpn_set_primal_mirrors.argtypes = [vp, vp, C.c_size_t, C.c_int32, C.POINTER(vp)]
    lib.qpn_sweep_status.argtypes = [vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.POINTER(vp), C.c_uint64,
                                     C.c_int32]
    lib.qpn_nodes_upload.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int,
                                     C.POINTER(vp)]
    lib.qpn_nodes_update.argtypes = [vp, vp, C.c_int32, vp, C.c_int]
    lib.qpn_nodes_set_schedule.argtypes = [vp, vp, C.c_int32]
    lib.qpn_nodes_free.argtypes = [vp, vp]
    lib.qpn_nodes_info.argtypes = [vp, vp, vp]
    lib.qpn_solve_nodes_h.argtypes = [vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.POINTER(AviOpts), C.c_int, vp, C.c_int64]
    lib.qpn_verify_nodes_h.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_double, vp, vp, vp, C.c_int]
    lib.qpn_pool_size.argtypes = [C.POINTER(PoolShape), C.c_int, C.POINTER(C.c_int32)]
    lib.qpn_assemble_pools.argtypes = [vp, C.POINTER(PoolShape), C.c_int, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64,
                                       vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, vp, vp,
                                       vp, C.c_int]
    lib.qpn_local_pieces.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                     vp, vp, vp, vp, C.c_int]
    lib.qpn_recipes_from_masks.argtypes = [vp, C.c_int32, vp, C.c_int64, C.c_int32, vp, C.POINTER(C.c_int64), C.c_int]
    lib.qpn_recipes_batch.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int]
    lib.qpn_reduced_pieces.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                       C.c_double, vp, vp, vp, vp, vp, C.c_int]
    del dp, ip, bp
    _lib = lib
    return lib
