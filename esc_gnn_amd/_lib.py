"""
ctypes binding of libescat.so (the C ABI declared in include/escat.h).

The library is built in-tree by `make -C esc_gnn_amd/csrc` (or
__graft_entry__.build()).  There is NO fallback: if the shared object is
missing or a symbol is absent, importing the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ESCAT_LIB", os.path.join(_HERE, "libescat.so"))      # (ESCAT_LIB: tuning builds, tools/ only)

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
vp = ctypes.c_void_p
i64 = ctypes.c_int64
ci = ctypes.c_int
sz = ctypes.c_size_t


class EscatEvent(ctypes.Structure):
    _fields_ = [("step", ctypes.c_int32), ("slot", ctypes.c_int32),
                ("sign", ctypes.c_int32), ("chan", ctypes.c_int32)]


class EscatSell(ctypes.Structure):
    _fields_ = [("val", ctypes.c_void_p), ("lc", ctypes.c_void_p), ("lane_meta", ctypes.c_void_p),
                ("dmax", ctypes.c_int32), ("n_slices", ctypes.c_int32), ("slices_per_part", ctypes.c_int32),
                ("val_planes", ctypes.c_int32),
                ("col32", ctypes.c_void_p), ("perm", ctypes.c_void_p), ("slice_doff", ctypes.c_void_p),
                ("part_list", ctypes.c_void_p), ("part_row0", ctypes.c_void_p), ("n_list", ctypes.c_int32),
                ("orders", ctypes.c_int32), ("frames", ctypes.c_void_p),
                ("layout", ctypes.c_int32), ("pos_of_row", ctypes.c_void_p),
                ("half_val", ctypes.c_void_p), ("half_aux", ctypes.c_void_p), ("ent_of_slot", ctypes.c_void_p)]


class EscatSellDims(ctypes.Structure):
    _fields_ = [("dmax", ctypes.c_int32), ("n_slices", ctypes.c_int32), ("slices_per_part", ctypes.c_int32),
                ("layout", ctypes.c_int32), ("total_d", ctypes.c_int64), ("total_p", ctypes.c_int64),
                ("val_floats", ctypes.c_int64), ("lc_words", ctypes.c_int64), ("row_len_hist", ctypes.c_int32 * 34)]


# name -> (restype, argtypes); every symbol include/escat.h declares
SIGNATURES = {
    "escat_last_error": (ctypes.c_char_p, []),
    "escat_abi_version": (ci, []),
    "escat_csr_spmm_f32": (ci, [vp, vp, vp, i64, i64, vp, ci, vp, ci, ci, vp]),
    "escat_bsr3_spmv_f32": (ci, [vp, vp, vp, i64, i64, vp, vp, vp]),
    "escat_chain_workspace_bytes": (sz, [i64, ci]),
    "escat_chain_csr_f32": (ci, [vp, vp, vp, i64, i64, vp, ci, ci, vp, ci, ci, vp, vp, ci, ci, vp, vp, sz, vp]),
    "escat_chain_csr_static_f32": (ci, [vp, vp, vp, i64, i64, vp, ci, ci, vp, ci, ci, vp, vp, ci, ci, vp, vp]),
    "escat_chain_csr_bwd_f32": (ci, [vp, vp, vp, i64, i64, vp, ci, ci, vp, ci, ci, vp, vp, ci, ci, vp, vp, sz, vp]),
    "escat_chain_bsr3_f32": (ci, [vp, vp, vp, i64, i64, vp, ci, vp, ci, ci, vp, vp, ci, ci, vp, vp, sz, vp]),
    "escat_chain_bsr3_bwd_f32": (ci, [vp, vp, vp, i64, i64, vp, ci, vp, ci, ci, vp, vp, ci, ci, vp, vp, sz, vp]),
    "escat_chain_both_f32": (ci, [vp, vp, vp, i64, i64, vp, vp, ci, vp, ci, ci, vp, vp, vp, ci, ci, vp, vp]),
    "escat_chain_both_bwd_f32": (ci, [vp, vp, vp, i64, i64, vp, vp, ci, vp, ci, ci, vp, vp, vp, ci, ci, vp, vp]),
    "escat_scatter_orders_f32": (ci, [vp, vp, i64, ci, ci, vp, vp]),
    "escat_scatter_orders_bwd_f32": (ci, [vp, i64, vp, i64, ci, ci, vp, vp, vp]),
    "escat_sell_plan_slots": (ci, [vp, vp, vp, vp, vp, ci, ci, vp, vp, vp]),
    "escat_sell_pack_workspace_bytes": (sz, [i64, ci]),
    "escat_sell_pack_plan": (ci, [vp, i64, vp, ci, ci, ci, ci, vp, sz, vp, vp]),
    "escat_sell_pack_fill": (ci, [vp, vp, vp, vp, i64, i64, vp, ci, ci, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp]),
    "escat_sell_pack_fill_unplanned": (ci, [vp, vp, vp, vp, i64, i64, vp, ci, ci, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp]),
    "escat_sell_pack_refill": (ci, [vp, vp, i64, vp, ci, ci, vp, vp, sz, vp, vp, vp]),
    "escat_sell_pack_refill_signed": (ci, [vp, vp, vp, i64, vp, ci, vp, vp, sz, vp, vp, vp, vp, vp]),
    "escat_sell_pack_refill_half": (ci, [vp, vp, i64, vp, ci, vp, vp, sz, vp, vp, vp, vp, vp, vp]),
    "escat_ingest_coo": (ci, [vp, vp, i64, i64, ci, vp, vp, vp, vp]),
    "escat_ingest_coo_cover": (ci, [vp, vp, i64, i64, vp, vp, vp, vp, vp]),
    "escat_row_blocks_workspace_bytes": (sz, [i64]),
    "escat_row_blocks": (i64, [vp, i64, i64, vp, sz, vp, i64, vp, vp, vp, vp]),
    "escat_ingest_coo_like": (ci, [vp, vp, i64, i64, vp, vp, i64, vp, vp]),
    "escat_row_cover": (ci, [vp, vp, i64, vp, vp]),
    "escat_merge_row_ranges": (i64, [vp, i64, i64, vp, vp]),
    "escat_transpose_sym": (ci, [vp, vp, vp, i64, i64, ci, vp, vp, vp]),
    "escat_neighbor_pool_f32": (ci, [vp, vp, i64, ci, ci, ci, ci, vp, vp, vp, vp]),
    "escat_neighbor_pool_bwd_f32": (ci, [vp, vp, i64, ci, ci, ci, vp, vp, vp, vp]),
    "escat_vector_invariants_f32": (ci, [vp, vp, vp, i64, ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    "escat_vector_invariants_bwd_f32": (ci, [vp, vp, vp, i64, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
    "escat_rotate_frames_f32": (ci, [vp, vp, vp, i64, ci, ci, vp]),
    "escat_infogain_divergence_f32": (ci, [vp, vp, i64, ci, ci, ci, ci, ci, ctypes.c_float, ci, vp, vp, vp]),
    "escat_recover_factored_lds_bytes": (sz, [ci, ci]),
    "escat_recover_factored": (ci, [vp, vp, vp, i64, i64, vp, ci, ci, ci, vp, vp, vp, vp]),
    "escat_lookup3": (ctypes.c_uint32, [ctypes.c_char_p, sz, ctypes.c_uint32]),
    "escat_build_workspace_bytes": (sz, [i64, ci, ci]),
    "escat_build_knn_P": (ci, [vp, vp, ci, i64, ci, ci, vp, vp, vp, i64, vp, vp, sz, vp]),
    "escat_build_grid_workspace_bytes": (sz, [i64, ci]),
    "escat_build_knn_P_grid": (ci, [vp, i64, ci, vp, vp, vp, i64, vp, vp, sz, vp]),
    "escat_build_frames": (ci, [vp, vp, vp, i64, ci, ctypes.c_float, ci, ctypes.c_float, ci, ci, ci, vp, vp, vp]),
    "escat_build_frames_ex": (ci, [vp, vp, vp, i64, ci, ctypes.c_float, ci, ctypes.c_float, ci, ci, ci, ci, ctypes.c_float, vp, vp, vp]),
    "escat_build_Q_bsr3": (ci, [vp, vp, vp, vp, i64, ci, ci, vp, vp]),
}

_lib = None


class EscatError(RuntimeError):
    """a library call failed; `code` is its return value (include/escat.h: ESCAT_EINVAL -1, ESCAT_ERANGE -2, ...), None when
    the failure is on the Python side"""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


ESCAT_ERANGE = -2


def load():
    """Load libescat.so and bind every declared entry point. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EscatError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C esc_gnn_amd/csrc` "
            "(hipcc --offload-arch=gfx950) or __graft_entry__.build(). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().escat_last_error()
        raise EscatError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}", code=int(rc))
