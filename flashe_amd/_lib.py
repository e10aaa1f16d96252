"""ctypes binding of libflashe_hip.so -- the ONLY compute path of this package.

There is deliberately no CPU fallback: if the shared library is missing or no HIP device is
usable, loading / context creation raises.  Signatures follow include/flashe.h.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("FLASHE_LIB_NAME", "libflashe_hip.so"))

OK = 0
SCHEME_SINGLE = 0
SCHEME_DOUBLE = 1

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_u64pp = ctypes.POINTER(c_u64p)
c_vp = ctypes.c_void_p
c_int = ctypes.c_int
c_u32 = ctypes.c_uint32
c_u64 = ctypes.c_uint64
c_size = ctypes.c_size_t


class PrfJob(ctypes.Structure):
    """flashe_prf_job of include/flashe.h."""
    _fields_ = [("add_idx", ctypes.c_uint32), ("minus_idx", ctypes.c_uint32), ("has_minus", ctypes.c_int32),
                ("in_limbs", ctypes.c_int32), ("first", ctypes.c_uint64), ("count", ctypes.c_uint64),
                ("in_dev", ctypes.c_void_p), ("out_dev", ctypes.c_void_p), ("n_in", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("in_stride", ctypes.c_uint64), ("sum_out_dev", ctypes.c_void_p)]



class CodecLayer(ctypes.Structure):
    """flashe_codec_layer of include/flashe.h."""
    _fields_ = [("start", ctypes.c_uint64), ("x_dev", ctypes.c_void_p), ("alpha", ctypes.c_double), ("x_is_f64", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class BatchLayer(ctypes.Structure):
    """flashe_batch_layer of include/flashe.h."""
    _fields_ = [("size", ctypes.c_uint64), ("x_dev", ctypes.c_void_p), ("alpha", ctypes.c_double), ("x_is_f64", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class FlasheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"flashe error {code}: {msg}")
        self.code = code


_SIGNATURES = {
    # name: (restype, argtypes)
    "flashe_abi_version": (c_int, []),
    "flashe_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "flashe_device_peer_access": (c_int, [c_int, c_int, ctypes.POINTER(c_int)]),
    "flashe_limbs": (c_int, [c_int]),
    "flashe_ctx_create": (c_int, [ctypes.POINTER(c_vp), c_u8p, c_int, c_int, c_vp]),
    "flashe_ctx_destroy": (c_int, [c_vp]),
    "flashe_ctx_set_key": (c_int, [c_vp, c_u8p]),
    "flashe_ctx_int_bits": (c_int, [c_vp]),
    "flashe_ctx_compact_layout": (c_int, [c_vp]),
    "flashe_ctx_set_cu_limit": (c_int, [c_vp, c_int]),
    "flashe_ctx_cu_count": (c_int, [c_vp]),
    "flashe_ctx_set_prf_backend": (c_int, [c_vp, c_int]),
    "flashe_last_error": (ctypes.c_char_p, [c_vp]),
    "flashe_selftest": (c_int, [c_vp]),
    "flashe_chunks": (c_int, [c_u64, c_u32, c_u64p]),
    "flashe_telescope": (c_int, [c_u32p, c_int, c_u32p, c_u32p, ctypes.POINTER(c_int)]),
    "flashe_prp_block": (c_int, [c_u8p, c_u8p, c_u8p]),
    "flashe_dev_alloc": (c_int, [c_vp, c_size, ctypes.POINTER(c_vp)]),
    "flashe_dev_free": (c_int, [c_vp, c_vp]),
    "flashe_dev_trim": (c_int, [c_int]),
    "flashe_dev_pool_stats": (c_int, [c_int, c_u64p, c_u64p, c_u64p]),
    "flashe_host_alloc": (c_int, [c_size, ctypes.POINTER(c_vp)]),
    "flashe_host_free": (c_int, [c_vp]),
    "flashe_memcpy_h2d": (c_int, [c_vp, c_vp, c_vp, c_size]),
    "flashe_memcpy_d2h": (c_int, [c_vp, c_vp, c_vp, c_size]),
    "flashe_memcpy_d2d": (c_int, [c_vp, c_vp, c_vp, c_size]),
    "flashe_memset_dev": (c_int, [c_vp, c_vp, c_int, c_size]),
    "flashe_sync": (c_int, [c_vp]),
    "flashe_event_create": (c_int, [c_vp, ctypes.POINTER(c_vp)]),
    "flashe_event_destroy": (c_int, [c_vp, c_vp]),
    "flashe_event_record": (c_int, [c_vp, c_vp]),
    "flashe_stream_wait_event": (c_int, [c_vp, c_vp]),
    "flashe_event_elapsed_ms": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(ctypes.c_float)]),
    "flashe_mask_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u64, c_u32, c_vp]),
    "flashe_mask": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u64, c_u32, c_vp]),
    "flashe_encrypt_dev": (c_int, [c_vp, c_u32, c_u32, c_int, c_u64, c_u32, c_vp, c_int, c_vp]),
    "flashe_encrypt": (c_int, [c_vp, c_u32, c_u32, c_int, c_u64, c_u32, c_vp, c_int, c_vp]),
    "flashe_graph_begin": (c_int, [c_vp]),
    "flashe_graph_end": (c_int, [c_vp, ctypes.POINTER(c_vp)]),
    "flashe_graph_launch": (c_int, [c_vp, c_vp]),
    "flashe_graph_launch_shifted": (c_int, [c_vp, c_vp, c_u32]),
    "flashe_graph_destroy": (c_int, [c_vp]),
    "flashe_prf_jobs_dev": (c_int, [c_vp, c_u32, c_u64, c_u32, c_int, ctypes.POINTER(PrfJob)]),
    "flashe_encrypt_batch_dev": (c_int, [c_vp, c_u32, c_int, c_u64, c_u32, c_int, c_u32p, ctypes.POINTER(c_vp), c_int, ctypes.POINTER(c_vp)]),
    "flashe_encrypt_batch_sum_dev": (c_int, [c_vp, c_u32, c_int, c_u64, c_u32, c_int, c_u32p, ctypes.POINTER(c_vp), c_int, ctypes.POINTER(c_vp),
                                             c_vp]),
    "flashe_span_bounds_create": (c_int, [c_vp, c_u64, c_int, ctypes.POINTER(c_vp), c_u64p, ctypes.POINTER(c_vp)]),
    "flashe_span_bounds_recompute": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_u64p]),
    "flashe_span_bounds_destroy": (None, [c_vp]),
    "flashe_sparse_aggregate_bounds_dev": (c_int, [c_vp, c_u64, c_int, ctypes.POINTER(c_vp), c_u64p, ctypes.POINTER(c_vp), c_u64p, c_vp, c_vp]),
    "flashe_sparse_decrypt_bounds_dev": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64, c_u32, c_vp, c_vp, c_vp]),
    "flashe_sparse_encrypt_aggregate_dev": (c_int, [c_vp, c_u32, c_u32, c_u64, c_int, c_u32p, ctypes.POINTER(c_vp), c_u64p, ctypes.POINTER(c_vp), c_int,
                                                    c_u64p, c_vp, ctypes.POINTER(c_vp), c_vp]),
    "flashe_sparse_span": (c_int, []),
    "flashe_sparse_encrypt_aggregate_range_dev": (c_int, [c_vp, c_u32, c_u32, c_u64, c_int, c_u32p, ctypes.POINTER(c_vp), c_u64p, ctypes.POINTER(c_vp), c_int,
                                                          c_u64p, c_vp, c_u64, c_u64, ctypes.POINTER(c_vp), c_vp]),
    "flashe_sparse_decrypt_range_dev": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64, c_u32, c_vp, c_u64, c_u64, c_vp, c_vp]),
    "flashe_aggregate_elem_u32_dev": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), c_u64, c_vp]),
    "flashe_dynamic_masking_cost_dev": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64p, c_u64p]),
    "flashe_prepare_encrypt": (c_int, [c_vp, c_u32, c_u32, c_int, c_u64, c_u32]),
    "flashe_prepare_decrypt": (c_int, [c_vp, c_u32, c_u32, c_u64, c_u32]),
    "flashe_prepared_query": (c_int, [c_vp, c_int, c_u64p, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "flashe_prepared_discard": (c_int, [c_vp, c_int]),
    "flashe_encrypt_prepared_dev": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp]),
    "flashe_encrypt_prepared": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp]),
    "flashe_decrypt_prepared_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_vp, c_vp]),
    "flashe_decrypt_prepared": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_vp, c_vp]),
    "flashe_encrypt_batch_range_dev": (c_int, [c_vp, c_u32, c_int, c_u64, c_u32, c_u64, c_u64, c_int, c_u32p, ctypes.POINTER(c_vp), c_int,
                                               ctypes.POINTER(c_vp), c_vp]),
    "flashe_decrypt_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_vp, c_vp]),
    "flashe_decrypt": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_vp, c_vp]),
    "flashe_mask_range_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u64, c_u32, c_u64, c_u64, c_vp]),
    "flashe_encrypt_range_dev": (c_int, [c_vp, c_u32, c_u32, c_int, c_u64, c_u32, c_u64, c_u64, c_vp, c_int, c_vp]),
    "flashe_decrypt_range_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_u64, c_u64, c_vp, c_vp]),
    "flashe_combine_dev": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp, c_vp, c_vp]),
    "flashe_combine": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp, c_vp, c_vp]),
    "flashe_combine_batch_dev": (c_int, [c_vp, c_u64, c_int, ctypes.POINTER(c_vp), c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                         ctypes.POINTER(c_vp)]),
    "flashe_combine_batch_sum_dev": (c_int, [c_vp, c_u64, c_int, ctypes.POINTER(c_vp), c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                             ctypes.POINTER(c_vp), c_vp]),
    "flashe_combine_batch_sum_decrypt_dev": (c_int, [c_vp, c_u64, c_int, ctypes.POINTER(c_vp), c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                                     ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_vp]),
    "flashe_aggregate_elem_dev": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), c_u64, c_vp]),
    "flashe_aggregate_elem": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), c_u64, c_vp]),
    "flashe_aggregate_decrypt_range_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_u64, c_u64, c_int,
                                                   ctypes.POINTER(c_vp), c_vp, c_vp]),
    "flashe_encrypt_batch_u32_dev": (c_int, [c_vp, c_u32, c_int, c_u64, c_u32, c_int, c_u32p, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "flashe_encrypt_batch_sum_u32_dev": (c_int, [c_vp, c_u32, c_int, c_u64, c_u32, c_int, c_u32p, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_vp]),
    "flashe_aggregate_decrypt_u32_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_u64, c_u64, c_int,
                                                 ctypes.POINTER(c_vp), c_vp, c_vp, c_int]),
    "flashe_widen_u32_dev": (c_int, [c_vp, c_u64, c_vp, c_vp]),
    "flashe_narrow_u32_dev": (c_int, [c_vp, c_u64, c_vp, c_vp]),
    "flashe_aggregate_packed_dev": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), c_u64, c_u64, c_vp]),
    "flashe_aggregate_packed": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp), c_u64, c_u64, c_vp]),
    "flashe_packed_probe_dev": (c_int, [c_vp, c_u64, c_vp, c_vp]),
    "flashe_packed_add_carry_dev": (c_int, [c_vp, c_u64, c_u64, c_u64, c_vp]),
    "flashe_packed_resolve_carry_dev": (c_int, [c_vp, c_u64, c_u64, c_vp, c_int, c_vp]),
    "flashe_packed_resolve_carry_strided_dev": (c_int, [c_vp, c_u64, c_u64, c_vp, c_int, c_int, c_vp]),
    "flashe_pack_dev": (c_int, [c_vp, c_u64, c_vp, c_vp]),
    "flashe_pack": (c_int, [c_vp, c_u64, c_vp, c_vp]),
    "flashe_unpack_dev": (c_int, [c_vp, c_u64, c_vp, c_vp]),
    "flashe_unpack": (c_int, [c_vp, c_u64, c_vp, c_vp]),
    "flashe_expand_to_dense_dev": (c_int, [c_vp, c_u64, c_u64, c_vp, c_vp, c_u64p, c_vp]),
    "flashe_expand_to_dense": (c_int, [c_vp, c_u64, c_u64, c_vp, c_vp, c_u64p, c_vp]),
    "flashe_sparse_aggregate_dev": (c_int, [c_vp, c_u64, c_int, ctypes.POINTER(c_vp), c_u64p, ctypes.POINTER(c_vp), c_u64p, c_int, c_vp]),
    "flashe_sparse_minus_mask_sorted_dev": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64, c_u32, c_vp]),
    "flashe_sparse_decrypt_dev": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64, c_u32, c_int, c_vp, c_vp]),
    "flashe_sparse_minus_mask_dev": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64, c_u32, c_vp]),
    "flashe_sparse_minus_mask": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64, c_u32, c_vp]),
    "flashe_sparse_double_masks_dev": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64p, c_u64, c_vp, c_vp]),
    "flashe_sparse_dense_mask_dev": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64, c_vp]),
    "flashe_sparse_dense_mask": (c_int, [c_vp, c_u32, c_int, ctypes.POINTER(c_vp), c_u64, c_vp]),
    "flashe_quantize_encrypt_dev": (c_int, [c_vp, c_u32, c_u32, c_int, c_u64, c_u32, c_vp, c_int, ctypes.c_double, c_int, c_vp, c_vp]),
    "flashe_decrypt_unquantize_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_vp, ctypes.c_double, c_int, c_int, c_vp]),
    "flashe_quantize_batch_model_dev": (c_int, [c_vp, ctypes.POINTER(BatchLayer), c_int, c_int, c_int, c_vp, c_u64, c_vp]),
    "flashe_unbatch_unquantize_model_dev": (c_int, [c_vp, ctypes.POINTER(BatchLayer), c_int, c_int, c_int, c_int, c_vp, c_u64, c_vp]),
    "flashe_quantize_encrypt_model_dev": (c_int, [c_vp, c_u32, c_u32, c_int, c_u64, c_u32, c_u64, c_u64, ctypes.POINTER(CodecLayer), c_int, c_int, c_vp, c_vp]),
    "flashe_decrypt_unquantize_model_dev": (c_int, [c_vp, c_u32, c_u32p, c_int, c_u32p, c_int, c_u64, c_u32, c_u64, c_u64, c_vp,
                                                    ctypes.POINTER(CodecLayer), c_int, c_int, c_int, c_vp]),
    "flashe_unquantize_model_dev": (c_int, [c_vp, c_u64, c_u64, c_u64, c_vp, ctypes.POINTER(CodecLayer), c_int, c_int, c_int, c_vp]),
    "flashe_shift_dev": (c_int, [c_vp, c_u64, c_vp, c_int, ctypes.c_double, c_int]),
    "flashe_mean_std_dev": (c_int, [c_vp, c_u64, c_vp, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "flashe_mt19937_random_dev": (c_int, [c_vp, c_u32p, c_u32p, c_u64, c_vp]),
    "flashe_mt19937_jump_selfcheck": (c_int, []),
    "flashe_mt19937_plan": (c_int, [c_u32, c_u64, c_u32p, c_u32p, c_u32p]),
    "flashe_quantize_dev": (c_int, [c_vp, c_u64, c_vp, c_int, ctypes.c_double, c_int, c_vp, c_vp]),
    "flashe_quantize": (c_int, [c_vp, c_u64, c_vp, c_int, ctypes.c_double, c_int, c_vp, c_vp]),
    "flashe_unquantize_dev": (c_int, [c_vp, c_u64, c_vp, c_int, ctypes.c_double, c_int, c_int, c_vp]),
    "flashe_unquantize": (c_int, [c_vp, c_u64, c_vp, c_int, ctypes.c_double, c_int, c_int, c_vp]),
    "flashe_batch_dev": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp]),
    "flashe_batch": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp]),
    "flashe_unbatch_dev": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp]),
    "flashe_unbatch": (c_int, [c_vp, c_u64, c_vp, c_int, c_vp]),
    "flashe_sparsify_dev": (c_int, [c_vp, c_u64, c_u64, c_vp, c_int, c_vp, c_vp, c_vp]),
    "flashe_sparsify": (c_int, [c_vp, c_u64, c_u64, c_vp, c_int, c_vp, c_vp, c_vp]),
    "flashe_sparsify_batch_dev": (c_int, [c_vp, c_int, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), c_vp, c_int, c_vp, c_vp, c_vp]),
    "flashe_sparsify_batch": (c_int, [c_vp, c_int, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), c_vp, c_int, c_vp, c_vp, c_vp]),
    "flashe_rccl_unique_id": (c_int, [c_u8p]),
    "flashe_rccl_init": (c_int, [c_vp, c_u8p, c_int, c_int, ctypes.POINTER(c_vp)]),
    "flashe_rccl_destroy": (c_int, [c_vp]),
    "flashe_rccl_rank": (c_int, [c_vp]),
    "flashe_rccl_world": (c_int, [c_vp]),
    "flashe_rccl_all_to_all": (c_int, [c_vp, c_vp, c_vp, c_size, c_vp, c_size, c_size]),
    "flashe_rccl_all_gather": (c_int, [c_vp, c_vp, c_vp, c_vp, c_size]),
    "flashe_rccl_reduce_scatter_modadd": (c_int, [c_vp, c_vp, c_vp, c_u64, c_vp, c_vp, c_vp]),
    "flashe_rccl_allreduce_modadd_u64": (c_int, [c_vp, c_vp, c_vp, c_u64]),
    "flashe_rccl_allreduce_f64": (c_int, [c_vp, c_vp, ctypes.POINTER(ctypes.c_double), c_int]),
    "flashe_rccl_barrier": (c_int, [c_vp, c_vp]),
    "flashe_rccl_version": (c_int, [ctypes.POINTER(c_int)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load libflashe_hip.so (built by `python -c 'import __graft_entry__ as g; g.build()'`
    or `make -C flashe_amd/csrc`).  Raises if it is missing -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlasheError(-2, f"{LIB_PATH} not found: build it with `make -C flashe_amd/csrc` "
                              "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, rc):
    if rc != OK:
        msg = load().flashe_last_error(ctx_handle)
        raise FlasheError(rc, msg.decode() if msg else "unknown")
    return rc
