"""ctypes loader for libssw_hip.so (the C ABI declared in include/ssw.h).

There is deliberately no fallback: if the shared library has not been built, or
no GPU is present when a context is created, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSW_LIB_PATH: a diagnostic build of the same library (e.g. -DSSW_TILE_TRACE), never a different implementation
LIB_PATH = os.environ.get("SSW_LIB_PATH") or os.path.join(_HERE, "lib", "libssw_hip.so")

SSW_OK = 0
STATUS_NAMES = {
    0: "SSW_OK", 1: "SSW_ERR_BAD_ARG", 2: "SSW_ERR_BAD_DIMS", 3: "SSW_ERR_LENGTH_MISMATCH",
    4: "SSW_ERR_K_TOO_LARGE", 5: "SSW_ERR_NOT_BASE", 6: "SSW_ERR_UNSUPPORTED", 7: "SSW_ERR_CONSUMED",
    8: "SSW_ERR_HIP", 9: "SSW_ERR_NO_DEVICE", 10: "SSW_ERR_OUT_OF_MEMORY",
}
(SSW_ERR_BAD_ARG, SSW_ERR_BAD_DIMS, SSW_ERR_LENGTH_MISMATCH, SSW_ERR_K_TOO_LARGE, SSW_ERR_NOT_BASE,
 SSW_ERR_UNSUPPORTED, SSW_ERR_CONSUMED, SSW_ERR_HIP, SSW_ERR_NO_DEVICE, SSW_ERR_OUT_OF_MEMORY) = range(1, 11)

ORDER_ENERGY, ORDER_ENERGY_ORTHOGONAL, ORDER_LEGACY, ORDER_CUSTOM = 0, 1, 2, 3
OPTION1, OPTION2, OPTION3, METHOD_CUSTOM = 1, 2, 3, 4
DCT2, DCT2_ORTHOGONAL, DCT3 = 0, 1, 2
PRECISION_F32, PRECISION_F64 = 0, 1
STAGES = ["rgb_to_yiq", "dct_row", "dct_col", "select", "embed", "extract", "similarity", "yiq_to_rgb",
          "resize", "convert", "dct_prep", "dct_row_main", "dct_col_main"]
DCT_FOLDING_DEFAULT = 5
PLAN_FLAGS = {"pair_f64": 1, "rows_deep": 2, "cols_deep": 4, "rows_level2": 8, "cols_level2": 16, "class_major": 32, "fused_cols": 64}
TRANSFER_STATS = ["h2d_bytes", "d2h_bytes", "h2d_seconds", "d2h_seconds", "staged_bytes", "direct_bytes"]


class Config(C.Structure):
    _fields_ = [("ordering", C.c_int32), ("method", C.c_int32), ("alpha", C.c_float), ("precision", C.c_int32)]


_vp, _f32p, _u32p, _u64p, _sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t
_cfgp = C.POINTER(Config)

# name -> (restype, argtypes); this table is also what tests check against include/ssw.h
SIGNATURES = {
    "ssw_version": (C.c_char_p, []),
    "ssw_build_all_strategies": (C.c_int, []),
    "ssw_ctx_transform_plan": (C.c_int, [_vp, _sz, _sz, _sz, C.c_int, C.POINTER(C.c_uint32)]),
    "ssw_status_string": (C.c_char_p, [C.c_int]),
    "ssw_last_error": (C.c_char_p, []),
    "ssw_config_default": (None, [_cfgp]),
    "ssw_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "ssw_ctx_destroy": (C.c_int, [_vp]),
    "ssw_ctx_synchronize": (C.c_int, [_vp]),
    "ssw_ctx_stream": (_vp, [_vp]),
    "ssw_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "ssw_ctx_wait_event": (C.c_int, [_vp, _vp]),
    "ssw_ctx_record_event": (C.c_int, [_vp, _vp]),
    "ssw_ctx_set_chunk_frames": (C.c_int, [_vp, _sz]),
    "ssw_ctx_pass_frames": (_sz, [_vp, _sz, _sz, _sz]),
    "ssw_ctx_set_dct_folding": (C.c_int, [_vp, C.c_int]),
    "ssw_ctx_enable_timing": (C.c_int, [_vp, C.c_int]),
    "ssw_ctx_reset_timing": (C.c_int, [_vp]),
    "ssw_ctx_get_timing": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "ssw_ctx_get_work": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "ssw_ctx_get_traffic": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "ssw_ctx_set_overlap": (C.c_int, [_vp, C.c_int]),
    "ssw_ctx_set_prune": (C.c_int, [_vp, C.c_int]),
    "ssw_ctx_set_odd_split": (C.c_int, [_vp, C.c_int]),
    "ssw_tuning_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "ssw_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "ssw_tuning_reset": (C.c_int, [C.c_char_p]),
    "ssw_ctx_get_prune_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "ssw_ctx_get_select_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "ssw_dev_mem_info": (C.c_int, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "ssw_dev_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "ssw_dev_free": (C.c_int, [_vp, _vp]),
    "ssw_copy_to_dev": (C.c_int, [_vp, _vp, _vp, _sz]),
    "ssw_copy_to_host": (C.c_int, [_vp, _vp, _vp, _sz]),
    "ssw_host_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "ssw_host_free": (C.c_int, [_vp, _vp]),
    "ssw_ctx_set_copy_threads": (C.c_int, [_vp, C.c_int]),
    "ssw_ctx_get_transfer_stats": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int]),
    "ssw_rgb_to_yiq": (C.c_int, [_vp, _f32p, _sz, _sz, _sz, _f32p, _f32p, _f32p]),
    "ssw_yiq_to_rgb": (C.c_int, [_vp, _f32p, _f32p, _f32p, _sz, _sz, _sz, _f32p]),
    "ssw_dct2d": (C.c_int, [_vp, C.c_int, C.c_int, _sz, _sz, _sz, _f32p]),
    "ssw_topk_indices": (C.c_int, [_vp, _f32p, _sz, _sz, _sz, C.c_int, _sz, _u32p]),
    "ssw_embed_coefficients": (C.c_int, [_vp, _f32p, _sz, _sz, _u32p, _sz, C.c_int, C.c_float, _f32p, _sz]),
    "ssw_extract_coefficients": (C.c_int, [_vp, _f32p, _f32p, _sz, _sz, _u32p, _sz, C.c_int, C.c_float, _f32p]),
    "ssw_similarity_batch": (C.c_int, [_vp, _f32p, _f32p, _sz, _sz, _f32p]),
    "ssw_similarity_matrix": (C.c_int, [_vp, _f32p, _sz, _f32p, _sz, _sz, _f32p]),
    "ssw_batch_embed": (C.c_int, [_vp, _cfgp, _f32p, _sz, _sz, _sz, _f32p, _sz, _f32p, _f32p, _u32p]),
    "ssw_batch_extract": (C.c_int, [_vp, _cfgp, _f32p, _f32p, _sz, _sz, _sz, _sz, _f32p, _f32p, _f32p]),
    "ssw_convert_rgb8_to_f32": (C.c_int, [_vp, _vp, _sz, _f32p]),
    "ssw_convert_f32_to_rgb8": (C.c_int, [_vp, _f32p, _sz, _vp]),
    "ssw_resize_rgb8": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "ssw_batch_embed_rgb8": (C.c_int, [_vp, _cfgp, _vp, _sz, _sz, _sz, _f32p, _sz, _vp]),
    "ssw_batch_embed_host_rgb8": (C.c_int, [_vp, _cfgp, C.POINTER(_vp), _sz, _sz, _sz, _f32p, _sz, C.POINTER(_vp)]),
    "ssw_batch_extract_host_rgb8": (C.c_int, [_vp, _cfgp, C.POINTER(_vp), C.POINTER(_vp), _sz, _sz, _sz, _sz, _f32p, _f32p, _f32p]),
    "ssw_convert_rgb16_to_f32": (C.c_int, [_vp, _vp, _sz, _f32p]),
    "ssw_convert_f32_to_rgb16": (C.c_int, [_vp, _f32p, _sz, _vp]),
    "ssw_batch_embed_rgb16": (C.c_int, [_vp, _cfgp, _vp, _sz, _sz, _sz, _f32p, _sz, _f32p]),
    "ssw_batch_extract_rgb16": (C.c_int, [_vp, _cfgp, _vp, _vp, _sz, _sz, _sz, _sz, _f32p, _f32p, _f32p]),
    "ssw_batch_extract_rgb8": (C.c_int, [_vp, _cfgp, _vp, _vp, _sz, _sz, _sz, _sz, _f32p, _f32p, _f32p]),
    "ssw_writer_create": (C.c_int, [_vp, _vp, _sz, _sz, _cfgp, C.POINTER(_vp)]),
    "ssw_writer_create_rgb8": (C.c_int, [_vp, _vp, _sz, _sz, _cfgp, C.POINTER(_vp)]),
    "ssw_writer_create_rgb16": (C.c_int, [_vp, _vp, _sz, _sz, _cfgp, C.POINTER(_vp)]),
    "ssw_writer_coefficients": (C.c_int, [_vp, _vp]),
    "ssw_writer_embed": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), _sz]),
    "ssw_writer_result": (C.c_int, [_vp, _vp]),
    "ssw_writer_mark": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), _sz, _vp]),
    "ssw_writer_result_rgb8": (C.c_int, [_vp, _vp]),
    "ssw_writer_mark_rgb8": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), _sz, _vp]),
    "ssw_writer_destroy": (C.c_int, [_vp]),
    "ssw_reader_create": (C.c_int, [_vp, _vp, _sz, _sz, C.c_int, _cfgp, C.POINTER(_vp)]),
    "ssw_reader_create_rgb8": (C.c_int, [_vp, _vp, _sz, _sz, C.c_int, _cfgp, C.POINTER(_vp)]),
    "ssw_reader_create_rgb16": (C.c_int, [_vp, _vp, _sz, _sz, C.c_int, _cfgp, C.POINTER(_vp)]),
    "ssw_reader_coefficients": (C.c_int, [_vp, _vp]),
    "ssw_reader_indices": (C.c_int, [_vp, _sz, _vp]),
    "ssw_reader_extract": (C.c_int, [_vp, _vp, _vp, _sz]),
    "ssw_reader_destroy": (C.c_int, [_vp]),
    "ssw_similarity": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(C.c_float)]),
    "ssw_synth_frames": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _sz, _sz, _sz, _f32p]),
}

_lib = None


class SswLibraryMissing(ImportError):
    pass


def load() -> C.CDLL:
    """dlopen the HIP library; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SswLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C spread_spectrum_watermarking_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def all_strategies() -> bool:
    """True when the loaded library is the diagnostic build with every transform strategy (make ALL_STRATEGIES=1)."""
    return bool(load().ssw_build_all_strategies())


class SswError(RuntimeError):
    """Raised where the Rust reference would panic (or on a HIP failure)."""

    def __init__(self, status: int, where: str = ""):
        self.status = status
        lib = load()
        msg = lib.ssw_status_string(status).decode()
        detail = lib.ssw_last_error().decode() if status in (SSW_ERR_HIP, SSW_ERR_NO_DEVICE, SSW_ERR_OUT_OF_MEMORY) else ""
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)}: {msg}" + (f" [{detail}]" if detail else ""))


def check(status: int, where: str = "ssw"):
    if status != SSW_OK:
        raise SswError(status, where)
