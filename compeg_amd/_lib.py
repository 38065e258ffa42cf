"""Loader of the in-tree native library (compeg_amd/libcompeg_hip.so).

There is no Python or CPU fallback: if the library is missing the import fails
loudly, and opening a Gpu without a gfx950 device raises Error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COMPEG_LIB: developer override for A/B runs of two builds of the same library (tools/ab_bench.sh)
LIB_PATH = os.environ.get("COMPEG_LIB") or os.path.join(_HERE, "libcompeg_hip.so")

OK, E_INVALID_ARG, E_UNSUPPORTED, E_MALFORMED, E_COUNT_MISMATCH, E_HIP = 0, -1, -2, -3, -4, -5
METADATA_BYTES = 1112
L1_BYTES = 2048


class Error(Exception):
    """String-only error like compeg::Error (src/error.rs:5-46); .code carries the C status."""

    def __init__(self, message, code=E_INVALID_ARG):
        super().__init__(message)
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C compeg_amd/csrc` (hipcc --offload-arch=gfx950). compeg_amd has no fallback path.")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, i = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    pvp, psz, pu32, pi = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_int)
    sig = {
        "compeg_last_error": (C.c_char_p, []),
        "compeg_version": (C.c_char_p, []),
        "compeg_gpu_open": (i, [i, pvp]),
        "compeg_gpu_from_stream": (i, [i, vp, pvp]),
        "compeg_gpu_retain": (None, [vp]),
        "compeg_gpu_release": (None, [vp]),
        "compeg_gpu_device": (i, [vp]),
        "compeg_gpu_name": (C.c_char_p, [vp]),
        "compeg_image_parse": (i, [vp, sz, i, pvp]),
        "compeg_image_parse_ext": (i, [vp, sz, i, C.c_uint, pvp]),
        "compeg_image_free": (None, [vp]),
        "compeg_image_width": (u32, [vp]),
        "compeg_image_height": (u32, [vp]),
        "compeg_image_parallelism": (u32, [vp]),
        "compeg_image_metadata": (vp, [vp]),
        "compeg_image_huffman_l1": (vp, [vp]),
        "compeg_image_huffman_l2": (vp, [vp, psz]),
        "compeg_image_scan_range": (None, [vp, psz, psz]),
        "compeg_scanbuffer_new": (vp, []),
        "compeg_scanbuffer_free": (None, [vp]),
        "compeg_scanbuffer_process": (i, [vp, vp, sz, u32]),
        "compeg_scanbuffer_process_on_gpu": (i, [vp, vp, vp, sz, u32]),
        "compeg_scanbuffer_set_threads": (i, [vp, C.c_uint]),
        "compeg_scanbuffer_data": (vp, [vp, psz]),
        "compeg_scanbuffer_start_positions": (vp, [vp, psz]),
        "compeg_decoder_new": (i, [vp, pvp]),
        "compeg_decoder_free": (None, [vp]),
        "compeg_decoder_enqueue": (i, [vp, vp, vp, pi]),
        "compeg_decoder_start_decode": (i, [vp, vp, pvp]),
        "compeg_decoder_decode_blocking": (i, [vp, vp, pvp]),
        "compeg_decoder_last_warning": (C.c_char_p, [vp]),
        "compeg_decoder_last_stage_times": (C.c_int, [vp, vp]),
        "compeg_decoder_set_device_preprocess": (i, [vp, i]),
        "compeg_decoder_set_scan_threads": (i, [vp, C.c_uint]),
        "compeg_op_wait": (i, [vp]),
        "compeg_op_texture_changed": (i, [vp]),
        "compeg_op_free": (None, [vp]),
        "compeg_decoder_output": (i, [vp, pvp, pu32, pu32, psz]),
        "compeg_decoder_take_output": (i, [vp, pvp, pu32, pu32, psz]),
        "compeg_device_free": (None, [vp]),
        "compeg_decoder_read_output": (i, [vp, vp, u32, u32]),
        "compeg_decoder_read_coefficients": (i, [vp, vp, sz]),
        "compeg_batch_new": (i, [vp, pvp]),
        "compeg_batch_free": (None, [vp]),
        "compeg_batch_upload": (i, [vp, pvp, sz, i]),
        "compeg_batch_upload_jpegs": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_int, C.c_uint]),
        "compeg_batch_upload_jpegs_begin": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_int, C.c_uint]),
        "compeg_batch_upload_end": (i, [vp]),
        "compeg_batch_decode": (i, [vp, vp]),
        "compeg_batch_set_chunk": (i, [vp, u32]),
        "compeg_batch_set_device_preprocess": (i, [vp, i]),
        "compeg_batch_host_fallbacks": (sz, [vp]),
        "compeg_batch_wait": (i, [vp]),
        "compeg_batch_count": (sz, [vp]),
        "compeg_batch_output": (i, [vp, sz, pvp, pu32, pu32, psz]),
        "compeg_batch_read_output": (i, [vp, sz, vp]),
        "compeg_batch_algorithmic_bytes": (C.c_uint64, [vp]),
        "compeg_batch_pixels": (C.c_uint64, [vp]),
        "compeg_batch_timing": (i, [vp, i, pu32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "compeg_batch_set_timing": (i, [vp, i]),
        "compeg_batch_last_kernel": (i, [vp]),
        "compeg_host_feed_work": (i, [vp, vp, C.c_size_t, i, C.c_uint, i, i, C.POINTER(C.c_double)]),
        "compeg_host_alloc": (i, [C.c_size_t, C.POINTER(C.c_void_p)]),
        "compeg_host_free": (None, [vp]),
        "compeg_host_register": (i, [vp, C.c_size_t]),
        "compeg_host_unregister": (i, [vp]),
        "compeg_decoder_last_kernel": (i, [vp]),
    }
    for name, (res, args) in sig.items():
        if os.environ.get("COMPEG_LIB") and not hasattr(L, name):
            continue           # (A/B runs against an older build: it may lack the newest entry points)
        fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    L._signatures = sig
    return L


lib = _load()


def check(rc):
    if rc != OK:
        raise Error(lib.compeg_last_error().decode("utf-8", "replace"), rc)
