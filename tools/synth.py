"""Deterministic synthetic baseline-JPEG inputs for tests and bench (ctypes over
tools/libsynthjpeg.so).  Input generation only -- not part of the decoder."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsynthjpeg.so")
NO_DHT, NO_EOI, JFIF = 1, 2, 4
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "synth_jpeg.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.synth_encode.restype = C.c_size_t
        L.synth_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_uint, C.c_void_p, C.c_size_t]
        L.synth_fill.restype = None
        L.synth_fill.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
        _lib = L
    return _lib


def fill(w, h, seed=0, kind=0, noise=12):
    rgb = np.empty((h, w, 3), dtype=np.uint8)
    lib().synth_fill(rgb.ctypes.data, w, h, seed, kind, noise)
    return rgb


def encode(rgb, quality=85, sampling=(2, 1), ri=4, flags=0):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    cap = w * h * 3 + 4096
    while True:
        out = np.empty(cap, dtype=np.uint8)
        n = lib().synth_encode(rgb.ctypes.data, w, h, quality, sampling[0], sampling[1], ri, flags,
                               out.ctypes.data, cap)
        if n <= cap:
            return out[:n].tobytes()
        cap = n


def make_jpeg(w, h, seed=0, kind=0, noise=12, quality=85, sampling=(2, 1), ri=4, flags=0):
    """One synthetic 4:2:2 restart-interval JPEG."""
    return encode(fill(w, h, seed, kind, noise), quality, sampling, ri, flags)
