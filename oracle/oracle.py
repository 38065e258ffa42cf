"""ctypes binding of the CPU oracle (oracle/libcompeg_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importers: tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped package (compeg_amd) must never
import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcompeg_oracle.so")
ERRLEN = 256


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("compeg_oracle.c", "compeg_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, sz, u32, cp = C.c_void_p, C.c_size_t, C.c_uint32, C.c_char_p
        psz = C.POINTER(C.c_size_t)
        sig = {
            "orc_scanbuf_new": (vp, []),
            "orc_scanbuf_free": (None, [vp]),
            "orc_scanbuf_process": (C.c_int, [vp, vp, sz, u32, cp]),
            "orc_scanbuf_data": (vp, [vp, psz]),
            "orc_scanbuf_starts": (vp, [vp, psz]),
            "orc_table_build": (vp, [vp, vp, sz]),
            "orc_table_free": (None, [vp]),
            "orc_table_lookup": (C.c_uint16, [vp, C.c_uint16]),
            "orc_table_l2_len": (sz, [vp]),
            "orc_table_debug": (sz, [vp, cp, sz]),
            "orc_table_default": (vp, [C.c_int]),
            "orc_bits_init": (None, [vp, vp, sz, u32]),
            "orc_bits_refill": (None, [vp]),
            "orc_bits_consume": (None, [vp, u32]),
            "orc_bits_peek": (u32, [vp, u32]),
            "orc_bits_huffdecode_table": (u32, [vp, vp]),
            "orc_huff_extend": (C.c_int32, [C.c_int32, u32]),
            "orc_parser_dump": (sz, [vp, sz, cp, sz]),
            "orc_image_parse": (vp, [vp, sz, cp]),
            "orc_image_parse_ext": (vp, [vp, sz, C.c_uint, cp]),
            "orc_image_free": (None, [vp]),
            "orc_image_width": (u32, [vp]),
            "orc_image_height": (u32, [vp]),
            "orc_image_parallelism": (u32, [vp]),
            "orc_image_metadata": (vp, [vp]),
            "orc_image_l1": (vp, [vp]),
            "orc_image_l2": (vp, [vp, psz]),
            "orc_image_scan": (None, [vp, psz, psz]),
            "orc_huffman_pass": (None, [vp, vp, vp, sz, vp, sz, vp, sz, vp, sz]),
            "orc_dct_pass": (None, [vp, vp, sz]),
            "orc_finalize_pass": (None, [vp, vp, sz, vp, u32, u32]),
            "orc_image_decode": (C.c_int, [vp, vp, vp, u32, u32, vp, cp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _buf(b):
    """bytes-like -> (ctypes pointer value, length, keepalive)."""
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    a = np.ascontiguousarray(a)
    return a.ctypes.data, a.nbytes, a


class OracleError(Exception):
    pass


class ScanBuffer:
    """ref: src/scan.rs:15-77"""

    def __init__(self):
        self._p = lib().orc_scanbuf_new()

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_scanbuf_free(self._p)
            self._p = None

    def process(self, scan, expected):
        ptr, n, keep = _buf(scan)
        err = C.create_string_buffer(ERRLEN)
        rc = lib().orc_scanbuf_process(self._p, ptr, n, expected, err)
        if rc:
            raise OracleError(err.value.decode())

    def _get(self, fn):
        n = C.c_size_t()
        p = fn(self._p, C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def processed_scan_data(self):
        return self._get(lib().orc_scanbuf_data)

    def start_positions(self):
        return self._get(lib().orc_scanbuf_starts)


class Table:
    """ref: src/huffman.rs:23-232"""

    def __init__(self, li=None, vij=None, default=None):
        if default is not None:
            self._p = lib().orc_table_default(default)
        else:
            li = np.asarray(li, dtype=np.uint8)
            vij = np.asarray(vij, dtype=np.uint8)
            assert li.size == 16
            self._p = lib().orc_table_build(li.ctypes.data, vij.ctypes.data, vij.size)
        if not self._p:
            raise OracleError("panic: malformed huffman table")

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_table_free(self._p)
            self._p = None

    def lookup(self, code):
        e = lib().orc_table_lookup(self._p, code)
        return e >> 8, e & 0xFF  # (bits, value)

    def l2_len(self):
        return lib().orc_table_l2_len(self._p)

    def debug(self):
        out = C.create_string_buffer(1 << 16)
        lib().orc_table_debug(self._p, out, len(out))
        return out.value.decode()


class _BitsStruct(C.Structure):
    _fields_ = [("words", C.c_void_p), ("nwords", C.c_size_t), ("next_word", C.c_uint32),
                ("cur", C.c_uint32), ("next", C.c_uint32), ("left", C.c_uint32)]


class BitStream:
    """ref: src/huffman.wgsl:35-79, src/bits.rs:18-67"""

    def __init__(self, words, start=0):
        self._w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
        self._s = _BitsStruct()
        lib().orc_bits_init(C.byref(self._s), self._w.ctypes.data, self._w.size, start)

    def refill(self):
        lib().orc_bits_refill(C.byref(self._s))

    def consume(self, n):
        lib().orc_bits_consume(C.byref(self._s), n)

    def peek(self, n):
        return lib().orc_bits_peek(C.byref(self._s), n)

    def huffdecode(self, table):
        return lib().orc_bits_huffdecode_table(C.byref(self._s), table._p)

    @property
    def left(self):
        return self._s.left


def huff_extend(v, t):
    return lib().orc_huff_extend(v, t)


def parser_dump(jpeg):
    ptr, n, keep = _buf(jpeg)
    need = lib().orc_parser_dump(ptr, n, None, 0)
    out = C.create_string_buffer(need + 1)
    lib().orc_parser_dump(ptr, n, out, need + 1)
    return out.value.decode("latin-1")


class ImageData:
    """ref: src/lib.rs:576-851"""

    def __init__(self, jpeg, allow_sampling=False, standard_entropy=False):
        """Extensions beyond the reference: allow_sampling -- 4:4:4, 4:4:0 and 4:2:0 are accepted too;
        standard_entropy -- refill in front of DC codes and ZRL = 16 positions, as T.81 has it."""
        self.jpeg = bytes(jpeg)
        ptr, n, self._keep = _buf(self.jpeg)
        err = C.create_string_buffer(ERRLEN)
        if allow_sampling or standard_entropy:
            self._p = lib().orc_image_parse_ext(ptr, n, (1 if allow_sampling else 0) | (2 if standard_entropy else 0), err)
        else:
            self._p = lib().orc_image_parse(ptr, n, err)
        if not self._p:
            raise OracleError(err.value.decode())

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_image_free(self._p)
            self._p = None

    def width(self):
        return lib().orc_image_width(self._p)

    def height(self):
        return lib().orc_image_height(self._p)

    def parallelism(self):
        return lib().orc_image_parallelism(self._p)

    def metadata(self):
        return C.string_at(lib().orc_image_metadata(self._p), 1112)

    def l1(self):
        return C.string_at(lib().orc_image_l1(self._p), 2048)

    def l2(self):
        n = C.c_size_t()
        p = lib().orc_image_l2(self._p, C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def scan_range(self):
        o, n = C.c_size_t(), C.c_size_t()
        lib().orc_image_scan(self._p, C.byref(o), C.byref(n))
        return o.value, n.value

    def scan_data(self):
        o, n = self.scan_range()
        return self.jpeg[o:o + n]

    def total_dus(self):
        md = np.frombuffer(self.metadata(), dtype=np.uint32)
        return int(md[272]) * int(md[256]) * int(md[276])

    def decode(self, tex_w=None, tex_h=None, want_coefficients=False, strict=False):
        """Full reference path.  Returns rgba [tex_h, tex_w, 4] u8 (and the
        post-huffman int32 coefficient buffer when asked)."""
        tex_w = tex_w or self.width()
        tex_h = tex_h or self.height()
        rgba = np.zeros((tex_h, tex_w, 4), dtype=np.uint8)
        coef = np.zeros(max(1, self.total_dus() * 32), dtype=np.int32) if want_coefficients else None
        err = C.create_string_buffer(ERRLEN)
        ptr, n, keep = _buf(self.jpeg)
        rc = lib().orc_image_decode(self._p, ptr, rgba.ctypes.data, tex_w, tex_h,
                                    coef.ctypes.data if coef is not None else None, err)
        if rc and strict:
            raise OracleError(err.value.decode())
        self.last_warning = err.value.decode() if rc else None
        return (rgba, coef[: self.total_dus() * 32]) if want_coefficients else rgba
