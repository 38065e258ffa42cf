"""compeg_amd -- host-side mirror of the reference's public API over the C ABI of
libcompeg_hip.so (hand-written gfx950 kernels).

Same names, argument meaning and error behaviour as SludgePhD/Compeg's Rust
crate (src/lib.rs), so tests read like the reference's own:

    gpu = Gpu.open()
    decoder = Decoder(gpu)
    data = ImageData(jpeg_bytes)          # raises compeg_amd.Error like ImageData::new -> Err
    op = decoder.decode_blocking(data)
    rgba = decoder.read_texture(data.width(), data.height())

Everything that computes runs in the native library; this package only moves
handles around.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from ._lib import (E_COUNT_MISMATCH, E_HIP, E_INVALID_ARG, E_MALFORMED, E_UNSUPPORTED, L1_BYTES,
                   LIB_PATH, METADATA_BYTES, Error, check, lib)

__all__ = ["Gpu", "Decoder", "DecodeOp", "ImageData", "ScanBuffer", "Batch", "Texture", "Error", "HostBuffer", "JpegList",
           "host_register", "host_unregister", "version", "LIB_PATH"]


# compeg_decoder_last_kernel / compeg_batch_last_kernel (include/compeg_hip.h: COMPEG_KERNEL_*)
KERNEL_NAMES = ("none", "fused", "pair", "coop_team", "generic", "split", "fused_layout", "fused_stream", "walk_mcu")


def version():
    return lib.compeg_version().decode()


def _host_view(data):
    a = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
    return np.ascontiguousarray(a)


class HostBuffer:
    """Page-locked host memory (compeg_host_alloc).  JPEG bytes that lie here -- `place()` packs a list of them -- go
    to the card without being touched on the host when a Batch with device preprocessing uploads them: the
    `Cow::Borrowed` road of ImageData::new + the uploads straight from the borrowed bytes (src/lib.rs:577-595,397-407)."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib.compeg_host_alloc(nbytes, C.byref(p)))
        self._p, self.nbytes = p, nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(nbytes, 1)).from_address(p.value))[:nbytes]

    def place(self, jpegs, align=64):
        """Copies the byte strings in, one behind the other; returns views of them (to hand to Batch.upload_jpegs)."""
        views, at = [], 0
        for j in jpegs:
            a = _host_view(j)
            if at + a.nbytes > self.nbytes:
                raise ValueError("HostBuffer too small")
            self.array[at:at + a.nbytes] = a
            views.append(self.array[at:at + a.nbytes])
            at = (at + a.nbytes + align - 1) // align * align
        return views

    def close(self):
        if self._p:
            self.array = None
            lib.compeg_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_feed_work(jpegs, host_threads, road, reps=1):
    """Seconds the host's share of feeding this batch takes, `reps` times over (compeg_host_feed_work: no device)."""
    jl = jpegs if isinstance(jpegs, JpegList) else JpegList(jpegs)
    sec = C.c_double()
    check(lib.compeg_host_feed_work(jl.ptrs, jl.lens, jl.count, host_threads, 0, road, reps, C.byref(sec)))
    return sec.value


def host_register(array):
    """Page-locks a caller-owned buffer (numpy array) in place: compeg_host_register."""
    check(lib.compeg_host_register(C.c_void_p(array.ctypes.data), array.nbytes))


def host_unregister(array):
    check(lib.compeg_host_unregister(C.c_void_p(array.ctypes.data)))


class JpegList:
    """The (pointer, length) arrays compeg_batch_upload_jpegs takes, gathered once from bytes-like objects (which this
    object keeps alive)."""

    def __init__(self, jpegs):
        self.views = [_host_view(j) for j in jpegs]
        self.count = len(self.views)
        self.ptrs = (C.c_void_p * self.count)(*[v.__array_interface__["data"][0] for v in self.views])
        self.lens = (C.c_size_t * self.count)(*[v.nbytes for v in self.views])


class Gpu:
    """An open handle to one MI355X (ref: `Gpu`, src/lib.rs:64-270)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def open(cls, device=-1):
        """ref: Gpu::open (src/lib.rs:78-102)."""
        h = C.c_void_p()
        check(lib.compeg_gpu_open(device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_stream(cls, device, hip_stream):
        """`Gpu::from_wgpu` analogue (src/lib.rs:105): adopt a caller-owned HIP stream
        (an int / pointer value, e.g. torch.cuda.current_stream().cuda_stream)."""
        h = C.c_void_p()
        check(lib.compeg_gpu_from_stream(device, C.c_void_p(hip_stream), C.byref(h)))
        return cls(h)

    def device(self):
        return lib.compeg_gpu_device(self._h)

    def name(self):
        return lib.compeg_gpu_name(self._h).decode()

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (lib is None once the interpreter is shutting down)
            lib.compeg_gpu_release(self._h)
            self._h = None


class ImageData:
    """A parsed JPEG (ref: `ImageData`, src/lib.rs:576-851)."""

    def __init__(self, jpeg, copy=True, allow_sampling=False, standard_entropy=False):
        """Extensions beyond the reference (compeg_image_parse_ext): allow_sampling -- 4:4:4, 4:4:0 and
        4:2:0 are accepted as well as 4:2:2; standard_entropy -- the bit reader is topped up in front of
        DC codes and ZRL skips 16 positions, as ITU-T T.81 has it (the reference deviates in both)."""
        self._h = None
        self._keep = _host_view(jpeg)
        h = C.c_void_p()
        flags = (1 if allow_sampling else 0) | (2 if standard_entropy else 0)
        if flags:
            check(lib.compeg_image_parse_ext(self._keep.ctypes.data, self._keep.nbytes, 1 if copy else 0, flags,
                                             C.byref(h)))
        else:
            check(lib.compeg_image_parse(self._keep.ctypes.data, self._keep.nbytes, 1 if copy else 0, C.byref(h)))
        self._h = h
        if copy:
            self._keep = None

    new = classmethod(lambda cls, jpeg: cls(jpeg))

    def width(self):
        return lib.compeg_image_width(self._h)

    def height(self):
        return lib.compeg_image_height(self._h)

    def parallelism(self):
        return lib.compeg_image_parallelism(self._h)

    # what the reference uploads for this image (src/lib.rs:397-407)
    def metadata(self):
        return C.string_at(lib.compeg_image_metadata(self._h), METADATA_BYTES)

    def huffman_l1(self):
        return C.string_at(lib.compeg_image_huffman_l1(self._h), L1_BYTES)

    def huffman_l2(self):
        n = C.c_size_t()
        p = lib.compeg_image_huffman_l2(self._h, C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def scan_range(self):
        o, n = C.c_size_t(), C.c_size_t()
        lib.compeg_image_scan_range(self._h, C.byref(o), C.byref(n))
        return o.value, n.value

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (lib is None once the interpreter is shutting down)
            lib.compeg_image_free(self._h)
            self._h = None


class ScanBuffer:
    """ref: `ScanBuffer` (src/scan.rs:15-77; bench-only re-export, src/lib.rs:44-46)."""

    def __init__(self):
        self._h = C.c_void_p(lib.compeg_scanbuffer_new())

    def process(self, scan_data, expected_restart_intervals):
        a = _host_view(scan_data)
        check(lib.compeg_scanbuffer_process(self._h, a.ctypes.data, a.nbytes, expected_restart_intervals))

    def set_threads(self, threads):
        """Extension: threads that share one process() call (same output)."""
        check(lib.compeg_scanbuffer_set_threads(self._h, threads))

    def process_on_gpu(self, gpu, scan_data, expected_restart_intervals):
        """Same buffers, filled by the device-side scan kernels (extension)."""
        a = _host_view(scan_data)
        check(lib.compeg_scanbuffer_process_on_gpu(self._h, gpu._h, a.ctypes.data, a.nbytes,
                                                   expected_restart_intervals))

    def _get(self, fn):
        n = C.c_size_t()
        p = fn(self._h, C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def processed_scan_data(self):
        return self._get(lib.compeg_scanbuffer_data)

    def start_positions(self):
        return self._get(lib.compeg_scanbuffer_start_positions)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (lib is None once the interpreter is shutting down)
            lib.compeg_scanbuffer_free(self._h)
            self._h = None


class Texture:
    """Device-resident RGBA8 output (the role of wgpu::Texture in the reference).
    Exposes __cuda_array_interface__ so torch / cupy can wrap it without a copy."""

    def __init__(self, ptr, width, height, pitch, owner=None, owned=False):
        self.ptr, self.width, self.height, self.pitch = ptr, width, height, pitch
        self._owner, self._owned = owner, owned

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.height, self.width, 4), "typestr": "|u1", "data": (self.ptr, False),
                "strides": (self.pitch, 4, 1), "version": 3}

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr and lib is not None:
            lib.compeg_device_free(C.c_void_p(self.ptr))
            self.ptr = None


class DecodeOp:
    """ref: `DecodeOp` (src/lib.rs:541-574)."""

    def __init__(self, handle, decoder):
        self._h, self._decoder = handle, decoder

    def wait(self):
        check(lib.compeg_op_wait(self._h))

    def texture(self):
        return self._decoder.texture()

    def texture_changed(self):
        return bool(lib.compeg_op_texture_changed(self._h))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (lib is None once the interpreter is shutting down)
            lib.compeg_op_free(self._h)
            self._h = None


class Decoder:
    """A GPU JPEG decode context (ref: `Decoder`, src/lib.rs:273-529).  One thread at a time."""

    def __init__(self, gpu):
        self._gpu = gpu
        h = C.c_void_p()
        check(lib.compeg_decoder_new(gpu._h, C.byref(h)))
        self._h = h

    new = classmethod(lambda cls, gpu: cls(gpu))

    def enqueue(self, data, hip_stream=0):
        """ref: Decoder::enqueue (src/lib.rs:385).  Returns texture_changed."""
        changed = C.c_int()
        check(lib.compeg_decoder_enqueue(self._h, data._h, C.c_void_p(hip_stream), C.byref(changed)))
        return bool(changed.value)

    def start_decode(self, data):
        op = C.c_void_p()
        check(lib.compeg_decoder_start_decode(self._h, data._h, C.byref(op)))
        return DecodeOp(op, self)

    def decode_blocking(self, data):
        op = C.c_void_p()
        check(lib.compeg_decoder_decode_blocking(self._h, data._h, C.byref(op)))
        return DecodeOp(op, self)

    def last_warning(self):
        return lib.compeg_decoder_last_warning(self._h).decode()

    def last_stage_times(self):
        """Host microseconds of the last decode's stages: the reference's three trace timers
        (t_preprocess, t_enqueue_writes, t_poll: src/lib.rs:391-396,452-475,516-522)."""
        t = (C.c_double * 3)()
        check(lib.compeg_decoder_last_stage_times(self._h, t))
        return {"preprocess_us": t[0], "enqueue_writes_us": t[1], "poll_us": t[2]}

    def last_kernel(self):
        """Diagnostics: the decode kernel the last enqueue went to (KERNEL_* names)."""
        return KERNEL_NAMES[lib.compeg_decoder_last_kernel(self._h)]

    def set_device_preprocess(self, on=True):
        """Extension: preprocess scans with the device-side scan kernels instead of on the host."""
        check(lib.compeg_decoder_set_device_preprocess(self._h, 1 if on else 0))

    def set_scan_threads(self, threads):
        """Extension: threads of this decoder's host scan preprocessor."""
        check(lib.compeg_decoder_set_scan_threads(self._h, threads))

    def texture(self):
        p, w, h, pitch = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_size_t()
        check(lib.compeg_decoder_output(self._h, C.byref(p), C.byref(w), C.byref(h), C.byref(pitch)))
        return Texture(p.value, w.value, h.value, pitch.value, owner=self)

    def into_texture(self):
        p, w, h, pitch = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_size_t()
        check(lib.compeg_decoder_take_output(self._h, C.byref(p), C.byref(w), C.byref(h), C.byref(pitch)))
        self._h = None
        return Texture(p.value, w.value, h.value, pitch.value, owned=True)

    def read_texture(self, width, height):
        """Tightly packed host copy of the WxH corner (what src/tests.rs:52-84 does)."""
        out = np.empty((height, width, 4), dtype=np.uint8)
        check(lib.compeg_decoder_read_output(self._h, out.ctypes.data, width, height))
        return out

    def read_coefficients(self, total_dus):
        out = np.empty(total_dus * 32, dtype=np.int32)
        check(lib.compeg_decoder_read_coefficients(self._h, out.ctypes.data, out.size))
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (lib is None once the interpreter is shutting down)
            lib.compeg_decoder_free(self._h)
            self._h = None


class Batch:
    """Extension (not in the reference): many independent images per launch sequence."""

    def __init__(self, gpu):
        self._gpu = gpu
        h = C.c_void_p()
        check(lib.compeg_batch_new(gpu._h, C.byref(h)))
        self._h = h
        self._images = []

    def upload(self, images, host_threads=0):
        self._images = list(images)
        arr = (C.c_void_p * len(self._images))(*[im._h for im in self._images])
        check(lib.compeg_batch_upload(self._h, arr, len(self._images), host_threads))

    def upload_jpegs(self, jpegs, host_threads=0, allow_sampling=False, standard_entropy=False):
        """Host-fed use: parse (on the worker threads), preprocess, upload.  jpegs: bytes-like objects, or a
        JpegList made from them once (saves this wrapper's per-call pointer gathering, nothing else)."""
        jl = jpegs if isinstance(jpegs, JpegList) else JpegList(jpegs)
        flags = (1 if allow_sampling else 0) | (2 if standard_entropy else 0)
        self._images = []
        check(lib.compeg_batch_upload_jpegs(self._h, jl.ptrs, jl.lens, jl.count, host_threads, flags))

    def upload_jpegs_begin(self, jpegs, host_threads=0, allow_sampling=False, standard_entropy=False):
        """compeg_batch_upload_jpegs_begin: returns when the transfers are queued; upload_end() (or decode()) finishes."""
        jl = jpegs if isinstance(jpegs, JpegList) else JpegList(jpegs)
        flags = (1 if allow_sampling else 0) | (2 if standard_entropy else 0)
        self._images = []
        self._feeding = jl   # (the bytes stay alive until the upload has ended)
        check(lib.compeg_batch_upload_jpegs_begin(self._h, jl.ptrs, jl.lens, jl.count, host_threads, flags))

    def upload_end(self):
        check(lib.compeg_batch_upload_end(self._h))
        self._feeding = None

    def set_device_preprocess(self, mode):
        """0 host (default), 1 scan kernels once at upload, 2 scan kernels in every decode."""
        check(lib.compeg_batch_set_device_preprocess(self._h, mode))

    def host_fallbacks(self):
        return lib.compeg_batch_host_fallbacks(self._h)

    def set_chunk(self, images_per_launch):
        check(lib.compeg_batch_set_chunk(self._h, images_per_launch))

    def decode(self, hip_stream=0):
        check(lib.compeg_batch_decode(self._h, C.c_void_p(hip_stream)))

    def wait(self):
        check(lib.compeg_batch_wait(self._h))

    def count(self):
        return lib.compeg_batch_count(self._h)

    def output(self, index):
        p, w, h, pitch = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_size_t()
        check(lib.compeg_batch_output(self._h, index, C.byref(p), C.byref(w), C.byref(h), C.byref(pitch)))
        return Texture(p.value, w.value, h.value, pitch.value, owner=self)

    def read_output(self, index):
        t = self.output(index)
        out = np.empty((t.height, t.width, 4), dtype=np.uint8)
        check(lib.compeg_batch_read_output(self._h, index, out.ctypes.data))
        return out

    def algorithmic_bytes(self):
        return lib.compeg_batch_algorithmic_bytes(self._h)

    def pixels(self):
        return lib.compeg_batch_pixels(self._h)

    def last_kernel(self):
        """Diagnostics: the decode kernel the first launch of the last decode went to (KERNEL_NAMES)."""
        return KERNEL_NAMES[lib.compeg_batch_last_kernel(self._h)]

    def set_timing(self, on=True):
        """compeg_batch_set_timing: off = decodes record no timing events (they run closer together)."""
        check(lib.compeg_batch_set_timing(self._h, 1 if on else 0))

    def timing(self, reset=True):
        """(decodes, total_ms, huffman_ms, idct_composite_ms) summed over the decodes since the
        last upload/reset, from HIP events recorded on the stream the kernels ran on."""
        n, total = C.c_uint32(), C.c_double()
        stages = (C.c_double * 2)()
        check(lib.compeg_batch_timing(self._h, 1 if reset else 0, C.byref(n), C.byref(total), stages))
        return n.value, total.value, stages[0], stages[1]

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (lib is None once the interpreter is shutting down)
            lib.compeg_batch_free(self._h)
            self._h = None
