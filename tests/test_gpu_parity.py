"""GPU parity tests: the HIP path, called through the C ABI (compeg_amd is a thin ctypes
mirror of the reference's API), against the CPU oracle on the same inputs.  Bit-exact."""
import hashlib
import os

import numpy as np
import pytest

from conftest import read_golden
from oracle import oracle as orc
from tools import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    import compeg_amd
    return compeg_amd


@pytest.fixture(scope="module")
def gpu(ca):
    return ca.Gpu.open(0)


def _decode(ca, gpu, jpeg):
    dec = ca.Decoder(gpu)
    data = ca.ImageData(jpeg)
    dec.decode_blocking(data)
    return dec, data, dec.read_texture(data.width(), data.height())


def _assert_equal(got, want):
    if not np.array_equal(got, want):
        diff = (got != want).any(axis=2)
        ys, xs = np.nonzero(diff)
        raise AssertionError(f"{diff.sum()} pixels differ; first at x={xs[0]} y={ys[0]}: "
                             f"got {got[ys[0], xs[0]]} want {want[ys[0], xs[0]]}")


def _fnv1a_fast(flat):
    """64-bit FNV-1a of a uint8 array without a Python loop per byte: the hash is h -> (h ^ b) * P mod 2^64, which a small
    C routine does in a moment -- compiled on the fly with gcc (test infrastructure only)."""
    import ctypes
    import subprocess
    import tempfile
    src = "unsigned long long f(const unsigned char*p,unsigned long long n){unsigned long long h=14695981039346656037ull;for(unsigned long long i=0;i<n;i++){h^=p[i];h*=1099511628211ull;}return h;}"
    d = tempfile.mkdtemp()
    open(os.path.join(d, "f.c"), "w").write(src)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(d, "f.c"), "-o", os.path.join(d, "f.so")])
    lib = ctypes.CDLL(os.path.join(d, "f.so"))
    lib.f.restype = ctypes.c_ulonglong
    lib.f.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong]
    buf = np.ascontiguousarray(flat)
    return int(lib.f(buf.ctypes.data, buf.size))


def test_native_library_is_loaded(ca, gpu):
    assert "gfx950" in gpu.name()
    maps = open("/proc/self/maps").read()
    assert "libcompeg_hip.so" in maps


@pytest.mark.parametrize("name", ["64x8-Ri-1.jpg", "64x8-Ri-2.jpg"])
def test_reference_reftest_images(ca, gpu, name):
    """ref src/tests.rs:131-135, but bit-exact against the oracle instead of +-3 vs the PNG."""
    jpeg = read_golden("refs", name)
    dec, data, got = _decode(ca, gpu, jpeg)
    _assert_equal(got, orc.ImageData(jpeg).decode())
    assert hashlib.sha256(got.tobytes()).hexdigest() == \
        "30d5ae4c2ae877f80b33d923736c97f164e424ab7bb21bb23a26d0c707a944c6"


def test_reference_444_fixture_through_the_extension(ca, gpu):
    """ref src/tests.rs:137-142 (`#[ignore]`d there): the reference's own 4:4:4 fixture, accepted through
    compeg_image_parse_ext; bit-exact against the oracle's extension, which tests/test_oracle_pinning.py
    pins to the reference's 64x8.png within its +-3.  The default parse keeps the reference's rejection."""
    jpeg = read_golden("refs", "64x8-Hi1-Vi1.jpg")
    with pytest.raises(ca.Error) as e:
        ca.ImageData(jpeg)
    assert "invalid sampling factors 1x1 for Y component" in str(e.value)
    dec = ca.Decoder(gpu)
    data = ca.ImageData(jpeg, allow_sampling=True)
    dec.decode_blocking(data)
    _assert_equal(dec.read_texture(data.width(), data.height()), orc.ImageData(jpeg, allow_sampling=True).decode())


def test_reference_mjpeg_fixture(ca, gpu):
    """960x720 webcam MJPEG frame, DRI=10, no DHT (Annex-K tables), contains ZRL runs."""
    jpeg = read_golden("parser", "mjpeg.jpg")
    dec, data, got = _decode(ca, gpu, jpeg)
    assert data.parallelism() == 540
    assert dec.last_kernel() == "coop_team"   # (10 MCUs per interval: six whole intervals per team)
    _assert_equal(got, orc.ImageData(jpeg).decode())
    assert hashlib.sha256(got.tobytes()).hexdigest() == \
        "502b1b9c9a0401b20a04c3220710ae6c6c8a1068719a58018739b258602f9905"


CASES = [
    # w, h, kind, quality, ri, seed
    (64, 8, 0, 100, 1, 1),
    (640, 360, 0, 85, 4, 2),
    (256, 64, 1, 95, 1, 3),       # random RGB: long codes, L2 LUT, ZRL (quirk Q2)
    (250, 70, 0, 50, 3, 4),       # ragged: width/height not MCU multiples
    (33, 17, 0, 90, 1, 5),        # tiny ragged
    (512, 128, 2, 85, 7, 6),      # mostly-EOB blocks, odd restart interval
    (1920, 1080, 0, 85, 4, 7),    # config 3 frame
    (1920, 1080, 1, 95, 120, 8),  # one interval per MCU row, max entropy
    (3840, 2160, 0, 85, 4, 9),    # config 2
]


@pytest.mark.parametrize("w,h,kind,q,ri,seed", CASES)
def test_synthetic_parity(ca, gpu, w, h, kind, q, ri, seed):
    jpeg = synth.make_jpeg(w, h, seed=seed, kind=kind, quality=q, ri=ri)
    dec, data, got = _decode(ca, gpu, jpeg)
    ref = orc.ImageData(jpeg)
    want, coef = ref.decode(want_coefficients=True)
    got_coef = dec.read_coefficients(ref.total_dus())
    assert np.array_equal(got_coef, coef), "huffman stage differs from the oracle"
    _assert_equal(got, want)


def test_no_dht_and_jfif_variants(ca, gpu):
    for flags in (synth.NO_DHT, synth.JFIF, synth.NO_DHT | synth.JFIF):
        jpeg = synth.make_jpeg(320, 240, seed=11, ri=5, flags=flags)
        _, _, got = _decode(ca, gpu, jpeg)
        _assert_equal(got, orc.ImageData(jpeg).decode())


def test_no_dri_single_interval(ca, gpu):
    """Without DRI the whole image is one restart interval (lib.rs:784)."""
    jpeg = synth.make_jpeg(128, 64, seed=12, ri=0)
    dec, data, got = _decode(ca, gpu, jpeg)
    assert data.parallelism() == 1
    _assert_equal(got, orc.ImageData(jpeg).decode())


def test_decoder_takes_the_walk_route_for_long_intervals(ca, gpu):
    """compeg_decoder (the reference's Decoder: one image at a time) takes the walk + lane-per-MCU route where the
    cooperative kernel cannot take the image or is the slower of the two (runtime.cpp: coop_preferred): no DRI at all
    (the whole image one restart interval of 1800 MCUs, lib.rs:784: a single lane would decode it otherwise),
    intervals of 300 MCUs (longer than the cooperative kernel's teams take), a scan with flipped bits (the walk's slow
    road), the same decoder used for all of them and for images of the cooperative kernel's in between (DRI = 4; an
    interval of 250 MCUs that does not divide the image: its speculative walks); a blocking decode with the scan
    preprocessed on the device keeps the other kernels.  Bit-exact either way."""
    cases = [(synth.make_jpeg(640, 360, seed=61, quality=85, ri=0), "walk_mcu"),
             (synth.make_jpeg(960, 720, seed=62, quality=85, ri=4), "coop_team"),
             (synth.make_jpeg(1280, 720, seed=63, quality=85, ri=300), "walk_mcu"),
             (synth.make_jpeg(1280, 720, seed=66, quality=85, ri=250), "coop_team"),
             (_flip_bits(synth.make_jpeg(1280, 720, seed=64, quality=85, ri=300), 640, flips=12), "walk_mcu"),
             (synth.make_jpeg(640, 360, seed=65, kind=1, quality=70, ri=0), None)]   # (dense: whichever kernel the terms name)
    dec = ca.Decoder(gpu)
    for jpeg, kernel in cases:
        data = ca.ImageData(jpeg)
        want = orc.ImageData(jpeg).decode()
        if jpeg in (cases[3][0], cases[4][0]):
            # (250 MCUs an interval leave the image's last 200 MCUs unwritten, lib.rs:784-785; so may a scan that has lost
            # a marker: a texture of their own, like the oracle's -- the decoder before them has written every texel)
            dec = ca.Decoder(gpu)
        dec.decode_blocking(data)
        if kernel:
            assert dec.last_kernel() == kernel, dec.last_kernel()
        _assert_equal(dec.read_texture(data.width(), data.height()), want)
    dev = ca.Decoder(gpu)
    dev.set_device_preprocess(True)
    # (with device preprocessing a blocking decode sends the descriptors up in front of the scan kernels, which fill in
    # the scan's counts -- of one descriptor: the cooperative kernel's images keep it; images whose intervals only the
    # route decodes well, beyond 256 MCUs, have the counts read back first and take the route)
    for jpeg, kernel in cases[:4]:
        data = ca.ImageData(jpeg)
        dev.decode_blocking(data)
        assert dev.last_kernel() == kernel, dev.last_kernel()
        _assert_equal(dev.read_texture(data.width(), data.height()), orc.ImageData(jpeg).decode())
        if kernel == "walk_mcu":
            dev = ca.Decoder(gpu)
            dev.set_device_preprocess(True)
    # (a non-blocking decode has the counts in hand when it writes the descriptors, whatever the interval)
    jpeg = cases[2][0]
    data = ca.ImageData(jpeg)
    dev.start_decode(data).wait()
    assert dev.last_kernel() == "walk_mcu"
    _assert_equal(dev.read_texture(data.width(), data.height()), orc.ImageData(jpeg).decode())


def test_walk_route_dc_code_cut_by_the_readers_buffer(ca, gpu):
    """tests/golden/route/cut_dc_code.jpg (see its README): the reference's reader, six bits left in front of a DC code,
    reads a code the stream's own bits do not hold and runs dry -- the route's walk hands that MCU to its slow road.
    Through the decoder and in a batch beside an ordinary frame."""
    jpeg = read_golden("route", "cut_dc_code.jpg")
    want = orc.ImageData(jpeg).decode()
    dec, data, got = _decode(ca, gpu, jpeg)
    assert dec.last_kernel() == "walk_mcu"
    _assert_equal(got, want)
    other = synth.make_jpeg(1016, 408, seed=91, quality=85, ri=0)
    batch = ca.Batch(gpu)
    batch.upload([ca.ImageData(jpeg), ca.ImageData(other), ca.ImageData(jpeg)])
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "walk_mcu"
    _assert_equal(batch.read_output(0), want)
    _assert_equal(batch.read_output(1), orc.ImageData(other).decode())
    _assert_equal(batch.read_output(2), want)


def test_decoder_reuse_grows_and_reports_texture_changed(ca, gpu):
    """ref lib.rs:564-573 + dynamic.rs:214-248: first decode and every growth report true."""
    dec = ca.Decoder(gpu)
    small = synth.make_jpeg(320, 200, seed=20)
    big = synth.make_jpeg(800, 600, seed=21)
    a, b = ca.ImageData(small), ca.ImageData(big)
    assert dec.decode_blocking(a).texture_changed() is True
    assert dec.decode_blocking(a).texture_changed() is False
    assert dec.decode_blocking(b).texture_changed() is True
    _assert_equal(dec.read_texture(800, 600), orc.ImageData(big).decode())
    # smaller image into the larger texture: only the WxH corner is defined
    assert dec.decode_blocking(a).texture_changed() is False
    tex = dec.texture()
    assert (tex.width, tex.height) == (800, 600)
    want = orc.ImageData(small).decode(tex_w=800, tex_h=600)
    _assert_equal(dec.read_texture(320, 200), want[:200, :320])


def test_enqueue_on_caller_stream_and_start_decode(ca, gpu):
    jpeg = synth.make_jpeg(640, 480, seed=30, ri=2)
    want = orc.ImageData(jpeg).decode()
    dec = ca.Decoder(gpu)
    data = ca.ImageData(jpeg)
    assert dec.enqueue(data, 0) is True          # default stream
    _assert_equal(dec.read_texture(640, 480), want)
    op = dec.start_decode(data)
    op.wait()
    assert op.texture_changed() is False
    _assert_equal(dec.read_texture(640, 480), want)


def test_stage_times_through_the_c_abi(ca, gpu):
    """The reference traces three host timers per decode (t_preprocess, t_enqueue_writes, t_poll:
    src/lib.rs:391-396,452-475,516-522); here they come back through compeg_decoder_last_stage_times."""
    import time
    jpeg = synth.make_jpeg(1920, 1080, seed=33, ri=4)
    data = ca.ImageData(jpeg)
    dec = ca.Decoder(gpu)
    assert dec.last_stage_times() == {"preprocess_us": 0.0, "enqueue_writes_us": 0.0, "poll_us": 0.0}
    dec.decode_blocking(data)
    t0 = time.perf_counter()
    dec.decode_blocking(data)
    wall_us = (time.perf_counter() - t0) * 1e6
    t = dec.last_stage_times()
    assert t["preprocess_us"] > 0 and t["enqueue_writes_us"] > 0 and t["poll_us"] > 0
    assert sum(t.values()) <= wall_us * 1.05
    for threads_device in ((1, False), (4, True)):
        dec.set_scan_threads(threads_device[0])
        dec.set_device_preprocess(threads_device[1])
        dec.decode_blocking(data)
        t = dec.last_stage_times()
        assert t["preprocess_us"] > 0 and t["poll_us"] > 0
    dec.start_decode(data).wait()
    assert dec.last_stage_times()["poll_us"] == 0.0      # nothing waited inside start_decode


def test_count_mismatch_is_a_warning_like_the_reference(ca, gpu):
    """lib.rs:391-394 drops process() errors; we decode and surface the text as a warning."""
    jpeg = bytearray(synth.make_jpeg(128, 32, seed=40, ri=2))
    # corrupt the DRI segment: claim Ri=4 while the stream has markers every 2 MCUs
    i = jpeg.find(b"\xff\xdd")
    jpeg[i + 4:i + 6] = (4).to_bytes(2, "big")
    dec = ca.Decoder(gpu)
    data = ca.ImageData(bytes(jpeg))
    dec.decode_blocking(data)
    assert dec.last_warning().startswith("restart interval count mismatch: counted")
    ref = orc.ImageData(bytes(jpeg))
    want = ref.decode()
    _assert_equal(dec.read_texture(128, 32), want)


def test_into_texture_transfers_ownership(ca, gpu):
    jpeg = synth.make_jpeg(160, 120, seed=50)
    dec = ca.Decoder(gpu)
    dec.decode_blocking(ca.ImageData(jpeg))
    tex = dec.into_texture()
    assert tex.ptr and (tex.width, tex.height) == (160, 120)
    iface = tex.__cuda_array_interface__
    assert iface["shape"] == (120, 160, 4)


def test_batch_matches_single_decodes(ca, gpu):
    jpegs = [synth.make_jpeg(w, h, seed=60 + i, kind=k, quality=q, ri=ri)
             for i, (w, h, k, q, ri) in enumerate([(640, 360, 0, 85, 4), (320, 240, 1, 95, 1),
                                                   (1280, 720, 0, 70, 8), (64, 8, 0, 100, 2),
                                                   (250, 70, 2, 85, 3)])]
    images = [ca.ImageData(j) for j in jpegs]
    batch = ca.Batch(gpu)
    batch.upload(images)
    batch.decode()
    batch.wait()
    assert batch.count() == len(jpegs)
    for i, j in enumerate(jpegs):
        _assert_equal(batch.read_output(i), orc.ImageData(j).decode())
    n, total, huff, idct = batch.timing()
    assert n == 1 and total > 0 and huff > 0 and idct >= 0   # (one kernel does the whole path: no event in the middle)
    # chunked launches give the same pixels
    batch.set_chunk(2)
    batch.decode()
    for i, j in enumerate(jpegs):
        _assert_equal(batch.read_output(i), orc.ImageData(j).decode())


def test_batch_fed_with_jpeg_bytes(ca, gpu):
    """compeg_batch_upload_jpegs: ImageData::new on the batch's worker threads, in the pass that preprocesses and
    uploads (layout from the files' headers); same outputs as uploading parsed images; a file the front-end
    rejects fails the call with the reference's message and the image's index; files with unusual headers (a
    fill byte in front of a marker, trailing data) take the parse-first road and decode the same."""
    jpegs = [synth.make_jpeg(w, h, seed=700 + i, kind=k, quality=q, ri=ri)
             for i, (w, h, k, q, ri) in enumerate([(640, 360, 0, 85, 4), (320, 240, 1, 95, 1), (1280, 720, 0, 70, 8),
                                                   (64, 8, 0, 100, 2), (250, 70, 2, 85, 3), (1000, 1000, 0, 80, 16)])]
    wants = [orc.ImageData(j).decode() for j in jpegs]
    batch = ca.Batch(gpu)
    for threads in (1, 3, 8):
        batch.upload_jpegs(jpegs, host_threads=threads)
        assert batch.count() == len(jpegs)
        batch.decode()
        batch.wait()
        for i, want in enumerate(wants):
            _assert_equal(batch.read_output(i), want)
    # a rejected file: progressive SOF (the reference: "not a baseline JPEG ..."), reported with its index
    bad = bytearray(jpegs[1])
    sof = bad.find(b"\xff\xc0")
    bad[sof + 1] = 0xC2
    with pytest.raises(ca.Error) as e:
        batch.upload_jpegs(jpegs[:2] + [bytes(bad)] + jpegs[3:], host_threads=4)
    assert str(e.value).startswith("image 2: ")
    with pytest.raises(ca.Error):
        ca.ImageData(bytes(bad))
    # unusual but valid framing: a fill byte (FF FF) in front of the SOF marker; the headers are then not "peeked"
    odd = bytearray(jpegs[0])
    sof = odd.find(b"\xff\xc0")
    odd = bytes(odd[:sof] + b"\xff" + odd[sof:])
    if _accepts(orc, odd):
        batch.upload_jpegs([odd, jpegs[2]], host_threads=2)
        batch.decode()
        batch.wait()
        _assert_equal(batch.read_output(0), orc.ImageData(odd).decode())
        _assert_equal(batch.read_output(1), wants[2])
    # extension layouts through the same entry point
    j420 = synth.make_jpeg(320, 200, seed=710, ri=2, sampling=(2, 2))
    batch.upload_jpegs([j420, jpegs[0]], host_threads=2, allow_sampling=True)
    batch.decode()
    batch.wait()
    _assert_equal(batch.read_output(0), orc.ImageData(j420, allow_sampling=True).decode())
    _assert_equal(batch.read_output(1), wants[0])


def _accepts(orc_mod, jpeg):
    try:
        orc_mod.ImageData(jpeg)
        return True
    except orc_mod.OracleError:
        return False


def test_full_size_8k_dri1_roundtrip_properties(ca, gpu):
    """Config 5 (7680x4320, DRI=1): size-independent checks -- alpha is 255 everywhere,
    decoding twice is idempotent, and a sampled set of MCU rows matches the oracle run on a
    crop-equivalent stream is too slow, so compare full output hashes with the oracle once."""
    jpeg = synth.make_jpeg(7680, 4320, seed=70, ri=1, quality=60)
    dec, data, got = _decode(ca, gpu, jpeg)
    assert data.parallelism() == 480 * 540
    assert np.all(got[:, :, 3] == 255)
    again = ca.Decoder(gpu)
    again.decode_blocking(data)
    assert np.array_equal(again.read_texture(7680, 4320), got)
    want = orc.ImageData(jpeg).decode()
    _assert_equal(got, want)


def test_single_large_frame_with_long_intervals_takes_the_streamed_windows(ca, gpu):
    """One 8K frame with DRI = 32: too many data units for the cooperative kernel, 64 intervals too long for a
    whole-interval window -- with the device's preprocessing the decoder's launch goes to
    decode_fused_422_stream_kernel (word counts patched into the descriptor behind the launch's planning); with the
    host's, to the walk + lane-per-MCU route (127 waves of intervals for 1024 SIMDs: 149 against 471 us)."""
    jpeg = synth.make_jpeg(7680, 4320, seed=71, ri=32, quality=75, kind=0)
    want = orc.ImageData(jpeg).decode()
    data = ca.ImageData(jpeg)
    for device in (False, True):
        dec = ca.Decoder(gpu)
        dec.set_device_preprocess(device)
        dec.decode_blocking(data)
        assert dec.last_kernel() == ("fused_stream" if device else "walk_mcu")
        _assert_equal(dec.read_texture(7680, 4320), want)


@pytest.mark.parametrize("every_ri", [0, 1])
def test_fused_kernel_ragged_batch(ca, gpu, every_ri):
    """A batch big enough for the throughput kernel (more than 1024 waves) of images whose
    geometry exercises its corner cases: MCUs cut by the right and bottom edge (stored by
    their own lane), a last wave with unused lanes (they only help their quad store), an
    interval count that is not a multiple of the restart interval.  every_ri = 1: the kernel for
    launches of one-MCU intervals, whose rows leave wave-wide (sixteen MCUs to a store)."""
    shapes = [(1000, 1000, every_ri or 3), (1016, 990, every_ri or 5), (1000, 1004, every_ri or 4)]
    jpegs = [synth.make_jpeg(w, h, seed=200 + i, kind=i % 3, quality=80, ri=ri)
             for i, (w, h, ri) in enumerate(shapes * 11)]
    images = [ca.ImageData(j) for j in jpegs]
    waves = sum((im.parallelism() + 63) // 64 for im in images)
    assert waves > 1024
    batch = ca.Batch(gpu)
    batch.upload(images)
    batch.decode()
    batch.wait()
    for i, j in enumerate(jpegs):
        _assert_equal(batch.read_output(i), orc.ImageData(j).decode())
    # the same batch with the scan preprocess on the device
    dev = ca.Batch(gpu)
    dev.set_device_preprocess(2)
    dev.upload(images)
    dev.decode()
    dev.wait()
    for i in (0, 7, len(jpegs) - 1):
        assert np.array_equal(dev.read_output(i), batch.read_output(i))


@pytest.mark.parametrize("ri,uniform", [(10, True), (16, True), (30, True), (16, False)])
def test_batch_kernel_with_streamed_windows(ca, gpu, ri, uniform):
    """Batches whose restart intervals are too long for whole-interval windows (64 intervals of DRI MCUs per wave)
    go to decode_fused_422_stream_kernel: every lane's stream staged MCU by MCU.  Frames of one stream (the flat
    grid) and mixed images (a grid row per image; ragged edges, a corrupt scan among them)."""
    if uniform:
        # (kind 0: photograph-like, 1.7 bit per pixel -- streams beyond 3 bit per pixel keep whole-interval windows)
        # (DRI = 10: whole windows would still fit -- six waves a CU --, the launch has to want more than that)
        # (more than a wave of intervals per SIMD: smaller launches of long intervals take the walk + lane-per-MCU route,
        # test_walk_route_batches)
        frames = [synth.make_jpeg(960, 720, seed=900 + i + ri, kind=0, quality=85, ri=ri) for i in range(32)]
        jpegs = [frames[i % 32] for i in range({10: 320, 16: 192, 30: 352}[ri])]
    else:
        shapes = [(1000, 1000), (1016, 990), (936, 1004), (1280, 720)]
        distinct = [synth.make_jpeg(w, h, seed=950 + i + ri, kind=0, quality=(70, 85, 90)[i % 3], ri=ri)
                    for i, (w, h) in enumerate(shapes * 30)]
        jpegs = distinct + (distinct[:24] if ri == 16 else [])
        bad = bytearray(jpegs[5])
        at = bad.find(b"\xff\xda") + 14
        rng = np.random.default_rng(ri)
        for _ in range(40):
            pos = int(rng.integers(at, len(bad) - 2))
            if bad[pos] != 0xFF and bad[pos - 1] != 0xFF:
                bad[pos] ^= 1 << int(rng.integers(0, 8))
                if bad[pos] == 0xFF:
                    bad[pos] = 0xFE
        jpegs[5] = bytes(bad)
    images = [ca.ImageData(j) for j in jpegs]
    batch = ca.Batch(gpu)
    batch.upload(images)
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "fused_stream"
    for i in list(range(0, len(jpegs), 7)) + [5, len(jpegs) - 1]:
        _assert_equal(batch.read_output(i), orc.ImageData(jpegs[i]).decode())
    # the same with the scan preprocess on the device (the descriptors' word counts are patched in there)
    dev = ca.Batch(gpu)
    dev.set_device_preprocess(2)
    dev.upload(images)
    dev.decode()
    dev.wait()
    assert dev.last_kernel() == "fused_stream"
    for i in (0, 5, len(jpegs) - 1):
        assert np.array_equal(dev.read_output(i), batch.read_output(i))


def test_uniform_batch_spans_images_with_its_workgroups(ca, gpu):
    """Frames of one stream (same interval count, same LUT bytes): the throughput kernel runs a one-dimensional
    grid over the waves of all images; 112.5 waves per image, so workgroups and even the image's last wave are
    shared out unevenly.  A batch with one odd image takes the per-image grid; both equal the oracle."""
    jpegs = [synth.make_jpeg(1280, 720, seed=500 + i, kind=i % 3, quality=85, ri=1) for i in range(10)]
    wants = [orc.ImageData(j).decode() for j in jpegs]
    images = [ca.ImageData(j) for j in jpegs]
    assert sum((im.parallelism() + 63) // 64 for im in images) > 1024
    batch = ca.Batch(gpu)
    batch.upload(images)
    batch.decode()
    batch.wait()
    for i, want in enumerate(wants):
        _assert_equal(batch.read_output(i), want)
    # 64 such frames are more than two rounds of workgroups over the chip
    batch.upload([images[i % 10] for i in range(64)])
    batch.decode()
    batch.wait()
    for i in range(64):
        _assert_equal(batch.read_output(i), wants[i % 10])
    odd = synth.make_jpeg(1280, 720, seed=77, kind=1, quality=60, ri=1)      # other quantisers, same LUTs: still uniform
    other = synth.make_jpeg(640, 360, seed=78, kind=0, quality=85, ri=1)     # other interval count: not uniform
    for extra in (odd, other):
        batch.upload(images + [ca.ImageData(extra)])
        batch.decode()
        batch.wait()
        for i, want in enumerate(wants):
            _assert_equal(batch.read_output(i), want)
        _assert_equal(batch.read_output(len(images)), orc.ImageData(extra).decode())


@pytest.mark.parametrize("ri,n", [(2, 230), (8, 830)])
def test_resident_waves_walk_over_their_units(ca, gpu, ri, n):
    """A uniform batch of a few more units of 64 intervals than the chip holds waves (3072): the throughput
    kernel's waves stay and some of them take a second unit, whose window they have asked for while decoding the
    first (DRI = 2: the touch and the interval starts fall on the same data unit; DRI = 8: 32 data units per
    interval; the image's last wave is partly filled in both).  Every slot against the oracle."""
    jpegs = [synth.make_jpeg(640, 360, seed=900 + 10 * ri + i, kind=i % 2, quality=80 + i, ri=ri) for i in range(6)]
    wants = [orc.ImageData(j).decode() for j in jpegs]
    images = [ca.ImageData(j) for j in jpegs]
    waves = (images[0].parallelism() + 63) // 64
    assert images[0].parallelism() % 64 != 0 and 3072 < waves * n < 2 * 3072
    src = [(5 * i + i // 6) % 6 for i in range(n)]
    batch = ca.Batch(gpu)
    batch.upload([images[s] for s in src])
    batch.decode()
    batch.wait()
    for i, s in enumerate(src):
        if not np.array_equal(batch.read_output(i), wants[s]):
            _assert_equal(batch.read_output(i), wants[s])


def test_concurrent_batch_uploads_share_the_worker_threads(ca, gpu):
    """Two batches uploaded at the same time from two threads of the caller (ctypes releases the GIL): one finds the
    library's worker threads busy and brings its own; then the same two one after the other on the kept threads."""
    import threading
    jpegs = [synth.make_jpeg(1280, 720, seed=1200 + i, kind=i % 3, quality=85, ri=4) for i in range(6)]
    wants = [orc.ImageData(j).decode() for j in jpegs]
    batches = [ca.Batch(gpu), ca.Batch(gpu)]
    errors = []

    def feed(k):
        try:
            for _ in range(3):
                batches[k].upload_jpegs([jpegs[(i + k) % 6] for i in range(24)], host_threads=8)
        except Exception as e:  # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=feed, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for k in range(2):
        batches[k].upload_jpegs([jpegs[(i + k) % 6] for i in range(24)], host_threads=8)
        batches[k].decode()
        batches[k].wait()
        for i in range(24):
            _assert_equal(batches[k].read_output(i), wants[(i + k) % 6])


def test_restart_interval_changes_pixels_only_where_the_reference_does(ca, gpu):
    """The restart interval only changes where the DC predictors are reset and how the scan is
    cut into lanes -- except for the reference's quirk Q1: its reader is not refilled in front
    of a DC code, so on rare bit alignments a valid stream underflows it and the rest of that
    interval decodes to something else.  Where that happens depends on the alignment, hence
    on DRI.  The same 1080p source encoded with six DRI values must therefore agree between
    any two encodings exactly where the oracle's outputs agree, and always match the oracle."""
    got, want = [], []
    for ri in (1, 2, 4, 7, 120, 0):
        jpeg = synth.make_jpeg(1920, 1080, seed=77, kind=0, quality=85, ri=ri)
        got.append(_decode(ca, gpu, jpeg)[2])
        want.append(orc.ImageData(jpeg).decode())
        _assert_equal(got[-1], want[-1])
    for a in range(len(got)):
        for b in range(a + 1, len(got)):
            assert np.array_equal(got[a] == got[b], want[a] == want[b])
    # the quirk is rare, but an underflow spoils the rest of its interval: with short intervals
    # (DRI 1, 2, 4, 7) the encodings differ in a few percent of the pixels at most; with one
    # interval per image (DRI 0) everything behind the first underflow differs
    diff = max(int((got[0] != g).any(axis=2).sum()) for g in got[1:4])
    assert diff < 0.05 * 1920 * 1080, diff


def test_two_kernel_pipeline_in_a_subprocess(ca, gpu):
    """The development pipeline (entropy_kernel + idct_composite_kernel) and the forced kernel choices: experiment
    switches of the laboratory build (compeg_amd/csrc/lab.h, libcompeg_hip_lab.so), chosen once per process.  The
    shipped library ignores them (its own run below stays on its dispatch)."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import compeg_amd as ca
from oracle import oracle as orc
from tools import synth
gpu = ca.Gpu.open(0)
for (w, h, k, q, ri, s) in [(640, 360, 0, 85, 4, 2), (250, 70, 1, 50, 3, 4), (1920, 1080, 0, 85, 4, 7)]:
    j = synth.make_jpeg(w, h, seed=s, kind=k, quality=q, ri=ri)
    data = ca.ImageData(j)
    dec = ca.Decoder(gpu)
    dec.decode_blocking(data)
    assert np.array_equal(dec.read_texture(w, h), orc.ImageData(j).decode()), (w, h)
    print("kernel", dec.last_kernel())
print("split ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lab = os.path.join(os.path.dirname(ca.LIB_PATH), "libcompeg_hip_lab.so")
    assert os.path.exists(lab), "make -C compeg_amd/csrc lab (__graft_entry__.build() does)"
    for env_extra, kernels in (({"COMPEG_LIB": lab, "COMPEG_PIPELINE": "split"}, {"split"}),
                               ({"COMPEG_LIB": lab, "COMPEG_COOP": "0", "COMPEG_PAIR": "0"}, {"fused"}),
                               ({"COMPEG_LIB": lab, "COMPEG_COOP": "0", "COMPEG_PAIR": "1"}, {"pair"}),
                               ({"COMPEG_PIPELINE": "split", "COMPEG_COOP": "0"}, {"coop_team"})):   # the shipped library
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "split ok" in r.stdout, r.stdout + r.stderr[-2000:]
        assert {l.split()[1] for l in r.stdout.splitlines() if l.startswith("kernel ")} == kernels, (env_extra, r.stdout)


@pytest.mark.parametrize("sampling", [(1, 1), (1, 2), (2, 2)])
def test_extension_layouts_with_streamed_windows(ca, gpu, sampling):
    """The extension layouts' kernels in their streamed form (decode_fused_444 / _440 / _420_stream_kernel): long
    restart intervals and dense streams, where whole-interval windows cost a CU its waves -- a uniform batch (flat
    grid), a mixed-size one with a corrupt scan (a grid row per image), one large frame through the Decoder."""
    frames = [synth.make_jpeg(960, 720, seed=700 + i, kind=0, quality=85, ri=16, sampling=sampling) for i in range(16)]
    uniform = [frames[i % 16] for i in range(240)]   # (a launch that wants more waves a CU than whole windows leave it)
    shapes = [(1000, 600), (1016, 590), (936, 604), (1280, 720)]
    mixed = [synth.make_jpeg(w, h, seed=760 + i, kind=0, quality=(75, 85)[i % 2], ri=(8, 16, 40)[i % 3], sampling=sampling)
             for i, (w, h) in enumerate(shapes * 16)]
    bad = bytearray(mixed[3])
    at = bad.find(b"\xff\xda") + 14
    rng = np.random.default_rng(5)
    for _ in range(40):
        pos = int(rng.integers(at, len(bad) - 2))
        if bad[pos] != 0xFF and bad[pos - 1] != 0xFF:
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            if bad[pos] == 0xFF:
                bad[pos] = 0xFE
    mixed[3] = bytes(bad)
    for jpegs in (uniform, mixed):
        batch = ca.Batch(gpu)
        batch.upload([ca.ImageData(j, allow_sampling=True) for j in jpegs])
        batch.decode()
        batch.wait()
        assert batch.last_kernel() == "fused_stream"
        for i in list(range(0, len(jpegs), 9)) + [3, len(jpegs) - 1]:
            _assert_equal(batch.read_output(i), orc.ImageData(jpegs[i], allow_sampling=True).decode())
    big = synth.make_jpeg(3840, 2160, seed=799, quality=85, ri=48, sampling=sampling)
    dec = ca.Decoder(gpu)
    dec.decode_blocking(ca.ImageData(big, allow_sampling=True))
    assert dec.last_kernel() == "fused_stream"
    _assert_equal(dec.read_texture(3840, 2160), orc.ImageData(big, allow_sampling=True).decode())


@pytest.mark.parametrize("sampling", [(1, 1), (1, 2), (2, 2)])
def test_extension_layouts(ca, gpu, sampling):
    """4:4:4, 4:4:0 and 4:2:0 (opt-in): bit-exact against the oracle with the same
    extension, through the Decoder and -- mixed with a 4:2:2 image -- through a Batch."""
    # (even restart intervals: the 8-pixel MCUs of 4:4:4 / 4:4:0 are composited in pairs; odd ones: singly)
    cases = [(640, 360, 0, 85, 4, 301), (250, 70, 1, 75, 3, 302), (33, 17, 2, 85, 1, 303), (1920, 1080, 0, 85, 8, 304),
             (250, 70, 0, 85, 2, 306), (33, 17, 1, 90, 6, 307)]
    jpegs = [synth.make_jpeg(w, h, seed=s, kind=k, quality=q, ri=ri, sampling=sampling) for (w, h, k, q, ri, s) in cases]
    for j, (w, h, *_) in zip(jpegs, cases):
        # a fresh decoder per image: texels behind a truncated last interval are only defined
        # (zero) in a new texture, in the reference as here
        dec = ca.Decoder(gpu)
        data = ca.ImageData(j, allow_sampling=True)
        dec.decode_blocking(data)
        want, coef = orc.ImageData(j, allow_sampling=True).decode(want_coefficients=True)
        assert dec.last_kernel() == "fused_layout"     # decode_fused_444 / _440 / _420_kernel: one pass, nothing but RGBA to HBM
        _assert_equal(dec.read_texture(w, h), want)
        assert np.array_equal(dec.read_coefficients(data.total_dus() if hasattr(data, "total_dus") else len(coef) // 32), coef)
    # a batch of one layout: the same fused kernel (40 images: several workgroups per image, images of different sizes)
    same = [jpegs[i % 4] for i in range(40)]   # (restart intervals 4, 3, 1, 8: a batch with an odd one composites singly)
    batch = ca.Batch(gpu)
    batch.upload([ca.ImageData(j, allow_sampling=True) for j in same])
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "fused_layout"
    wants = [orc.ImageData(j, allow_sampling=True).decode() for j in jpegs]
    for i in range(40):
        _assert_equal(batch.read_output(i), wants[i % 4])
    # corrupt streams in this layout (bit flips inside the scan)
    rng = np.random.default_rng(17)
    for it in range(4):
        j = bytearray(jpegs[1])
        scan_at = j.find(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 20))):
            pos = int(rng.integers(scan_at, len(j) - 2))
            if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                j[pos] ^= 1 << int(rng.integers(0, 8))
                if j[pos] == 0xFF:
                    j[pos] = 0xFE
        dec = ca.Decoder(gpu)
        dec.decode_blocking(ca.ImageData(bytes(j), allow_sampling=True))
        _assert_equal(dec.read_texture(250, 70), orc.ImageData(bytes(j), allow_sampling=True).decode())
    # layouts mixed in one batch: the two-kernel route (sample records, generic composite)
    mixed = jpegs[:3] + [synth.make_jpeg(320, 240, seed=305, ri=2)]
    batch = ca.Batch(gpu)
    batch.upload([ca.ImageData(j, allow_sampling=True) for j in mixed])
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "generic"
    for i, j in enumerate(mixed):
        _assert_equal(batch.read_output(i), orc.ImageData(j, allow_sampling=True).decode())


def test_zero_copy_consumer_and_caller_stream(ca, gpu):
    """the decoded frame is consumed where it is -- torch wraps the output through
    __cuda_array_interface__ without a copy (same device pointer), and a decode recorded on the
    consumer's own stream (the viewer's command encoder in the reference, examples/viewer.rs:244-246)
    is ordered in front of the consumer's work on that stream.  Runs in a process of its own that
    imports torch first (torch brings its own copy of the HIP runtime; bench.py has the same order)."""
    import subprocess
    import sys
    code = r'''
import sys
import numpy as np
import torch
sys.path.insert(0, %r)
import compeg_amd as ca
from oracle import oracle as orc
from tools import synth
jpeg = synth.make_jpeg(640, 360, seed=401, ri=4)
want = orc.ImageData(jpeg).decode()
data = ca.ImageData(jpeg)
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
g = ca.Gpu.from_stream(0, stream.cuda_stream)
dec = ca.Decoder(g)
with torch.cuda.stream(stream):
    changed = dec.enqueue(data, stream.cuda_stream)
    tex = dec.texture()
    view = torch.as_tensor(tex, device="cuda")           # no copy
    luma = view[..., :3].to(torch.float32).mean()         # consumer work, same stream, no host sync
    host = view.cpu()
stream.synchronize()
assert changed
assert view.data_ptr() == tex.ptr and tuple(view.shape) == (360, 640, 4)
assert np.array_equal(host.numpy(), want)
assert abs(float(luma) - want[..., :3].mean()) < 1e-3
batch = ca.Batch(g)                                        # a batch output is consumable the same way
batch.upload([data])
batch.decode(stream.cuda_stream)
with torch.cuda.stream(stream):
    bview = torch.as_tensor(batch.output(0), device="cuda")
    assert np.array_equal(bview.cpu().numpy(), want)
# a uniform batch with more units than resident waves (they draw their units from the batch's queue), decoded on two
# streams in a row: the second decode waits for the first, each is whole
frames = [synth.make_jpeg(960, 540, seed=410 + i, ri=2) for i in range(4)]
big = ca.Batch(g)
big.upload([ca.ImageData(frames[i %% 4]) for i in range(104)])
other = torch.cuda.Stream()
big.decode(stream.cuda_stream)
big.decode(other.cuda_stream)
other.synchronize()
assert big.last_kernel() == "fused"
wants = [orc.ImageData(j).decode() for j in frames]
for i in (0, 1, 50, 103):
    assert np.array_equal(torch.as_tensor(big.output(i), device="cuda").cpu().numpy(), wants[i %% 4])
print("zero-copy ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "zero-copy ok" in r.stdout, r.stdout + r.stderr[-3000:]


def _corrupt_variants(n):
    """Bit flips inside the scan of a small image (the generator of the emulation tests): decoding
    runs off the rails -- long codes, huge runs, reads past the interval and past the scan."""
    rng = np.random.default_rng(5)
    base = synth.make_jpeg(192, 48, seed=21, kind=0, quality=75, ri=2)
    scan_at = base.find(b"\xff\xda") + 14
    out = []
    for _ in range(n):
        j = bytearray(base)
        for _ in range(int(rng.integers(1, 30))):
            pos = int(rng.integers(scan_at, len(j) - 2))
            if j[pos] != 0xFF and j[pos - 1] != 0xFF:
                j[pos] ^= 1 << int(rng.integers(0, 8))
                if j[pos] == 0xFF:
                    j[pos] = 0xFE
        out.append(bytes(j))
    return out


def _hostile_table_variants(n):
    """DHT segments with DC categories up to 255 and other symbols a baseline encoder never emits."""
    rng = np.random.default_rng(8)
    base = synth.make_jpeg(128, 32, seed=31, kind=1, quality=90, ri=1)
    out = []
    for it in range(n):
        counts = np.zeros(16, dtype=np.uint8)
        counts[1], counts[2] = 2, 3
        counts[4] = int(rng.integers(1, 4))
        counts[8] = int(rng.integers(0, 6))
        counts[15] = int(rng.integers(0, 30))
        nsym = int(counts.sum())
        syms = rng.integers(0, 256, nsym, dtype=np.uint8)
        tcth = [0x00, 0x01, 0x10, 0x11][it % 4]
        seg = bytes([0xFF, 0xC4]) + (2 + 17 + nsym).to_bytes(2, "big") + bytes([tcth]) + counts.tobytes() + syms.tobytes()
        sos = base.find(b"\xff\xda")
        out.append(base[:sos] + seg + base[sos:])
    return out


def test_corrupt_streams_and_hostile_tables_on_the_gpu(ca, gpu):
    """The rare paths of the entropy decoder (cut DC codes, reader underflow -> exact mode, escapes,
    reads past the window and past the scan) as the GPU executes them -- through the paired kernel
    (single decodes) and through the throughput kernel (one batch of more than 1024 waves) --
    byte-equal to the oracle.  tests/test_kernel_emulation.py runs the same inputs under ASan."""
    variants = []
    for j in _corrupt_variants(12) + _hostile_table_variants(10):
        try:
            want = orc.ImageData(j).decode()
        except orc.OracleError:
            with pytest.raises(ca.Error):
                ca.ImageData(j)
            continue
        variants.append((j, want))
    assert len(variants) >= 12
    for j, want in variants:
        _, _, got = _decode(ca, gpu, j)
        _assert_equal(got, want)
    # the same streams with the scan preprocessed by the device kernels (blocking decode: no read-back between
    # the scan kernels and the decode kernel, window sized from an upper bound) and on one host thread
    for device, threads in ((True, 4), (False, 1)):
        dec = ca.Decoder(gpu)
        dec.set_device_preprocess(device)
        dec.set_scan_threads(threads)
        for j, want in variants:
            data = ca.ImageData(j)
            dec.decode_blocking(data)
            _assert_equal(dec.read_texture(data.width(), data.height()), want)
    images = [ca.ImageData(j) for j, _ in variants]
    n = 1100
    assert sum((images[i % len(images)].parallelism() + 63) // 64 for i in range(n)) > 1024
    batch = ca.Batch(gpu)
    batch.upload([images[i % len(images)] for i in range(n)])
    batch.decode()
    batch.wait()
    for i in list(range(len(variants))) + [n - 1]:
        _assert_equal(batch.read_output(i), variants[i % len(variants)][1])


# Restart intervals the cooperative kernel's teams handle differently (device_types.h: coop_shape): a lane per
# interval through the walk tables (1..4; 4 with the decoding under the walk), speculative walks with whole
# intervals laid end to end in the team's four rounds (5..64: intervals straddle rounds unless 4 * DRI divides 64),
# one interval per team in more than four rounds and longer lists (65..256); 60 / 120 / 240: one interval per MCU
# row of 960 / 1920 / 3840 pixels.
COOP_DRIS = [1, 2, 3, 4, 5, 6, 7, 8, 10, 16, 17, 30, 60, 64, 65, 120, 240, 256]


@pytest.mark.parametrize("ri", COOP_DRIS)
def test_cooperative_kernel_takes_any_restart_interval(ca, gpu, ri):
    """Single frames of every kind of restart interval go to decode_coop_team_422_kernel (asserted through
    compeg_decoder_last_kernel) and come out bit-exact: smooth content, and busy content with long zero runs
    (ZRL, quirk Q2: some intervals run over their end) -- host preprocessing, device preprocessing (window sized from
    an estimate) and as a one-image batch.  ref: src/huffman.wgsl:118-204 loops over any restart_interval."""
    for (w, h, kind, q, seed) in ((960, 720, 0, 85, 300 + ri), (640, 360, 1, 60, 400 + ri)):
        jpeg = synth.make_jpeg(w, h, seed=seed, kind=kind, quality=q, ri=ri)
        want = orc.ImageData(jpeg).decode()
        for device in (False, True):
            dec = ca.Decoder(gpu)
            dec.set_device_preprocess(device)
            data = ca.ImageData(jpeg)
            dec.decode_blocking(data)
            # (the busy 640x360 frame with an interval of 240 or 256 MCUs, preprocessed on the device: the window the
            # device path estimates for a team's interval -- twice the average -- exceeds what a team can have (32 KB), and
            # the frame takes the streamed batch kernel; every other case is the cooperative kernel's)
            # (the busy 640x360 frame with an interval of 240 or 256 MCUs, preprocessed on the device: the window the
            # device path estimates for a team's interval -- twice the average -- exceeds what a team can have (32 KB), and
            # the frame takes the streamed batch kernel; every other case is the cooperative kernel's)
            expect = "fused_stream" if (kind == 1 and device and ri in (240, 256)) else "coop_team"
            assert dec.last_kernel() == expect, (ri, w, h, device, dec.last_kernel())
            _assert_equal(dec.read_texture(w, h), want)
        batch = ca.Batch(gpu)
        batch.upload([ca.ImageData(jpeg)])
        batch.decode()
        batch.wait()
        assert batch.last_kernel() == "coop_team"
        _assert_equal(batch.read_output(0), want)


def _flip_bits(jpeg, seed, flips=40):
    bad = bytearray(jpeg)
    at = bad.find(b"\xff\xda") + 14
    rng = np.random.default_rng(seed)
    for _ in range(flips):
        pos = int(rng.integers(at, len(bad) - 2))
        if bad[pos] != 0xFF and bad[pos - 1] != 0xFF:
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            if bad[pos] == 0xFF:
                bad[pos] = 0xFE
    return bytes(bad)


@pytest.mark.parametrize("ri", [10, 30, 60, 120, 240])
@pytest.mark.parametrize("content", ["uniform", "mixed", "corrupt"])
def test_walk_route_batches(ca, gpu, ri, content):
    """Launches of few or long restart intervals (no more than a wave of them per SIMD) take the walk + lane-per-MCU
    route: walk_mcus_422_kernel finds where the MCUs begin -- a lane per interval through the walk tables --, then
    decode_fused_422_mcu_rec_kernel decodes with a lane per MCU.  Frames of one stream (flat grids, units' queues),
    images of different sizes (a grid row per image, ragged edges), corrupt scans among them; host and device
    preprocessing.  Bit-exact, kernel identity asserted."""
    if content == "uniform":
        w, h = (3840, 2160) if ri == 240 else (960, 720)
        frames = [synth.make_jpeg(w, h, seed=700 + i + ri, kind=0, quality=85, ri=ri) for i in range(6)]
        jpegs = [frames[i % 6] for i in range(6 if ri == 240 else (48 if ri >= 30 else 40))]
    else:
        shapes = [(1000, 1000), (1016, 990), (936, 1004), (1280, 720), (250, 70), (1921, 1081)]
        jpegs = [synth.make_jpeg(w, h, seed=750 + i + ri, kind=(0, 0, 2)[i % 3], quality=(70, 85, 90)[i % 3], ri=ri)
                 for i, (w, h) in enumerate(shapes * 3)]
        if content == "corrupt":
            for i in (1, 5, 8, 12):
                jpegs[i] = _flip_bits(jpegs[i], seed=ri + i)
    images = [ca.ImageData(j) for j in jpegs]
    batch = ca.Batch(gpu)
    batch.upload(images)
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "walk_mcu", (ri, content, batch.last_kernel())
    check = range(len(jpegs)) if content != "uniform" else (0, 1, 5, len(jpegs) - 1)
    for i in check:
        _assert_equal(batch.read_output(i), orc.ImageData(jpegs[i]).decode())
    # a second decode of the same batch (the records are rewritten), and the scan preprocessed on the device
    batch.decode()
    batch.wait()
    _assert_equal(batch.read_output(len(jpegs) - 1), orc.ImageData(jpegs[-1]).decode())
    dev = ca.Batch(gpu)
    dev.set_device_preprocess(2)
    dev.upload(images)
    dev.decode()
    dev.wait()
    assert dev.last_kernel() == "walk_mcu"
    for i in check:
        assert np.array_equal(dev.read_output(i), batch.read_output(i)), i


def test_walk_route_quirk_q1_inside_long_intervals(ca, gpu):
    """Quirk Q1 on the walk + lane-per-MCU route: restart intervals of a whole MCU row (DRI = 64) with the tiles of
    Q1_TILES inside -- the walk's test at the DC codes finds the reference reader running dry, the MCUs behind it in
    the interval get records that say so, and the second kernel decodes them from zeros.  Both entropy modes."""
    rgb = synth.fill(1024, 256, seed=5, kind=0, noise=0).copy()
    n = 0
    for row in range(0, 256 // 8, 2):
        seeds = Q1_TILES[n % 4]
        x = 64 * (1 + (3 * n) % 13)
        rgb[row * 8:row * 8 + 8, x:x + 64] = synth.fill(64, 8, seed=seeds[(n // 4) % 2], kind=1)
        n += 1
    jpeg = synth.encode(rgb, quality=100, ri=64)
    want = orc.ImageData(jpeg).decode()
    plain = orc.ImageData(jpeg, standard_entropy=True).decode()
    assert (want != plain).any(axis=2).sum() > 100 * n   # (the underflows are there: the rest of those MCU rows differs)
    for standard, expect in ((False, want), (True, plain)):
        batch = ca.Batch(gpu)
        batch.upload([ca.ImageData(jpeg, standard_entropy=standard)] * 72)   # (up to 64 of them are the cooperative kernel's)
        batch.decode()
        batch.wait()
        assert batch.last_kernel() == "walk_mcu"
        for i in (0, 71):
            _assert_equal(batch.read_output(i), expect)


def test_walk_route_chunked_launches_and_short_scans(ca, gpu):
    """The route with launches of part of a batch (set_chunk: the records of every launch's images), with a scan that
    holds fewer restart intervals than its header announces (COUNT_MISMATCH is tolerated: the reference decodes the
    missing ones from the scan's first words), and with an image whose last interval is cut short by the end of the data."""
    frames = [synth.make_jpeg(960, 720, seed=820 + i, kind=0, quality=85, ri=30) for i in range(4)]
    cut = frames[1][:len(frames[1]) * 2 // 3] + b"\xff\xd9"          # (the scan ends early: fewer intervals than announced)
    j = bytearray(frames[2])
    i = j.find(b"\xff\xdd")
    j[i + 4:i + 6] = (45).to_bytes(2, "big")                           # (DRI says 45, the stream has markers every 30 MCUs)
    jpegs = [frames[0], cut, bytes(j), frames[3]] * 6
    batch = ca.Batch(gpu)
    batch.upload([ca.ImageData(x) for x in jpegs])
    batch.set_chunk(5)
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "walk_mcu"
    for i in (0, 1, 2, 3, 22, 23):
        _assert_equal(batch.read_output(i), orc.ImageData(jpegs[i]).decode())


@pytest.mark.parametrize("ri", [4, 10])
def test_cooperative_kernel_launch_size_boundary(ca, gpu, ri):
    """The dispatch boundary: launches of up to 2 x 1024 x 256 data units are the cooperative kernel's -- two 4K
    frames (518 400 data units) are, three are not -- and either side of it is bit-exact; so are launches in which a
    workgroup holds one, two and four teams (small, medium, full launches) and the last team of an image is short."""
    jpegs = [synth.make_jpeg(3840, 2160, seed=500 + i, quality=85, ri=ri) for i in range(3)]
    wants = [orc.ImageData(j).decode() for j in jpegs]
    # (DRI = 10: one frame is 1080 teams for the chip's 1024 places -- a second round of teams that walk a lane per
    # interval -- and the walk + lane-per-MCU route is faster, 86 against 98 us; two frames 93 against 145; with DRI = 4
    # the cooperative kernel's second round still wins; three frames: a lane per interval, or that route)
    for n, kernel in ((1, "coop_team" if ri == 4 else "walk_mcu"), (2, "coop_team" if ri == 4 else "walk_mcu"), (3, "pair" if ri == 4 else "walk_mcu")):
        batch = ca.Batch(gpu)
        batch.upload([ca.ImageData(j) for j in jpegs[:n]])
        batch.decode()
        batch.wait()
        assert batch.last_kernel() == kernel, (n, batch.last_kernel())
        for i in range(n):
            _assert_equal(batch.read_output(i), wants[i])
    # chunked: launches of two frames and a last one of one, all cooperative (walk tables made for the chunk size)
    batch = ca.Batch(gpu)
    batch.upload([ca.ImageData(j) for j in jpegs])
    batch.set_chunk(2)
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "coop_team"
    for i in range(3):
        _assert_equal(batch.read_output(i), wants[i])
    small = synth.make_jpeg(250, 70, seed=77, quality=85, ri=ri)   # ragged, 32 / 4 or 10 intervals: one short team
    want_small = orc.ImageData(small).decode()
    for n in (1, 7, 40):
        batch = ca.Batch(gpu)
        batch.upload([ca.ImageData(small)] * n)
        batch.decode()
        batch.wait()
        assert batch.last_kernel() == "coop_team"
        for i in (0, n - 1):
            _assert_equal(batch.read_output(i), want_small)


def _with_noise_patch(w, h, seed, x0, y0, pw, ph):
    rgb = synth.fill(w, h, seed=seed, kind=0).copy()
    rng = np.random.default_rng(seed)
    rgb[y0:y0 + ph, x0:x0 + pw] = rng.integers(0, 256, (ph, pw, 3), dtype=np.uint8)
    return rgb


@pytest.mark.parametrize("ri", [4, 10, 30])
def test_cooperative_kernel_window_estimate_too_small(ca, gpu, ri):
    """Device preprocessing knows the largest span of 64 consecutive intervals only; a team's window is sized from
    its share of that.  A frame whose entropy sits in one small patch makes that estimate too small for the teams
    of the patch: their walks leave the window and the intervals go to the serial decoder inside the same kernel
    -- bit-exact all the same.  (With host preprocessing the spans are known exactly.)"""
    w, h = 1920, 360
    jpeg = synth.encode(_with_noise_patch(w, h, 900 + ri, 256, 64, 16 * ri, 24), quality=90, ri=ri)
    want = orc.ImageData(jpeg).decode()
    for device in (True, False):
        dec = ca.Decoder(gpu)
        dec.set_device_preprocess(device)
        data = ca.ImageData(jpeg)
        dec.decode_blocking(data)
        if device:
            assert dec.last_kernel() == "coop_team"
        _assert_equal(dec.read_texture(w, h), want)


# Seeds of synth.fill(64, 8, seed, kind=1): as one restart interval of four MCUs at quality 100, the reference's reader
# runs dry at a DC code (quirk Q1) in quarter q of the interval -- MCU q -- and decodes zeros from there on.  Found
# with the emulated kernel's counters (tests/test_kernel_emulation.py checks them); one such interval in a thousand.
Q1_TILES = {0: (2336, 19408), 1: (860, 1469), 2: (1674, 7303), 3: (77, 6532)}


def q1_frame(w, h, every=16):
    """A smooth frame in which every `every`-th restart interval (64 x 8 pixels, DRI = 4) is one of the Q1 tiles."""
    rgb = synth.fill(w, h, seed=5, kind=0, noise=0).copy()
    per_row, n = w // 64, 0
    for i in range(3, per_row * (h // 8), every):
        seeds = Q1_TILES[n % 4]
        y, x = (i // per_row) * 8, (i % per_row) * 64
        rgb[y:y + 8, x:x + 64] = synth.fill(64, 8, seed=seeds[(n // 4) % 2], kind=1)
        n += 1
    return synth.encode(rgb, quality=100, ri=4), n


def test_q1_underflow_in_every_quarter_under_the_walk(ca, gpu):
    """Quirk Q1 inside the cooperative kernel's decoding-under-the-walk (DRI = 4, more than one team per workgroup):
    the wave of quarter q finds the reference reader's underflow, the later quarters' lanes drop what they decoded
    and take the zero-stream levels, the DC sums cross the quarters -- for underflows in each of the four quarters,
    in one frame (272 teams) and in a two-frame launch."""
    jpeg, tiles = q1_frame(2048, 1088)
    assert tiles >= 250
    want = orc.ImageData(jpeg).decode()
    plain = orc.ImageData(jpeg, standard_entropy=True).decode()
    assert (want != plain).any(axis=2).sum() > 20 * tiles          # (the underflows are there: whole MCUs differ)
    dec, data, got = _decode(ca, gpu, jpeg)
    assert dec.last_kernel() == "coop_team"
    _assert_equal(got, want)
    batch = ca.Batch(gpu)
    batch.upload([ca.ImageData(jpeg)] * 2)
    batch.decode()
    batch.wait()
    assert batch.last_kernel() == "coop_team"
    for i in range(2):
        _assert_equal(batch.read_output(i), want)


def test_large_frames_clean_and_corrupt(ca, gpu):
    """A short seed of tools/fuzz_gpu_big.py: 720p to 4K+ frames, random quality / restart interval / content, bit
    flips in every other scan, both entropy modes, host and device preprocessing.  (Its first run found a lost
    segment tail in the device path's threaded staging copy; iteration 18 of this seed is that input.)"""
    from tools import fuzz_gpu_big
    n, bad = fuzz_gpu_big.run(seed=20261004, iters=20, log=lambda *a, **k: None)
    assert n >= 30 and bad == 0
    # ... and with the restart intervals of webcams and row-per-interval encoders
    n, bad = fuzz_gpu_big.run(seed=20261005, iters=12, log=lambda *a, **k: None, dris=(10, 30, 60, 120, 240, 7))
    assert n >= 18 and bad == 0


def test_walk_tables_follow_the_huffman_tables(ca, gpu):
    """The cooperative kernel's walk tables are made once per set of Huffman tables and kept (by a decoder: across
    decodes; by a batch: per image unless all images share their tables).  One decoder, images whose tables
    alternate; then small batches -- which the cooperative kernel takes -- of images with different tables."""
    base = synth.make_jpeg(640, 360, seed=31, quality=85, ri=4)
    others = []
    for j in _hostile_table_variants(10):
        try:
            others.append((j, orc.ImageData(j).decode()))
        except orc.OracleError:
            pass
    assert len(others) >= 4
    want_base = orc.ImageData(base).decode()
    dec = ca.Decoder(gpu)
    for j, want in [(base, want_base), others[0], (base, want_base), others[1], others[2], (base, want_base)]:
        data = ca.ImageData(j)
        dec.decode_blocking(data)
        _assert_equal(dec.read_texture(data.width(), data.height()), want)
    for group in ([(base, want_base), others[0], others[3]], [(base, want_base), (base, want_base)],
                  [others[1], (base, want_base)]):
        batch = ca.Batch(gpu)
        batch.upload([ca.ImageData(j) for j, _ in group])
        batch.decode()
        batch.wait()
        for i, (_, want) in enumerate(group):
            _assert_equal(batch.read_output(i), want)


def test_standard_entropy_extension(ca, gpu):
    """COMPEG_PARSE_STANDARD_ENTROPY (opt-in): bit-exact against the oracle with the same switch, through
    the paired kernel, the throughput kernel and an extension layout; and what the switch is for: the
    output no longer depends on the restart interval."""
    outs = []
    for ri in (1, 4, 8, 0):   # divisors of the 16 200 MCUs: a trailing partial interval is not decoded (lib.rs:785)
        jpeg = synth.make_jpeg(1920, 1080, seed=77, kind=0, quality=85, ri=ri)
        dec = ca.Decoder(gpu)
        dec.decode_blocking(ca.ImageData(jpeg, standard_entropy=True))
        got = dec.read_texture(1920, 1080)
        _assert_equal(got, orc.ImageData(jpeg, standard_entropy=True).decode())
        outs.append(got)
    for other in outs[1:]:
        assert np.array_equal(outs[0], other)
    # throughput kernel (more than 1024 waves) and a 4:2:0 image, both with the switch
    jpegs = [synth.make_jpeg(1000, 1000, seed=210 + i, kind=i % 2, quality=92, ri=3) for i in range(26)]
    batch = ca.Batch(gpu)
    batch.upload([ca.ImageData(j, standard_entropy=True) for j in jpegs])
    batch.decode()
    batch.wait()
    assert sum((2625 + 63) // 64 for _ in jpegs) > 1024
    for i in (0, 1, 25):
        _assert_equal(batch.read_output(i), orc.ImageData(jpegs[i], standard_entropy=True).decode())
    j420 = synth.make_jpeg(640, 360, seed=230, kind=0, quality=95, ri=2, sampling=(2, 2))
    dec = ca.Decoder(gpu)
    dec.decode_blocking(ca.ImageData(j420, allow_sampling=True, standard_entropy=True))
    _assert_equal(dec.read_texture(640, 360), orc.ImageData(j420, allow_sampling=True, standard_entropy=True).decode())


def _full_size_batch(ca, gpu, w, h, n=256, distinct=8, ri=4):
    """n frames of distinct synthetic sources through compeg_batch_*; EVERY output slot is read back and compared
    with the oracle's output of its source (slot i holds source (3 i + i // distinct) % distinct, so that
    neighbouring slots, and slots a workgroup may share, come from different sources)."""
    jpegs = [synth.make_jpeg(w, h, seed=0xC0FFEE + i, kind=0, quality=85, ri=ri) for i in range(distinct)]
    wants = [orc.ImageData(j).decode() for j in jpegs]
    images = [ca.ImageData(j) for j in jpegs]
    src = [(3 * i + i // distinct) % distinct for i in range(n)]
    assert len(set(src)) == distinct
    batch = ca.Batch(gpu)
    batch.upload([images[s] for s in src])
    assert batch.count() == n and batch.pixels() == n * w * h
    batch.decode()
    batch.wait()
    for i, s in enumerate(src):
        got = batch.read_output(i)
        if not np.array_equal(got, wants[s]):
            _assert_equal(got, wants[s])
    # a second decode into the same outputs (what a benchmark step is) changes nothing: first, middle, last slot
    batch.decode()
    batch.wait()
    for i in (0, n // 2 - 1, n - 1):
        _assert_equal(batch.read_output(i), wants[src[i]])
    return batch, images, src, wants


def test_full_size_batch_256_x_1080p(ca, gpu):
    """BASELINE configs[2] at size: a batch of 256 1920x1080 4:2:2 DRI=4 JPEGs in one launch, every slot checked."""
    batch, images, src, wants = _full_size_batch(ca, gpu, 1920, 1080)
    # the same batch with the scan preprocessed by the device kernels inside every decode
    dev = ca.Batch(gpu)
    dev.set_device_preprocess(2)
    dev.upload([images[s] for s in src])
    dev.decode()
    dev.wait()
    assert dev.host_fallbacks() == 0
    for i in range(0, 256, 17):
        _assert_equal(dev.read_output(i), wants[src[i]])


def test_full_size_batch_256_x_4k(ca, gpu):
    """BASELINE configs[3], one GPU's share at size: 256 3840x2160 4:2:2 DRI=4 JPEGs (8.5 GB of RGBA) in one
    launch -- the workload bench.py times -- every slot checked against the oracle output of its source."""
    _full_size_batch(ca, gpu, 3840, 2160)


def test_bounded_gpu_fuzz_seed(ca, gpu):
    """A bounded seed of tools/fuzz_gpu.py (random geometry / quality / DRI / sampling / bit flips / both entropy
    modes; host preprocessor on 1 and 4 threads and the device scan kernels, blocking and non-blocking, plus
    batches): every decode equal to the oracle.  The device-only instruction sequences (inline-asm LDS reads,
    sbfe, the rounding-mode block) are not seen by the sanitised emulation, so they get random inputs here."""
    from tools import fuzz_gpu
    lines = []
    runs, bad, skipped = fuzz_gpu.run(seed=20260, iters=60, budget_s=20.0,
                                      log=lambda *a, **k: lines.append(" ".join(str(x) for x in a)))
    assert bad == 0, "\n".join(lines[-20:])
    assert runs >= 60, (runs, skipped)


def test_bench_multi_rank_path_rehearsed_on_one_gpu(tmp_path):
    """bench.py's multi-rank code path as the driver launches it (torch.distributed.run, one process per rank):
    two fresh ranks share this box's one card, gloo stands in for RCCL (--rehearse-on-one-gpu).  Not a
    measurement -- it proves that sharding, per-rank core shares, barriers, the max-over-ranks reduction, the
    per-rank host-fed pipeline and the single JSON line work with more than one rank."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rehearse-on-one-gpu", "--batch", "6", "--width", "640", "--height", "360", "--cpu-seconds", "0",
           "--no-extra-configs", "--e2e-reps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["verified_bit_exact_vs_oracle"] is True
    e2e = j["end_to_end"]
    assert e2e["verified_bit_exact_vs_oracle"] is True and e2e["from_jpeg_bytes_parse_included"]["whole_job_mpix_s"] > 0
    assert j["config"]["images_per_gpu"] == 6


def test_c_consumer_decodes_on_the_gpu(ca, gpu, tmp_path):
    """The boundary without Python: tests/c_consumer/consumer.c --decode is the reference's own test flow
    (src/tests.rs:41-72: Gpu::open, ImageData::new, Decoder::decode_blocking, texture read-back) as a plain C99 program
    linked against libcompeg_hip.so.  Its FNV-1a hash of the RGBA bytes equals the hash of the oracle's pixels, for the
    reference's MJPEG fixture (DRI = 10, no DHT) and for a 4K frame."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "consumer")
    libdir = os.path.dirname(ca.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_consumer", "consumer.c"), "-o", exe,
                           "-L", libdir, "-l:libcompeg_hip.so", "-Wl,-rpath," + libdir])

    big = tmp_path / "frame4k.jpg"
    big.write_bytes(synth.make_jpeg(3840, 2160, seed=0xC0FFEE, kind=0, quality=85, ri=4))
    for path in (os.path.join(root, "tests", "golden", "parser", "mjpeg.jpg"), str(big)):
        r = subprocess.run([exe, "--decode", path], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = dict(l.split(" ", 1) for l in r.stdout.splitlines())
        want = orc.ImageData(open(path, "rb").read()).decode()
        h, w = want.shape[:2]
        acc = _fnv1a_fast(want.reshape(-1))   # (the same hash of the oracle's pixels)
        assert lines["decoded"] == f"{w} {h} fnv1a {acc:016x}", (path, lines)
        assert lines["changed"].startswith("1 ") and lines["changed_again"] == "0"


@pytest.mark.parametrize("sampling", [(1, 1), (1, 2)])
def test_mcu_pairs_across_mcu_rows(ca, gpu, sampling):
    """4:4:4 / 4:4:0 with an even restart interval (MCUs composited in pairs) and an odd number of MCUs a row -- 1000,
    360, 1080 pixels across: every second MCU row begins with the second MCU of a pair.  That pair's halves go to two
    places (`pair_limits`, `pair_second_offset`); the pairs of such rows lie across two 64-byte segments (ordinary
    stores, `layout_store`).  On the GPU a select around the lane exchange once sent the second halves of half the
    quads' pairs to the wrong place (tests/test_code_objects.py has the scan for it): batches and single frames,
    ragged heights, bit-exact."""
    for (w, h, ri) in ((1000, 600, 4), (360, 642, 2), (1080, 250, 8), (1001, 97, 4), (8, 93, 4), (8, 90, 3)):   # (... and one MCU across)
        jpegs = [synth.make_jpeg(w, h, seed=90 + i, kind=0, quality=85, ri=ri, sampling=sampling) for i in range(3)]
        batch = ca.Batch(gpu)
        batch.upload([ca.ImageData(jpegs[i % 3], allow_sampling=True) for i in range(24)])
        batch.decode()
        batch.wait()
        assert batch.last_kernel() in ("fused_layout", "fused_stream")
        for i in (0, 1, 23):
            _assert_equal(batch.read_output(i), orc.ImageData(jpegs[i % 3], allow_sampling=True).decode())
        dec = ca.Decoder(gpu)
        dec.decode_blocking(ca.ImageData(jpegs[2], allow_sampling=True))
        _assert_equal(dec.read_texture(w, h), orc.ImageData(jpegs[2], allow_sampling=True).decode())


@pytest.mark.parametrize("sampling,single,paired", [((1, 1), "decode_fused_444_single_kernel", "decode_fused_444_kernel"),
                                                     ((1, 2), "decode_fused_440_single_kernel", "decode_fused_440_kernel")])
def test_extension_layouts_with_odd_restart_intervals(ca, gpu, sampling, single, paired):
    """4:4:4 and 4:4:0 with odd restart intervals: the paired kernels composite an interval's MCUs two at a time and its
    last one alone (the second halves store nothing); intervals of one MCU take the single form
    (decode_fused_444_single_kernel / decode_fused_440_single_kernel: 32-byte rows).  The batch reports "fused_layout"
    for all of them; all are bit-exact against the oracle's extension.  Odd intervals were two to three times slower per
    frame than even ones while their 32-byte rows were non-temporal stores -- each a write of its own at the memory; as
    ordinary stores they cost no such factor (kernels_body.h, layout_store; every second interval's pairs lie across two
    64-byte segments)."""
    times = {}
    for ri in (1, 3, 4, 5):
        jpegs = [synth.make_jpeg(1280, 720, seed=60 + i, kind=0, quality=85, ri=ri, sampling=sampling) for i in range(4)]
        images = [ca.ImageData(j, allow_sampling=True) for j in jpegs]
        batch = ca.Batch(gpu)
        batch.upload([images[i % 4] for i in range(64)])
        for _ in range(3):
            batch.decode()
        batch.wait()
        assert batch.last_kernel() == "fused_layout"
        batch.timing(reset=True)
        for _ in range(5):
            batch.decode()
        batch.wait()
        n, total, _, _ = batch.timing(reset=True)
        times[ri] = total / n
        for i in (0, 3, 63):
            _assert_equal(batch.read_output(i), orc.ImageData(jpegs[i % 4], allow_sampling=True).decode())
    assert times[3] < 1.9 * times[4] and times[5] < 1.9 * times[4], (single, paired, times)
